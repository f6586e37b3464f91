ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bundle.py -x -q -m gpu -k "stage or golden or degenerate or midsize or bundle or ecoli or yeast" 2>&1 | tail -3
BARGS="" WLS="ecoli yeast human" HWQS="16" bash tools/r4_bench_x.sh
