ulimit -c 0
timeout 1200 python -m pytest tests/test_gpu_cli.py tests/test_gpu_bundle.py tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" 2>&1 | tail -3
BARGS="--no-side-legs" WLS="human yeast ecoli" HWQS="16" bash tools/r4_bench_x.sh
WL="human yeast" STEPS=12 bash tools/tl1.sh
