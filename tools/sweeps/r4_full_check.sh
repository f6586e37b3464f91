ulimit -c 0
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
BARGS="" WLS="human yeast ecoli adversarial" HWQS="16" bash tools/r4_bench_x.sh
