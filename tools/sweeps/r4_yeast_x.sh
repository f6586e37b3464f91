#!/bin/bash
# -sen parity (default path: bundles of dense chunks through the sweep, short contigs through k_dense_search) and the yeast legs
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bundle.py tests/test_gpu_cli.py -x -q -m gpu -k "sen or yeast or bundle or stage" 2>&1 | tail -3
BARGS="" WLS="yeast" HWQS="16" bash tools/r4_bench_x.sh
GSA_SEED_MODE=search BARGS="--no-side-legs" WLS="yeast" HWQS="16" bash tools/r4_bench_x.sh
