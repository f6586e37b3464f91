#!/bin/bash
# -sen (yeast) through the sweep: parity of the -sen cases with every chunk swept, then the yeast bench leg per seed mode / segment length / shape
ulimit -c 0
GSA_SEED_MODE=sweep timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_bundle.py -x -q -m gpu -k "sen or yeast or bundle" 2>&1 | tail -3
export GSA_LIB_PATH=$PWD/gsalign_amd/lib/libgsa_hip_x.so
echo "default (k_dense_search)"; BARGS="--no-side-legs" WLS="yeast" HWQS="16" bash tools/r4_bench_x.sh
for cfg in ${CFGS:-"1 40" "1 80" "0 160" "0 40"}; do set -- $cfg; echo "sweep shape $1 seg $2"
  GSA_SEED_MODE=sweep GSA_SWEEP_SHAPE=$1 GSA_SWEEP_SEG=$2 BARGS="--no-side-legs" WLS="yeast" HWQS="16" bash tools/r4_bench_x.sh; done
