#!/bin/bash
# Parity of the dense seed kernels (repeat / adversarial cases; every chunk through the sweep with GSA_SEED_MODE=sweep), then the adversarial bench leg
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "repeat or adversarial" 2>&1 | tail -3
GSA_SEED_MODE=sweep timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -x -q -m gpu -k "stage or degenerate or midsize or drop_in" 2>&1 | tail -3
GSA_SEED_MODE=search timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stage or degenerate" 2>&1 | tail -3
WLS="adversarial" HWQS="16" bash tools/r4_bench_x.sh
for v in ${VARIANTS}; do echo "variant $v"; GSA_LIB_PATH=$PWD/gsalign_amd/lib/libgsa_hip_$v.so BARGS="--no-side-legs" WLS="adversarial" HWQS="16" bash tools/r4_bench_x.sh; done
