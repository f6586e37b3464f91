#!/bin/bash
# Seed kernel's persistent workgroups per CU (LDS left to the other contexts' kernels) against the step with four contexts:
#   WLS="human human_full" PS="12 10 8" bash tools/r4_persist.sh      (needs the -DGSA_EXPERIMENTS library variant x)
ulimit -c 0
export GSA_LIB_PATH=$PWD/gsalign_amd/lib/libgsa_hip_x.so
for w in ${WLS:-human}; do for p in ${PS:-12 10 8 6}; do echo "persist $p"; GSA_SEED_PERSIST=$p BARGS="--no-side-legs" WLS="$w" HWQS="16" bash tools/r4_bench_x.sh; done; done
