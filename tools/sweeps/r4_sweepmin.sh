ulimit -c 0
export GSA_LIB_PATH=$PWD/gsalign_amd/lib/libgsa_hip_x.so
for m in 1024 1; do echo "sweep_min $m"; GSA_SWEEP_MIN=$m BARGS="--no-side-legs" WLS="human" HWQS="16" bash tools/r4_bench_x.sh; done
