ulimit -c 0
export GSA_LIB_PATH=$PWD/gsalign_amd/lib/libgsa_hip_x.so
for cfg in "0 0" "1 0" "1 1" "2 1"; do set -- $cfg; export GSA_STREAM_PRIO=$1 GSA_DP_SIDE=$2; echo "prio $1 dp_side $2"
  BARGS="--no-side-legs" WLS="human adversarial" HWQS="16" bash tools/r4_bench_x.sh; done
