#!/bin/bash
# main bench leg of chosen workloads per library variant: VARIANTS="- g3 g4" WLS="human yeast" bash tools/r4_var_bench.sh   ("-" = the default library)
ulimit -c 0
for v in ${VARIANTS}; do echo "variant $v"; L=$PWD/gsalign_amd/lib/libgsa_hip_$v.so; [ "$v" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  GSA_LIB_PATH=$L BARGS="--no-side-legs" WLS="${WLS:-human}" HWQS="16" bash tools/r4_bench_x.sh; done
