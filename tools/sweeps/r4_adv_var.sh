#!/bin/bash
# adversarial bench leg (main leg only) per library variant: VARIANTS="a b" bash tools/r4_adv_var.sh   ("-" = the default library)
ulimit -c 0
for v in ${VARIANTS}; do echo "variant $v"; L=$PWD/gsalign_amd/lib/libgsa_hip_$v.so; [ "$v" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  GSA_LIB_PATH=$L BARGS="--no-side-legs" WLS="adversarial" HWQS="16" bash tools/r4_bench_x.sh; done
