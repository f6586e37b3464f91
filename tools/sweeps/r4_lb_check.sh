#!/bin/bash
# parity of the stages behind the fused passes (goldens, oracle cases, bundles), then the human one-context timeline (durations of the k_lb_pass launches)
# per library variant: VARIANTS="- i8 i16" ("-" = the default library)
ulimit -c 0
[ -n "$NOTEST" ] || timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bundle.py tests/test_gpu_cli.py -x -q -m gpu -k "${K:-not full_size}" 2>&1 | tail -3
for v in ${VARIANTS:--}; do
  L=$PWD/gsalign_amd/lib/libgsa_hip_$v.so; [ "$v" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  export GSA_LIB_PATH=$L
  WL="human" STEPS=12 bash tools/tl1.sh
  echo "variant $v: $(grep 'k_lb_pass' gpurun_out/tl1_human.txt | awk '{printf "%s ", $3}')"
  grep "k_lb_pass" gpurun_out/tl1_human.txt | awk '{s+=$3} END {print "  sum of the fused passes:", s, "us"}'
  BARGS="--no-side-legs" WLS="human" HWQS="16" bash tools/r4_bench_x.sh
done
