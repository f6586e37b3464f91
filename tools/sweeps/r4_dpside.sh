#!/bin/bash
# dp_side (the striped DP's lower size class beside the upper one): parity with the option on, then bench legs and the one-context timeline with and without
ulimit -c 0
GSA_DP_SIDE=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bundle.py -x -q -m gpu -k "stage or golden or degenerate or midsize or bundle or config or repeat" 2>&1 | tail -3
for s in 0 1; do export GSA_DP_SIDE=$s; echo "dp_side $s"
  BARGS="--no-side-legs" WLS="${WLS:-human adversarial}" HWQS="16" bash tools/r4_bench_x.sh
  WL=human STEPS=12 bash tools/tl1.sh; head -1 gpurun_out/tl1_human.txt | cut -c1-100; grep "k_dp_stripe" gpurun_out/tl1_human.txt | cut -c1-100
done
