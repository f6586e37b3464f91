#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb; mkdir -p $GSA_BENCH_TMP
run() {
  L=$PWD/gsalign_amd/lib/libgsa_hip_$3.so; [ "$3" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  env $4 GSA_LIB_PATH=$L GSA_BENCH_DETAIL=gpurun_out/r6v_detail_$1.json timeout 900 python bench.py --workload $2 --extra "" --no-cpu-baseline --no-side-legs --no-e2e $5 2>gpurun_out/r6v_$1.err | tail -1 > gpurun_out/r6v_$1.json
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/r6v_$1.json")); print("run $1", d["value"], "Gbp/s", d["ms_per_step"], "ms", d.get("stage_ms_alone"))
except Exception as e:
    print("run $1 FAILED", e); print(open("gpurun_out/r6v_$1.err").read()[-800:])
P
}
F="--steps 20 --warmup 4"
run warm human_full - GSA_X=0 "$F"
for rep in 1 2; do
  run full_ctx4_$rep human_full - GSA_X=0 "$F"
  run full_ctx6_$rep human_full - GSA_X=0 "$F --inflight 6"
  run full_ctx8_$rep human_full - GSA_X=0 "$F --inflight 8"
  run full_ctx7_$rep human_full - GSA_X=0 "$F --inflight 7"
done
run hum_ctx4 human - GSA_X=0 ""
run hum_ctx6 human - GSA_X=0 "--inflight 6"
run hum_ctx8 human - GSA_X=0 "--inflight 8"
run hl_ctx4 human_like - GSA_X=0 ""
run hl_ctx6 human_like - GSA_X=0 "--inflight 6"
run adv_ctx6 adversarial - GSA_X=0 "--inflight 6"
run yeast_ctx3 yeast - GSA_X=0 ""
run yeast_ctx6 yeast - GSA_X=0 "--inflight 6"
