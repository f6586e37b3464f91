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
F="--steps 10 --warmup 2"
for v in - lw512 lw768 -; do run full_$v human_full $v GSA_X=0 "$F"; done
( time timeout 1200 python bench.py ) > gpurun_out/r6v_bench_default.txt 2> gpurun_out/r6v_bench_default.err; tail -c 1500 gpurun_out/r6v_bench_default.txt; cp gpurun_out/bench_detail.json gpurun_out/r6v_bench_default_detail.json
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=5 ) > gpurun_out/r6v_gputest_full.txt 2>&1; tail -14 gpurun_out/r6v_gputest_full.txt
