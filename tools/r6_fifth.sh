#!/bin/bash
# round 6, fifth GPU call: the speculative seed kernel's give-up budget (wave-iterations a chunk may take before the dense kernels redo it; 256) on the repeat workloads -- a chunk that is
# given up on has burnt the whole budget first -- and two chunks per wave (SEED_NCH=2) on the human index
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb; mkdir -p $GSA_BENCH_TMP
run() {   # tag, workload, library variant, env assignment, extra bench args
  L=$PWD/gsalign_amd/lib/libgsa_hip_$3.so; [ "$3" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  env $4 GSA_LIB_PATH=$L GSA_BENCH_DETAIL=gpurun_out/r6_fifth_detail_$1.json timeout 900 python bench.py --workload $2 --extra "" --no-cpu-baseline --no-side-legs --no-e2e $5 2>gpurun_out/r6_fifth_$1.err | tail -1 > gpurun_out/r6_fifth_$1.json
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/r6_fifth_$1.json")); print("run $1", d["value"], "Gbp/s", d["ms_per_step"], "ms", d.get("stage_ms_alone"))
except Exception as e:
    print("run $1 FAILED", e); print(open("gpurun_out/r6_fifth_$1.err").read()[-800:])
P
}
for b in 256 128 96 64 48; do run hl_b$b human_like - GSA_SEED_BUDGET=$b ""; done
for b in 256 96 64; do run adv_b$b adversarial - GSA_SEED_BUDGET=$b ""; done
for b in 256 96; do run hum_b$b human - GSA_SEED_BUDGET=$b ""; done
run full_b128 human_full - GSA_SEED_BUDGET=128 "--steps 10 --warmup 2"
run full_nch2 human_full nch2 GSA_X=0 "--steps 10 --warmup 2"
