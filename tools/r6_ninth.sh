#!/bin/bash
# round 6, ninth GPU call: the pre-classifier of repeat-heavy chunks (k_chunk_preclass, option seed_preclass): parity with it forced on, then what it buys on the repeat workloads
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time GSA_SEED_PRECLASS=2 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bundle.py -m gpu -x -q -k "stages_vs or human_like_repeats or adversarial_repeats or repeat_stress or sweep_launch or drop_in or midsize or config3 or bundle or (scaled_pairs and not 50000000)" ) > gpurun_out/r6_ninth_tests.txt 2>&1; tail -6 gpurun_out/r6_ninth_tests.txt
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb; mkdir -p $GSA_BENCH_TMP
run() {   # tag, workload, env assignment
  env $3 GSA_BENCH_DETAIL=gpurun_out/r6_ninth_detail_$1.json timeout 900 python bench.py --workload $2 --extra "" --no-cpu-baseline --no-side-legs --no-e2e 2>gpurun_out/r6_ninth_$1.err | tail -1 > gpurun_out/r6_ninth_$1.json
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/r6_ninth_$1.json")); print("run $1", d["value"], "Gbp/s", d["ms_per_step"], "ms", d.get("stage_ms_alone"))
except Exception as e:
    print("run $1 FAILED", e); print(open("gpurun_out/r6_ninth_$1.err").read()[-800:])
P
}
run hl_pre0 human_like GSA_SEED_PRECLASS=0
run hl_pre1 human_like GSA_SEED_PRECLASS=1
run hl_pre2 human_like GSA_SEED_PRECLASS=2
run hl_pre1_min1 human_like "GSA_SEED_PRECLASS=1 GSA_SEED_PRECLASS_MIN=1"
run hl_pre1_min4 human_like "GSA_SEED_PRECLASS=1 GSA_SEED_PRECLASS_MIN=4"
run adv_pre0 adversarial GSA_SEED_PRECLASS=0
run adv_pre1 adversarial GSA_SEED_PRECLASS=1
run adv_pre1_min1 adversarial "GSA_SEED_PRECLASS=1 GSA_SEED_PRECLASS_MIN=1"
run hum_pre0 human GSA_SEED_PRECLASS=0
run hum_pre2 human GSA_SEED_PRECLASS=2
