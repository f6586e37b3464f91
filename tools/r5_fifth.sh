#!/bin/bash
# round 5, fifth GPU call: DP job histogram of the human_like workload, contexts per GPU on the human index (4 / 5 / 6), end_to_end with page-locked query sequences
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP
timeout 600 python tools/dp_hist.py human_like > gpurun_out/r5_dp_hist_human_like.txt 2>&1; cat gpurun_out/r5_dp_hist_human_like.txt
for n in 4 6 8; do
  timeout 900 python bench.py --workload human_full --extra "" --steps 10 --warmup 3 --no-side-legs --no-cpu-baseline $( [ $n = 4 ] || echo --no-e2e ) --inflight $n > gpurun_out/r5_hf_inflight$n.json 2> gpurun_out/r5_hf_inflight$n.err
  python - $n <<'P'
import json, sys; d = json.loads(open(f"gpurun_out/r5_hf_inflight{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("inflight", sys.argv[1], "human_full", round(d["value"], 2), "Gbp/s", round(d["ms_per_step"], 2), "ms"); 
if "end_to_end" in d: print(json.dumps(d["end_to_end"])[:800])
P
done
