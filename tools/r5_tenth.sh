#!/bin/bash
# round 5, tenth GPU call: fourth DP size class + host register test (suite), flake hunt with the new hand-off code, dp_occupancy experiment
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "not config5" ) > gpurun_out/r5_gputest10.log 2>&1; tail -5 gpurun_out/r5_gputest10.log
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP
show() { python - "$1" "$2" <<'P'
import json, sys; d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), "Gbp/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_one_context_alone"].items()}, "latency", round(d["one_contig_latency"]["ms"], 2))
P
}
for occ in 0 7 6; do
  for w in human human_like; do
    GSA_DP_OCCUPANCY=$occ timeout 600 python bench.py --workload $w --extra "" --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_${w}_occ$occ.json 2> gpurun_out/r5_${w}_occ$occ.err; show "$w dp_occupancy=$occ" gpurun_out/r5_${w}_occ$occ.json
  done
done
for occ in 0 7; do
  GSA_DP_OCCUPANCY=$occ timeout 900 python bench.py --workload human_full --extra "" --steps 10 --warmup 3 --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_hf_occ$occ.json 2> gpurun_out/r5_hf_occ$occ.err; show "human_full dp_occupancy=$occ" gpurun_out/r5_hf_occ$occ.json
done
( for a in "human 0 12" "human_like 0 8" "adversarial 0 6" "ecoli 0 150" "yeast 0 30"; do timeout 600 python tools/stress_consistency.py $a 2>&1 | tail -3; done ) > gpurun_out/r5_stress.txt 2>&1; cat gpurun_out/r5_stress.txt
