#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "not config5" ) > gpurun_out/r5_gputest13.log 2>&1; tail -5 gpurun_out/r5_gputest13.log
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP
show() { python - "$1" "$2" <<'P'
import json, sys; d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), "Gbp/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_one_context_alone"].items()}, "latency", round(d["one_contig_latency"]["ms"], 2))
P
}
timeout 900 python bench.py --workload human_full --extra "" --steps 10 --warmup 3 --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_hf_13.json 2> gpurun_out/r5_hf_13.err; show human_full gpurun_out/r5_hf_13.json
for w in human ecoli yeast adversarial human_like; do timeout 600 python bench.py --workload $w --extra "" --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_${w}_13.json 2> gpurun_out/r5_${w}_13.err; show $w gpurun_out/r5_${w}_13.json; done
