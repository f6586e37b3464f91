#!/bin/bash
# round 5, eighth GPU call: early-DP rule with the 1/16 clause (human_like), the striped launches it leaves, yeast (-sen) with and without the PosDiff bitmap
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "human_like or adversarial or scaled or golden or config3" ) > gpurun_out/r5_gputest8.log 2>&1; tail -3 gpurun_out/r5_gputest8.log
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP
show() { python - "$1" "$2" <<'P'
import json, sys; d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), "Gbp/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_one_context_alone"].items()}, "latency", round(d["one_contig_latency"]["ms"], 2))
P
}
timeout 900 python bench.py --workload human_like --extra "" --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_human_like_8.json 2> gpurun_out/r5_human_like_8.err; show human_like gpurun_out/r5_human_like_8.json
rocprofv3 --kernel-trace -d gpurun_out/tl1_hl -o t -- python bench.py --workload human_like --inflight 1 --steps 2 --warmup 1 --extra "" --no-cpu-baseline --no-side-legs --no-e2e > gpurun_out/tl1_hl.log 2>&1
python tools/stripe_launches.py gpurun_out/tl1_hl/t_results.db 6 > gpurun_out/r5_stripe_launches_human_like_b.txt 2>&1; cat gpurun_out/r5_stripe_launches_human_like_b.txt
python tools/timeline.py gpurun_out/tl1_hl/t_results.db v > gpurun_out/tl1_human_like_b.txt 2>&1
rm -rf gpurun_out/tl1_hl
timeout 600 python bench.py --workload yeast --extra "" --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_yeast_8.json 2> gpurun_out/r5_yeast_8.err; show yeast gpurun_out/r5_yeast_8.json
GSA_PD_BITMAP=0 timeout 600 python bench.py --workload yeast --extra "" --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_yeast_8s.json 2> gpurun_out/r5_yeast_8s.err; show yeast_pdsort gpurun_out/r5_yeast_8s.json
