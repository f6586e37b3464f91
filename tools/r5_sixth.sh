#!/bin/bash
# round 5, sixth GPU call: the whole suite on the current code (two-level bitmap scan at human scale, page-locked CLI buffers), fixtures with the
# large-input paths forced, the striped-DP launches of the human_like workload
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/r5_gputest6.log 2>&1; tail -16 gpurun_out/r5_gputest6.log
( GSA_WALK_CHAIN_MIN=0 GSA_PD_TWO_LEVEL_MIN=0 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bundle.py -m gpu -x -q -k "golden or drop_in or complex or degenerate or bundle" ) > gpurun_out/r5_gputest6b.log 2>&1; tail -3 gpurun_out/r5_gputest6b.log
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP
timeout 600 python tools/dp_hist.py human_like 2>&1 | tail -4
rocprofv3 --kernel-trace -d gpurun_out/tl1_hl -o t -- python bench.py --workload human_like --inflight 1 --steps 2 --warmup 1 --extra "" --no-cpu-baseline --no-side-legs --no-e2e > gpurun_out/tl1_hl.log 2>&1
python tools/stripe_launches.py gpurun_out/tl1_hl/t_results.db 12 > gpurun_out/r5_stripe_launches_human_like.txt 2>&1; cat gpurun_out/r5_stripe_launches_human_like.txt
rm -rf gpurun_out/tl1_hl
timeout 900 python bench.py --workload human_full --extra "" --steps 10 --warmup 3 --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_hf_6.json 2> gpurun_out/r5_hf_6.err
python - <<'P'
import json; d = json.loads(open("gpurun_out/r5_hf_6.json").read().strip().splitlines()[-1])
print("human_full", round(d["value"], 2), "Gbp/s", round(d["ms_per_step"], 2), "ms", d["stage_ms_one_context_alone"])
P
