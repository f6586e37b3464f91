#!/bin/bash
# round 5, seventh GPU call: early striped DP only for blocks that are likely to survive the redundancy filter -- parity, then human_like / human / adversarial
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "not config5 and not full_size" ) > gpurun_out/r5_gputest7.log 2>&1; tail -5 gpurun_out/r5_gputest7.log
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP
for w in human_like human adversarial; do
  timeout 900 python bench.py --workload $w --extra "" --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_${w}_7.json 2> gpurun_out/r5_${w}_7.err
  python - $w <<'P'
import json, sys; d = json.loads(open(f"gpurun_out/r5_{sys.argv[1]}_7.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), "Gbp/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_one_context_alone"].items()}, "latency", round(d["one_contig_latency"]["ms"], 2))
P
done
python - <<'P'
import sys, time; sys.path.insert(0, ".")
import argparse, bench, tempfile
from gsalign_amd import capi
wl = dict(bench.WORKLOADS["human_like"]); args = argparse.Namespace(fasta_ref="", fasta_query="")
px, idx, refs = bench.build_reference("/tmp/gsa_round", "human_like", wl, 0, 1, args)
q = bench.make_queries(wl, refs, args)[0][0]
g = capi.Aligner(idx); g.align_contig(q); g.align_contig(q); st = g.seed_stats()
print("human_like: large gaps launched early", int(st[6]), " large jobs in the late launch", int(st[7]))
P
