#!/bin/bash
# dp_side again (the striped DP's lower size classes beside the upper one) now that the fused passes are shorter
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
for w in ${WLS:-human human_full}; do for ds in 0 1; do
  echo "== $w dp_side=$ds"
  GSA_DP_SIDE=$ds timeout 400 python bench.py --workload $w --extra '' --no-cpu-baseline --no-side-legs --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('stage_ms_one_context_alone')), (d.get('one_contig_latency') or {}).get('ms'))"
done; done
