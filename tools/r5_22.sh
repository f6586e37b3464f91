#!/bin/bash
# fused look-back passes: predecessors looked at per round trip (LB_WIN 16 against 64) -- is a pass bound by the pace of the inclusive-prefix frontier?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
for v in ${VARIANTS:-- w16}; do
  L=$PWD/gsalign_amd/lib/libgsa_hip_$v.so; [ "$v" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  for w in ${WLS:-yeast human}; do
    for inf in ${INFS:-1}; do
      echo "== variant $v workload $w inflight $inf (0 = the workload's own)"
      x=""; [ "$inf" != "0" ] && x="--inflight $inf"
      GSA_LIB_PATH=$L timeout 150 python bench.py --workload $w --extra '' --no-cpu-baseline --no-side-legs --no-e2e $x 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('stage_ms_one_context_alone')))"
    done
  done
done
