#!/bin/bash
# fused look-back passes: persistent workgroups per CU (LB_GRID_PER_CU 1 / 2 / 4 / 8) on yeast (alone, three contexts) and human / human_full (four contexts)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
for v in ${VARIANTS:-g1 - g4 g8}; do
  L=$PWD/gsalign_amd/lib/libgsa_hip_$v.so; [ "$v" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  for w in ${WLS:-yeast human}; do
    for inf in ${INFS:-1 0}; do
      echo "== variant $v workload $w inflight $inf (0 = the workload's own)"
      x=""; [ "$inf" != "0" ] && x="--inflight $inf"
      GSA_LIB_PATH=$L timeout 600 python bench.py --workload $w --extra '' --no-cpu-baseline --no-side-legs --no-e2e $x 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('stage_ms_one_context_alone')))"
    done
  done
done
