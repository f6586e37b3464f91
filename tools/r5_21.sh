#!/bin/bash
# fused look-back passes: is the tile ticket (one same-address device-scope atomic per tile) what a pass costs?  st = tiles by workgroup number (no ticket), t512 = 512-thread tiles
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
for v in ${VARIANTS:-- st t512 st512}; do
  L=$PWD/gsalign_amd/lib/libgsa_hip_$v.so; [ "$v" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  echo "== variant $v parity subset"
  GSA_LIB_PATH=$L timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "yeast_sized or align_many_contexts or pd_byte_map" 2>&1 | tail -2
  for w in ${WLS:-yeast human}; do
    for inf in ${INFS:-1 0}; do
      echo "== variant $v workload $w inflight $inf (0 = the workload's own)"
      x=""; [ "$inf" != "0" ] && x="--inflight $inf"
      GSA_LIB_PATH=$L timeout 600 python bench.py --workload $w --extra '' --no-cpu-baseline --no-side-legs --no-e2e $x 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('stage_ms_one_context_alone')))"
    done
  done
done
