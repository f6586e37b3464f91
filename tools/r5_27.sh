#!/bin/bash
# fused passes with all loads of a thread's elements issued together (clamped Ops): parity subset, then stage times
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${K:-not config5 and not full_size and not two_devices}" 2>&1 | tail -4
for w in ${WLS:-yeast human}; do for inf in ${INFS:-1 0}; do
  echo "== $w inflight $inf"
  x=""; [ "$inf" != "0" ] && x="--inflight $inf"
  timeout 400 python bench.py --workload $w --extra '' --no-cpu-baseline --no-side-legs --no-e2e $x 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('stage_ms_one_context_alone')))"
done; done
