#!/bin/bash
# window-bucket table of 1.25 na entries (any capacity) + sums zeroed per window: parity subset, then the workloads
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not config5 and not full_size and not two_devices" 2>&1 | tail -4
for w in ${WLS:-human human_full adversarial yeast}; do
  echo "== $w"
  timeout 400 python bench.py --workload $w --extra '' --no-cpu-baseline --no-side-legs --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('stage_ms_one_context_alone')))"
done
