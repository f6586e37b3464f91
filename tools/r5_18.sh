#!/bin/bash
# the PosDiff byte map (pd_bytes): parity, then yeast with and without it
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pd_byte_map or yeast_sized or sweep_launch or chunk_ranges or align_many_contexts" 2>&1 | tail -5
B="--workload yeast --extra '' --no-cpu-baseline --no-side-legs --no-e2e"
for pb in 1 0; do for inf in 1 3; do
  echo "== pd_bytes $pb inflight $inf"
  eval GSA_PD_BYTES=$pb timeout 600 python bench.py $B --warmup 2 --inflight $inf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('stage_ms_one_context_alone')))"
done; done
