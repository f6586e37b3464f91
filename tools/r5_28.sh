#!/bin/bash
# clamped Ops everywhere: parity subset on the product build, then the workloads with the loads of 4 (product) / 2 / 8 elements of a thread in flight together
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not config5 and not full_size and not two_devices" 2>&1 | tail -4
for v in ${VARIANTS:-- b2 b8}; do
  L=$PWD/gsalign_amd/lib/libgsa_hip_$v.so; [ "$v" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  for w in ${WLS:-human yeast human_full}; do
    echo "== variant $v $w"
    GSA_LIB_PATH=$L timeout 400 python bench.py --workload $w --extra '' --no-cpu-baseline --no-side-legs --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('stage_ms_one_context_alone')))"
  done
done
