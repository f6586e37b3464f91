#!/bin/bash
# GSA_SEED_CUS (seed kernels confined to part of the CUs) on the human-sized workload (GPU box):  gpurun -- 'bash tools/seedcus.sh 0 192 128 64'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb
for n in "$@"; do
  GSA_SEED_CUS=$n python bench.py --steps ${SLOTS_STEPS:-24} --warmup 6 --inflight ${CTX:-4} --extra "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('seed CUs $n', round(d['value'],2), 'Gbp/s', round(d['ms_per_step'],3), 'ms/step; seed kernels live', round(d['kernels'][0]['ms_per_step'],2), 'ms; alone', round(d['stage_ms_one_context_alone']['seed_search'],2))"
done
