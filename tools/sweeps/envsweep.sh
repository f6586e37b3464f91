#!/bin/bash
# bench.py (human workload) under a list of environment settings:  gpurun -- 'bash tools/envsweep.sh "A=1" "GSA_DP_LANE=1024" ...'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb
for e in "$@"; do
  env $e python bench.py --workload ${WL:-human} --steps ${SLOTS_STEPS:-24} --warmup 6 --extra "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', round(d['value'],2), 'Gbp/s', round(d['ms_per_step'],3), 'ms/step; extend alone', round(d['stage_ms_one_context_alone']['extend'],2), 'seed alone', round(d['stage_ms_one_context_alone']['seed_search'],2))"
done
