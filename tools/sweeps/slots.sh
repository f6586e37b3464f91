#!/bin/bash
# GSA_SEED_SLOTS x contexts in flight on the human-sized workload (GPU box):  gpurun -- 'bash tools/slots.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb
CFG=${SLOTS_CFG:-0:4 1:4 2:4 1:3 1:5 1:6 2:6 0:4}
for cfg in $CFG; do
  s=${cfg%%:*}; n=${cfg##*:}
  GSA_SEED_SLOTS=$s python bench.py --steps ${SLOTS_STEPS:-24} --warmup 6 --inflight $n --extra "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slots $s contexts $n', round(d['value'],2), 'Gbp/s', round(d['ms_per_step'],3), 'ms/step  pcie', round(d['pcie_inclusive']['value'],2))"
done
