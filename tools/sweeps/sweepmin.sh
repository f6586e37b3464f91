#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in "1024 40" "1 40" "1 20" "1 10" "1 5"; do set -- $cfg
  v=$(GSA_SWEEP_MIN=$1 GSA_SWEEP_SEG=$2 python bench.py --workload human --inflight 1 --steps 12 --warmup 4 --extra "" --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.3f Gbp/s %.3f ms/step stages %s" % (d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["stage_ms_one_context_alone"].items()}))')
  echo "GSA_SWEEP_MIN=$1 GSA_SWEEP_SEG=$2: $v"
done
