#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for o in 0 1; do for n in 1 3; do
  v=$(GSA_DP_ORDER=$o python bench.py --workload human --inflight $n --steps 24 --warmup 6 --extra "" --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.3f Gbp/s %.3f ms/step pcie %.3f  stages %s" % (d["value"], d["ms_per_step"], d["pcie_inclusive"]["value"], {k: round(v, 2) for k, v in d["stage_ms_one_context_alone"].items()}))')
  echo "GSA_DP_ORDER=$o inflight $n: $v"
done; done
