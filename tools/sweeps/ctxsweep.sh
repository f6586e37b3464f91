#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in ${CFGS:-"human 2" "human 3" "human 4" "human 5"}; do set -- $cfg
  v=$(python bench.py --workload $1 --inflight $2 --steps ${STEPS:-24} --warmup 6 --extra "" --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.3f Gbp/s %.3f ms/step pcie %.3f roofline %.3f" % (d["value"], d["ms_per_step"], d["pcie_inclusive"]["value"], d["roofline"]["frac"]))')
  echo "$1 inflight $2: $v"
done
