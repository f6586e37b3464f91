#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for o in 0 1; do for cfg in "human 1" "human 4" "ecoli 2"; do set -- $cfg
  v=$(GSA_DP_AFTER_EARLY=$o python bench.py --workload $1 --inflight $2 --steps 24 --warmup 6 --extra "" --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.3f Gbp/s %.3f ms/step pcie %.3f  extend alone %.2f" % (d["value"], d["ms_per_step"], d["pcie_inclusive"]["value"], d["stage_ms_one_context_alone"]["extend"]))')
  echo "GSA_DP_AFTER_EARLY=$o $1 inflight $2: $v"
done; done
