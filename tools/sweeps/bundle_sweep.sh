#!/bin/bash
# Bundle size / contexts sweep on the small-contig workloads:  gpurun -- 'bash tools/bundle_sweep.sh'
cd "$GRAFT_REPO_ROOT"
for w in ecoli yeast; do
  for t in ${TARGETS:-0 6000000 11000000 16000000 21000000 31000000}; do
    for n in 1 2 3 4; do
      v=$(GSA_BUNDLE_CAP=64000000 GSA_BUNDLE_TARGET=$t python bench.py --workload $w --inflight $n --steps ${STEPS:-48} --warmup 6 --extra "" --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.3f Gbp/s %.3f ms/step pcie %.3f" % (d["value"], d["ms_per_step"], d["pcie_inclusive"]["value"]))')
      echo "$w target $t inflight $n: $v"
    done
  done
done
