#!/bin/bash
# One-context kernel timelines (last pass of the run) of the small-contig workloads:  gpurun -- 'bash tools/tl1.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for w in ${WL:-ecoli yeast}; do
  rocprofv3 --kernel-trace -d gpurun_out/tl1_$w -o t -- python bench.py --workload $w --inflight 1 --steps ${STEPS:-20} --warmup 5 --extra "" --no-cpu-baseline --no-side-legs --no-e2e > gpurun_out/tl1_$w.log 2>&1
  python tools/timeline.py gpurun_out/tl1_$w/t_results.db v > gpurun_out/tl1_$w.txt 2>&1
  python tools/rocprof_summary.py gpurun_out/tl1_$w/t_results.db 40 > gpurun_out/tl1_${w}_kernels.txt 2>&1
  [ -n "$KEEP" ] || rm -rf gpurun_out/tl1_$w
done
