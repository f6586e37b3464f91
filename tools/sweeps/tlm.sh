#!/bin/bash
# Multi-context kernel timelines of the small-contig workloads:  gpurun -- 'bash tools/tlm.sh'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for w in ecoli yeast; do
  for n in 1 2 3 4; do
    rocprofv3 --kernel-trace -d gpurun_out/tlm_${w}_$n -o t -- python bench.py --workload $w --inflight $n --steps 24 --warmup 6 --extra "" --no-cpu-baseline > gpurun_out/tlm_${w}_$n.log 2>&1
    python tools/timeline_multi.py gpurun_out/tlm_${w}_$n/t_results.db 3000 $([ $n = 3 ] && echo dump) > gpurun_out/tlm_${w}_$n.txt 2>&1
    rm -rf gpurun_out/tlm_${w}_$n
    grep -o '"value": [0-9.]*' gpurun_out/tlm_${w}_$n.log | head -1
  done
done
