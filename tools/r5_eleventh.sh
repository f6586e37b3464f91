#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP
for w in human human_full; do
  rocprofv3 --kernel-trace -d gpurun_out/tlm_$w -o t -- python bench.py --workload $w --steps $([ $w = human ] && echo 60 || echo 5) --warmup 2 --extra "" --no-cpu-baseline --no-side-legs --no-e2e > gpurun_out/tlm_$w.log 2>&1
  python tools/timeline_share.py gpurun_out/tlm_$w/t_results.db $([ $w = human ] && echo 120 || echo 96) > gpurun_out/r5_timeline_share_$w.txt 2>&1; python tools/contig_phases.py gpurun_out/tlm_$w/t_results.db 96 > gpurun_out/r5_contig_phases_$w.txt 2>&1; cat gpurun_out/r5_contig_phases_$w.txt
  rm -rf gpurun_out/tlm_$w
done
