#!/bin/bash
# yeast, three contexts, steady state: who shares the chip with whom (last 9 bundles)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
B="--workload yeast --extra '' --no-cpu-baseline --no-side-legs --no-e2e"
eval timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/prof_y3 -o p -- python bench.py $B --steps 120 --warmup 2 > gpurun_out/prof_y3.log 2>&1
python tools/timeline_share.py gpurun_out/prof_y3/p_results.db 9 > gpurun_out/r5_timeline_share_yeast.txt 2>&1
python tools/contig_phases.py gpurun_out/prof_y3/p_results.db > gpurun_out/r5_contig_phases_yeast.txt 2>&1
rm -rf gpurun_out/prof_y3
head -45 gpurun_out/r5_timeline_share_yeast.txt
cat gpurun_out/r5_contig_phases_yeast.txt | head -20
for inf in 2 3 4 6; do
  echo "== inflight $inf"
  eval timeout 600 python bench.py $B --steps 120 --warmup 2 --inflight $inf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
