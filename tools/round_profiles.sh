#!/bin/bash
# Round-end measurements on the GPU box: default bench line, rocprofv3 kernel-trace summaries of the three workloads, PMC passes.
#   gpurun -- 'bash tools/round_profiles.sh'   ->  gpurun_out/{bench_default.json, kernels_<w>.txt, pmc_human/*.csv}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/bench_default.json
for w in human ecoli yeast; do
  timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$w -o p -- python bench.py --workload $w --steps 20 --warmup 5 --extra "" --no-cpu-baseline > gpurun_out/prof_$w.log 2>&1
  python tools/rocprof_summary.py gpurun_out/prof_$w/p_results.db > gpurun_out/kernels_$w.txt
  rm -rf gpurun_out/prof_$w
done
bash tools/pmc_top.sh human > gpurun_out/pmc_top.log 2>&1
bash tools/pmc_sq.sh human > gpurun_out/pmc_sq.log 2>&1
# then, back in the container: python tools/pmc_top.py human ; python tools/pmc_sq.py human > profiles/r03_sq_human.txt
