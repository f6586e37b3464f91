#!/bin/bash
# Round-end measurements on the GPU box: default bench line, rocprofv3 kernel-trace summaries of the workloads, one-context timelines, PMC passes.
#   gpurun -- 'bash tools/round_profiles.sh'   ->  gpurun_out/{bench_default.json, kernels_<w>.txt, tl1_<w>.txt, pmc_<w>/*.csv, pmc_sq_human/*.csv}
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cp gpurun_out/bench_detail.json gpurun_out/bench_default_detail.json
tail -c 400 gpurun_out/bench_default.json; echo
for w in ${KW:-human_full human human_like ecoli yeast adversarial}; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$w -o p -- python bench.py --workload $w --extra "" --no-cpu-baseline --no-side-legs --no-e2e > gpurun_out/prof_$w.log 2>&1
  python tools/rocprof_summary.py gpurun_out/prof_$w/p_results.db > gpurun_out/kernels_$w.txt
  rm -rf gpurun_out/prof_$w
done
WL="human human_like ecoli yeast adversarial" STEPS=8 bash tools/sweeps/tl1.sh
STEPS=1 bash tools/pmc_top.sh human_full > gpurun_out/pmc_top_full.log 2>&1
bash tools/pmc_top.sh human > gpurun_out/pmc_top.log 2>&1
bash tools/pmc_sq.sh human > gpurun_out/pmc_sq.log 2>&1
STEPS=1 bash tools/pmc_sq.sh human_full > gpurun_out/pmc_sq_full.log 2>&1
# then, back in the container: python tools/pmc_top.py human_full; python tools/pmc_top.py human ; python tools/pmc_sq.py human > profiles/r06_sq_human.txt ; python tools/pmc_sq.py human_full > profiles/r06_sq_human_full.txt
# the configs[4] test with its output (whole-genome parity, the time of gsa_clone_to_device on the human index)
( time timeout 1100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k config5 ) > gpurun_out/config5_test.txt 2>&1; grep -v "^contig \|^  contig" gpurun_out/config5_test.txt | tail -30
