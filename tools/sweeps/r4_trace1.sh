cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_sweep GSA_BENCH_KEEP=1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_h -o p -- python bench.py --workload human --steps 40 --warmup 6 --extra "" --no-cpu-baseline --no-side-legs > gpurun_out/prof_h.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_h/p_results.db 40 > gpurun_out/kernels_human_upl.txt
rm -rf gpurun_out/prof_h
head -30 gpurun_out/kernels_human_upl.txt | cut -c1-150
