# kernel-trace summary of one bench workload (value leg only): W=yeast bash tools/r4_trace_w.sh
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_sweep GSA_BENCH_KEEP=1
for w in ${W:-yeast}; do
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$w -o p -- python bench.py --workload $w --extra "" --no-cpu-baseline --no-side-legs ${BARGS} > gpurun_out/prof_$w.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_$w/p_results.db 45 > gpurun_out/kernels_$w.txt
rm -rf gpurun_out/prof_$w
tail -1 gpurun_out/prof_$w.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', d['value'], d['ms_per_step'], d['steps'])"
done
