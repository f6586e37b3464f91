cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1; do
[ $v = 1 ] && export GSA_SJ_DEV=1
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl -o tl -- python bench.py --workload human --inflight 1 --steps 3 --warmup 1 --extra "" --no-cpu-baseline > gpurun_out/tl.log 2>&1; python tools/timeline.py gpurun_out/tl/tl_results.db v > gpurun_out/timeline_human_1ctx_$v.txt; rm -rf gpurun_out/tl
done
