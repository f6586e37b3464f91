#!/bin/bash
# round 6, first GPU call: the -m gpu suite and the driver's bench command on the tree as it stands (compact last line)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r6_gputest.txt 2>&1; tail -5 gpurun_out/r6_gputest.txt
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6_bench_first.txt 2> gpurun_out/r6_bench_first.err; tail -c 3000 gpurun_out/r6_bench_first.txt; tail -5 gpurun_out/r6_bench_first.err
cp gpurun_out/bench_detail.json gpurun_out/r6_bench_first_detail.json
