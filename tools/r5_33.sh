#!/bin/bash
# the very last call of round 5: whole GPU suite and the default bench line on the final tree (the CLI's tail changed after the previous one)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) > gpurun_out/r5_gputest_final.log 2>&1; tail -12 gpurun_out/r5_gputest_final.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 200 gpurun_out/bench_default.json; echo
