#!/bin/bash
# round 5, first GPU call: the whole -m gpu suite (with the new scale-parity tests), then the driver's bench command
cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r5_gputest.log 2>&1
tail -25 gpurun_out/r5_gputest.log
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5_bench_a.json 2> gpurun_out/r5_bench_a.err
tail -c 1500 gpurun_out/r5_bench_a.err
python tools/show_bench.py gpurun_out/r5_bench_a.json 2>/dev/null | head -60
