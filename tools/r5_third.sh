#!/bin/bash
# round 5, third GPU call: the native-index oracle parity after the presence-table fix, CLI / bundle tests with the new host pipeline, the driver's bench command
cd "$GRAFT_REPO_ROOT"
( time timeout 1200 python tests/human_scale_check.py ) > gpurun_out/r5_human_check.log 2>&1
tail -25 gpurun_out/r5_human_check.log
( time timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_bundle.py -m gpu -x -q ) > gpurun_out/r5_gputest3.log 2>&1
tail -6 gpurun_out/r5_gputest3.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r5_bench_b.json 2> gpurun_out/r5_bench_b.err
tail -c 800 gpurun_out/r5_bench_b.err
python tools/show_bench.py gpurun_out/r5_bench_b.json 2>/dev/null | cut -c1-1500
