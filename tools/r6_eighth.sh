#!/bin/bash
# round 6, eighth GPU call: the whole -m gpu suite on the tree as it stands (timed: the driver allows 1 200 s), then the driver's bench command
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r6_gputest_full.txt 2>&1; tail -25 gpurun_out/r6_gputest_full.txt
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6_bench_driver_cmd.txt 2> gpurun_out/r6_bench_driver_cmd.err; tail -c 3000 gpurun_out/r6_bench_driver_cmd.txt
cp gpurun_out/bench_detail.json gpurun_out/r6_bench_driver_cmd_detail.json
python - <<'P'
import json
d=json.load(open("gpurun_out/r6_bench_driver_cmd_detail.json"))
e=d.get("end_to_end", {}); e.pop("note", None); e.pop("command", None); print(json.dumps(e))
P
