#!/bin/bash
# round 6, seventh GPU call: gsa_reserve_index + page-locked index uploads: parity subset, then the CLI end to end at human scale three times (spread)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -x -q -k "reserved or pac_bytes or clone_to_device or cli or stages_vs_golden or drop_in" ) > gpurun_out/r6_seventh_tests.txt 2>&1; tail -6 gpurun_out/r6_seventh_tests.txt
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb; mkdir -p $GSA_BENCH_TMP
for k in 1 2 3; do
GSA_BENCH_DETAIL=gpurun_out/r6_seventh_detail_$k.json timeout 900 python bench.py --steps 10 --warmup 2 --extra "" --no-cpu-baseline --no-side-legs 2>gpurun_out/r6_seventh_$k.err | tail -1 > gpurun_out/r6_seventh_$k.json
python - <<P
import json
d=json.load(open("gpurun_out/r6_seventh_$k.json")); print(d["value"], d["ms_per_step"], d.get("end_to_end"))
d=json.load(open("gpurun_out/r6_seventh_detail_$k.json"))
e=d.get("end_to_end", {}); e.pop("note", None); e.pop("command", None); print(json.dumps(e))
P
done
