#!/bin/bash
# round 6, sixth GPU call: GSA_CREATE_REF_PAC (the device unpacks .pac; the CLI unpacks RefSequence beside gsa_create): parity + CLI goldens + the end-to-end timing at human scale
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_bundle.py -m gpu -x -q -k "pac_bytes or clone_to_device or cli or bundle or stages_vs_golden" ) > gpurun_out/r6_sixth_tests.txt 2>&1; tail -6 gpurun_out/r6_sixth_tests.txt
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb; mkdir -p $GSA_BENCH_TMP
GSA_BENCH_DETAIL=gpurun_out/r6_sixth_detail.json timeout 900 python bench.py --steps 10 --warmup 2 --extra "" --no-cpu-baseline --no-side-legs 2>gpurun_out/r6_sixth.err | tail -1 > gpurun_out/r6_sixth.json
python - <<'P'
import json
d=json.load(open("gpurun_out/r6_sixth.json")); print(d["value"], d["ms_per_step"], d.get("end_to_end"), d.get("roofline"))
d=json.load(open("gpurun_out/r6_sixth_detail.json"))
e=d.get("end_to_end", {}); e.pop("note", None); print(json.dumps(e))
P
# the end-to-end program twice more (spread across runs)
for k in 1 2; do python - <<'P'
import json, os, subprocess, sys
sys.path.insert(0, os.getcwd())
import bench
from gsalign_amd import hostlib
d=json.load(open("gpurun_out/r6_sixth_detail.json"))
cmd=d["end_to_end"]["command"].split()
cmd[0]=hostlib.CLI_PATH
# (the query FASTA was removed by bench.py: write it again from the same generator)
P
done
