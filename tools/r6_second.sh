#!/bin/bash
# round 6, second GPU call: parity of the fat-workgroup seed kernel (SEED_WPW waves per workgroup, one workgroup per CU), the new tests (clone_to_device, two ranks on one GPU),
# then the human_full step per library variant (SEED_WPW = 1 = round 5's launch shape, 6 = default, 8, 10), and the CLI's -timing with the allocation statistics
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_two_ranks.py -m gpu -x -q -k "not config5" ) > gpurun_out/r6_second_tests.txt 2>&1; tail -8 gpurun_out/r6_second_tests.txt
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb; mkdir -p $GSA_BENCH_TMP
for v in wpw1 - wpw8 wpw10; do
  L=$PWD/gsalign_amd/lib/libgsa_hip_$v.so; [ "$v" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  X="--no-e2e"; [ "$v" = "-" ] && X=""
  GSA_LIB_PATH=$L GSA_BENCH_DETAIL=gpurun_out/r6_second_detail_$v.json timeout 900 python bench.py --steps 10 --warmup 2 --extra "" --no-cpu-baseline --no-side-legs $X 2>gpurun_out/r6_second_$v.err | tail -1 > gpurun_out/r6_second_$v.json
  python - <<P
import json
d=json.load(open("gpurun_out/r6_second_$v.json"))
print("variant $v", d["value"], "Gbp/s", d["ms_per_step"], "ms", d.get("stage_ms_alone"), d.get("end_to_end"))
P
done
python - <<'P'
import json
d=json.load(open("gpurun_out/r6_second_detail_-.json"))
print(json.dumps(d.get("end_to_end")))
P
