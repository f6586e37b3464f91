#!/bin/bash
# round 6, fourth GPU call (the third ran out of disk after a core dump): (1) the human_full step with two- / one-wave workgroups in the fused passes (LB_TPB = 128 / 64) and with
# 8 / 24 / 32 hardware queues (20 streams share 16 today), (2) the CLI end to end, (3) the new whole-genome / human-like / -sen 50 Mb parity tests, timed
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
df -h /tmp | tail -1
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb; mkdir -p $GSA_BENCH_TMP
run() {   # tag, library variant, extra bench args
  L=$PWD/gsalign_amd/lib/libgsa_hip_$2.so; [ "$2" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  GSA_LIB_PATH=$L GSA_BENCH_DETAIL=gpurun_out/r6_fourth_detail_$1.json timeout 900 python bench.py --steps 10 --warmup 2 --extra "" --no-cpu-baseline --no-side-legs $3 2>gpurun_out/r6_fourth_$1.err | tail -1 > gpurun_out/r6_fourth_$1.json
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/r6_fourth_$1.json")); print("run $1", d["value"], "Gbp/s", d["ms_per_step"], "ms", d.get("stage_ms_alone"), d.get("end_to_end"))
except Exception as e:
    print("run $1 FAILED", e); print(open("gpurun_out/r6_fourth_$1.err").read()[-800:])
P
}
run base - ""
run lbt128 lbt128 "--no-e2e"
run lbt64 lbt64 "--no-e2e"
run hwq8 - "--no-e2e --hwq 8"
run hwq24 - "--no-e2e --hwq 24"
run hwq32 - "--no-e2e --hwq 32"
python - <<'P'
import json
d=json.load(open("gpurun_out/r6_fourth_detail_base.json"))
e=d.get("end_to_end", {}); e.pop("note", None); print(json.dumps(e))
P
rm -rf /tmp/gb; df -h /tmp | tail -1
( time timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s --durations=8 -k "config5 or (full_size and human_like) or (scaled_pairs and 50000000)" ) > gpurun_out/r6_fourth_tests.txt 2>&1; grep -v "^contig \|^  contig\|^pass " gpurun_out/r6_fourth_tests.txt | tail -40 | cut -c1-400
