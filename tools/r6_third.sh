#!/bin/bash
# round 6, third GPU call: (1) the human_full step with smaller-footprint fused passes (LB_MIN_WAVES=5: <= 96 VGPRs; LB_TPB=128 / 64: two- / one-wave workgroups) against the default,
# (2) the CLI end to end with the buffer list (what a context's 28 GB are) and the parallel result copies, (3) the new whole-genome / human-like / -sen 50 Mb parity tests, timed
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb; mkdir -p $GSA_BENCH_TMP
for v in - lbw5 lbt128 lbt64; do
  L=$PWD/gsalign_amd/lib/libgsa_hip_$v.so; [ "$v" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  X="--no-e2e"; [ "$v" = "-" ] && X=""
  GSA_DUMP_BUFFERS=1 GSA_LIB_PATH=$L GSA_BENCH_DETAIL=gpurun_out/r6_third_detail_$v.json timeout 900 python bench.py --steps 10 --warmup 2 --extra "" --no-cpu-baseline --no-side-legs $X 2>gpurun_out/r6_third_$v.err | tail -1 > gpurun_out/r6_third_$v.json
  python - <<P
import json
d=json.load(open("gpurun_out/r6_third_$v.json"))
print("variant $v", d["value"], "Gbp/s", d["ms_per_step"], "ms", d.get("stage_ms_alone"), d.get("end_to_end"))
P
done
python - <<'P'
import json
d=json.load(open("gpurun_out/r6_third_detail_-.json"))
e=d.get("end_to_end", {}); e.pop("note", None); print(json.dumps(e))
P
grep -A26 "gsa_debug_buffers" gpurun_out/e2e_human_full_stderr.txt | head -60
( time timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s --durations=8 -k "config5 or (full_size and human_like) or (scaled_pairs and 50000000)" ) > gpurun_out/r6_third_tests.txt 2>&1; grep -v "^contig \|^  contig\|^pass " gpurun_out/r6_third_tests.txt | tail -40
