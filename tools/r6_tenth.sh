#!/bin/bash
# round 6, tenth GPU call: the early hand-over rule of the speculative seed kernel (a long match rejected for its frequency hands the chunk to the dense kernels at once): parity, then A/B
# against a build without it (SEED_NO_EARLY_HANDOVER) on the repeat workloads and the headline workload
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bundle.py -m gpu -x -q -k "stages_vs or human_like_repeats or adversarial_repeats or repeat_stress or sweep_launch or drop_in or midsize or bundle or long_kmer or (scaled_pairs and not 50000000) or (full_size and not human_like)" ) > gpurun_out/r6_tenth_tests.txt 2>&1; tail -6 gpurun_out/r6_tenth_tests.txt
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb; mkdir -p $GSA_BENCH_TMP
run() {   # tag, workload, library variant, extra args
  L=$PWD/gsalign_amd/lib/libgsa_hip_$3.so; [ "$3" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  GSA_LIB_PATH=$L GSA_BENCH_DETAIL=gpurun_out/r6_tenth_detail_$1.json timeout 900 python bench.py --workload $2 --extra "" --no-cpu-baseline --no-side-legs --no-e2e $4 2>gpurun_out/r6_tenth_$1.err | tail -1 > gpurun_out/r6_tenth_$1.json
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/r6_tenth_$1.json")); print("run $1", d["value"], "Gbp/s", d["ms_per_step"], "ms", d.get("stage_ms_alone"))
except Exception as e:
    print("run $1 FAILED", e); print(open("gpurun_out/r6_tenth_$1.err").read()[-800:])
P
}
run hl_noeh human_like noeh ""
run hl_eh human_like - ""
run adv_noeh adversarial noeh ""
run adv_eh adversarial - ""
run hum_noeh human noeh ""
run hum_eh human - ""
run hl_eh2 human_like - ""
run hl_noeh2 human_like noeh ""
run full_eh human_full - "--steps 10 --warmup 2"
run full_noeh human_full noeh "--steps 10 --warmup 2"
