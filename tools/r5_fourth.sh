#!/bin/bash
# round 5, fourth GPU call: presence table from the k-mer table (test + gsa_create time in end_to_end), human_like one-context timeline,
# SEED_MULTI=4 variant on the human index, pinned-memory probe
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_kmer or golden" ) > gpurun_out/r5_gputest4.log 2>&1; tail -3 gpurun_out/r5_gputest4.log
python tools/pin_probe.py 4 > gpurun_out/r5_pin_probe.txt 2>&1; cat gpurun_out/r5_pin_probe.txt
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP
( time timeout 900 python bench.py --workload human_full --extra "" --steps 10 --warmup 3 --no-side-legs ) > gpurun_out/r5_hf_base.json 2> gpurun_out/r5_hf_base.err
python - <<'P'
import json; d = json.loads(open("gpurun_out/r5_hf_base.json").read().strip().splitlines()[-1])
print("base human_full", round(d["value"], 2), "Gbp/s", round(d["ms_per_step"], 2), "ms", d["stage_ms_one_context_alone"], d.get("cpu_baseline", {}).get("parity_sample")); print(json.dumps(d.get("end_to_end"))[:900])
P
( time GSA_LIB_PATH=$PWD/gsalign_amd/lib/libgsa_hip_multi.so timeout 900 python bench.py --workload human_full --extra "" --steps 10 --warmup 3 --no-side-legs --no-e2e ) > gpurun_out/r5_hf_multi.json 2> gpurun_out/r5_hf_multi.err
python - <<'P'
import json; d = json.loads(open("gpurun_out/r5_hf_multi.json").read().strip().splitlines()[-1])
print("SEED_MULTI=4 human_full", round(d["value"], 2), "Gbp/s", round(d["ms_per_step"], 2), "ms", d["stage_ms_one_context_alone"], d.get("cpu_baseline", {}).get("parity_sample"))
P
rocprofv3 --kernel-trace -d gpurun_out/tl1_human_like -o t -- python bench.py --workload human_like --inflight 1 --steps 4 --warmup 2 --extra "" --no-cpu-baseline --no-side-legs --no-e2e > gpurun_out/tl1_human_like.log 2>&1
python tools/timeline.py gpurun_out/tl1_human_like/t_results.db v > gpurun_out/tl1_human_like.txt 2>&1
python tools/rocprof_summary.py gpurun_out/tl1_human_like/t_results.db 40 > gpurun_out/tl1_human_like_kernels.txt 2>&1
rm -rf gpurun_out/tl1_human_like
head -30 gpurun_out/tl1_human_like_kernels.txt | cut -c1-160
