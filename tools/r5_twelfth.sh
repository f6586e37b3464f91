#!/bin/bash
# round 5, twelfth GPU call: stream priorities (GSA_CREATE_PRIO modes 0..3) -- parity subset under mode 1, then the workloads under every mode
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
( GSA_PRIO=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bundle.py -m gpu -x -q -k "golden or complex or scaled or adversarial or human_like or bundle or many" ) > gpurun_out/r5_gputest12.log 2>&1; tail -3 gpurun_out/r5_gputest12.log
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP
show() { python - "$1" "$2" <<'P'
import json, sys; d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), "Gbp/s", round(d["ms_per_step"], 2), "ms", "latency", round(d["one_contig_latency"]["ms"], 2))
P
}
for p in 0 1 2 3; do
  GSA_PRIO=$p timeout 900 python bench.py --workload human_full --extra "" --steps 10 --warmup 3 --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_hf_prio$p.json 2> gpurun_out/r5_hf_prio$p.err; show "human_full prio=$p" gpurun_out/r5_hf_prio$p.json
  for w in human adversarial human_like ecoli yeast; do
    GSA_PRIO=$p timeout 600 python bench.py --workload $w --extra "" --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_${w}_prio$p.json 2> gpurun_out/r5_${w}_prio$p.err; show "$w prio=$p" gpurun_out/r5_${w}_prio$p.json
  done
done
