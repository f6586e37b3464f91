#!/bin/bash
# k_seed_select under -sen: where do the PosDiff-bitmap path's 3.3 ms go?  Variants: no LDS table (every hit straight to HBM), one probe, a plain look at the coarse bitmap
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP
show() { python - "$1" "$2" <<'P'
import json, sys; d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"], 2), "Gbp/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["stage_ms_one_context_alone"].items()})
P
}
for v in "" selA selB selC; do
  for w in yeast human; do
    GSA_LIB_PATH=$PWD/gsalign_amd/lib/libgsa_hip${v:+_$v}.so timeout 600 python bench.py --workload $w --extra "" --no-side-legs --no-cpu-baseline --no-e2e > gpurun_out/r5_${w}_sel$v.json 2> gpurun_out/r5_${w}_sel$v.err; show "$w variant=${v:-product}" gpurun_out/r5_${w}_sel$v.json
  done
done
