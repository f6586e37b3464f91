#!/bin/bash
# (1) the default bench line on the final build (PMC traffic of this build); (2) experiment: k_seed_select with 2 / 4 windows of 256 hits per round (SEL_BATCH)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 200 gpurun_out/bench_default.json; echo
GSA_LIB_PATH=$PWD/gsalign_amd/lib/libgsa_hip_sb4.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "yeast_sized or align_many_contexts or pd_byte_map or config1 or repeat or chunk_ranges or sweep_launch" 2>&1 | tail -2
for v in - sb2 sb4; do
  L=$PWD/gsalign_amd/lib/libgsa_hip_$v.so; [ "$v" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  for w in yeast human; do
    echo "== variant $v $w (one context)"
    GSA_LIB_PATH=$L timeout 300 python bench.py --workload $w --extra '' --no-cpu-baseline --no-side-legs --no-e2e --inflight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d.get('stage_ms_one_context_alone')))"
  done
done
