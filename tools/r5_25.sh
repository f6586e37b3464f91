#!/bin/bash
# wave slots: seed kernel 12 one-wave workgroups per CU + striped DP up to 8 four-wave workgroups = a full CU (32 waves); do the fused passes of the other contexts starve for slots?
# dp_occupancy (striped workgroups per CU, LDS padding) x GSA_SEED_PERSIST (seed workgroups per CU; experiments build)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
L=$PWD/gsalign_amd/lib/libgsa_hip_ex.so
for w in ${WLS:-human_full}; do for cfg in ${CFGS:-0:12 4:12 3:12 4:8 0:8 5:10}; do
  occ=${cfg%%:*}; per=${cfg##*:}
  echo "== $w dp_occupancy=$occ seed_persist=$per"
  GSA_LIB_PATH=$L GSA_DP_OCCUPANCY=$occ GSA_SEED_PERSIST=$per timeout 400 python bench.py --workload $w --extra '' --no-cpu-baseline --no-side-legs --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
