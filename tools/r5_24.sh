#!/bin/bash
# what the results' way home (D2H of records and string pools) costs the step: GSA_SKIP_D2H 0 / 1 (pools stay) / 2 (records too) on the experiments build -- timing only
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
L=$PWD/gsalign_amd/lib/libgsa_hip_ex.so
for w in ${WLS:-human human_full}; do for sk in 0 1 2; do
  echo "== $w GSA_SKIP_D2H=$sk"
  GSA_LIB_PATH=$L GSA_SKIP_D2H=$sk timeout 400 python bench.py --workload $w --extra '' --no-cpu-baseline --no-side-legs --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
