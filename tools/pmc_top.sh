#!/bin/bash
# HBM traffic of the top kernels of one bench.py workload from rocprofv3 PMC passes (GPU box).
# Separate passes (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2), --pmc only together with --kernel-trace.
#   tools/pmc_top.sh human   ->  gpurun_out/pmc_human/*.csv ; then tools/pmc_top.py human -> profiles/r06_pmc_human.json
W=${1:-human}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_$W
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 500 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$W/$tag -o p -- python bench.py --workload $W --steps ${STEPS:-4} --warmup 1 ${INFLIGHT:+--inflight $INFLIGHT} --no-e2e --extra "" --no-cpu-baseline --no-side-legs > gpurun_out/pmc_$W/$tag.log 2>&1
  find gpurun_out/pmc_$W/$tag -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_reduce.py {} gpurun_out/pmc_$W/$tag.csv
  rm -rf gpurun_out/pmc_$W/$tag
done
ls -la gpurun_out/pmc_$W/
