#!/bin/bash
# HBM traffic of the dominant kernel (k_seed_wg) from rocprofv3 PMC passes on the default bench workload.
# Separate passes (TCC slots), --pmc only together with --kernel-trace.  Output: gpurun_out/pmc/*.csv,
# then tools/pmc_seed.py turns them into profiles/archive/r01_pmc_seed.json.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc/$tag -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/$tag.log 2>&1
  find gpurun_out/pmc/$tag -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pmc/$tag.csv
done
ls -la gpurun_out/pmc/
