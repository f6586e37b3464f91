#!/bin/bash
# Counters of the adversarial workload's kernels (k_dense_sweep first): SQ issue/wait sets and the TCC request counts.
#   gpurun -- 'bash tools/r4_adv_pmc.sh'  ->  gpurun_out/pmc_sq_adversarial/*.csv, gpurun_out/pmc_adversarial/*.csv
ulimit -c 0
STEPS=3 bash tools/pmc_sq.sh adversarial
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
W=adversarial
mkdir -p gpurun_out/pmc_$W
for set in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$W/$tag -o p -- python bench.py --workload $W --steps 3 --warmup 1 --inflight 1 --extra "" --no-cpu-baseline --no-side-legs > gpurun_out/pmc_$W/$tag.log 2>&1
  find gpurun_out/pmc_$W/$tag -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_reduce.py {} gpurun_out/pmc_$W/$tag.csv
  rm -rf gpurun_out/pmc_$W/$tag
done
ls -la gpurun_out/pmc_$W/
