#!/bin/bash
# SQ / LDS counters of the top kernels (GPU box): issue utilisation, wait states, LDS bank conflicts.
#   tools/pmc_sq.sh human -> gpurun_out/pmc_sq_human/*.csv ; tools/pmc_sq.py human -> profiles/r06_sq_human.txt
W=${1:-human}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_sq_$W
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 500 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_sq_$W/s$i -o p -- python bench.py --workload $W --steps ${STEPS:-3} --warmup 1 ${INFLIGHT:+--inflight $INFLIGHT} --no-e2e --extra "" --no-cpu-baseline --no-side-legs > gpurun_out/pmc_sq_$W/s$i.log 2>&1
  find gpurun_out/pmc_sq_$W/s$i -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_reduce.py {} gpurun_out/pmc_sq_$W/s$i.csv
  rm -rf gpurun_out/pmc_sq_$W/s$i
done
ls -la gpurun_out/pmc_sq_$W/
