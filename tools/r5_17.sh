#!/bin/bash
# yeast, one context: SQ counters of the longest dispatch of every kernel (the bundles)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum"; do
  i=$((i+1))
  timeout 500 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_y/s$i -o p -- python bench.py --workload yeast --steps 20 --warmup 2 --inflight 1 --no-e2e --extra "" --no-cpu-baseline --no-side-legs > gpurun_out/pmc_y_s$i.log 2>&1
  find gpurun_out/pmc_y/s$i -name "*counter_collection.csv" | head -1 | xargs -I{} python tools/pmc_big.py {} > gpurun_out/r5_pmc_big_yeast_$i.txt
  rm -rf gpurun_out/pmc_y/s$i
  cut -c1-260 gpurun_out/r5_pmc_big_yeast_$i.txt | head -24
done
