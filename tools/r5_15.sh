#!/bin/bash
# multi-rank plumbing with the real aligner on the one-GPU box: two ranks share GPU 0 (gloo), contig shard + staged gather, and the one-contig chunk-range split
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP
timeout 900 python bench.py --gpus 2 --backend gloo --same-gpu --workload yeast --steps 4 --warmup 1 --extra "" 2>&1 | tail -2 | cut -c1-600
timeout 900 python bench.py --gpus 2 --backend gloo --same-gpu --workload human --split --steps 4 --warmup 1 --extra "" 2>&1 | tail -2 | cut -c1-600
