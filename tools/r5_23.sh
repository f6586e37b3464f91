#!/bin/bash
# shader clock and power while the bench runs: one context against the workload's own four (is the chip throttling when the contexts overlap?)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
for inf in 1 4; do
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/r5_clocks_$inf.txt &
  P=$!
  timeout 300 python bench.py --workload human --extra '' --no-cpu-baseline --no-side-legs --no-e2e --inflight $inf --steps 400 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight', $inf, d['value'], d['ms_per_step'])"
  kill $P
  sort gpurun_out/r5_clocks_$inf.txt | uniq -c | sort -rn | head -8 | cut -c1-300
done
