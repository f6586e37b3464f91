#!/bin/bash
# last call of round 5: the default bench line (reads the PMC traffic of this build), then the flake hunt -- its fingerprints must equal profiles/r05_stress.txt (same inputs, earlier build)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 300 gpurun_out/bench_default.json; echo
( for a in "human 0 6" "human_like 0 4" "adversarial 0 4" "ecoli 0 60" "yeast 0 12"; do timeout 600 python tools/stress_consistency.py $a 2>&1 | tail -3; done ) > gpurun_out/r5_stress_final.txt 2>&1; cut -c1-400 gpurun_out/r5_stress_final.txt
