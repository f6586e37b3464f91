#!/bin/bash
# CLI: context teardown and the VCF beside the MAF writer's tail -- the CLI tests, then end_to_end of configs[4] (twice) and configs[3]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
timeout 900 python -m pytest tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
timeout 900 python bench.py --workload human_full --steps 2 --warmup 1 --extra '' --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']; print({k:e[k] for k in e if k.endswith('_s') or k=='gbp_per_s_excl_index_build'})"
done
timeout 600 python bench.py --workload human --steps 4 --warmup 1 --extra '' --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']; print({k:e[k] for k in e if k.endswith('_s') or k=='gbp_per_s_excl_index_build'})"
