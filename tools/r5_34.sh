#!/bin/bash
# CLI: .bwt / .sa used where they lie in a private file mapping instead of read into buffers -- end_to_end of configs[4] with (default) and without (GSA_HOST_NO_MMAP=1)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_BENCH_TMP=/tmp/gsa_round GSA_BENCH_KEEP=1; mkdir -p $GSA_BENCH_TMP gpurun_out
for nm in 0 1 0 1; do
echo "== GSA_HOST_NO_MMAP=$nm"
GSA_HOST_NO_MMAP=$nm timeout 400 python bench.py --workload human_full --steps 1 --warmup 1 --extra '' --no-cpu-baseline --no-side-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['end_to_end']; print({k:e[k] for k in e if k in ('total_s','index_load_s','gsa_create_s','output_drain_after_align_s','gbp_per_s_excl_index_build','wall_s')})"
done
