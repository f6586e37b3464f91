# like tools/seedx.sh with an environment prefix per variant:  bash tools/seedx3.sh "ENV=1|-DFLAG" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb
for x in "$@"; do
e="${x%%|*}"; f="${x#*|}"
rm -f gsalign_amd/csrc/build/k_seed.o; make -C gsalign_amd/csrc -j32 lib EXTRA="$f" > /tmp/mk.log 2>&1 || tail -5 /tmp/mk.log
env $e python bench.py --steps 16 --warmup 4 --extra "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$x', round(d['value'],2), round(d['ms_per_step'],3), 'seed alone', round(d['stage_ms_one_context_alone']['seed_search'],3), 'occ_read', d['counters_per_step']['occ_blocks_read'])"
done
