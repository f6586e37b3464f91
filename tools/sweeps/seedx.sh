# seed-kernel tunables on the human-sized workload (GPU box): rebuild with the given -D flags and print the seed stage time
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb
for x in "$@"; do
rm -f gsalign_amd/csrc/build/k_seed.o; make -C gsalign_amd/csrc -j32 lib EXTRA="$x" > /tmp/mk.log 2>&1 || tail -5 /tmp/mk.log
python bench.py --steps 16 --warmup 4 --extra "" --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$x', round(d['value'],2), round(d['ms_per_step'],3), 'seed', round(d['stage_ms_one_context_alone']['seed_search'],3))"
done
