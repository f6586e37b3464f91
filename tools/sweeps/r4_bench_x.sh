# bench legs of chosen workloads with chosen --hwq: WLS="human yeast" HWQS="8 16" bash tools/r4_bench_x.sh
export GSA_BENCH_TMP=/tmp/gsa_sweep GSA_BENCH_KEEP=1
mkdir -p gpurun_out $GSA_BENCH_TMP
for w in ${WLS:-human}; do for q in ${HWQS:-16}; do
  python bench.py --workload $w --extra "" --no-cpu-baseline --hwq $q ${BARGS} > gpurun_out/bx.json 2> gpurun_out/bx.err || tail -5 gpurun_out/bx.err
  python - "$w" "$q" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/bx.json").read().strip().splitlines()[-1])
g=lambda k:(d.get(k) or {}).get("value") or 0
print("%s hwq %s: value %.3f (%.2f ms/step)  resident %.3f  no_prefetch %.3f  bundled %.3f  ratio %.3f  latency %.2f ms  stage %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], g("resident"), g("no_prefetch"), g("bundled"), d.get("h2d_inclusive_over_resident") or 0, d["one_contig_latency"]["ms"], {k: round(v,2) for k,v in d["stage_ms_one_context_alone"].items()}))
PY
done; done | tee -a gpurun_out/r4_bench_x.txt
