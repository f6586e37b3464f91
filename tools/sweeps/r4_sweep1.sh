# round 4: what the upload costs -- hardware queues x contexts in flight, H2D-inclusive (value) vs resident vs no prefetch
export GSA_BENCH_TMP=/tmp/gsa_sweep GSA_BENCH_KEEP=1
mkdir -p gpurun_out $GSA_BENCH_TMP
IFS=","; for cfg in ${SWEEP:-8 4,16 4,8 5,16 6,4 4}; do IFS=" "
  set -- $cfg
  python bench.py --workload ${WL:-human} --extra "" --no-cpu-baseline --steps ${STEPS:-40} --warmup 6 --hwq $1 --inflight $2 > gpurun_out/sw.json 2> gpurun_out/sw.err
  python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1])
print("hwq %s inflight %s: value %.2f  resident %.2f  no_prefetch %.2f  ratio %.3f" % (sys.argv[1], sys.argv[2], d["value"], d["resident"]["value"], d["no_prefetch"]["value"], d["h2d_inclusive_over_resident"]))
PY
done | tee gpurun_out/r4_sweep1.txt
