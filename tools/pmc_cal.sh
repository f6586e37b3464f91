#!/bin/bash
# Calibration of the TCC request counters on gfx950 (round-3 verdict, hygiene item): tools/rand_probe.hip `cal` issues 16.8 M random reads of
# 16 / 32 / 64 / 128 bytes (one launch each) on an 18 GB table; the counters per launch say what a read of each width costs.
#   gpurun -- 'bash tools/pmc_cal.sh'  ->  gpurun_out/pmc_cal.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/rand_probe tools/rand_probe.hip || exit 1
mkdir -p gpurun_out/pmc_cal
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "TCC_REQ_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_cal/$tag -o p -- /tmp/rand_probe cal > gpurun_out/pmc_cal/$tag.log 2>&1
  find gpurun_out/pmc_cal/$tag -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} gpurun_out/pmc_cal/$tag.csv
  rm -rf gpurun_out/pmc_cal/$tag
done
python - <<'PY' | tee gpurun_out/pmc_cal.txt
import csv, glob, collections
READS = 4096 * 256 * 16
acc = collections.defaultdict(dict)
for fn in glob.glob("gpurun_out/pmc_cal/*.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "k_rand" not in k: continue
        w = k.split("<")[1].split(">")[0] if "<" in k else k
        acc[w][r["Counter_Name"]] = acc[w].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(f"# gfx950, {READS} random reads per launch on an 18 GB table (tools/rand_probe.hip cal); counters per READ")
for w in sorted(acc, key=lambda x: int(x) if x.isdigit() else 0):
    print(f"read of {w:>3} B:", "  ".join(f"{c} {v / READS:.3f}" for c, v in sorted(acc[w].items())))
PY
