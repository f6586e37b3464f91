#!/bin/bash
# A/B of seed-kernel variants on the GPU box: rebuild k_seed.o with the given -D flags, then (1) un-profiled stage time from
# tools/seed_probe.py, (2) SQ instruction / cycle counters of k_seed_wg from one rocprofv3 --pmc pass.
#   gpurun -- 'bash tools/seedx2.sh "-DSEED_FLAT=1" "-DSEED_FAST_MAX=3 -DSEED_SLOW_MAX=2"'  -> gpurun_out/seedx2.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_PROBE_KEEP=/tmp/seedprobe_keep
N=${SEEDX_N:-100000000}; V=${SEEDX_V:-both}
: > gpurun_out/seedx2.txt
for x in "$@"; do
  rm -f gsalign_amd/csrc/build/k_seed.o; make -C gsalign_amd/csrc -j32 lib EXTRA="$x" > /tmp/mk.log 2>&1 || tail -5 /tmp/mk.log
  echo "=== $x" >> gpurun_out/seedx2.txt
  python tools/seed_probe.py $N $V 2>&1 | grep -E "^==|seed stats" >> gpurun_out/seedx2.txt
  rm -rf /tmp/sx; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/sx -o p -- python tools/seed_probe.py $N $V > /tmp/sx.log 2>&1
  python - >> gpurun_out/seedx2.txt <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for fn in glob.glob('/tmp/sx/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_seed_wg" in k or "k_dense" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, d in acc.items():
    print("   ", k[:40], "launches", n[k], " ".join(f"{c[3:]}={v / max(1, n[k]):.4g}" for c, v in sorted(d.items())))
PY
done
cat gpurun_out/seedx2.txt
