#!/bin/bash
# Counter sets (one rocprofv3 --pmc pass each) for the seed kernels of tools/seed_probe.py (every pass under a timeout: a call with the
# sets "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_SPI_BUSY" / TCC_* / TA_* never came back and cost 15 GPU-minutes;
# the SQ_* and TCC_EA0_* sets of tools/pmc_sq.sh / pmc_top.sh are the ones known to work):  gpurun -- 'bash tools/pmc_probe.sh "SET A" "SET B" ...'
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_PROBE_KEEP=/tmp/seedprobe_keep
N=${SEEDX_N:-100000000}; V=${SEEDX_V:-both}
mkdir -p gpurun_out; : > gpurun_out/pmc_probe.txt
python tools/seed_probe.py $N $V > /dev/null 2>&1      # (builds the index once)
for set in "$@"; do
  rm -rf /tmp/sx; timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sx -o p -- python tools/seed_probe.py $N $V > /tmp/sx.log 2>&1
  python - >> gpurun_out/pmc_probe.txt <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for fn in glob.glob('/tmp/sx/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_seed_wg" in k or "k_dense_search" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, d in acc.items():
    print("   ", k[:32], "launches", len(n[k]), " ".join(f"{c}={v / max(1, len(n[k])):.4g}" for c, v in sorted(d.items())))
PY
done
cat gpurun_out/pmc_probe.txt
