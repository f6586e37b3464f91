#!/usr/bin/env python3
"""gpurun_out/pmc_sq_<workload>/s*.csv -> per-kernel SQ / LDS counter table (text, for profiles/)."""
import csv, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = sys.argv[1] if len(sys.argv) > 1 else "human"
D = os.path.join(ROOT, "gpurun_out", f"pmc_sq_{W}")
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for fn in sorted(os.listdir(D)):
    if not fn.endswith(".csv"): continue
    for r in csv.DictReader(open(os.path.join(D, fn))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += int(r["Launches"]) if "Launches" in r else 1
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
        "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
print(f"# rocprofv3 --pmc SQ counters, bench.py --workload {W} the workload's own --inflight (4 contexts; counter collection serialises the dispatches), per launch (mean); two passes of 8 counters")
print(f"{'kernel':44s} {'launches':>8s} " + " ".join(f"{c[3:][:13]:>13s}" for c in cols))
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
for k, d in rows[:32]:
    ln = max(n[k].values())
    print(f"{k:44s} {ln:8d} " + " ".join(f"{d.get(c, 0) / max(1, n[k].get(c, 1)):13.3g}" for c in cols))
print("# derived: VALU issue share = ACTIVE_INST_VALU / WAVE_CYCLES ; waiting share = WAIT_ANY / WAVE_CYCLES ; LDS conflict share = LDS_BANK_CONFLICT / LDS_IDX_ACTIVE")
for k, d in rows[:32]:
    wc = d.get("SQ_WAVE_CYCLES", 0) / max(1, n[k].get("SQ_WAVE_CYCLES", 1))
    if wc <= 0: continue
    g = lambda c: d.get(c, 0) / max(1, n[k].get(c, 1))
    print(f"{k:44s} valu {g('SQ_ACTIVE_INST_VALU') / wc:6.3f}  any {g('SQ_ACTIVE_INST_ANY') / wc:6.3f}  wait_any {g('SQ_WAIT_ANY') / wc:6.3f}  wait_inst {g('SQ_WAIT_INST_ANY') / wc:6.3f}  lds_conf {g('SQ_LDS_BANK_CONFLICT') / max(1.0, g('SQ_LDS_IDX_ACTIVE')):6.3f}  valu/wave {g('SQ_INSTS_VALU') / max(1.0, g('SQ_WAVES')):9.1f}")
