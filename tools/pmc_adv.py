#!/usr/bin/env python3
"""Counters of the adversarial workload's dense seed kernels per launch (gpurun_out/pmc_adversarial/*.csv and pmc_sq_adversarial/*.csv,
written by tools/r4_adv_pmc.sh): python tools/pmc_adv.py > profiles/archive/r04_pmc_adversarial.txt"""
import csv
import glob

rows = {}
for fn in sorted(glob.glob("gpurun_out/pmc_adversarial/*.csv") + glob.glob("gpurun_out/pmc_sq_adversarial/*.csv")):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("k_dense_sweep", "k_dense_resolve", "k_seed_select", "k_dp_stripe", "k_seed_wg")):
            continue
        short = k.split("(")[0].replace("void ", "")
        rows.setdefault(short, {})[r["Counter_Name"]] = (float(r["Counter_Value"]), int(r["Launches"]))
print("# per-launch counters, adversarial workload (250 Mb, one context, bench.py --inflight 1): tools/r4_adv_pmc.sh")
print("# SQ_* as rocprofv3 reports them (SQ_INSTS_* = wave-instructions, SQ_WAVE_CYCLES / SQ_BUSY_CYCLES in the units of the SQ counters); TCC_* summed over the channels")
for k, d in rows.items():
    print(k)
    for c in sorted(d):
        v, n = d[c]
        print(f"    {c:28s} {v / n:18.0f} per launch   ({n} launches)")
    g = lambda c: d[c][0] / d[c][1] if c in d else 0.0
    if g("SQ_WAVES") and g("SQ_INSTS_VMEM_RD"):
        print(f"    -> VALU wave-instructions per wave {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.0f}; read requests to the fabric x 128 B = {g('TCC_EA0_RDREQ_sum') * 128 / 1e9:.2f} GB, write requests {g('TCC_EA0_WRREQ_sum') / 1e6:.1f} M; L2 hit rate {g('TCC_HIT_sum') / max(1.0, g('TCC_REQ_sum')):.2f}")
