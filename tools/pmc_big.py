#!/usr/bin/env python3
"""A rocprofv3 counter_collection.csv -> for every kernel, the counters of its LONGEST dispatch (grid, duration, one column per counter).
For workloads whose launches differ in size by orders of magnitude (yeast: single contigs of the counting pass beside 60 Mb bundles) the
per-kernel means of pmc_reduce.py say nothing.   python tools/pmc_big.py raw.csv > out.txt"""
import csv, sys
from collections import defaultdict
d = defaultdict(dict); meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"], r["Dispatch_Id"])
    d[key][r["Counter_Name"]] = d[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    meta[key] = (int(r.get("End_Timestamp", 0)) - int(r.get("Start_Timestamp", 0)), int(r.get("Grid_Size", 0)), int(r.get("Workgroup_Size", 0)), int(r.get("LDS_Block_Size", 0)))
best = {}
for (k, disp), c in d.items():
    if k not in best or meta[(k, disp)][0] > meta[(k, best[k])][0]: best[k] = disp
cols = sorted({c for v in d.values() for c in v})
print(f"{'kernel':44s} {'dur_us':>9s} {'grid':>10s} {'wg':>5s} {'lds':>7s} " + " ".join(f"{c[3:][:13]:>13s}" for c in cols))
for k in sorted(best, key=lambda k: -meta[(k, best[k])][0])[:40]:
    m = meta[(k, best[k])]; c = d[(k, best[k])]
    print(f"{k.split('(')[0].replace('void ', '')[:44]:44s} {m[0] / 1000.0:9.1f} {m[1]:10d} {m[2]:5d} {m[3]:7d} " + " ".join(f"{c.get(x, 0):13.4g}" for x in cols))
