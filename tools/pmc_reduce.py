#!/usr/bin/env python3
"""A rocprofv3 counter_collection.csv (a row per dispatch and counter: tens of MB for a 24-contig step) -> per-kernel sums, on the GPU box, so
that the result fits what gpurun carries home.  Columns: Kernel_Name, Counter_Name, Counter_Value (sum over the launches), Launches.
    python tools/pmc_reduce.py raw.csv reduced.csv"""
import csv, sys
from collections import defaultdict
acc = defaultdict(float); seen = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"], r["Counter_Name"])
    acc[k] += float(r["Counter_Value"]); seen[k].add(r.get("Dispatch_Id"))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Launches"])
for (kn, cn), v in sorted(acc.items()):
    w.writerow([kn, cn, repr(v), len(seen[(kn, cn)])])
