#!/usr/bin/env python3
"""Histogram of the gap-closing DP jobs of one contig (GPU box): which (m, n) classes hold the cells and the jobs.
usage: dp_hist.py [workload of bench.py: human|human_like|adversarial|ecoli|yeast] [genome_len]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from gsalign_amd import capi

name = sys.argv[1] if len(sys.argv) > 1 else "human"
wl = dict(bench.WORKLOADS[name])
if len(sys.argv) > 2: wl["lengths"] = [int(sys.argv[2])]
tmp = tempfile.mkdtemp(prefix="dphist_")
import argparse
args = argparse.Namespace(fasta_ref="", fasta_query="")
px, idx, refs = bench.build_reference(tmp, name, wl, 0, 1, args)
q = bench.make_queries(wl, refs, args)[0][0]
g = capi.Aligner(idx, **wl["params"])
r = g.align_contig(q)
F = r["frags"]; B = r["blocks"]
gap = F[(F["bseed"] == 0)]
m, n = gap["rlen"].astype(np.int64), gap["qlen"].astype(np.int64)
both = (m > 0) & (n > 0)
# a gap with equal sides and <= 5 mismatches is copied without DP; approximate: alnlen == qlen == rlen and no '-' => count as nodp if equal length (upper bound of DP jobs otherwise)
print(f"{name}: contig {q.size} bp, {B.size} blocks, {F.size} records, gaps {gap.size}, both-sided {int(both.sum())}, equal-length {int((both & (m == n)).sum())}")
mm, nn = m[both & (m != n)], n[both & (m != n)]
cells = mm * nn
def cls(m_, n_):
    return np.where((n_ <= 16) & (m_ + n_ - 1 <= 64), 0, np.where((n_ <= 64) & (m_ + n_ - 1 <= 128), 1, 2))
c = cls(mm, nn)
for k, nm in enumerate(["tiny", "small", "striped"]):
    s = c == k
    print(f"  {nm:8s} jobs {int(s.sum()):9d}  cells {int(cells[s].sum()):14d}")
s = c == 2
edges = [0, 64, 128, 256, 512, 1024, 2048, 4096, 1 << 20]
print("  striped jobs by n (query side) x m (reference side): jobs / Mcells / stripe-steps M (sum ceil(n/64) * (m+63) * 64)")
for i in range(len(edges) - 1):
    for j in range(len(edges) - 1):
        t = s & (nn > edges[i]) & (nn <= edges[i + 1]) & (mm > edges[j]) & (mm <= edges[j + 1])
        if t.any():
            steps = (((nn[t] + 63) // 64) * (mm[t] + 63) * 64).sum()
            print(f"    n ({edges[i]},{edges[i+1]}] m ({edges[j]},{edges[j+1]}]: {int(t.sum()):8d} jobs {cells[t].sum()/1e6:10.1f} Mcells {steps/1e6:10.1f} Msteps")
o = np.argsort(-cells)[:24]
print('  largest jobs (m x n):', ' '.join(f'{int(mm[i])}x{int(nn[i])}' for i in o))
print('  jobs with m > 3968 (one wave per workgroup, hand-off through HBM):', int((mm > 3968).sum()), ' with m in (768, 3968]:', int(((mm > 768) & (mm <= 3968)).sum()), ' cells there:', int(cells[(mm > 768) & (mm <= 3968)].sum()))
g.close()
