#!/usr/bin/env python3
"""Flake hunt (GPU box): the same contigs aligned over and over on four contexts at once (uploads prefetched since round 4) must give byte-identical results
every time (blocks, records, both string pools) -- the synchronisation of the striped DP, the fused passes and the seed
kernels runs without agent-scope fences, so a missing ordering would show up here as a rare difference.
usage: stress_consistency.py [workload=human] [genome_len] [rounds=12]"""
import os, sys, tempfile, threading, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ctypes as C
import bench
from gsalign_amd import capi
name = sys.argv[1] if len(sys.argv) > 1 else "human"
wl = dict(bench.WORKLOADS[name])
if len(sys.argv) > 2 and int(sys.argv[2]) > 0: wl["lengths"] = [int(sys.argv[2])]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 12
tmp = tempfile.mkdtemp(prefix="stress_")
import types
args = types.SimpleNamespace(fasta_ref="", fasta_query="")
px, idx, refs = bench.build_reference(tmp, name, wl, 0, 1, args)
contigs = [c for gq in bench.make_queries(wl, refs, args) for c in gq]
nctx = int(os.environ.get("STRESS_CTX", "4"))      # (round 6: eight, as bench.py runs the unique-text workloads)
g0 = capi.Aligner(idx, **wl["params"]); ctxs = [g0] + [g0.clone() for _ in range(nctx - 1)]
pinned = [g0.pinned_copy(c) for c in contigs]
sums = {}; lock = threading.Lock(); bad = []

def on_result(ci, res):
    k = ci % len(contigs)
    nb, nf, na = res.n_blocks, res.n_frags, res.n_aln
    h = zlib.crc32(C.string_at(res.blocks, nb * 40)) if nb else 0
    h = zlib.crc32(C.string_at(res.recs, nf * 16), h) if nf else h
    if na:
        h = zlib.crc32(C.string_at(res.aln1, na), h); h = zlib.crc32(C.string_at(res.aln2, na), h)
    with lock:
        if k in sums and sums[k] != (nb, nf, na, h): bad.append((ci, k, sums[k], (nb, nf, na, h)))
        sums.setdefault(k, (nb, nf, na, h))
    return 0

capi.align_many(ctxs, pinned * rounds, on_result)
print(f"{name}: {len(contigs)} contigs x {rounds} rounds on {len(ctxs)} contexts: {len(bad)} differences", sums if len(sums) < 6 else len(sums))
for b in bad[:5]: print("  DIFF", b)
sys.exit(1 if bad else 0)
