#!/usr/bin/env python3
"""Striped / small DP kernels alone on batches of one shape (GPU box); run under rocprofv3 --kernel-trace --stats to get the kernel times.
usage: dp_batch_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gsalign_amd import capi, indexio, synth
import gzip, shutil, tempfile
tmp = tempfile.mkdtemp()
for ext in ("bwt", "sa", "pac", "ann", "amb"):
    with gzip.open(os.path.join(ROOT, "tests", "golden", f"small.{ext}.gz"), "rb") as a, open(os.path.join(tmp, f"small.{ext}"), "wb") as b:
        shutil.copyfileobj(a, b)
g = capi.Aligner(indexio.load_index(os.path.join(tmp, "small")))
for (cnt, L) in ((20000, 30), (8000, 100), (6000, 200), (3000, 400), (600, 800), (150, 1600), (16, 3000)):
    s1, s2 = [], []
    for i in range(cnt):
        a = synth.fast_genome(L, 100 + i); b = synth.fast_mutate(a, 0.08, 200 + i)
        s1.append(a.tobytes()); s2.append(b.tobytes())
    cells = sum(len(x) * len(y) for x, y in zip(s1, s2))
    g.ksw2_batch(s1, s2)
    t = time.time(); g.ksw2_batch(s1, s2); dt = time.time() - t
    print(f"{cnt} pairs of ~{L}x{L}: {cells / 1e6:.1f} Mcells, call {dt * 1e3:.2f} ms (incl. copies)", flush=True)
