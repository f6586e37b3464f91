"""The striped-class DP jobs of one human-sized contig replayed through gsa_ksw2_batch (only them)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from gsalign_amd import capi
name = "human"
wl = dict(bench.WORKLOADS[name])
tmp = tempfile.mkdtemp(prefix="replay_")
px, idx, refs = bench.build_reference(tmp, name, wl, 0, 1)
ref = refs[0][1]
q = bench.make_queries(wl, refs, 0)[0][0]
g = capi.Aligner(idx, **wl["params"])
r = g.align_contig(q)
F = r["frags"]
gap = F[(F["bseed"] == 0)]
m, n = gap["rlen"].astype(np.int64), gap["qlen"].astype(np.int64)
rp, qp = gap["rpos"].astype(np.int64), gap["qpos"].astype(np.int64)
ok = (m > 0) & (n > 0) & (m != n) & ~((n <= 64) & (m + n - 1 <= 128)) & (rp + m <= ref.size)
print("striped jobs", int(ok.sum()), "cells", int((m[ok] * n[ok]).sum()), flush=True)
rb = ref.tobytes(); qb = q.tobytes()
s1 = [rb[a:a + l] for a, l in zip(rp[ok], m[ok])]; s2 = [qb[a:a + l] for a, l in zip(qp[ok], n[ok])]
for mode in ("all", "sorted", "mid"):
    if mode == "sorted":
        o = np.argsort(-(m[ok] * n[ok])); a1 = [s1[i] for i in o]; a2 = [s2[i] for i in o]
    elif mode == "mid":
        sel = [i for i in range(len(s1)) if len(s1[i]) <= 768]; a1 = [s1[i] for i in sel]; a2 = [s2[i] for i in sel]
        print("mid jobs", len(a1), "cells", sum(len(x) * len(y) for x, y in zip(a1, a2)))
    else: a1, a2 = s1, s2
    g.ksw2_batch(a1, a2)
    t = time.time(); g.ksw2_batch(a1, a2); print(mode, f"call {1e3 * (time.time() - t):.2f} ms (incl. copies)", flush=True)
