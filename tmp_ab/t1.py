import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from gsalign_amd import capi
wl = dict(bench.WORKLOADS["human"])
tmp = tempfile.mkdtemp(prefix="replay_")
px, idx, refs = bench.build_reference(tmp, "human", wl, 0, 1)
qs = [x[0] for x in bench.make_queries(wl, refs, 0)]
g = capi.Aligner(idx, **wl["params"])
pq = [g.pinned_copy(q) for q in qs]
for rep in range(3):
    for q in pq:
        t = time.time(); g.align_contig_raw(q); print(f"align_contig_raw {1e3 * (time.time() - t):.2f} ms", flush=True)
