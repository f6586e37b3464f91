#!/usr/bin/env python3
"""Round 4: A/B of the upload modes of gsa_align_many in ONE process, alternating, several rounds: contigs resident / uploaded with the
prefetch / uploaded when their turn comes.  Four contexts, 250 Mb contigs.   GPU_MAX_HW_QUEUES=8 python tools/h2d_ab.py [inflight] [steps]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch; torch.zeros(1, device="cuda")
from gsalign_amd import synth, hostlib, indexio, capi
n = 250_000_000; nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 4; m = int(sys.argv[2]) if len(sys.argv) > 2 else 80
tmp = os.environ.get("GSA_BENCH_TMP") or tempfile.mkdtemp(prefix="pfprobe_"); os.makedirs(tmp, exist_ok=True)
r = synth.fast_genome(n, 11000); synth.inject_repeats(r, 11000)
px = os.path.join(tmp, f"human_{n}")
if not os.path.exists(px + ".done"):
    synth.write_fasta(px + ".fa", [("chr1", r)]); hostlib.build_index(px + ".fa", px); open(px + ".done", "w").close()
idx = indexio.load_index(px)
g = capi.Aligner(idx); ctx = [g] + [g.clone() for _ in range(nctx - 1)]
qs = [g.pinned_copy(synth.fast_mutate(r, 0.01, 7000 + 10 * k)) for k in range(4)]
dv = [g.device_copy(q) for q in qs]
import ctypes as C
def walls():
    tot = np.zeros(10); n = 0
    for a in ctx:
        ms = (C.c_double * 10)(); k = C.c_int64()
        a.lib.gsa_get_wall_sums(a.ctx, ms, C.byref(k)); v = np.array(list(ms))
        if a is ctx[0]: up = v[9] / max(1, k.value)
        v[9] = 0; tot += v; n += k.value
    tot /= max(1, n); tot[9] = up
    return tot
def run(label, src, **kw):
    capi.align_many(ctx, src * 2, in_order=True, **kw)
    for a in ctx: a.set_profiling(False)
    torch.cuda.synchronize(); t = time.perf_counter(); capi.align_many(ctx, (src * (m // 4 + 1))[:m], in_order=True, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{label:58s} {m * n / dt / 1e9:6.2f} Gbp/s   {1e3 * dt / m:6.2f} ms/contig   host wall per contig: set-up %.2f | s1 %.2f | s2 %.2f | s3 %.2f | s4-6 %.2f | s7 %.2f | s8 %.2f | one upload %.2f" % ((lambda w: (w[0], w[1], w[2], w[3], w[4] + w[5] + w[6], w[7], w[8], w[9]))(walls())), flush=True)
for rnd in range(int(os.environ.get("AB_ROUNDS", "2"))):
    run("resident", dv)
    run("uploaded, prefetch", qs)
    run("uploaded, no prefetch", qs, prefetch=False)
for c in ctx[1:]: c.close()
g.close()
