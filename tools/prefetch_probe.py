#!/usr/bin/env python3
"""Round 4: does the upload of the next contig really hide behind the stages of the current one?  One context, 250 Mb contigs.
    python tools/prefetch_probe.py [genome_len]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch; torch.zeros(1, device="cuda")
from gsalign_amd import synth, hostlib, indexio, capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
tmp = os.environ.get("GSA_BENCH_TMP") or tempfile.mkdtemp(prefix="pfprobe_"); os.makedirs(tmp, exist_ok=True)
r = synth.fast_genome(n, 11000); synth.inject_repeats(r, 11000)
px = os.path.join(tmp, f"human_{n}")
if not os.path.exists(px + ".done"):
    synth.write_fasta(px + ".fa", [("chr1", r)]); hostlib.build_index(px + ".fa", px); open(px + ".done", "w").close()
idx = indexio.load_index(px)
g = capi.Aligner(idx)
qs = [g.pinned_copy(synth.fast_mutate(r, 0.01, 7000 + 10 * k)) for k in range(4)]
dv = [g.device_copy(q) for q in qs]
def T(f, reps=8):
    f(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
k = [0]
def nxt(): k[0] += 1; return k[0] % 4
import ctypes as C
def res_raw(i):
    res = capi.Result(); rc = g.lib.gsa_align_contig_device(g.ctx, C.c_void_p(dv[i].ptr), C.c_int32(dv[i].size), C.byref(res)); assert rc == 0
print("resident                 %.2f ms" % T(lambda: res_raw(nxt())))
print("pinned, upload in call   %.2f ms" % T(lambda: g.align_contig_raw(qs[nxt()]) and None))
def pre_wait():
    i = nxt(); g.prefetch_contig(qs[i]); time.sleep(0.02); t = time.perf_counter(); g.align_contig_raw(qs[i]); return time.perf_counter() - t
pre_wait(); print("prefetched + idle 20 ms, align only  %.2f ms" % (1e3 * float(np.median([pre_wait() for _ in range(8)]))))
t = time.perf_counter(); g.prefetch_contig(qs[0]); dt = time.perf_counter() - t; g.cancel_prefetch()
print("gsa_prefetch_contig call itself      %.3f ms" % (1e3 * dt))
def chain(m=12):
    g.prefetch_contig(qs[0]); t = time.perf_counter()
    for j in range(m):
        if j + 1 < m: g.prefetch_contig(qs[(j + 1) % 4])
        g.align_contig_raw(qs[j % 4])
    return (time.perf_counter() - t) / m
chain(4); print("chain prefetch(next); align(cur)     %.2f ms per contig" % (1e3 * chain()))
def chain_res(m=12):
    t = time.perf_counter()
    for j in range(m): res_raw(j % 4)
    return (time.perf_counter() - t) / m
chain_res(4); print("chain resident                       %.2f ms per contig" % (1e3 * chain_res()))
# a bare copy beside the stages: the next contig's upload issued through the library, then resident alignment
def chain_res_copy(m=12):
    t = time.perf_counter()
    for j in range(m):
        g.prefetch_contig(qs[(j + 1) % 4]); res_raw(j % 4); g.cancel_prefetch()
    return (time.perf_counter() - t) / m
chain_res_copy(4); print("chain resident + an unrelated upload beside it   %.2f ms per contig" % (1e3 * chain_res_copy()), flush=True)
# which stage pays for an upload beside it?  stage timers (hipEvents), resident contig, with and without an unrelated 250 MB upload started right before
g.set_profiling(True)
def res_raw2(i):
    res = capi.Result(); rc = g.lib.gsa_align_contig_device(g.ctx, C.c_void_p(dv[i].ptr), C.c_int32(dv[i].size), C.byref(res)); assert rc == 0
for label, up in (("quiet", False), ("upload beside", True), ("quiet", False), ("upload beside", True)):
    acc = np.zeros(8)
    for j in range(8):
        if up: g.prefetch_contig(qs[(j + 1) % 4])
        t = time.perf_counter(); res_raw2(j % 4); w = time.perf_counter() - t
        if up: g.cancel_prefetch()
        acc += g.timings().astype(np.float64); acc[6] += 1e3 * w
    acc /= 8
    print("%-14s seed %.2f | locate %.2f | sort %.2f | chain %.2f | refine %.2f | extend %.2f | wall %.2f" % (label, acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6]))
g.close()
