#!/usr/bin/env python3
"""Round 4: what does bulk H2D traffic beside the stages cost?  Four contexts align RESIDENT contigs (gsa_align_many) while another host
thread uploads an unrelated 250 MB buffer over and over (duty = fraction of the time the link is kept busy).
    python tools/h2d_noise_probe.py"""
import os, sys, time, tempfile, threading, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch; torch.zeros(1, device="cuda")
from gsalign_amd import synth, hostlib, indexio, capi
n = 250_000_000
tmp = os.environ.get("GSA_BENCH_TMP") or tempfile.mkdtemp(prefix="pfprobe_"); os.makedirs(tmp, exist_ok=True)
r = synth.fast_genome(n, 11000); synth.inject_repeats(r, 11000)
px = os.path.join(tmp, f"human_{n}")
if not os.path.exists(px + ".done"):
    synth.write_fasta(px + ".fa", [("chr1", r)]); hostlib.build_index(px + ".fa", px); open(px + ".done", "w").close()
idx = indexio.load_index(px)
g = capi.Aligner(idx); ctx = [g] + [g.clone() for _ in range(3)]
qs = [g.pinned_copy(synth.fast_mutate(r, 0.01, 7000 + 10 * k)) for k in range(4)]
dv = [g.device_copy(q) for q in qs]
lib = g.lib
hbuf = lib.gsa_host_alloc(n); dbuf = lib.gsa_device_alloc(0, n)
stop = [False]; stats = [0, 0.0]
hip_path = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln][0]
hip = C.CDLL(hip_path)
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
ns = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(ns), 1) == 0      # a non-blocking stream of its own (the null stream would synchronise with the library's streams)
def noise(duty, piece):
    while not stop[0]:
        t = time.perf_counter()
        for off in range(0, n, piece):
            hip.hipMemcpyAsync(C.c_void_p(dbuf + off), C.c_void_p(hbuf + off), min(piece, n - off), 1, ns)
        hip.hipStreamSynchronize(ns)
        dt = time.perf_counter() - t; stats[0] += 1; stats[1] += dt
        if duty < 1.0: time.sleep(dt * (1 - duty) / duty)
def run(label, duty=0.0, piece=n, m=48):
    capi.align_many(ctx, dv * 2, in_order=True)
    stop[0] = False; stats[0] = 0; stats[1] = 0.0
    th = threading.Thread(target=noise, args=(duty, piece)) if duty > 0 else None
    if th: th.start()
    t = time.perf_counter(); capi.align_many(ctx, (dv * (m // 4 + 1))[:m], in_order=True); dt = time.perf_counter() - t
    stop[0] = True
    if th: th.join()
    print(f"{label:60s} {m * n / dt / 1e9:6.2f} Gbp/s   {1e3 * dt / m:6.2f} ms/contig   noise copies {stats[0]}, {1e3 * stats[1] / max(1, stats[0]):.2f} ms each", flush=True)
run("resident, quiet link")
run("resident + H2D noise, duty 0.45, whole buffer per copy", 0.45)
run("resident + H2D noise, duty 1.0", 1.0)
run("resident + H2D noise, duty 0.45, 8 MB pieces", 0.45, 8 << 20)
run("resident + H2D noise, duty 0.45, 1 MB pieces", 0.45, 1 << 20)
run("resident, quiet link (again)")
for c in ctx[1:]: c.close()
g.close()
