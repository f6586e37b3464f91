#!/usr/bin/env python3
"""Aggregate H2D bandwidth of k concurrent pinned 250 MB copies, each on a stream of its own (what four contexts uploading at once do)."""
import ctypes as C, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch; torch.zeros(1, device="cuda")
from gsalign_amd import capi
lib = capi.load_library()
hip = C.CDLL([ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln][0])
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
N = 250_000_000
bufs = [(lib.gsa_host_alloc(N), lib.gsa_device_alloc(0, N)) for _ in range(8)]
for h, d in bufs: C.memset(h, 65, N)
def worker(h, d, reps, out):
    s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1)
    for _ in range(reps):
        t = time.perf_counter(); hip.hipMemcpyAsync(C.c_void_p(d), C.c_void_p(h), N, 1, s); hip.hipStreamSynchronize(s); out.append(time.perf_counter() - t)
for k in (1, 2, 4, 8):
    outs = [[] for _ in range(k)]; th = [threading.Thread(target=worker, args=(bufs[i][0], bufs[i][1], 12, outs[i])) for i in range(k)]
    t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; dt = time.perf_counter() - t
    import statistics
    print(f"{k} concurrent: aggregate {k * 12 * N / dt / 1e9:.1f} GB/s, one copy {1e3 * statistics.median(sum(outs, [])):.2f} ms")
