#!/usr/bin/env python3
"""Are async D2H copies behind a kernel on the same stream shader blits (__amd_rocclr_copyBuffer) or SDMA?  Run under
rocprofv3 --kernel-trace --stats and look for __amd_rocclr_copyBuffer.   python tools/blit_probe.py [torch]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch; torch.zeros(1, device="cuda")
from gsalign_amd import capi
lib = capi.load_library()
hip = C.CDLL([ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln][0])
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
N = 64 << 20
h = lib.gsa_host_alloc(N); d = lib.gsa_device_alloc(0, N)
s = C.c_void_p(); hip.hipStreamCreate(C.byref(s))
for rep in range(5):
    t = time.perf_counter()
    hip.hipMemsetAsync(C.c_void_p(d), rep, N, s)              # a (fill) kernel in front of the copy, same stream
    hip.hipMemcpyAsync(C.c_void_p(h), C.c_void_p(d), N, 2, s)  # D2H into pinned memory
    hip.hipStreamSynchronize(s)
    print(f"fill + D2H 64 MB: {1e3 * (time.perf_counter() - t):.2f} ms")
# variants: the copy on a SECOND stream behind an event / behind a host wait / with the producing kernel writing pinned memory itself
s2 = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s2), 1)
ev = C.c_void_p(); hip.hipEventCreateWithFlags(C.byref(ev), 2)
for rep in range(3):
    t = time.perf_counter()
    hip.hipMemsetAsync(C.c_void_p(d), rep, N, s); hip.hipEventRecord(ev, s); hip.hipStreamWaitEvent(s2, ev, 0)
    hip.hipMemcpyAsync(C.c_void_p(h), C.c_void_p(d), N, 2, s2); hip.hipStreamSynchronize(s2)
    print(f"[event] fill on A, D2H 64 MB on B behind an event: {1e3 * (time.perf_counter() - t):.2f} ms")
for rep in range(3):
    t = time.perf_counter()
    hip.hipMemsetAsync(C.c_void_p(d), rep, N, s); hip.hipStreamSynchronize(s)
    hip.hipMemcpyAsync(C.c_void_p(h), C.c_void_p(d), 32 << 20, 2, s2); hip.hipStreamSynchronize(s2)
    print(f"[host wait] fill on A, host waits, D2H 32 MB on B: {1e3 * (time.perf_counter() - t):.2f} ms")
