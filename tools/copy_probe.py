#!/usr/bin/env python3
"""How fast do pinned-host <-> device copies run in the process bench.py lives in (torch imported first: its bundled HIP runtime
serves libgsa_hip.so too) against a process that never imports torch (the system runtime)?  256 MB H2D and 64 MB D2H through the
library's own helpers, wall clock.   python tools/copy_probe.py [torch]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch; torch.zeros(1, device="cuda")
from gsalign_amd import capi
lib = capi.load_library()
hip = C.CDLL(None)      # whatever libamdhip64 the process resolved
for name in ("libamdhip64.so.7", "libamdhip64.so"):
    try:
        hip = C.CDLL(name); break
    except OSError:
        pass
print("maps:", sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "amdhip64" in ln or "hsa-runtime" in ln}))
N = 256 << 20
h = lib.gsa_host_alloc(N); d = lib.gsa_device_alloc(0, N)
C.memset(h, 65, N)
for rep in range(3):
    t = time.perf_counter(); lib.gsa_device_upload(0, C.c_void_p(d), C.c_void_p(h), N); dt = time.perf_counter() - t
    print(f"H2D 256 MB pinned: {dt * 1e3:.2f} ms = {N / dt / 1e9:.1f} GB/s")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
for rep in range(3):
    t = time.perf_counter(); hip.hipMemcpy(C.c_void_p(h), C.c_void_p(d), N // 4, 2); dt = time.perf_counter() - t
    print(f"D2H 64 MB pinned: {dt * 1e3:.2f} ms = {N / 4 / dt / 1e9:.1f} GB/s")
