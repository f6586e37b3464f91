#!/usr/bin/env python3
"""What does pinned host memory cost on this host?  hipHostMalloc of a few GB (and a first touch of it), hipHostRegister of malloc'ed memory that
is already touched, and the H2D rate from pageable / registered / hipHostMalloc memory.  GPU box."""
import ctypes as C, time, sys
import numpy as np
hip = C.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]; hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostUnregister.argtypes = [C.c_void_p]; hip.hipHostFree.argtypes = [C.c_void_p]
GB = 1 << 30; n = int(sys.argv[1]) * GB if len(sys.argv) > 1 else 4 * GB
d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), n) == 0
hip.hipDeviceSynchronize()
a = np.empty(n, np.uint8); t = time.time(); a[::4096] = 1; print(f"first touch of {n / GB:.0f} GB of malloc memory: {time.time() - t:.2f} s")
t = time.time(); assert hip.hipMemcpy(d, a.ctypes.data, n, 1) == 0; dt = time.time() - t; print(f"H2D from pageable: {dt:.2f} s = {n / dt / 1e9:.1f} GB/s")
t = time.time(); assert hip.hipMemcpy(d, a.ctypes.data, n, 1) == 0; dt = time.time() - t; print(f"H2D from pageable (again): {dt:.2f} s = {n / dt / 1e9:.1f} GB/s")
t = time.time(); rc = hip.hipHostRegister(a.ctypes.data, n, 0); print(f"hipHostRegister: rc {rc}, {time.time() - t:.2f} s")
t = time.time(); assert hip.hipMemcpy(d, a.ctypes.data, n, 1) == 0; dt = time.time() - t; print(f"H2D from registered: {dt:.2f} s = {n / dt / 1e9:.1f} GB/s")
hip.hipHostUnregister(a.ctypes.data)
p = C.c_void_p(); t = time.time(); rc = hip.hipHostMalloc(C.byref(p), n, 0); print(f"hipHostMalloc: rc {rc}, {time.time() - t:.2f} s")
b = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)); t = time.time(); b[::4096] = 1; print(f"first touch of it: {time.time() - t:.2f} s")
t = time.time(); assert hip.hipMemcpy(d, p, n, 1) == 0; dt = time.time() - t; print(f"H2D from hipHostMalloc memory: {dt:.2f} s = {n / dt / 1e9:.1f} GB/s")
