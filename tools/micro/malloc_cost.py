#!/usr/bin/env python
"""GPU-box micro-measurement: what hipMalloc / hipFree / hipMemset / a pageable H2D copy cost per
size (the per-chromosome preparation makes dozens of them)."""
import ctypes as C, time, numpy as np
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipDeviceSynchronize()
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 1 << 20); hip.hipFree(p)
print("%10s %10s %10s %10s %12s %12s" % ("MB", "malloc us", "free us", "memset us", "H2D page us", "H2D pinned us"))
for mb in (0.001, 0.064, 1, 4, 16, 64, 128, 256, 1024):
    nb = int(mb * (1 << 20))
    src = np.ones(nb, np.uint8)
    pin = C.c_void_p(); hip.hipHostMalloc(C.byref(pin), nb, 0)
    tm = tf = ts = tc = tp = 0.0
    reps = 5
    for _ in range(reps):
        t0 = time.perf_counter(); hip.hipMalloc(C.byref(p), nb); t1 = time.perf_counter()
        hip.hipMemset(p, 0, nb); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
        hip.hipMemcpy(p, src.ctypes.data, nb, 1); t3 = time.perf_counter()
        hip.hipMemcpy(p, pin, nb, 1); t4 = time.perf_counter()
        hip.hipFree(p); t5 = time.perf_counter()
        tm += t1 - t0; ts += t2 - t1; tc += t3 - t2; tp += t4 - t3; tf += t5 - t4
    print("%10.3f %10.1f %10.1f %10.1f %12.1f %12.1f" % (mb, tm / reps * 1e6, tf / reps * 1e6, ts / reps * 1e6, tc / reps * 1e6, tp / reps * 1e6))
