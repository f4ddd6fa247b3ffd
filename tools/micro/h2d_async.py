#!/usr/bin/env python
"""GPU-box micro-measurement: hipMemcpy vs hipMemcpyAsync (+ sync) from pageable and pinned host
memory, 200 MB (a chr1-sized pixel table's column array)."""
import ctypes as C, time, numpy as np
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
nb = 200 << 20
d = C.c_void_p(); hip.hipMalloc(C.byref(d), nb)
s = C.c_void_p(); hip.hipStreamCreate(C.byref(s))
pin = C.c_void_p(); hip.hipHostMalloc(C.byref(pin), nb, 0)
for name in ("hipMemcpy pageable", "hipMemcpyAsync pageable + sync", "hipMemcpyAsync pinned + sync", "hipHostRegister + async + unregister"):
    ts = []
    for rep in range(4):
        src = np.full(nb, rep, np.uint8)   # a fresh array, as the reader hands one over
        t0 = time.perf_counter()
        if name == "hipMemcpy pageable":
            hip.hipMemcpy(d, src.ctypes.data, nb, 1)
        elif name == "hipMemcpyAsync pageable + sync":
            hip.hipMemcpyAsync(d, src.ctypes.data, nb, 1, s); hip.hipStreamSynchronize(s)
        elif name == "hipMemcpyAsync pinned + sync":
            hip.hipMemcpyAsync(d, pin, nb, 1, s); hip.hipStreamSynchronize(s)
        else:
            hip.hipHostRegister(src.ctypes.data, nb, 0)
            hip.hipMemcpyAsync(d, src.ctypes.data, nb, 1, s); hip.hipStreamSynchronize(s)
            hip.hipHostUnregister(src.ctypes.data)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%-40s %s ms  -> %.1f GB/s" % (name, " ".join("%6.1f" % t for t in ts), nb / min(ts) / 1e6))
