#!/usr/bin/env python3
"""how long hipMalloc / hipFree of the index-sized buffers take on this box (the lean build's first allocation is 49 GB): one piece or many,
first time in the process and again after a free; and hipMemset of the same bytes for scale"""
import ctypes as C, time, sys
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
def t(f):
    t0 = time.perf_counter(); r = f(); hip.hipDeviceSynchronize(); return (time.perf_counter() - t0) * 1e3, r
def malloc(n):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), n)
    assert rc == 0, rc
    return p
GB = 1 << 30
ms, _ = t(lambda: hip.hipSetDevice(0)); print("hipSetDevice %.1f ms" % ms)
ms, p0 = t(lambda: malloc(64)); print("first hipMalloc(64 B) %.1f ms" % ms, flush=True)
for rep in range(3):
    ms, p = t(lambda: malloc(49 * GB)); print("hipMalloc(49 GB) #%d: %.1f ms" % (rep, ms), flush=True)
    ms2, _ = t(lambda: hip.hipMemset(p, 0, 49 * GB)); print("   hipMemset of it: %.1f ms" % ms2)
    ms3, _ = t(lambda: hip.hipFree(p)); print("   hipFree: %.1f ms" % ms3, flush=True)
ms, ps = t(lambda: [malloc(GB) for _ in range(49)]); print("49 x hipMalloc(1 GB): %.1f ms" % ms, flush=True)
ms, _ = t(lambda: [hip.hipFree(p) for p in ps]); print("49 x hipFree: %.1f ms" % ms)
ms, p = t(lambda: malloc(49 * GB)); print("hipMalloc(49 GB) after the pieces: %.1f ms" % ms)
