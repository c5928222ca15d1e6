"""How long does hipMalloc of a 4 GiB Krylov slab take?  (a) in a fresh process on a box whose VRAM nobody has touched; (b) after
other processes have allocated, written and released most of the VRAM.  tools/malloc_probe.py [fill_GiB]: with an argument the
process first allocates and writes that much device memory and exits (run it before the measurement)."""
import ctypes
import sys
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
GiB = 1 << 30
if len(sys.argv) > 1:
    ptrs = []
    for _ in range(int(sys.argv[1]) // 4):
        p = ctypes.c_void_p()
        if hip.hipMalloc(ctypes.byref(p), 4 * GiB) != 0:
            break
        hip.hipMemset(p, 1, 4 * GiB)
        ptrs.append(p)
    hip.hipDeviceSynchronize()
    print("filled %d GiB" % (4 * len(ptrs)), flush=True)
    sys.exit(0)
hip.hipDeviceSynchronize()
ts = []
ptrs = []
for _ in range(8):
    p = ctypes.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(ctypes.byref(p), 4 * GiB)
    ts.append((time.perf_counter() - t0) * 1e3)
    ptrs.append(p)
print("hipMalloc(4 GiB) x 8: " + " ".join("%.1f" % t for t in ts) + " ms", flush=True)
t0 = time.perf_counter()
for p in ptrs:
    hip.hipFree(p)
print("hipFree x 8: %.1f ms in total" % ((time.perf_counter() - t0) * 1e3), flush=True)
