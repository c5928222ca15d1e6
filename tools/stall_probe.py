"""Where do the rare ~80 ms outliers of launch-bound runs come from?  One process, `runs` Exponentiator runs of config 5
(2.9 ms each); prints every run slower than twice the median with its start time and the library's own phase times.
  python tools/stall_probe.py [runs]      (env: LL_TRIDIAG_THREAD=0/1, PIN=cpu)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("PIN"):
    os.sched_setaffinity(0, {int(os.environ["PIN"])})
import numpy as np  # noqa: E402

T_IMPORT = time.perf_counter()

import lambda_lanczos_amd as L  # noqa: E402
from lambda_lanczos_amd import generators as G  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1500


def vmstat():
    want = ("numa_pte_updates", "numa_hint_faults", "numa_pages_migrated", "pgfault", "thp_fault_alloc", "compact_stall")
    out = {}
    for line in open("/proc/vmstat"):
        k, v = line.split()
        if k in want:
            out[k] = int(v)
    return out


try:
    print("kernel.numa_balancing =", open("/proc/sys/kernel/numa_balancing").read().strip())
except OSError as e:
    print("numa_balancing: ", e)
ctx = L.Context(0)
side = 1000
n = side * side
csr = G.torus(side, 0, n)
op = L.CsrOperator(ctx, csr[0], csr[1], csr[2])
init = G.start_vector_fast(n, 1, np.complex128, 0)
eng = L.Exponentiator(op, n)
eng.max_iteration = 100
d_in, d_out = ctx.to_device(init), ctx.empty((n,), np.complex128)
for _ in range(5):
    eng.run(-5j, d_in, out=d_out)
if os.environ.get("SLEEP"):
    time.sleep(float(os.environ["SLEEP"]))
rec = []
vm0 = vmstat()
t00 = time.perf_counter()
for _ in range(runs):
    t0 = time.perf_counter()
    eng.run(-5j, d_in, out=d_out)
    t1 = time.perf_counter()
    s = eng.last_stats
    rec.append((t0 - t00, t1 - t0, s["seconds_total"], s["seconds_host_enqueue"], s["seconds_host_wait"]))
vm1 = vmstat()
print("vmstat deltas over the timed runs:", {k: vm1[k] - vm0[k] for k in vm0})
d = sorted(r[1] for r in rec)
med = d[len(d) // 2]
print("since import %.2f s;" % (time.perf_counter() - T_IMPORT), end=" ")
print("thread=%s pin=%s: %d runs in %.2f s, median %.3f ms, p99 %.3f ms, max %.3f ms, mean %.3f ms" % (
    os.environ.get("LL_TRIDIAG_THREAD", "default"), os.environ.get("PIN", "-"), runs, time.perf_counter() - t00, med * 1e3,
    d[int(len(d) * 0.99)] * 1e3, d[-1] * 1e3, sum(d) / len(d) * 1e3))
for r in rec:
    if r[1] > 2 * med:
        print("  at %8.3f s (%.3f s after import): run %.2f ms = library %.2f ms (enqueue %.2f, wait %.2f) + outside %.2f ms" % (
            r[0], r[0] + t00 - T_IMPORT, r[1] * 1e3, r[2] * 1e3, r[3] * 1e3, r[4] * 1e3, (r[1] - r[2]) * 1e3))
