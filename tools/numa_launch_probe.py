"""Is the bimodal run time of launch-bound problems (config 5: 3.0-3.5 ms or 6.4-7.4 ms per Exponentiator run, fixed for
the life of a process) the NUMA placement of the host thread relative to the GPU?  Runs the same measurement in child
processes pinned to (a) the GPU's local CPU list, (b) a CPU list of another NUMA node, (c) not pinned, and prints the
CPU each child actually ran on.
  python tools/numa_launch_probe.py > gpurun_out/numa_launch_probe.txt"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, time
sys.path.insert(0, %r)
cpus = sys.argv[1]
if cpus != "none":
    os.sched_setaffinity(0, {int(c) for c in cpus.split(",")})
import numpy as np
import lambda_lanczos_amd as L
from lambda_lanczos_amd import generators as G
ctx = L.Context(0)
side = 1000
n = side * side
csr = G.torus(side, 0, n)
op = L.CsrOperator(ctx, csr[0], csr[1], csr[2])
init = G.start_vector_fast(n, 1, np.complex128, 0)
eng = L.Exponentiator(op, n)
eng.max_iteration = 100
d_in, d_out = ctx.to_device(init), ctx.empty((n,), np.complex128)
for _ in range(3):
    eng.run(-5j, d_in, out=d_out)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); _, it = eng.run(-5j, d_in, out=d_out); ts.append(time.perf_counter() - t0)
ts.sort()
print("cpu now %%d  iterations %%d  median %%.3f ms  min %%.3f ms  max %%.3f ms" %% (int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36]), it, ts[10] * 1e3, ts[0] * 1e3, ts[-1] * 1e3))
""" % ROOT


def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


def main():
    nodes = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(os.path.join(d, "cpulist")).read())
    print("host NUMA nodes:", {k: "%d cpus (%d..%d)" % (len(v), v[0], v[-1]) for k, v in nodes.items() if v})
    gpu_nodes = []
    for dev in glob.glob("/sys/class/drm/card*/device"):
        try:
            vendor = open(os.path.join(dev, "vendor")).read().strip()
            if vendor != "0x1002":
                continue
            gpu_nodes.append((os.path.realpath(dev).rsplit("/", 1)[1], int(open(os.path.join(dev, "numa_node")).read()),
                              open(os.path.join(dev, "local_cpulist")).read().strip()))
        except OSError:
            pass
    print("AMD GPUs (pci id, numa node, local cpulist):", gpu_nodes)
    print("affinity of this process: %d cpus" % len(os.sched_getaffinity(0)))
    allowed = sorted(os.sched_getaffinity(0))
    local = [c for c in (cpulist(gpu_nodes[0][2]) if gpu_nodes and gpu_nodes[0][2] else []) if c in allowed]
    remote = [c for c in allowed if c not in local]
    plans = [("not pinned", "none")] * 3
    if local:
        plans += [("local cpus, one core", str(local[len(local) // 2]))] * 2 + [("local cpus, all", ",".join(map(str, local)))] * 2
    if remote:
        plans += [("remote cpus, one core", str(remote[len(remote) // 2]))] * 2
    for name, arg in plans:
        r = subprocess.run([sys.executable, "-c", CHILD, arg], capture_output=True, text=True, timeout=300)
        print("%-24s %s" % (name, (r.stdout.strip() or r.stderr.strip()[-300:])))


if __name__ == "__main__":
    main()
