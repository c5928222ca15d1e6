"""Kernel-to-kernel gaps from a rocprofv3 --kernel-trace csv: busy time, idle time between consecutive kernels, and the
per-kernel mean duration / mean gap before it, for the dispatches of the steady part of the trace.
  python tools/trace_gaps.py <..._kernel_trace.csv> [skip_first_n]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
busy = 0
gaps = defaultdict(list)
durs = defaultdict(list)
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    durs[name].append(e - s)
    if prev_end is not None:
        gaps[name].append(max(0, s - prev_end))
    busy += e - s
    prev_end = max(prev_end or 0, e)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("dispatches %d, span %.3f ms, busy %.3f ms (%.1f %%)" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span))
print("%-62s %7s %10s %12s" % ("kernel", "calls", "mean us", "gap before us"))
for name in sorted(durs, key=lambda k: -sum(durs[k])):
    g = gaps.get(name, [0])
    g2 = sorted(g)
    print("%-62s %7d %10.2f %12.2f (median %.2f)" % (name, len(durs[name]), sum(durs[name]) / len(durs[name]) / 1e3, sum(g) / len(g) / 1e3, g2[len(g2) // 2] / 1e3))
