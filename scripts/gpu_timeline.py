#!/usr/bin/env python3
"""How busy the device is during the timed steps of bench.py: reads the kernel trace of `rocprofv3 --kernel-trace --output-format csv`
and prints, for the window of the last N k_stage_a launches, the fraction of time with 0, 1, 2, ... kernels of this library running."""
import csv
import glob
import sys

d, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "mtgi::" in r["Kernel_Name"]]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows]
sa = sorted(e for e in ev if "k_stage_a" in e[2])[-last:]
t0, t1 = sa[0][0], sa[-1][1]
pts = []
for s, e, _ in ev:
    if e <= t0 or s >= t1:
        continue
    pts.append((max(s, t0), 1))
    pts.append((min(e, t1), -1))
pts.sort()
busy, cur, prev = {}, 0, t0
for t, dlt in pts:
    busy[cur] = busy.get(cur, 0) + (t - prev)
    cur += dlt
    prev = t
busy[cur] = busy.get(cur, 0) + (t1 - prev)
tot = t1 - t0
print("window %.1f ms, %d traversals -> %.2f ms per step" % (tot / 1e6, len(sa), tot / 1e6 / len(sa)))
for k in sorted(busy):
    print("  %d kernels running: %5.1f %%" % (k, 100.0 * busy[k] / tot))
for name in ("k_stage_a", "k_post"):
    ds = [e - s for s, e, n in ev if name in n and s >= t0 and e <= t1]
    print("  %s: %d launches, %.3f ms average" % (name, len(ds), sum(ds) / max(len(ds), 1) / 1e6))
