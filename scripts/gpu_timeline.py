#!/usr/bin/env python3
"""Device timeline of bench.py from `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv`: for the window of the last N
k_stage_a launches, how much of the time kernels / copies were running, average duration per kernel and per copy direction (with the
copy rate when the trace has byte counts), and the events of the last few steps in time order."""
import csv
import glob
import sys

d, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
kf = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(kf)) if "mtgi::" in r["Kernel_Name"]]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("mtgi::", ""), r.get("Queue_Id", "?")) for r in rows]
cf = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
cp = []
if cf:
    for r in csv.DictReader(open(cf[0])):
        b = None
        for key in ("Bytes", "Size", "bytes"):
            if key in r and r[key]:
                b = int(r[key])
        cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Kind", "?")), b))
sa = sorted(e for e in ev if "k_stage_a" in e[2])[-last:]
t0, t1 = sa[0][0], sa[-1][1]


def busy(intervals):
    pts = []
    for s, e in intervals:
        if e <= t0 or s >= t1:
            continue
        pts.append((max(s, t0), 1))
        pts.append((min(e, t1), -1))
    pts.sort()
    out, cur, prev = {}, 0, t0
    for t, dl in pts:
        out[cur] = out.get(cur, 0) + (t - prev)
        cur += dl
        prev = t
    out[cur] = out.get(cur, 0) + (t1 - prev)
    return out


tot = t1 - t0
print("window %.2f ms, %d traversals -> %.3f ms per step" % (tot / 1e6, len(sa), tot / 1e6 / len(sa)))
for label, iv in (("kernels", [(s, e) for s, e, _, _ in ev]), ("copies", [(s, e) for s, e, _, _ in cp])):
    bz = busy(iv)
    print("  %s running: " % label + ", ".join("%d: %.1f%%" % (k, 100.0 * bz[k] / tot) for k in sorted(bz)))
names = sorted({n for _, _, n, _ in ev})
for name in names:
    ds = [e - s for s, e, n, _ in ev if n == name and s >= t0 and e <= t1]
    if ds:
        print("  %-12s %4d launches, avg %.3f ms, max %.3f ms" % (name, len(ds), sum(ds) / len(ds) / 1e6, max(ds) / 1e6))
dirs = sorted({c[2] for c in cp})
for dr in dirs:
    cs = [c for c in cp if c[2] == dr and c[0] >= t0 and c[1] <= t1]
    if cs:
        tb = sum(c[3] or 0 for c in cs)
        tt = sum(c[1] - c[0] for c in cs)
        print("  copy %-22s %5d copies, avg %.3f ms, total %.1f MB, %.1f GB/s while copying" % (dr, len(cs), tt / len(cs) / 1e6, tb / 1e6, tb / max(tt, 1)))
if len(sys.argv) > 3:
    w0 = sa[-int(sys.argv[3])][0]
    allev = [(s, e, n, q) for s, e, n, q in ev if s >= w0] + [(s, e, "COPY " + str(dr) + (" %.1fMB" % (b / 1e6) if b else ""), "-") for s, e, dr, b in cp if s >= w0 and (e - s) > 20000]
    for s, e, n, q in sorted(allev):
        print("    %9.3f -> %9.3f  (%7.3f ms)  q%-3s %s" % ((s - w0) / 1e6, (e - w0) / 1e6, (e - s) / 1e6, q, n))
