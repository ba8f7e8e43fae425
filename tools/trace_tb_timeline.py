#!/usr/bin/env python3
"""Timeline of the LAST fe_offline call in a rocprofv3 kernel trace: start / end (us, relative), queue, kernel, grid - and how much
of the call's span had 1, 2, 3 ... kernels in flight."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "tb_" in r["Kernel_Name"] or "istft_ola" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one call = everything up to and including an istft_ola launch
calls, cur = [], []
for r in rows:
    cur.append(r)
    if "istft_ola" in r["Kernel_Name"]:
        calls.append(cur)
        cur = []
call = calls[-1]
t0 = int(call[0]["Start_Timestamp"])
print(f"# {len(calls)} calls in the trace; the last one: {len(call)} launches")
for r in call:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("<")[0].split("::")[-1].split("(")[0]
    grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f} us  q={r.get('Queue_Id', '?'):>3s} grid={grid:>7s}  {name}")
ev = []
for r in call:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
hist, n, last = {}, 0, ev[0][0]
for t, d in ev:
    hist[n] = hist.get(n, 0) + (t - last)
    n += d
    last = t
span = ev[-1][0] - ev[0][0]
print("# kernels in flight -> share of the call's span (%.1f us)" % (span / 1e3))
for k in sorted(hist):
    print(f"#   {k}: {hist[k] / span * 100:5.1f} %")
