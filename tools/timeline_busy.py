#!/usr/bin/env python
"""How busy is the GPU with two clips in flight?  Reads a rocprofv3 kernel trace (CSV) of `bench.py --clips-only` and reports, over
the span between the first and the last attn_sparse_kernel of the run's second half (steady state): the fraction of wall time with
0 / 1 / >= 2 kernels running, and the time-weighted mean number of kernels in flight.
    python tools/timeline_busy.py <rocprof_out_dir>"""
import csv
import glob
import os
import sys

traces = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for t in traces:
    for r in csv.DictReader(open(t)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
mid = t0 + (t1 - t0) // 2
sp = [r for r in rows if ("attn_sparse" in r[2] or "attn_shared" in r[2]) and r[0] >= mid]
a, b = sp[0][0], sp[-1][1]
ev = []
for s, e, _ in rows:
    if e <= a or s >= b:
        continue
    ev.append((max(s, a), 1))
    ev.append((min(e, b), -1))
ev.sort()
hist, cur, last = {}, 0, a
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += d
    last = t
hist[cur] = hist.get(cur, 0) + (b - last)
span = b - a
print(f"span {span / 1e6:.1f} ms, {len(ev) // 2} kernels")
for k in sorted(hist):
    print(f"  {k} kernels in flight: {100 * hist[k] / span:5.1f} % of the time")
print(f"  mean kernels in flight: {sum(k * v for k, v in hist.items()) / span:.2f}")
