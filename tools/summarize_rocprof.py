#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel trace, kernel stats, PMC counter collection) into small text
summaries that can be committed under profiles/.

    python tools/summarize_rocprof.py <rocprof_out_dir> <summary.txt> [--delete] [--by-grid KERNEL_SUBSTRING]

--by-grid adds a per-launch-grid breakdown of one kernel (one row per distinct workgroup count), so that a single
problem shape of a templated kernel (e.g. the QKV projection bench.py times live) can be read out of the trace.
"""
import collections
import csv
import glob
import os
import shutil
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:90]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = []
    traces = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    counters = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    if traces:
        agg = collections.defaultdict(lambda: [0, 0, 10 ** 18, 0])
        by_grid = collections.defaultdict(lambda: [0, 0, 10 ** 18, 0])
        grid_of = sys.argv[sys.argv.index("--by-grid") + 1] if "--by-grid" in sys.argv else None
        t_first, t_last = 10 ** 30, 0
        for r in csv.DictReader(open(traces[0])):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a = agg[short(r["Kernel_Name"])]
            if grid_of and grid_of in short(r["Kernel_Name"]) and "Grid_Size_X" in r:
                wgs = int(r["Grid_Size_X"]) // max(int(r.get("Workgroup_Size_X", 1) or 1), 1)
                g = by_grid[(short(r["Kernel_Name"]), wgs)]
                g[0] += 1
                g[1] += d
                g[2] = min(g[2], d)
                g[3] = max(g[3], d)
            a[0] += 1
            a[1] += d
            a[2] = min(a[2], d)
            a[3] = max(a[3], d)
            t_first, t_last = min(t_first, int(r["Start_Timestamp"])), max(t_last, int(r["End_Timestamp"]))
        tot = sum(a[1] for a in agg.values())
        out.append(f"# kernel trace: {sum(a[0] for a in agg.values())} dispatches, {tot / 1e6:.2f} ms of kernel time, "
                   f"{(t_last - t_first) / 1e6:.2f} ms first-start to last-end")
        out.append(f"{'kernel':92s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            out.append(f"{k:92s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.1f} {a[2] / 1e3:9.1f} {a[3] / 1e3:9.1f} {100 * a[1] / tot:6.2f}")
        if "--gaps" in sys.argv:   # idle time between kernels inside the busiest stretch of the trace (the timed loop)
            iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(traces[0])))
            segs, cur = [], [iv[0]]
            for x in iv[1:]:             # stretches separated by more than 2 ms of idle device
                if x[0] - max(e for _, e in cur[-8:]) > 2_000_000:
                    segs.append(cur)
                    cur = []
                cur.append(x)
            segs.append(cur)
            iv = max(segs, key=len)
            busy_end, idle, hist, launches = iv[0][1], 0, collections.Counter(), len(iv)
            for a, b in iv[1:]:
                if a > busy_end:
                    g = a - busy_end
                    idle += g
                    hist["<2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else "20-100us" if g < 100000 else ">=100us"] += g
                busy_end = max(busy_end, b)
            span = busy_end - iv[0][0]
            out.append(f"# gaps inside the busiest stretch: {launches} dispatches, span {span / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms "
                       f"({100 * idle / span:.1f} %), {idle / max(launches - 1, 1) / 1e3:.2f} us per dispatch")
            out.append("# idle time by gap size: " + ", ".join(f"{k} {v / 1e6:.2f} ms" for k, v in sorted(hist.items())))
        if traces and by_grid:
            out.append(f"# per-grid breakdown of kernels matching {grid_of!r} (workgroups per launch)")
            out.append(f"{'kernel':60s} {'workgroups':>10s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
            for (k, wgs), g in sorted(by_grid.items(), key=lambda kv: -kv[1][1]):
                out.append(f"{k:60s} {wgs:10d} {g[0]:7d} {g[1] / 1e6:10.3f} {g[1] / g[0] / 1e3:9.1f} {g[2] / 1e3:9.1f} {g[3] / 1e3:9.1f}")
    if counters:
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.defaultdict(int)
        for r in csv.DictReader(open(counters[0])):
            k = short(r["Kernel_Name"])
            per[r["Counter_Name"]][k] += float(r["Counter_Value"])
            calls[(r["Counter_Name"], k)] += 1
        for cname, d in per.items():
            tot = sum(d.values())
            out.append(f"# counter {cname}: total {tot:.6e} over {sum(calls[(cname, k)] for k in d)} dispatches")
            for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:40]:
                out.append(f"{k:92s} {calls[(cname, k)]:7d} {v:16.6e} {v / max(calls[(cname, k)], 1):14.4e}")
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    open(dst, "w").write("\n".join(out) + "\n")
    if "--delete" in sys.argv:
        shutil.rmtree(src, ignore_errors=True)
    print("\n".join(out[:12]))


if __name__ == "__main__":
    main()
