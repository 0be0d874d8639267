#!/usr/bin/env python
"""Matrix-pipe utilisation per kernel from a `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES` pass
(tools/summarize_rocprof.py summary of it; tools/collect_profiles.sh makes one over an eager clip of bench.py).

SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the chip's 1024 SIMDs, the cycles a SIMD's matrix pipe is busy (16 per
v_mfma_f32_16x16x32_bf16, 32 per 32x32x16: MI355X_MICROARCH.md, cycle table), so
    utilisation = MFMA_BUSY / (1024 SIMDs x kernel time x clock)
and, independent of the clock, MFMA_BUSY x 1024 flop/cycle/SIMD = the bf16 flops the kernel executed (a 16x16x32 MFMA is 16384
flops in 16 cycles).  The second column group uses SQ_BUSY_CYCLES as rocprofv3 reports it (summed over its SQ instances).
    python tools/mfma_busy_table.py <summary.txt> <out.txt> [clock_GHz=2.4]"""
import re
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    ghz = float(sys.argv[3]) if len(sys.argv) > 3 else 2.4
    dur, cnt, section = {}, {}, None
    for line in open(src):
        if line.startswith("# kernel trace"):
            section = "trace"
            continue
        m = re.match(r"# counter (\w+):", line)
        if m:
            section = m.group(1)
            cnt[section] = {}
            continue
        if line.startswith("#") or line.startswith("kernel "):
            continue
        if section == "trace":
            f = line.rsplit(None, 6)
            if len(f) == 7:
                dur[f[0].strip()] = (int(f[1]), float(f[2]))            # calls, total ms
        elif section:
            f = line.rsplit(None, 3)
            if len(f) == 4:
                cnt[section][f[0].strip()] = float(f[2])
    mf, busy = cnt.get("SQ_VALU_MFMA_BUSY_CYCLES", {}), cnt.get("SQ_BUSY_CYCLES", {})
    rows = []
    for k, (calls, ms) in dur.items():
        if k in mf and mf[k] > 0:
            util = mf[k] / (1024 * ms * 1e-3 * ghz * 1e9)
            rows.append((ms, k, calls, mf[k], busy.get(k, 0.0), util))
    rows.sort(reverse=True)
    tot_ms = sum(r[0] for r in rows)
    tot_mf = sum(r[3] for r in rows)
    all_ms = sum(ms for _, ms in dur.values())
    out = [f"# matrix-pipe utilisation per kernel, {src}",
           f"# utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x {ghz} GHz); executed bf16 flops = MFMA_BUSY x 1024",
           f"# kernels with MFMAs: {tot_ms:.1f} ms of {all_ms:.1f} ms kernel time in the run; MFMA_BUSY total {tot_mf:.4e} => {tot_mf * 1024 / 1e12:.1f} TF executed, "
           f"{tot_mf / (1024 * tot_ms * 1e-3 * ghz * 1e9):.3f} of the matrix pipe over those kernels, {tot_mf / (1024 * all_ms * 1e-3 * ghz * 1e9):.3f} over all kernel time",
           f"{'kernel':72s} {'calls':>6s} {'total_ms':>9s} {'MFMA_BUSY':>12s} {'SQ_BUSY':>12s} {'mfma/sq_busy':>12s} {'TFLOP/s':>8s} {'pipe_util':>9s}"]
    for ms, k, calls, m, b, util in rows:
        out.append(f"{k[:72]:72s} {calls:6d} {ms:9.2f} {m:12.4e} {b:12.4e} {(m / b if b else 0):12.3f} {m * 1024 / (ms * 1e-3) / 1e12:8.0f} {util:9.3f}")
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:14]))


if __name__ == "__main__":
    main()
