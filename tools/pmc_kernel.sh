#!/bin/bash
# SQ counter breakdown of ONE kernel: where its waves' cycles go.  Two rocprofv3 --pmc passes (8 SQ slots each) over a command, then the
# per-kernel sums of the kernels whose name contains the pattern.
#   gpurun -- 'bash tools/pmc_kernel.sh ff_fused gpurun_out/pmc_ff python3 tools/fused_probe.py ff'
set -eo pipefail
pat=$1; out=$2; shift 2
export TMPDIR=/tmp
mkdir -p "$out"
pass() {
    local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/rp_$name -- "${CMD[@]}" > "$out/$name.log" 2>&1 || { tail -20 "$out/$name.log"; return 1; }
    python3 - "$pat" /tmp/rp_$name "$out/$name.txt" <<'PY'
import csv, glob, sys, collections
pat, d, dst = sys.argv[1:4]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in files:
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
with open(dst, "w") as o:
    for k in sorted(tot):
        line = f"{k:32s} sum {tot[k]:.4e}  per dispatch {tot[k] / max(n[k], 1):.4e}  ({n[k]} dispatches)"
        print(line); o.write(line + "\n")
PY
}
CMD=("$@")
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
pass b SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA
