#!/bin/bash
# The round's rocprofv3 evidence of bench.py, all from ONE gpurun call (one box): kernel traces of the hipGraph benchmark with one
# and two clips in flight, then the three PMC passes (each its own run, kernel trace only beside the counters) of one eager clip.
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r03'
# Summaries land in gpurun_out/<tag>_prof/ ; copy what is to be judged into profiles/.
set -eo pipefail
tag=${1:-r04}
out=gpurun_out/${tag}_prof
mkdir -p "$out"
export TMPDIR=/tmp
root=$(pwd)

trace() {            # name, bench flags...
    local name=$1; shift
    rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_$name -- python3 bench.py "$@" > "$out/$name.log" 2>&1
    python3 tools/summarize_rocprof.py /tmp/rp_$name "$out/${tag}_rocprofv3_kernel_stats_$name.txt" --gaps --delete > /dev/null
    grep -h "^{" "$out/$name.log" | tail -n 1 > "$out/${tag}_bench_under_rocprof_$name.json"
    echo "done $name"
}
pmc() {              # name, counters...
    local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/rp_$name -- python3 bench.py --steps 1 --warmup 1 --no-graph --clips-only --lanes 1 \
        > "$out/pmc_$name.log" 2>&1
    python3 tools/summarize_rocprof.py /tmp/rp_$name "$out/${tag}_rocprofv3_pmc_${name}_bench_eager.txt" --delete > /dev/null
    echo "done pmc $name"
}

trace bench_graph --steps 4 --warmup 1 --lanes 1 --clips-only
trace bench_graph_2lanes --steps 4 --warmup 2 --lanes 2 --clips-only
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc MFMA_BUSY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
python3 tools/mfma_busy_table.py "$out/${tag}_rocprofv3_pmc_MFMA_BUSY_bench_eager.txt" "$out/${tag}_mfma_busy_table.txt" > /dev/null
python3 tools/traffic_from_pmc.py "$out/${tag}_rocprofv3_pmc_FETCH_SIZE_bench_eager.txt" "$out/${tag}_rocprofv3_pmc_WRITE_SIZE_bench_eager.txt" 2 "$out/${tag}_traffic.json" > /dev/null
cd "$root"
python3 bench.py > "$out/${tag}_bench_default.json" 2> "$out/bench_default.err"
tail -c 3000 "$out/${tag}_bench_default.json"
