# Wall-clock A/B of an environment switch on one box: bench.py (clips only) with the arms alternating; prints frames/s per run.
#   bash tools/ab_bench.sh VAR "0 1 0 1" [lanes] [steps]        (run on the GPU box)
VAR=$1; ARMS=${2:-"0 1 0 1"}; LANES=${3:-2}; STEPS=${4:-4}
R=${GRAFT_REPO_ROOT:-.}; mkdir -p $R/gpurun_out/ab
for v in $ARMS; do
  export $VAR=$v
  timeout -k 10 300 python $R/bench.py --steps $STEPS --warmup 2 --lanes $LANES --clips-only > $R/gpurun_out/ab/last.log 2>&1 || { tail -5 $R/gpurun_out/ab/last.log; exit 1; }
  python - "$VAR" "$v" "$LANES" $R/gpurun_out/ab/last.log <<'PY'
import json, sys
var, v, lanes, path = sys.argv[1:5]
d = json.loads([x for x in open(path) if x.startswith("{")][-1])
print(f"{var}={v} lanes={lanes}: {d['value']:.2f} frames/s ({d['ms_per_step']:.1f} ms per clip)", flush=True)
PY
done
