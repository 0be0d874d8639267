# In-model A/B of an environment switch: kernel-time totals of rocprofv3 traces of bench.py, arms alternating on one box.
#   bash tools/ab_env.sh VAR "0 1 0 1"        (run on the GPU box)
VAR=$1; ARMS=${2:-"0 1 0 1"}
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/ab_$VAR; cd /tmp; export TMPDIR=/tmp
n=0
for v in $ARMS; do
  n=$((n+1))
  env $VAR=$v true
  export $VAR=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$VAR/raw -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  python $R/tools/summarize_rocprof.py $R/gpurun_out/ab_$VAR/raw $R/gpurun_out/ab_$VAR/stats_${n}_${VAR}_$v.txt --delete > /dev/null || exit 1
done
python $R/tools/sum_gemm_time.py $R/gpurun_out/ab_$VAR/stats_*.txt
