# A/B of the 160-column family-tile planner rules (CCV_GEMM_F160=0 disables them): kernel-time totals of rocprofv3 traces
# of bench.py, all arms on one box in one call.  Run on the GPU box: bash tools/ab_f160.sh
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/f160; cd /tmp; export TMPDIR=/tmp
for lvl in 0 1 0 1; do
  for rep in a; do
    CCV_GEMM_F160=$lvl timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f160/raw -- python $R/bench.py --steps 2 --warmup 1 > /dev/null 2>&1 || exit 1
    python $R/tools/summarize_rocprof.py $R/gpurun_out/f160/raw $R/gpurun_out/f160/stats_${lvl}_$RANDOM.txt --delete > /dev/null || exit 1
  done
done
python $R/tools/sum_gemm_time.py $R/gpurun_out/f160/stats_*.txt
