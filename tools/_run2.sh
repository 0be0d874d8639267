set -e
for S in 3 4 5; do echo "== ring depth $S"; CCV_ATTN_SHARED_S=$S SPARSE_PROBE_VARIANTS=5,4 timeout -k 10 200 python tools/sparse_probe.py 20; done > gpurun_out/probe2.log 2>&1
cat gpurun_out/probe2.log
SPARSE_PROBE_VARIANTS=5 timeout -k 10 500 bash tools/pmc_kernel.sh attn_shared gpurun_out/pmc_shared python3 tools/sparse_probe.py 3 > gpurun_out/pmc_shared.log 2>&1
SPARSE_PROBE_VARIANTS=6 timeout -k 10 500 bash tools/pmc_kernel.sh attn_sparse gpurun_out/pmc_sparse python3 tools/sparse_probe.py 3 > gpurun_out/pmc_sparse.log 2>&1
