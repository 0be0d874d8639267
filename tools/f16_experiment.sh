#!/bin/bash
# Round 4: fp16 MFMA operands (CCV_OPERANDS=f16, libccv_hip_f16.so) against the default bf16 build on one box: the parity numbers that carry
# the stated tolerances (medium fixture vs the reference, 25-step medium trajectory vs the reference's sampler + UNet, full-size CFG step vs
# the oracle) and frames/s with two clips in flight and one at a time.   gpurun --timeout 1190 -- 'bash tools/f16_experiment.sh'
out=gpurun_out/f16
mkdir -p $out
for mode in f16 bf16; do
  export CCV_OPERANDS=$mode
  timeout -k 10 600 python -m pytest tests/test_unet_gpu.py tests/test_trajectory_gpu.py -q -m gpu -s -k "medium_fixture or trajectory or error_is_bf16 or no_camera_per_frame or camera_repeat" > $out/parity_$mode.log 2>&1
  echo "== $mode small/medium rc=$?"; grep "\[parity\]\|passed\|failed\|Error" $out/parity_$mode.log | tail -30
done
export CCV_OPERANDS=f16
timeout -k 10 700 python -m pytest tests/test_fullsize_oracle_gpu.py -q -m gpu -s -k "cfg_step" > $out/fullsize_f16.log 2>&1
echo "== f16 full size rc=$?"; grep "\[parity\]\|passed\|failed\|Error" $out/fullsize_f16.log | tail
unset CCV_OPERANDS
bash tools/ab_bench.sh CCV_OPERANDS "bf16 f16 bf16 f16" 2 6 > $out/ab_2lanes.log 2>&1; cat $out/ab_2lanes.log
bash tools/ab_bench.sh CCV_OPERANDS "bf16 f16" 1 4 > $out/ab_1lane.log 2>&1; cat $out/ab_1lane.log
