#!/bin/bash
# Build (here or on the GPU box) and run tools/mix_probe.hip: the instruction mix of the attention step without memory or barriers.
#   tools/mix_probe.sh [steps]        (gpurun -- 'tools/mix_probe.sh > gpurun_out/mix_probe.txt')
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
if [ ! -x tools/bin/mix_probe ] || [ tools/mix_probe.hip -nt tools/bin/mix_probe ] || [ camc2v_amd/csrc/ccv_attn.hip -nt tools/bin/mix_probe ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Iinclude -o tools/bin/mix_probe tools/mix_probe.hip
fi
exec tools/bin/mix_probe "${1:-20000}"
