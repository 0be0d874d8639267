// Write-burst rate of a GEMM epilogue's store pattern against fully coalesced stores: 512 workgroups x 4 waves each write a
// 128 x 160 bf16 tile (40 KB) of a [32768, 320] row-major matrix, all at once (what the end of a one-round GEMM launch does).
//   pattern 0: the MFMA fragment layout with the pair swap -- per store instruction 16 rows x 64 contiguous bytes (16 B per lane)
//   pattern 1: the same without the swap -- 16 rows x 32 bytes (8 B per lane), two instructions
//   pattern 2: rows through LDS first: per store instruction 3 whole 320-byte rows of the tile (16 B per lane, 20 lanes per row)
//   hipcc --offload-arch=gfx950 -O2 store_pattern_probe.hip -o store_probe.bin && ./store_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int PATTERN>
__global__ __launch_bounds__(256) void store_tile(uint16_t* C, int ldc, int reps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, fr = lane & 15, fg = lane >> 4;
    const int tiles_n = 2, m0 = (blockIdx.x / tiles_n) * 128, n0 = (blockIdx.x % tiles_n) * 160;
    const uint4 v = make_uint4(tid, tid + 1, tid + 2, tid + 3);
    for (int rep = 0; rep < reps; ++rep) {
        if (PATTERN == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 5; j += 2) {      // fragment pairs (j, j+1): lanes fg = 0..3 cover 64 contiguous bytes of the row
                    const int m = m0 + wm * 64 + 16 * i + fr;
                    if (j + 1 < 5) *reinterpret_cast<uint4*>(C + (long)m * ldc + n0 + wn * 80 + 16 * j + 8 * fg) = v;
                    else *reinterpret_cast<uint2*>(C + (long)m * ldc + n0 + wn * 80 + 16 * j + 4 * fg) = make_uint2(v.x, v.y);
                }
        } else if (PATTERN == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int m = m0 + wm * 64 + 16 * i + fr;
                    *reinterpret_cast<uint2*>(C + (long)m * ldc + n0 + wn * 80 + 16 * j + 4 * fg) = make_uint2(v.x, v.y);
                }
        } else {
            // 128 rows x 320 B = 2560 16-byte chunks per tile, 256 threads: 10 chunks per thread, chunk c -> row c / 20, column 8 (c % 20)
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const int c = k * 256 + tid, r = c / 20, col = 8 * (c % 20);
                *reinterpret_cast<uint4*>(C + (long)(m0 + r) * ldc + n0 + col) = v;
            }
        }
    }
}

template <int PATTERN>
float run(uint16_t* C, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(store_tile<PATTERN>, dim3(512), dim3(256), 0, 0, C, 320, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(store_tile<PATTERN>, dim3(512), dim3(256), 0, 0, C, 320, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}

int main() {
    uint16_t* C;
    hipMalloc(&C, 32768l * 320 * 2);
    for (int reps : {1, 4}) {
        const float t0 = run<0>(C, reps), t1 = run<1>(C, reps), t2 = run<2>(C, reps);
        const double mb = 32768.0 * 320 * 2 * reps / 1e6;
        printf("reps %d (%.0f MB per launch): 16 rows x 64 B per instruction %.1f us (%.2f TB/s) | 16 rows x 32 B %.1f us (%.2f TB/s) | 3 whole rows %.1f us (%.2f TB/s)\n",
               reps, mb, t0 * 1e3, mb / t0 / 1e3, t1 * 1e3, mb / t1 / 1e3, t2 * 1e3, mb / t2 / 1e3);
    }
    return 0;
}
