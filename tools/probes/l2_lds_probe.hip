// Diagnostic (not part of the library): how fast do the GEMM kernels' operand tiles arrive in LDS when nothing else
// happens?  Replays the LDS-DMA traffic of a tiled GEMM (tile BM x BN, workgroup (i, j) streams its A and W strips slab by
// slab into an ST-deep LDS ring behind counted s_waitcnt vmcnt + s_barrier) without MFMA or LDS reads.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_lds_probe tools/probes/l2_lds_probe.hip && /tmp/l2_lds_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// RB: bytes of one row piece (128 = 8 rows x 128 B per wave-instruction, 64 = 16 rows x 64 B); SWZ: XOR the 16-B chunk index
// with the row (source-side swizzle); ST: ring depth; PER: DMA instructions per wave and slab
template <int RB, int SWZ, int ST, int PER>
__global__ __launch_bounds__(256) void probe(const uint16_t* A, const uint16_t* W, int M, int N, int K, int BM, int BN, int* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RPP = 1024 / RB;          // rows per piece
    constexpr int CPR = RB / 16;            // 16-B chunks per row
    constexpr int SLAB = RB / 2;            // bf16 elements per slab row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane / CPR, lchunk = lane % CPR;
    const int tiles_n = N / BN;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int a_pieces = BM / RPP;          // pieces 0 .. a_pieces-1 are A rows, the rest W rows
    const uint16_t* src[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int piece = 4 * j + wave;
        const int row = (piece < a_pieces ? piece : piece - a_pieces) * RPP + lrow;
        const int chunk = SWZ ? (lchunk ^ ((row >> 1) & (CPR - 1))) : lchunk;
        src[j] = (piece < a_pieces ? A + (long)(m0 + row) * K : W + (long)(n0 + row) * K) + chunk * 8;
    }
    const int nslab = K / SLAB;
    const int stage_bytes = (BM + BN) * RB;
    auto issue = [&](int s) {
        unsigned char* dst = smem + (s % ST) * stage_bytes;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t*)src[j], (lptr_t*)(dst + (4 * j + wave) * 1024), 16, 0, 0);
            src[j] += SLAB;
        }
    };
    for (int s = 0; s < ST - 1 && s < nslab; ++s) issue(s);
    for (int s = 0; s < nslab; ++s) {
        if (s + ST - 1 < nslab) { issue(s + ST - 1); wait_vm<PER * (ST - 1)>(); }
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
    }
    if (smem[threadIdx.x * 16] == 0x5a && smem[4096 + threadIdx.x] == 0xa5 && sink) atomicAdd(sink, 1);
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int RB, int SWZ, int ST, int PER>
void run(const char* name, const uint16_t* A, const uint16_t* W, int M, int N, int K, int BM, int BN, int* sink) {
    if (PER * 4 * (1024 / RB) != BM + BN) { printf("  %-28s skipped (PER mismatch)\n", name); return; }
    const int tiles = (M / BM) * (N / BN);
    const size_t lds = (size_t)ST * (BM + BN) * RB;
    auto k = probe<RB, SWZ, ST, PER>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), lds, 0, A, W, M, N, K, BM, BN, sink);
    CHECK(hipDeviceSynchronize());
    const int iters = 30;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), lds, 0, A, W, M, N, K, BM, BN, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    const double bytes = (double)tiles * (BM + BN) * K * 2.0;
    printf("  %-28s LDS %3zu KiB  %7.1f us  %6.2f TB/s into LDS  (%.0f MB)\n", name, lds >> 10, us, bytes / us / 1e6, bytes / 1e6);
}

int main() {
    struct Shape { int M, N, K, BM, BN; } shapes[] = {{32768, 960, 320, 128, 64}, {32768, 320, 320, 128, 64}, {8192, 640, 640, 128, 64},
                                                      {2048, 1280, 1280, 64, 64}, {32768, 960, 320, 128, 160}, {32768, 320, 1280, 128, 160}};
    int* sink;
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(sink, 0, 4));
    for (const Shape& s : shapes) {
        std::vector<uint16_t> h((size_t)(s.M + s.N) * s.K);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(rand() & 0x3fff);
        uint16_t *A, *W;
        CHECK(hipMalloc(&A, (size_t)s.M * s.K * 2)); CHECK(hipMalloc(&W, (size_t)s.N * s.K * 2));
        CHECK(hipMemcpy(A, h.data(), (size_t)s.M * s.K * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(W, h.data() + (size_t)s.M * s.K, (size_t)s.N * s.K * 2, hipMemcpyHostToDevice));
        printf("M=%d N=%d K=%d tile %dx%d (%d workgroups)\n", s.M, s.N, s.K, s.BM, s.BN, (s.M / s.BM) * (s.N / s.BN));
        if (s.BM + s.BN == 192) {
            run<128, 1, 2, 6>("128B rows swz  2 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
            run<128, 0, 2, 6>("128B rows lin  2 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
            run<128, 1, 3, 6>("128B rows swz  3 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
            run<128, 1, 4, 6>("128B rows swz  4 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
            run<64, 1, 4, 3>("64B rows swz   4 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
            run<64, 1, 8, 3>("64B rows swz   8 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
        } else if (s.BM + s.BN == 128) {
            run<128, 1, 2, 4>("128B rows swz  2 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
            run<128, 1, 4, 4>("128B rows swz  4 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
            run<64, 1, 8, 2>("64B rows swz   8 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
        } else {   // 128 x 160: 288 rows
            run<128, 1, 2, 9>("128B rows swz  2 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
            run<128, 1, 3, 9>("128B rows swz  3 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
            run<64, 1, 2, 5>("64B rows swz   2 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
            run<64, 1, 4, 5>("64B rows swz   4 stages", A, W, s.M, s.N, s.K, s.BM, s.BN, sink);
        }
        CHECK(hipFree(A)); CHECK(hipFree(W));
    }
    return 0;
}
