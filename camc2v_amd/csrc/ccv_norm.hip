// GroupNorm(32)(+SiLU) and LayerNorm over token-major activations (gfx950).
//
// Both are HBM/L2-bound streaming passes: reads are fully coalesced along the channel axis
// (a row of C channels is contiguous), statistics are fp32.
//
// GroupNorm runs as two launches (no atomics, fixed summation tree => bitwise reproducible):
//   gn_stats grid (chunks, instances): per-chunk per-group (sum, sumsq) partials
//   gn_apply grid (chunks, instances): partials -> (mean, rstd), then y = (x-mean)*rstd*gamma+beta [, SiLU] -> bf16
// Every load is a 16-byte (4-channel) access; all reductions run in a fixed order, so the result is
// bitwise reproducible (channels-per-group is even for every layer here, so a channel pair never
// straddles two groups).
#include "ccv_common.h"

namespace {

constexpr int GN_GROUPS = 32;
constexpr int GN_MAX_CHUNKS = 128;
constexpr int GN_MAX_PARTS = 512;     // slots per instance ccv_groupnorm_apply_parts accepts (statistics from GEMM epilogues)
constexpr int GN_UNROLL = 8;       // rows a thread has in flight per batch (memory-level parallelism)

// threads per block: a multiple of C/4 (each thread owns 4 fixed channels), <= 1024, >= 256
inline int gn_threads(int C) {
    const int cols = C / 4;
    int r = (256 + cols - 1) / cols;   // at least 256 threads (the statistics prologue uses 256)
    if (r < 1) r = 1;
    return cols * r;
}
// chunks per instance: aim at ~1024 workgroups in total, at least one full batch of rows per thread
inline int gn_chunks(int instances, int rows, int C) {
    const int R = gn_threads(C) / (C / 4);
    int c = (1024 + instances - 1) / instances;
    const int by_rows = (rows + GN_UNROLL * R - 1) / (GN_UNROLL * R);
    if (c > by_rows) c = by_rows;
    if (c > GN_MAX_CHUNKS) c = GN_MAX_CHUNKS;
    return c < 1 ? 1 : c;
}

template <int XK>      // CCV_BF16 / CCV_F32 / CCV_F16
__device__ __forceinline__ float4 load4(const void* x, long idx4) { return ccv_load4<XK>(x, idx4); }

// grid (nchunk, instances), gn_threads(C) threads.  Thread (roff, col) owns the 4 channels of float4 column
// `col` on rows r0+roff, r0+roff+R, ...; GN_UNROLL row loads are issued back to back before any is consumed
// (the pass is latency-bound otherwise: a chunk is only a few dozen rows).  Its partial sums belong to two
// fixed channel pairs; the block reduces over roff and over the pairs of each group in a FIXED order.
template <int X_F32>
__global__ __launch_bounds__(1024) void gn_stats(const void* x, float* partial, int rows_per_instance, int C) {
    __shared__ __attribute__((aligned(16))) float gsm[1024 * 4];  // [R][cols] x (sum01, sq01, sum23, sq23)
    const int inst = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int cols = C >> 2;
    const int col = threadIdx.x % cols, roff = threadIdx.x / cols, R = blockDim.x / cols;
    const int rows_per_chunk = (rows_per_instance + nchunk - 1) / nchunk;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(rows_per_instance, r0 + rows_per_chunk);
    const long base = (long)inst * rows_per_instance * cols + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int r = r0 + roff; r < r1; r += GN_UNROLL * R) {
        float4 v[GN_UNROLL];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
            const int rr = r + u * R;
            v[u] = (rr < r1) ? load4<X_F32>(x, base + (long)rr * cols) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
            a0 += v[u].x + v[u].y; a1 += v[u].x * v[u].x + v[u].y * v[u].y;
            a2 += v[u].z + v[u].w; a3 += v[u].z * v[u].z + v[u].w * v[u].w;
        }
    }
    *reinterpret_cast<float4*>(gsm + (roff * cols + col) * 4) = make_float4(a0, a1, a2, a3);
    __syncthreads();
    if (threadIdx.x < GN_GROUPS * 2) {
        const int g = threadIdx.x >> 1, k = threadIdx.x & 1;
        const int ppg = (C / GN_GROUPS) >> 1;  // channel pairs per group (channels per group is even)
        float a = 0.f;
        for (int i = 0; i < ppg; ++i) {
            const int pr = g * ppg + i;        // pair -> float4 column pr/2, half pr&1
            float t = 0.f;
            for (int ro = 0; ro < R; ++ro) t += gsm[(ro * cols + (pr >> 1)) * 4 + (pr & 1) * 2 + k];
            a += t;
        }
        partial[((long)inst * nchunk + chunk) * GN_GROUPS * 2 + threadIdx.x] = a;
    }
}

// grid (nchunk, instances), gn_threads(C) threads.  Prologue: the block reduces the chunk partials of its
// instance to (mean, rstd) per group in a fixed order (4 threads per (group, moment), all loads of a thread in
// flight together; every block of an instance computes the same bits), which saves a separate finalize launch
// per GroupNorm; then y = (x-mean)*rstd*gamma+beta [SiLU], GN_UNROLL/2 rows in flight per thread.
template <int X_F32>
__global__ __launch_bounds__(1024) void gn_apply(const void* x, uint16_t* y, const float* gamma, const float* beta,
                                                 const float* partial, int rows_per_instance, int C, int silu,
                                                 float inv_count, float eps, int npart) {   // npart: partial slots per instance (<= 128)
    __shared__ float s_sum[GN_GROUPS * 2];
    const int inst = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int cols = C >> 2;
    const int col = threadIdx.x % cols, roff = threadIdx.x / cols, R = blockDim.x / cols;
    const int rows_per_chunk = (rows_per_instance + nchunk - 1) / nchunk;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(rows_per_instance, r0 + rows_per_chunk);
    const long base = (long)inst * rows_per_instance * cols + col;
    constexpr int U = GN_UNROLL / 2;
    // the first batch of rows and the affine parameters are requested before the statistics prologue, whose
    // partial-sum loads would otherwise add a full memory round trip in front of them
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int rr = r0 + roff + u * R;
        v[u] = (rr < r1) ? load4<X_F32>(x, base + (long)rr * cols) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 gm = reinterpret_cast<const float4*>(gamma)[col];
    const float4 bt = reinterpret_cast<const float4*>(beta)[col];
    if (threadIdx.x < 256) {
        const int gk = threadIdx.x >> 2, j = threadIdx.x & 3;   // gk = 2*group + moment
        const float* pp = partial + (long)inst * npart * GN_GROUPS * 2 + gk;
        float a = 0.f;
        for (int base = 0; base < npart; base += GN_MAX_CHUNKS) {   // one round for the chunked statistics pass (<= 128 slots), up to four
            float pv[GN_MAX_CHUNKS / 4];                            // for statistics from GEMM epilogues (clip-wide norms: <= 512 tiles)
#pragma unroll
            for (int i = 0; i < GN_MAX_CHUNKS / 4; ++i) {
                const int c = base + j + 4 * i;
                pv[i] = (c < npart) ? pp[(long)c * GN_GROUPS * 2] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < GN_MAX_CHUNKS / 4; ++i) a += pv[i];
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (j == 0) s_sum[gk] = a;
    }
    __syncthreads();
    const int cpg = C / GN_GROUPS;
    const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
    float sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int g = (4 * col + k) / cpg;
        const float mean = s_sum[2 * g] * inv_count;
        const float var = fmaxf(s_sum[2 * g + 1] * inv_count - mean * mean, 0.f);
        sc[k] = rsqrtf(var + eps) * gmv[k];
        sh[k] = btv[k] - mean * sc[k];
    }
    uint2* yo = reinterpret_cast<uint2*>(y);
    for (int r = r0 + roff; r < r1; r += U * R) {
        float4 nv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {      // next batch in flight while this one is normalised and stored
            const int rr = r + (U + u) * R;
            nv[u] = (rr < r1) ? load4<X_F32>(x, base + (long)rr * cols) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * R;
            float o0 = v[u].x * sc[0] + sh[0], o1 = v[u].y * sc[1] + sh[1], o2 = v[u].z * sc[2] + sh[2], o3 = v[u].w * sc[3] + sh[3];
            if (silu) { o0 = silu_f(o0); o1 = silu_f(o1); o2 = silu_f(o2); o3 = silu_f(o3); }
            if (rr < r1) yo[base + (long)rr * cols] = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = nv[u];
    }
}

// gn_apply for the two-byte inputs (bf16 conv outputs, the fp16 stream) with EIGHT channels per thread: one 16-byte load and one 16-byte store
// per lane and row.  With 8-byte accesses the kernel ran at the same time for fp32 and for two-byte inputs of the same shape (11.4 / 11.8 us at
// 32768 x 320: 5.5 against 3.5 TB/s) -- bound by instructions per byte, not by bytes.  Same arithmetic per element as gn_apply: identical results.
template <int XK>
__global__ __launch_bounds__(1024) void gn_apply8(const void* x, uint16_t* y, const float* gamma, const float* beta,
                                                  const float* partial, int rows_per_instance, int C, int silu,
                                                  float inv_count, float eps, int npart) {
    static_assert(XK == CCV_BF16 || XK == CCV_F16, "gn_apply8: two-byte inputs");
    __shared__ float s_sum[GN_GROUPS * 2];
    const int inst = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int cols = C >> 3;
    const int col = threadIdx.x % cols, roff = threadIdx.x / cols, R = blockDim.x / cols;
    const int rows_per_chunk = (rows_per_instance + nchunk - 1) / nchunk;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(rows_per_instance, r0 + rows_per_chunk);
    const long base = (long)inst * rows_per_instance * cols + col;       // in 16-byte units
    constexpr int U = GN_UNROLL / 2;
    const uint4* xin = static_cast<const uint4*>(x);
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int rr = r0 + roff + u * R;
        v[u] = (rr < r1) ? xin[base + (long)rr * cols] : make_uint4(0u, 0u, 0u, 0u);
    }
    const float4 gm0 = reinterpret_cast<const float4*>(gamma)[2 * col], gm1 = reinterpret_cast<const float4*>(gamma)[2 * col + 1];
    const float4 bt0 = reinterpret_cast<const float4*>(beta)[2 * col], bt1 = reinterpret_cast<const float4*>(beta)[2 * col + 1];
    if (threadIdx.x < 256) {
        const int gk = threadIdx.x >> 2, j = threadIdx.x & 3;   // gk = 2*group + moment
        const float* pp = partial + (long)inst * npart * GN_GROUPS * 2 + gk;
        float a = 0.f;
        for (int base = 0; base < npart; base += GN_MAX_CHUNKS) {
            float pv[GN_MAX_CHUNKS / 4];
#pragma unroll
            for (int i = 0; i < GN_MAX_CHUNKS / 4; ++i) {
                const int c = base + j + 4 * i;
                pv[i] = (c < npart) ? pp[(long)c * GN_GROUPS * 2] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < GN_MAX_CHUNKS / 4; ++i) a += pv[i];
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (j == 0) s_sum[gk] = a;
    }
    __syncthreads();
    const int cpg = C / GN_GROUPS;
    const float gmv[8] = {gm0.x, gm0.y, gm0.z, gm0.w, gm1.x, gm1.y, gm1.z, gm1.w}, btv[8] = {bt0.x, bt0.y, bt0.z, bt0.w, bt1.x, bt1.y, bt1.z, bt1.w};
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int g = (8 * col + k) / cpg;
        const float mean = s_sum[2 * g] * inv_count;
        const float var = fmaxf(s_sum[2 * g + 1] * inv_count - mean * mean, 0.f);
        sc[k] = rsqrtf(var + eps) * gmv[k];
        sh[k] = btv[k] - mean * sc[k];
    }
    uint4* yo = reinterpret_cast<uint4*>(y);
    for (int r = r0 + roff; r < r1; r += U * R) {
        uint4 nv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {      // next batch in flight while this one is normalised and stored
            const int rr = r + (U + u) * R;
            nv[u] = (rr < r1) ? xin[base + (long)rr * cols] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * R;
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            uint32_t pk[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                float a, b;
                if (XK == CCV_F16) { const float2 f = ccv_unpack_f16x2(w[h]); a = f.x; b = f.y; }
                else { ccv_opnd2_to_f32(w[h], a, b); }
                float o0 = a * sc[2 * h] + sh[2 * h], o1 = b * sc[2 * h + 1] + sh[2 * h + 1];
                if (silu) { o0 = silu_f(o0); o1 = silu_f(o1); }
                pk[h] = pack_bf16x2(o0, o1);
            }
            if (rr < r1) yo[base + (long)rr * cols] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = nv[u];
    }
}


// Single-launch GroupNorm for instances whose per-group slice is small (the 16x16 .. 4x4 latent layers): grid
// (32 / gpb, instances); a workgroup owns `gpb` adjacent groups of one instance (gpb * C/32 channels, >= 128
// contiguous bytes per row), reads its slice once for the statistics and once more (an L1/L2 hit) to normalise.
// No partials, no second launch: at these sizes two launches cost more than the bytes they move.  Fixed-order
// reductions as above (bitwise reproducible).
template <int X_F32>
__global__ __launch_bounds__(256) void gn_small(const void* x, uint16_t* y, const float* gamma, const float* beta,
                                                int rows, int C, int gpb, int silu, float inv_count, float eps) {
    __shared__ __attribute__((aligned(16))) float part[256 * 4];
    __shared__ float colsum[64 * 4];
    __shared__ float s_stat[GN_GROUPS * 2];
    const int inst = blockIdx.y;
    const int cpg = C / GN_GROUPS;
    const int w4 = (cpg * gpb) >> 2;                 // float4 columns of this workgroup's channel span
    const int c4 = threadIdx.x % w4, rr = threadIdx.x / w4, RS = 256 / w4;
    const bool active = rr < RS;
    const int cols = C >> 2;
    const int col = blockIdx.x * w4 + c4;            // float4 column in the full row
    const long base = (long)inst * rows * cols + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (active) {
        for (int r = rr; r < rows; r += GN_UNROLL * RS) {
            float4 v[GN_UNROLL];
#pragma unroll
            for (int u = 0; u < GN_UNROLL; ++u) {
                const int q = r + u * RS;
                v[u] = (q < rows) ? load4<X_F32>(x, base + (long)q * cols) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < GN_UNROLL; ++u) {
                a0 += v[u].x + v[u].y; a1 += v[u].x * v[u].x + v[u].y * v[u].y;
                a2 += v[u].z + v[u].w; a3 += v[u].z * v[u].z + v[u].w * v[u].w;
            }
        }
        *reinterpret_cast<float4*>(part + (rr * w4 + c4) * 4) = make_float4(a0, a1, a2, a3);
    }
    const float4 gm = active ? reinterpret_cast<const float4*>(gamma)[col] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bt = active ? reinterpret_cast<const float4*>(beta)[col] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (threadIdx.x < w4 * 4) {                      // column sums over the RS row slots: part[ro][c4][comp], 8 reads in flight
        float a = 0.f;
        for (int ro = 0; ro < RS; ro += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = (ro + u < RS) ? part[(ro + u) * w4 * 4 + threadIdx.x] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) a += t[u];
        }
        colsum[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x < 2 * gpb) {
        const int g = threadIdx.x >> 1, k = threadIdx.x & 1;
        const int ppg = cpg >> 1;                    // channel pairs per group
        float a = 0.f;
#pragma unroll 4
        for (int i = 0; i < ppg; ++i) {
            const int pr = g * ppg + i;              // pair inside the span -> float4 column pr/2, half pr&1
            a += colsum[(pr >> 1) * 4 + (pr & 1) * 2 + k];
        }
        s_stat[threadIdx.x] = a;
    }
    __syncthreads();
    if (!active) return;
    const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
    float sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int g = (4 * c4 + k) / cpg;            // group inside the span
        const float mean = s_stat[2 * g] * inv_count;
        const float var = fmaxf(s_stat[2 * g + 1] * inv_count - mean * mean, 0.f);
        sc[k] = rsqrtf(var + eps) * gmv[k];
        sh[k] = btv[k] - mean * sc[k];
    }
    uint2* yo = reinterpret_cast<uint2*>(y);
    for (int r = rr; r < rows; r += GN_UNROLL * RS) {
        float4 v[GN_UNROLL];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
            const int q = r + u * RS;
            v[u] = (q < rows) ? load4<X_F32>(x, base + (long)q * cols) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
            const int q = r + u * RS;
            float o0 = v[u].x * sc[0] + sh[0], o1 = v[u].y * sc[1] + sh[1], o2 = v[u].z * sc[2] + sh[2], o3 = v[u].w * sc[3] + sh[3];
            if (silu) { o0 = silu_f(o0); o1 = silu_f(o1); o2 = silu_f(o2); o3 = silu_f(o3); }
            if (q < rows) yo[base + (long)q * cols] = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
        }
    }
}

// groups per workgroup of gn_small (0: use the two-launch path): the span must be a whole number of float4 columns,
// at least 128 bytes per row and at most 64 columns.  Measured on MI355X (tools/bench_kernels.py norm): the single
// launch wins while one pass of a workgroup stays under ~24 KiB and there are >= 128 workgroups (frame-wise norms
// at 8x8 and 4x4 latents: 6-10 us instead of 12-15 us); larger slices are faster through the chunked two-launch path.
inline int gn_small_gpb(int instances, int rows_per_instance, int C, bool x_f32) {   // x_f32: 4-byte elements
    const int cpg = C / GN_GROUPS;
    for (int gpb = 1; gpb <= GN_GROUPS; gpb *= 2) {
        const int span = cpg * gpb;
        if (span % 4 != 0 || span * (x_f32 ? 4 : 2) < 128) continue;
        if (span / 4 > 64) return 0;
        const bool fits = (long)rows_per_instance * span * (x_f32 ? 4 : 2) <= (24l << 10);
        return (fits && (long)instances * (GN_GROUPS / gpb) >= 128) ? gpb : 0;
    }
    return 0;
}

// LPR lanes per row (64/LPR rows per wave); a lane owns float4 columns l + LPR*i, so every load instruction of a
// wave reads 16*LPR contiguous bytes of each of its rows and every store writes half of that.  Two-pass statistics
// on registers (mean, then centred sum of squares), reduced over the LPR lanes with shuffles.  LN_MAX4 = float4
// columns per lane the instance is compiled for (C <= 4 * LPR * LN_MAX4).  The layers here have C = 320 / 640 /
// 1280 at 32768 / 8192 / 2048 rows: LPR = C/20 keeps 5 float4 (20 registers) per lane at every width, 8 waves per
// SIMD resident, and the narrow-and-long as well as the wide-and-short activations spread over all CUs.
template <int LPR, int LN_MAX4, int NR, int XK>     // XK: CCV_F32 / CCV_F16 rows
__global__ __launch_bounds__(256) void ln_kernel(const void* x, uint16_t* y, const float* gamma, const float* beta,
                                                 int rows, int C, float eps, const uint16_t* addend, int addend_rows, uint16_t* y2) {
    // NR rows per lane group: all their loads are in flight together and gamma / beta are fetched once for them
    constexpr int RPB = 256 / LPR * NR;      // rows per block
    const int row0 = blockIdx.x * RPB + (threadIdx.x / LPR) * NR;
    const int l = threadIdx.x % LPR;
    const int cols = C >> 2;                 // float4 columns per row
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
    float4 v[NR][LN_MAX4], gm[LN_MAX4], bt[LN_MAX4];
#pragma unroll
    for (int r = 0; r < NR; ++r) {           // every load of the kernel is issued here, before the first reduction
        const long xr = (long)(row0 + r < rows ? row0 + r : 0) * cols;
#pragma unroll
        for (int i = 0; i < LN_MAX4; ++i) {
            const int c = l + LPR * i;
            v[r][i] = c < cols ? ccv_load4<XK>(x, xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int i = 0; i < LN_MAX4; ++i) {
        const int c = l + LPR * i;
        const bool in = c < cols;
        gm[i] = in ? g4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[i] = in ? b4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float mean[NR], rstd[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX4; ++i) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o, 64);
        mean[r] = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAX4; ++i) {
            const int c = l + LPR * i;
            if (c < cols) {
                const float a = v[r][i].x - mean[r], b = v[r][i].y - mean[r], d = v[r][i].z - mean[r], e = v[r][i].w - mean[r];
                q += (a * a + b * b) + (d * d + e * e);
            }
        }
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) q += __shfl_xor(q, o, 64);
        rstd[r] = rsqrtf(q / (float)C + eps);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int row = row0 + r;
        if (row >= rows) break;
        uint2* yo = reinterpret_cast<uint2*>(y) + (long)row * cols;
        uint2* y2o = y2 ? reinterpret_cast<uint2*>(y2) + (long)row * cols : nullptr;
        const uint2* ad = addend ? reinterpret_cast<const uint2*>(addend) + (long)(row % addend_rows) * cols : nullptr;
#pragma unroll
        for (int i = 0; i < LN_MAX4; ++i) {
            const int c = l + LPR * i;
            if (c < cols) {
                const float4 g = gm[i], bb = bt[i];
                const float o0 = (v[r][i].x - mean[r]) * rstd[r] * g.x + bb.x, o1 = (v[r][i].y - mean[r]) * rstd[r] * g.y + bb.y;
                const float o2 = (v[r][i].z - mean[r]) * rstd[r] * g.z + bb.z, o3 = (v[r][i].w - mean[r]) * rstd[r] * g.w + bb.w;
                yo[c] = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
                if (y2o) {
                    const uint2 a = ad[c];
                    y2o[c] = make_uint2(pack_bf16x2(o0 + bf16_to_f32((uint16_t)(a.x & 0xffffu)), o1 + bf16_to_f32((uint16_t)(a.x >> 16))),
                                        pack_bf16x2(o2 + bf16_to_f32((uint16_t)(a.y & 0xffffu)), o3 + bf16_to_f32((uint16_t)(a.y >> 16))));
                }
            }
        }
    }
}

template <int LPR, int LN_MAX4, int NR>
void launch_ln(hipStream_t st, const void* x, int kind, uint16_t* y, const float* gamma, const float* beta, int rows, int C, float eps,
               const uint16_t* addend, int ar, uint16_t* y2) {
    constexpr int RPB = 256 / LPR * NR;
    if (kind == CCV_F16)
        hipLaunchKernelGGL((ln_kernel<LPR, LN_MAX4, NR, CCV_F16>), dim3((rows + RPB - 1) / RPB), dim3(256), 0, st, x, y, gamma, beta, rows, C, eps, addend, ar, y2);
    else
        hipLaunchKernelGGL((ln_kernel<LPR, LN_MAX4, NR, CCV_F32>), dim3((rows + RPB - 1) / RPB), dim3(256), 0, st, x, y, gamma, beta, rows, C, eps, addend, ar, y2);
}

// fp32 -> fp32 LayerNorm for the small once-per-clip tensors (the adaptor's 4-channel output norm, the Resampler's
// 1024-wide one): one wave per row, lane-strided columns, two-pass statistics; rows are a few thousand at most.
__global__ __launch_bounds__(256) void ln_small_kernel(const float* x, float* y, const float* gamma, const float* beta, long rows, int C,
                                                       long ldx, float eps) {
    const long r = blockIdx.x * 4l + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* xr = x + r * ldx;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    for (int c = lane; c < C; c += 64) y[r * C + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}

// x_f32 of the C ABI is the element kind of x: 0 bf16, 1 fp32, 2 fp16
// gn_apply's launch: the 8-channels-per-thread kernel for two-byte inputs (same grid: the chunk count fixes the layout of the partial sums)
inline int gn_threads8(int C) {
    const int cols = C / 8;
    int r = (256 + cols - 1) / cols;   // at least 256 threads (the statistics prologue uses 256)
    return cols * (r < 1 ? 1 : r);
}
inline bool gn_aligned16(const void* x, const void* y) {      // gn_apply8 moves 16 bytes per lane: x and y on 16-byte boundaries (rows are: C % 8 == 0)
    return ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}
#define GN_FIRST2_(a, b, ...) a, b
#define GN_FIRST2(...) GN_FIRST2_(__VA_ARGS__)
#define GN_APPLY(kind, grid, st, C, ...)                                                                                   \
    do {                                                                                                                   \
        static const bool wide_ = [] { const char* e = getenv("CCV_GN_APPLY8"); return !(e && e[0] == '0'); }();           \
        if ((kind) != CCV_F32 && wide_ && (C) % 8 == 0 && gn_threads8(C) <= 1024 && gn_aligned16(GN_FIRST2(__VA_ARGS__))) {    \
            if ((kind) == CCV_F16) hipLaunchKernelGGL(gn_apply8<CCV_F16>, grid, dim3(gn_threads8(C)), 0, st, __VA_ARGS__); \
            else hipLaunchKernelGGL(gn_apply8<CCV_BF16>, grid, dim3(gn_threads8(C)), 0, st, __VA_ARGS__);                  \
        } else {                                                                                                           \
            GN_DISPATCH(gn_apply, kind, grid, dim3(gn_threads(C)), 0, st, __VA_ARGS__);                                    \
        }                                                                                                                  \
    } while (0)

#define GN_DISPATCH(KERNEL, kind, ...)                                                  \
    do {                                                                                \
        if ((kind) == CCV_F32) hipLaunchKernelGGL(KERNEL<CCV_F32>, __VA_ARGS__);        \
        else if ((kind) == CCV_F16) hipLaunchKernelGGL(KERNEL<CCV_F16>, __VA_ARGS__);   \
        else hipLaunchKernelGGL(KERNEL<CCV_BF16>, __VA_ARGS__);                         \
    } while (0)

}  // namespace

extern "C" int64_t ccv_groupnorm_ws_bytes(int32_t instances, int32_t C) {
    (void)C;
    return (int64_t)instances * (GN_MAX_CHUNKS + 1) * GN_GROUPS * 2 * (int64_t)sizeof(float);
}

extern "C" int ccv_groupnorm(const void* x, int32_t x_f32, uint16_t* y, const float* gamma, const float* beta,
                             int32_t instances, int32_t rows_per_instance, int32_t C, float eps, int32_t silu,
                             void* ws, void* stream) {
    CCV_REQUIRE(x && y && gamma && beta && ws, CCV_EINVAL, "ccv_groupnorm: null pointer");
    CCV_REQUIRE(x_f32 >= 0 && x_f32 <= 2, CCV_EINVAL, "ccv_groupnorm: x_f32 must be 0 (bf16), 1 (fp32) or 2 (fp16)");
    CCV_REQUIRE(instances > 0 && rows_per_instance > 0, CCV_EINVAL, "ccv_groupnorm: non-positive sizes");
    CCV_REQUIRE(instances <= 65535, CCV_ESHAPE, "ccv_groupnorm: too many instances");
    CCV_REQUIRE(C % 64 == 0 && C > 0 && C <= 4096, CCV_ESHAPE, "ccv_groupnorm: C=%d must be a multiple of 64 and <= 4096", C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float inv_n = 1.0f / ((float)rows_per_instance * (float)(C / GN_GROUPS));
    static const bool small_on = [] { const char* e = getenv("CCV_GN_SMALL"); return !(e && e[0] == '0'); }();
    const int gpb = small_on ? gn_small_gpb(instances, rows_per_instance, C, x_f32 == CCV_F32) : 0;
    if (gpb > 0) {
        dim3 grid_s(GN_GROUPS / gpb, instances);
        GN_DISPATCH(gn_small, x_f32, grid_s, dim3(256), 0, st, x, y, gamma, beta, rows_per_instance, C, gpb, silu, inv_n, eps);
        CCV_LAUNCH_CHECK("ccv_groupnorm(small)");
        return CCV_OK;
    }
    const int nchunk = gn_chunks(instances, rows_per_instance, C);
    const int nthreads = gn_threads(C);
    float* partial = static_cast<float*>(ws);
    dim3 grid(nchunk, instances);
    GN_DISPATCH(gn_stats, x_f32, grid, dim3(nthreads), 0, st, x, partial, rows_per_instance, C);
    CCV_LAUNCH_CHECK("ccv_groupnorm(stats)");
    const float inv_count = 1.0f / ((float)rows_per_instance * (float)(C / GN_GROUPS));
    GN_APPLY(x_f32, grid, st, C, x, y, gamma, beta, partial, rows_per_instance, C, silu, inv_count, eps, (int)grid.x);
    CCV_LAUNCH_CHECK("ccv_groupnorm(apply)");
    return CCV_OK;
}

// 1 when ccv_groupnorm would run this problem as ONE launch (gn_small): statistics from a producer's epilogue then save nothing
extern "C" int32_t ccv_groupnorm_single_launch(int32_t instances, int32_t rows_per_instance, int32_t C, int32_t x_kind) {
    if (instances <= 0 || rows_per_instance <= 0 || C <= 0 || C % 64 != 0) return 0;
    static const bool small_on = [] { const char* e = getenv("CCV_GN_SMALL"); return !(e && e[0] == '0'); }();
    return (small_on && gn_small_gpb(instances, rows_per_instance, C, x_kind == CCV_F32) > 0) ? 1 : 0;
}

// The two halves of ccv_groupnorm as separate calls, for statistics that span more rows than this process holds (a clip whose
// frames are sharded over GPUs: the caller sums the per-chunk partials, all-reduces them and hands the totals back).
extern "C" int32_t ccv_groupnorm_chunks(int32_t instances, int32_t rows_per_instance, int32_t C) {
    if (instances <= 0 || rows_per_instance <= 0 || C <= 0 || C % 64 != 0) return 0;
    return gn_chunks(instances, rows_per_instance, C);
}

extern "C" int ccv_groupnorm_stats(const void* x, int32_t x_f32, int32_t instances, int32_t rows_per_instance, int32_t C, void* ws, void* stream) {
    CCV_REQUIRE(x && ws, CCV_EINVAL, "ccv_groupnorm_stats: null pointer");
    CCV_REQUIRE(instances > 0 && instances <= 65535 && rows_per_instance > 0, CCV_EINVAL, "ccv_groupnorm_stats: bad sizes");
    CCV_REQUIRE(C % 64 == 0 && C > 0 && C <= 4096, CCV_ESHAPE, "ccv_groupnorm_stats: C=%d must be a multiple of 64 and <= 4096", C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(gn_chunks(instances, rows_per_instance, C), instances);
    GN_DISPATCH(gn_stats, x_f32, grid, dim3(gn_threads(C)), 0, st, x, static_cast<float*>(ws), rows_per_instance, C);
    CCV_LAUNCH_CHECK("ccv_groupnorm_stats");
    return CCV_OK;
}

extern "C" int ccv_groupnorm_apply(const void* x, int32_t x_f32, uint16_t* y, const float* gamma, const float* beta, int32_t instances,
                                   int32_t rows_per_instance, int32_t C, float eps, int32_t silu, const void* ws, float inv_count, void* stream) {
    CCV_REQUIRE(x && y && gamma && beta && ws, CCV_EINVAL, "ccv_groupnorm_apply: null pointer");
    CCV_REQUIRE(instances > 0 && instances <= 65535 && rows_per_instance > 0 && inv_count > 0.f, CCV_EINVAL, "ccv_groupnorm_apply: bad sizes");
    CCV_REQUIRE(C % 64 == 0 && C > 0 && C <= 4096, CCV_ESHAPE, "ccv_groupnorm_apply: C=%d must be a multiple of 64 and <= 4096", C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(gn_chunks(instances, rows_per_instance, C), instances);
    const float* partial = static_cast<const float*>(ws);
    GN_APPLY(x_f32, grid, st, C, x, y, gamma, beta, partial, rows_per_instance, C, silu, inv_count, eps, (int)grid.x);
    CCV_LAUNCH_CHECK("ccv_groupnorm_apply");
    return CCV_OK;
}

// The normalise half on statistics some other kernel produced (the producing GEMM's epilogue, ccv_gemm with gn_partial): `parts`
// slots of 64 floats per instance, [instances][parts][32 groups][sum, sum of squares], summed here in slot order.
extern "C" int ccv_groupnorm_apply_parts(const void* x, int32_t x_f32, uint16_t* y, const float* gamma, const float* beta, int32_t instances,
                                         int32_t rows_per_instance, int32_t C, float eps, int32_t silu, const void* partial, int32_t parts,
                                         void* stream) {
    CCV_REQUIRE(x && y && gamma && beta && partial, CCV_EINVAL, "ccv_groupnorm_apply_parts: null pointer");
    CCV_REQUIRE(instances > 0 && instances <= 65535 && rows_per_instance > 0, CCV_EINVAL, "ccv_groupnorm_apply_parts: bad sizes");
    CCV_REQUIRE(parts > 0 && parts <= GN_MAX_PARTS, CCV_ESHAPE, "ccv_groupnorm_apply_parts: parts=%d must be in 1..%d", parts, GN_MAX_PARTS);
    CCV_REQUIRE(C % 64 == 0 && C > 0 && C <= 4096, CCV_ESHAPE, "ccv_groupnorm_apply_parts: C=%d must be a multiple of 64 and <= 4096", C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(gn_chunks(instances, rows_per_instance, C), instances);
    const float inv_count = 1.0f / ((float)rows_per_instance * (float)(C / GN_GROUPS));
    const float* pr = static_cast<const float*>(partial);
    GN_APPLY(x_f32, grid, st, C, x, y, gamma, beta, pr, rows_per_instance, C, silu, inv_count, eps, (int)parts);
    CCV_LAUNCH_CHECK("ccv_groupnorm_apply_parts");
    return CCV_OK;
}

extern "C" int ccv_layernorm(const void* x, int32_t x_kind, uint16_t* y, const float* gamma, const float* beta,
                             int32_t rows, int32_t C, float eps, const uint16_t* addend, int32_t addend_rows, uint16_t* y2, void* stream) {
    CCV_REQUIRE(x && y && gamma && beta, CCV_EINVAL, "ccv_layernorm: null pointer");
    CCV_REQUIRE(x_kind == CCV_F32 || x_kind == CCV_F16, CCV_EINVAL, "ccv_layernorm: x_kind must be 1 (fp32) or 2 (fp16)");
    CCV_REQUIRE(rows > 0, CCV_EINVAL, "ccv_layernorm: rows=%d", rows);
    CCV_REQUIRE(C % 64 == 0 && C > 0 && C <= 2048, CCV_ESHAPE, "ccv_layernorm: C=%d must be a multiple of 64 and <= 2048", C);
    CCV_REQUIRE((addend == nullptr) == (y2 == nullptr), CCV_EINVAL, "ccv_layernorm: addend and y2 go together");
    CCV_REQUIRE(!addend || addend_rows > 0, CCV_EINVAL, "ccv_layernorm: addend_rows must be positive");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ar = addend_rows > 0 ? addend_rows : 1;
    // two rows per lane group once that still leaves >= 4 workgroups per CU (A/B aid: CCV_LN_ROWS=1)
    static const int nr_env = [] { const char* e = getenv("CCV_LN_ROWS"); return e ? atoi(e) : 2; }();
    const bool two = nr_env == 2 && (long)rows * (C <= 320 ? 16 : C <= 640 ? 32 : 64) >= 2l * 256 * 1024;
    if (C <= 320)
        two ? launch_ln<16, 5, 2>(st, x, x_kind, y, gamma, beta, rows, C, eps, addend, ar, y2) : launch_ln<16, 5, 1>(st, x, x_kind, y, gamma, beta, rows, C, eps, addend, ar, y2);
    else if (C <= 640)
        two ? launch_ln<32, 5, 2>(st, x, x_kind, y, gamma, beta, rows, C, eps, addend, ar, y2) : launch_ln<32, 5, 1>(st, x, x_kind, y, gamma, beta, rows, C, eps, addend, ar, y2);
    else if (C <= 1280)
        launch_ln<64, 5, 1>(st, x, x_kind, y, gamma, beta, rows, C, eps, addend, ar, y2);
    else
        launch_ln<64, 8, 1>(st, x, x_kind, y, gamma, beta, rows, C, eps, addend, ar, y2);
    CCV_LAUNCH_CHECK("ccv_layernorm");
    return CCV_OK;
}

extern "C" int ccv_layernorm_small(const float* x, float* y, const float* gamma, const float* beta, int64_t rows, int32_t C,
                                   int64_t ldx, float eps, void* stream) {
    CCV_REQUIRE(x && y && gamma && beta && rows > 0, CCV_EINVAL, "ccv_layernorm_small: bad args");
    CCV_REQUIRE(C > 0 && ldx >= C, CCV_ESHAPE, "ccv_layernorm_small: C=%d must be positive and ldx >= C", C);
    hipLaunchKernelGGL(ln_small_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, gamma, beta,
                       (long)rows, C, (long)ldx, eps);
    CCV_LAUNCH_CHECK("ccv_layernorm_small");
    return CCV_OK;
}
