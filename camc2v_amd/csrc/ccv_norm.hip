// GroupNorm(32)(+SiLU) and LayerNorm over token-major activations (gfx950).
//
// Both are HBM/L2-bound streaming passes: reads are fully coalesced along the channel axis
// (a row of C channels is contiguous), statistics are fp32.
//
// GroupNorm runs as three launches so that it is deterministic and needs no atomics:
//   gn_stats    grid (chunks, instances): per-chunk per-group (sum, sumsq) partials
//   gn_finalize grid (instances): partials -> (mean, rstd) per group
//   gn_apply    grid (chunks, instances): y = (x - mean) * rstd * gamma + beta [, SiLU] -> bf16
// A thread owns fixed channel PAIRS (pair index t, t+256, ...) for every row of its chunk, so
// its partial sums belong to fixed groups (channels-per-group is even for every layer here).
#include "ccv_common.h"

namespace {

constexpr int GN_GROUPS = 32;
constexpr int GN_MAX_CHUNKS = 256;
constexpr int GN_SLOTS = 8;  // C/2 <= 256 * 8  =>  C <= 4096

__host__ __device__ inline int gn_chunks(int rows) {
    int c = (rows + 31) / 32;
    return c < 1 ? 1 : (c > GN_MAX_CHUNKS ? GN_MAX_CHUNKS : c);
}

template <bool X_F32>
__device__ __forceinline__ float2 load_pair(const void* x, long idx_pair) {
    if (X_F32) return reinterpret_cast<const float2*>(x)[idx_pair];
    const uint32_t u = reinterpret_cast<const uint32_t*>(x)[idx_pair];
    return make_float2(bf16_to_f32((uint16_t)(u & 0xffffu)), bf16_to_f32((uint16_t)(u >> 16)));
}

template <bool X_F32>
__global__ __launch_bounds__(256) void gn_stats(const void* x, float* partial, int rows_per_instance, int C) {
    __shared__ float s_sum[256 * GN_SLOTS], s_sq[256 * GN_SLOTS];
    const int inst = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int pairs = C >> 1;
    const int rows_per_chunk = (rows_per_instance + nchunk - 1) / nchunk;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(rows_per_instance, r0 + rows_per_chunk);
    float sum[GN_SLOTS], sq[GN_SLOTS];
#pragma unroll
    for (int k = 0; k < GN_SLOTS; ++k) sum[k] = sq[k] = 0.f;
    const long base = (long)inst * rows_per_instance * pairs;
    for (int r = r0; r < r1; ++r) {
#pragma unroll
        for (int k = 0; k < GN_SLOTS; ++k) {
            const int pi = threadIdx.x + 256 * k;
            if (pi < pairs) {
                const float2 v = load_pair<X_F32>(x, base + (long)r * pairs + pi);
                sum[k] += v.x + v.y;
                sq[k] += v.x * v.x + v.y * v.y;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < GN_SLOTS; ++k) {
        s_sum[threadIdx.x + 256 * k] = sum[k];
        s_sq[threadIdx.x + 256 * k] = sq[k];
    }
    __syncthreads();
    if (threadIdx.x < GN_GROUPS) {
        const int ppg = pairs / GN_GROUPS;  // pairs per group
        float a = 0.f, b = 0.f;
        for (int i = 0; i < ppg; ++i) {
            a += s_sum[threadIdx.x * ppg + i];
            b += s_sq[threadIdx.x * ppg + i];
        }
        float* o = partial + (((long)inst * nchunk + chunk) * GN_GROUPS + threadIdx.x) * 2;
        o[0] = a;
        o[1] = b;
    }
}

__global__ void gn_finalize(const float* partial, float* stats, int nchunk, float inv_count, float eps) {
    const int inst = blockIdx.x, g = threadIdx.x;
    if (g >= GN_GROUPS) return;
    float a = 0.f, b = 0.f;
    for (int c = 0; c < nchunk; ++c) {
        const float* pp = partial + (((long)inst * nchunk + c) * GN_GROUPS + g) * 2;
        a += pp[0];
        b += pp[1];
    }
    const float mean = a * inv_count;
    const float var = fmaxf(b * inv_count - mean * mean, 0.f);
    stats[((long)inst * GN_GROUPS + g) * 2] = mean;
    stats[((long)inst * GN_GROUPS + g) * 2 + 1] = rsqrtf(var + eps);
}

template <bool X_F32>
__global__ __launch_bounds__(256) void gn_apply(const void* x, uint16_t* y, const float* gamma, const float* beta,
                                                const float* stats, int rows_per_instance, int C, int silu) {
    const int inst = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int pairs = C >> 1;
    const int ppg = pairs / GN_GROUPS;
    const int rows_per_chunk = (rows_per_instance + nchunk - 1) / nchunk;
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(rows_per_instance, r0 + rows_per_chunk);
    float sc0[GN_SLOTS], sc1[GN_SLOTS], sh0[GN_SLOTS], sh1[GN_SLOTS];
#pragma unroll
    for (int k = 0; k < GN_SLOTS; ++k) {
        const int pi = threadIdx.x + 256 * k;
        sc0[k] = sc1[k] = sh0[k] = sh1[k] = 0.f;
        if (pi < pairs) {
            const int g = pi / ppg;
            const float mean = stats[((long)inst * GN_GROUPS + g) * 2];
            const float rstd = stats[((long)inst * GN_GROUPS + g) * 2 + 1];
            sc0[k] = rstd * gamma[2 * pi];
            sc1[k] = rstd * gamma[2 * pi + 1];
            sh0[k] = beta[2 * pi] - mean * sc0[k];
            sh1[k] = beta[2 * pi + 1] - mean * sc1[k];
        }
    }
    const long base = (long)inst * rows_per_instance * pairs;
    uint32_t* yo = reinterpret_cast<uint32_t*>(y);
    for (int r = r0; r < r1; ++r) {
#pragma unroll
        for (int k = 0; k < GN_SLOTS; ++k) {
            const int pi = threadIdx.x + 256 * k;
            if (pi < pairs) {
                const long idx = base + (long)r * pairs + pi;
                const float2 v = load_pair<X_F32>(x, idx);
                float a = v.x * sc0[k] + sh0[k], b = v.y * sc1[k] + sh1[k];
                if (silu) { a = silu_f(a); b = silu_f(b); }
                yo[idx] = pack_bf16x2(a, b);
            }
        }
    }
}

// one wave per row, C/64 elements per lane (C <= 2048)
__global__ __launch_bounds__(256) void ln_kernel(const float* x, uint16_t* y, const float* gamma, const float* beta,
                                                 int rows, int C, float eps, const uint16_t* addend, int addend_rows, uint16_t* y2) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int per = C >> 6;
    const float* xr = x + (long)row * C;
    float v[32];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        v[i] = 0.f;
        if (i < per) { v[i] = xr[lane + 64 * i]; s += v[i]; }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (i < per) { const float d = v[i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 32; ++i)
        if (i < per) {
            const int c = lane + 64 * i;
            const float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
            y[(long)row * C + c] = f32_to_bf16(o);
            if (y2) y2[(long)row * C + c] = f32_to_bf16(o + bf16_to_f32(addend[(long)(row % addend_rows) * C + c]));
        }
}

}  // namespace

extern "C" int64_t ccv_groupnorm_ws_bytes(int32_t instances, int32_t C) {
    (void)C;
    return (int64_t)instances * (GN_MAX_CHUNKS + 1) * GN_GROUPS * 2 * (int64_t)sizeof(float);
}

extern "C" int ccv_groupnorm(const void* x, int32_t x_f32, uint16_t* y, const float* gamma, const float* beta,
                             int32_t instances, int32_t rows_per_instance, int32_t C, float eps, int32_t silu,
                             void* ws, void* stream) {
    CCV_REQUIRE(x && y && gamma && beta && ws, CCV_EINVAL, "ccv_groupnorm: null pointer");
    CCV_REQUIRE(instances > 0 && rows_per_instance > 0, CCV_EINVAL, "ccv_groupnorm: non-positive sizes");
    CCV_REQUIRE(instances <= 65535, CCV_ESHAPE, "ccv_groupnorm: too many instances");
    CCV_REQUIRE(C % 64 == 0 && C > 0 && C <= 512 * GN_SLOTS, CCV_ESHAPE, "ccv_groupnorm: C=%d must be a multiple of 64 and <= 4096", C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nchunk = gn_chunks(rows_per_instance);
    float* partial = static_cast<float*>(ws);
    float* stats = partial + (long)instances * GN_MAX_CHUNKS * GN_GROUPS * 2;
    dim3 grid(nchunk, instances);
    if (x_f32)
        hipLaunchKernelGGL(gn_stats<true>, grid, dim3(256), 0, st, x, partial, rows_per_instance, C);
    else
        hipLaunchKernelGGL(gn_stats<false>, grid, dim3(256), 0, st, x, partial, rows_per_instance, C);
    CCV_LAUNCH_CHECK("ccv_groupnorm(stats)");
    const float inv_count = 1.0f / ((float)rows_per_instance * (float)(C / GN_GROUPS));
    hipLaunchKernelGGL(gn_finalize, dim3(instances), dim3(64), 0, st, partial, stats, nchunk, inv_count, eps);
    CCV_LAUNCH_CHECK("ccv_groupnorm(finalize)");
    if (x_f32)
        hipLaunchKernelGGL(gn_apply<true>, grid, dim3(256), 0, st, x, y, gamma, beta, stats, rows_per_instance, C, silu);
    else
        hipLaunchKernelGGL(gn_apply<false>, grid, dim3(256), 0, st, x, y, gamma, beta, stats, rows_per_instance, C, silu);
    CCV_LAUNCH_CHECK("ccv_groupnorm(apply)");
    return CCV_OK;
}

extern "C" int ccv_layernorm(const float* x, uint16_t* y, const float* gamma, const float* beta,
                             int32_t rows, int32_t C, float eps, const uint16_t* addend, int32_t addend_rows, uint16_t* y2, void* stream) {
    CCV_REQUIRE(x && y && gamma && beta, CCV_EINVAL, "ccv_layernorm: null pointer");
    CCV_REQUIRE(rows > 0, CCV_EINVAL, "ccv_layernorm: rows=%d", rows);
    CCV_REQUIRE(C % 64 == 0 && C > 0 && C <= 2048, CCV_ESHAPE, "ccv_layernorm: C=%d must be a multiple of 64 and <= 2048", C);
    CCV_REQUIRE((addend == nullptr) == (y2 == nullptr), CCV_EINVAL, "ccv_layernorm: addend and y2 go together");
    CCV_REQUIRE(!addend || addend_rows > 0, CCV_EINVAL, "ccv_layernorm: addend_rows must be positive");
    hipLaunchKernelGGL(ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, gamma, beta,
                       rows, C, eps, addend, addend_rows > 0 ? addend_rows : 1, y2);
    CCV_LAUNCH_CHECK("ccv_layernorm");
    return CCV_OK;
}
