// Layout, elementwise, DDIM-step and epipolar-mask kernels + library-wide error plumbing.
// All of these are HBM/L2-bound streaming passes with coalesced accesses along the
// fastest-varying output axis.
#include <stdarg.h>

#include "ccv_common.h"

// ---- error plumbing ---------------------------------------------------------------------
static thread_local char g_err[512] = "";

void ccv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* ccv_last_error(void) { return g_err; }
extern "C" int ccv_version(void) { return CCV_VERSION; }

namespace {

inline dim3 grid1d(int64_t n, int block = 256, int64_t cap = 1 << 20) {
    int64_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

// x [b, c1, t, hw] (+ x2 [b, c2, t, hw]) -> out [(b t hw), ldo], zero padded channels.
// One thread per output element; reads are strided by t*hw (small tensors: 8 channels).
__global__ void pack_nchw_kernel(const float* x, int c1, const float* x2, int c2, float* out, int ldo,
                                 int b, int t, int hw) {
    const int64_t n = (int64_t)b * t * hw * ldo;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ldo);
        const int64_t row = i / ldo;
        const int p = (int)(row % hw);
        const int f = (int)((row / hw) % t);
        const int bb = (int)(row / ((int64_t)hw * t));
        float v = 0.f;
        if (c < c1)
            v = x[(((int64_t)bb * c1 + c) * t + f) * hw + p];
        else if (c < c1 + c2)
            v = x2[(((int64_t)bb * c2 + (c - c1)) * t + f) * hw + p];
        out[i] = v;
    }
}

// in [(b t hw), ldi] -> out [b, c, t, hw]; one thread per output element (coalesced writes).
__global__ void unpack_nchw_kernel(const float* in, int ldi, float* out, int c, int b, int t, int hw) {
    const int64_t n = (int64_t)b * c * t * hw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % hw);
        const int f = (int)((i / hw) % t);
        const int cc = (int)((i / ((int64_t)hw * t)) % c);
        const int bb = (int)(i / ((int64_t)hw * t * c));
        out[i] = in[(((int64_t)bb * t + f) * hw + p) * ldi + cc];
    }
}

// out[row] = [a[row, 0:ca] | b[row, 0:cb]] in units of 4 elements (fp32: float4, fp16: 8 bytes); out16 = its bf16 rounding
template <int KIND>
__global__ void concat_rows_kernel(const void* a, int ca4, const void* b, int cb4, void* out, uint2* out16, int64_t rows) {
    const int w = ca4 + cb4;
    const int64_t n = rows * w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % w);
        const int64_t r = i / w;
        if constexpr (KIND == CCV_F32) {
            const float4 v = (c < ca4) ? reinterpret_cast<const float4*>(a)[r * ca4 + c] : reinterpret_cast<const float4*>(b)[r * cb4 + (c - ca4)];
            reinterpret_cast<float4*>(out)[i] = v;
            if (out16) out16[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        } else {
            const uint2 u = (c < ca4) ? reinterpret_cast<const uint2*>(a)[r * ca4 + c] : reinterpret_cast<const uint2*>(b)[r * cb4 + (c - ca4)];
            reinterpret_cast<uint2*>(out)[i] = u;
            if (out16) {
                const float2 lo = ccv_unpack_f16x2(u.x), hi = ccv_unpack_f16x2(u.y);
                out16[i] = make_uint2(pack_bf16x2(lo.x, lo.y), pack_bf16x2(hi.x, hi.y));
            }
        }
    }
}

template <int KIND>
__global__ void cast_bf16_kernel(const void* x, uint2* y, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = ccv_load4<KIND>(x, i);
        y[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
}

// x [b, c, t, hw] fp32 -> y [(b t hw), c] bf16; 32x32 tiles through LDS so both sides coalesce.
__global__ void nchw_to_rows_bf16_kernel(const float* x, uint16_t* y, int c, int t, int hw) {
    __shared__ float tile[32][33];
    const int bt = blockIdx.z;  // b * t + f
    const int bb = bt / t, f = bt % t;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int cc = c0 + j, p = p0 + tx;
        tile[j][tx] = (cc < c && p < hw) ? x[(((int64_t)bb * c + cc) * t + f) * hw + p] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, cc = c0 + tx;
        if (p < hw && cc < c) y[((int64_t)bt * hw + p) * c + cc] = f32_to_bf16(tile[tx][j]);
    }
}

__global__ void timestep_embedding_kernel(const float* t, uint16_t* out, int n, int dim) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * half) return;
    const int row = i / half, k = i % half;
    // freqs = exp(-ln(10000) * k / half)  (lvdm/models/utils_diffusion.py:19-23)
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);
    const float arg = t[row] * freq;
    out[(int64_t)row * dim + k] = f32_to_bf16(cosf(arg));
    out[(int64_t)row * dim + half + k] = f32_to_bf16(sinf(arg));
    if ((dim & 1) && k == 0) out[(int64_t)row * dim + dim - 1] = 0;
}

__global__ void add_silu_bf16_kernel(const float* a, const float* b, uint16_t* out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = a[i] + (b ? b[i] : 0.f);
        out[i] = f32_to_bf16(silu_f(v));
    }
}

// ---- DDIM: per-sample sums for the std rescale, then the elementwise update --------------
__global__ __launch_bounds__(1024) void ddim_stats_kernel(const float* e_c, const float* e_uc, float scale,
                                                          int64_t per_sample, float* ws) {
    __shared__ float red[4][16];
    const int smp = blockIdx.x;
    const float* ec = e_c + smp * per_sample;
    const float* eu = e_uc + smp * per_sample;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int64_t i = threadIdx.x; i < per_sample; i += blockDim.x) {
        const float c = ec[i], u = eu[i];
        const float e = u + scale * (c - u);
        s0 += c; s1 += c * c; s2 += e; s3 += e * e;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = s0; red[1][w] = s1; red[2][w] = s2; red[3][w] = s3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        float a = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) a += red[threadIdx.x][i];
        ws[smp * 4 + threadIdx.x] = a;
    }
}

__global__ void ddim_update_kernel(const float* x, const float* e_c, const float* e_uc, const float* noise,
                                   float* x_prev, float* pred_x0, const float* coef, float scale, float gr,
                                   int64_t per_sample, int64_t n, const float* ws) {
    const float a_t = coef[0], a_prev = coef[1], sigma = coef[2], sqrt_1m_at = coef[3];
    const float inv_sqrt_at = 1.0f / sqrtf(a_t);
    const float sqrt_aprev = sqrtf(a_prev);
    const float dir = sqrtf(fmaxf(1.0f - a_prev - sigma * sigma, 0.f));
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float e = e_c[i];
        if (e_uc) {
            const float u = e_uc[i];
            e = u + scale * (e - u);
            if (gr > 0.f) {
                const float* s = ws + (i / per_sample) * 4;
                const float np = (float)per_sample;
                // unbiased std (torch.std default), lvdm/models/utils_diffusion.py:152-153
                const float var_c = fmaxf((s[1] - s[0] * s[0] / np) / (np - 1.f), 0.f);
                const float var_e = fmaxf((s[3] - s[2] * s[2] / np) / (np - 1.f), 0.f);
                const float fac = sqrtf(var_c) / sqrtf(var_e);
                e = gr * (e * fac) + (1.f - gr) * e;
            }
        }
        const float x0 = (x[i] - sqrt_1m_at * e) * inv_sqrt_at;
        float xp = sqrt_aprev * x0 + dir * e;
        if (noise) xp += sigma * noise[i];
        x_prev[i] = xp;
        if (pred_x0) pred_x0[i] = x0;
    }
}

// Camera guidance (third forward, lvdm/models/samplers/ddim.py:268-280) folded into the unconditional prediction:
// out = e_uc + coeff * w(t) * (e_c - e_nc), w = 1 ('constant') or cos((1 - t/999) pi/2) ('cosine', per sample).
__global__ void camera_cfg_fold_kernel(const float* e_uc, const float* e_c, const float* e_nc, const int64_t* t, float coeff,
                                       float* out, int64_t per_sample, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float w = coeff;
        if (t) w *= cosf((1.0f - (float)t[i / per_sample] / 999.0f) * 3.14159265358979323846f * 0.5f);
        out[i] = e_uc[i] + w * (e_c[i] - e_nc[i]);
    }
}

// ---- epipolar mask preparation -----------------------------------------------------------
// bool bytes [B, Lq, Lk] -> words [B, Lq, W]; one thread per word (32 contiguous bytes).
__device__ __forceinline__ void mark_block(uint32_t* wave_bits, int64_t b, int Lq, int q, int w, int words) {
    // 64-query group (q / 64) needs key block w (32 keys)
    const int wave_words = (words + 31) / 32;
    atomicOr(wave_bits + (b * ((Lq + 63) / 64) + q / 64) * wave_words + (w >> 5), 1u << (w & 31));
}

__global__ void pack_mask_kernel(const uint8_t* mask, uint32_t* bits, uint8_t* flags, uint32_t* wave_bits, int Lq, int Lk, int words,
                                 int ktiles, int64_t nwords_total, int perm_hw, int perm_w) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nwords_total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % words);
        const int64_t rowi = i / words;  // b * Lq + q
        const int q = (int)(rowi % Lq);
        const int64_t b = rowi / Lq;
        uint32_t word = 0;
        const int nk = min(32, Lk - w * 32);
        if (perm_w) {  // patch order: output row q / bit j <-> stored row patch_row(q) / column patch_row(32 w + j)
            const uint8_t* srow = mask + (b * Lq + ccv_patch_row(q, perm_hw, perm_w)) * (int64_t)Lk;
            for (int j = 0; j < nk; ++j) word |= (srow[ccv_patch_row(w * 32 + j, perm_hw, perm_w)] ? 1u : 0u) << j;
            bits[i] = word;
            if (word && flags) flags[(b * ((Lq + 127) / 128) + q / 128) * ktiles + (w >> 1)] = 1;
            if (word && wave_bits) mark_block(wave_bits, b, Lq, q, w, words);
            continue;
        }
        const uint8_t* src = mask + rowi * Lk + (int64_t)w * 32;
        if (nk == 32 && (((uintptr_t)src) & 15) == 0) {
            const uint4 lo = reinterpret_cast<const uint4*>(src)[0], hi = reinterpret_cast<const uint4*>(src)[1];
            const uint32_t v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) word |= (((v[k] >> (8 * j)) & 0xffu) ? 1u : 0u) << (4 * k + j);
        } else {
            for (int j = 0; j < nk; ++j) word |= (src[j] ? 1u : 0u) << j;
        }
        bits[i] = word;
        if (word && flags) flags[(b * ((Lq + 127) / 128) + q / 128) * ktiles + (w >> 1)] = 1;
        if (word && wave_bits) mark_block(wave_bits, b, Lq, q, w, words);
    }
}

// F [B, T, T, 3, 3] -> packed mask words; query (t1,p1) row, key (t2,p2) column.
// Arithmetic mirrors model/camcontexti2v.py:229-239 in fp32:  l = F x1; l /= ||l_xy||; visible <=> |l . x2| < d*sqrt(2)/2.
// The reference evaluates both 3-term dot products as K = 3 matrix products, i.e. as the sequential chain
// acc = a0 b0 (rounded), acc = fma(a1, b1, acc), acc = fma(a2, 1, acc) = acc + a2: written out below with explicit rounding
// steps, which reproduces the reference's masks bit for bit (tests/golden/geometry_bits.npz: 0 of 10.4 M bits differ at
// every resolution; plain mul/add or the reverse chain flips 3-19 bits within an ulp of the threshold).
__device__ __forceinline__ float dot3_chain(float a0, float b0, float a1, float b1, float a2) {
#pragma clang fp contract(off)
    const float prod = a0 * b0;                  // rounded product
    const float acc = __builtin_fmaf(a1, b1, prod);
    return acc + a2;                             // (HIP's __fmul_rn / __fadd_rn are plain operators, __fsqrt_rn is the NATIVE square root)
}

__global__ void epipolar_bits_kernel(const float* F, uint32_t* bits, uint8_t* flags, uint32_t* wave_bits, int T, int Tk, int H, int W, float d,
                                     int words, int ktiles, int64_t nwords_total, int perm_w) {
#pragma clang fp contract(off)
    const int HW = H * W, L = T * HW, Lk = Tk * HW;   // L query rows (T frames), Lk key columns (Tk frames)
    const float thr = d * 0.70710678118654752440f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nwords_total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % words);
        const int64_t rowi = i / words;
        const int q = (int)(rowi % L);
        const int64_t b = rowi / L;
        const int qs = ccv_patch_row(q, HW, perm_w);   // stored (raster) token of output row q
        const int t1 = qs / HW, p1 = qs % HW;
        const float x1 = (float)(p1 % W) * d + d / 2.0f - 0.5f;
        const float y1 = (float)(p1 / W) * d + d / 2.0f - 0.5f;
        uint32_t word = 0;
        int t2_cached = -1;
        float l0 = 0.f, l1 = 0.f, l2 = 0.f;
        for (int j = 0; j < 32; ++j) {
            if (w * 32 + j >= Lk) break;
            const int key = ccv_patch_row(w * 32 + j, HW, perm_w);
            const int t2 = key / HW, p2 = key % HW;
            if (t2 != t2_cached) {
                const float* f = F + ((b * T + t1) * Tk + t2) * 9;
                const float a0 = dot3_chain(f[0], x1, f[1], y1, f[2]);
                const float a1 = dot3_chain(f[3], x1, f[4], y1, f[5]);
                const float a2 = dot3_chain(f[6], x1, f[7], y1, f[8]);
                const float nrm = sqrtf(a0 * a0 + a1 * a1);      // correctly rounded square root and divisions (hipcc default); no contraction (pragma above)
                l0 = a0 / nrm; l1 = a1 / nrm; l2 = a2 / nrm;
                t2_cached = t2;
            }
            const float x2 = (float)(p2 % W) * d + d / 2.0f - 0.5f;
            const float y2 = (float)(p2 / W) * d + d / 2.0f - 0.5f;
            const float dist = fabsf(dot3_chain(l0, x2, l1, y2, l2));
            word |= (dist < thr ? 1u : 0u) << j;
        }
        bits[i] = word;
        if (word && flags) flags[(b * ((L + 127) / 128) + q / 128) * ktiles + (w >> 1)] = 1;
        if (word && wave_bits) mark_block(wave_bits, b, L, q, w, words);
    }
}

// softmax over the last axis: x fp32 [rows, ldx] -> y bf16 [rows, ldy], one wave per row, three passes over a row that
// stays in L1/L2 (rows here are 1024 logits of the first-stage decoder's single-head attention)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, uint16_t* y, int rows, int L, long ldx, long ldy) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * ldx);
    const int n4 = L >> 2;
    float m = -3.0e38f;
    for (int i = lane; i < n4; i += 64) {
        const float4 v = xr[i];
        m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const float4 v = xr[i];
        s += (__expf(v.x - m) + __expf(v.y - m)) + (__expf(v.z - m) + __expf(v.w - m));
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    uint2* yr = reinterpret_cast<uint2*>(y + (long)row * ldy);
    for (int i = lane; i < n4; i += 64) {
        const float4 v = xr[i];
        yr[i] = make_uint2(pack_bf16x2(__expf(v.x - m) * inv, __expf(v.y - m) * inv), pack_bf16x2(__expf(v.z - m) * inv, __expf(v.w - m) * inv));
    }
}

// Longest-first schedule of the sparse attention kernel: one workgroup per mask batch counts the needed key
// blocks of every 64-query group (popcount of its wave_bits row) and rank-sorts the groups (O(n^2), n <= 8192,
// once per clip).
// merge > 1: an item is `merge` consecutive groups and counts the blocks ANY of them needs (the OR of their rows): the items of the
// workgroup-shared sparse kernel.
__global__ __launch_bounds__(256) void group_order_kernel(const uint32_t* wave_bits, int ngroups, int words, int merge, int32_t* order) {
    __shared__ int cnt[8192];
    const uint32_t* wb = wave_bits + (long)blockIdx.x * ngroups * words;
    const int nitems = (ngroups + merge - 1) / merge;
    for (int i = threadIdx.x; i < nitems; i += 256) {
        int c = 0;
        for (int w = 0; w < words; ++w) {
            uint32_t u = 0u;
            for (int j = 0; j < merge; ++j)
                if (i * merge + j < ngroups) u |= wb[(long)(i * merge + j) * words + w];
            c += __popc(u);
        }
        cnt[i] = c;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nitems; i += 256) {
        const int ci = cnt[i];
        int rank = 0;
        for (int j = 0; j < nitems; ++j) rank += (cnt[j] > ci || (cnt[j] == ci && j < i)) ? 1 : 0;
        order[(long)blockIdx.x * nitems + rank] = i;
    }
}

}  // namespace

extern "C" int ccv_pack_nchw_to_rows(const float* x, int32_t c1, const float* x2, int32_t c2, float* out, int32_t ldo,
                                     int32_t b, int32_t t, int32_t hw, void* stream) {
    CCV_REQUIRE(x && out, CCV_EINVAL, "ccv_pack_nchw_to_rows: null pointer");
    CCV_REQUIRE(c1 > 0 && c2 >= 0 && (c2 == 0 || x2) && ldo >= c1 + c2 && b > 0 && t > 0 && hw > 0, CCV_EINVAL,
                "ccv_pack_nchw_to_rows: bad sizes");
    const int64_t n = (int64_t)b * t * hw * ldo;
    hipLaunchKernelGGL(pack_nchw_kernel, grid1d(n), dim3(256), 0, static_cast<hipStream_t>(stream), x, c1, x2, c2, out, ldo, b, t, hw);
    CCV_LAUNCH_CHECK("ccv_pack_nchw_to_rows");
    return CCV_OK;
}

extern "C" int ccv_unpack_rows_to_nchw(const float* in, int32_t ldi, float* out, int32_t c, int32_t b, int32_t t,
                                       int32_t hw, void* stream) {
    CCV_REQUIRE(in && out && c > 0 && ldi >= c && b > 0 && t > 0 && hw > 0, CCV_EINVAL, "ccv_unpack_rows_to_nchw: bad args");
    const int64_t n = (int64_t)b * c * t * hw;
    hipLaunchKernelGGL(unpack_nchw_kernel, grid1d(n), dim3(256), 0, static_cast<hipStream_t>(stream), in, ldi, out, c, b, t, hw);
    CCV_LAUNCH_CHECK("ccv_unpack_rows_to_nchw");
    return CCV_OK;
}

extern "C" int ccv_concat_rows(const void* a, int32_t ca, const void* b, int32_t cb, void* out, uint16_t* out_bf16, int64_t rows,
                               int32_t kind, void* stream) {
    CCV_REQUIRE(a && b && out && rows > 0, CCV_EINVAL, "ccv_concat_rows: bad args");
    CCV_REQUIRE(kind == CCV_F32 || kind == CCV_F16, CCV_EINVAL, "ccv_concat_rows: kind must be 1 (fp32) or 2 (fp16)");
    CCV_REQUIRE(ca % 4 == 0 && cb % 4 == 0 && ca > 0 && cb > 0, CCV_ESHAPE, "ccv_concat_rows: channel counts must be multiples of 4");
    const int64_t n = rows * ((ca + cb) / 4);
    if (kind == CCV_F32)
        hipLaunchKernelGGL(concat_rows_kernel<CCV_F32>, grid1d(n), dim3(256), 0, static_cast<hipStream_t>(stream), a, ca / 4, b, cb / 4, out,
                           reinterpret_cast<uint2*>(out_bf16), rows);
    else
        hipLaunchKernelGGL(concat_rows_kernel<CCV_F16>, grid1d(n), dim3(256), 0, static_cast<hipStream_t>(stream), a, ca / 4, b, cb / 4, out,
                           reinterpret_cast<uint2*>(out_bf16), rows);
    CCV_LAUNCH_CHECK("ccv_concat_rows");
    return CCV_OK;
}

extern "C" int ccv_cast_bf16(const void* x, int32_t x_kind, uint16_t* y, int64_t n, void* stream) {
    CCV_REQUIRE(x && y && n > 0, CCV_EINVAL, "ccv_cast_bf16: bad args");
    CCV_REQUIRE(x_kind == CCV_F32 || x_kind == CCV_F16, CCV_EINVAL, "ccv_cast_bf16: x_kind must be 1 (fp32) or 2 (fp16)");
    CCV_REQUIRE(n % 4 == 0, CCV_ESHAPE, "ccv_cast_bf16: n must be a multiple of 4");
    if (x_kind == CCV_F32)
        hipLaunchKernelGGL(cast_bf16_kernel<CCV_F32>, grid1d(n / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, reinterpret_cast<uint2*>(y), n / 4);
    else
        hipLaunchKernelGGL(cast_bf16_kernel<CCV_F16>, grid1d(n / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, reinterpret_cast<uint2*>(y), n / 4);
    CCV_LAUNCH_CHECK("ccv_cast_bf16");
    return CCV_OK;
}

extern "C" int ccv_nchw_to_rows_bf16(const float* x, uint16_t* y, int32_t b, int32_t c, int32_t t, int32_t hw, void* stream) {
    CCV_REQUIRE(x && y && b > 0 && c > 0 && t > 0 && hw > 0, CCV_EINVAL, "ccv_nchw_to_rows_bf16: bad args");
    CCV_REQUIRE((int64_t)b * t <= 65535, CCV_ESHAPE, "ccv_nchw_to_rows_bf16: b*t too large");
    dim3 grid((hw + 31) / 32, (c + 31) / 32, b * t);
    hipLaunchKernelGGL(nchw_to_rows_bf16_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, y, c, t, hw);
    CCV_LAUNCH_CHECK("ccv_nchw_to_rows_bf16");
    return CCV_OK;
}

extern "C" int ccv_timestep_embedding(const float* t, uint16_t* out, int32_t n, int32_t dim, void* stream) {
    CCV_REQUIRE(t && out && n > 0 && dim > 1, CCV_EINVAL, "ccv_timestep_embedding: bad args");
    const int total = n * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), t, out, n, dim);
    CCV_LAUNCH_CHECK("ccv_timestep_embedding");
    return CCV_OK;
}

extern "C" int ccv_add_silu_bf16(const float* a, const float* b, uint16_t* out, int64_t n, void* stream) {
    CCV_REQUIRE(a && out && n > 0, CCV_EINVAL, "ccv_add_silu_bf16: bad args");
    hipLaunchKernelGGL(add_silu_bf16_kernel, grid1d(n), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, out, n);
    CCV_LAUNCH_CHECK("ccv_add_silu_bf16");
    return CCV_OK;
}

extern "C" int ccv_ddim_cfg_step(const float* x, const float* e_c, const float* e_uc, const float* noise, float* x_prev,
                                 float* pred_x0, const float* coef, float scale, float guidance_rescale,
                                 int32_t n_samples, int64_t per_sample, float* ws, void* stream) {
    CCV_REQUIRE(x && e_c && x_prev && coef, CCV_EINVAL, "ccv_ddim_cfg_step: null pointer");
    CCV_REQUIRE(n_samples > 0 && per_sample > 1, CCV_EINVAL, "ccv_ddim_cfg_step: bad sizes");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool rescale = e_uc && guidance_rescale > 0.f;
    if (rescale) {
        CCV_REQUIRE(ws != nullptr, CCV_EINVAL, "ccv_ddim_cfg_step: guidance rescale needs the workspace");
        hipLaunchKernelGGL(ddim_stats_kernel, dim3(n_samples), dim3(1024), 0, st, e_c, e_uc, scale, per_sample, ws);
        CCV_LAUNCH_CHECK("ccv_ddim_cfg_step(stats)");
    }
    const int64_t n = (int64_t)n_samples * per_sample;
    hipLaunchKernelGGL(ddim_update_kernel, grid1d(n), dim3(256), 0, st, x, e_c, e_uc, noise, x_prev, pred_x0, coef, scale,
                       rescale ? guidance_rescale : 0.f, per_sample, n, ws);
    CCV_LAUNCH_CHECK("ccv_ddim_cfg_step(update)");
    return CCV_OK;
}

extern "C" int ccv_camera_cfg_fold(const float* e_uc, const float* e_c, const float* e_nc, const int64_t* t, float coeff, float* out,
                                   int32_t n_samples, int64_t per_sample, void* stream) {
    CCV_REQUIRE(e_uc && e_c && e_nc && out, CCV_EINVAL, "ccv_camera_cfg_fold: null pointer");
    CCV_REQUIRE(n_samples > 0 && per_sample > 0, CCV_EINVAL, "ccv_camera_cfg_fold: bad sizes");
    const int64_t n = (int64_t)n_samples * per_sample;
    hipLaunchKernelGGL(camera_cfg_fold_kernel, grid1d(n), dim3(256), 0, static_cast<hipStream_t>(stream), e_uc, e_c, e_nc, t, coeff, out,
                       per_sample, n);
    CCV_LAUNCH_CHECK("ccv_camera_cfg_fold");
    return CCV_OK;
}

extern "C" int ccv_pack_mask(const uint8_t* mask, uint32_t* bits, uint8_t* flags, uint32_t* wave_bits, int32_t B, int32_t Lq,
                             int32_t Lk, int32_t perm_hw, int32_t perm_w, void* stream) {
    CCV_REQUIRE(mask && bits && B > 0 && Lq > 0 && Lk > 0, CCV_EINVAL, "ccv_pack_mask: bad args");
    CCV_REQUIRE(perm_w == 0 || (perm_w % 8 == 0 && perm_hw > 0 && perm_hw % (4 * perm_w) == 0 && Lq % perm_hw == 0 && Lk % perm_hw == 0),
                CCV_ESHAPE, "ccv_pack_mask: patch order needs W %% 8 == 0, H %% 4 == 0 and whole frames");
    const int words = (Lk + 31) / 32, ktiles = (Lk + 63) / 64;
    const int64_t n = (int64_t)B * Lq * words;
    hipLaunchKernelGGL(pack_mask_kernel, grid1d(n), dim3(256), 0, static_cast<hipStream_t>(stream), mask, bits, flags, wave_bits, Lq, Lk, words, ktiles, n, perm_hw, perm_w);
    CCV_LAUNCH_CHECK("ccv_pack_mask");
    return CCV_OK;
}

extern "C" int ccv_epipolar_mask_bits_rect(const float* F, uint32_t* bits, uint8_t* flags, uint32_t* wave_bits, int32_t B, int32_t Tq,
                                           int32_t Tk, int32_t H, int32_t W, int32_t downsample, int32_t patch_order, void* stream);

extern "C" int ccv_epipolar_mask_bits(const float* F, uint32_t* bits, uint8_t* flags, uint32_t* wave_bits, int32_t B, int32_t T,
                                      int32_t H, int32_t W, int32_t downsample, int32_t patch_order, void* stream) {
    CCV_REQUIRE(F && bits && B > 0 && T > 0 && H > 0 && W > 0 && downsample > 0, CCV_EINVAL, "ccv_epipolar_mask_bits: bad args");
    CCV_REQUIRE(!patch_order || (W % 8 == 0 && H % 4 == 0), CCV_ESHAPE, "ccv_epipolar_mask_bits: patch order needs W %% 8 == 0 and H %% 4 == 0");
    return ccv_epipolar_mask_bits_rect(F, bits, flags, wave_bits, B, T, T, H, W, downsample, patch_order, stream);
}

extern "C" int ccv_epipolar_mask_bits_rect(const float* F, uint32_t* bits, uint8_t* flags, uint32_t* wave_bits, int32_t B, int32_t Tq,
                                           int32_t Tk, int32_t H, int32_t W, int32_t downsample, int32_t patch_order, void* stream) {
    CCV_REQUIRE(F && bits && B > 0 && Tq > 0 && Tk > 0 && H > 0 && W > 0 && downsample > 0, CCV_EINVAL, "ccv_epipolar_mask_bits_rect: bad args");
    CCV_REQUIRE(!patch_order || (W % 8 == 0 && H % 4 == 0), CCV_ESHAPE, "ccv_epipolar_mask_bits_rect: patch order needs W %% 8 == 0 and H %% 4 == 0");
    const int L = Tq * H * W, Lk = Tk * H * W;
    const int words = (Lk + 31) / 32, ktiles = (Lk + 63) / 64;
    const int64_t n = (int64_t)B * L * words;
    hipLaunchKernelGGL(epipolar_bits_kernel, grid1d(n), dim3(256), 0, static_cast<hipStream_t>(stream), F, bits, flags, wave_bits, Tq, Tk, H, W,
                       (float)downsample, words, ktiles, n, patch_order ? W : 0);
    CCV_LAUNCH_CHECK("ccv_epipolar_mask_bits_rect");
    return CCV_OK;
}

extern "C" int ccv_attn_group_order(const uint32_t* wave_bits, int32_t B, int32_t ngroups, int32_t wave_words, int32_t* order, void* stream) {
    CCV_REQUIRE(wave_bits && order && B > 0 && ngroups > 0 && wave_words > 0, CCV_EINVAL, "ccv_attn_group_order: bad args");
    CCV_REQUIRE(ngroups <= 8192, CCV_ESHAPE, "ccv_attn_group_order: at most 8192 query groups (Lq <= 524288)");
    hipLaunchKernelGGL(group_order_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), wave_bits, ngroups, wave_words, 1, order);
    CCV_LAUNCH_CHECK("ccv_attn_group_order");
    return CCV_OK;
}

extern "C" int ccv_attn_group_order_merged(const uint32_t* wave_bits, int32_t B, int32_t ngroups, int32_t wave_words, int32_t merge, int32_t* order,
                                           void* stream) {
    CCV_REQUIRE(wave_bits && order && B > 0 && ngroups > 0 && wave_words > 0, CCV_EINVAL, "ccv_attn_group_order_merged: bad args");
    CCV_REQUIRE(merge >= 1 && merge <= 8, CCV_EINVAL, "ccv_attn_group_order_merged: merge must be 1 .. 8 (got %d)", merge);
    CCV_REQUIRE((ngroups + merge - 1) / merge <= 8192, CCV_ESHAPE, "ccv_attn_group_order_merged: at most 8192 items");
    hipLaunchKernelGGL(group_order_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), wave_bits, ngroups, wave_words, merge, order);
    CCV_LAUNCH_CHECK("ccv_attn_group_order_merged");
    return CCV_OK;
}

extern "C" int ccv_softmax_rows(const float* x, uint16_t* y, int32_t rows, int32_t L, int64_t ldx, int64_t ldy, void* stream) {
    CCV_REQUIRE(x && y && rows > 0 && L > 0, CCV_EINVAL, "ccv_softmax_rows: bad args");
    CCV_REQUIRE(L % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= L && ldy >= L, CCV_ESHAPE,
                "ccv_softmax_rows: L and the leading dimensions must be multiples of 4 (L=%d)", L);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, rows, L, (long)ldx, (long)ldy);
    CCV_LAUNCH_CHECK("ccv_softmax_rows");
    return CCV_OK;
}
