// Shared device/host helpers for libccv_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ccv.h"

// The MFMA operand type: bf16 (the default build, libccv_hip.so) or, with -DCCV_OPERANDS_F16 (libccv_hip_f16.so, CCV_OPERANDS=f16;
// round 4's numerics experiment: the reference itself runs under fp16 autocast, main/trainer.py:193), IEEE half.  Both forms of the
// MFMA issue at the same rate on gfx950.  Everything below that says "bf16" in its name means "the operand type": the element kind 0
// of the C ABI, the 16-bit storage format of weights, normalised activations, q / k / v, P and attention outputs.
#if defined(CCV_OPERANDS_F16)
typedef _Float16 ccv_opnd_t;
#else
typedef __bf16 ccv_opnd_t;
#endif
typedef __attribute__((ext_vector_type(8))) ccv_opnd_t bf16x8;
typedef __attribute__((ext_vector_type(4))) ccv_opnd_t bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ f32x16 ccv_mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
#if defined(CCV_OPERANDS_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 ccv_mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
#if defined(CCV_OPERANDS_F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 ccv_mfma_16x16x16(bf16x4 a, bf16x4 b, f32x4 c) {
#if defined(CCV_OPERANDS_F16)
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
#else
    typedef __attribute__((ext_vector_type(4))) short ccv_short4_t;
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(ccv_short4_t, a), __builtin_bit_cast(ccv_short4_t, b), c, 0, 0, 0);
#endif
}
// ds_read_b64_tr_b16 (the instruction does not care what the 16 bits mean)
__device__ __forceinline__ bf16x4 ccv_ds_read_tr16(const unsigned char* lds_ptr) {
#if defined(CCV_OPERANDS_F16)
    typedef __fp16 ccv_h4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) ccv_h4_t*)(lds_ptr)));
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds_ptr));
#endif
}

// ---- error plumbing -----------------------------------------------------------------
void ccv_set_error(const char* fmt, ...);

#define CCV_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            ccv_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

#define CCV_LAUNCH_CHECK(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            ccv_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return (int)e_;                                                      \
        }                                                                        \
    } while (0)

// ---- bf16 helpers (raw uint16 storage) ------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t v) {
#if defined(CCV_OPERANDS_F16)
    return (float)__builtin_bit_cast(_Float16, v);
#else
    return __uint_as_float(((uint32_t)v) << 16);
#endif
}

__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN preserving) on gfx950 (v_cvt_f16_f32 in the fp16-operand build)
    ccv_opnd_t b = (ccv_opnd_t)f;
    return __builtin_bit_cast(uint16_t, b);
}
// a packed pair of operand elements -> two floats
__device__ __forceinline__ void ccv_opnd2_to_f32(uint32_t pk, float& lo, float& hi) {
#if defined(CCV_OPERANDS_F16)
    typedef __attribute__((ext_vector_type(2))) _Float16 ccv_h2_t;
    const ccv_h2_t v = __builtin_bit_cast(ccv_h2_t, pk);
    lo = (float)v[0]; hi = (float)v[1];
#else
    lo = __uint_as_float(pk << 16); hi = __uint_as_float(pk & 0xffff0000u);
#endif
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#if defined(CCV_OPERANDS_F16)
    // the vector form: one v_cvt_pk_f16_f32 (the scalar form becomes two converts + an or)
    typedef __attribute__((ext_vector_type(2))) _Float16 ccv_h2p_t;
    const ccv_h2p_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
#else
    // the vector form here too: ONE v_cvt_pk_bf16_f32 dst, lo, hi wherever the call sits (the scalar form is matched to it in straight-line code
    // only; behind the runtime branches of the GEMM epilogues it became two converts + a shift + an or)
    typedef __attribute__((ext_vector_type(2))) __bf16 ccv_b2p_t;
    const ccv_b2p_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
#endif
}
// eight floats -> one MFMA operand fragment
__device__ __forceinline__ bf16x8 ccv_opnd8(float f0, float f1, float f2, float f3, float f4, float f5, float f6, float f7) {
#if defined(CCV_OPERANDS_F16)
    typedef __attribute__((ext_vector_type(4))) uint32_t ccv_u4_t;
    const ccv_u4_t u = {pack_bf16x2(f0, f1), pack_bf16x2(f2, f3), pack_bf16x2(f4, f5), pack_bf16x2(f6, f7)};
    return __builtin_bit_cast(bf16x8, u);
#else
    bf16x8 r;      // (element-wise casts: hipcc pairs them into v_cvt_pk_bf16_f32 by itself)
    r[0] = (ccv_opnd_t)f0; r[1] = (ccv_opnd_t)f1; r[2] = (ccv_opnd_t)f2; r[3] = (ccv_opnd_t)f3;
    r[4] = (ccv_opnd_t)f4; r[5] = (ccv_opnd_t)f5; r[6] = (ccv_opnd_t)f6; r[7] = (ccv_opnd_t)f7;
    return r;
#endif
}

// ---- fp16 helpers (the residual stream's hand-off format; raw uint16 storage) ---------------------------
__device__ __forceinline__ uint32_t ccv_pack_f16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float2 ccv_unpack_f16x2(uint32_t u) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const h2 v = __builtin_bit_cast(h2, u);
    return make_float2((float)v[0], (float)v[1]);
}
// Element kinds of activation tensors handed to the norm / layout kernels
enum { CCV_BF16 = 0, CCV_F32 = 1, CCV_F16 = 2 };
// four consecutive elements (index in units of 4 elements) of a bf16 / fp32 / fp16 tensor as floats
template <int KIND>
__device__ __forceinline__ float4 ccv_load4(const void* x, long idx4) {
    if constexpr (KIND == CCV_F32) return reinterpret_cast<const float4*>(x)[idx4];
    const uint2 u = reinterpret_cast<const uint2*>(x)[idx4];
    if constexpr (KIND == CCV_F16) {
        const float2 a = ccv_unpack_f16x2(u.x), b = ccv_unpack_f16x2(u.y);
        return make_float4(a.x, a.y, b.x, b.y);
    }
    float4 r;
    ccv_opnd2_to_f32(u.x, r.x, r.y);
    ccv_opnd2_to_f32(u.y, r.z, r.w);
    return r;
}

// x * rcp(1 + e^-x): v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup per element --
// a third of gn_apply's instructions, and vector instructions of a streaming kernel are not free beside another clip's MFMAs on the same SIMD);
// the results are rounded to bf16 by every caller.  Same limits: x -> -inf gives -0, x -> +inf gives x.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf-GELU x Phi(x) from Abramowitz-Stegun 7.1.26 (erf(z) = 1 - P(t) exp(-z^2), t = 1 / (1 + p z), |abs err| <= 1.5e-7: below the fp32 rounding of
// the product and far below the bf16 rounding of the stored result), written on the tail q = Phi(-|x|) = 0.5 P(t) exp(-x^2 / 2), t = 1 / (1 + p |x| / sqrt 2):
// Phi(x) = q for x < 0 and 1 - q otherwise, so x Phi(x) = max(x, 0) - |x| q.  12 vector instructions + v_rcp + v_exp instead of 16 + 2 for
// 0.5 x (1 + erf(x / sqrt 2)) (no copysign, no 1 + erf), and no cancellation in the negative tail (max |err| 3.3e-7, relative 1.7e-3 at x = -5
// against 4.6e-7 / 6.8e-2 of that form).  (The GELU of a GEGLU epilogue is vector
// work beside the tile's MFMAs: profiles/r04_attention_step_mix.txt.)
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.23164188f, 1.0f));
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752f);
    const float q = t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f + t * (-0.7265760135f + t * 0.5307027145f))));
    return __builtin_fmaf(-ax, q * e, fmaxf(x, 0.f));
}

// permuted token index -> stored row: frames of `hw` tokens, `w` wide, walked in 4x8-pixel patches (w == 0: identity).  Patches are
// numbered in 2x2 QUADS (8x16 pixels, quads row-major, patches row-major inside a quad) when the frame has an even number of patch rows and
// columns, else row-major: 128 consecutive tokens -- the queries of one workgroup of the shared-K/V attention kernel -- are then a compact
// 8x16-pixel block whatever the camera does.  (Round 4; before: four patches in a row, a 4x32 strip, whose queries see nearly the same key
// blocks only when the epipolar lines run along the strip: key blocks needed per 128 queries at 32x32 latents, of 512: benchmark camera
// 225 -> 211, vertical translation 506 -> 260, forward motion 476 -> 297.)  16x16 latents: the same order either way.
__host__ __device__ __forceinline__ int ccv_patch_row(int idx, int hw, int w) {
    if (w == 0) return idx;
    const int f = idx / hw, rem = idx - f * hw;
    const int patch = rem >> 5, within = rem & 31;
    const int ppr = w >> 3;                       // patches per patch-row
    int py, px;
    if (ppr > 2 && !(ppr & 1) && !((hw / w) & 7)) {      // (two patches per row: quad order = row-major order)
        const int quad = patch >> 2, sub = patch & 3, qpr = ppr >> 1;
        const int qy = quad / qpr, qx = quad - qy * qpr;
        py = 2 * qy + (sub >> 1);
        px = 2 * qx + (sub & 1);
    } else {
        py = patch / ppr;
        px = patch - py * ppr;
    }
    return f * hw + (py * 4 + (within >> 3)) * w + px * 8 + (within & 7);
}

// ---- work queues of the sparse attention kernel (ccv_attn.hip): XCD `xq` owns the slices s with s % 8 == xq ("home",
// n_full = nbh / 8 of them) in full and, of the n_extra = nbh % 8 leftover slices, the ranks r with r % c == q of slice
// 8 n_full + k, where k = xq % n_extra, c = number of XCDs with that k and q = xq / n_extra.  Items are merged in rank
// order (rank = position in the longest-first order): per block of c ranks, the home items of each rank and, after rank
// r0 + q, the one leftover item.  Every (slice, rank) pair appears in exactly one queue.
__host__ __device__ __forceinline__ void ccv_sparse_queue_geom(int nbh, int xq, int& n_full, int& k, int& c, int& q) {
    n_full = nbh >> 3;
    const int n_extra = nbh & 7;
    k = 0; c = 1; q = 0;
    if (n_extra > 0) {
        k = xq % n_extra;
        c = (7 - k) / n_extra + 1;
        q = xq / n_extra;
    }
}
__host__ __device__ __forceinline__ long ccv_sparse_queue_items(int nbh, int ngroups, int xq) {
    int n_full, k, c, q;
    ccv_sparse_queue_geom(nbh, xq, n_full, k, c, q);
    const int per_blk = c * n_full + ((nbh & 7) ? 1 : 0);
    return (long)((ngroups + c - 1) / c) * per_blk;
}
__host__ __device__ __forceinline__ void ccv_sparse_queue_item(int nbh, int ngroups, int xq, long idx, int& bh, int& rank) {
    int n_full, k, c, q;
    ccv_sparse_queue_geom(nbh, xq, n_full, k, c, q);
    (void)ngroups;
    const bool extra = (nbh & 7) != 0;
    const int per_blk = c * n_full + (extra ? 1 : 0);
    const int blk = (int)(idx / per_blk);
    int rem = (int)(idx - (long)blk * per_blk);
    const int r0 = blk * c;
    if (extra) {
        const int ext_pos = (q + 1) * n_full;
        if (rem == ext_pos) { bh = 8 * n_full + k; rank = r0 + q; return; }
        if (rem > ext_pos) --rem;
    }
    rank = r0 + rem / n_full;
    bh = xq + 8 * (rem % n_full);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
