// fp8 (OCP e4m3) variant of the sparse epipolar attention -- BASELINE.json configs[4] / north_star "fp8 MFMA attention path".
//
// The bf16 sparse kernel (ccv_attn.hip) is bound by the K/V bytes it moves per visited 32-key block (8 KiB through a
// wave-private LDS ring, 9 DMA wave-instructions) and by its softmax VALU work, not by MFMA rate.  This variant halves the
// bytes and the DMA instructions:
//   ccv_attn_fp8_pack   once per attention call: per-(batch, head) amax of q / k / v (register tokens included) -> scales;
//                       q -> e4m3 rows; K -> e4m3 in BLOCK-contiguous order [(b h)][block][32 keys][64]; V -> e4m3 TRANSPOSED
//                       blocks [(b h)][block][64 d][32 keys] (keys permuted inside each 16 so that the 8 keys one lane feeds to
//                       the PV MFMA are 8 contiguous bytes).  The 4x8-patch token order of the mask is applied here, so the
//                       attention kernel fetches every block as one contiguous 2 KiB piece.
//   ccv_attn_sparse_fp8_fwd   same schedule, masks, online softmax (fp32) and output layout as attn_sparse_kernel; S^T = K Q^T
//                       and O^T += V^T P^T on v_mfma_f32_32x32x16_fp8_fp8 (P in e4m3, scales folded into the softmax scale
//                       and the final normalisation); 5 DMA wave-instructions per block instead of 9.
// Reference call site it serves: F.scaled_dot_product_attention in model/modules/epipolar.py:99.
#include <atomic>

#include "ccv_common.h"

namespace {

constexpr float NEG_INF = -__builtin_inff();
constexpr float E4M3_MAX = 448.0f;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
__device__ __forceinline__ uint2 pack8_fp8(const uint4 raw, float inv) {   // 8 bf16 -> 8 e4m3, scaled by inv
    const unsigned u[4] = {raw.x, raw.y, raw.z, raw.w};
    float f[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf16_to_f32((uint16_t)(u[i] & 0xffffu)) * inv;
        f[2 * i + 1] = bf16_to_f32((uint16_t)(u[i] >> 16)) * inv;
    }
    return make_uint2(pack4_fp8(f[0], f[1], f[2], f[3]), pack4_fp8(f[4], f[5], f[6], f[7]));
}
__device__ __forceinline__ float amax8(const uint4 raw) {
    const unsigned u[4] = {raw.x, raw.y, raw.z, raw.w};
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) m = fmaxf(m, fmaxf(fabsf(bf16_to_f32((uint16_t)(u[i] & 0xffffu))), fabsf(bf16_to_f32((uint16_t)(u[i] >> 16)))));
    return m;
}

// ---- amax per (batch, head): grid (B*H, chunks), 256 threads; amax[bh][0..2] = max |q|, |k|, |v| (non-negative floats order like ints)
__global__ __launch_bounds__(256) void fp8_amax_kernel(const CcvAttn p, unsigned int* amax) {
    const int bh = blockIdx.x, head = bh % p.H, b = bh / p.H;
    const int c = threadIdx.x & 7, rsub = threadIdx.x >> 3;
    float mq = 0.f, mk = 0.f, mv = 0.f;
    for (int t = blockIdx.y * 32 + rsub; t < p.Lk; t += gridDim.y * 32) {
        mq = fmaxf(mq, amax8(*reinterpret_cast<const uint4*>(p.q + (long)b * p.q_bso + (long)t * p.q_ls + head * 64 + 8 * c)));
        mk = fmaxf(mk, amax8(*reinterpret_cast<const uint4*>(p.k + (long)b * p.k_bso + (long)t * p.k_ls + head * 64 + 8 * c)));
        mv = fmaxf(mv, amax8(*reinterpret_cast<const uint4*>(p.v + (long)b * p.v_bso + (long)t * p.v_ls + head * 64 + 8 * c)));
    }
    if (blockIdx.y == 0 && p.kreg && rsub < p.nreg) {
        mk = fmaxf(mk, amax8(*reinterpret_cast<const uint4*>(p.kreg + (long)rsub * p.H * 64 + head * 64 + 8 * c)));
        mv = fmaxf(mv, amax8(*reinterpret_cast<const uint4*>(p.vreg + (long)rsub * p.H * 64 + head * 64 + 8 * c)));
    }
    mq = wave_max(mq); mk = wave_max(mk); mv = wave_max(mv);
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&amax[bh * 4 + 0], __float_as_uint(mq));
        atomicMax(&amax[bh * 4 + 1], __float_as_uint(mk));
        atomicMax(&amax[bh * 4 + 2], __float_as_uint(mv));
    }
}

// position p (0..31) of a key inside its V^T block row -> key index within the block: inside each 16 keys the order is
// [0 1 2 3 8 9 10 11 | 4 5 6 7 12 13 14 15]: the 8 keys that lane half hh of the PV MFMA's B operand holds (rows
// (i & 3) + 8 (i >> 2) + 4 hh of the 32x32 score accumulator, i = 8 s2 .. 8 s2 + 7) are then 8 contiguous bytes.
__device__ __forceinline__ int vt_key_of_pos(int pos) {
    const int s2 = pos >> 4, w = pos & 15, hh = w >> 3, idx = w & 7;
    return 16 * s2 + (idx < 4 ? idx : idx + 4) + 4 * hh;
}

// grid (nblk + 1, B*H), 256 threads: block `blk` of slice bh (blk == nblk: the register tokens)
__global__ __launch_bounds__(256) void fp8_pack_kernel(const CcvAttn p, const unsigned int* amax, uint8_t* q8, uint8_t* k8b, uint8_t* v8tb,
                                                       float* scales, int nblk) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[32 * 64];
    const int blk = blockIdx.x, bh = blockIdx.y, head = bh % p.H, b = bh / p.H;
    const float aq = __uint_as_float(amax[bh * 4 + 0]), ak = __uint_as_float(amax[bh * 4 + 1]), av = __uint_as_float(amax[bh * 4 + 2]);
    const float sq = aq > 0.f ? aq / E4M3_MAX : 1.f, sk = ak > 0.f ? ak / E4M3_MAX : 1.f, sv = av > 0.f ? av / E4M3_MAX : 1.f;
    if (blk == 0 && threadIdx.x == 0) { scales[bh * 4 + 0] = sq; scales[bh * 4 + 1] = sk; scales[bh * 4 + 2] = sv; scales[bh * 4 + 3] = 0.f; }
    const int j = threadIdx.x >> 3, c = threadIdx.x & 7;
    const bool reg = blk == nblk;
    const int idx = 32 * blk + j;                              // permuted token index of this key
    uint4 kr = make_uint4(0u, 0u, 0u, 0u), vr = kr;
    if (reg) {
        if (j < p.nreg) {
            kr = *reinterpret_cast<const uint4*>(p.kreg + (long)j * p.H * 64 + head * 64 + 8 * c);
            vr = *reinterpret_cast<const uint4*>(p.vreg + (long)j * p.H * 64 + head * 64 + 8 * c);
        }
    } else if (idx < p.Lk) {
        const long row = ccv_patch_row(idx, p.perm_hw, p.perm_w);
        kr = *reinterpret_cast<const uint4*>(p.k + (long)b * p.k_bso + row * p.k_ls + head * 64 + 8 * c);
        vr = *reinterpret_cast<const uint4*>(p.v + (long)b * p.v_bso + row * p.v_ls + head * 64 + 8 * c);
        const uint4 qr = *reinterpret_cast<const uint4*>(p.q + (long)b * p.q_bso + row * p.q_ls + head * 64 + 8 * c);
        *reinterpret_cast<uint2*>(q8 + ((long)b * p.Lq + row) * (p.H * 64) + head * 64 + 8 * c) = pack8_fp8(qr, 1.f / sq);
    }
    const long slot = (long)bh * (nblk + 1) + blk;
    *reinterpret_cast<uint2*>(k8b + (slot * 32 + j) * 64 + 8 * c) = pack8_fp8(kr, 1.f / sk);
    *reinterpret_cast<uint2*>(tile + j * 64 + 8 * c) = pack8_fp8(vr, 1.f / sv);
    __syncthreads();
    const int d = threadIdx.x >> 2, pc = threadIdx.x & 3;      // V^T row d, 8 positions 8 pc .. 8 pc + 7
    uint8_t out[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) out[q] = tile[vt_key_of_pos(8 * pc + q) * 64 + d];
    uint2 o2;
    o2.x = out[0] | (out[1] << 8) | (out[2] << 16) | ((unsigned)out[3] << 24);
    o2.y = out[4] | (out[5] << 8) | (out[6] << 16) | ((unsigned)out[7] << 24);
    *reinterpret_cast<uint2*>(v8tb + (slot * 64 + d) * 32 + 8 * pc) = o2;
}

// Online-softmax update of one 32-key x 32-query score block (same arithmetic as softmax_block32 in ccv_attn.hip) with P
// emitted as e4m3 for the PV MFMA: pfo[s2] = 8 values = accumulator registers 8 s2 .. 8 s2 + 7.
__device__ __forceinline__ void softmax_block32_fp8(f32x16& sa, uint32_t w, bool all_visible, int hh, float sl2, float& m_r, float& l_r,
                                                   f32x16 (&oa)[2], long (&pfo)[2]) {
    if (!all_visible) {
        const int wsh = (int)(w >> (4 * hh));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int keep = __builtin_amdgcn_sbfe(wsh, (i & 3) + 8 * (i >> 2), 1);
            float sel;
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(sel) : "v"(keep), "v"(sa[i]), "v"(NEG_INF));
            sa[i] = sel;
        }
    }
    float tmax = NEG_INF;
#pragma unroll
    for (int i = 0; i < 16; i += 2) tmax = fmaxf(fmaxf(tmax, sa[i]), sa[i + 1]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sl2;
    const float m_new = fmaxf(m_r, tmax);
    const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
    if (!__all(m_new == m_r)) {
        const float alpha = __builtin_amdgcn_exp2f(m_r - m_use);
        l_r *= alpha;
        oa[0] = oa[0] * alpha;
        oa[1] = oa[1] * alpha;
        m_r = m_new;
    }
    const f32x2 scale2 = {sl2, sl2}, negm2 = {-m_use, -m_use};
    f32x2 psum2 = {0.f, 0.f};
    float pv[16];
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const f32x2 s2 = {sa[i], sa[i + 1]};
        const f32x2 x = __builtin_elementwise_fma(s2, scale2, negm2);
        const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
        pv[i] = e[0];
        pv[i + 1] = e[1];
        psum2 = psum2 + e;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const unsigned lo = pack4_fp8(pv[8 * s2], pv[8 * s2 + 1], pv[8 * s2 + 2], pv[8 * s2 + 3]);
        const unsigned hi = pack4_fp8(pv[8 * s2 + 4], pv[8 * s2 + 5], pv[8 * s2 + 6], pv[8 * s2 + 7]);
        pfo[s2] = (long)(((unsigned long long)hi << 32) | lo);
    }
    l_r += psum2[0] + psum2[1];
}

__device__ unsigned int g_sparse8_ctr[64];
__global__ void sparse8_ctr_reset(int slot) {
    if (threadIdx.x == 0) g_sparse8_ctr[slot] = 0u;
}

// p.q = q8 rows [(b Lq)][H*64] bytes, p.k = k8b, p.v = v8tb (blocked, see above), p.o bf16 as the bf16 kernel writes it.
__global__ __launch_bounds__(256, 2) void attn_sparse_fp8_kernel(const CcvAttn p, const float* scales, int nblk, int slot) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[4 * 2 * 4096 + 4 * 512];  // [wave][stage][K 2 KiB | V^T 2 KiB] + mask words
    unsigned char* smw = sm + 4 * 2 * 4096;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    unsigned char* ring = sm + wave * 8192;
    unsigned char* mwring = smw + wave * 512;
    constexpr int DONE = 0x7fffffff;
    const uint8_t* q8 = reinterpret_cast<const uint8_t*>(p.q);
    const uint8_t* k8b = reinterpret_cast<const uint8_t*>(p.k);
    const uint8_t* v8tb = reinterpret_cast<const uint8_t*>(p.v);
    const bool has_reg = p.nreg > 0;
    const int ngroups = (p.Lq + 63) >> 6;
    const long total = (long)p.B * p.H * ngroups;
    const int nbh = p.B * p.H;
    const int C = p.H * 64;

    // 16-byte piece `lane + 64 jj` of a 2 KiB block image; the LDS image is swizzled at 16-byte granularity through the
    // SOURCE piece each lane fetches (DMA writes LDS linearly):
    //   K  rows of 64 B (4 pieces): piece (row, c16) sits at slot c16 ^ ((row >> 2) & 3)
    //   V^T rows of 32 B (2 pieces): piece (d, c16) sits at slot c16 ^ ((d >> 3) & 1)
    int ksrc[2], vsrc[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int L = lane + 64 * jj;
        const int krow = L >> 2, kslot = L & 3;
        ksrc[jj] = (krow * 4 + (kslot ^ ((krow >> 2) & 3))) * 16;
        const int vrow = L >> 1, vslot = L & 1;
        vsrc[jj] = (vrow * 2 + (vslot ^ ((vrow >> 3) & 1))) * 16;
    }

  for (;;) {
    unsigned int idx = 0;
    if (lane == 0) idx = atomicAdd(p.queue_counters ? p.queue_counters : &g_sparse8_ctr[slot], 1u);
    const long item = (long)(unsigned int)__builtin_amdgcn_readfirstlane((int)idx);
    if (item >= total) break;
    const int rank = (int)(item / nbh);
    const int bh = (int)(item % nbh);
    const int head = bh % p.H, b = bh / p.H;
    const int mb = b % p.mask_nb;
    const int qg = p.group_order ? p.group_order[(long)mb * p.order_bs + rank] : rank;
    const int q0 = qg * 64;
    const float sq = scales[bh * 4 + 0], sk = scales[bh * 4 + 1], sv = scales[bh * 4 + 2];
    const float sl2 = p.scale * 1.4426950408889634f * sq * sk;

    long qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = min(q0 + 32 * qb + r, p.Lq - 1);
        const uint8_t* qp = q8 + ((long)b * p.Lq + ccv_patch_row(qi, p.perm_hw, p.perm_w)) * C + head * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qb][s] = *reinterpret_cast<const long*>(qp + 16 * s);
    }
    const uint32_t* wrow = p.wave_bits + (long)mb * p.wave_bs + (long)qg * p.wave_words;
    const uint8_t* kslice = k8b + (long)bh * (nblk + 1) * 2048;
    const uint8_t* vslice = v8tb + (long)bh * (nblk + 1) * 2048;

    const uint32_t my_word = (lane < p.wave_words) ? wrow[lane] : 0u;
    int widx = -1;
    uint32_t cbits = 0;
    bool reg_pending = has_reg;
    auto next_block = [&]() -> int {
        if (reg_pending) { reg_pending = false; return -1; }
        while (cbits == 0) {
            if (++widx >= p.wave_words) return DONE;
            cbits = (uint32_t)__builtin_amdgcn_readlane((int)my_word, widx);
        }
        const int bit = __builtin_ctz(cbits);
        cbits &= cbits - 1;
        return widx * 32 + bit;
    };
    auto issue = [&](int blk, int stage) {   // 2 K + 2 V^T DMA pieces + 1 mask-word DMA = 5 vector-memory operations
        unsigned char* sK = ring + stage * 4096;
        unsigned char* sV = sK + 2048;
        const long boff = (long)(blk < 0 ? nblk : blk) * 2048;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            __builtin_amdgcn_global_load_lds((gptr_t*)(kslice + boff + ksrc[jj]), (lptr_t*)(sK + jj * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t*)(vslice + boff + vsrc[jj]), (lptr_t*)(sV + jj * 1024), 16, 0, 0);
        }
        const int wi = blk < 0 ? 0 : blk;
        const uint32_t* mp = p.mask_bits + (long)mb * p.mask_bs + (long)min(q0 + lane, p.Lq - 1) * p.mask_words + wi;
        __builtin_amdgcn_global_load_lds((gptr_t*)mp, (lptr_t*)(mwring + stage * 256), 4, 0, 0);
    };

    float m_run[2] = {NEG_INF, NEG_INF}, l_run[2] = {0.f, 0.f};
    f32x16 oacc[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[qb][d][i] = 0.f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int blk, int stage) {
        const unsigned char* sK = ring + stage * 4096;
        const unsigned char* sV = sK + 2048;
        const int left = blk < 0 ? p.nreg : min(32, p.Lk - 32 * blk);
        const uint32_t lim = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
        const uint32_t* wl = reinterpret_cast<const uint32_t*>(mwring + stage * 256);
        uint32_t mw[2];
        mw[0] = (blk < 0 ? 0xffffffffu : wl[r]) & lim;
        mw[1] = (blk < 0 ? 0xffffffffu : wl[32 + r]) & lim;
        const bool on0 = __ballot(mw[0] != 0u) != 0ull;
        const bool on1 = (q0 + 32 < p.Lq) && (__ballot(mw[1] != 0u) != 0ull);
        if (!(on0 || on1)) return;
        f32x16 sa0 = zero16, sa1 = zero16;
#pragma unroll
        for (int s = 0; s < 4; ++s) {        // K row r, bytes 16 s + 8 hh .. + 7: piece s, half hh
            const long kf = *reinterpret_cast<const long*>(sK + (r * 4 + (s ^ ((r >> 2) & 3))) * 16 + 8 * hh);
            if (on0) sa0 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kf, qf[0][s], sa0, 0, 0, 0);
            if (on1) sa1 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kf, qf[1][s], sa1, 0, 0, 0);
        }
        long pf0[2], pf1[2];
        auto softmax_block = [&](f32x16& sa, uint32_t w, float& m_r, float& l_r, f32x16 (&oa)[2], long (&pfo)[2]) {
            const bool all_visible = __builtin_amdgcn_readfirstlane((int)__all(w == 0xffffffffu)) != 0;
            softmax_block32_fp8(sa, w, all_visible, hh, sl2, m_r, l_r, oa, pfo);
        };
        if (on0) softmax_block(sa0, mw[0], m_run[0], l_run[0], oacc[0], pf0);
        if (on1) softmax_block(sa1, mw[1], m_run[1], l_run[1], oacc[1], pf1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int d = 0; d < 2; ++d) {     // V^T row dd = 32 d + r, positions 16 s2 + 8 hh .. + 7: piece s2, half hh
                const int dd = 32 * d + r;
                const long vf = *reinterpret_cast<const long*>(sV + (dd * 2 + (s2 ^ ((dd >> 3) & 1))) * 16 + 8 * hh);
                if (on0) oacc[0][d] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf, pf0[s2], oacc[0][d], 0, 0, 0);
                if (on1) oacc[1][d] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vf, pf1[s2], oacc[1][d], 0, 0, 0);
            }
    };

    // retire the ordinary vector loads where hipcc can see it (see attn_sparse_kernel)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) asm volatile("" ::"v"(qf[qb][s4]));
    asm volatile("" ::"v"(my_word), "v"(sl2), "v"(sv));

    int cur = next_block(), stage = 0;
    if (cur != DONE) issue(cur, 0);
    while (cur != DONE) {
        const int nxt = next_block();
        if (nxt != DONE) {
            issue(nxt, stage ^ 1);
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // the 5 operations of `nxt` may stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        compute(cur, stage);
        cur = nxt;
        stage ^= 1;
    }

    const long bo = b / p.inner, bi = b % p.inner;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float wgt = (l_tot > 0.f ? 1.0f / l_tot : 0.f) * sv;
        const int q = q0 + 32 * qb + r;
        if (q < p.Lq) {
            uint16_t* op = p.o + bo * p.o_bso + bi * p.o_bsi + (long)ccv_patch_row(q, p.perm_hw, p.perm_w) * p.o_ls + head * 64;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dd = 32 * d + 8 * g4 + 4 * hh;
                    uint2 pk = make_uint2(pack_bf16x2(oacc[qb][d][4 * g4] * wgt, oacc[qb][d][4 * g4 + 1] * wgt),
                                          pack_bf16x2(oacc[qb][d][4 * g4 + 2] * wgt, oacc[qb][d][4 * g4 + 3] * wgt));
                    *reinterpret_cast<uint2*>(op + dd) = pk;
                }
        }
    }
  }
}

}  // namespace

extern "C" int64_t ccv_attn_fp8_ws_bytes(const CcvAttn* p, int64_t* q8_off, int64_t* k8_off, int64_t* v8_off, int64_t* scales_off,
                                         int64_t* amax_off) {
    if (p == nullptr || p->B <= 0 || p->H <= 0 || p->Lq <= 0 || p->Lk <= 0) return 0;
    const int64_t nblk = (p->Lk + 31) / 32, nbh = (int64_t)p->B * p->H;
    auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
    int64_t off = 0;
    const int64_t o_q = off; off += up((int64_t)p->B * p->Lq * p->H * 64);
    const int64_t o_k = off; off += up(nbh * (nblk + 1) * 2048);
    const int64_t o_v = off; off += up(nbh * (nblk + 1) * 2048);
    const int64_t o_s = off; off += up(nbh * 16);
    const int64_t o_a = off; off += up(nbh * 16);
    if (q8_off) *q8_off = o_q;
    if (k8_off) *k8_off = o_k;
    if (v8_off) *v8_off = o_v;
    if (scales_off) *scales_off = o_s;
    if (amax_off) *amax_off = o_a;
    return off;
}

extern "C" int ccv_attn_sparse_fp8_fwd(const CcvAttn* pp, void* ws, int64_t ws_bytes, void* stream) {
    CCV_REQUIRE(pp != nullptr && ws != nullptr, CCV_EINVAL, "ccv_attn_sparse_fp8_fwd: null params / workspace");
    const CcvAttn& p = *pp;
    CCV_REQUIRE(p.q && p.k && p.v && p.o, CCV_EINVAL, "ccv_attn_sparse_fp8_fwd: null q/k/v/o");
    CCV_REQUIRE(p.B > 0 && p.H > 0 && p.Lq > 0 && p.Lk == p.Lq && p.inner == 1, CCV_ESHAPE,
                "ccv_attn_sparse_fp8_fwd: self attention over one token sequence per batch expected (B=%d H=%d Lq=%d Lk=%d inner=%d)", p.B, p.H, p.Lq, p.Lk, p.inner);
    CCV_REQUIRE(p.mask_bits && p.wave_bits && p.mask_nb > 0 && p.mask_words * 32 >= p.Lk && (long)p.wave_words * 1024 >= p.Lk, CCV_EINVAL,
                "ccv_attn_sparse_fp8_fwd: needs mask_bits and wave_bits covering Lk");
    CCV_REQUIRE(!p.k2 && p.nreg <= 32 && (p.nreg == 0 || (p.kreg && p.vreg)), CCV_EINVAL, "ccv_attn_sparse_fp8_fwd: single context, at most 32 register tokens");
    CCV_REQUIRE((p.q_ls % 8 == 0) && (p.k_ls % 8 == 0) && (p.v_ls % 8 == 0) && (p.o_ls % 4 == 0), CCV_ESHAPE, "ccv_attn_sparse_fp8_fwd: token strides must keep 16-byte alignment");
    CCV_REQUIRE(p.perm_w == 0 || (p.perm_w % 8 == 0 && p.perm_hw > 0 && p.perm_hw % (4 * p.perm_w) == 0 && p.Lq % p.perm_hw == 0), CCV_ESHAPE,
                "ccv_attn_sparse_fp8_fwd: patch order needs W %% 8 == 0, H %% 4 == 0 and whole frames");
    int64_t o_q, o_k, o_v, o_s, o_a;
    const int64_t need = ccv_attn_fp8_ws_bytes(pp, &o_q, &o_k, &o_v, &o_s, &o_a);
    CCV_REQUIRE(ws_bytes >= need, CCV_EINVAL, "ccv_attn_sparse_fp8_fwd: workspace of %ld bytes needed", (long)need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint8_t* base = static_cast<uint8_t*>(ws);
    uint8_t *q8 = base + o_q, *k8b = base + o_k, *v8tb = base + o_v;
    float* scales = reinterpret_cast<float*>(base + o_s);
    unsigned int* amax = reinterpret_cast<unsigned int*>(base + o_a);
    const int nblk = (p.Lk + 31) / 32, nbh = p.B * p.H;
    if (hipMemsetAsync(amax, 0, (size_t)nbh * 16, st) != hipSuccess) { ccv_set_error("ccv_attn_sparse_fp8_fwd: memset failed"); return CCV_EINVAL; }
    int chunks = (p.Lk + 31) / 32;
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(fp8_amax_kernel, dim3(nbh, chunks), dim3(256), 0, st, p, amax);
    hipLaunchKernelGGL(fp8_pack_kernel, dim3(nblk + 1, nbh), dim3(256), 0, st, p, amax, q8, k8b, v8tb, scales, nblk);
    CcvAttn a = p;
    a.q = reinterpret_cast<const uint16_t*>(q8);
    a.k = reinterpret_cast<const uint16_t*>(k8b);
    a.v = reinterpret_cast<const uint16_t*>(v8tb);
    static std::atomic<int> next_slot{0};
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const int slot = next_slot.fetch_add(1) & 63;
    const long groups = (long)((p.Lq + 63) / 64) * nbh;
    const long wgs = (groups + 3) / 4 < 2l * n_cu ? (groups + 3) / 4 : 2l * n_cu;
    if (!p.queue_counters) hipLaunchKernelGGL(sparse8_ctr_reset, dim3(1), dim3(64), 0, st, slot);
    hipLaunchKernelGGL(attn_sparse_fp8_kernel, dim3((unsigned)wgs), dim3(256), 0, st, a, scales, nblk, slot);
    CCV_LAUNCH_CHECK("ccv_attn_sparse_fp8_fwd");
    return CCV_OK;
}
