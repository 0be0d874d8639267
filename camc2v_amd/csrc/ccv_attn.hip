// ccv_attn_fwd: fused attention forward for head dim 64 on bf16 MFMA (gfx950).
//
// One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.
// K/V are walked in 64-key tiles staged global -> registers -> LDS (shared by the 4 waves).
// Per tile and wave:
//   S^T[key][query] = K . Q^T        v_mfma_f32_32x32x16_bf16, K rows as A, Q rows as B
//                                     => a lane holds one query column: the softmax row
//                                        reduction is in-lane + one exchange with lane^32
//   online softmax in fp32 (base 2), optional bit mask / key bound
//   O^T[d][query] += V^T . P^T       the S^T accumulator, converted to bf16 in place, IS the
//                                     B operand (no LDS round trip); V^T fragments come from
//                                     the row-major V tile through ds_read_b64_tr_b16
// Masked (epipolar) attention skips whole 128x64 tiles via caller-provided tile flags and,
// per wave, 32x32 blocks whose mask words are all zero (both exact).  Register tokens are a
// leading, always-visible K/V segment.  An optional second context (image tokens) is a second
// softmax pass over the same Q tile, added with a gate.
#include <atomic>
#include <type_traits>

#include "ccv_common.h"

namespace {

constexpr int KT = 64;          // keys per tile
constexpr int K_ROW = 128;      // bytes per K row in LDS (64 bf16), XOR swizzled
constexpr int V_ROW = 192;      // bytes per V row in LDS (128 + 64 pad: tr reads conflict free)
constexpr int VT_ROW = 144;     // bytes per V^T row in LDS (variant 1)
constexpr float NEG_INF = -__builtin_inff();

// First stored row of the 32-token patch that starts at permuted index k0 (csrc/ccv_common.h: ccv_patch_row with within = 0), in wave-uniform
// scalar arithmetic: shifts for the shipped power-of-two latent sizes instead of ~22-instruction scalar divisions per block.
struct PatchGeom {
    int hw, w, ppr, quads, pow2, sh_hw, sh_ppr;
};
__device__ __forceinline__ PatchGeom patch_geom(int perm_hw, int perm_w) {
    PatchGeom g;
    g.hw = perm_hw;
    g.w = perm_w;
    g.ppr = perm_w >> 3;
    g.pow2 = perm_w && (perm_hw & (perm_hw - 1)) == 0 && (g.ppr & (g.ppr - 1)) == 0;
    g.quads = perm_w && g.ppr > 2 && !(g.ppr & 1) && !((perm_hw / perm_w) & 7);      // 2x2 quads of patches (ccv_patch_row; two patches per row: the same order)
    g.sh_hw = g.pow2 ? __builtin_ctz(perm_hw) : 0;
    g.sh_ppr = g.pow2 ? __builtin_ctz(g.ppr) : 0;
    return g;
}
__device__ __forceinline__ int patch_first_row(int k0, const PatchGeom& g) {
    int f, patch, py, px;
    if (g.pow2) {
        f = k0 >> g.sh_hw;
        patch = (k0 & (g.hw - 1)) >> 5;
    } else {
        f = k0 / g.hw;
        patch = (k0 - f * g.hw) >> 5;
    }
    if (g.quads) {
        const int quad = patch >> 2, sub = patch & 3;
        int qy, qx;
        if (g.pow2) {
            qy = quad >> (g.sh_ppr - 1);
            qx = quad & ((g.ppr >> 1) - 1);
        } else {
            const int qpr = g.ppr >> 1;
            qy = quad / qpr;
            qx = quad - qy * qpr;
        }
        py = 2 * qy + (sub >> 1);
        px = 2 * qx + (sub & 1);
    } else if (g.pow2) {
        py = patch >> g.sh_ppr;
        px = patch & (g.ppr - 1);
    } else {
        py = patch / g.ppr;
        px = patch - py * g.ppr;
    }
    return f * g.hw + py * 4 * g.w + px * 8;
}

struct Seg {            // one key/value segment of a softmax pass
    const uint16_t* k;
    const uint16_t* v;
    long k_ls, v_ls;
    int len;
    int masked;         // uses mask_bits / tile_flags
};

template <bool TR>
__global__ __launch_bounds__(256) void attn_kernel(const CcvAttn p) {
    __shared__ __attribute__((aligned(16))) unsigned char sK[KT * K_ROW];
    __shared__ __attribute__((aligned(16))) unsigned char sV[KT * V_ROW];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar for the compiler (M0 of the K/V DMA)
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y, qblk = blockIdx.x;
    const long bo = b / p.inner, bi = b % p.inner;
    const int q0 = qblk * 128 + wave * 32;
    const bool wave_active = q0 < p.Lq;   // wave-uniform
    const int qi = min(q0 + r, p.Lq - 1); // clamped query index of this lane

    // ---- Q fragments (B operand of S^T): Q[qi][16 s + 8 hh + j] --------------------------
    bf16x8 qf[4];
    {
        const uint16_t* qp = p.q + bo * p.q_bso + bi * p.q_bsi + (long)qi * p.q_ls + head * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    const float sl2 = p.scale * 1.4426950408889634f;  // softmax in base 2

    f32x16 ofin[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) ofin[d][i] = 0.f;

    // Flattened tile schedule: [register tokens][main K/V tiles][second-context tiles].
    // The second context is its own softmax pass: crossing into it finalises pass 0.
    const int n_reg = (p.kreg != nullptr && p.nreg > 0) ? 1 : 0;
    const int n_main = (p.Lk + KT - 1) / KT;
    const int n_two = (p.k2 != nullptr) ? (p.Lk2 + KT - 1) / KT : 0;
    const int n_total = n_reg + n_main + n_two;
    const uint8_t* flags = (p.mask_bits && p.tile_flags) ? p.tile_flags + (long)(b % p.mask_nb) * p.flags_bs + (long)qblk * p.flags_ktiles : nullptr;

    float m_run = NEG_INF, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;

    {
        {
            for (int it = 0; it < n_total; ++it) {
                if (it == n_reg + n_main) {
                    // ---- end of pass 0: normalise into the final accumulator, restart the softmax
                    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
                    const float wgt = l_tot > 0.f ? 1.0f / l_tot : 0.f;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int i = 0; i < 16; ++i) { ofin[d][i] = oacc[d][i] * wgt; oacc[d][i] = 0.f; }
                    m_run = NEG_INF;
                    l_run = 0.f;
                }
                Seg seg;
                int kt;
                if (it < n_reg) {
                    seg = Seg{p.kreg + head * 64, p.vreg + head * 64, (long)p.H * 64, (long)p.H * 64, p.nreg, 0};
                    kt = 0;
                } else if (it < n_reg + n_main) {
                    seg = Seg{p.k + bo * p.k_bso + bi * p.k_bsi + head * 64, p.v + bo * p.v_bso + bi * p.v_bsi + head * 64,
                              (long)p.k_ls, (long)p.v_ls, p.Lk, p.mask_bits != nullptr};
                    kt = it - n_reg;
                    if (flags && flags[kt] == 0) continue;  // block-uniform: nothing visible in this tile
                } else {
                    seg = Seg{p.k2 + bo * p.k2_bso + bi * p.k2_bsi + head * 64, p.v2 + bo * p.v2_bso + bi * p.v2_bsi + head * 64,
                              (long)p.k2_ls, (long)p.v2_ls, p.Lk2, 0};
                    kt = it - n_reg - n_main;
                }
                const int k0 = kt * KT;
                const int nvalid = min(KT, seg.len - k0);

                // ---- stage K and V tiles (zero fill past the end: 0 * garbage must stay 0) ----
                __syncthreads();  // previous tile fully consumed
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idx = tid + 256 * i;
                    const int key = idx >> 3, c = idx & 7;
                    uint4 kv = make_uint4(0u, 0u, 0u, 0u), vv = make_uint4(0u, 0u, 0u, 0u);
                    if (key < nvalid) {
                        kv = *reinterpret_cast<const uint4*>(seg.k + (long)(k0 + key) * seg.k_ls + c * 8);
                        vv = *reinterpret_cast<const uint4*>(seg.v + (long)(k0 + key) * seg.v_ls + c * 8);
                    }
                    *reinterpret_cast<uint4*>(sK + key * K_ROW + ((c ^ ((key >> 1) & 7)) << 4)) = kv;
                    if (TR) {
                        *reinterpret_cast<uint4*>(sV + key * V_ROW + (c << 4)) = vv;
                    } else {
                        const uint16_t* e = reinterpret_cast<const uint16_t*>(&vv);
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            *reinterpret_cast<uint16_t*>(sV + (8 * c + j) * VT_ROW + key * 2) = e[j];
                    }
                }
                __syncthreads();
                if (!wave_active) continue;  // wave-uniform; barriers above are still reached next iteration

                // ---- which 32-key blocks does this wave need? --------------------------------
                uint32_t mw[2] = {0xffffffffu, 0xffffffffu};
                if (seg.masked) {
                    const uint32_t* mrow = p.mask_bits + (long)(b % p.mask_nb) * p.mask_bs + (long)qi * p.mask_words;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const int w = (k0 >> 5) + kb;
                        mw[kb] = (w < p.mask_words) ? mrow[w] : 0u;
                    }
                }
                bool on[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    on[kb] = (32 * kb < nvalid) && (__ballot(mw[kb] != 0u) != 0ull);
                if (!on[0] && !on[1]) continue;

                // ---- S^T = K Q^T ------------------------------------------------------------
                f32x16 sacc[2];
                float tmax = NEG_INF;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
                    if (!on[kb]) continue;
                    const int krow = 32 * kb + r;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int c = 2 * s + hh;
                        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + krow * K_ROW + ((c ^ ((krow >> 1) & 7)) << 4));
                        sacc[kb] = ccv_mfma_32x32x16(kf, qf[s], sacc[kb]);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int ko = (i & 3) + 8 * (i >> 2) + 4 * hh;  // key offset inside the 32 block
                        const bool vis = (32 * kb + ko < nvalid) && ((mw[kb] >> ko) & 1u);
                        const float sv = vis ? sacc[kb][i] * sl2 : NEG_INF;
                        sacc[kb][i] = sv;
                        tmax = fmaxf(tmax, sv);
                    }
                }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run, tmax);
                const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);  // m_run = -inf -> 0
                m_run = m_new;
                float psum = 0.f;
                bf16x8 pf[2][2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (!on[kb]) continue;
                    float pv[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        pv[i] = __builtin_amdgcn_exp2f(sacc[kb][i] - m_use);
                        psum += pv[i];
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
                        pf[kb][s2] = ccv_opnd8(pv[8 * s2], pv[8 * s2 + 1], pv[8 * s2 + 2], pv[8 * s2 + 3], pv[8 * s2 + 4], pv[8 * s2 + 5], pv[8 * s2 + 6], pv[8 * s2 + 7]);
                }
                l_run = l_run * alpha + psum;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int i = 0; i < 16; ++i) oacc[d][i] *= alpha;

                // ---- O^T += V^T P^T -----------------------------------------------------------
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (!on[kb]) continue;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int kb0 = 32 * kb + 16 * s2 + 4 * hh;  // keys kb0..kb0+3 and kb0+8..kb0+11
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            bf16x8 vf;
                            if (TR) {
                                const int li = lane & 15, g = (lane >> 4) & 1;
                                const int dcol = 32 * d + 16 * g + 4 * (li & 3);
                                const unsigned char* a0 = sV + (kb0 + (li >> 2)) * V_ROW + dcol * 2;
                                const bf16x4 lo = ccv_ds_read_tr16(a0);
                                const bf16x4 hi = ccv_ds_read_tr16(a0 + 8 * V_ROW);
#pragma unroll
                                for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
                            } else {
                                const unsigned char* a0 = sV + (32 * d + r) * VT_ROW + kb0 * 2;
                                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(a0);
                                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(a0 + 16);
#pragma unroll
                                for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
                            }
                            oacc[d] = ccv_mfma_32x32x16(vf, pf[kb][s2], oacc[d]);
                        }
                    }
                }
            }
        }
    }
    // ---- finish the last pass -------------------------------------------------------------
    {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float wgt = (l_tot > 0.f ? 1.0f / l_tot : 0.f) * (n_two > 0 ? p.gate2 : 1.0f);
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) ofin[d][i] += oacc[d][i] * wgt;
    }

    // ---- store O[query][head*64 + d]: lane holds column `query`, rows d ------------------
    if (wave_active && q0 + r < p.Lq) {
        uint16_t* op = p.o + bo * p.o_bso + bi * p.o_bsi + (long)(q0 + r) * p.o_ls + head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = 32 * d + 8 * g4 + 4 * hh;
                uint2 pk = make_uint2(pack_bf16x2(ofin[d][4 * g4], ofin[d][4 * g4 + 1]),
                                      pack_bf16x2(ofin[d][4 * g4 + 2], ofin[d][4 * g4 + 3]));
                *reinterpret_cast<uint2*>(op + dd) = pk;
            }
    }
}


// =================================================================================================
// attn2_kernel: second-generation forward for single-context attention (self / epipolar / temporal).
//   * a wave owns 64 queries (two 32-row blocks): every K and V^T fragment read from LDS feeds two
//     MFMAs, halving the LDS bytes per FLOP (the first kernel was LDS-read bound at 2 workgroups/CU);
//   * K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a 2-deep ring: the DMA of the
//     next non-empty tile is in flight while the current one is multiplied; no VGPR staging, no ds_write;
//     the LDS swizzles are applied to the per-lane SOURCE address (DMA writes 1 KiB pieces linearly);
//   * softmax: scale folded into one FMA per score, accumulator rescale only when a row maximum moved,
//     mask bit tests only on 32x32 blocks that are neither full nor out of range.
// Workgroup = 4 waves = 256 queries of one (batch, head).  Tile flags are given per 128 query rows.
// =================================================================================================
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
__device__ __attribute__((aligned(16))) unsigned char g_attn_zero_line[16];

typedef float f32x2 __attribute__((ext_vector_type(2)));

// V^T fragments of one 32-key block [32 rows x 128 B, 64-byte halves swapped on rows with ((row >> 1) & 1)] for the four
// (16-key half s2, 32-column half d) PV MFMAs, read with ds_read_b64_tr_b16 from INLINE ASM.  Through
// __builtin_amdgcn_ds_read_tr16_b64 hipcc's waitcnt pass puts an s_waitcnt vmcnt(0) in front of the first read (it
// cannot tell that the LDS-DMA writes still in flight go to another ring stage), which drains the prefetch ring once
// per block: the next block's DMA then overlaps only QK^T + softmax instead of the whole block.  The reads are waited for
// here (lgkmcnt(0)): hipcc does not see them.
__device__ __forceinline__ void read_vt_block(const unsigned char* sV, int lane, int hh, bf16x8 (&vf)[2][2]) {
    const int li = lane & 15, g = (lane >> 4) & 1;
    const int row0 = 4 * hh + (li >> 2);
    const int x = ((row0 >> 1) & 1) << 6;
    const int c = (16 * g + 4 * (li & 3)) * 2;                               // < 64: (c ^ x) = c + x, ((64 + c) ^ x) = c + (64 - x)
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(sV) + row0 * 128 + c;
    const uint32_t ad0 = base + x, ad1 = base + 64 - x;
    bf16x4 t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
        "ds_read_b64_tr_b16 %2, %9\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:1024\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:2048\n\t"
        "ds_read_b64_tr_b16 %5, %8 offset:3072\n\t"
        "ds_read_b64_tr_b16 %6, %9 offset:2048\n\t"
        "ds_read_b64_tr_b16 %7, %9 offset:3072\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(ad0), "v"(ad1)
        : "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        vf[0][0][j] = t0[j]; vf[0][0][4 + j] = t1[j];
        vf[0][1][j] = t2[j]; vf[0][1][4 + j] = t3[j];
        vf[1][0][j] = t4[j]; vf[1][0][4 + j] = t5[j];
        vf[1][1][j] = t6[j]; vf[1][1][4 + j] = t7[j];
    }
}

// Online-softmax update of one 32-key x 32-query score block held as the 32x32x16 MFMA accumulator (lane = query
// column lane&31, register i = key row (i&3) + 8 (i>>2) + 4 hh): mask -> running max -> rescale -> P = exp2(S c - m)
// in bf16 fragments for the PV MFMA.  w = the query's 32 mask bits for this key block (ignored when all_visible,
// which must be wave-uniform).  Instruction budget per element: v_bfe_i32 + v_bfi_b32 for the mask (the select is a
// bit-field insert of -inf under a sign-extended mask bit), half a v_max3, half a v_pk_fma, one v_exp, half a v_pk_add.
__device__ __forceinline__ void softmax_block32(f32x16& sa, uint32_t w, bool all_visible, int hh, float sl2, float& m_r, float& l_r,
                                               f32x16 (&oa)[2], bf16x8 (&pfo)[2]) {
    if (!all_visible) {
        const int wsh = (int)(w >> (4 * hh));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int keep = __builtin_amdgcn_sbfe(wsh, (i & 3) + 8 * (i >> 2), 1);       // 0 or -1
            float sel;   // (keep & s) | (~keep & -inf): hipcc lowers the C form to and + or, the instruction exists
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(sel) : "v"(keep), "v"(sa[i]), "v"(NEG_INF));
            sa[i] = sel;
        }
    }
    float tmax = NEG_INF;
#pragma unroll
    for (int i = 0; i < 16; i += 2) tmax = fmaxf(fmaxf(tmax, sa[i]), sa[i + 1]);
    {   // max over the two half-waves: v_permlane32_swap leaves {lo, lo} / {hi, hi} in the two registers -- no LDS round trip
        // (ds_bpermute) on the critical path of every block
        const uint32_t tb = __float_as_uint(tmax);
        const auto sw = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);
        asm("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(sw[0]), "v"(sw[1]));
        tmax *= sl2;
    }
    const float m_new = fmaxf(m_r, tmax);
    const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
    if (!__all(m_new == m_r)) {  // some row maximum moved: rescale the running sums
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
        const f32x2 x = __builtin_elementwise_fma(s2, scale2, negm2);          // -inf -> exp2 = 0
        const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
        pv[i] = e[0];
        pv[i + 1] = e[1];
        psum2 = psum2 + e;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
        pfo[s2] = ccv_opnd8(pv[8 * s2], pv[8 * s2 + 1], pv[8 * s2 + 2], pv[8 * s2 + 3], pv[8 * s2 + 4], pv[8 * s2 + 5], pv[8 * s2 + 6], pv[8 * s2 + 7]);
    l_r += psum2[0] + psum2[1];
}

// 1-D grid -> (query block, head, batch) with an XCD-aware remap: workgroups are dealt round-robin over the 8 XCDs,
// so giving XCD x the contiguous range [x*n/8, (x+1)*n/8) of (batch, head, query block) makes the workgroups that
// share an L2 walk the SAME (batch, head) K/V slice (4 MB at 32x32 latents = one XCD's L2) instead of eight
// different ones that evict each other to the Infinity Cache.  Bijective for any grid size; speed only.
__device__ __forceinline__ void attn_block_coords(int nq, int H, int& qblk, int& head, int& b) {
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    qblk = bid % nq;
    const int t = bid / nq;
    head = t % H;
    b = t / H;
}

// TWO: a second context (k2 / v2 / Lk2, the gated image tokens of the cross-attention, attention.py:205-209) with a softmax
// of its own follows the first in the same tile sequence: when the first tile of the second context comes up, the
// normalised result of the first is parked as packed bf16 (32 registers) and the running max / sum / accumulators start
// over; the epilogue writes o1 + gate2 * o2.  Unmasked calls only.
// NQB = 32-query blocks per wave: 2 (64 queries per wave, 256 per workgroup, two workgroups per CU) or -- round 4 -- 1 (32 queries per wave,
// 128 per workgroup: half the accumulators, four workgroups = 16 waves per CU, so that one wave's softmax runs under three others' MFMAs and
// LDS reads; the same arithmetic per query, bit-identical results).
template <bool MASKED, bool TWO = false, int NQB = 2>
__global__ __launch_bounds__(256, NQB == 1 ? 4 : 2) void attn2_kernel(const CcvAttn p) {
    static_assert(NQB == 1 || NQB == 2, "one or two 32-query blocks per wave");
    static_assert(!(MASKED && NQB == 1), "the masked tiled form stays at 64 queries per wave: its 32-query instance gave wrong results for partially "
                                         "masked second key blocks of a tile (round 4, not understood) and is not built");
    constexpr int QW = 128 * NQB;        // queries per workgroup
    static_assert(!(MASKED && TWO), "the two-context form has no mask path");
    __shared__ __attribute__((aligned(16))) unsigned char sm[2 * 2 * KT * 128];  // [stage][K|V][64 rows][128 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar for the compiler (M0 of the K/V DMA)
    const int r = lane & 31, hh = lane >> 5;
    int qblk, head, b;
    attn_block_coords((p.Lq + QW - 1) / QW, p.H, qblk, head, b);
    const long bo = b / p.inner, bi = b % p.inner;
    const int q0 = qblk * QW + wave * (32 * NQB);
    const bool wave_active = q0 < p.Lq;

    bf16x8 qf[NQB][4];
    int qi[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        qi[qb] = min(q0 + 32 * qb + r, p.Lq - 1);   // index in mask / schedule order
        const uint16_t* qp = p.q + bo * p.q_bso + bi * p.q_bsi + (long)ccv_patch_row(qi[qb], p.perm_hw, p.perm_w) * p.q_ls + head * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qb][s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    const float sl2 = p.scale * 1.4426950408889634f;

    const int n_reg = (p.kreg != nullptr && p.nreg > 0) ? 1 : 0;
    const int n_main = (p.Lk + KT - 1) / KT;
    const int n_first = n_reg + n_main;                       // tiles of the first softmax
    const int n_total = n_first + (TWO ? (p.Lk2 + KT - 1) / KT : 0);
    constexpr bool masked = MASKED;
    const uint8_t* flags = (masked && p.tile_flags) ? p.tile_flags + (long)(b % p.mask_nb) * p.flags_bs : nullptr;
    const int fq0 = NQB * qblk, fq1 = min(NQB * qblk + NQB - 1, (p.Lq + 127) / 128 - 1);     // the flag rows (128 queries each) of this workgroup

    auto next_tile = [&](int it) {  // first schedulable tile index >= it (block-uniform)
        while (it < n_total) {
            if (it < n_reg || !flags) break;
            const int kt = it - n_reg;
            if (flags[(long)fq0 * p.flags_ktiles + kt] | flags[(long)fq1 * p.flags_ktiles + kt]) break;
            ++it;
        }
        return it;
    };
    const uint16_t* kmain = p.k + bo * p.k_bso + bi * p.k_bsi + head * 64;
    const uint16_t* vmain = p.v + bo * p.v_bso + bi * p.v_bsi + head * 64;
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_attn_zero_line);

    auto issue = [&](int it, int stage) {  // DMA tile `it` into ring slot `stage`
        const uint16_t *kb_, *vb_;
        long kls, vls;
        int len, k0;
        if (it < n_reg) { kb_ = p.kreg + head * 64; vb_ = p.vreg + head * 64; kls = vls = (long)p.H * 64; len = p.nreg; k0 = 0; }
        else if (TWO && it >= n_first) {
            kb_ = p.k2 + bo * p.k2_bso + bi * p.k2_bsi + head * 64;
            vb_ = p.v2 + bo * p.v2_bso + bi * p.v2_bsi + head * 64;
            kls = p.k2_ls; vls = p.v2_ls; len = p.Lk2; k0 = (it - n_first) * KT;
        }
        else { kb_ = kmain; vb_ = vmain; kls = p.k_ls; vls = p.v_ls; len = p.Lk; k0 = (it - n_reg) * KT; }
        unsigned char* sK = sm + stage * (2 * KT * 128);
        unsigned char* sV = sK + KT * 128;
#if defined(__HIP_DEVICE_COMPILE__)
        if ((p.perm_w == 0 || it < n_reg || (TWO && it >= n_first)) && ((long)(len - 1) * kls + 64) * 2 < 0x7ff00000l && ((long)(len - 1) * vls + 64) * 2 < 0x7ff00000l) {
            // keys stored in the order they are visited: the tile through buffer descriptors over this (batch, head) slice -- a lane's
            // offset is row * stride + its swizzled column, the tile's first row rides in the scalar offset, and rows past the last key
            // fall outside the descriptor's range: the hardware writes zeros for them (no zero line, no per-lane pointer arithmetic)
            const int kls32 = (int)kls, vls32 = (int)vls;
            const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(kb_), 0, ((len - 1) * kls32 + 64) * 2, 0x00020000);
            const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(vb_), 0, ((len - 1) * vls32 + 64) * 2, 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int piece = wave * 2 + j;             // 8 rows of 128 B
                const int row = 8 * piece + (lane >> 3), pc = lane & 7;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lptr_t*)(sK + piece * 1024), 16, (row * kls32 + ((pc ^ ((row >> 1) & 7)) << 3)) * 2, k0 * kls32 * 2, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lptr_t*)(sV + piece * 1024), 16, (row * vls32 + ((pc ^ (((row >> 1) & 1) << 2)) << 3)) * 2, k0 * vls32 * 2, 0, 0);
            }
            return;
        }
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave * 2 + j;             // 8 rows of 128 B
            const int row = 8 * piece + (lane >> 3), pc = lane & 7;
            const bool ok = (k0 + row) < len;
            const long krow_g = (it < n_reg || (TWO && it >= n_first)) ? (long)(k0 + row) : (long)ccv_patch_row(k0 + row, p.perm_hw, p.perm_w);
            const uint16_t* gk = ok ? kb_ + krow_g * kls + ((pc ^ ((row >> 1) & 7)) << 3) : zero;
            const uint16_t* gv = ok ? vb_ + krow_g * vls + ((pc ^ (((row >> 1) & 1) << 2)) << 3) : zero;
            __builtin_amdgcn_global_load_lds((gptr_t*)gk, (lptr_t*)(sK + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t*)gv, (lptr_t*)(sV + piece * 1024), 16, 0, 0);
        }
    };

    float m_run[NQB], l_run[NQB];
    f32x16 oacc[NQB][2];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) { m_run[qb] = NEG_INF; l_run[qb] = 0.f; }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[qb][d][i] = 0.f;

    uint32_t o1pk[TWO ? NQB : 1][2][8];   // first context's normalised output, packed bf16 pairs
    // mask words of a tile: [query block][32-key block]; loaded one tile ahead so their latency hides behind
    // the current tile's math (register tokens and unmasked calls see all-ones)
    auto load_words = [&](int it, uint32_t (&w)[NQB][2]) {
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                w[qb][kb] = 0xffffffffu;
                if (masked && it >= n_reg && it < n_total) {
                    const int wi = (((it - n_reg) * KT) >> 5) + kb;
                    w[qb][kb] = (wi < p.mask_words)
                                    ? p.mask_bits[(long)(b % p.mask_nb) * p.mask_bs + (long)qi[qb] * p.mask_words + wi] : 0u;
                }
            }
    };

    // retire the Q-fragment loads where hipcc can see it (see attn_sparse_kernel): otherwise every use of qf in the
    // loop is preceded by s_waitcnt vmcnt(0), which waits for the NEXT tile's DMA and serialises the ring
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) asm volatile("" ::"v"(qf[qb][s4]));

    int cur = next_tile(0), stage = 0;
    uint32_t mwc[NQB][2], mwn[NQB][2];
    if (cur < n_total) issue(cur, 0);
    load_words(cur, mwc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    while (cur < n_total) {
        const int nxt = next_tile(cur + 1);
        if (nxt < n_total) issue(nxt, stage ^ 1);
        load_words(nxt, mwn);

        if (TWO && cur == n_first) {   // block-uniform: park the first softmax's result, start the second
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
                const float wgt = l_tot > 0.f ? 1.0f / l_tot : 0.f;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        o1pk[TWO ? qb : 0][d][e] = pack_bf16x2(oacc[qb][d][2 * e] * wgt, oacc[qb][d][2 * e + 1] * wgt);
                        oacc[qb][d][2 * e] = 0.f;
                        oacc[qb][d][2 * e + 1] = 0.f;
                    }
                m_run[qb] = NEG_INF;
                l_run[qb] = 0.f;
            }
        }
        if (wave_active) {
            const bool is_reg = cur < n_reg;
            const bool second = TWO && cur >= n_first;
            const int k0 = is_reg ? 0 : (second ? (cur - n_first) * KT : (cur - n_reg) * KT);
            const int nvalid = min(KT, (is_reg ? p.nreg : (second ? p.Lk2 : p.Lk)) - k0);
            const unsigned char* sK = sm + stage * (2 * KT * 128);
            const unsigned char* sV = sK + KT * 128;
            uint32_t mw[NQB][2];
            bool on[NQB][2];
            bool any_on = false;
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    uint32_t w = mwc[qb][kb];
                    const int left = nvalid - 32 * kb;  // keys of this 32-block that exist
                    if (left < 32) w = (left <= 0) ? 0u : (w & ((1u << left) - 1u));
                    mw[qb][kb] = w;
                    on[qb][kb] = (q0 + 32 * qb < p.Lq) && (__ballot(w != 0u) != 0ull);
                    any_on |= on[qb][kb];
                }
            if (any_on) {
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                // one online-softmax step per 32-key block: S^T (8 MFMAs) -> softmax -> O^T += V^T P^T (8 MFMAs);
                // every K / V^T fragment read from LDS feeds both query blocks
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    bool any_q = false;
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb) any_q |= on[qb][kb];
                    if (!any_q) continue;
                    f32x16 sa[NQB];
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb) sa[qb] = zero16;
                    const int krow = 32 * kb + r;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int c = 2 * s + hh;
                        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + krow * 128 + ((c ^ ((krow >> 1) & 7)) << 4));
#pragma unroll
                        for (int qb = 0; qb < NQB; ++qb)
                            if (!MASKED || on[qb][kb]) sa[qb] = ccv_mfma_32x32x16(kf, qf[qb][s], sa[qb]);
                    }
                    bf16x8 pf[NQB][2];
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb)
                        if (!MASKED || on[qb][kb]) {
                            const bool all_visible = __builtin_amdgcn_readfirstlane((int)__all(mw[qb][kb] == 0xffffffffu)) != 0;
                            softmax_block32(sa[qb], mw[qb][kb], all_visible, hh, sl2, m_run[qb], l_run[qb], oacc[qb], pf[qb]);
                        }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const int kb0 = 32 * kb + 16 * s2 + 4 * hh;
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            const int li = lane & 15, g = (lane >> 4) & 1;
                            const int row0 = kb0 + (li >> 2);
                            const int colb = (32 * d + 16 * g + 4 * (li & 3)) * 2;  // byte column inside the 128-B row
                            // rows row0 and row0+8 have the same ((row>>1)&1): one swizzle term serves both reads
                            const unsigned char* a0 = sV + row0 * 128 + (colb ^ (((row0 >> 1) & 1) << 6));
                            const bf16x4 lo = ccv_ds_read_tr16(a0);
                            const bf16x4 hi = ccv_ds_read_tr16(a0 + 8 * 128);
                            bf16x8 vf;
#pragma unroll
                            for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
#pragma unroll
                            for (int qb = 0; qb < NQB; ++qb)
                                if (!MASKED || on[qb][kb]) oacc[qb][d] = ccv_mfma_32x32x16(vf, pf[qb][s2], oacc[qb][d]);
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur = nxt;
        stage ^= 1;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) mwc[qb][kb] = mwn[qb][kb];
    }

#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float wgt = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        const int q = q0 + 32 * qb + r;
        if (wave_active && q < p.Lq) {
            uint16_t* op = p.o + bo * p.o_bso + bi * p.o_bsi + (long)ccv_patch_row(q, p.perm_hw, p.perm_w) * p.o_ls + head * 64;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dd = 32 * d + 8 * g4 + 4 * hh;
                    float o0 = oacc[qb][d][4 * g4] * wgt, o1 = oacc[qb][d][4 * g4 + 1] * wgt;
                    float o2 = oacc[qb][d][4 * g4 + 2] * wgt, o3 = oacc[qb][d][4 * g4 + 3] * wgt;
                    if (TWO) {
                        const uint32_t a = o1pk[TWO ? qb : 0][d][2 * g4], c = o1pk[TWO ? qb : 0][d][2 * g4 + 1];
                        o0 = bf16_to_f32((uint16_t)(a & 0xffffu)) + p.gate2 * o0;
                        o1 = bf16_to_f32((uint16_t)(a >> 16)) + p.gate2 * o1;
                        o2 = bf16_to_f32((uint16_t)(c & 0xffffu)) + p.gate2 * o2;
                        o3 = bf16_to_f32((uint16_t)(c >> 16)) + p.gate2 * o3;
                    }
                    uint2 pk = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
                    *reinterpret_cast<uint2*>(op + dd) = pk;
                }
        }
    }
}


// =================================================================================================
// attn_sparse_kernel: masked (epipolar) attention where only 15-20 % of the 32x32 score blocks hold a
// visible key.  Sharing 64-key tiles between the four waves of a workgroup (attn2_kernel<true>) makes every
// wave pay the DMA, the barriers and the slowest wave of every tile that ANY of them needs; here each wave
// is on its own: it owns 64 queries, walks the bitmap of 32-key blocks its query group needs (wave_bits,
// built once per clip next to the mask), DMAs just those K/V blocks into a wave-private 2-deep LDS ring and
// never meets a barrier.  Math per block is attn2's: S^T = K Q^T, in-lane softmax, O^T += V^T P^T.
// =================================================================================================
// Work distribution: a 64-query group costs between ~0.5x and ~1.4x the average (its bitmap popcount), and the
// 2560 groups of the 32x32-latent layers do not divide over the 2048 resident waves, so a static grid ends with
// half-empty CUs waiting for the last workgroups (measured: 47 % average wave occupancy).  The kernel is therefore
// persistent: every wave pulls its next item from a counter, and items are ordered longest first (p.group_order,
// built with the mask).
// XCD-local queues: an item is (batch-head slice, rank).  One slice's K and V are 4 MiB at 32x32 latents -- a whole
// XCD L2 -- and a launch has 10 (32x32) or 20 (16x16) slices; dealing items round-robin over the slices made every
// XCD stream ALL slices through its 4 MiB L2 (measured 1.5 GB fetched per launch against 80 MB of operands).  Each XCD
// therefore has a queue of its own (one counter per XCD): its "home" slices (slice % 8 == XCD) in full, plus an equal
// share of one of the nbh % 8 leftover slices (the ranks r with r % c == q, c = XCDs sharing that slice), merged in
// rank order so the queue is still longest-first.  A wave reads its XCD from HW_REG_XCC_ID, drains that queue and
// then helps the other XCDs' queues (x+1, x+2, ...), so any placement of workgroups finishes all the work; the
// placement only decides how local the K/V traffic is.
__device__ unsigned int g_sparse_ctr[64][8];

__global__ void sparse_ctr_reset(int slot) {
    if (threadIdx.x < 8) g_sparse_ctr[slot][threadIdx.x] = 0u;
}

// Kernel arguments: only what this kernel reads, token strides and slice sizes in 32 bits (checked by the launcher).  The whole
// CcvAttn is ~100 scalar registers of arguments; held live across the block loop it cost 79 scalar spills (v_readlane reloads
// inside the loop) and a scratch slot.
struct SparseArgs {
    const uint16_t* q; const uint16_t* k; const uint16_t* v; uint16_t* o;
    const uint16_t* kreg; const uint16_t* vreg;
    const uint32_t* mask_bits; const uint32_t* wave_bits; const int32_t* group_order; uint32_t* queue_counters;
    int64_t q_bso, q_bsi, k_bso, k_bsi, v_bso, v_bsi, o_bso, o_bsi;
    int32_t q_ls, k_ls, v_ls, o_ls;
    int32_t mask_bs, mask_words, mask_nb, wave_bs, wave_words, order_bs;
    int32_t B, inner, H, Lq, Lk, nreg, perm_hw, perm_w;
    float scale;
    int32_t slot, use_xcd_queues;
    const int32_t* wg_order; int32_t wg_order_bs;   // attn_shared_kernel: items (groups of NW * 32 queries) longest first
    // attn_shared_kernel, key-split items (see the kernel): ranks >= split_rank0 come in `split_parts` parts; workspace = one counter per
    // split (slice, rank) followed by the parts' partial results
    int32_t split_rank0, split_parts;
    uint32_t* split_ctr; float* split_part;
};

__global__ __launch_bounds__(256, 2) void attn_sparse_kernel(const SparseArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // (buffer-descriptor builtins: device pass only)
    const int slot = p.slot, use_xcd_queues = p.use_xcd_queues;
    // aligned(256): the seed-table address is formed as (nibble << 4) | lut (v_lshl_or_b32), which needs the low 8 bits of the table's
    // LDS address clear (table offset 0x10800 + a 256-byte aligned base)
    __shared__ __attribute__((aligned(256))) unsigned char sm[4 * 2 * 8192 + 4 * 512 + 256];  // [wave][stage][K 4 KiB | V 4 KiB] + mask words + seed table
    static_assert((4 * 2 * 8192 + 4 * 512) % 256 == 0, "seed table offset must keep the low 8 address bits clear");
    // mask nibble -> four fp32 accumulator seeds (0 for a visible key, -inf for a masked one): the score MFMA chain starts from
    // them instead of from zero, so S^T comes out of the matrix pipe already masked (S + -inf = -inf) and the 2 vector
    // instructions per score element of a select are replaced by 2 per FOUR elements (table address) + one ds_read_b128
    // (carved out of `sm`: with a second __shared__ object hipcc's waitcnt pass drains vmcnt(0) before every LDS read of the loop)
    float* seed_lut = reinterpret_cast<float*>(sm + 4 * 2 * 8192 + 4 * 512);
    unsigned char* smw = sm + 4 * 2 * 8192;
    const int tid = threadIdx.x;
    if (tid < 64) seed_lut[tid] = ((tid >> 2) >> (tid & 3)) & 1 ? 0.f : NEG_INF;
    __syncthreads();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler: LDS-DMA destinations stay in SGPRs
    const int r = lane & 31, hh = lane >> 5;
    const float sl2 = p.scale * 1.4426950408889634f;
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_attn_zero_line);
    unsigned char* ring = sm + wave * 16384;
    unsigned char* mwring = smw + wave * 512;   // [stage][64 words]
    constexpr int DONE = 0x7fffffff;
    const bool has_reg = p.kreg != nullptr && p.nreg > 0;
    const int ngroups = (p.Lq + 63) >> 6;
    const int nbh = p.B * p.H;
    const int k_ls32 = (int)p.k_ls, v_ls32 = (int)p.v_ls;
    const PatchGeom pgeom = patch_geom(p.perm_hw, p.perm_w);
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    xcc = (use_xcd_queues == 2) ? (int)(blockIdx.x & 7) : (xcc & 7);   // 2 (A/B aid): the round-robin dispatch label instead

  for (int qstep = 0; qstep < (use_xcd_queues ? 8 : 1); ++qstep) {
    const int xq = use_xcd_queues ? (xcc + qstep) & 7 : 0;      // own queue first, then the others'
    // use_xcd_queues == 0 (A/B aid): one queue for the whole chip, slices dealt round-robin (item = rank * nbh + slice)
    const long qitems = use_xcd_queues ? ccv_sparse_queue_items(nbh, ngroups, xq) : (long)nbh * ngroups;
  for (;;) {
    unsigned int idx = 0;
    if (lane == 0) idx = atomicAdd(p.queue_counters ? p.queue_counters + xq : &g_sparse_ctr[slot][xq], 1u);
    const long item = (long)(unsigned int)__builtin_amdgcn_readfirstlane((int)idx);
    if (item >= qitems) break;
    int bh, rank;
    if (use_xcd_queues) ccv_sparse_queue_item(nbh, ngroups, xq, item, bh, rank);
    else { rank = (int)(item / nbh); bh = (int)(item % nbh); }
    if (rank >= ngroups) continue;
    const int head = bh % p.H, b = bh / p.H;
    const int qg = p.group_order ? p.group_order[(long)(b % p.mask_nb) * p.order_bs + rank] : rank;
    const long bo = b / p.inner, bi = b % p.inner;
    const int q0 = qg * 64;

    bf16x8 qf[2][4];
    int qi[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qi[qb] = min(q0 + 32 * qb + r, p.Lq - 1);
        const uint16_t* qp = p.q + bo * p.q_bso + bi * p.q_bsi + (long)ccv_patch_row(qi[qb], p.perm_hw, p.perm_w) * p.q_ls + head * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qb][s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    const int mb = b % p.mask_nb;
    const uint32_t* wrow = p.wave_bits + (long)mb * p.wave_bs + (long)qg * p.wave_words;
    const uint16_t* kmain = p.k + bo * p.k_bso + bi * p.k_bsi + head * 64;
    const uint16_t* vmain = p.v + bo * p.v_bso + bi * p.v_bsi + head * 64;

    // ---- block schedule: -1 = register tokens, then the set bits of this query group's bitmap row ----
    // The whole bitmap row (<= 64 words) is fetched once, one word per lane, and read back with v_readlane:
    // an ordinary vector load inside the loop would make hipcc drain vmcnt(0), i.e. every DMA in flight.
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
    // per-lane DMA source offsets (elements) of the 4 K and 4 V pieces of a 32-key block; a block is either 32
    // consecutive stored rows (raster) or one 4x8-pixel patch: row 8j + (lane>>3) of the block is stored row
    // base + j*W + (lane>>3), with `base` wave-uniform
    const int lr8 = lane >> 3, pc = lane & 7;
    long koff[4], voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rowin = 8 * j + lr8;
        const long step = p.perm_w ? (long)j * p.perm_w + lr8 : (long)rowin;
        koff[j] = step * p.k_ls + ((pc ^ ((rowin >> 1) & 7)) << 3);
        voff[j] = step * p.v_ls + ((pc ^ (((rowin >> 1) & 1) << 2)) << 3);
    }
    // buffer descriptors over this item's K / V slice and mask rows (all uniform) + the lane's 32-bit byte offsets into them
    const __amdgpu_buffer_rsrc_t rsrc_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(kmain), 0, ((p.Lk - 1) * k_ls32 + 64) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(vmain), 0, ((p.Lk - 1) * v_ls32 + 64) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.mask_bits + (long)mb * p.mask_bs), 0, p.Lq * p.mask_words * 4, 0x00020000);
    int koff32[4], voff32[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        koff32[j] = (int)koff[j] * 2;
        voff32[j] = (int)voff[j] * 2;
    }
    const int moff32 = min(q0 + lane, p.Lq - 1) * p.mask_words * 4;
    auto issue = [&](int blk, int stage) __attribute__((always_inline)) {   // 8 K/V DMA pieces + 1 mask-word DMA = 9 vector-memory operations
        unsigned char* sK = ring + stage * 8192;
        unsigned char* sV = sK + 4096;
        if (blk < 0) {   // register tokens: rows >= nreg read the zero line
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rowin = 8 * j + lr8;
                const bool ok = rowin < p.nreg;
                const uint16_t* gk = ok ? p.kreg + (long)rowin * p.H * 64 + head * 64 + ((pc ^ ((rowin >> 1) & 7)) << 3) : zero;
                const uint16_t* gv = ok ? p.vreg + (long)rowin * p.H * 64 + head * 64 + ((pc ^ (((rowin >> 1) & 1) << 2)) << 3) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t*)gk, (lptr_t*)(sK + j * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t*)gv, (lptr_t*)(sV + j * 1024), 16, 0, 0);
            }
        } else {
            const int k0 = 32 * blk;
            int base = k0;
            if (p.perm_w) base = patch_first_row(k0, pgeom);   // first stored row of the patch (wave-uniform scalar arithmetic)
            // through the slice's buffer descriptors: the lane's 32-bit offsets are fixed for the item, the block's first stored row rides
            // in the scalar offset (no vector address arithmetic per block); rows past the last key of a ragged last block (raster order
            // only: a patch-ordered sequence is a whole number of 32-key patches, checked on the host) fall outside the descriptor's
            // range and arrive as zeros
            const int ks = base * k_ls32 * 2, vs = base * v_ls32 * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_k, (lptr_t*)(sK + j * 1024), 16, koff32[j], ks, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, (lptr_t*)(sV + j * 1024), 16, voff32[j], vs, 0, 0);
            }
        }
        // mask words of the 64 queries (lane l <-> query q0 + l) go to LDS by DMA as well: 4 bytes per lane
        const int wi = blk < 0 ? 0 : blk;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_m, (lptr_t*)(mwring + stage * 256), 4, moff32, wi * 4, 0, 0);
    };

    float m_run[2] = {NEG_INF, NEG_INF}, l_run[2] = {0.f, 0.f};
    f32x16 oacc[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[qb][d][i] = 0.f;

    auto compute = [&](int blk, int stage) __attribute__((always_inline)) {
        const unsigned char* sK = ring + stage * 8192;
        const unsigned char* sV = sK + 4096;
        const int left = blk < 0 ? p.nreg : min(32, p.Lk - 32 * blk);
        const uint32_t lim = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
        const uint32_t* wl = reinterpret_cast<const uint32_t*>(mwring + stage * 256);
        uint32_t mw[2];
        mw[0] = (blk < 0 ? 0xffffffffu : wl[r]) & lim;
        mw[1] = (blk < 0 ? 0xffffffffu : wl[32 + r]) & lim;
        const bool on0 = __ballot(mw[0] != 0u) != 0ull;
        const bool on1 = (q0 + 32 < p.Lq) && (__ballot(mw[1] != 0u) != 0ull);
        if (!(on0 || on1)) return;
        // both 32-query halves always (the matrix pipe has slack, vector issue does not: computing only the half that sees a key cost
        // eight accumulator moves per k-step and a branch per MFMA in the one-sided blocks, a third of all visited blocks)
        f32x16 sa0, sa1;
        {   // accumulator seeds: register i of a lane is key (i & 3) + 8 (i >> 2) + 4 hh, i.e. nibble 2 (i >> 2) + hh of the mask word
            const uint32_t lut = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)(seed_lut);
            uint32_t ad[8];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {   // two instructions per address, kept opaque: hipcc's own form of the expression takes three
                uint32_t n0, n1;
                asm("v_bfe_u32 %0, %1, %2, 4" : "=v"(n0) : "v"(mw[0]), "v"(8 * g4 + 4 * hh));
                asm("v_bfe_u32 %0, %1, %2, 4" : "=v"(n1) : "v"(mw[1]), "v"(8 * g4 + 4 * hh));
                asm("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(ad[g4]) : "v"(n0), "v"(lut));
                asm("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(ad[4 + g4]) : "v"(n1), "v"(lut));
            }
            // the four K fragments of the lane travel in the same batch: one LDS round trip in front of the eight score MFMAs
            const uint32_t kb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(sK) + r * 128;
            uint32_t ka[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) ka[s] = kb + (((2 * s + hh) ^ ((r >> 1) & 7)) << 4);
            f32x4 t0, t1, t2, t3, t4, t5, t6, t7;
            bf16x8 kf0, kf1, kf2, kf3;
            asm volatile(   // inline: a C++ LDS read here makes hipcc drain vmcnt(0), i.e. the DMA ring
                "ds_read_b128 %8, %12\n\t"
                "ds_read_b128 %0, %16\n\t"
                "ds_read_b128 %1, %17\n\t"
                "ds_read_b128 %2, %18\n\t"
                "ds_read_b128 %3, %19\n\t"
                "ds_read_b128 %4, %20\n\t"
                "ds_read_b128 %5, %21\n\t"
                "ds_read_b128 %6, %22\n\t"
                "ds_read_b128 %7, %23\n\t"
                "ds_read_b128 %9, %13\n\t"
                "ds_read_b128 %10, %14\n\t"
                "ds_read_b128 %11, %15\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7),
                  "=&v"(kf0), "=&v"(kf1), "=&v"(kf2), "=&v"(kf3)
                : "v"(ka[0]), "v"(ka[1]), "v"(ka[2]), "v"(ka[3]),
                  "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
                : "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sa0[j] = t0[j]; sa0[4 + j] = t1[j]; sa0[8 + j] = t2[j]; sa0[12 + j] = t3[j];
                sa1[j] = t4[j]; sa1[4 + j] = t5[j]; sa1[8 + j] = t6[j]; sa1[12 + j] = t7[j];
            }
            // one chain after the other (a dependent MFMA issues back to back): the first half's scores are complete while the
            // second chain still runs, so its softmax starts 4 MFMAs earlier
            const bf16x8 kfr[4] = {kf0, kf1, kf2, kf3};
#pragma unroll
            for (int s = 0; s < 4; ++s) sa0 = ccv_mfma_32x32x16(kfr[s], qf[0][s], sa0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) sa1 = ccv_mfma_32x32x16(kfr[s], qf[1][s], sa1);
        }
        bf16x8 vf[2][2];
        read_vt_block(sV, lane, hh, vf);   // while the score MFMAs run
        __builtin_amdgcn_sched_barrier(0);
        // the scores arrive masked (seeds above): the softmax runs its all-visible form.  Each half's O^T MFMAs follow its
        // softmax directly and run under the other half's softmax.
        if (on0) {
            bf16x8 pf[2];
            softmax_block32(sa0, 0xffffffffu, true, hh, sl2, m_run[0], l_run[0], oacc[0], pf);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < 2; ++d) oacc[0][d] = ccv_mfma_32x32x16(vf[s2][d], pf[s2], oacc[0][d]);
        }
        if (on1) {
            bf16x8 pf[2];
            softmax_block32(sa1, 0xffffffffu, true, hh, sl2, m_run[1], l_run[1], oacc[1], pf);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < 2; ++d) oacc[1][d] = ccv_mfma_32x32x16(vf[s2][d], pf[s2], oacc[1][d]);
        }
    };

    // Retire every ordinary vector load (Q fragments, bitmap row) where hipcc can see it: its waitcnt bookkeeping
    // cannot see the counted waits below, and a load it still believes pending inside the loop becomes an
    // s_waitcnt vmcnt(0) in front of every use, i.e. it would drain the DMA ring on every MFMA.
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) asm volatile("" ::"v"(qf[qb][s4]));
    asm volatile("" ::"v"(my_word));

    // ---- two-deep software pipeline: the DMA of the next needed block flies while the current one is multiplied ----
    // the loop body exists once per ring stage, so that every LDS address of a stage is lane base + immediate offset
    int cur = next_block();
    auto step = [&](int stage) __attribute__((always_inline)) -> bool {
        const int nxt = next_block();
        if (nxt != DONE) {
            issue(nxt, stage ^ 1);
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // the 9 operations of `nxt` may stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        compute(cur, stage);
        cur = nxt;
        return cur != DONE;
    };
    if (cur != DONE) {
        issue(cur, 0);
        for (;;) {
            if (!step(0)) break;
            if (!step(1)) break;
        }
    }

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float wgt = l_tot > 0.f ? 1.0f / l_tot : 0.f;
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
  }  // next query group
  }  // next queue
#endif
}

// =================================================================================================
// attn_shared_kernel<NW, S> (round 4): the same masked attention with the K / V blocks SHARED by a workgroup.
// attn_sparse_kernel's waves each stream their own copy of every 32-key block they need: 8 KiB of DMA per 16 MFMAs, 12x the
// algorithmic bytes per launch, two thirds of it missing the XCD's L2.  Patch-ordered neighbours need almost the same
// blocks (benchmark masks, 32x32 latents: a 64-query group needs 173 of the 512 blocks, the 256 queries of four
// neighbouring groups 254 together), so here a workgroup of NW waves owns NW * 32 consecutive queries -- ONE 4x8-pixel
// patch per wave -- and walks the union of its groups' bitmaps (the OR of their wave_bits rows).  Every block of that
// union is staged ONCE per workgroup into an S-deep ring: each wave issues 8 / NW of its eight 1 KiB pieces (+ the mask words
// of its own 32 queries), so a wave issues 2-3 vector-memory operations per block instead of 9.  One s_barrier per block:
//     wait for my own pieces of block i (counted vmcnt: blocks i+1 .. i+S-2 stay in flight) -> s_barrier (everybody's pieces
//     of block i have landed AND everybody has finished block i-1) -> issue block i+S-1 into the stage of block i-1
//     -> multiply block i if one of my 32 queries sees a key of it (ballot over the mask words), else go on.
// A wave holds 32 queries instead of 64: half the accumulators (<= 128 registers per lane: four waves per SIMD instead of
// two), and a half without a visible key costs nothing instead of a full softmax (attn_sparse_kernel multiplies both halves
// of every block it visits: 346 half-blocks per group against the 290 needed).  Math per half-block is attn_sparse_kernel's:
// seeded S^T = K Q^T, in-lane softmax, O^T += V^T P^T.  Items (batch-head slice, query group) come from the same persistent
// queue, longest first by p.wg_order (ccv_attn_group_order_merged).
//
// Key-split items (the tail): the 1280 items of a 32x32-latent launch do not divide over the 1024 resident workgroups -- the queue's
// makespan is 1.39x the mean load (the last quarter of the items starts when the first workgroups finish), and the 640 items of a
// 16x16-latent launch leave 384 workgroup slots empty.  The items of rank >= split_rank0 (the shortest ones, handed out last) therefore
// come in split_parts parts, each walking a contiguous share of the item's schedule.  A part leaves its running (max, sum, O^T
// accumulators) in the workspace (write-through stores) and takes a ticket from the item's counter; the part that draws the
// LAST ticket reads the others back (L1-bypassing loads) and merges all parts in part order -- nobody ever waits for anybody, so the
// persistent grid cannot deadlock, and the merge (max, two products, one sum per value, no contraction) does not depend on who came last:
// the result is bitwise reproducible.  The counter is reset by the merging part (the workspace is zeroed once by the caller).
// =================================================================================================
constexpr int SPLIT_LANE_FLOATS = 36;          // per lane and part: 32 accumulators + running sum + running max + 2 pad (nine 16-byte stores)

constexpr int SHARED_MAX_BLOCKS = 1037;  // schedule entries (16 bits each) per item kept in LDS: Lk <= 32 * 1037 (+ the register-token step)

template <int NW, int S>
__global__ __launch_bounds__(NW * 64, 4) void attn_shared_kernel(const SparseArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(S >= 3 && S <= 5, "ring depth");
    constexpr int RING = S * 8192, MWB = NW * S * 256;
    constexpr int PP = 8 / NW;          // K / V pieces per wave and block
    constexpr int P = PP + 1;           // + the wave's mask words: vector-memory operations per wave and block
    constexpr int GQ = NW * 32;         // queries per item
    constexpr int SCHED = RING + MWB + 256 + 16;                    // the item's schedule: 16 bits per step
    __shared__ __attribute__((aligned(256))) unsigned char sm[SCHED + 2 * (SHARED_MAX_BLOCKS + 3)];   // ring | mask words [wave][stage][64] | seed table | item, n | schedule
    static_assert((RING + MWB) % 256 == 0, "seed table offset must keep the low 8 address bits clear");
    float* seed_lut = reinterpret_cast<float*>(sm + RING + MWB);
    unsigned int* item_word = reinterpret_cast<unsigned int*>(sm + RING + MWB + 256);   // [0] item, [1] steps of the item
    uint16_t* sched = reinterpret_cast<uint16_t*>(sm + SCHED);   // (plain LDS accesses: a volatile generic pointer becomes flat_load + vmcnt(0))
    const int tid = threadIdx.x;
    if (tid < 64) seed_lut[tid] = ((tid >> 2) >> (tid & 3)) & 1 ? 0.f : NEG_INF;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int hh4 = 4 * hh;
    const float sl2 = p.scale * 1.4426950408889634f;
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_attn_zero_line);
    unsigned char* mwring = sm + RING + wave * (S * 256);
    const bool has_reg = p.kreg != nullptr && p.nreg > 0;
    const int ngroups = (p.Lq + GQ - 1) / GQ;
    const int ngroups64 = (p.Lq + 63) >> 6;
    const int nbh = p.B * p.H;
    const PatchGeom pgeom = patch_geom(p.perm_hw, p.perm_w);
    const int parts = max(p.split_parts, 1);
    const int rank0 = parts > 1 ? min(p.split_rank0, ngroups) : ngroups;        // ranks >= rank0 come in `parts` parts
    const long q_whole = (long)nbh * rank0;
    const long qitems = q_whole + (long)nbh * (ngroups - rank0) * parts;
    const int lr8 = lane >> 3, pc = lane & 7;
    // this wave's pieces of a block (piece id = wave * PP + t; ids 0-3 are the four 8-row pieces of K, 4-7 those of V) are all K or all V
    const bool wave_k = wave * PP < 4;
    const int ls32 = wave_k ? (int)p.k_ls : (int)p.v_ls;      // token stride of the operand this wave stages
    // per-lane LDS addresses of stage 0 (see compute): seed table, the four K fragments, the two V^T read bases (read_vt_block's layout)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(sm);
    const uint32_t lut_addr = lds0 + RING + MWB;
    const uint32_t mw_addr = lds0 + RING + wave * (S * 256) + 4 * r;
    uint32_t ka[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ka[s] = lds0 + r * 128 + (((2 * s + hh) ^ ((r >> 1) & 7)) << 4);
    uint32_t va0, va1;
    {
        const int li = lane & 15, g = (lane >> 4) & 1;
        const int row0 = 4 * hh + (li >> 2);
        const int x = ((row0 >> 1) & 1) << 6;
        const int c = (16 * g + 4 * (li & 3)) * 2;
        va0 = lds0 + row0 * 128 + c + x;
        va1 = lds0 + row0 * 128 + c + 64 - x;
    }
    int poff32[PP];     // per-lane byte offsets of this wave's pieces inside a block (fixed for the whole launch)
#pragma unroll
    for (int t = 0; t < PP; ++t) {
        const int j = (wave * PP + t) & 3;
        const int rowin = 8 * j + lr8;
        const long step = p.perm_w ? (long)j * p.perm_w + lr8 : (long)rowin;
        const int swz = wave_k ? ((pc ^ ((rowin >> 1) & 7)) << 3) : ((pc ^ (((rowin >> 1) & 1) << 2)) << 3);
        poff32[t] = (int)(step * ls32 + swz) * 2;
    }

  for (;;) {
    __syncthreads();   // everybody is done with the previous item: the ring, the mask words, the schedule and the item word are free
    if (tid == 0) item_word[0] = atomicAdd(p.queue_counters ? p.queue_counters : &g_sparse_ctr[p.slot][0], 1u);
    __syncthreads();
    const long item = (long)(unsigned int)__builtin_amdgcn_readfirstlane((int)item_word[0]);
    if (item >= qitems) break;
    int rank, bh, part = 0, nparts = 1;
    if (item < q_whole) { rank = (int)(item / nbh); bh = (int)(item % nbh); }
    else {   // the parts of one (slice, rank) are nbh queue positions apart: they run at the same time on different workgroups
        const long t = item - q_whole;
        const int per_rank = nbh * parts;
        rank = rank0 + (int)(t / per_rank);
        const int rem = (int)(t % per_rank);
        part = rem / nbh;
        bh = rem % nbh;
        nparts = parts;
    }
    const int head = bh % p.H, b = bh / p.H;
    const int mb = b % p.mask_nb;
    const int qg = p.wg_order ? p.wg_order[(long)mb * p.wg_order_bs + rank] : rank;
    const long bo = b / p.inner, bi = b % p.inner;
    const int q0w = qg * GQ + 32 * wave;          // this wave's 32 queries
    const bool wave_active = q0w < p.Lq;

    // ---- the item's schedule, built once by wave 0 into LDS: step -> block + 1 (0 = register tokens), the set bits of the OR of the
    // item's bitmap rows in ascending order.  Lane l expands word l.
    if (wave == 0) {
        uint32_t u = 0u;
        const uint32_t* wrow = p.wave_bits + (long)mb * p.wave_bs;
#pragma unroll
        for (int j = 0; j < NW / 2; ++j) {
            const int g64 = qg * (NW / 2) + j;
            if (g64 < ngroups64 && lane < p.wave_words) u |= wrow[(long)g64 * p.wave_words + lane];
        }
        const int cnt = __popc(u);
        int pre = cnt;                                // inclusive prefix sum over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(pre, o, 64);
            if (lane >= o) pre += t;
        }
        const int total = __shfl(pre, 63, 64);
        int at = pre - cnt + (has_reg ? 1 : 0);
        if (lane == 0) {
            if (has_reg) sched[0] = (uint16_t)0;
            item_word[1] = (unsigned int)(total + (has_reg ? 1 : 0));
        }
        while (u) {
            const int bit = __builtin_ctz(u);
            u &= u - 1;
            sched[at++] = (uint16_t)(lane * 32 + bit + 1);
        }
    }

    bf16x8 qf[4];
    {
        const int qi = min(q0w + r, p.Lq - 1);
        const uint16_t* qp = p.q + bo * p.q_bso + bi * p.q_bsi + (long)ccv_patch_row(qi, p.perm_hw, p.perm_w) * p.q_ls + head * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    // buffer descriptors over the operand slice this wave stages (K or V of this batch and head) and over the mask rows
    const uint16_t* opnd = (wave_k ? p.k + bo * p.k_bso + bi * p.k_bsi : p.v + bo * p.v_bso + bi * p.v_bsi) + head * 64;
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(opnd), 0, ((p.Lk - 1) * ls32 + 64) * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.mask_bits + (long)mb * p.mask_bs), 0, p.Lq * p.mask_words * 4, 0x00020000);
    const int moff32 = min(q0w + r, p.Lq - 1) * p.mask_words * 4;   // both half-waves fetch the 32 words (lane l <-> query q0w + (l & 31))
    const uint16_t* regsrc = nullptr;
    if (has_reg) regsrc = (wave_k ? p.kreg : p.vreg) + head * 64;

    // e = schedule entry; stage = ring slot
    auto issue = [&](uint32_t e, int stage) __attribute__((always_inline)) {
        unsigned char* dst = sm + stage * 8192 + wave * (PP * 1024);
        const int blk = (int)(e & 0xffffu) - 1;
        if (blk < 0) {   // register tokens: rows >= nreg read the zero line
#pragma unroll
            for (int t = 0; t < PP; ++t) {
                const int j = (wave * PP + t) & 3;
                const int rowin = 8 * j + lr8;
                const int swz = wave_k ? ((pc ^ ((rowin >> 1) & 7)) << 3) : ((pc ^ (((rowin >> 1) & 1) << 2)) << 3);
                const uint16_t* g = rowin < p.nreg ? regsrc + (long)rowin * p.H * 64 + swz : zero;
                __builtin_amdgcn_global_load_lds((gptr_t*)g, (lptr_t*)(dst + t * 1024), 16, 0, 0);
            }
        } else {
            const int k0 = 32 * blk;
            int base = k0;
            if (p.perm_w) base = patch_first_row(k0, pgeom);   // first stored row of the 4x8-pixel patch (wave-uniform scalar arithmetic)
            const int so = base * ls32 * 2;
#pragma unroll
            for (int t = 0; t < PP; ++t) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_o, (lptr_t*)(dst + t * 1024), 16, poff32[t], so, 0, 0);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_m, (lptr_t*)(mwring + stage * 256), 4, moff32, max(blk, 0) * 4, 0, 0);
    };

    float m_run = NEG_INF, l_run = 0.f;
    f32x16 oacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;

    // every LDS address of the loop is a per-lane constant of stage 0 + an immediate offset (stage * 8192 ...): one set of address
    // registers for all S stages.  mw = the wave's mask words of the block (already waited for).
    auto compute = [&](int blk, uint32_t mw_raw, auto stc) __attribute__((always_inline)) {
        constexpr int stage = decltype(stc)::value;
        constexpr int KO = stage * 8192, VO = stage * 8192 + 4096;
        const int left = blk < 0 ? p.nreg : min(32, p.Lk - 32 * blk);
        const uint32_t lim = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
        const uint32_t mw = (blk < 0 ? 0xffffffffu : mw_raw) & lim;
        if (!wave_active || __ballot(mw != 0u) == 0ull) return;
        f32x16 sa;
        {
            uint32_t ad[4];
            const uint32_t mwsh = mw >> hh4;            // the half-wave's nibbles at bits 0, 8, 16, 24 (constant bit offsets: no offset registers)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint32_t n0;
                asm("v_bfe_u32 %0, %1, %2, 4" : "=v"(n0) : "v"(mwsh), "n"(8 * g4));
                asm("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(ad[g4]) : "v"(n0), "v"(lut_addr));
            }
            f32x4 t0, t1, t2, t3;
            bf16x8 kf0, kf1, kf2, kf3;
            asm volatile(
                "ds_read_b128 %4, %8 offset:%16\n\t"
                "ds_read_b128 %0, %12\n\t"
                "ds_read_b128 %1, %13\n\t"
                "ds_read_b128 %2, %14\n\t"
                "ds_read_b128 %3, %15\n\t"
                "ds_read_b128 %5, %9 offset:%16\n\t"
                "ds_read_b128 %6, %10 offset:%16\n\t"
                "ds_read_b128 %7, %11 offset:%16\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(kf0), "=&v"(kf1), "=&v"(kf2), "=&v"(kf3)
                : "v"(ka[0]), "v"(ka[1]), "v"(ka[2]), "v"(ka[3]), "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "n"(KO)
                : "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) { sa[j] = t0[j]; sa[4 + j] = t1[j]; sa[8 + j] = t2[j]; sa[12 + j] = t3[j]; }
            const bf16x8 kfr[4] = {kf0, kf1, kf2, kf3};
#pragma unroll
            for (int s = 0; s < 4; ++s) sa = ccv_mfma_32x32x16(kfr[s], qf[s], sa);
        }
        bf16x8 pf[2];
        softmax_block32(sa, 0xffffffffu, true, hh, sl2, m_run, l_run, oacc, pf);
        __builtin_amdgcn_sched_barrier(0);
        // V^T fragments only now: 16 registers fewer across the softmax keep the wave within 128 (four waves per SIMD, whose MFMAs and
        // vector work cover this read's latency)
        bf16x8 vf[2][2];
        {
            bf16x4 u0, u1, u2, u3, u4, u5, u6, u7;
            asm volatile(
                "ds_read_b64_tr_b16 %0, %8 offset:%10\n\t"
                "ds_read_b64_tr_b16 %1, %8 offset:%11\n\t"
                "ds_read_b64_tr_b16 %2, %9 offset:%10\n\t"
                "ds_read_b64_tr_b16 %3, %9 offset:%11\n\t"
                "ds_read_b64_tr_b16 %4, %8 offset:%12\n\t"
                "ds_read_b64_tr_b16 %5, %8 offset:%13\n\t"
                "ds_read_b64_tr_b16 %6, %9 offset:%12\n\t"
                "ds_read_b64_tr_b16 %7, %9 offset:%13\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7)
                : "v"(va0), "v"(va1), "n"(VO), "n"(VO + 1024), "n"(VO + 2048), "n"(VO + 3072)
                : "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vf[0][0][j] = u0[j]; vf[0][0][4 + j] = u1[j];
                vf[0][1][j] = u2[j]; vf[0][1][4 + j] = u3[j];
                vf[1][0][j] = u4[j]; vf[1][0][4 + j] = u5[j];
                vf[1][1][j] = u6[j]; vf[1][1][4 + j] = u7[j];
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int d = 0; d < 2; ++d) oacc[d] = ccv_mfma_32x32x16(vf[s2][d], pf[s2], oacc[d]);
    };

    // retire the ordinary vector loads where hipcc can see it (see attn_sparse_kernel)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) asm volatile("" ::"v"(qf[s4]));

    __syncthreads();                                   // the schedule is complete
    const int n_all = (int)__builtin_amdgcn_readfirstlane((int)item_word[1]);
    // this part's share of the schedule: steps [lo, n) (the prologue / loop below index the schedule from `lo`)
    const int lo = (int)((long)n_all * part / nparts);
    const int n = (int)((long)n_all * (part + 1) / nparts);

    // ---- S-deep ring: steps i .. i+S-2 are in flight or landed while step i is multiplied ----
    int cur_blk[S - 1];                                // block ids of steps i .. i+S-2 (scalars)
#pragma unroll
    for (int j = 0; j < S - 1; ++j) {
        cur_blk[j] = -2;
        if (lo + j < n) {
            const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)sched[lo + j]);
            cur_blk[j] = (int)(e & 0xffffu) - 1;
            issue(e, j);
        }
    }
    const uint32_t sched_addr = lds0 + SCHED;
    uint32_t e_next;                                    // entry of the step the next iteration issues (prefetched one step ahead)
    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e_next) : "v"(sched_addr + 2 * max(min(lo + S - 1, n - 1), 0)) : "memory");   // (hipcc does not
                                                        // track an inline-asm read: without the wait the first step could use the entry before it arrives)
    int i = lo;
    auto step = [&](auto stc) __attribute__((always_inline)) -> bool {
        constexpr int st = decltype(stc)::value;
        if (i >= n) return false;
        // my own pieces of step i have landed when at most the operations of the steps behind it are outstanding
        const int behind = n - 1 - i;                 // steps after this one
        if (behind >= S - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * P) : "memory");
        else if (S > 3 && behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * P) : "memory");
        else if (S > 4 && behind == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // the mask words of this step first: their LDS latency runs under the issue of step i + S - 1
        uint32_t mw_raw;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(mw_raw) : "v"(mw_addr), "n"(st * 256) : "memory");
        int nblk = -2;
        if (i + S - 1 < n) {
            const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)e_next);
            nblk = (int)(e & 0xffffu) - 1;
            issue(e, (st + S - 1) % S);
            asm volatile("ds_read_u16 %0, %1" : "=v"(e_next) : "v"(sched_addr + 2 * min(i + S, n - 1)) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mw_raw), "+v"(e_next)::"memory");
        compute(cur_blk[0], mw_raw, stc);
#pragma unroll
        for (int j = 0; j < S - 2; ++j) cur_blk[j] = cur_blk[j + 1];
        cur_blk[S - 2] = nblk;
        ++i;
        return true;
    };
    for (;;) {
        bool go = true;
        if (go) go = step(std::integral_constant<int, 0>{});
        if (go) go = step(std::integral_constant<int, 1>{});
        if (go) go = step(std::integral_constant<int, 2>{});
        if constexpr (S > 3) { if (go) go = step(std::integral_constant<int, 3>{}); }
        if constexpr (S > 4) { if (go) go = step(std::integral_constant<int, 4>{}); }
        if (!go) break;
    }

    if (nparts > 1) {
        // ---- a part of a key-split item: publish (max, sum, accumulators), take a ticket; the last ticket merges all parts ----
        const long sidx = (long)(rank - rank0) * nbh + bh;
        const long lane_f4 = SPLIT_LANE_FLOATS / 4;                               // 16-byte chunks per lane
        auto area = [&](int q) { return reinterpret_cast<float4*>(p.split_part) + (((sidx * nparts + q) * NW + wave) * lane_f4) * 64 + lane; };
        // The hand-off uses write-through stores and L1-bypassing loads (sc1) instead of a release / acquire fence pair: a release fence
        // writes back the XCD's whole L2 and an acquire drops the CU's L1 under the feet of the 15 other waves (measured: the fences ate
        // most of the split's gain).  MI355X_MICROARCH.md, "Valid forms": every handed-off byte stored sc1, every storing wave drained
        // (vmcnt(0)) before the workgroup barrier, ONE lane's agent-scope atomic add behind that barrier, the workgroup whose add came
        // last (told by the returned value) loads -- every load sc1 -- after a barrier that lane has joined.
        {
            float4* dst = area(part);                                           // chunk c of lane l at [c][l]: one 1 KiB store per chunk and wave
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 v4 = {oacc[c >> 2][4 * (c & 3)], oacc[c >> 2][4 * (c & 3) + 1], oacc[c >> 2][4 * (c & 3) + 2], oacc[c >> 2][4 * (c & 3) + 3]};
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + c * 64), "v"(v4) : "memory");
            }
            const f32x4 tail = {l_run, m_run, 0.f, 0.f};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + 8 * 64), "v"(tail) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // every storing wave, then the workgroup ...
        __syncthreads();
        if (tid == 0) item_word[2] = atomicAdd(p.split_ctr + sidx, 1u);          // ... then ONE lane signals (agent scope, returned value)
        __syncthreads();
        const unsigned ticket = (unsigned)__builtin_amdgcn_readfirstlane((int)item_word[2]);
        if (ticket != (unsigned)(nparts - 1)) continue;                          // somebody else will merge
        if (tid == 0) p.split_ctr[sidx] = 0u;                                    // for the next launch that uses this workspace
        // merge (own part from registers): O = sum over the parts IN PART ORDER of O_q 2^(m_q - max_q m_q), products and sums not contracted:
        // the same arithmetic whoever arrived last.  Two passes so that at most 16 + 16 values are live beside the accumulators.
        float mqs[4], lqs[4], M = NEG_INF;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mqs[q] = NEG_INF; lqs[q] = 0.f;
            if (q < nparts) {
                if (q == part) { mqs[q] = m_run; lqs[q] = l_run; }
                else {
                    f32x4 tl;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(tl) : "v"(area(q) + 8 * 64) : "memory");
                    lqs[q] = tl[0]; mqs[q] = tl[1];
                }
                M = fmaxf(M, mqs[q]);
            }
        }
        const float mu = (M == NEG_INF) ? 0.f : M;
        float wq[4], Ls = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wq[q] = __builtin_amdgcn_exp2f(mqs[q] - mu);
            if (q < nparts) Ls = __fadd_rn(Ls, __fmul_rn(lqs[q], wq[q]));
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 sum;
#pragma unroll
            for (int e = 0; e < 16; ++e) sum[e] = 0.f;
            for (int q = 0; q < nparts; ++q) {
                f32x4 t[4];
                if (q == part) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) { t[c][0] = oacc[half][4 * c]; t[c][1] = oacc[half][4 * c + 1]; t[c][2] = oacc[half][4 * c + 2]; t[c][3] = oacc[half][4 * c + 3]; }
                } else {
                    const float4* src = area(q) + (4 * half) * 64;
#pragma unroll
                    for (int c = 0; c < 4; ++c) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[c]) : "v"(src + c * 64) : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])::"memory");
                }
                const float w = q == 0 ? wq[0] : (q == 1 ? wq[1] : (q == 2 ? wq[2] : wq[3]));
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sum[4 * c + e] = __fadd_rn(sum[4 * c + e], __fmul_rn(t[c][e], w));
            }
            oacc[half] = sum;
        }
        l_run = Ls;
    }
    {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float wgt = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        const int q = q0w + r;
        if (q < p.Lq) {
            uint16_t* op = p.o + bo * p.o_bso + bi * p.o_bsi + (long)ccv_patch_row(q, p.perm_hw, p.perm_w) * p.o_ls + head * 64;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dd = 32 * d + 8 * g4 + 4 * hh;
                    uint2 pk = make_uint2(pack_bf16x2(oacc[d][4 * g4] * wgt, oacc[d][4 * g4 + 1] * wgt),
                                          pack_bf16x2(oacc[d][4 * g4 + 2] * wgt, oacc[d][4 * g4 + 3] * wgt));
                    *reinterpret_cast<uint2*>(op + dd) = pk;
                }
        }
    }
  }  // next item
#endif
}

// =================================================================================================
// attn_temporal_kernel: self attention over <= 16 tokens (the frames of one pixel), one wave per
// (pixel, head).  HBM/L2-bound: Q and K fragments are loaded straight from the token-major activations
// (16 rows x 64 B per instruction), S^T = K Q^T is two v_mfma_f32_16x16x32_bf16, the softmax runs on the
// 16x16 accumulator (4 keys per lane + two cross-lane exchanges), and because the S^T accumulator layout
// IS the B-operand layout of v_mfma_f32_16x16x16_bf16, P^T feeds O^T = V^T P^T without any lane movement;
// only V takes a 2 KiB wave-private LDS round trip to be read transposed (ds_read_b64_tr_b16).
// =================================================================================================
typedef __attribute__((ext_vector_type(4))) short short4_t;

__global__ __launch_bounds__(256) void attn_temporal_kernel(const CcvAttn p) {
    __shared__ __attribute__((aligned(16))) unsigned char sV[4][16 * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nitem = (long)p.B * p.H;
    const long item = (long)blockIdx.x * 4 + wave;
    const bool active = item < nitem;       // wave-uniform
    const long it = active ? item : 0;
    const int head = (int)(it % p.H);
    const long b = it / p.H;
    const long bo = b / p.inner, bi = b % p.inner;
    const int fr = lane & 15, g = lane >> 4;
    const int T = p.Lk;
    const int frow = min(fr, T - 1);

    const uint16_t* qp = p.q + bo * p.q_bso + bi * p.q_bsi + (long)frow * p.q_ls + head * 64 + 8 * g;
    const uint16_t* kp = p.k + bo * p.k_bso + bi * p.k_bsi + (long)frow * p.k_ls + head * 64 + 8 * g;
    bf16x8 qf[2], kf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        qf[s] = *reinterpret_cast<const bf16x8*>(qp + 32 * s);
        kf[s] = *reinterpret_cast<const bf16x8*>(kp + 32 * s);
    }
    // V rows -> wave-private LDS image [16 frames][128 B] (rows >= T zero)
    const uint16_t* vb = p.v + bo * p.v_bso + bi * p.v_bsi + head * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = lane + 64 * i, row = idx >> 3, c = idx & 7;
        uint4 vv = make_uint4(0u, 0u, 0u, 0u);
        if (row < T) vv = *reinterpret_cast<const uint4*>(vb + (long)row * p.v_ls + c * 8);
        *reinterpret_cast<uint4*>(&sV[wave][row * 128 + c * 16]) = vv;
    }
    // S^T[key 4g + r][query fr]
    f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) sacc = ccv_mfma_16x16x32(kf[s], qf[s], sacc);
    const float sl2 = p.scale * 1.4426950408889634f;
    float sv[4], m = NEG_INF;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sv[r] = (4 * g + r < T) ? sacc[r] * sl2 : NEG_INF;
        m = fmaxf(m, sv[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
    float ev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ev[r] = __builtin_amdgcn_exp2f(sv[r] - m);
        l += ev[r];
    }
    typedef __attribute__((ext_vector_type(2))) uint32_t ccv_u2_t;
    const ccv_u2_t pbu = {pack_bf16x2(ev[0], ev[1]), pack_bf16x2(ev[2], ev[3])};
    const bf16x4 pb = __builtin_bit_cast(bf16x4, pbu);
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    __syncthreads();  // V image complete (written by this wave's own lanes; block barrier keeps it simple)
    const int li = lane & 15;
    uint16_t* op = p.o + bo * p.o_bso + bi * p.o_bsi + (long)fr * p.o_ls + head * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const unsigned char* a0 = &sV[wave][(4 * g + (li >> 2)) * 128 + (16 * dt + 4 * (li & 3)) * 2];
        const bf16x4 vt = ccv_ds_read_tr16(a0);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        o = ccv_mfma_16x16x16(vt, pb, o);  // O^T[d = 16 dt + 4 g + r][query fr]
        if (active && fr < T) {
            uint2 pk = make_uint2(pack_bf16x2(o[0] * inv, o[1] * inv), pack_bf16x2(o[2] * inv, o[3] * inv));
            *reinterpret_cast<uint2*>(op + 16 * dt) = pk;
        }
    }
}

}  // namespace

// Host-side view of the sparse kernel's work queues (tests): item `idx` of XCD `xq`'s queue -> (slice, rank); returns the
// queue length (rank >= ngroups marks a padding item the kernel skips).
extern "C" int64_t ccv_attn_sparse_queue_item(int32_t nbh, int32_t ngroups, int32_t xq, int64_t idx, int32_t* bh, int32_t* rank) {
    const long n = ccv_sparse_queue_items(nbh, ngroups, xq);
    if (bh && rank && idx >= 0 && idx < n) {
        int b_, r_;
        ccv_sparse_queue_item(nbh, ngroups, xq, idx, b_, r_);
        *bh = b_;
        *rank = r_;
    }
    return n;
}

// ---- routing of the masked calls that carry a block bitmap (shared by ccv_attn_fwd and ccv_attn_split_ws_bytes) ----------------------
static int attn_n_cu() {
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return n_cu;
}
static long sparse_pct() {   // persistent workgroups in percent of the CU count (A/B aid; 200 = two 256-thread workgroups of the per-wave kernel per CU)
    static const long pct = [] { const char* e = getenv("CCV_ATTN_SPARSE_PCT"); const int v = e ? atoi(e) : 200; return (long)(v >= 50 && v <= 200 ? v : 200); }();
    return pct;
}
static bool takes_sparse_path(const CcvAttn& p) {
    return (p.variant == 0 || p.variant >= 3) && p.k2 == nullptr && p.mask_bits && p.wave_bits &&
           ((p.variant >= 3 && p.variant <= 6) || (long)((p.Lq + 63) / 64) * p.H * p.B >= 1024);
}
// waves per workgroup of the workgroup-shared kernel for this call, 0 = the per-wave kernel.  variant 4 / 5 force the 8- / 4-wave form,
// variant 6 the per-wave kernel; otherwise CCV_ATTN_SHARED = 4 (default: measured fastest, profiles/r04_sparse_shared_kv.txt) | 8 | 0
static int shared_waves(const CcvAttn& p) {
    static const int shared_env = [] { const char* e = getenv("CCV_ATTN_SHARED"); const int v = e ? atoi(e) : 4; return (v == 8 || v == 4) ? v : 0; }();
    int nw = p.variant == 4 ? 8 : (p.variant == 5 ? 4 : (p.variant == 6 ? 0 : shared_env));
    if ((p.Lk + 31) / 32 > SHARED_MAX_BLOCKS) nw = 0;   // the item's schedule lives in LDS (16 bits per step): longer key sequences stay on the per-wave kernel
    return nw;
}
struct SplitPlan { int rank0, parts; long n_split, ctr_bytes, total_bytes, slots, items_per; };
// Key-split plan of the workgroup-shared kernel (see the kernel): T items over P resident workgroups.  T > P (32x32 latents: 1280 over
// 1024): the ranks that do not fit the first round -- the shortest items -- come in two parts, so the workgroups that finish first share
// them.  T <= P (16x16 latents: 640 over 1024, 384 slots idle): every item in s parts, s in 1 .. 4 minimising ceil(T s / P) / s (3: two
// rounds of thirds instead of one round of wholes) -- built and measured slower, so only on request (split_all_parts).  CCV_ATTN_SPLIT=0
// switches the tail split off (A/B aid); short key sequences are never split.
static SplitPlan split_plan(const CcvAttn& p, int nw) {
    static const bool on = [] { const char* e = getenv("CCV_ATTN_SPLIT"); return !(e && e[0] == '0'); }();
    SplitPlan sp{0, 1, 0, 0, 0, 0, 0};
    const int merge = nw / 2;
    const long g64 = (p.Lq + 63) / 64;
    sp.items_per = (g64 + merge - 1) / merge;
    const long nbh = (long)p.H * p.B, T = sp.items_per * nbh;
    const long cap = sparse_pct() * attn_n_cu() / 100 * (nw == 8 ? 1 : 2);     // 2 x 512 or 4 x 256 threads per CU
    sp.slots = T < cap ? T : cap;
    sp.rank0 = (int)sp.items_per;
    if (!on || (p.Lk + 31) / 32 < 96 || T < 64) return sp;
    if (T > cap) {
        if (T >= 3 * cap) return sp;                     // many rounds: the tail is a small share
        sp.rank0 = (int)(cap / nbh);
        sp.parts = 2;
    } else {
        // measured (profiles/r04_sparse_shared_kv.txt): splitting EVERY item of a launch costs more partial-result traffic than the idle slots
        // are worth (16x16 latents, 3 parts: 108 -> 123 us), so it happens only on request (CcvAttn.split_all_parts; tests, A/B runs)
        static const int env_all = [] { const char* e = getenv("CCV_ATTN_SPLIT_ALL"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 4 ? v : 1; }();
        const int max_all = p.split_all_parts >= 2 && p.split_all_parts <= 4 ? p.split_all_parts : env_all;
        int best = 1;
        double cost = 1.0;
        for (int s2 = 2; s2 <= max_all; ++s2) {
            const double c = (double)((T * s2 + cap - 1) / cap) / s2;
            if (c < cost - 1e-9) { cost = c; best = s2; }
        }
        if (best == 1) return sp;
        sp.rank0 = 0;
        sp.parts = best;
        sp.slots = T * best < cap ? T * best : cap;
    }
    if (sp.rank0 >= sp.items_per) { sp.rank0 = (int)sp.items_per; sp.parts = 1; return sp; }
    sp.n_split = (sp.items_per - sp.rank0) * nbh;
    // the counters live in a FIXED prefix of the workspace: calls of different sizes share one buffer (one per stream), and a smaller
    // call's partial results must never land on a larger call's (zero, self-cleaning) counters
    sp.ctr_bytes = 65536;
    if (sp.n_split * 4 > sp.ctr_bytes) { sp.rank0 = (int)sp.items_per; sp.parts = 1; sp.n_split = 0; sp.ctr_bytes = 0; return sp; }
    sp.total_bytes = sp.ctr_bytes + sp.n_split * sp.parts * nw * 64 * SPLIT_LANE_FLOATS * 4;
    return sp;
}

extern "C" int64_t ccv_attn_split_ws_bytes(const CcvAttn* pp) {
    if (!pp || !takes_sparse_path(*pp)) return 0;
    const int nw = shared_waves(*pp);
    if (!nw) return 0;
    return split_plan(*pp, nw).total_bytes;
}

// CCV_ATTN2_Q32: bit mask of the tiled-attention forms that run with 32 queries per wave (attn2_kernel<.., .., 1>: four workgroups per CU)
// instead of 64: 1 = two-context cross attention, 2 = unmasked single context (the masked tiled form has no such instance).  Default 3:
// +1.0 % frames/s one clip at a time, within the noise with two in flight (profiles/r04_ab_switches.txt); CCV_ATTN2_Q32=0 is the A/B arm.
static int attn2_q32(const CcvAttn& p) {
    static const int v = [] { const char* e = getenv("CCV_ATTN2_Q32"); return e ? atoi(e) & 3 : 3; }();
    return p.variant == 7 ? 3 : (p.variant == 8 ? 0 : v);      // variants 7 / 8: as 0 with 32 / 64 queries per wave whatever the default (tests)
}

static bool two_ctx_on() {   // CCV_ATTN_TWO=0: two-context calls on the first-generation kernel (A/B aid)
    static const bool v = [] { const char* e = getenv("CCV_ATTN_TWO"); return !(e && e[0] == '0'); }();
    return v;
}

// Inner batches that share their keys and values (k/v inner-batch stride 0: the frames of a clip against the clip's text /
// image tokens) and whose queries and outputs lie back to back are one longer query sequence: a workgroup then fills its
// four waves with 256 queries from several frames and streams the shared K/V tiles once for all of them (at 8x8 latents a
// frame has 64 queries: one active wave per workgroup and 16x the K/V traffic otherwise).  CCV_ATTN_FOLD=0 disables it.
static CcvAttn fold_shared_kv(const CcvAttn& in) {
    static const bool on = [] { const char* e = getenv("CCV_ATTN_FOLD"); return !(e && e[0] == '0'); }();
    CcvAttn p = in;
    const bool shared = p.k_bsi == 0 && p.v_bsi == 0 && (!p.k2 || (p.k2_bsi == 0 && p.v2_bsi == 0));
    if (on && shared && p.inner > 1 && p.B % p.inner == 0 && !p.mask_bits && !p.kreg && p.perm_w == 0 && (p.variant == 0 || p.variant >= 7) &&
        p.q_bsi == (int64_t)p.Lq * p.q_ls && p.o_bsi == (int64_t)p.Lq * p.o_ls && (long)p.inner * p.Lq < (1l << 30) && p.Lq > 16) {
        p.Lq *= p.inner;
        p.B /= p.inner;
        p.inner = 1;
        p.q_bsi = p.o_bsi = 0;
    }
    return p;
}

extern "C" int ccv_attn_fwd(const CcvAttn* pp, void* stream) {
    CCV_REQUIRE(pp != nullptr, CCV_EINVAL, "ccv_attn_fwd: null params");
    const CcvAttn p_folded = fold_shared_kv(*pp);
    const CcvAttn& p = p_folded;
    CCV_REQUIRE(p.q && p.k && p.v && p.o, CCV_EINVAL, "ccv_attn_fwd: null q/k/v/o");
    CCV_REQUIRE(p.B > 0 && p.H > 0 && p.Lq > 0 && p.Lk > 0 && p.inner > 0, CCV_EINVAL,
                "ccv_attn_fwd: non-positive B/H/Lq/Lk/inner (%d,%d,%d,%d,%d)", p.B, p.H, p.Lq, p.Lk, p.inner);
    CCV_REQUIRE(p.H <= 65535, CCV_ESHAPE, "ccv_attn_fwd: too many heads");
    CCV_REQUIRE((p.q_ls % 8 == 0) && (p.k_ls % 8 == 0) && (p.v_ls % 8 == 0) && (p.o_ls % 4 == 0), CCV_ESHAPE,
                "ccv_attn_fwd: token strides must keep 16-byte alignment");
    CCV_REQUIRE(!p.k2 || (p.v2 && p.Lk2 > 0), CCV_EINVAL, "ccv_attn_fwd: second context needs k2, v2, Lk2");
    CCV_REQUIRE(!p.mask_bits || p.mask_words * 32 >= p.Lk, CCV_EINVAL, "ccv_attn_fwd: mask_words too small");
    CCV_REQUIRE(!p.mask_bits || p.mask_nb > 0, CCV_EINVAL, "ccv_attn_fwd: mask_nb must be positive");
    CCV_REQUIRE(!p.tile_flags || p.flags_ktiles * KT >= p.Lk, CCV_EINVAL, "ccv_attn_fwd: flags_ktiles too small");
    CCV_REQUIRE(!(p.variant >= 3 && p.variant <= 6) || (p.wave_bits && !p.k2), CCV_EINVAL, "ccv_attn_fwd: variants 3-6 need wave_bits and a single context");
    CCV_REQUIRE(p.variant >= 0 && p.variant <= 8, CCV_EINVAL, "ccv_attn_fwd: unknown variant %d", p.variant);
    CCV_REQUIRE(!p.wave_bits || (p.mask_bits && (long)p.wave_words * 32 * 32 >= p.Lk), CCV_EINVAL,
                "ccv_attn_fwd: wave_bits needs mask_bits and wave_words covering Lk");
    CCV_REQUIRE(!p.kreg || p.vreg, CCV_EINVAL, "ccv_attn_fwd: kreg without vreg");
    CCV_REQUIRE(p.nreg <= KT, CCV_ESHAPE, "ccv_attn_fwd: at most 64 register tokens");
    CCV_REQUIRE(p.perm_w == 0 || ((p.variant == 0 || p.variant >= 3) && !p.k2 && p.perm_w % 8 == 0 && p.perm_hw > 0 && p.perm_hw % (4 * p.perm_w) == 0 &&
                                  p.Lq % p.perm_hw == 0 && p.Lk % p.perm_hw == 0),
                CCV_ESHAPE, "ccv_attn_fwd: patch order needs the single-context kernel, W %% 8 == 0, H %% 4 == 0 and whole frames");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // gridDim.z is limited to 65535: fold large batches (temporal attention: one batch per pixel)
    const bool temporal_path = (p.variant == 0 || p.variant >= 3) && !p.k2 && !p.mask_bits && !p.kreg && p.Lq == p.Lk && p.Lk <= 16 && p.perm_w == 0;
    CCV_REQUIRE(temporal_path || p.B <= 65535, CCV_ESHAPE, "ccv_attn_fwd: B=%d exceeds 65535 (split the call)", p.B);
    dim3 grid((p.Lq + 127) / 128, p.H, p.B);
    if (temporal_path) {
        // frames-of-a-pixel attention: one wave per (batch, head), no key tiling
        const long items = (long)p.B * p.H;
        hipLaunchKernelGGL(attn_temporal_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, p);
    } else if ((p.variant == 0 || p.variant >= 3) && p.k2 == nullptr) {  // single-context attention: second-generation kernel, 256 queries per workgroup
        const long nwg2 = (long)((p.Lq + 255) / 256) * p.H * p.B;
        CCV_REQUIRE(nwg2 < (1l << 31), CCV_ESHAPE, "ccv_attn_fwd: grid too large");
        dim3 grid2((unsigned)nwg2);
        const long nwg2h = (long)((p.Lq + 127) / 128) * p.H * p.B;      // 32 queries per wave (attn2_kernel<.., .., 1>)
        CCV_REQUIRE(nwg2h < (1l << 31), CCV_ESHAPE, "ccv_attn_fwd: grid too large");
        dim3 grid2h((unsigned)nwg2h);
        if (takes_sparse_path(p)) {
            // persistent: 2 workgroups per CU (LDS-bound), fewer when there are fewer 64-query groups than waves
            static std::atomic<int> next_slot{0};
            CCV_REQUIRE(p.k_ls >= 0 && p.v_ls >= 0 && (long)p.Lk * p.k_ls < (1l << 31) && (long)p.Lk * p.v_ls < (1l << 31), CCV_ESHAPE,
                        "ccv_attn_fwd: sparse kernel addresses one K/V slice with 32-bit element offsets");
            const int n_cu = attn_n_cu();
            const int slot = next_slot.fetch_add(1) & 63;   // successive launches use different counter rows; callers whose launches
                                                            // overlap on several streams pass their own zeroed queue_counters
            const long groups = (long)((p.Lq + 63) / 64) * p.H * p.B;
            // persistent workgroups in percent of the CU count (A/B aid; 200 = two per CU = all the LDS)
            const long pct = sparse_pct();
            const long cap = pct * n_cu / 100;
            const long wgs = (groups + 3) / 4 < cap ? (groups + 3) / 4 : cap;
            if (!p.queue_counters) hipLaunchKernelGGL(sparse_ctr_reset, dim3(1), dim3(64), 0, st, slot);
            static const int xcd_queues = [] { const char* e = getenv("CCV_ATTN_XCD"); return e ? atoi(e) : 0; }();   // 0 (default): one chip-wide queue; 1: per-XCD queues
            // (HW_REG_XCC_ID); 2: per-XCD queues with blockIdx & 7 as the label.  Measured on MI355X (profiles/r02_sparse_xcd_queues.txt):
            // the XCD-local queues are SLOWER (32x32 latents 585 -> 621 us, 16x16 168 -> 208 us per b=2 launch), so they stay off
            const long mask_bs = p.mask_bs, wave_bs = p.wave_bs, order_bs = p.order_bs;
            CCV_REQUIRE(p.q_ls >= 0 && p.q_ls < (1l << 31) && p.o_ls >= 0 && p.o_ls < (1l << 31) && mask_bs >= 0 && mask_bs < (1l << 31) &&
                            wave_bs >= 0 && wave_bs < (1l << 31) && order_bs >= 0 && order_bs < (1l << 31), CCV_ESHAPE,
                        "ccv_attn_fwd: sparse kernel takes 32-bit token strides and mask slice sizes");
            // K / V blocks and mask words reach LDS through buffer descriptors over one (batch, head) slice / one mask: 32-bit byte offsets
            CCV_REQUIRE(p.k_ls >= 0 && p.v_ls >= 0 && ((long)(p.Lk - 1) * p.k_ls + 64) * 2 < 0x7ff00000l && ((long)(p.Lk - 1) * p.v_ls + 64) * 2 < 0x7ff00000l &&
                            (long)p.Lq * p.mask_words * 4 < 0x7ff00000l, CCV_ESHAPE, "ccv_attn_fwd: sparse kernel: a K / V slice or a mask spans 2 GiB or more");
            CCV_REQUIRE(p.perm_w == 0 || p.Lk % 32 == 0, CCV_ESHAPE, "ccv_attn_fwd: patch-ordered keys come in whole 32-key patches (Lk = %d)", p.Lk);
            SparseArgs a;
            a.q = p.q; a.k = p.k; a.v = p.v; a.o = p.o; a.kreg = p.kreg; a.vreg = p.vreg;
            a.mask_bits = p.mask_bits; a.wave_bits = p.wave_bits; a.group_order = p.group_order; a.queue_counters = p.queue_counters;
            a.q_bso = p.q_bso; a.q_bsi = p.q_bsi; a.k_bso = p.k_bso; a.k_bsi = p.k_bsi; a.v_bso = p.v_bso; a.v_bsi = p.v_bsi; a.o_bso = p.o_bso; a.o_bsi = p.o_bsi;
            a.q_ls = (int32_t)p.q_ls; a.k_ls = (int32_t)p.k_ls; a.v_ls = (int32_t)p.v_ls; a.o_ls = (int32_t)p.o_ls;
            a.mask_bs = (int32_t)mask_bs; a.mask_words = p.mask_words; a.mask_nb = p.mask_nb; a.wave_bs = (int32_t)wave_bs; a.wave_words = p.wave_words;
            a.order_bs = (int32_t)order_bs;
            a.B = p.B; a.inner = p.inner; a.H = p.H; a.Lq = p.Lq; a.Lk = p.Lk; a.nreg = p.kreg ? p.nreg : 0; a.perm_hw = p.perm_hw; a.perm_w = p.perm_w;
            a.scale = p.scale; a.slot = slot; a.use_xcd_queues = xcd_queues;
            a.wg_order = nullptr; a.wg_order_bs = 0;
            a.split_rank0 = 0; a.split_parts = 1; a.split_ctr = nullptr; a.split_part = nullptr;
            const int nw = shared_waves(p);      // round 4: K / V blocks shared by the workgroup (attn_shared_kernel)
            if (nw) {
                const int merge = nw / 2;                         // 64-query groups per item
                const long g64 = (p.Lq + 63) / 64, items_per = (g64 + merge - 1) / merge;
                if (p.wg_order && p.wg_merge == merge) {
                    CCV_REQUIRE(p.wg_order_bs >= items_per && p.wg_order_bs < (1l << 31), CCV_EINVAL, "ccv_attn_fwd: wg_order_bs too small");
                    a.wg_order = p.wg_order; a.wg_order_bs = (int32_t)p.wg_order_bs;
                }
                // key-split tail (split_plan): needs the caller's workspace (ccv_attn_split_ws_bytes; counters zeroed once, self-cleaning)
                const SplitPlan sp = split_plan(p, nw);
                if (sp.parts > 1 && p.split_ws != nullptr && p.split_ws_bytes >= sp.total_bytes) {
                    a.split_rank0 = sp.rank0; a.split_parts = sp.parts;
                    a.split_ctr = static_cast<uint32_t*>(p.split_ws);
                    a.split_part = reinterpret_cast<float*>(static_cast<unsigned char*>(p.split_ws) + sp.ctr_bytes);
                }
                const long items = items_per * p.H * p.B;
                const long cap_s = pct * n_cu / 100 * (nw == 8 ? 1 : 2);   // 2 x 512 or 4 x 256 threads per CU
                const long work = a.split_parts > 1 ? (long)sp.rank0 * p.H * p.B + sp.n_split * sp.parts : items;
                const long wgs_s = work < cap_s ? work : cap_s;
                static const int ring_s = [] { const char* e = getenv("CCV_ATTN_SHARED_S"); const int v = e ? atoi(e) : 4; return (v >= 3 && v <= 5) ? v : 4; }();   // ring depth (A/B aid)
                if (nw == 8) {
                    if (ring_s == 3)      hipLaunchKernelGGL((attn_shared_kernel<8, 3>), dim3((unsigned)wgs_s), dim3(512), 0, st, a);
                    else if (ring_s == 5) hipLaunchKernelGGL((attn_shared_kernel<8, 5>), dim3((unsigned)wgs_s), dim3(512), 0, st, a);
                    else                  hipLaunchKernelGGL((attn_shared_kernel<8, 4>), dim3((unsigned)wgs_s), dim3(512), 0, st, a);
                } else {
                    if (ring_s == 3)      hipLaunchKernelGGL((attn_shared_kernel<4, 3>), dim3((unsigned)wgs_s), dim3(256), 0, st, a);
                    else if (ring_s == 5) hipLaunchKernelGGL((attn_shared_kernel<4, 5>), dim3((unsigned)wgs_s), dim3(256), 0, st, a);
                    else                  hipLaunchKernelGGL((attn_shared_kernel<4, 4>), dim3((unsigned)wgs_s), dim3(256), 0, st, a);
                }
            } else
            hipLaunchKernelGGL(attn_sparse_kernel, dim3((unsigned)wgs), dim3(256), 0, st, a);
        }
        else if (p.mask_bits) {
            hipLaunchKernelGGL(attn2_kernel<true>, grid2, dim3(256), 0, st, p);
        } else {
            if (attn2_q32(p) & 2) hipLaunchKernelGGL((attn2_kernel<false, false, 1>), grid2h, dim3(256), 0, st, p);
            else                 hipLaunchKernelGGL(attn2_kernel<false>, grid2, dim3(256), 0, st, p);
        }
    } else if ((p.variant == 0 || p.variant >= 7) && p.k2 && !p.mask_bits && p.perm_w == 0 && two_ctx_on()) {   // two contexts (text + gated image tokens)
        const long nwg2 = (long)((p.Lq + 255) / 256) * p.H * p.B;
        CCV_REQUIRE(nwg2 < (1l << 31), CCV_ESHAPE, "ccv_attn_fwd: grid too large");
        if (attn2_q32(p) & 1) {
            const long nwg2h = (long)((p.Lq + 127) / 128) * p.H * p.B;
            CCV_REQUIRE(nwg2h < (1l << 31), CCV_ESHAPE, "ccv_attn_fwd: grid too large");
            hipLaunchKernelGGL((attn2_kernel<false, true, 1>), dim3((unsigned)nwg2h), dim3(256), 0, st, p);
        } else
        hipLaunchKernelGGL((attn2_kernel<false, true>), dim3((unsigned)nwg2), dim3(256), 0, st, p);
    } else if (p.variant == 0 || p.variant == 1)
        hipLaunchKernelGGL(attn_kernel<true>, grid, dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL(attn_kernel<false>, grid, dim3(256), 0, st, p);
    CCV_LAUNCH_CHECK("ccv_attn_fwd");
    return CCV_OK;
}
