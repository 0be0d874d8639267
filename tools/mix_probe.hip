// Instruction-mix probe for the attention inner step (no memory, no barriers): per iteration and wave the 32-query x 32-key step of
// attn_shared_kernel -- 4 dependent QK MFMAs (32x32x16), the online softmax of csrc/ccv_attn.hip (softmax_block32, all keys visible),
// 4 PV MFMAs -- with either part switched off.  Answers: how much of the softmax's vector work runs under OTHER waves' MFMAs on one
// SIMD, as a function of waves per SIMD.  Build + run: tools/mix_probe.sh (hipcc, standalone executable).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../camc2v_amd/csrc/ccv_attn.hip"

void ccv_set_error(const char*, ...) {}      // csrc/ccv_misc.hip's, not linked here

namespace {

template <int MODE, int OCC>   // MODE 1 = MFMAs only, 2 = softmax only, 3 = both; OCC = workgroups (of 4 waves) per CU wanted
__global__ __launch_bounds__(256, OCC) void mix_kernel(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63, hh = lane >> 5;
    bf16x8 kf[4], qf[4], vf[2][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            kf[s][j] = (ccv_opnd_t)(seed * (float)((lane + s + j) & 7) * 0.01f);
            qf[s][j] = (ccv_opnd_t)(seed * (float)((lane * 3 + s + j) & 7) * 0.01f);
        }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) vf[a][b][j] = (ccv_opnd_t)(seed * (float)((lane + a + 2 * b + j) & 3) * 0.1f);
    f32x16 oacc[2];
    for (int i = 0; i < 16; ++i) oacc[0][i] = oacc[1][i] = 0.f;
    float m_run = NEG_INF, l_run = 0.f;
    bf16x8 pf[2];
    for (int j = 0; j < 8; ++j) pf[0][j] = pf[1][j] = (ccv_opnd_t)0.5f;
    f32x16 sa;
    for (int i = 0; i < 16; ++i) sa[i] = seed * (float)i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
            f32x16 s0;
            for (int i = 0; i < 16; ++i) s0[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) s0 = ccv_mfma_32x32x16(kf[s], qf[s], s0);
            sa = s0;
        } else if (MODE == 1) {      // the same four dependent MFMAs, accumulated (no vector work at all in this mode)
#pragma unroll
            for (int s = 0; s < 4; ++s) sa = ccv_mfma_32x32x16(kf[s], qf[s], sa);
        }
        if (MODE & 2) {
            if (!(MODE & 1)) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(sa[i]));     // opaque scores: the softmax cannot be hoisted, no instruction added
            }
            softmax_block32(sa, 0xffffffffu, true, hh, 0.18f, m_run, l_run, oacc, pf);
        }
        if (MODE & 1) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) oacc[d] = ccv_mfma_32x32x16(vf[d][s2], pf[s2], oacc[d]);
        } else {
            asm volatile("" :: "v"(pf[0]), "v"(pf[1]));
        }
    }
    float acc = l_run + m_run;
    for (int i = 0; i < 16; ++i) acc += oacc[0][i] + oacc[1][i] + sa[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// Waves of two kinds on one SIMD: workgroups with (blockIdx / 256) even issue only MFMAs (8 per step), the others only vector work
// (VEXP v_exp_f32 + VFMA v_pk_fma_f32 per step).  If the matrix pipe runs under other waves' vector instructions the mixed launch takes
// max(parts); if a SIMD issues one or the other, their sum.
template <int VEXP, int VFMA>
__global__ __launch_bounds__(256, 4) void role_kernel(float* out, int iters, int roles, float seed) {
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    // roles 1: all MFMA, 2: all vector, 3: by workgroup (blockIdx / 256), 4: by (wave + blockIdx), 5: by (wave + blockIdx / 256) -- whichever way
    // the dispatcher places workgroups (consecutive ids on one CU, or ids 256 apart), exactly one of 4 / 5 puts two waves of each kind on every SIMD
    const int kind = roles == 3 ? ((blockIdx.x >> 8) & 1) : roles == 4 ? ((wv + blockIdx.x) & 1) : roles == 5 ? ((wv + (blockIdx.x >> 8)) & 1) : roles - 1;
    float acc = 0.f;
    if (kind == 0) {
        bf16x8 a[4], b[4];
        for (int s = 0; s < 4; ++s)
            for (int j = 0; j < 8; ++j) {
                a[s][j] = (ccv_opnd_t)(seed * (float)((lane + s + j) & 7) * 0.01f);
                b[s][j] = (ccv_opnd_t)(seed * (float)((lane * 3 + s + j) & 7) * 0.01f);
            }
        f32x16 c0, c1;
        for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                c0 = ccv_mfma_32x32x16(a[s], b[s], c0);
                c1 = ccv_mfma_32x32x16(b[s], a[s], c1);
            }
        }
        for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
    } else {
        f32x2 x[8];
        float e[8];
        for (int i = 0; i < 8; ++i) {
            x[i] = f32x2{seed * (float)(lane + i) * 1e-3f, seed * 0.5f};
            e[i] = seed * (float)(lane + i) * 1e-2f;
        }
        const f32x2 m = {0.999f, 0.9999f}, c = {1e-4f, 1e-5f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < VFMA / 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], m, c);
#pragma unroll
            for (int r = 0; r < VEXP / 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]) - 1.0f;
        }
        for (int i = 0; i < 8; ++i) acc += x[i][0] + x[i][1] + e[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// One wave issuing both streams itself, interleaved by the compiler's scheduling groups: 1 MFMA, then VFMA / 8 independent v_pk_fma_f32.
template <int VFMA>
__global__ __launch_bounds__(256, 4) void inwave_kernel(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[4];
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < 8; ++j) {
            a[s][j] = (ccv_opnd_t)(seed * (float)((lane + s + j) & 7) * 0.01f);
            b[s][j] = (ccv_opnd_t)(seed * (float)((lane * 3 + s + j) & 7) * 0.01f);
        }
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
    f32x2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f32x2{seed * (float)(lane + i) * 1e-3f, seed * 0.5f};
    const f32x2 m = {0.999f, 0.9999f}, c = {1e-4f, 1e-5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c0 = ccv_mfma_32x32x16(a[s], b[s], c0);
            c1 = ccv_mfma_32x32x16(b[s], a[s], c1);
        }
#pragma unroll
        for (int r = 0; r < VFMA / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], m, c);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, VFMA / 8, 0);     // VFMA / 8 vector instructions
        }
    }
    float acc = 0.f;
    for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
    for (int i = 0; i < 8; ++i) acc += x[i][0] + x[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int VFMA>
float run_inwave(float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    inwave_kernel<VFMA><<<blocks, 256>>>(out, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    inwave_kernel<VFMA><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int VEXP, int VFMA>
float run_roles(float* out, int blocks, int iters, int roles) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    role_kernel<VEXP, VFMA><<<blocks, 256>>>(out, 16, roles, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    role_kernel<VEXP, VFMA><<<blocks, 256>>>(out, iters, roles, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

// The step software-pipelined inside one wave: the four QK MFMAs of step i + 1 (independent of step i) are issued among the softmax of
// step i, the four PV MFMAs of step i follow.  SCHED 0: source order only; 1: sched_group_barrier pattern 1 MFMA : VPER vector instructions.
template <int OCC, int SCHED, int VPER>
__global__ __launch_bounds__(256, OCC) void piped_kernel(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63, hh = lane >> 5;
    bf16x8 kf[4], qf[4], vf[2][2];
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < 8; ++j) {
            kf[s][j] = (ccv_opnd_t)(seed * (float)((lane + s + j) & 7) * 0.01f);
            qf[s][j] = (ccv_opnd_t)(seed * (float)((lane * 3 + s + j) & 7) * 0.01f);
        }
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int j = 0; j < 8; ++j) vf[a][b][j] = (ccv_opnd_t)(seed * (float)((lane + a + 2 * b + j) & 3) * 0.1f);
    f32x16 oacc[2];
    for (int i = 0; i < 16; ++i) oacc[0][i] = oacc[1][i] = 0.f;
    float m_run = NEG_INF, l_run = 0.f;
    bf16x8 pf[2];
    f32x16 s_next;
    for (int i = 0; i < 16; ++i) s_next[i] = seed * (float)i;
    for (int it = 0; it < iters; ++it) {
        f32x16 sa = s_next;
        for (int i = 0; i < 16; ++i) s_next[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) s_next = ccv_mfma_32x32x16(kf[s], qf[s], s_next);
        softmax_block32(sa, 0xffffffffu, true, hh, 0.18f, m_run, l_run, oacc, pf);
        if (SCHED == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, VPER, 0);     // VALU or transcendental
            }
        }
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) oacc[d] = ccv_mfma_32x32x16(vf[d][s2], pf[s2], oacc[d]);
    }
    float acc = l_run + m_run;
    for (int i = 0; i < 16; ++i) acc += oacc[0][i] + oacc[1][i] + s_next[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OCC, int SCHED, int VPER>
float run_piped(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    piped_kernel<OCC, SCHED, VPER><<<256 * OCC, 256>>>(out, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    piped_kernel<OCC, SCHED, VPER><<<256 * OCC, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

// Two 32-query blocks A and B per wave, half a step apart: while the softmax of one block issues its vector instructions, the other block's eight
// independent MFMAs (PV of its previous step, QK of its next) go into the gaps.  The softmax is written in two parts so that the rare rescale
// branch does not separate the MFMAs from the exponentials they are interleaved with.
__device__ __forceinline__ void sm_head(const f32x16& sa, float sl2, float& m_r, float& l_r, f32x16 (&oa)[2], float& m_use) {
    float tmax = NEG_INF;
#pragma unroll
    for (int i = 0; i < 16; i += 2) tmax = fmaxf(fmaxf(tmax, sa[i]), sa[i + 1]);
    const uint32_t tb = __float_as_uint(tmax);
    const auto sw = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);
    asm("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(sw[0]), "v"(sw[1]));
    tmax *= sl2;
    const float m_new = fmaxf(m_r, tmax);
    m_use = (m_new == NEG_INF) ? 0.f : m_new;
    if (!__all(m_new == m_r)) {
        const float alpha = __builtin_amdgcn_exp2f(m_r - m_use);
        l_r *= alpha;
        oa[0] = oa[0] * alpha;
        oa[1] = oa[1] * alpha;
        m_r = m_new;
    }
}
__device__ __forceinline__ void sm_tail(const f32x16& sa, float sl2, float m_use, float& l_r, bf16x8 (&pfo)[2]) {
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
    for (int s2 = 0; s2 < 2; ++s2)
        pfo[s2] = ccv_opnd8(pv[8 * s2], pv[8 * s2 + 1], pv[8 * s2 + 2], pv[8 * s2 + 3], pv[8 * s2 + 4], pv[8 * s2 + 5], pv[8 * s2 + 6], pv[8 * s2 + 7]);
    l_r += psum2[0] + psum2[1];
}

template <int OCC, int SCHED>
__global__ __launch_bounds__(256, OCC) void twoblock_kernel(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    bf16x8 kf[4], qa[4], qb[4], vf[2][2];
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < 8; ++j) {
            kf[s][j] = (ccv_opnd_t)(seed * (float)((lane + s + j) & 7) * 0.01f);
            qa[s][j] = (ccv_opnd_t)(seed * (float)((lane * 3 + s + j) & 7) * 0.01f);
            qb[s][j] = (ccv_opnd_t)(seed * (float)((lane * 5 + s + j) & 7) * 0.01f);
        }
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int j = 0; j < 8; ++j) vf[a][b][j] = (ccv_opnd_t)(seed * (float)((lane + a + 2 * b + j) & 3) * 0.1f);
    f32x16 oA[2], oB[2], sA, sB;
    for (int i = 0; i < 16; ++i) {
        oA[0][i] = oA[1][i] = oB[0][i] = oB[1][i] = 0.f;
        sA[i] = seed * (float)i;
        sB[i] = seed * (float)(15 - i);
    }
    float mA = NEG_INF, lA = 0.f, mB = NEG_INF, lB = 0.f;
    bf16x8 pA[2], pB[2];
    for (int j = 0; j < 8; ++j) pA[0][j] = pA[1][j] = pB[0][j] = pB[1][j] = (ccv_opnd_t)0.5f;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        float mu;
        // phase X: softmax of A's scores; MFMAs: PV of B (previous step), QK of B (this step)
        sm_head(sA, 0.18f, mA, lA, oA, mu);
        {
            f32x16 t = ccv_mfma_32x32x16(kf[0], qb[0], zero);
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) oB[d] = ccv_mfma_32x32x16(vf[d][s2], pB[s2], oB[d]);
#pragma unroll
            for (int s = 1; s < 4; ++s) t = ccv_mfma_32x32x16(kf[s], qb[s], t);
            sm_tail(sA, 0.18f, mu, lA, pA);
            sB = t;
            if (SCHED) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, SCHED, 0);
                }
            }
        }
        // phase Y: softmax of B's scores; MFMAs: PV of A (this step), QK of A (next step)
        sm_head(sB, 0.18f, mB, lB, oB, mu);
        {
            f32x16 t = ccv_mfma_32x32x16(kf[0], qa[0], zero);
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) oA[d] = ccv_mfma_32x32x16(vf[d][s2], pA[s2], oA[d]);
#pragma unroll
            for (int s = 1; s < 4; ++s) t = ccv_mfma_32x32x16(kf[s], qa[s], t);
            sm_tail(sB, 0.18f, mu, lB, pB);
            sA = t;
            if (SCHED) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, SCHED, 0);
                }
            }
        }
    }
    float acc = lA + mA + lB + mB;
    for (int i = 0; i < 16; ++i) acc += oA[0][i] + oA[1][i] + oB[0][i] + oB[1][i] + sA[i] + sB[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// The same two-block schedule with the order pinned in the source: one MFMA, then a slice of the other block's softmax, sched_barrier(0) between
// the slices (nothing moves across).  o = the softmax block's accumulators (rescale), po / pi = its P fragments out, the other block's in.
#define PIN() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void pinned_phase(f32x16& s_sm, float& m_r, float& l_r, f32x16 (&o_sm)[2], bf16x8 (&p_out)[2],
                                             f32x16 (&o_mm)[2], const bf16x8 (&p_in)[2], f32x16& s_out, const bf16x8 (&kf)[4], const bf16x8 (&qf)[4],
                                             const bf16x8 (&vf)[2][2], float sl2) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // slices 0, 1: PV MFMAs of the other block; row maximum of this one
    o_mm[0] = ccv_mfma_32x32x16(vf[0][0], p_in[0], o_mm[0]);
    float t0 = fmaxf(fmaxf(s_sm[0], s_sm[1]), s_sm[2]), t1 = fmaxf(fmaxf(s_sm[3], s_sm[4]), s_sm[5]);
    t0 = fmaxf(fmaxf(t0, s_sm[6]), s_sm[7]);
    t1 = fmaxf(fmaxf(t1, s_sm[8]), s_sm[9]);
    PIN();
    o_mm[0] = ccv_mfma_32x32x16(vf[0][1], p_in[1], o_mm[0]);
    t0 = fmaxf(fmaxf(t0, s_sm[10]), s_sm[11]);
    t1 = fmaxf(fmaxf(t1, s_sm[12]), s_sm[13]);
    float tmax = fmaxf(fmaxf(fmaxf(t0, t1), s_sm[14]), s_sm[15]);
    const uint32_t tb = __float_as_uint(tmax);
    const auto sw = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);
    asm("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(sw[0]), "v"(sw[1]));
    tmax *= sl2;
    const float m_new = fmaxf(m_r, tmax);
    const float m_use = (m_new == NEG_INF) ? 0.f : m_new;
    PIN();
    if (!__all(m_new == m_r)) {
        const float alpha = __builtin_amdgcn_exp2f(m_r - m_use);
        l_r *= alpha;
        o_sm[0] = o_sm[0] * alpha;
        o_sm[1] = o_sm[1] * alpha;
        m_r = m_new;
    }
    PIN();
    const f32x2 scale2 = {sl2, sl2}, negm2 = {-m_use, -m_use};
    f32x2 psum2 = {0.f, 0.f};
    float pv[16];
    f32x16 t;
#define EXP4(i0)                                                                                             \
    _Pragma("unroll") for (int i = i0; i < i0 + 4; i += 2) {                                                 \
        const f32x2 s2 = {s_sm[i], s_sm[i + 1]};                                                             \
        const f32x2 x = __builtin_elementwise_fma(s2, scale2, negm2);                                        \
        const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};                        \
        pv[i] = e[0];                                                                                        \
        pv[i + 1] = e[1];                                                                                    \
        psum2 = psum2 + e;                                                                                   \
    }
    o_mm[1] = ccv_mfma_32x32x16(vf[1][0], p_in[0], o_mm[1]);
    EXP4(0)
    PIN();
    o_mm[1] = ccv_mfma_32x32x16(vf[1][1], p_in[1], o_mm[1]);
    EXP4(4)
    PIN();
    t = ccv_mfma_32x32x16(kf[0], qf[0], zero);
    EXP4(8)
    PIN();
    t = ccv_mfma_32x32x16(kf[1], qf[1], t);
    EXP4(12)
    PIN();
    t = ccv_mfma_32x32x16(kf[2], qf[2], t);
    p_out[0] = ccv_opnd8(pv[0], pv[1], pv[2], pv[3], pv[4], pv[5], pv[6], pv[7]);
    PIN();
    t = ccv_mfma_32x32x16(kf[3], qf[3], t);
    p_out[1] = ccv_opnd8(pv[8], pv[9], pv[10], pv[11], pv[12], pv[13], pv[14], pv[15]);
    l_r += psum2[0] + psum2[1];
    PIN();
    s_out = t;
#undef EXP4
}

template <int OCC>
__global__ __launch_bounds__(256, OCC) void pinned_kernel(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    bf16x8 kf[4], qa[4], qb[4], vf[2][2];
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < 8; ++j) {
            kf[s][j] = (ccv_opnd_t)(seed * (float)((lane + s + j) & 7) * 0.01f);
            qa[s][j] = (ccv_opnd_t)(seed * (float)((lane * 3 + s + j) & 7) * 0.01f);
            qb[s][j] = (ccv_opnd_t)(seed * (float)((lane * 5 + s + j) & 7) * 0.01f);
        }
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int j = 0; j < 8; ++j) vf[a][b][j] = (ccv_opnd_t)(seed * (float)((lane + a + 2 * b + j) & 3) * 0.1f);
    f32x16 oA[2], oB[2], sA, sB;
    for (int i = 0; i < 16; ++i) {
        oA[0][i] = oA[1][i] = oB[0][i] = oB[1][i] = 0.f;
        sA[i] = seed * (float)i;
        sB[i] = seed * (float)(15 - i);
    }
    float mA = NEG_INF, lA = 0.f, mB = NEG_INF, lB = 0.f;
    bf16x8 pA[2], pB[2];
    for (int j = 0; j < 8; ++j) pA[0][j] = pA[1][j] = pB[0][j] = pB[1][j] = (ccv_opnd_t)0.5f;
    for (int it = 0; it < iters; ++it) {
        pinned_phase(sA, mA, lA, oA, pA, oB, pB, sB, kf, qb, vf, 0.18f);     // softmax A | PV B (previous step), QK B
        pinned_phase(sB, mB, lB, oB, pB, oA, pA, sA, kf, qa, vf, 0.18f);     // softmax B | PV A, QK A (next step)
    }
    float acc = lA + mA + lB + mB;
    for (int i = 0; i < 16; ++i) acc += oA[0][i] + oA[1][i] + oB[0][i] + oB[1][i] + sA[i] + sB[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OCC>
float run_pinned(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    pinned_kernel<OCC><<<256 * OCC, 256>>>(out, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    pinned_kernel<OCC><<<256 * OCC, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int OCC, int SCHED>
float run_twoblock(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    twoblock_kernel<OCC, SCHED><<<256 * OCC, 256>>>(out, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    twoblock_kernel<OCC, SCHED><<<256 * OCC, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int MODE, int OCC>
float run(float* out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    mix_kernel<MODE, OCC><<<blocks, 256>>>(out, 16, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix_kernel<MODE, OCC><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

}  // namespace

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("# per step and wave: 8 v_mfma_f32_32x32x16 (256 matrix-pipe cycles) + softmax_block32; clock attribute %.2f GHz; %d steps per wave\n", clk_khz * 1e-6, iters);
    printf("# waves/SIMD  mode      ms      ns per step-round (all waves of a SIMD advance one step)   cycles at 2.4 GHz   per wave-step\n");
    const char* names[4] = {"", "mfma", "softmax", "both"};
#define ROW(MODE, OCC)                                                                                                    \
    {                                                                                                                     \
        const float ms = run<MODE, OCC>(out, 256 * OCC, iters);                                                           \
        const double ns = ms * 1e6 / iters;                                                                               \
        printf("  %d          %-8s %8.3f   %8.1f   %8.0f   %8.0f\n", OCC, names[MODE], ms, ns, ns * 2.4, ns * 2.4 / OCC); \
    }
    ROW(1, 1) ROW(2, 1) ROW(3, 1)
    ROW(1, 2) ROW(2, 2) ROW(3, 2)
    ROW(1, 4) ROW(2, 4) ROW(3, 4)
    printf("# role split, 4 workgroups per CU (4 waves per SIMD); cycles per step at 2.4 GHz (a step = 8 MFMAs or the vector mix)\n");
#define ROLES(VEXP, VFMA)                                                                                                                   \
    {                                                                                                                                       \
        const float a2 = run_roles<VEXP, VFMA>(out, 512, iters, 1), b2 = run_roles<VEXP, VFMA>(out, 512, iters, 2);                       \
        const float a4 = run_roles<VEXP, VFMA>(out, 1024, iters, 1), b4 = run_roles<VEXP, VFMA>(out, 1024, iters, 2);                     \
        const float mix = run_roles<VEXP, VFMA>(out, 1024, iters, 3);                                                                      \
        const float mix4 = run_roles<VEXP, VFMA>(out, 1024, iters, 4), mix5 = run_roles<VEXP, VFMA>(out, 1024, iters, 5);                  \
        const double k = 1e6 / iters * 2.4;                                                                                                 \
        printf("  vector mix %2d v_exp + %2d v_pk_fma:  2 MFMA waves/SIMD %6.0f   2 vector waves/SIMD %6.0f   4 MFMA %6.0f   4 vector %6.0f   "  \
               "2 + 2 mixed by workgroup %6.0f, by wave + workgroup %6.0f, by wave + workgroup / 256 %6.0f  (max of parts %6.0f, sum %6.0f)\n", VEXP, VFMA, a2 * k, b2 * k, a4 * k, b4 * k, mix * k, mix4 * k, mix5 * k, \
               (a2 > b2 ? a2 : b2) * k, (a2 + b2) * k);                                                                                     \
    }
    ROLES(16, 32) ROLES(0, 64) ROLES(32, 0) ROLES(16, 64) ROLES(0, 16)
    printf("# one wave issuing 8 MFMAs and N v_pk_fma_f32 per step, interleaved 1 : N / 8; cycles per step-round at 2.4 GHz, 1 / 2 / 4 waves per SIMD\n");
#define INWAVE(VFMA)                                                                                                                  \
    {                                                                                                                                 \
        const double k = 1e6 / iters * 2.4;                                                                                           \
        printf("  8 MFMA + %2d v_pk_fma in one wave: %6.0f %6.0f %6.0f   (MFMA alone 256 / 512 / 1024, vector alone %d / %d / %d)\n", VFMA, \
               run_inwave<VFMA>(out, 256, iters) * k, run_inwave<VFMA>(out, 512, iters) * k, run_inwave<VFMA>(out, 1024, iters) * k, 4 * VFMA, 8 * VFMA, 16 * VFMA); \
    }
    INWAVE(8) INWAVE(32) INWAVE(56) INWAVE(64)
    printf("# step pipelined in the wave (QK of step i + 1 among the softmax of step i); cycles per wave-step at 2.4 GHz, 1 / 2 / 4 waves per SIMD\n");
    {
        const double k = 1e6 / iters * 2.4;
        printf("  source order only:              %6.0f %6.0f %6.0f\n", run_piped<1, 0, 0>(out, iters) * k, run_piped<2, 0, 0>(out, iters) * k / 2, run_piped<4, 0, 0>(out, iters) * k / 4);
        printf("  1 MFMA : 8 vector instructions: %6.0f %6.0f %6.0f\n", run_piped<1, 1, 8>(out, iters) * k, run_piped<2, 1, 8>(out, iters) * k / 2, run_piped<4, 1, 8>(out, iters) * k / 4);
        printf("  1 MFMA : 14 vector instructions:%6.0f %6.0f %6.0f\n", run_piped<1, 1, 14>(out, iters) * k, run_piped<2, 1, 14>(out, iters) * k / 2, run_piped<4, 1, 14>(out, iters) * k / 4);
    }
    printf("# two query blocks per wave half a step apart (softmax of one among the other's 8 MFMAs); cycles per 32-query wave-step at 2.4 GHz, 1 / 2 waves per SIMD\n");
    {
        const double k = 1e6 / iters * 2.4 / 2;      // an iteration = two wave-steps
        printf("  source order:                    %6.0f %6.0f\n", run_twoblock<1, 0>(out, iters) * k, run_twoblock<2, 0>(out, iters) * k / 2);
        printf("  1 MFMA : 5 vector instructions:  %6.0f %6.0f\n", run_twoblock<1, 5>(out, iters) * k, run_twoblock<2, 5>(out, iters) * k / 2);
        printf("  1 MFMA : 6 vector instructions:  %6.0f %6.0f\n", run_twoblock<1, 6>(out, iters) * k, run_twoblock<2, 6>(out, iters) * k / 2);
        printf("  1 MFMA : 7 vector instructions:  %6.0f %6.0f\n", run_twoblock<1, 7>(out, iters) * k, run_twoblock<2, 7>(out, iters) * k / 2);
        printf("  order pinned in the source:      %6.0f %6.0f\n", run_pinned<1>(out, iters) * k, run_pinned<2>(out, iters) * k / 2);
    }
    hipFree(out);
    return 0;
}
