// ccv_gemm: implicit-GEMM (linear / conv3x3 / temporal conv) on bf16 MFMA for gfx950.
//
// C[M,N] = epilogue( sum_tap gather_tap(A)[M,K] . W_tap[N,K]^T ),  fp32 accumulate.
//
// Tiling: one workgroup = 256 threads = 4 waves (2 x 2), block tile (32*MT) x (32*NT),
// wave tile (16*MT) x (16*NT) built from v_mfma_f32_16x16x32_bf16.  The MFMA is issued
// with the WEIGHT fragment as the A operand and the ACTIVATION fragment as the B operand,
// so a lane ends up with 4 consecutive output columns of one output row: bias, residual
// and the store are 8/16-byte vector accesses and GEGLU pairs live in one lane.
// K is walked in 64-wide slabs; both operands are staged global -> registers -> LDS
// (register staging lets the loader zero-fill conv padding, convert fp32 activations and
// gather rows per tap), LDS is double buffered and XOR-swizzled so the ds_read_b128
// fragment reads are bank-conflict free (16-byte chunk c of row r lives at c ^ ((r>>1)&7)).
// blockIdx is remapped so that the workgroups of one XCD walk neighbouring tiles (L2 reuse).
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "ccv_common.h"

namespace {

constexpr int BK = 64;  // K granularity every problem must satisfy (K % 64 == 0)

std::atomic<int> g_streams_in_flight{1};   // ccv_set_streams_in_flight

// epilogue for 4 consecutive output columns n..n+3 of row m (acc already holds the full K sum)
// compile-time loop: f(std::integral_constant<int, B>), f(<B + S>), ... while < E (accumulator fragments must be indexed
// by constants, or the whole accumulator array moves to scratch memory)
template <int B, int E, int S, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + S, E, S>(f);
    }
}

// fp16 <-> fp32 (the residual stream's hand-off format, CcvGemm.out_f32 == 2 / res_f16): v_cvt_f16_f32 / v_cvt_f32_f16, round to nearest even
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) { return ccv_pack_f16x2(lo, hi); }
__device__ __forceinline__ float2 unpack_f16x2(uint32_t u) { return ccv_unpack_f16x2(u); }
// two-byte outputs: bf16 (out_f32 == 0) or fp16 (out_f32 == 2)
__device__ __forceinline__ uint32_t pack_out2(const CcvGemm& p, float lo, float hi) {
    return p.out_f32 == 2 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi);
}

// bias_pre / bias2_pre: the fragment's bias values already in registers (tile epilogues fetch a row fragment's NT column groups together, ahead
// of the arithmetic: fetched here, every fragment pays a global load + s_waitcnt vmcnt(0) of its own -- 20 dependent round trips per wave and
// tile -- and, for bias2, a 25-instruction integer division).  Same values, same order of operations: bit-identical results.
__device__ __forceinline__ void epilogue_math(const CcvGemm& p, int m, int n, float o[4], bool add_residual = true, const float4* bias_pre = nullptr,
                                              const float4* bias2_pre = nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] *= p.alpha;
    if (p.bias) {
        const float4 bv = bias_pre ? *bias_pre : *reinterpret_cast<const float4*>(p.bias + n);
        o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
    }
    if (p.bias2) {
        const float4 bv = bias2_pre ? *bias2_pre : *reinterpret_cast<const float4*>(p.bias2 + (long)(m / p.rows_per_batch) * p.ldb2 + n);
        o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
    }
    if (p.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = silu_f(o[r]);
    } else if (p.act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = gelu_erf_f(o[r]);
    } else if (p.act == 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
    }
    if (p.residual && add_residual) {
        if (p.res_f16) {
            const uint2 rv = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(p.residual) + (long)m * p.ldr + n);
            const float2 a = unpack_f16x2(rv.x), b = unpack_f16x2(rv.y);
            o[0] += a.x; o[1] += a.y; o[2] += b.x; o[3] += b.y;
        } else {
            const float4 rv = *reinterpret_cast<const float4*>(static_cast<const float*>(p.residual) + (long)m * p.ldr + n);
            o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
        }
    }
}

// the NT column groups (16 apart) of one row of bias values, all loads in flight together; row == nullptr or a group past N: zeros (not used)
template <int NT>
__device__ __forceinline__ void preload_cols(const float* row, int n_first, int N, float4 (&v)[NT]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n_first + 16 * j;
        v[j] = (row != nullptr && n < N) ? *reinterpret_cast<const float4*>(row + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// row of the per-batch bias (bias2) that output row m takes, or nullptr
__device__ __forceinline__ const float* bias2_row(const CcvGemm& p, int m) {
    return (p.bias2 && m < p.M) ? p.bias2 + (long)(m / p.rows_per_batch) * p.ldb2 : nullptr;
}
// The 320-column tiles (NT = 10: 160 accumulator registers) keep the per-fragment fetch of bias2: 40 more live registers spill there.
template <int NT>
struct Bias2Pre {
    static constexpr bool on = NT <= 5;
    float4 v[on ? NT : 1];
    __device__ __forceinline__ void load(const CcvGemm& p, int m, int n_first) {
        if constexpr (on) preload_cols<NT>(bias2_row(p, m), n_first, p.N, v);
    }
    __device__ __forceinline__ const float4* at(int j) const { return on ? &v[on ? j : 0] : nullptr; }
};

__device__ __forceinline__ void epilogue_store(const CcvGemm& p, int m, int n, float o[4], const float4* bias_pre = nullptr, const float4* bias2_pre = nullptr) {
    epilogue_math(p, m, n, o, true, bias_pre, bias2_pre);
    if (p.out_f32 == 1) {
        *reinterpret_cast<float4*>(static_cast<float*>(p.C) + (long)m * p.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        uint2 pk = make_uint2(pack_out2(p, o[0], o[1]), pack_out2(p, o[2], o[3]));
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.C) + (long)m * p.ldc + n) = pk;
    }
}

// bf16 outputs, two 16-column fragments side by side: the lane holds columns c .. c+3 (a) and c+16 .. c+19 (b) of row m,
// c = 16-aligned base + 4 fg.  Lanes fg and fg ^ 1 (16 lanes apart, same row) swap one half each, so that every lane
// stores 8 consecutive bf16 with one 16-byte store instead of two 8-byte ones: half the store instructions and 64
// contiguous bytes per row and instruction.  (Output stores are issue-bound at the end of a tile.)  All 64 lanes of the
// wave must call it together with the same row activity per 16-lane row group (rows are per fr = lane & 15: both partners
// share it).  Needs 16-byte aligned rows (ldc % 8 == 0, aligned C): callers check p_wide.
__device__ __forceinline__ void store_pair_bf16(uint16_t* crow, int c, uint2 a, uint2 b) {
    const bool odd = (threadIdx.x >> 4) & 1;
    const uint2 send = odd ? a : b;
    uint2 recv;
    recv.x = (uint32_t)__shfl_xor((int)send.x, 16, 64);
    recv.y = (uint32_t)__shfl_xor((int)send.y, 16, 64);
    const uint4 out = odd ? make_uint4(recv.x, recv.y, b.x, b.y) : make_uint4(a.x, a.y, recv.x, recv.y);
    *reinterpret_cast<uint4*>(crow + (odd ? c + 12 : c)) = out;
}

__device__ int g_wide_store = 1;   // CCV_GEMM_WIDE_STORE=0 clears it (A/B aid)
__device__ __forceinline__ bool wide_bf16_ok(const CcvGemm& p) {
    return p.out_f32 != 1 && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && g_wide_store != 0;
}

__device__ __forceinline__ void epilogue_store_pair(const CcvGemm& p, int m, int n, float o0[4], float o1[4], const float4* bias_pre = nullptr,
                                                    const float4* bias2_pre = nullptr) {      // the pre-loaded values of BOTH fragments: [0], [1]
    epilogue_math(p, m, n, o0, true, bias_pre, bias2_pre);
    epilogue_math(p, m, n + 16, o1, true, bias_pre ? bias_pre + 1 : nullptr, bias2_pre ? bias2_pre + 1 : nullptr);
    store_pair_bf16(static_cast<uint16_t*>(p.C) + (long)m * p.ldc, n, make_uint2(pack_out2(p, o0[0], o0[1]), pack_out2(p, o0[2], o0[3])),
                    make_uint2(pack_out2(p, o1[0], o1[1]), pack_out2(p, o1[2], o1[3])));
}

// GEGLU: value columns n..n+3 and gate columns n+16..n+19 of the interleaved weight layout
// bias_pre: the value fragment's and (bias_pre[1]) the gate fragment's bias values, already in registers (see epilogue_math)
__device__ __forceinline__ uint2 geglu_value(const CcvGemm& p, int n, const float a_[4], const float g_[4], const float4* bias_pre = nullptr) {
    float o[4];
    float ba[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        const float4 va = bias_pre ? bias_pre[0] : *reinterpret_cast<const float4*>(p.bias + n);
        const float4 vg = bias_pre ? bias_pre[1] : *reinterpret_cast<const float4*>(p.bias + n + 16);
        ba[0] = va.x; ba[1] = va.y; ba[2] = va.z; ba[3] = va.w;
        bg[0] = vg.x; bg[1] = vg.y; bg[2] = vg.z; bg[3] = vg.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = a_[r] * p.alpha, g = g_[r] * p.alpha;
        if (p.bias) { a += ba[r]; g += bg[r]; }
        o[r] = a * gelu_erf_f(g);
    }
    return make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
}

__device__ __forceinline__ void epilogue_geglu(const CcvGemm& p, int m, int n, const float a_[4], const float g_[4], const float4* bias_pre = nullptr) {
    const int nc = (n >> 5) * 16 + (n & 15);
    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.C) + (long)m * p.ldc + nc) = geglu_value(p, n, a_, g_, bias_pre);
}


// Row-coalescing epilogue of gemm_dma_kernel for two-byte outputs (bf16 / fp16), with or without an fp16 residual.  In the MFMA
// fragment layout a store instruction covers 16 rows x 32-64 bytes; at the end of a launch every workgroup does that at once and the
// burst runs at the write path's rate for partial lines (profiles/r03_family_kernel_stamps.txt: 14-32 thousand cycles per tile).  Here
// the tile goes through LDS (the operand stages are free by now): the residual tile comes in with whole-row 16-byte loads, every lane
// adds its fragments' values (read from LDS, fp32 arithmetic, ONE rounding, as in the direct epilogue: bit-identical results), puts the
// packed result back in place, and the tile leaves with whole-row 16-byte stores (tools/probes/store_pattern_probe.hip: 1.2-1.45x the
// rate of the fragment pattern).  Needs 16-byte aligned rows of C (and of the residual); the caller falls back otherwise.  The
// statistics-emitting instances keep the direct stores: the same route through LDS measured -0.6 ... +1.7 % there (box noise).
template <int MT, int NT>
__device__ __forceinline__ bool rows_epilogue_ok(const CcvGemm& p) {
    return p.out_f32 != 1 && !p.geglu && p.split_k <= 1 && (p.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 &&
           (!p.residual || (p.res_f16 && (p.ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(p.residual) & 15) == 0));
}
template <int MT, int NT>
__device__ __forceinline__ void tile_epilogue_rows(const CcvGemm& p, const f32x4 (&acc)[MT][NT], int m0, int n0, unsigned char* smem) {
    constexpr int BM = 32 * MT, BN = 32 * NT, CPR = BN / 8, PITCH = BN * 2 + 16;     // 16-byte chunks per tile row; LDS row pitch in bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, fr = lane & 15, fg = lane >> 4;
    const bool rmw = p.residual != nullptr;
    __syncthreads();      // every wave is past its last fragment read of the operand stages
    // the tile's 16-byte chunks are dealt to the 256 threads IT at a time, fully unrolled with every load issued before the first use: as a
    // rolled loop each iteration waited for its own load (s_waitcnt vmcnt(0) + ds_write per chunk: IT dependent L2 round trips per tile)
    constexpr int CHUNKS = BM * CPR, IT = (CHUNKS + 255) / 256;
    if (rmw) {
        const uint16_t* R = static_cast<const uint16_t*>(p.residual);
        uint4 rv[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = tid + 256 * it;
            const int row = c / CPR, col = 8 * (c - row * CPR), m = m0 + row, n = n0 + col;
            rv[it] = make_uint4(0u, 0u, 0u, 0u);
            if (c < CHUNKS && m < p.M && n < p.N) rv[it] = *reinterpret_cast<const uint4*>(R + (long)m * p.ldr + n);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = tid + 256 * it;
            const int row = c / CPR, col = 8 * (c - row * CPR);
            if (c < CHUNKS) *reinterpret_cast<uint4*>(smem + row * PITCH + col * 2) = rv[it];
        }
        __syncthreads();
    }
    const int n_first = n0 + wn * 16 * NT + 4 * fg;
    float4 bpre[NT];
    preload_cols<NT>(p.bias, n_first, p.N, bpre);
    static_for<0, MT, 1>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        const int rl = wm * 16 * MT + 16 * i + fr, m = m0 + rl;
        Bias2Pre<NT> b2pre;
        b2pre.load(p, m, n_first);
        static_for<0, NT, 1>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            const int cl = wn * 16 * NT + 16 * j + 4 * fg, n = n0 + cl;
            if (m < p.M && n < p.N) {
                float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                epilogue_math(p, m, n, o, false, &bpre[j], b2pre.at(j));
                uint2* slot = reinterpret_cast<uint2*>(smem + rl * PITCH + cl * 2);
                if (rmw) {
                    const uint2 rv = *slot;
                    const float2 a = unpack_f16x2(rv.x), b = unpack_f16x2(rv.y);
                    o[0] += a.x; o[1] += a.y; o[2] += b.x; o[3] += b.y;
                }
                *slot = make_uint2(pack_out2(p, o[0], o[1]), pack_out2(p, o[2], o[3]));
            }
        });
    });
    __syncthreads();
    uint16_t* Cp = static_cast<uint16_t*>(p.C);
    uint4 ov[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = tid + 256 * it;
        const int row = c / CPR, col = 8 * (c - row * CPR);
        ov[it] = make_uint4(0u, 0u, 0u, 0u);
        if (c < CHUNKS) ov[it] = *reinterpret_cast<const uint4*>(smem + row * PITCH + col * 2);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = tid + 256 * it;
        const int row = c / CPR, col = 8 * (c - row * CPR), m = m0 + row, n = n0 + col;
        if (c < CHUNKS && m < p.M && n < p.N) *reinterpret_cast<uint4*>(Cp + (long)m * p.ldc + n) = ov[it];
    }
}

// Epilogue of the statistics-emitting kernel instances (p.gn_partial; the output feeds a GroupNorm(32)): the normal epilogue
// (alpha, bias, bias2, residual, store as bf16 / fp16 / fp32) AND the sums and sums of squares of every group's channels over this
// tile's rows, of the values AS STORED (rounded to the output type), written to slot (row tile inside the instance) * (column
// tiles) + (column tile) of the instance the tile's rows belong to -- the layout gn_apply's prologue reduces (ccv_norm.hip).
// Work per lane: its 4 columns of a row are two channel PAIRS, and a pair never straddles a group (channels per group are even);
// pair sums accumulate over the wave's row fragments, fold over the 16 row lanes with DPP steps, land in LDS (the operand stages
// are free by now) and 64 threads -- one per (group, moment) -- add up their group's entries in a fixed order: no atomics, bitwise
// reproducible.  Tile geometry shared by gemm_dma_kernel and gemm_ring_kernel: 2 x 2 waves, wave tile 16 MT x 16 NT, lane = row
// (lane & 15) x column quad (lane >> 4) of a 16 x 16 fragment.  The host admits only problems without activation / GEGLU, with
// every row inside M and whole tiles inside an instance (ccv_gemm_gn_slots).
__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 lanes of a DPP row (= the 16 rows of a fragment), no LDS
    auto step = [](float x, auto ctrl) {
        constexpr int c = decltype(ctrl)::value;
        return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), c, 0xf, 0xf, true));
    };
    v = step(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{});   // row_mirror
    return v;
}

// o[0..3] rounded to the two-byte output type: the packed values, and o[] replaced by the values as stored
__device__ __forceinline__ uint2 round_pack(const CcvGemm& p, float o[4]) {
    uint2 pk;
    if (p.out_f32 == 2) {
        pk = make_uint2(pack_f16x2(o[0], o[1]), pack_f16x2(o[2], o[3]));
        const float2 a = unpack_f16x2(pk.x), b = unpack_f16x2(pk.y);
        o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
    } else {
        pk = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        ccv_opnd2_to_f32(pk.x, o[0], o[1]);
        ccv_opnd2_to_f32(pk.y, o[2], o[3]);
    }
    return pk;
}

// store o[0..3] at C[m][n..n+3] in the output type and hand back the values as stored
__device__ __forceinline__ void store_rounded(const CcvGemm& p, int m, int n, float o[4]) {
    if (p.out_f32 == 1) {
        *reinterpret_cast<float4*>(static_cast<float*>(p.C) + (long)m * p.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
    } else if (p.out_f32 == 2) {
        const uint2 pk = make_uint2(pack_f16x2(o[0], o[1]), pack_f16x2(o[2], o[3]));
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.C) + (long)m * p.ldc + n) = pk;
        const float2 a = unpack_f16x2(pk.x), b = unpack_f16x2(pk.y);
        o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
    } else {
        const uint2 pk = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.C) + (long)m * p.ldc + n) = pk;
        ccv_opnd2_to_f32(pk.x, o[0], o[1]);
        ccv_opnd2_to_f32(pk.y, o[2], o[3]);
    }
}

template <int MT, int NT>
__device__ __forceinline__ void tile_epilogue_gn(const CcvGemm& p, const f32x4 (&acc)[MT][NT], int m0, int n0, int tiles_n, unsigned char* smem) {
    constexpr int BM = 32 * MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, fr = lane & 15, fg = lane >> 4;
    float gs[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j) gs[j][0] = gs[j][1] = gs[j][2] = gs[j][3] = 0.f;
    const bool wide = wide_bf16_ok(p);
    const int n_first = n0 + wn * 16 * NT + 4 * fg;
    float4 bpre[NT];
    preload_cols<NT>(p.bias, n_first, p.N, bpre);
    static_for<0, MT, 1>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        const int m = m0 + wm * 16 * MT + 16 * i + fr;      // every row is inside M (host check)
        Bias2Pre<NT> b2pre;
        b2pre.load(p, m, n_first);
        static_for<0, NT, 2>([&](auto J) __attribute__((always_inline)) {      // fragments two at a time: one 16-byte store per lane
            constexpr int j = decltype(J)::value;                              // (store_pair_bf16) where both are in range
            const int n = n0 + wn * 16 * NT + 16 * j + 4 * fg;
            if (n < p.N) {                                   // columns past N are not stored and add nothing
                float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                epilogue_math(p, m, n, o, true, &bpre[j], b2pre.at(j));
                bool paired = false;
                if constexpr (j + 1 < NT) {
                    if (wide && n - 4 * fg + 32 <= p.N) {
                        float o1[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
                        epilogue_math(p, m, n + 16, o1, true, &bpre[j + 1], b2pre.at(j + 1));
                        const uint2 a = round_pack(p, o), b = round_pack(p, o1);
                        store_pair_bf16(static_cast<uint16_t*>(p.C) + (long)m * p.ldc, n, a, b);
                        gs[j + 1][0] += o1[0] + o1[1]; gs[j + 1][1] += o1[0] * o1[0] + o1[1] * o1[1];
                        gs[j + 1][2] += o1[2] + o1[3]; gs[j + 1][3] += o1[2] * o1[2] + o1[3] * o1[3];
                        paired = true;
                    }
                }
                if (!paired) store_rounded(p, m, n, o);
                gs[j][0] += o[0] + o[1]; gs[j][1] += o[0] * o[0] + o[1] * o[1];
                gs[j][2] += o[2] + o[3]; gs[j][3] += o[2] * o[2] + o[3] * o[3];
                if constexpr (j + 1 < NT) {
                    if (!paired && n + 16 < p.N) {
                        float o1[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
                        epilogue_math(p, m, n + 16, o1, true, &bpre[j + 1], b2pre.at(j + 1));
                        store_rounded(p, m, n + 16, o1);
                        gs[j + 1][0] += o1[0] + o1[1]; gs[j + 1][1] += o1[0] * o1[0] + o1[1] * o1[1];
                        gs[j + 1][2] += o1[2] + o1[3]; gs[j + 1][3] += o1[2] * o1[2] + o1[3] * o1[3];
                    }
                }
            }
        });
    });
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) gs[j][k] = row16_sum(gs[j][k]);
    __syncthreads();   // every wave has read its last operand stage
    float* ent = reinterpret_cast<float*>(smem);   // [wave][NT][column quad] x (pair 0 sum, sq, pair 1 sum, sq): 4 * NT * 4 * 16 B <= 2.5 KiB
    if (fr == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<float4*>(ent + ((wave * NT + j) * 4 + fg) * 4) = make_float4(gs[j][0], gs[j][1], gs[j][2], gs[j][3]);
    }
    __syncthreads();
    if (tid < 64) {
        const int g = tid >> 1, k = tid & 1, cpg = p.N >> 5;
        float a = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; c += 2) {      // the pairs of group g, in channel order
            const int local = c - n0;
            if (local < 0 || local >= 32 * NT) continue;
            const int w = local / (16 * NT), rem = local - w * (16 * NT);     // wave column, column inside the wave tile
            const int j = rem >> 4, f = (rem >> 2) & 3, pr = (rem >> 1) & 1;
#pragma unroll
            for (int half = 0; half < 2; ++half)                               // the two wave rows
                a += ent[(((2 * half + w) * NT + j) * 4 + f) * 4 + 2 * pr + k];
        }
        const int inst = m0 / p.gn_rows, row_tile = (m0 - inst * p.gn_rows) / BM;
        p.gn_partial[((long)inst * p.gn_slots + row_tile * tiles_n + (n0 / (32 * NT))) * 64 + tid] = a;
    }
}

// Tiles / gathers that have a statistics-emitting kernel instance: every family tile for linear layers (proj_out of the
// transformers), 3x3 and temporal convolutions; the 4-deep 128x160 ring and the 2-deep 128x320 ring (decoder convolutions).
constexpr bool gn_dma_tile(int mt, int nt, int gather) {
    return gather >= 0 && gather <= 2 && ((mt == 4 && nt == 5) || (mt == 4 && nt == 4) || (mt == 2 && nt == 4) || (mt == 4 && nt == 2) || (mt == 2 && nt == 2));
}
constexpr bool gn_ring_tile(int mt, int nt, int stages, int gather) {
    return (gather == 1 || gather == 2) && ((mt == 4 && nt == 5 && stages == 4) || (mt == 4 && nt == 10 && stages == 2 && gather == 1));
}

// Workgroup -> (split, tile).  blockIdx is first remapped so that each XCD (blocks with equal blockIdx % 8 under the observed
// round-robin dispatch: speed only) owns a contiguous range of work items, then the range is walked
//   tile_order 0: output tiles row-major (N fastest): an XCD holds a few row bands -> its L2 reads A once and all of W;
//   tile_order 1: column-major (M fastest): an XCD holds a few column bands -> its L2 reads W once and all of A.
// The library picks the order by which operand is larger (ccv_gemm: weights of the 8x8 / 4x4-latent layers are 10-60 MB against
// 1-5 MB of activations; with order 0 every one of the 8 XCDs pulled the whole weight matrix through the fabric).
__device__ __forceinline__ void block_tile(const CcvGemm& p, int BM, int BN, int tiles_n, int& split, int& m0, int& n0) {
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    split = (p.split_k > 1) ? bid % p.split_k : 0;
    if (p.split_k > 1) bid /= p.split_k;
    if (p.tile_order) {
        const int tiles_m = (p.M + BM - 1) / BM;
        m0 = (bid % tiles_m) * BM;
        n0 = (bid / tiles_m) * BN;
    } else {
        m0 = (bid / tiles_n) * BM;
        n0 = (bid % tiles_n) * BN;
    }
}

// BKT = K-slab depth (bf16 elements): 64 -> 128-byte LDS rows, 2 MFMA k-steps per slab, 64 KiB of LDS for a
// 128x128 tile (2 workgroups per CU); 32 -> 64-byte rows, 1 k-step per slab, 32 KiB (4-5 workgroups per CU:
// more waves in flight to hide the global-load latency of short-K problems).
template <int BKT>
__device__ __forceinline__ int lds_off(int row, int chunk) {
    // 16-byte chunk `chunk` of `row`, XOR-swizzled so that the 16 lanes of a ds_read_b128 group hit 16 distinct
    // 16-byte slots of the 256-byte bank row
    if (BKT == 64) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
    // 64-byte rows: 4 rows per bank row; slot = chunk ^ h(row>>2) with h = (0,2,3,1) keeps every
    // ds_read_b128 lane group ({0-3,12-15,20-27}, ...) on 16 distinct 16-byte slots
    return row * 64 + ((chunk ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3)) << 4);
}

template <int MT, int NT, bool A_F32, int GATHER, int BKT>
__global__ __launch_bounds__(256) void gemm_kernel(const CcvGemm p) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    constexpr int CH = BKT / 8;          // 16-byte chunks per LDS row
    constexpr int RPP = 256 / CH;        // rows staged per pass of the 256 threads
    constexpr int AR = BM / RPP, BR = BN / RPP;  // rows staged per thread
    constexpr int ROWB = BKT * 2;        // bytes per LDS row
    static_assert(AR >= 1 && BR >= 1, "tile too small for this slab depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                  // [2][BM][ROWB]
    unsigned char* sB = smem + 2 * BM * ROWB;  // [2][BN][ROWB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile assignment (bijective for any grid size) ----------------------
    const int tiles_n = (p.N + BN - 1) / BN;
    int split, m0, n0;
    block_tile(p, BM, BN, tiles_n, split, m0, n0);

    // ---- per-thread staging geometry ----------------------------------------------------
    const int chunk = tid % CH;  // 16-byte chunk (8 bf16) within the slab
    const int srow = tid / CH;   // 0..RPP-1
    int a_base[AR], a_y[AR], a_x[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + srow + RPP * i;
        if (GATHER == 0) {
            a_base[i] = (m < p.M) ? m : -1;
            a_y[i] = a_x[i] = 0;
        } else if (GATHER == 1) {
            const int pix = p.out_h * p.out_w;
            const int img = m / pix, rem = m - img * pix;
            const int oy = rem / p.out_w, ox = rem - oy * p.out_w;
            a_base[i] = (m < p.M) ? img * p.src_h * p.src_w : -1;
            a_y[i] = oy * p.stride - (p.no_lead_pad ? 0 : 1);
            a_x[i] = ox * p.stride - (p.no_lead_pad ? 0 : 1);
        } else if (GATHER == 2) {
            a_base[i] = (m < p.M) ? m : -1;
            a_y[i] = (m / p.hw) % p.frames;  // frame index within the clip
            a_x[i] = 0;
        } else {
            a_base[i] = (m < p.M) ? m : -1;
            a_y[i] = a_x[i] = 0;
        }
    }
    const int slabs_per_tap = p.K / BKT;
    const int nslab_all = p.taps * slabs_per_tap;
    const int ldw = p.taps * p.K;
    // split-K: this workgroup sums slabs [s_begin, s_end) and leaves the epilogue to the reduce kernel
    const int s_begin = (p.split_k > 1) ? (int)((long)nslab_all * split / p.split_k) : 0;
    const int s_end = (p.split_k > 1) ? (int)((long)nslab_all * (split + 1) / p.split_k) : nslab_all;

    uint4 ra[AR];            // bf16 path
    float4 rf[A_F32 ? 2 * AR : 1];  // fp32 path (converted when written to LDS)
    uint4 rb[BR];

    auto load_slab = [&](int s) {
        const int tap = s / slabs_per_tap;
        const int kc = (s - tap * slabs_per_tap) * BKT + chunk * 8;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            long src = -1;
            if (a_base[i] >= 0) {
                if (GATHER == 0) {
                    src = a_base[i];
                } else if (GATHER == 1) {
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                    const int vh = p.src_h << p.upsample, vw = p.src_w << p.upsample;
                    if (iy >= 0 && iy < vh && ix >= 0 && ix < vw)
                        src = a_base[i] + (iy >> p.upsample) * p.src_w + (ix >> p.upsample);
                } else if (GATHER == 2) {
                    const int f = a_y[i] + tap - 1;
                    if (f >= 0 && f < p.frames) src = (long)a_base[i] + (long)(tap - 1) * p.hw;
                } else {
                    src = (long)a_base[i] + (long)tap * p.hw;   // segment `tap` of the stacked operand
                }
            }
            if (A_F32) {
                if (src >= 0) {
                    const float4* g = reinterpret_cast<const float4*>(static_cast<const float*>(p.A) + src * p.lda + kc);
                    rf[2 * i] = g[0];
                    rf[2 * i + 1] = g[1];
                } else {
                    rf[2 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    rf[2 * i + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            } else {
                if (src >= 0)
                    ra[i] = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.A) + src * p.lda + kc);
                else
                    ra[i] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        const int kw = s * BKT + chunk * 8;
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int n = n0 + srow + RPP * i;
            if (n < p.N)
                rb[i] = *reinterpret_cast<const uint4*>(p.W + (long)n * ldw + kw);
            else
                rb[i] = make_uint4(0u, 0u, 0u, 0u);
        }
    };

    auto store_slab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int r = srow + RPP * i;
            uint4 v;
            if (A_F32) {
                const float4 lo = rf[2 * i], hi = rf[2 * i + 1];
                v.x = pack_bf16x2(lo.x, lo.y);
                v.y = pack_bf16x2(lo.z, lo.w);
                v.z = pack_bf16x2(hi.x, hi.y);
                v.w = pack_bf16x2(hi.z, hi.w);
            } else {
                v = ra[i];
            }
            *reinterpret_cast<uint4*>(sA + buf * BM * ROWB + lds_off<BKT>(r, chunk)) = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int r = srow + RPP * i;
            *reinterpret_cast<uint4*>(sB + buf * BN * ROWB + lds_off<BKT>(r, chunk)) = rb[i];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15;  // fragment row (m for activations, n for weights)
    const int fg = lane >> 4;  // k group: 8 consecutive k at 8*fg

    load_slab(s_begin);
    store_slab(0);
    __syncthreads();

    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) load_slab(s + 1);
#pragma unroll
        for (int ks = 0; ks < BKT / 32; ++ks) {
            const int c = ks * 4 + fg;
            bf16x8 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = wm * 16 * MT + 16 * i + fr;
                fa[i] = *reinterpret_cast<const bf16x8*>(sA + buf * BM * ROWB + lds_off<BKT>(r, c));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int r = wn * 16 * NT + 16 * j + fr;
                fb[j] = *reinterpret_cast<const bf16x8*>(sB + buf * BN * ROWB + lds_off<BKT>(r, c));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    // weights as the MFMA A operand (rows = n), activations as B (cols = m)
                    acc[i][j] = ccv_mfma_16x16x32(fb[j], fa[i], acc[i][j]);
        }
        if (s + 1 < s_end) store_slab(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n..n+3], m = ..+fr, n = ..+4*fg ---------------------
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * 16 * MT + 16 * i + fr;
        if (m >= p.M) continue;
        if (p.split_k > 1) {  // raw partial sums -> workspace [split][M][N]
            float* wsp = static_cast<float*>(p.ws) + ((long)split * p.M + m) * p.N;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * 16 * NT + 16 * j + 4 * fg;
                if (n < p.N) *reinterpret_cast<float4*>(wsp + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
            continue;
        }
        if (p.geglu) {
#pragma unroll
            for (int j = 0; j < NT; j += 2) {
                const int n = n0 + wn * 16 * NT + 16 * j + 4 * fg;  // value columns; gate at n + 16
                if (n >= p.N) continue;
                const float a_[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                const float g_[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
                epilogue_geglu(p, m, n, a_, g_);
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * 16 * NT + 16 * j + 4 * fg;
            if (n >= p.N) continue;
            float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            epilogue_store(p, m, n, o);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// LDS-DMA variant for bf16 activations: both operand tiles go global -> LDS with global_load_lds_dwordx4
// (no VGPR staging, no ds_write: on gfx950 a register->LDS store of 16 bytes costs ~13 cycles of the LDS
// pipe, which made the register-staged loop LDS-bound).  One wave-instruction writes 1 KiB = 8 rows of the
// 128-byte-row tile linearly (LDS address = wave-uniform base + 16*lane), so the XOR swizzle is applied to
// the per-lane SOURCE address (lane l fetches logical chunk (l&7) ^ swz(row)) and again on the fragment read.
// Conv padding / out-of-range rows fetch from a 16-byte zero line.  Two LDS stages: the DMA of slab s+1 is
// in flight while slab s is multiplied.
// -------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned char g_zero_line[16];

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
// LDS-DMA through buffer descriptors (gemm_dma_kernel / gemm_ring_kernel): the descriptors span DMA_RANGE bytes from the operand's
// base (the host checks that every real row lies below it), DMA_NOWHERE is the per-lane offset of a row that does not exist.
constexpr int DMA_RANGE = 0x7ff00000, DMA_NOWHERE = 0x7ffffff0;

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
// slab s must have landed; `younger` slabs (PER DMA instructions each per wave) may stay in flight
template <int PER, int Y>
struct WaitSlab {
    static __device__ __forceinline__ void run(int younger) {
        if (younger >= Y) wait_vm_barrier<PER * Y>();
        else WaitSlab<PER, Y - 1>::run(younger);
    }
};
template <int PER>
struct WaitSlab<PER, 0> {
    static __device__ __forceinline__ void run(int) { wait_vm_barrier<0>(); }
};


#ifdef CCV_FAMILY_STAMPS
// Diagnostic build (tools/family_stamps.py; never the shipped library): lane 0 of every wave of gemm_dma_kernel's two-stage loop
// stamps s_memtime (low 32 bits) at five points of every slab into a spare 4 KiB of LDS (ds_write by inline assembly: a C++ LDS
// store in a loop with LDS-DMA makes hipcc drain vmcnt) and the stamps are copied out when the tile is done.
constexpr int FAM_STAMP_SLABS = 48, FAM_STAMP_POINTS = 5, FAM_STAMP_WORDS = FAM_STAMP_SLABS * FAM_STAMP_POINTS + 4;
__device__ unsigned int* g_fam_stamps = nullptr;
extern "C" int ccv_debug_family_stamps(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fam_stamps), &buf, sizeof(buf));
}
__device__ __forceinline__ void fam_stamp(uint32_t lds_addr, int lane) {
    const uint32_t t = (uint32_t)__builtin_amdgcn_s_memtime();
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr), "v"(t) : "memory");
}
#define FAM_STAMP(slab, point) do { if ((slab) < FAM_STAMP_SLABS) fam_stamp(stamp_base + 4 * ((slab) * FAM_STAMP_POINTS + (point)), lane); } while (0)
#else
#define FAM_STAMP(slab, point) do { } while (0)
#endif

// ST = LDS stages: 2 = the loop above (DMA of slab s+1 behind the MFMAs of slab s, drained before every barrier); 3 = ring with
// counted waits (two slabs in flight, s_waitcnt vmcnt(N) + raw s_barrier, as gemm_ring_kernel but with whole 128-byte rows): for
// the layers whose time is the sum of their slabs' DMA latencies -- few tiles (<= 1 workgroup per CU anyway) and 10-80 slabs, the
// 8x8 / 4x4-latent linear layers and temporal convolutions: a two-stage loop pays one exposed L2 / HBM round trip per slab.
template <int MT, int NT, int GATHER, bool GN = false, int ST = 2>   // GN: instances whose epilogue is tile_epilogue_gn (their own kernels: it costs
__global__ __launch_bounds__(256) void gemm_dma_kernel(const CcvGemm p) {   // 15-50 VGPRs, which the plain instances must not pay)
#if defined(__HIP_DEVICE_COMPILE__)   // (the buffer-descriptor builtins and their type exist in the device pass only; the host pass needs the stub)
    constexpr int BM = 32 * MT, BN = 32 * NT;
    constexpr int AI = BM / 32, BI = BN / 32;  // DMA wave-instructions per wave and slab (8 rows each)
    static_assert(ST == 2 || ST == 3 || ST == 4, "stages");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                  // [ST][BM][128 B]
    unsigned char* sB = smem + ST * BM * 128;  // [ST][BN][128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler: LDS-DMA destinations (M0) stay scalar
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    int split, m0, n0;
    block_tile(p, BM, BN, tiles_n, split, m0, n0);

    // staging geometry: DMA instruction j of this wave fills tile rows 8*(4j+wave) .. +7
    const int lrow = lane >> 3, lchunk = lane & 7;
    int a_base[AI], a_y[AI], a_x[AI], a_col[AI];
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int r = 8 * (4 * j + wave) + lrow;
        a_col[j] = (lchunk ^ ((r >> 1) & 7)) * 8;
        const int m = m0 + r;
        if (GATHER == 0) {
            a_base[j] = (m < p.M) ? m : -1;
            a_y[j] = a_x[j] = 0;
        } else if (GATHER == 1) {
            const int pix = p.out_h * p.out_w;
            const int img = m / pix, rem = m - img * pix;
            const int oy = rem / p.out_w, ox = rem - oy * p.out_w;
            a_base[j] = (m < p.M) ? img * p.src_h * p.src_w : -1;
            a_y[j] = oy * p.stride - (p.no_lead_pad ? 0 : 1);
            a_x[j] = ox * p.stride - (p.no_lead_pad ? 0 : 1);
        } else if (GATHER == 2) {
            a_base[j] = (m < p.M) ? m : -1;
            a_y[j] = (m / p.hw) % p.frames;
            a_x[j] = 0;
        } else {
            a_base[j] = (m < p.M) ? m : -1;
            a_y[j] = a_x[j] = 0;
        }
    }
    const int ldw = p.taps * p.K;
    const int slabs_per_tap = p.K / BK;
    const int nslab_all = p.taps * slabs_per_tap;
    const int s_begin = (p.split_k > 1) ? (int)((long)nslab_all * split / p.split_k) : 0;
    const int s_end = (p.split_k > 1) ? (int)((long)nslab_all * (split + 1) / p.split_k) : nslab_all;
    // Operands reach LDS through raw buffer descriptors (buffer_load_dwordx4 ... offen lds): a lane's source is base + its own 32-bit
    // byte offset (fixed per tap: the gather is re-evaluated 9 / 3 / 1 times per tile) + ONE scalar offset that walks K, so the
    // steady-state loop has no per-lane address arithmetic at all (the 64-bit pointer per piece it replaces cost two vector
    // instructions per piece and slab); rows that do not exist -- zero padding of a convolution, rows past M or N -- carry an
    // offset past the descriptor's range and the hardware writes zeros for them (tools/probes/buffer_lds_oob_probe.hip).
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, DMA_RANGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.W), 0, DMA_RANGE, 0x00020000);
    int b_off[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int r = 8 * (4 * j + wave) + lrow;
        const int n = n0 + r;
        b_off[j] = (n < p.N) ? (int)(((long)n * ldw + (lchunk ^ ((r >> 1) & 7)) * 8) * 2) : DMA_NOWHERE;
    }
    int w_soff = s_begin * BK * 2;      // scalar byte offsets along K: the weights', and the activations' inside the current tap
    int a_soff = 0;
    int a_off[AI];
    auto set_tap = [&](int tap, int kc) {
#pragma unroll
        for (int j = 0; j < AI; ++j) {
            long src = -1;
            if (a_base[j] >= 0) {
                if (GATHER == 0) {
                    src = a_base[j];
                } else if (GATHER == 1) {
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    const int iy = a_y[j] + ky, ix = a_x[j] + kx;
                    const int vh = p.src_h << p.upsample, vw = p.src_w << p.upsample;
                    if (iy >= 0 && iy < vh && ix >= 0 && ix < vw)
                        src = a_base[j] + (iy >> p.upsample) * p.src_w + (ix >> p.upsample);
                } else if (GATHER == 2) {
                    const int f = a_y[j] + tap - 1;
                    if (f >= 0 && f < p.frames) src = (long)a_base[j] + (long)(tap - 1) * p.hw;
                } else {
                    src = (long)a_base[j] + (long)tap * p.hw;   // segment `tap` of the stacked operand
                }
            }
            a_off[j] = (src >= 0) ? (int)((src * p.lda + a_col[j]) * 2) : DMA_NOWHERE;
        }
        a_soff = kc * 2;
    };
    int tap_cur = s_begin / slabs_per_tap;
    int slab_in_tap = s_begin - tap_cur * slabs_per_tap;
    set_tap(tap_cur, slab_in_tap * BK);

    auto issue = [&](int buf) {  // DMA the next slab (slabs are issued strictly in order)
        if (slab_in_tap == slabs_per_tap) {
            slab_in_tap = 0;
            ++tap_cur;
            set_tap(tap_cur, 0);
        }
#pragma unroll
        for (int j = 0; j < AI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t*)(sA + (buf * BM + 8 * (4 * j + wave)) * 128), 16, a_off[j], a_soff, 0, 0);
#pragma unroll
        for (int j = 0; j < BI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t*)(sB + (buf * BN + 8 * (4 * j + wave)) * 128), 16, b_off[j], w_soff, 0, 0);
        a_soff += BK * 2;
        w_soff += BK * 2;
        ++slab_in_tap;
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;

    auto multiply = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 4 + fg;
            bf16x8 fa[MT], fb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = wm * 16 * MT + 16 * i + fr;
                fa[i] = *reinterpret_cast<const bf16x8*>(sA + buf * BM * 128 + lds_off<64>(r, c));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int r = wn * 16 * NT + 16 * j + fr;
                fb[j] = *reinterpret_cast<const bf16x8*>(sB + buf * BN * 128 + lds_off<64>(r, c));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = ccv_mfma_16x16x32(fb[j], fa[i], acc[i][j]);
        }
    };
    if constexpr (ST == 2) {
#ifdef CCV_FAMILY_STAMPS
        const uint32_t stamp_base = (uint32_t)(uintptr_t)(lptr_t*)(smem + ST * (BM + BN) * 128) + wave * (FAM_STAMP_WORDS * 4);
        FAM_STAMP(0, 0);          // (slab 0, point 0 is overwritten below: the kernel's first stamp goes to the spare words)
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(stamp_base + 4 * (FAM_STAMP_SLABS * FAM_STAMP_POINTS)), "v"((uint32_t)__builtin_amdgcn_s_memtime()) : "memory");
#endif
        issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int s = s_begin; s < s_end; ++s) {
            const int buf = (s - s_begin) & 1;
            FAM_STAMP(s - s_begin, 0);                 // top of the slab
            if (s + 1 < s_end) issue(buf ^ 1);
            FAM_STAMP(s - s_begin, 1);                 // next slab's DMA issued
            multiply(buf);
            FAM_STAMP(s - s_begin, 2);                 // fragment reads done, all MFMAs issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FAM_STAMP(s - s_begin, 3);                 // next slab landed (this wave's pieces)
            __syncthreads();
            FAM_STAMP(s - s_begin, 4);                 // past the barrier
        }
#ifdef CCV_FAMILY_STAMPS
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(stamp_base + 4 * (FAM_STAMP_SLABS * FAM_STAMP_POINTS + 1)), "v"((uint32_t)__builtin_amdgcn_s_memtime()) : "memory");
#endif
    } else {
        // ring: ST-1 slabs in flight.  Top of iteration s: slab s has landed (the younger ones may still fly), every wave is past
        // its reads of slab s-1 (program order + barrier), whose stage therefore takes slab s+ST-1.
#pragma unroll
        for (int k = 0; k < ST - 1; ++k)
            if (s_begin + k < s_end) issue(k);
        int stage = 0;
        for (int s = s_begin; s < s_end; ++s) {
            WaitSlab<AI + BI, ST - 2>::run(s_end - 1 - s);
            if (s + ST - 1 < s_end) issue(stage == 0 ? ST - 1 : stage - 1);
            multiply(stage);
            stage = (stage + 1 == ST) ? 0 : stage + 1;
        }
        __syncthreads();   // (the statistics tail reuses the operand stages)
    }

    if constexpr (GN) {      // statistics-emitting instance (launched only with p.gn_partial set and no split-K): its own epilogue
        tile_epilogue_gn<MT, NT>(p, acc, m0, n0, tiles_n, smem);
        return;
    }
    if (rows_epilogue_ok<MT, NT>(p)) {
        tile_epilogue_rows<MT, NT>(p, acc, m0, n0, smem);
        return;
    }
    const bool wide = wide_bf16_ok(p);
    static_for<0, MT, 1>([&](auto I) __attribute__((always_inline)) {   // (a plain loop this large is not unrolled any more)
        constexpr int i = decltype(I)::value;
        const int m = m0 + wm * 16 * MT + 16 * i + fr;
        if (m >= p.M) return;
        if (p.split_k > 1) {
            float* wsp = static_cast<float*>(p.ws) + ((long)split * p.M + m) * p.N;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * 16 * NT + 16 * j + 4 * fg;
                if (n < p.N) *reinterpret_cast<float4*>(wsp + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
            return;
        }
        if (NT % 2 == 0 && p.geglu) {
            float4 gpre[NT];                  // bias of the row's value / gate fragments, fetched together
            preload_cols<NT>(p.bias, n0 + wn * 16 * NT + 4 * fg, p.N, gpre);
            static_for<0, NT, 4>([&](auto J) __attribute__((always_inline)) {   // (value, gate) fragment pairs, two pairs at a time
                constexpr int j = decltype(J)::value;
                if constexpr (j + 1 < NT) {
                    const int n = n0 + wn * 16 * NT + 16 * j + 4 * fg;
                    if (n < p.N) {
                        const float a_[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        const float g_[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
                        if constexpr (j + 3 < NT) {
                            const float a2[4] = {acc[i][j + 2][0], acc[i][j + 2][1], acc[i][j + 2][2], acc[i][j + 2][3]};
                            const float g2[4] = {acc[i][j + 3][0], acc[i][j + 3][1], acc[i][j + 3][2], acc[i][j + 3][3]};
                            if (wide && n - 4 * fg + 64 <= p.N) {   // both output groups in range: one 16-byte store per lane
                                store_pair_bf16(static_cast<uint16_t*>(p.C) + (long)m * p.ldc, (n >> 5) * 16 + (n & 15),
                                                geglu_value(p, n, a_, g_, &gpre[j]), geglu_value(p, n + 32, a2, g2, &gpre[j + 2]));
                            } else {
                                epilogue_geglu(p, m, n, a_, g_, &gpre[j]);
                                if (n + 32 < p.N) epilogue_geglu(p, m, n + 32, a2, g2, &gpre[j + 2]);
                            }
                        } else {
                            epilogue_geglu(p, m, n, a_, g_, &gpre[j]);
                        }
                    }
                }
            });
            return;
        }
        float4 bpre[NT];                  // this row's bias fragments, fetched together (epilogue_math)
        Bias2Pre<NT> b2pre;
        preload_cols<NT>(p.bias, n0 + wn * 16 * NT + 4 * fg, p.N, bpre);
        b2pre.load(p, m, n0 + wn * 16 * NT + 4 * fg);
        static_for<0, NT, 2>([&](auto J) __attribute__((always_inline)) {   // fragments two at a time
            constexpr int j = decltype(J)::value;
            const int n = n0 + wn * 16 * NT + 16 * j + 4 * fg;
            if (n < p.N) {
                float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if constexpr (j + 1 < NT) {
                    float o1[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
                    if (wide && n - 4 * fg + 32 <= p.N) {   // bf16 out, both fragments in range: one 16-byte store per lane
                        epilogue_store_pair(p, m, n, o, o1, &bpre[j], b2pre.at(j));
                    } else {
                        epilogue_store(p, m, n, o, &bpre[j], b2pre.at(j));
                        if (n + 16 < p.N) epilogue_store(p, m, n + 16, o1, &bpre[j + 1], b2pre.at(j + 1));
                    }
                } else {
                    epilogue_store(p, m, n, o, &bpre[j], b2pre.at(j));
                }
            }
        });
    });
#ifdef CCV_FAMILY_STAMPS
    if constexpr (ST == 2) {
        if (g_fam_stamps != nullptr) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned int* src = reinterpret_cast<const unsigned int*>(smem + ST * (BM + BN) * 128) + wave * FAM_STAMP_WORDS;
            unsigned int* dst = g_fam_stamps + ((long)blockIdx.x * 4 + wave) * FAM_STAMP_WORDS;
            for (int i = lane; i < FAM_STAMP_WORDS; i += 64) dst[i] = src[i];
            if (lane == 0) { dst[FAM_STAMP_SLABS * FAM_STAMP_POINTS + 2] = (unsigned)(s_end - s_begin); dst[FAM_STAMP_SLABS * FAM_STAMP_POINTS + 3] = (unsigned)__builtin_amdgcn_s_memtime(); }   // behind the epilogue's last store
        }
    }
#endif
#endif
}

// -------------------------------------------------------------------------------------------------
// Ring LDS-DMA variant: block tile (32 MT) x (32 NT) with 2 x 2 waves (wave tile 16 MT x 16 NT), 32-deep K
// slabs in an ST-stage LDS ring: ST-1 slabs of DMA stay in flight behind counted s_waitcnt vmcnt + raw
// s_barrier.  Instances: 128x320 / 64x320 / 128x160 / 64x160 -- every channel count of the model is a multiple
// of 160, so N tiles exactly and the dispatcher can pick the shape whose tile count fills the 256 CUs evenly.
// Why wide wave tiles: with 64 x 64 wave tiles the fragment reads alone need 128 B/clk of LDS per workgroup
// (2 workgroups per CU = the whole LDS port), which bounds the loop at ~25 % of the MFMA peak; a 64 x 160 wave
// tile reads (64+160) rows per 40 MFMAs.  Why the deep ring: the small-M layers (8x8 / 4x4 latents) are bound by
// the global->LDS latency of each slab, not by bandwidth; with 3-7 slabs in flight it overlaps.
// -------------------------------------------------------------------------------------------------
constexpr int RING_BK = 32;

template <int MT, int NT, int ST, int GATHER, bool GN = false>
__global__ __launch_bounds__(256, (ST == 2 ? (MT * NT <= 20 ? 3 : 2) : 1)) void gemm_ring_kernel(const CcvGemm p) {
#if defined(__HIP_DEVICE_COMPILE__)   // (as gemm_dma_kernel)
    constexpr int BM = 32 * MT, BN = 32 * NT;
    constexpr int AI = BM / 64;           // A pieces (16 rows of 64 B = one DMA wave-instruction) per wave and slab
    constexpr int BP = BN / 16;           // B pieces per slab in total
    constexpr int BI = (BP + 3) / 4;      // ... per wave; a wave without a piece of its own re-fetches the last one
    constexpr int PER_SLAB = AI + BI;     //     (same bytes to the same LDS slot) so that vmcnt counts stay uniform
    constexpr int STAGE_BYTES = (BM + BN) * RING_BK * 2;
    static_assert((ST & (ST - 1)) == 0 && PER_SLAB * (ST - 2) <= 63, "ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler: LDS-DMA destinations (M0) stay scalar
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = p.N / BN;
    int split, m0, n0;
    block_tile(p, BM, BN, tiles_n, split, m0, n0);

    const int lrow = lane >> 2, lchunk = lane & 3;
    int a_base[AI], a_y[AI], a_x[AI], a_col[AI];
#pragma unroll
    for (int j = 0; j < AI; ++j) {
        const int r = 16 * (4 * j + wave) + lrow;
        a_col[j] = (lchunk ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3)) * 8;
        const int m = m0 + r;
        if (GATHER == 0) {
            a_base[j] = (m < p.M) ? m : -1;
            a_y[j] = a_x[j] = 0;
        } else if (GATHER == 1) {
            const int pix = p.out_h * p.out_w;
            const int img = m / pix, rem = m - img * pix;
            const int oy = rem / p.out_w, ox = rem - oy * p.out_w;
            a_base[j] = (m < p.M) ? img * p.src_h * p.src_w : -1;
            a_y[j] = oy * p.stride - (p.no_lead_pad ? 0 : 1);
            a_x[j] = ox * p.stride - (p.no_lead_pad ? 0 : 1);
        } else if (GATHER == 2) {
            a_base[j] = (m < p.M) ? m : -1;
            a_y[j] = (m / p.hw) % p.frames;
            a_x[j] = 0;
        } else {
            a_base[j] = (m < p.M) ? m : -1;
            a_y[j] = a_x[j] = 0;
        }
    }
    const int ldw = p.taps * p.K;
    const int slabs_per_tap = p.K / RING_BK;
    const int nslab_all = p.taps * slabs_per_tap;
    const int s_begin = (p.split_k > 1) ? (int)((long)nslab_all * split / p.split_k) : 0;
    const int s_end = (p.split_k > 1) ? (int)((long)nslab_all * (split + 1) / p.split_k) : nslab_all;
    // buffer descriptors + per-lane 32-bit offsets + one scalar K offset, as in gemm_dma_kernel
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, DMA_RANGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.W), 0, DMA_RANGE, 0x00020000);
    int b_off[BI];
    int b_piece[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        b_piece[j] = min(4 * j + wave, BP - 1);
        const int r = 16 * b_piece[j] + lrow;
        b_off[j] = (int)(((long)(n0 + r) * ldw + (lchunk ^ ((0x78 >> (2 * ((r >> 2) & 3))) & 3)) * 8) * 2);
    }
    int w_soff = s_begin * RING_BK * 2;
    int a_soff = 0;
    int a_off[AI];
    auto set_tap = [&](int tap, int kc) {   // gather re-evaluated only when the tap changes
#pragma unroll
        for (int j = 0; j < AI; ++j) {
            long src = -1;
            if (a_base[j] >= 0) {
                if (GATHER == 0) {
                    src = a_base[j];
                } else if (GATHER == 1) {
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    const int iy = a_y[j] + ky, ix = a_x[j] + kx;
                    const int vh = p.src_h << p.upsample, vw = p.src_w << p.upsample;
                    if (iy >= 0 && iy < vh && ix >= 0 && ix < vw)
                        src = a_base[j] + (iy >> p.upsample) * p.src_w + (ix >> p.upsample);
                } else if (GATHER == 2) {
                    const int f = a_y[j] + tap - 1;
                    if (f >= 0 && f < p.frames) src = (long)a_base[j] + (long)(tap - 1) * p.hw;
                } else {
                    src = (long)a_base[j] + (long)tap * p.hw;   // segment `tap` of the stacked operand
                }
            }
            a_off[j] = (src >= 0) ? (int)((src * p.lda + a_col[j]) * 2) : DMA_NOWHERE;
        }
        a_soff = kc * 2;
    };
    int tap_cur = s_begin / slabs_per_tap;
    int slab_in_tap = s_begin - tap_cur * slabs_per_tap;
    set_tap(tap_cur, slab_in_tap * RING_BK);

    auto issue = [&](int stage) {  // DMA the next slab into ring slot `stage` (slabs are issued strictly in order)
        if (slab_in_tap == slabs_per_tap) {
            slab_in_tap = 0;
            ++tap_cur;
            set_tap(tap_cur, 0);
        }
        unsigned char* sA = smem + stage * STAGE_BYTES;
        unsigned char* sB = sA + BM * RING_BK * 2;
#pragma unroll
        for (int j = 0; j < AI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lptr_t*)(sA + 16 * (4 * j + wave) * 64), 16, a_off[j], a_soff, 0, 0);
#pragma unroll
        for (int j = 0; j < BI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t*)(sB + 16 * b_piece[j] * 64), 16, b_off[j], w_soff, 0, 0);
        a_soff += RING_BK * 2;
        w_soff += RING_BK * 2;
        ++slab_in_tap;
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;

    // prologue: ST-1 slabs in flight
#pragma unroll
    for (int k = 0; k < ST - 1; ++k)
        if (s_begin + k < s_end) issue(k);

    for (int s = s_begin; s < s_end; ++s) {
        const int stage = (s - s_begin) & (ST - 1);
        WaitSlab<PER_SLAB, ST - 2>::run(s_end - 1 - s);
        // the stage consumed in the previous iteration is free now: every wave passed the barrier after reading it
        if (s + ST - 1 < s_end) issue((stage + ST - 1) & (ST - 1));
        const unsigned char* sA = smem + stage * STAGE_BYTES;
        const unsigned char* sB = sA + BM * RING_BK * 2;
        bf16x8 fa[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(sA + lds_off<32>(wm * 16 * MT + 16 * i + fr, fg));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const bf16x8 fb = *reinterpret_cast<const bf16x8*>(sB + lds_off<32>(wn * 16 * NT + 16 * j + fr, fg));
#pragma unroll
            for (int i = 0; i < MT; ++i)
                acc[i][j] = ccv_mfma_16x16x32(fb, fa[i], acc[i][j]);
        }
    }

    if constexpr (GN) {      // statistics-emitting instance: its own epilogue (the ring has drained: every issued slab was waited for)
        __syncthreads();
        tile_epilogue_gn<MT, NT>(p, acc, m0, n0, tiles_n, smem);
        return;
    }
    const bool wide = wide_bf16_ok(p);
    static_for<0, MT, 1>([&](auto I) __attribute__((always_inline)) {   // (a plain loop this large is not unrolled any more)
        constexpr int i = decltype(I)::value;
        const int m = m0 + wm * 16 * MT + 16 * i + fr;
        if (m >= p.M) return;
        if (p.split_k > 1) {
            float* wsp = static_cast<float*>(p.ws) + ((long)split * p.M + m) * p.N;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + wn * 16 * NT + 16 * j + 4 * fg;
                *reinterpret_cast<float4*>(wsp + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
            return;
        }
        if (NT % 2 == 0 && p.geglu) {   // N is a multiple of the tile width here: every fragment is in range
            float4 gpre[NT];                  // bias of the row's value / gate fragments, fetched together
            preload_cols<NT>(p.bias, n0 + wn * 16 * NT + 4 * fg, p.N, gpre);
            static_for<0, NT, 4>([&](auto J) __attribute__((always_inline)) {   // (value, gate) fragment pairs, two pairs at a time
                constexpr int j = decltype(J)::value;
                if constexpr (j + 1 < NT) {
                    const int n = n0 + wn * 16 * NT + 16 * j + 4 * fg;
                    const float a_[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    const float g_[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
                    if constexpr (j + 3 < NT) {
                        const float a2[4] = {acc[i][j + 2][0], acc[i][j + 2][1], acc[i][j + 2][2], acc[i][j + 2][3]};
                        const float g2[4] = {acc[i][j + 3][0], acc[i][j + 3][1], acc[i][j + 3][2], acc[i][j + 3][3]};
                        if (wide) {
                            store_pair_bf16(static_cast<uint16_t*>(p.C) + (long)m * p.ldc, (n >> 5) * 16 + (n & 15),
                                            geglu_value(p, n, a_, g_, &gpre[j]), geglu_value(p, n + 32, a2, g2, &gpre[j + 2]));
                        } else {
                            epilogue_geglu(p, m, n, a_, g_, &gpre[j]);
                            epilogue_geglu(p, m, n + 32, a2, g2, &gpre[j + 2]);
                        }
                    } else {
                        epilogue_geglu(p, m, n, a_, g_, &gpre[j]);
                    }
                }
            });
            return;
        }
        float4 bpre[NT];                  // this row's bias fragments, fetched together (epilogue_math)
        Bias2Pre<NT> b2pre;
        preload_cols<NT>(p.bias, n0 + wn * 16 * NT + 4 * fg, p.N, bpre);
        b2pre.load(p, m, n0 + wn * 16 * NT + 4 * fg);
        static_for<0, NT, 2>([&](auto J) __attribute__((always_inline)) {   // fragments two at a time
            constexpr int j = decltype(J)::value;
            const int n = n0 + wn * 16 * NT + 16 * j + 4 * fg;
            float o[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if constexpr (j + 1 < NT) {
                float o1[4] = {acc[i][j + 1][0], acc[i][j + 1][1], acc[i][j + 1][2], acc[i][j + 1][3]};
                if (wide) {
                    epilogue_store_pair(p, m, n, o, o1, &bpre[j], b2pre.at(j));
                } else {
                    epilogue_store(p, m, n, o, &bpre[j], b2pre.at(j));
                    epilogue_store(p, m, n + 16, o1, &bpre[j + 1], b2pre.at(j + 1));
                }
            } else {
                epilogue_store(p, m, n, o, &bpre[j], b2pre.at(j));
            }
        });
    });
#endif
}

// split-K second pass: sum the partial slabs and run the epilogue; one thread per 4 output columns
__global__ __launch_bounds__(256) void gemm_splitk_reduce(const CcvGemm p) {
    const int n4 = p.N >> 2;
    const long total = (long)p.M * n4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i % n4) * 4;
        if (p.geglu && (n & 16)) continue;  // gate columns are consumed together with their value columns
        // the partial slabs four at a time with their loads in flight together (rolled, every slab waited for its own load: split_k dependent L2
        // round trips per thread), added in slab order as before; the bias values are requested ahead of the sums
        auto sum = [&](int col, float o[4]) {
            o[0] = o[1] = o[2] = o[3] = 0.f;
            const float* src = static_cast<const float*>(p.ws) + (long)m * p.N + col;
            const long slab = (long)p.M * p.N;
            int sp = 0;
            for (; sp + 4 <= p.split_k; sp += 4) {
                const float4 v0 = *reinterpret_cast<const float4*>(src + sp * slab), v1 = *reinterpret_cast<const float4*>(src + (sp + 1) * slab);
                const float4 v2 = *reinterpret_cast<const float4*>(src + (sp + 2) * slab), v3 = *reinterpret_cast<const float4*>(src + (sp + 3) * slab);
                o[0] += v0.x; o[1] += v0.y; o[2] += v0.z; o[3] += v0.w;
                o[0] += v1.x; o[1] += v1.y; o[2] += v1.z; o[3] += v1.w;
                o[0] += v2.x; o[1] += v2.y; o[2] += v2.z; o[3] += v2.w;
                o[0] += v3.x; o[1] += v3.y; o[2] += v3.z; o[3] += v3.w;
            }
            if (sp + 2 <= p.split_k) {
                const float4 v0 = *reinterpret_cast<const float4*>(src + sp * slab), v1 = *reinterpret_cast<const float4*>(src + (sp + 1) * slab);
                o[0] += v0.x; o[1] += v0.y; o[2] += v0.z; o[3] += v0.w;
                o[0] += v1.x; o[1] += v1.y; o[2] += v1.z; o[3] += v1.w;
                sp += 2;
            }
            if (sp < p.split_k) {
                const float4 v = *reinterpret_cast<const float4*>(src + sp * slab);
                o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
            }
        };
        float o[4];
        if (p.geglu) {
            float4 gpre[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
            if (p.bias) { gpre[0] = *reinterpret_cast<const float4*>(p.bias + n); gpre[1] = *reinterpret_cast<const float4*>(p.bias + n + 16); }
            float g[4];
            sum(n, o);
            sum(n + 16, g);
            epilogue_geglu(p, m, n, o, g, gpre);
        } else {
            float4 bpre[1], b2pre[1];
            preload_cols<1>(p.bias, n, p.N, bpre);
            preload_cols<1>(bias2_row(p, m), n, p.N, b2pre);
            sum(n, o);
            epilogue_store(p, m, n, o, bpre, b2pre);
        }
    }
}

// split-K second pass for outputs that feed a GroupNorm(32) (p.gn_partial): as gemm_splitk_reduce (sum the partial slabs, run the
// epilogue, store) plus the statistics of the values as stored.  One workgroup owns RB consecutive rows x all N columns: thread
// (roff, col) sums the column quad `col` of rows r0 + roff, + R, ...; the block folds its pair sums exactly like gn_stats
// (ccv_norm.hip: fixed order, bitwise reproducible) into slot (r0 - instance start) / RB of the rows' instance.
__global__ __launch_bounds__(1024) void gemm_splitk_reduce_gn(const CcvGemm p, int RB) {
    __shared__ __attribute__((aligned(16))) float gsm[1024 * 4];  // [R][cols] x (sum01, sq01, sum23, sq23)
    const int cols = p.N >> 2;
    const int col = threadIdx.x % cols, roff = threadIdx.x / cols, R = blockDim.x / cols;
    const int r0 = blockIdx.x * RB, n = col * 4;
    const float* ws = static_cast<const float*>(p.ws);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int m = r0 + roff; m < r0 + RB; m += R) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        float4 bpre[1], b2pre[1];                 // bias values requested ahead of the sums (see gemm_splitk_reduce)
        preload_cols<1>(p.bias, n, p.N, bpre);
        preload_cols<1>(bias2_row(p, m), n, p.N, b2pre);
        {
            const float* src = ws + (long)m * p.N + n;
            const long slab = (long)p.M * p.N;
            int sp = 0;
            for (; sp + 4 <= p.split_k; sp += 4) {
                const float4 v0 = *reinterpret_cast<const float4*>(src + sp * slab), v1 = *reinterpret_cast<const float4*>(src + (sp + 1) * slab);
                const float4 v2 = *reinterpret_cast<const float4*>(src + (sp + 2) * slab), v3 = *reinterpret_cast<const float4*>(src + (sp + 3) * slab);
                o[0] += v0.x; o[1] += v0.y; o[2] += v0.z; o[3] += v0.w;
                o[0] += v1.x; o[1] += v1.y; o[2] += v1.z; o[3] += v1.w;
                o[0] += v2.x; o[1] += v2.y; o[2] += v2.z; o[3] += v2.w;
                o[0] += v3.x; o[1] += v3.y; o[2] += v3.z; o[3] += v3.w;
            }
            if (sp + 2 <= p.split_k) {
                const float4 v0 = *reinterpret_cast<const float4*>(src + sp * slab), v1 = *reinterpret_cast<const float4*>(src + (sp + 1) * slab);
                o[0] += v0.x; o[1] += v0.y; o[2] += v0.z; o[3] += v0.w;
                o[0] += v1.x; o[1] += v1.y; o[2] += v1.z; o[3] += v1.w;
                sp += 2;
            }
            if (sp < p.split_k) {
                const float4 v = *reinterpret_cast<const float4*>(src + sp * slab);
                o[0] += v.x; o[1] += v.y; o[2] += v.z; o[3] += v.w;
            }
        }
        epilogue_math(p, m, n, o, true, bpre, b2pre);
        store_rounded(p, m, n, o);
        a0 += o[0] + o[1]; a1 += o[0] * o[0] + o[1] * o[1];
        a2 += o[2] + o[3]; a3 += o[2] * o[2] + o[3] * o[3];
    }
    *reinterpret_cast<float4*>(gsm + (roff * cols + col) * 4) = make_float4(a0, a1, a2, a3);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int g = threadIdx.x >> 1, k = threadIdx.x & 1;
        const int ppg = (p.N >> 5) >> 1;   // channel pairs per group (channels per group is even)
        float a = 0.f;
        for (int i = 0; i < ppg; ++i) {
            const int pr = g * ppg + i;    // pair -> float4 column pr/2, half pr&1
            float t = 0.f;
            for (int ro = 0; ro < R; ++ro) t += gsm[(ro * cols + (pr >> 1)) * 4 + (pr & 1) * 2 + k];
            a += t;
        }
        const int inst = r0 / p.gn_rows, slot = (r0 - inst * p.gn_rows) / RB;
        p.gn_partial[((long)inst * p.gn_slots + slot) * 64 + threadIdx.x] = a;
    }
}

// rows per workgroup of gemm_splitk_reduce_gn: the largest power of two that divides the instance, leaves >= 256 workgroups (or one
// row each) and no more than GN_MAX_PARTS (512) slots per instance; 0 = this problem cannot take the statistics path
inline int reduce_gn_rows(int M, int N, int gn_rows) {
    if (gn_rows <= 0 || M % gn_rows != 0 || N % 64 != 0 || N > 4096) return 0;
    int rb = 1;
    while (rb * 2 <= gn_rows && gn_rows % (rb * 2) == 0 && M / (rb * 2) >= 256) rb *= 2;
    while (gn_rows / rb > 512 && gn_rows % (rb * 2) == 0) rb *= 2;
    return gn_rows / rb <= 512 ? rb : 0;
}

inline int launch_reduce(const CcvGemm& p, hipStream_t st) {
    if (p.gn_partial) {
        const int rb = reduce_gn_rows(p.M, p.N, p.gn_rows);
        const int cols = p.N / 4;
        const int threads = cols * ((256 + cols - 1) / cols);
        hipLaunchKernelGGL(gemm_splitk_reduce_gn, dim3((unsigned)(p.M / rb)), dim3(threads), 0, st, p, rb);
        CCV_LAUNCH_CHECK("ccv_gemm(split-K reduce + GroupNorm statistics)");
        return CCV_OK;
    }
    const long total = (long)p.M * (p.N / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)blocks), dim3(256), 0, st, p);
    CCV_LAUNCH_CHECK("ccv_gemm(split-K reduce)");
    return CCV_OK;
}

template <int MT, int NT, bool A_F32, int GATHER, int BKT>
int launch(const CcvGemm& p, hipStream_t st) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * (p.split_k > 1 ? p.split_k : 1);
    const size_t lds = 2 * (BM + BN) * BKT * 2;
    auto kern = gemm_kernel<MT, NT, A_F32, GATHER, BKT>;
    static bool attr_done = false;  // raise the dynamic-LDS cap once per instantiation
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, st, p);
    CCV_LAUNCH_CHECK("ccv_gemm");
    if (p.split_k > 1) return launch_reduce(p, st);
    return CCV_OK;
}

// Tile shape for a problem: the largest tile that still yields >= ~1 workgroup per CU (256 CUs).
inline int tune_env(const char* name);
inline void choose_tile(const CcvGemm& p, int& mt, int& nt) {
    const int forced = tune_env("CCV_GEMM_FAMTILE");   // tuning aid: 44 / 24 / 42 / 22 / 45 / 25
    if (forced == 44 || forced == 24 || forced == 42 || forced == 22 || forced == 45 || forced == 25) {
        mt = forced / 10; nt = forced % 10;
        if ((nt == 4 && p.N % 128 != 0) || (nt == 2 && p.N % 64 != 0) || (nt == 5 && (p.N % 160 != 0 || p.geglu || p.a_f32))) { mt = 2; nt = 2; }
        return;
    }
    auto tiles = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
    // measured on MI355X (tools/famtile_probe.py): the loop hides its DMA latency only with two or more workgroups per
    // CU in flight, so a tile size is taken once it yields >= 1.5 workgroups per CU (256 CUs); e.g. 8192x640x640:
    // 128x128 (320 tiles) 24.4 us, 64x128 (640 tiles) 20.3 us; 2048x1280x1280: 64x128 22.9 us, 64x64 19.3 us
    // long-K problems (>= 48 slabs of 64) keep the larger tile: split-K supplies their workgroups
    // (2048x1280 with K = 3840 / 5120 in-model: 64x128 + split 2 = 47 / 58 us, 64x64 unsplit = 54 / 67 us)
    const bool long_k = p.taps * (p.K / BK) >= 48;
    // (round 2 took the larger tile at one workgroup per CU whenever a second launch stream was busy: 27.5 -> 28.1 frames/s then; with
    // round 3's cheaper DMA issue and row epilogues the smaller tiles win again with two clips in flight, 31.0 -> 31.6: profiles/r03_ab_switches.txt)
    const long want = tune_env("CCV_GEMM_WANT") > 0 ? tune_env("CCV_GEMM_WANT") : (long_k ? 256 : 384);
    if (p.N % 128 == 0 && tiles(128, 128) >= want) { mt = 4; nt = 4; return; }
    if (p.N % 128 == 0 && tiles(64, 128) >= want) { mt = 2; nt = 4; return; }
    if (p.N % 64 == 0 && tiles(128, 64) >= (want == 256 ? 320 : want)) { mt = 4; nt = 2; return; }
    mt = 2; nt = 2;
}

// Split-K factor for the 128x128-family kernels: long-K problems that cannot fill the chip with output tiles
// alone (the 8x8 / 4x4 latent layers stream 10-60 MB of weights through a few dozen workgroups otherwise).
// Fitted to tools/gemm_tune.py on MI355X: aim at ~2.5 workgroups per CU, keep >= 20 slabs of 64 per split, and
// do not split when the fp32 partials (split * M * N * 4 bytes, written and re-read) outweigh the gain.
inline long split_scale_pct() {   // tuning aid: CCV_GEMM_SPLITSCALE (percent) scales the workgroup counts split-K aims at
    const int v = tune_env("CCV_GEMM_SPLITSCALE");
    return v > 0 ? v : 100;
}
inline int choose_split(const CcvGemm& p) {
    int mt, nt;
    choose_tile(p, mt, nt);
    const long tiles = (long)((p.M + 32 * mt - 1) / (32 * mt)) * ((p.N + 32 * nt - 1) / (32 * nt));
    const int nslab = p.taps * (p.K / BK);
    if ((long)p.M * p.N > (4l << 20) || tiles >= 512) return 1;
    long s = 640 * split_scale_pct() / 100 / tiles;
    if (s > nslab / 20) s = nslab / 20;
    if (s > 16) s = 16;
    return s < 2 ? 1 : (int)s;
}

thread_local int g_plan_stages = 0;     // LDS stages the current plan asks for (0 = family_stages' rule); set by dispatch_tile

// LDS stages of the family kernel for a problem (gemm_dma_kernel: 2 = two-stage loop, 3 / 4 = ring with counted waits).
// CCV_GEMM_ST (tuning aid, with CCV_GEMM_TUNE=1 re-read per call): 2 / 3 force a depth, 0 / unset = the rule below.
inline int family_stages(const CcvGemm& p, int bm, int bn) {
    const int forced = tune_env("CCV_GEMM_ST");
    if (forced >= 2 && forced <= 4) return forced;
    if (g_plan_stages >= 2) return g_plan_stages;
    static const int env = [] { const char* e = getenv("CCV_GEMM_ST"); return e ? atoi(e) : 0; }();
    if (env >= 2 && env <= 4) return env;
    const long tiles = (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * (p.split_k > 1 ? p.split_k : 1);
    const int nslab = p.taps * (p.K / BK) / (p.split_k > 1 ? p.split_k : 1);
    // measured (tools/stage_probe.py, cold operands, profiles/r03_gemm_stages.txt): the ring wins only on the M = 512 layers of the
    // 4x4 latents (-16 ... -40 %: up to ~2 workgroups per CU, each a chain of exposed DMA round trips); from M = 2048 up the third
    // stage costs the second workgroup per CU and loses 10-30 %
    return (p.M <= 1024 && tiles <= 512 && nslab >= 4) ? 3 : 2;
}

template <int MT, int NT, int GATHER, int ST>
int launch_dma_ring(const CcvGemm& p, hipStream_t st) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * (p.split_k > 1 ? p.split_k : 1);
    const size_t lds = (size_t)ST * (BM + BN) * 128;
    auto kern = gemm_dma_kernel<MT, NT, GATHER, false, ST>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, st, p);
    CCV_LAUNCH_CHECK("ccv_gemm(dma ring)");
    if (p.split_k > 1) return launch_reduce(p, st);
    return CCV_OK;
}

template <int MT, int NT, int GATHER>
int launch_dma(const CcvGemm& p, hipStream_t st) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * (p.split_k > 1 ? p.split_k : 1);
    if (!p.gn_partial || p.split_k > 1) {
        const int stages = family_stages(p, BM, BN);
        if constexpr ((BM + BN) * 128 * 3 <= 160 * 1024) {
            if (stages >= 3) return launch_dma_ring<MT, NT, GATHER, 3>(p, st);     // (4 stages measured equal to 3: not instantiated)
        }
    }
#ifdef CCV_FAMILY_STAMPS
    const size_t lds = 2 * (BM + BN) * 128 + 4 * FAM_STAMP_WORDS * 4;
#else
    const size_t lds = 2 * (BM + BN) * 128;
#endif
    if constexpr (gn_dma_tile(MT, NT, GATHER)) {
        if (p.gn_partial && p.split_k <= 1) {   // the statistics-emitting instance (split-K: the reduce kernel emits them)
            auto kern_gn = gemm_dma_kernel<MT, NT, GATHER, true>;
            static bool attr_gn = false;
            if (!attr_gn) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern_gn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_gn = true;
            }
            hipLaunchKernelGGL(kern_gn, dim3(tiles), dim3(256), lds, st, p);
            CCV_LAUNCH_CHECK("ccv_gemm(dma, GroupNorm statistics)");
            return CCV_OK;
        }
    }
    auto kern = gemm_dma_kernel<MT, NT, GATHER>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, st, p);
    CCV_LAUNCH_CHECK("ccv_gemm(dma)");
    if (p.split_k > 1) return launch_reduce(p, st);
    return CCV_OK;
}

template <int MT, int NT, int ST, int GATHER>
int launch_ring(const CcvGemm& p, hipStream_t st) {
    constexpr int BM = 32 * MT, BN = 32 * NT;
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN) * (p.split_k > 1 ? p.split_k : 1);
    const size_t lds = (size_t)ST * (BM + BN) * RING_BK * 2;
    if constexpr (gn_ring_tile(MT, NT, ST, GATHER)) {
        if (p.gn_partial && p.split_k <= 1) {   // the statistics-emitting instance (split-K: the reduce kernel emits them)
            auto kern_gn = gemm_ring_kernel<MT, NT, ST, GATHER, true>;
            static bool attr_gn = false;
            if (!attr_gn) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern_gn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_gn = true;
            }
            hipLaunchKernelGGL(kern_gn, dim3(tiles), dim3(256), lds, st, p);
            CCV_LAUNCH_CHECK("ccv_gemm(ring, GroupNorm statistics)");
            return CCV_OK;
        }
    }
    auto kern = gemm_ring_kernel<MT, NT, ST, GATHER>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, st, p);
    CCV_LAUNCH_CHECK("ccv_gemm(ring)");
    if (p.split_k > 1) return launch_reduce(p, st);
    return CCV_OK;
}

// ---- kernel selection -----------------------------------------------------------------------------------
// ring configurations: block tile and ring depth
struct RingCfg { int bm, bn, st; };
constexpr int N_RING = 8;
const RingCfg kRing[N_RING] = {
    {128, 320, 4},   // 112 KiB LDS, 1 workgroup per CU
    {64, 320, 4},    //  96 KiB
    {128, 160, 4},   //  72 KiB, 2 per CU
    {64, 160, 4},    //  56 KiB, 2 per CU
    {64, 160, 8},    // 112 KiB
    {128, 320, 2},   //  56 KiB, 2 per CU (register budget 256 per lane)
    {128, 160, 2},   //  36 KiB, 3 per CU
    {64, 320, 2},    //  48 KiB, 3 per CU
};

struct Plan {
    int ring;    // index into kRing, -1: 128x128-family kernels (gemm_dma_kernel / gemm_kernel) with the tile choose_tile
                 // picks, FAM_128x160 / FAM_64x160: the family kernel on a 160-column tile
    int split;   // split-K factor (1 = none)
    int famtile = 0;   // ring == -1 only: 0 = the tile choose_tile picks, else 10 MT + NT of the family tile (44 / 24 / 42 / 22 / 45 / 25)
    int stages = 0;    // family kernel: 0 = family_stages' rule, 2 / 3 = LDS stages
};
constexpr int FAM_128x160 = -2, FAM_64x160 = -3;   // (64x160: reachable through CCV_GEMM_FAMTILE=25 only)

inline int tune_env(const char* name) {  // CCV_GEMM_TUNE=1 re-reads the tuning variables on every call (probe tools)
    static const bool live = [] { const char* e = getenv("CCV_GEMM_TUNE"); return e && e[0] == '1'; }();
    if (!live) return -2;
    const char* e = getenv(name);
    return e ? atoi(e) : -2;
}

inline bool ring_fits(const CcvGemm& p, int r) {
    const RingCfg& c = kRing[r];
    if (p.a_f32 || p.N % c.bn != 0 || p.K % RING_BK != 0) return false;
    if (p.geglu && (c.bn / 32) % 2 != 0) return false;   // the GEGLU epilogue pairs 16-column groups inside a wave tile
    return true;
}

// Which kernel runs a problem.  Fitted to the sweep of tools/gemm_tune.py on MI355X (profiles/r01_gemm_tune.txt):
//  * linear layers (taps == 1): the 128x128 family (64-deep slabs, 2 workgroups per CU) wins or ties, except the
//    GEGLU up-projections (N = 8C), which take the 2-stage 128x320 tile (2 workgroups per CU, up to 770 TFLOP/s);
//  * 3x3 / temporal convolutions: long K (>= 256 slabs of 32) on the 2-stage 128x320 tile with split-K up to ~512
//    workgroups (up to 910 TFLOP/s), shorter K on the 128x160 ring tile (with split-K to reach 256-512 workgroups
//    on the 16x16 .. 4x4 latent layers).
// Per-shape plans from the cold-operand sweep (tools/plan_sweep.py, profiles/r03_gemm_plan_sweep.txt: every family tile x LDS
// stages x split-K and every ring tile x split-K per GEMM signature of the CFG-pair forward, weights and activations rotating
// through > 256 MB of copies).  Listed: signatures whose best plan beat the rules below by >= 5 %; K = -1 matches any K.
// CCV_GEMM_TABLE=0 (A/B aid) ignores the table.
struct PlanRow { int M, N, K, taps, gather, res; Plan plan; };     // res: -1 any, 0 without / 1 with a residual operand
inline const Plan* plan_override(const CcvGemm& p) {
    static const PlanRow rows[] = {
        {8192, 1920, 640, 1, 0, -1, {6, 1}},                    // QKV projection, 16x16 latents: 2-deep 128x160 ring      42.2 -> 37.1 us
        {2048, 1280, 1280, 3, 2, 0, {-1, 1, 25, 3}},            // temporal conv, 8x8: 64x160 tile, 3 stages, no split-K     46.1 -> 39.1
        {2048, 1280, 1280, 3, 2, 1, {-1, 4, 45, 2}},            // ... the one that updates the stream                      46.2 -> 40.9
        {32768, 320, 1280, 1, 0, -1, {-1, 1, 45, 2}},           // feed-forward down-projection, 32x32: 128x160 tile        63.4 -> 49.8
        {2048, 1280, 1280, 1, 0, -1, {-1, 1, 42, 3}},           // K = C projections, 8x8: 128x64 tile, 3 stages            23.0 -> 21.2
        {2048, 1280, 2560, 1, 0, -1, {-1, 1, 42, 3}},           // skip convolution 1x1, 8x8                                37.3 -> 30.4
        {8192, 640, 640, 3, 2, 0, {-1, 1, 45, 3}},              // temporal conv, 16x16                                     36.6 -> 34.6
        {8192, 640, 2560, 1, 0, -1, {-1, 1, 45, 3}},            // feed-forward down-projection, 16x16                      56.3 -> 47.7
        {2048, 1280, 5120, 1, 0, -1, {-1, 2, 45, 3}},           // ... 8x8                                                  50.1 -> 47.7
        {8192, 640, -1, 9, 1, -1, {-1, 2, 45, 2}},              // 3x3 convolutions, 16x16 (K = 640 ... 1920): 128x160 x 2  91 -> 74, 111 -> 96, 139 -> 128, 188 -> 177
        {2048, 1280, 1280, 3, 3, -1, {-1, 2, 45, 3}},           // stacked camera projections, 8x8                          48.4 -> 41.2
        {8192, 640, 640, 3, 3, -1, {-1, 1, 45, 3}},             // ... 16x16                                                45.0 -> 39.8
        {512, 1280, -1, 9, 1, -1, {-1, 8, 45, 3}},              // 3x3 convolutions, 4x4: 128x160 tile, 3 stages, split 8    34.8 -> 32.9, 53.9 -> 49.7
        {512, 1280, 5120, 1, 0, -1, {-1, 4, 25, 3}},            // feed-forward down-projection, 4x4                        27.0 -> 23.0
        {8192, 640, 640, 1, 0, 0, {-1, 1, 45, 3}},              // proj_in, 16x16                                           19.3 -> 18.3
    };
    static const bool on = [] { const char* e = getenv("CCV_GEMM_TABLE"); return !(e && e[0] == '0'); }();
    if (!on || p.a_f32 || p.geglu) return nullptr;
    for (const PlanRow& r : rows)
        if (r.M == p.M && r.N == p.N && (r.K < 0 || r.K == p.K) && r.taps == p.taps && r.gather == p.gather &&
            (r.res < 0 || r.res == (p.residual != nullptr ? 1 : 0)))
            return &r.plan;
    return nullptr;
}

inline Plan make_plan(const CcvGemm& p, bool allow_split) {
    static const bool ring_on = [] { const char* e = getenv("CCV_GEMM_WIDE"); return !(e && e[0] == '0'); }();
    const int forced_ring = tune_env("CCV_GEMM_RING"), forced_split = tune_env("CCV_GEMM_SPLIT");
    const int nslab = p.taps * (p.K / RING_BK);
    auto family = [&]() {   // 128x128-family kernels: 64-deep slabs, every split needs at least two of them
        Plan pl{-1, 1};
        if (!allow_split) return pl;
        if (forced_split <= 0) { pl.split = choose_split(p); return pl; }
        const int cap = p.taps * (p.K / BK) / 2;
        pl.split = forced_split > cap ? (cap < 1 ? 1 : cap) : forced_split;
        return pl;
    };
    auto ring = [&](int r, int sp) {   // ring tile r with split sp, clamped to >= 16 slabs per workgroup
        if (forced_split > 0) sp = forced_split;
        if (!allow_split) sp = 1;
        if (sp > nslab / 16) sp = nslab / 16;
        if (sp > 16) sp = 16;
        return Plan{r, sp < 1 ? 1 : sp};
    };
    if (forced_ring == -2 && forced_split <= 0 && tune_env("CCV_GEMM_FAMTILE") == -2 && tune_env("CCV_GEMM_ST") == -2) {
        if (const Plan* ov = plan_override(p)) {
            Plan pl = *ov;
            if (!allow_split) pl.split = 1;
            if (pl.ring < 0 || ring_fits(p, pl.ring)) return pl;
        }
    }
    if (forced_ring == -1 || !ring_on || p.a_f32) return family();
    if (forced_ring >= 0) return ring_fits(p, forced_ring) ? ring(forced_ring, forced_split > 0 ? forced_split : 1) : family();
    const long tiles0 = (long)((p.M + 127) / 128) * (p.N / 320);   // 128x320 tiles (meaningful when N % 320 == 0)
    // GEGLU up-projections (N = 8C): the 128x320 tile at two workgroups per CU once it fills them
    if (p.taps == 1 && p.geglu && ring_fits(p, 5) && tiles0 >= 512) return ring(5, 1);
    // 160-column family tiles (2 stages of 64-deep slabs = whole 128-byte rows per DMA piece, 72 KiB of LDS, two workgroups
    // per CU): fewer operand bytes per flop than 128x64 / 64x64 and faster delivery than the ring tiles' 64-byte rows
    // (tools/probes/l2_lds_probe.hip: 19-25 TB/s against 14-16 TB/s into LDS).  The kernel-level sweep (tools/ring_probe.py)
    // favours them on most 32x32- and 16x16-latent shapes, but IN THE MODEL (kernel-time totals of rocprofv3 traces of
    // bench.py, same box, tools/ab_env.sh CCV_GEMM_F160) only two rules pay: the 3x3 / temporal convolutions at 32x32 latents (-0.5 %
    // of all kernel time) and the long-K layers at 8x8 latents with split-K 4 (-0.7 %); on the 32x32-latent linear
    // layers the GEMMs gain nothing and the kernels that consume their outputs get slower (+17 ms per 77 steps).
    static const bool f160_on = [] { const char* e = getenv("CCV_GEMM_F160"); return !(e && e[0] == '0'); }();   // A/B aid
    if (f160_on && forced_ring == -2 /* none forced */ && !p.geglu && p.N % 160 == 0 && p.K % BK == 0) {
        const long tiles_a = (long)((p.M + 127) / 128) * (p.N / 160);
        const long kk = (long)p.taps * p.K;
        auto fam160 = [&](int code, int sp) {
            const int cap = (int)(kk / BK / 2);
            if (!allow_split || sp > cap) sp = !allow_split ? 1 : (cap < 1 ? 1 : cap);
            return Plan{code, sp < 1 ? 1 : sp};
        };
        if (p.taps > 1 && tiles_a >= 512 && p.N <= 960) return fam160(FAM_128x160, 1);          // convolutions, 32x32 latents
        if (tiles_a >= 128 && tiles_a < 256 && kk >= 5120) return fam160(FAM_128x160, 4);       // 8x8 latents, long K
    }
    if (p.taps == 1) return family();
    if (!ring_fits(p, 2)) return family();
    const long tiles2 = (long)((p.M + 127) / 128) * (p.N / 160);   // 128x160 tiles
    constexpr int r160 = 2;   // 4-stage 128x160 tile (the 2-stage instance, index 6, measured equal in-model)
    if (p.taps == 3) return tiles2 >= 512 ? ring(r160, 1) : family();
    // long-K 3x3 convolutions: 128x320 tiles, two workgroups per CU, split-K up to ~512 workgroups
    if (ring_fits(p, 5) && nslab >= 256 && tiles0 >= 64) return ring(5, (int)(tiles0 >= 512 ? 1 : 512 * split_scale_pct() / 100 / tiles0));
    if (tiles2 >= 512) return ring(r160, 1);
    int sp = (int)(256 * split_scale_pct() / 100 / tiles2);
    if (sp < 1) sp = 1;
    if (tiles2 >= 256 && nslab / sp >= 180) sp *= 2;
    return ring(r160, sp);
}

inline bool dma_enabled() {  // CCV_GEMM_DMA=0 falls back to the register-staged loop (tuning aid)
    static const bool v = [] { const char* e = getenv("CCV_GEMM_DMA"); return !(e && e[0] == '0'); }();
    return v;
}

inline int slab_depth_override() {  // CCV_GEMM_BK=32|64 forces the slab depth (tuning aid)
    static const int v = [] { const char* e = getenv("CCV_GEMM_BK"); return e ? atoi(e) : 0; }();
    return v;
}

// the family tile of a plan: the table's, else what choose_tile picks
inline void plan_tile(const CcvGemm& p, const Plan& pl, int& mt, int& nt) {
    if (pl.ring == -1 && pl.famtile > 0) { mt = pl.famtile / 10; nt = pl.famtile % 10; return; }
    choose_tile(p, mt, nt);
}

template <int GATHER>
int dispatch_ring(const CcvGemm& p, int ring, hipStream_t st) {
    switch (ring) {
        case 0: return launch_ring<4, 10, 4, GATHER>(p, st);
        case 1: return launch_ring<2, 10, 4, GATHER>(p, st);
        case 2: return launch_ring<4, 5, 4, GATHER>(p, st);
        case 3: return launch_ring<2, 5, 4, GATHER>(p, st);
        case 4: return launch_ring<2, 5, 8, GATHER>(p, st);
        case 5: return launch_ring<4, 10, 2, GATHER>(p, st);
        case 6: return launch_ring<4, 5, 2, GATHER>(p, st);
        default: return launch_ring<2, 10, 2, GATHER>(p, st);
    }
}

template <bool A_F32, int GATHER>
int dispatch_tile(const CcvGemm& p, const Plan& pl, hipStream_t st) {
    const int ring = pl.ring;
    int mt, nt;
    plan_tile(p, pl, mt, nt);
    g_plan_stages = pl.stages;
    if (!A_F32 && ring >= 0) return dispatch_ring<GATHER>(p, ring, st);
    if (!A_F32 && ring == FAM_128x160) return launch_dma<4, 5, GATHER>(p, st);
    if (!A_F32 && ring == FAM_64x160) return launch_dma<2, 5, GATHER>(p, st);
    if (!A_F32 && dma_enabled()) {
        // (256-row / 256-column tiles -- 8x4, 4x8, 8x5, 8x8 fragments per wave, one workgroup per CU -- were instantiated for the
        // round-3 sweep and never came within 10 % of the best plan of any shape: this two-barrier loop needs two waves per SIMD)
        if (mt == 4 && nt == 5) return launch_dma<4, 5, GATHER>(p, st);
        if (mt == 2 && nt == 5) return launch_dma<2, 5, GATHER>(p, st);
        if (mt == 4 && nt == 4) return launch_dma<4, 4, GATHER>(p, st);
        if (mt == 2 && nt == 4) return launch_dma<2, 4, GATHER>(p, st);
        if (mt == 4 && nt == 2) return launch_dma<4, 2, GATHER>(p, st);
        return launch_dma<2, 2, GATHER>(p, st);
    }
    int bk = slab_depth_override();
    if (bk != 32 && bk != 64) bk = 64;
    if (bk == 64) {
        if (mt == 4 && nt == 4) return launch<4, 4, A_F32, GATHER, 64>(p, st);
        if (mt == 2 && nt == 4) return launch<2, 4, A_F32, GATHER, 64>(p, st);
        if (mt == 4 && nt == 2) return launch<4, 2, A_F32, GATHER, 64>(p, st);
        return launch<2, 2, A_F32, GATHER, 64>(p, st);
    }
    if (mt == 4 && nt == 4) return launch<4, 4, A_F32, GATHER, 32>(p, st);
    if (mt == 2 && nt == 4) return launch<2, 4, A_F32, GATHER, 32>(p, st);
    if (mt == 4 && nt == 2) return launch<4, 2, A_F32, GATHER, 32>(p, st);
    return launch<2, 2, A_F32, GATHER, 32>(p, st);
}

// -------------------------------------------------------------------------------------------------
// A-stationary kernel for the short-K linear layers of the 32x32-latent blocks (K = 320: QKV / out / GEGLU-up
// projections over M = 32768 token rows).  There the tiled kernels above are bound by neither MFMA nor HBM: with five
// 64-deep slabs per tile every workgroup spends its life in prologue (first DMA round trip), five exposed DMA
// latencies and an epilogue, and re-fetches its 128 activation rows once per N tile (472 MB into LDS for the QKV
// projection, profiles/r01_l2_lds_probe.txt).  Here one 8-wave workgroup per CU owns 128 rows for the WHOLE N range:
//   * its activation tile (128 x K bf16) is staged ONCE through LDS in whole 128-byte lines (borrowing two ring stages) and
//     read into registers (20 x 16 B per lane at K = 320), where it stays;
//   * the weights stream through a 3-deep LDS ring in strips of 64 output columns x K (40 KiB, LDS-DMA, same
//     swizzled 128-byte-row slab image as gemm_dma_kernel), two strips ahead of the MFMAs, behind counted vmcnt waits;
//   * per strip a wave multiplies its 32 rows x 32 columns (2 x 2 accumulators of v_mfma_f32_16x16x32_bf16, weights
//     as the A operand like everywhere in this file) and runs the epilogue for it while the next strips' DMA flies:
//     LDS traffic is the weight fragments only, operand bytes into LDS drop from 472 MB to 157 MB per QKV projection.
// EST = vector stores one wave issues per strip epilogue (counted in vmcnt next to the DMA pieces).
// -------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vm_only() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

enum { AS_BF16_WIDE = 0, AS_BF16 = 1, AS_F32 = 2, AS_GEGLU = 3, AS_F16 = 4 };   // AS_F16: fp16 output (16-byte pair stores), fp16 residual

// LNF (round 3): the activation operand is the fp16 residual stream itself and the LayerNorm in front of the projection
// (attention.py:248-253: norm1 -> attn1.to_q/k/v, norm2 -> attn2.to_q, norm3 -> ff) runs HERE, on the whole K = 320 rows the
// workgroup stages anyway: statistics over the row (two passes over the LDS tile: mean, then centred sum of squares, folded over the
// four lanes that share a row), then (x - mean) rstd gamma + beta rounded to bf16 straight into the fragment registers.  No
// LayerNorm launch, no normalised copy of the stream in memory.
template <int NSLAB, int MODE, bool HAS_BIAS, bool HAS_RES, bool LNF = false>
__global__ __launch_bounds__(512, 2) void gemm_astat_kernel(const CcvGemm p) {
#if defined(__HIP_DEVICE_COMPILE__)   // (buffer-descriptor builtins: device pass only, as gemm_dma_kernel)
    // 8 waves = 4 (rows) x 2 (columns), wave tile 32 rows x 32 columns, two waves per SIMD.  Measured on the way here
    // (profiles/r02_astat_notes.txt): 4 waves of 64 rows, one per SIMD, add the phases up (knock-out: MFMA 9.6 us + stores 7.8 us
    // + weight DMA 5.3 us over a 17.8 us floor for the QKV projection, 36.4 us); 8 waves: 32.5 us; letting waves 4-7 run their
    // epilogue one strip late (under the other half's MFMAs) or holding a whole strip's fragments in registers: slower again.
    // s_memtime stamps: per strip ~3100 cycles (wait + barrier 680, DMA issue + 40 MFMAs 1630, epilogue 800) and a 14k-cycle
    // prologue (160 KiB of activation fragments per CU arrive at ~11 B/clk/CU): the kernel is bound by vector issue (short K:
    // ~6 VALU instructions per output element against 2.5 MFMAs) and that prologue, not by MFMA, LDS or HBM.
    constexpr int BM = 128, BN = 64, MT = 2, NT = 2, KS = 2 * NSLAB, K = 64 * NSLAB, NWAVE = 8;
    constexpr int STAGE = NSLAB * BN * 128;            // bytes of one weight strip in LDS
    constexpr int NST = 3;                             // ring depth
    constexpr int PIECES = NSLAB * BN / 8 / NWAVE;     // DMA wave-instructions per wave and strip (8 rows x 128 B each)
    constexpr int EST = (MODE == AS_BF16_WIDE || MODE == AS_F16) ? MT : MODE == AS_GEGLU ? MT : MT * NT;     // vector stores per strip epilogue
    constexpr int NL = (HAS_BIAS ? NT : 0) + (HAS_RES ? MT * NT : 0);                    // epilogue-operand loads per strip
    static_assert(BN / 8 == NWAVE && PIECES == NSLAB && 2 * PIECES + 2 * EST + NL <= 63, "strip geometry: wave w stages row group w of every slab");
    static_assert(!HAS_RES || MODE == AS_F32 || MODE == AS_F16, "residual needs a stream output mode (fp32 / fp16)");
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    using res_t = typename std::conditional<MODE == AS_F16, u32x2, f32x4>::type;     // one lane's 4 residual values of a fragment
    static_assert(MT == 2 && NT == 2, "operand lists of the counted waits below");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar for the compiler (M0 of the weight DMA)
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int nstrips = p.N / BN;

    // diagnosis (tools/astat_stamps.py passes a buffer in p.ws): wave 0 of every workgroup stamps s_memtime at the phase borders
    unsigned long long* stamps = (p.ws != nullptr && wave == 0 && lane == 0) ? static_cast<unsigned long long*>(p.ws) + (long)blockIdx.x * (2 + 3 * nstrips) : nullptr;
    if (stamps) stamps[0] = __builtin_amdgcn_s_memtime();

    // ---- weight strip DMA: piece q of this wave = slab q, row group `wave` (8 rows x 128 B) ---------------------------
    const int lrow = lane >> 3, lchunk = lane & 7;
    const int wr = 8 * wave + lrow;                     // row of the strip (0..63) this lane fetches
    // weights through a buffer descriptor (as gemm_dma_kernel): the lane's offset inside a strip is fixed, the strip and the slab ride in
    // the scalar offset.  Past the last strip the same number of pieces is issued with an out-of-range PER-LANE offset (the probed form,
    // tools/probes/buffer_lds_oob_probe.hip: the hardware writes zeros into a stage nobody reads any more; the scalar offset stays that of
    // a real strip, so no signed overflow and no reliance on range-checking the scalar part): every iteration then issues exactly
    // PIECES DMA operations, so the counted waits below hold on every path.
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.W), 0, DMA_RANGE, 0x00020000);
    const int w_off = (wr * K + ((lchunk ^ ((wr >> 1) & 7)) << 3)) * 2;
    auto issue = [&](int strip, int stage) {
        const bool real = strip < nstrips;
        const int soff = (real ? strip : 0) * BN * K * 2;
        const int voff = real ? w_off : DMA_NOWHERE;
#pragma unroll
        for (int q = 0; q < PIECES; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lptr_t*)(smem + stage * STAGE + (q * BN + 8 * wave) * 128), 16, voff, soff + q * 128, 0, 0);
    };
    // Strip s lives in ring stage (s + 2) % 3: stages 0 and 1 first hold the ACTIVATION tile, staged by LDS-DMA in whole
    // 128-byte lines (8 rows x 128 B per piece, [slab][128 rows][128 B], same source-side swizzle as the weight image) and read
    // ONCE into registers: 80 KiB per workgroup from L2 instead of 160 KiB of fragment-shaped (16 rows x 64 B) loads issued twice
    // (the two column halves of the wave grid need the same rows).  The first weight strip flies into stage 2 meanwhile.
    issue(0, 2 % NST);
    {
        const uint16_t* A = static_cast<const uint16_t*>(p.A);
        constexpr int APIECES = NSLAB * (BM / 8) / NWAVE;     // 10 at K = 320
        static_assert(NSLAB * BM * 128 <= 2 * STAGE, "the activation tile must fit the two stages it borrows");
#pragma unroll
        for (int q = 0; q < APIECES; ++q) {
            const int pid = wave + NWAVE * q, slab = pid / (BM / 8), rg = pid % (BM / 8);
            const int r = 8 * rg + lrow;
            const uint16_t* src = A + (long)(m0 + r) * p.lda + slab * 64 + ((lchunk ^ ((r >> 1) & 7)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(smem + (slab * BM + 8 * rg) * 128), 16, 0, 0);
        }
    }
    wait_vm_only<0>();
    asm volatile("s_barrier" ::: "memory");
    // ---- this wave's activation fragments: rows 32 wm + 16 i + fr of the tile, k = 32 ks + 8 fg .. + 7 -----------------
    bf16x8 fa[KS][MT];
    if constexpr (LNF) {
        typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
        auto frag = [&](int ks, int i) {       // 8 consecutive k of row 32 wm + 16 i + fr, as stored (fp16)
            const int slab = ks >> 1, c = (ks & 1) * 4 + fg;
            return *reinterpret_cast<const f16x8*>(smem + slab * BM * 128 + lds_off<64>(wm * 16 * MT + 16 * i + fr, c));
        };
        auto row_sum = [](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };   // the 4 lanes (fg) of a row
        float mean[MT], rstd[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float a = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 h = frag(ks, i);
#pragma unroll
                for (int e = 0; e < 8; ++e) a += (float)h[e];
            }
            mean[i] = row_sum(a) * (1.0f / (float)K);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 h = frag(ks, i);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)h[e] - mean[i]; q += d * d; }
            }
            rstd[i] = rsqrtf(row_sum(q) * (1.0f / (float)K) + p.ln_eps);
        }
        static_for<0, KS, 1>([&](auto Q) __attribute__((always_inline)) {
            constexpr int ks = decltype(Q)::value;
            const int k0 = 32 * ks + 8 * fg;
            const float4 g0 = *reinterpret_cast<const float4*>(p.ln_gamma + k0), g1 = *reinterpret_cast<const float4*>(p.ln_gamma + k0 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(p.ln_beta + k0), b1 = *reinterpret_cast<const float4*>(p.ln_beta + k0 + 4);
            const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const f16x8 h = frag(ks, i);
                float nv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) nv[e] = ((float)h[e] - mean[i]) * rstd[i] * gm[e] + bt[e];
                fa[ks][i] = ccv_opnd8(nv[0], nv[1], nv[2], nv[3], nv[4], nv[5], nv[6], nv[7]);
            }
        });
    } else {
        static_for<0, KS, 1>([&](auto Q) __attribute__((always_inline)) {
            constexpr int ks = decltype(Q)::value;
            constexpr int slab = ks >> 1;
            const int c = (ks & 1) * 4 + fg;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                fa[ks][i] = *reinterpret_cast<const bf16x8*>(smem + slab * BM * 128 + lds_off<64>(wm * 16 * MT + 16 * i + fr, c));
        });
    }
    // retire the fragment reads (an lgkmcnt(0) where hipcc can see it), then every wave is done with stages 0 / 1
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(fa[ks][i]));
    asm volatile("s_barrier" ::: "memory");
    issue(1, (1 + 2) % NST);
    if (stamps) stamps[1] = __builtin_amdgcn_s_memtime();

    // Epilogue operands (bias, residual) are inline-asm loads with counted waits of our own: hipcc's waitcnt pass answers mixed
    // pending loads / stores / LDS-DMA with vmcnt(0), which would drain the weight strips in flight.  HAS_BIAS / HAS_RES are
    // compile-time so that the loads are unconditional (uniform operation counts).
    auto load_operands = [&](int strip, f32x4 (&bz)[NT], res_t (&rz)[MT][NT]) {
        const int n = strip * BN + wn * 32 + 4 * fg;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if constexpr (HAS_BIAS) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bz[j]) : "v"(p.bias + n + 16 * j) : "memory");
            else bz[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if constexpr (HAS_RES && MODE == AS_F16)
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rz[i][j])
                                 : "v"(static_cast<const uint16_t*>(p.residual) + (long)(m0 + wm * 16 * MT + 16 * i + fr) * p.ldr + n + 16 * j) : "memory");
                else if constexpr (HAS_RES)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rz[i][j])
                                 : "v"(static_cast<const float*>(p.residual) + (long)(m0 + wm * 16 * MT + 16 * i + fr) * p.ldr + n + 16 * j) : "memory");
                else rz[i][j] = res_t{};
            }
    };
    // alpha, bias, (residual), store: exactly EST vector stores per lane.  WAITN: operations younger than this strip's operand
    // loads that may stay in flight (the "+v" operands order every use of the loaded values behind the wait).
    auto epilogue = [&](auto WAITN, int strip, f32x4 (&acc)[MT][NT], f32x4 (&bz)[NT], res_t (&rz)[MT][NT]) __attribute__((always_inline)) {
        constexpr int waitn = decltype(WAITN)::value;
        if constexpr (HAS_RES)
            asm volatile("s_waitcnt vmcnt(%6)" : "+v"(bz[0]), "+v"(bz[1]), "+v"(rz[0][0]), "+v"(rz[0][1]), "+v"(rz[1][0]), "+v"(rz[1][1]) : "n"(waitn) : "memory");
        else if constexpr (HAS_BIAS)
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bz[0]), "+v"(bz[1]) : "n"(waitn) : "memory");
        const int n = strip * BN + wn * 32 + 4 * fg;   // this lane's columns: n .. n+3 (fragment 0) and n+16 .. n+19 (fragment 1)
        static_for<0, MT, 1>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int m = m0 + wm * 16 * MT + 16 * i + fr;     // M % 128 == 0 (host check): always in range
            float o[NT][4];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                o[j][0] = acc[i][j][0] * p.alpha + bz[j][0]; o[j][1] = acc[i][j][1] * p.alpha + bz[j][1];
                o[j][2] = acc[i][j][2] * p.alpha + bz[j][2]; o[j][3] = acc[i][j][3] * p.alpha + bz[j][3];
            }
            if constexpr (MODE == AS_GEGLU) {             // fragment 0 = value columns, fragment 1 = their gates
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = o[0][r] * gelu_erf_f(o[1][r]);
                const int nc = (n >> 5) * 16 + (n & 15);
                *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.C) + (long)m * p.ldc + nc) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            } else if constexpr (MODE == AS_F32) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float4 v4 = make_float4(o[j][0], o[j][1], o[j][2], o[j][3]);
                    if constexpr (HAS_RES) { v4.x += rz[i][j][0]; v4.y += rz[i][j][1]; v4.z += rz[i][j][2]; v4.w += rz[i][j][3]; }
                    *reinterpret_cast<float4*>(static_cast<float*>(p.C) + (long)m * p.ldc + n + 16 * j) = v4;
                }
            } else if constexpr (MODE == AS_F16) {
                if constexpr (HAS_RES) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const float2 ra = unpack_f16x2(rz[i][j][0]), rb = unpack_f16x2(rz[i][j][1]);
                        o[j][0] += ra.x; o[j][1] += ra.y; o[j][2] += rb.x; o[j][3] += rb.y;
                    }
                }
                store_pair_bf16(static_cast<uint16_t*>(p.C) + (long)m * p.ldc, n, make_uint2(pack_f16x2(o[0][0], o[0][1]), pack_f16x2(o[0][2], o[0][3])),
                                make_uint2(pack_f16x2(o[1][0], o[1][1]), pack_f16x2(o[1][2], o[1][3])));
            } else {
                uint16_t* crow = static_cast<uint16_t*>(p.C) + (long)m * p.ldc;
                const uint2 a2 = make_uint2(pack_bf16x2(o[0][0], o[0][1]), pack_bf16x2(o[0][2], o[0][3]));
                const uint2 b2 = make_uint2(pack_bf16x2(o[1][0], o[1][1]), pack_bf16x2(o[1][2], o[1][3]));
                if constexpr (MODE == AS_BF16_WIDE) {
                    store_pair_bf16(crow, n, a2, b2);
                } else {
                    *reinterpret_cast<uint2*>(crow + n) = a2;
                    *reinterpret_cast<uint2*>(crow + n + 16) = b2;
                }
            }
        });
    };

    // Vector-memory operations of a wave, in issue order (vmcnt counts them in this order), iteration s:
    //     L(s) [NL]   DMA(s+2) [PIECES]   (MFMA s)   <wait L(s): PIECES younger>   stores(s) [EST]
    // Top of iteration s: strip s (DMA issued in iteration s-2) must have landed; younger than it are PIECES (DMA s+1) and the
    // stores (and operand loads) of the min(s, 2) epilogues since; the counts below never exceed the true number of younger
    // operations (a smaller count only waits for more).
    f32x4 acc[MT][NT], bz[NT];
    res_t rz[MT][NT];
    for (int s = 0; s < nstrips; ++s) {
        const int stage = (s + 2) % NST;
        const int ne = s >= 2 ? 2 : s;
        if (ne == 0) wait_vm_only<PIECES>();
        else if (ne == 1) wait_vm_only<PIECES + EST>();
        else wait_vm_only<PIECES + 2 * EST>();
        asm volatile("s_barrier" ::: "memory");
        if (stamps) stamps[2 + 3 * s] = __builtin_amdgcn_s_memtime();
        load_operands(s, bz, rz);
        asm volatile("" ::: "memory");
        // strip s + 2 goes where strip s - 1 was (stage (s + 1) % 3), read during iteration s - 1: every wave is past that (barrier above)
        issue(s + 2, (s + 4) % NST);

#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned char* sB = smem + stage * STAGE;
        bf16x8 fb[2][NT];   // weight fragments, one k-step ahead of the MFMAs
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(sB + lds_off<64>(wn * 32 + 16 * j + fr, fg));
        static_for<0, KS, 1>([&](auto Q) __attribute__((always_inline)) {
            constexpr int ks = decltype(Q)::value;
            if constexpr (ks + 1 < KS) {
                constexpr int slab = (ks + 1) >> 1;
                const int c = ((ks + 1) & 1) * 4 + fg;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    fb[(ks + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(sB + slab * BN * 128 + lds_off<64>(wn * 32 + 16 * j + fr, c));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = ccv_mfma_16x16x32(fb[ks & 1][j], fa[ks][i], acc[i][j]);
        });
        if (stamps) {   // after the MFMAs have produced their results (reading an accumulator waits for them)
            asm volatile("" ::"v"(acc[MT - 1][NT - 1]));
            stamps[3 + 3 * s] = __builtin_amdgcn_s_memtime();
        }
        epilogue(std::integral_constant<int, PIECES>{}, s, acc, bz, rz);
        if (stamps) stamps[4 + 3 * s] = __builtin_amdgcn_s_memtime();
    }
    wait_vm_only<0>();   // the zero-line pieces issued past the last strip target this workgroup's LDS: retire them before it is released
#endif
}

template <int NSLAB, int MODE, bool HAS_BIAS, bool HAS_RES, bool LNF = false>
int launch_astat(const CcvGemm& p, hipStream_t st) {
    constexpr size_t lds = 3 * (size_t)NSLAB * 64 * 128;
    auto kern = gemm_astat_kernel<NSLAB, MODE, HAS_BIAS, HAS_RES, LNF>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.M / 128), dim3(512), lds, st, p);
    CCV_LAUNCH_CHECK("ccv_gemm(a-stationary)");
    return CCV_OK;
}

// The A-stationary kernel takes the problem when one 128-row workgroup per CU covers M with at most a quarter of the chip idle,
// K fits the register-resident activation tile and nothing but a plain linear map is asked for.
constexpr int ASTAT_TILE = -4;   // ccv_gemm_plan's *tile code for it

// -------------------------------------------------------------------------------------------------
// Row-vector kernel: M <= 4 rows (one per sample: the timestep / frame-stride MLPs and the projection of their sum onto all 22
// ResBlocks' embedding inputs, openaimodel3d.py:583-592,140-152 -- five launches per forward).  A 128-row MFMA tile spends 25-33 us
// per launch walking K serially for one useful row; the work is a matrix-vector product bound by reading W once.  One wave per 4
// output columns, lane l takes the 16-byte chunks l, l + 64, ... of the K axis of those 4 weight rows, fp32 partial sums are folded
// across the wave in a fixed order and lanes 0 .. M-1 run the ordinary epilogue for their row.
// -------------------------------------------------------------------------------------------------
constexpr int SKINNY_TILE = -5;   // ccv_gemm_plan's *tile code for it
inline bool skinny_fits(const CcvGemm& p) {
    static const bool on = [] { const char* e = getenv("CCV_GEMM_SKINNY"); return !(e && e[0] == '0'); }();   // A/B aid
    return on && p.M <= 4 && p.gather == 0 && p.taps == 1 && !p.geglu && p.gn_partial == nullptr && p.ln_gamma == nullptr && p.N % 4 == 0 && p.K % 8 == 0;
}

template <bool A_F32>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const CcvGemm p) {
    const int lane = threadIdx.x & 63;
    const int n0 = 4 * (int)((blockIdx.x * 256u + threadIdx.x) >> 6);
    if (n0 >= p.N) return;
    float acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
    const int chunks = p.K >> 3;
    for (int c = lane; c < chunks; c += 64) {
        float a[4][8];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < p.M) {
                if (A_F32) {
                    const float4* ap = reinterpret_cast<const float4*>(static_cast<const float*>(p.A) + (long)m * p.lda + 8 * c);
                    const float4 lo = ap[0], hi = ap[1];
                    const float raw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[m][q] = bf16_to_f32(f32_to_bf16(raw[q]));   // "converted on load" like the tiled kernels
                } else {
                    const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.A) + (long)m * p.lda + 8 * c);
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) ccv_opnd2_to_f32(w[q], a[m][2 * q], a[m][2 * q + 1]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) a[m][q] = 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 v = *reinterpret_cast<const uint4*>(p.W + (long)(n0 + j) * p.K + 8 * c);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            float wf[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) ccv_opnd2_to_f32(w[q], wf[2 * q], wf[2 * q + 1]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[m][j] = fmaf(a[m][q], wf[q], acc[m][j]);
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc[m][j] += __shfl_xor(acc[m][j], off, 64);
#pragma unroll
    for (int m = 0; m < 4; ++m)
        if (lane == m && m < p.M) {
            float o[4] = {acc[m][0], acc[m][1], acc[m][2], acc[m][3]};
            epilogue_store(p, m, n0, o);
        }
}

static int dispatch_skinny(const CcvGemm& p, hipStream_t st) {
    const unsigned waves = (unsigned)(p.N / 4);
    const dim3 grid((waves + 3) / 4), block(256);
    if (p.a_f32) hipLaunchKernelGGL(gemm_skinny_kernel<true>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(gemm_skinny_kernel<false>, grid, block, 0, st, p);
    CCV_LAUNCH_CHECK("ccv_gemm(row vector)");
    return CCV_OK;
}
inline bool astat_fits(const CcvGemm& p) {
    static const bool on = [] { const char* e = getenv("CCV_GEMM_ASTAT"); return !(e && e[0] == '0'); }();   // A/B aid
    if (!on || tune_env("CCV_GEMM_RING") != -2 || tune_env("CCV_GEMM_FAMTILE") != -2 || tune_env("CCV_GEMM_SPLIT") > 0) return false;
    if (p.ln_gamma != nullptr &&      // LayerNorm prologue: instantiated for the projections that follow a LayerNorm (QKV / q: bf16 out, no bias; GEGLU)
        !(p.ln_beta != nullptr && p.residual == nullptr && (p.geglu ? p.bias != nullptr : (p.out_f32 == 0 && p.bias == nullptr && (p.ldc & 7) == 0))))
        return false;
    return p.gather == 0 && p.taps == 1 && !p.a_f32 && p.K == 320 && p.M % 128 == 0 && p.M / 128 >= 192 && p.M / 128 <= 512 &&
           p.N % 64 == 0 && p.N >= 128 && p.act == 0 && p.bias2 == nullptr && (p.out_f32 || p.residual == nullptr) &&
           (p.residual == nullptr || (p.res_f16 != 0) == (p.out_f32 == 2)) && (p.out_f32 != 2 || (p.ldc & 7) == 0) &&
           (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 &&
           (p.residual == nullptr || (reinterpret_cast<uintptr_t>(p.residual) & 15) == 0);
}
template <int MODE>
int dispatch_astat_mode(const CcvGemm& p, hipStream_t st) {
    if constexpr (MODE == AS_F32 || MODE == AS_F16) {
        if (p.residual) return p.bias ? launch_astat<5, MODE, true, true>(p, st) : launch_astat<5, MODE, false, true>(p, st);
    }
    return p.bias ? launch_astat<5, MODE, true, false>(p, st) : launch_astat<5, MODE, false, false>(p, st);
}
inline int dispatch_astat(const CcvGemm& p, hipStream_t st) {
    if (p.ln_gamma) {      // (astat_fits admitted only these two flavours)
        if (p.geglu) return launch_astat<5, AS_GEGLU, true, false, true>(p, st);
        return launch_astat<5, AS_BF16_WIDE, false, false, true>(p, st);
    }
    if (p.geglu) return dispatch_astat_mode<AS_GEGLU>(p, st);
    if (p.out_f32 == 2) return dispatch_astat_mode<AS_F16>(p, st);
    if (p.out_f32) return dispatch_astat_mode<AS_F32>(p, st);
    const bool wide = (p.ldc & 7) == 0;
    return wide ? dispatch_astat_mode<AS_BF16_WIDE>(p, st) : dispatch_astat_mode<AS_BF16>(p, st);
}

}  // namespace

// Which operand an XCD's L2 should read only once (block_tile): the larger one.  Unique bytes: the weights [N, taps K] against the
// source rows of A (a strided convolution reads 4x the output rows, an upsampling one a quarter; every tap re-reads the same rows).
inline int choose_tile_order(const CcvGemm& p) {
    static const int forced = [] { const char* e = getenv("CCV_GEMM_ORDER"); return e ? atoi(e) : -1; }();   // A/B aid: 0 / 1 force an order
    const int live = tune_env("CCV_GEMM_ORDER");
    if (live == 0 || live == 1) return live;
    if (forced == 0 || forced == 1) return forced;
    double src_rows = (double)p.M;
    if (p.gather == 1 && p.out_h > 0 && p.out_w > 0) src_rows = (double)(p.M / (p.out_h * p.out_w)) * p.src_h * p.src_w;
    const double a_bytes = src_rows * p.K * (p.a_f32 ? 4.0 : 2.0) * (p.gather == 3 ? p.taps : 1);
    const double w_bytes = (double)p.N * p.taps * p.K * 2.0;
    // measured (tools/order_probe.py, cold weights): M-fastest pays only where the weights dwarf the activations -- the M = 512 layers
    // of the 4x4 latents, -3 ... -18 %; at M = 2048 (weights 2-20x the activations) it costs 1-7 %, at M = 8192 up to 20 %
    static const int rule = [] { const char* e = getenv("CCV_GEMM_ORDER_RULE"); return e ? atoi(e) : 0; }();   // A/B aid: 1 = wherever the weights are larger
    // with a second launch stream busy (ccv_set_streams_in_flight) the fabric traffic saved counts for more than the few percent a
    // single kernel loses: wherever the weights are larger (in-model A/B on one box: two clips in flight +0.7 %, one clip -0.5 %)
    if (rule == 1 || (rule == 0 && g_streams_in_flight.load() >= 2)) return w_bytes > a_bytes ? 1 : 0;
    return (w_bytes >= 2.0 * a_bytes && p.M <= 1024) ? 1 : 0;
}

inline bool plan_ok(const CcvGemm& p) {
    return p.M > 0 && p.N > 0 && p.K > 0 && p.K % BK == 0 && p.taps > 0;
}

extern "C" int ccv_set_streams_in_flight(int32_t n) {
    return g_streams_in_flight.exchange(n < 1 ? 1 : n);
}

extern "C" int64_t ccv_gemm_ws_bytes(const CcvGemm* pp) {
    if (pp == nullptr || !plan_ok(*pp)) return 0;
    if (astat_fits(*pp) || skinny_fits(*pp)) return 0;
    const Plan pl = make_plan(*pp, true);
    return pl.split > 1 ? (int64_t)pl.split * pp->M * pp->N * (int64_t)sizeof(float) : 0;
}

extern "C" int32_t ccv_gemm_ln_fusable(const CcvGemm* pp) {
    return (pp != nullptr && pp->ln_gamma != nullptr && plan_ok(*pp) && astat_fits(*pp)) ? 1 : 0;
}

extern "C" int ccv_gemm_plan(const CcvGemm* pp, int32_t* tile, int32_t* split) {
    CCV_REQUIRE(pp && tile && split, CCV_EINVAL, "ccv_gemm_plan: null pointer");
    CCV_REQUIRE(plan_ok(*pp), CCV_ESHAPE, "ccv_gemm_plan: bad problem sizes");
    if (skinny_fits(*pp)) {
        *tile = SKINNY_TILE;
        *split = 1;
        return CCV_OK;
    }
    if (astat_fits(*pp)) {
        *tile = ASTAT_TILE;
        *split = 1;
        return CCV_OK;
    }
    const Plan pl = make_plan(*pp, true);
    *tile = pl.ring;
    *split = pl.split;
    return CCV_OK;
}

// Tile the kernel ccv_gemm would run this problem on, when that kernel can emit GroupNorm statistics (gemm_dma_kernel /
// gemm_ring_kernel, bf16 in and out, one pass over K)
static bool gn_tile_dims(const CcvGemm& p, const Plan& pl, int& bm, int& bn) {
    if (pl.ring >= 0) {
        bm = kRing[pl.ring].bm; bn = kRing[pl.ring].bn;
        return gn_ring_tile(bm / 32, bn / 32, kRing[pl.ring].st, p.gather);
    }
    int mt = 0, nt = 0;
    if (pl.ring == FAM_128x160) { mt = 4; nt = 5; }
    else if (pl.ring == FAM_64x160) { mt = 2; nt = 5; }
    else if (!dma_enabled()) return false;
    else plan_tile(p, pl, mt, nt);
    bm = 32 * mt;
    bn = 32 * nt;
    return gn_dma_tile(mt, nt, p.gather);
}

extern "C" int32_t ccv_gemm_gn_slots(const CcvGemm* pp, int32_t rows_per_instance) {
    if (pp == nullptr || rows_per_instance <= 0) return 0;
    const CcvGemm& p = *pp;
    if (!plan_ok(p) || astat_fits(p) || skinny_fits(p) || p.a_f32 || p.geglu || p.act != 0) return 0;
    if (p.N % 64 != 0 || p.M % rows_per_instance != 0) return 0;
    const Plan pl = make_plan(p, true);
    if (pl.split > 1) {      // the reduce kernel emits them: one slot per RB rows of an instance
        const int rb = reduce_gn_rows(p.M, p.N, rows_per_instance);
        return rb > 0 ? rows_per_instance / rb : 0;
    }
    int bm = 0, bn = 0;
    if (!gn_tile_dims(p, pl, bm, bn) || rows_per_instance % bm != 0) return 0;
    const long slots = (long)(rows_per_instance / bm) * ((p.N + bn - 1) / bn);
    return slots <= 512 ? (int32_t)slots : 0;   // ccv_norm.hip: GN_MAX_PARTS
}

extern "C" int ccv_gemm(const CcvGemm* pp, void* stream) {
    CCV_REQUIRE(pp != nullptr, CCV_EINVAL, "ccv_gemm: null params");
    static const bool wide_init = [] {
        const char* e = getenv("CCV_GEMM_WIDE_STORE");
        if (e && e[0] == '0') {
            const int zero = 0;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wide_store), &zero, sizeof(int));
        }
        return true;
    }();
    (void)wide_init;
    CcvGemm p = *pp;
    Plan pl{-1, 1};
    if (plan_ok(p)) {   // split-K only when the caller provided the workspace ccv_gemm_ws_bytes() asks for
        pl = make_plan(p, true);
        if (pl.split > 1 && !(p.ws != nullptr && p.ws_bytes >= (int64_t)pl.split * p.M * p.N * (int64_t)sizeof(float)))
            pl = make_plan(p, false);
    }
    p.split_k = pl.split;
    p.tile_order = choose_tile_order(p);
    (void)pl.ring;
    CCV_REQUIRE(p.A && p.W && p.C, CCV_EINVAL, "ccv_gemm: null A/W/C");
    CCV_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, CCV_EINVAL, "ccv_gemm: non-positive M/N/K (%d,%d,%d)", p.M, p.N, p.K);
    CCV_REQUIRE(p.K % BK == 0, CCV_ESHAPE, "ccv_gemm: K=%d must be a multiple of 64", p.K);
    CCV_REQUIRE(p.N % 16 == 0, CCV_ESHAPE, "ccv_gemm: N=%d must be a multiple of 16", p.N);
    CCV_REQUIRE(p.out_f32 >= 0 && p.out_f32 <= 2 && (p.res_f16 == 0 || p.res_f16 == 1), CCV_EINVAL, "ccv_gemm: out_f32 must be 0 / 1 / 2, res_f16 0 / 1");
    CCV_REQUIRE(!p.geglu || (p.N % 32 == 0 && !p.out_f32 && !p.residual && !p.bias2), CCV_ESHAPE,
                "ccv_gemm: geglu needs N%%32==0, bf16 output, no residual/bias2");
    CCV_REQUIRE(p.lda % 8 == 0 && p.ldc % 4 == 0 && (!p.residual || p.ldr % 4 == 0), CCV_ESHAPE,
                "ccv_gemm: leading dimensions must keep 16-byte row alignment (lda=%d ldc=%d ldr=%d)", p.lda, p.ldc, p.ldr);
    CCV_REQUIRE(p.lda >= p.K, CCV_ESHAPE, "ccv_gemm: lda=%d < K=%d", p.lda, p.K);
    CCV_REQUIRE(!p.bias2 || (p.rows_per_batch > 0 && p.ldb2 >= p.N && p.ldb2 % 4 == 0), CCV_EINVAL,
                "ccv_gemm: bias2 needs rows_per_batch > 0 and ldb2 >= N (multiple of 4)");
    CCV_REQUIRE(!p.gn_partial || (p.gn_rows > 0 && p.gn_slots > 0 && p.gn_slots == ccv_gemm_gn_slots(&p, p.gn_rows) &&
                                  (ccv_gemm_ws_bytes(&p) == 0 || p.split_k > 1)), CCV_EINVAL,
                "ccv_gemm: gn_partial needs gn_rows and gn_slots = ccv_gemm_gn_slots() > 0 for this problem (got rows %d, slots %d)", p.gn_rows, p.gn_slots);
    if (!p.a_f32 && !skinny_fits(p)) {   // the LDS-DMA kernels (family, ring, A-stationary) address both operands with 32-bit byte offsets from
        // their base (buffer descriptors); the register-staged fp32-activation kernel and the skinny kernel keep 64-bit pointers
        long a_rows = p.M;
        if (p.gather == 1 && p.out_h > 0 && p.out_w > 0) a_rows = (long)(p.M / (p.out_h * p.out_w)) * p.src_h * p.src_w;
        else if (p.gather == 3) a_rows = (long)(p.taps - 1) * p.hw + p.M;
        CCV_REQUIRE(a_rows * p.lda * (p.a_f32 ? 4 : 2) < (long)DMA_RANGE && (long)p.N * p.taps * p.K * 2 < (long)DMA_RANGE, CCV_ESHAPE,
                    "ccv_gemm: an operand spans 2 GiB or more (activations %ld rows x %d, weights %d x %d)", a_rows, p.lda, p.N, p.taps * p.K);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    CCV_REQUIRE(p.ln_gamma == nullptr || (plan_ok(p) && astat_fits(p)), CCV_ESHAPE,
                "ccv_gemm: the LayerNorm prologue (ln_gamma) exists in the A-stationary kernel only: ask ccv_gemm_ln_fusable() first");
    if (plan_ok(p) && astat_fits(p)) return dispatch_astat(p, st);
    if (skinny_fits(p)) {
        p.split_k = 1;
        return dispatch_skinny(p, st);
    }
    switch (p.gather) {
        case 0:
            CCV_REQUIRE(p.taps == 1, CCV_EINVAL, "ccv_gemm: linear gather needs taps == 1");
            return p.a_f32 ? dispatch_tile<true, 0>(p, pl, st) : dispatch_tile<false, 0>(p, pl, st);
        case 1:
            CCV_REQUIRE(p.taps == 9, CCV_EINVAL, "ccv_gemm: conv3x3 gather needs taps == 9");
            CCV_REQUIRE(p.out_h > 0 && p.out_w > 0 && p.src_h > 0 && p.src_w > 0 && (p.stride == 1 || p.stride == 2) &&
                            (p.upsample == 0 || p.upsample == 1) && p.M % (p.out_h * p.out_w) == 0,
                        CCV_EINVAL, "ccv_gemm: bad conv geometry");
            return p.a_f32 ? dispatch_tile<true, 1>(p, pl, st) : dispatch_tile<false, 1>(p, pl, st);
        case 2:
            CCV_REQUIRE(p.taps == 3, CCV_EINVAL, "ccv_gemm: tconv3 gather needs taps == 3");
            CCV_REQUIRE(p.frames > 0 && p.hw > 0 && p.M % (p.frames * p.hw) == 0, CCV_EINVAL, "ccv_gemm: bad tconv geometry");
            CCV_REQUIRE(!p.a_f32, CCV_ESHAPE, "ccv_gemm: tconv3 takes bf16 activations");
            return dispatch_tile<false, 2>(p, pl, st);
        case 3:
            // stacked operand: A = [taps][hw rows][lda], out[m] = sum_t A[t][m] . W[:, t*K:(t+1)*K]^T -- several linear maps
            // into the same output (e.g. the three projections a camera-conditioned temporal block adds to its stream,
            // modified_forwards.py:519-529) as ONE GEMM with one epilogue
            CCV_REQUIRE(p.taps >= 1 && p.hw >= p.M, CCV_EINVAL, "ccv_gemm: segment gather needs taps >= 1 and a segment stride (hw) >= M");
            CCV_REQUIRE(!p.a_f32, CCV_ESHAPE, "ccv_gemm: segment gather takes bf16 activations");
            return dispatch_tile<false, 3>(p, pl, st);
        default:
            ccv_set_error("ccv_gemm: unknown gather %d", p.gather);
            return CCV_EINVAL;
    }
}
