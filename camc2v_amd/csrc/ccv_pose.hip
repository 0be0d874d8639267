// Once-per-clip camera feeders: ray / Pluecker embedding, pixel-unshuffle into token rows,
// 2x2 average pooling on token rows and the pose encoder's temporal self-attention over <= 16 frames with arbitrary
// head width (40 / 80 / 160 at the shipped sizes; the d = 64 MFMA attention kernels do not cover those).  All tiny,
// HBM/latency bound; none of this runs inside the DDIM loop.
#include "ccv_common.h"

namespace {

inline dim3 grid1d_pose(int64_t n, int block = 256, int64_t cap = 1 << 20) {
    int64_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

// out [B, 6, V, H, W] fp32; one thread per (b, v, pixel)   (reference model/base.py:112-174)
__global__ void ray_condition_kernel(const float* K, const float* c2w, float* out, int B, int V, int H, int W, int plucker) {
    const int64_t n = (int64_t)B * V * H * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % W), py = (int)((i / W) % H);
        const int v = (int)((i / ((int64_t)W * H)) % V), b = (int)(i / ((int64_t)W * H * V));
        const float* Kp = K + ((int64_t)b * V + v) * 9;
        const float* M = c2w + ((int64_t)b * V + v) * 16;
        const float x = ((float)px + 0.5f - Kp[2]) / Kp[0], y = ((float)py + 0.5f - Kp[5]) / Kp[4];
        const float inv = 1.0f / sqrtf(x * x + y * y + 1.0f);
        const float dx = x * inv, dy = y * inv, dz = inv;
        const float rx = dx * M[0] + dy * M[1] + dz * M[2];
        const float ry = dx * M[4] + dy * M[5] + dz * M[6];
        const float rz = dx * M[8] + dy * M[9] + dz * M[10];
        const float ox = M[3], oy = M[7], oz = M[11];
        float e[6];
        if (plucker) { e[0] = oy * rz - oz * ry; e[1] = oz * rx - ox * rz; e[2] = ox * ry - oy * rx; }
        else { e[0] = ox; e[1] = oy; e[2] = oz; }
        e[3] = rx; e[4] = ry; e[5] = rz;
        const int64_t hw = (int64_t)H * W, pix = (int64_t)py * W + px;
#pragma unroll
        for (int c = 0; c < 6; ++c) out[(((int64_t)b * 6 + c) * V + v) * hw + pix] = e[c];
    }
}

// x [n, c, H, W] fp32 -> rows [(n H/r W/r), c r^2] bf16, channel = ch * r^2 + dy * r + dx (torch.nn.PixelUnshuffle)
__global__ void pixel_unshuffle_rows_kernel(const float* x, uint16_t* y, int n, int c, int H, int W, int r) {
    const int h = H / r, w = W / r, cc = c * r * r;
    const int64_t total = (int64_t)n * h * w * cc;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % cc);
        const int64_t row = i / cc;
        const int ox = (int)(row % w), oy = (int)((row / w) % h), img = (int)(row / ((int64_t)w * h));
        const int ch = k / (r * r), dy = (k / r) % r, dx = k % r;
        y[i] = f32_to_bf16(x[(((int64_t)img * c + ch) * H + oy * r + dy) * W + ox * r + dx]);
    }
}

// rows [(n H W), C] fp32 -> [(n H/2 W/2), C] fp32, mean of the 2x2 window (nn.AvgPool2d(2, 2)); float4 columns
__global__ void avgpool2_rows_kernel(const float4* x, float4* y, int n, int H, int W, int c4) {
    const int h = H >> 1, w = W >> 1;
    const int64_t total = (int64_t)n * h * w * c4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % c4);
        const int64_t row = i / c4;
        const int ox = (int)(row % w), oy = (int)((row / w) % h), img = (int)(row / ((int64_t)w * h));
        const int64_t r0 = ((int64_t)img * H + 2 * oy) * W + 2 * ox;
        const float4 a = x[r0 * c4 + k], b = x[(r0 + 1) * c4 + k], c = x[(r0 + W) * c4 + k], d = x[(r0 + W + 1) * c4 + k];
        y[i] = make_float4(0.25f * ((a.x + b.x) + (c.x + d.x)), 0.25f * ((a.y + b.y) + (c.y + d.y)),
                           0.25f * ((a.z + b.z) + (c.z + d.z)), 0.25f * ((a.w + b.w) + (c.w + d.w)));
    }
}

// Cross-normalisation (model/modules/utils.py:30-45): slice s of x (len_x contiguous fp32) is shifted and scaled to the
// mean / unbiased std of reference slice s / per_ref (len_ref contiguous fp32):
//   y = (x - mean_x) * (std_ref / (std_x + 1e-5)) + mean_ref.
// One workgroup per slice; mean first, then the centred sum of squares (both slices are read twice, the second time from
// cache); reductions run in a fixed order (lane tree, then wave 0..3), so the result is reproducible.
__device__ __forceinline__ float block_sum256(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();                      // sh may still be read by the previous call
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void cross_norm_kernel(const float* x, const float* ref, float* y, int64_t len_x, int64_t len_ref,
                                                         int per_ref, float eps) {
    __shared__ float sh[4];
    const float* xs = x + (int64_t)blockIdx.x * len_x;
    const float* rs = ref + (int64_t)(blockIdx.x / per_ref) * len_ref;
    float* ys = y + (int64_t)blockIdx.x * len_x;
    float a = 0.f, b = 0.f;
    for (int64_t i = threadIdx.x; i < len_x; i += 256) a += xs[i];
    for (int64_t i = threadIdx.x; i < len_ref; i += 256) b += rs[i];
    const float mean_x = block_sum256(a, sh) / (float)len_x;
    const float mean_r = block_sum256(b, sh) / (float)len_ref;
    a = b = 0.f;
    for (int64_t i = threadIdx.x; i < len_x; i += 256) { const float d = xs[i] - mean_x; a += d * d; }
    for (int64_t i = threadIdx.x; i < len_ref; i += 256) { const float d = rs[i] - mean_r; b += d * d; }
    const float std_x = sqrtf(block_sum256(a, sh) / (float)(len_x - 1));       // torch.std: unbiased
    const float std_r = sqrtf(block_sum256(b, sh) / (float)(len_ref - 1));
    const float scale = std_r / (std_x + eps);
    for (int64_t i = threadIdx.x; i < len_x; i += 256) ys[i] = (xs[i] - mean_x) * scale + mean_r;
}

// softmax(q k^T scale) v over T <= 16 tokens with head width D <= 256 (multiple of 8): one wave per (batch, head);
// q / k / v rows staged in LDS, lane (t = lane & 15, g = lane >> 4) scores keys g, g+4, g+8, g+12 for query t, the
// probabilities go through LDS and the lane then accumulates its quarter of the D output columns.
constexpr int SMALL_T = 16, SMALL_D = 256;
__global__ __launch_bounds__(128) void attn_small_kernel(const CcvAttn p, int D) {
    __shared__ __attribute__((aligned(16))) uint16_t sq[2][SMALL_T * SMALL_D];     // 3 x 16 KiB + 2 KiB
    __shared__ __attribute__((aligned(16))) uint16_t sk[2][SMALL_T * SMALL_D];
    __shared__ __attribute__((aligned(16))) uint16_t sv[2][SMALL_T * SMALL_D];
    __shared__ float sp[2][SMALL_T * SMALL_T];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 2 + wave;
    const long items = (long)p.B * p.H;
    if (item >= items) return;               // no workgroup barrier below: waves are independent
    const int head = (int)(item % p.H);
    const long b = item / p.H;
    const long bo = b / p.inner, bi = b % p.inner;
    const int T = p.Lq;
    const uint16_t* qg = p.q + bo * p.q_bso + bi * p.q_bsi + (long)head * D;
    const uint16_t* kg = p.k + bo * p.k_bso + bi * p.k_bsi + (long)head * D;
    const uint16_t* vg = p.v + bo * p.v_bso + bi * p.v_bsi + (long)head * D;
    const int d8 = D >> 3;
    for (int i = lane; i < T * d8; i += 64) {
        const int t = i / d8, c = i - t * d8;
        *reinterpret_cast<uint4*>(&sq[wave][t * D + 8 * c]) = *reinterpret_cast<const uint4*>(qg + (long)t * p.q_ls + 8 * c);
        *reinterpret_cast<uint4*>(&sk[wave][t * D + 8 * c]) = *reinterpret_cast<const uint4*>(kg + (long)t * p.k_ls + 8 * c);
        *reinterpret_cast<uint4*>(&sv[wave][t * D + 8 * c]) = *reinterpret_cast<const uint4*>(vg + (long)t * p.v_ls + 8 * c);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int t = lane & 15, g = lane >> 4;
    float s[4];
    float m = -3.0e38f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = g + 4 * jj;
        float acc = 0.f;
        if (t < T && j < T)
            for (int d = 0; d < D; ++d) acc += bf16_to_f32(sq[wave][t * D + d]) * bf16_to_f32(sk[wave][j * D + d]);
        s[jj] = (t < T && j < T) ? acc * p.scale : -3.0e38f;
        m = fmaxf(m, s[jj]);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = g + 4 * jj;
        s[jj] = (t < T && j < T) ? __expf(s[jj] - m) : 0.f;
        l += s[jj];
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) sp[wave][t * SMALL_T + g + 4 * jj] = s[jj] * inv;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t >= T) return;
    uint16_t* og = p.o + bo * p.o_bso + bi * p.o_bsi + (long)t * p.o_ls + (long)head * D;
    for (int d = g; d < D; d += 4) {          // this lane's quarter of the output columns (interleaved)
        float acc = 0.f;
        for (int j = 0; j < T; ++j) acc += sp[wave][t * SMALL_T + j] * bf16_to_f32(sv[wave][j * D + d]);
        og[d] = f32_to_bf16(acc);
    }
}

// Direct 3x3x3 convolution (padding 1) of a few channels: x [B, Cin, T, H, W] fp32 -> y [B, Cout, T, H, W] fp32 (+ bias), optionally
// added to a per-clip image `add` [B, Cout, H, W] broadcast over T: the zero-initialised latent projection behind the
// context-frame adaptor (model/camcontexti2v.py:81-84, 368-373).  Cin, Cout <= 8; one thread per output voxel.
__global__ void conv3d_small_kernel(const float* x, const float* w, const float* bias, const float* add, float* y, int B, int Cin, int Cout,
                                    int T, int H, int W) {
    const int64_t n = (int64_t)B * T * H * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(i % W), py = (int)((i / W) % H), t = (int)((i / ((int64_t)W * H)) % T), b = (int)(i / ((int64_t)W * H * T));
        float acc[8];
        for (int co = 0; co < Cout; ++co) acc[co] = bias ? bias[co] : 0.f;
        for (int ci = 0; ci < Cin; ++ci)
            for (int dt = 0; dt < 3; ++dt) {
                const int tt = t + dt - 1;
                if (tt < 0 || tt >= T) continue;
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = py + dy - 1;
                    if (yy < 0 || yy >= H) continue;
                    for (int dx = 0; dx < 3; ++dx) {
                        const int xx = px + dx - 1;
                        if (xx < 0 || xx >= W) continue;
                        const float v = x[((((int64_t)b * Cin + ci) * T + tt) * H + yy) * W + xx];
                        for (int co = 0; co < Cout; ++co) acc[co] += v * w[(((co * Cin + ci) * 3 + dt) * 3 + dy) * 3 + dx];
                    }
                }
            }
        for (int co = 0; co < Cout; ++co) {
            float o = acc[co];
            if (add) o += add[(((int64_t)b * Cout + co) * H + py) * W + px];
            y[((((int64_t)b * Cout + co) * T + t) * H + py) * W + px] = o;
        }
    }
}

}  // namespace

extern "C" int ccv_ray_condition(const float* K, const float* c2w, float* out, int32_t B, int32_t V, int32_t H, int32_t W,
                                 int32_t plucker, void* stream) {
    CCV_REQUIRE(K && c2w && out && B > 0 && V > 0 && H > 0 && W > 0, CCV_EINVAL, "ccv_ray_condition: bad args");
    hipLaunchKernelGGL(ray_condition_kernel, grid1d_pose((int64_t)B * V * H * W), dim3(256), 0, static_cast<hipStream_t>(stream), K, c2w, out,
                       B, V, H, W, plucker);
    CCV_LAUNCH_CHECK("ccv_ray_condition");
    return CCV_OK;
}

extern "C" int ccv_pixel_unshuffle_rows(const float* x, uint16_t* y, int32_t n, int32_t c, int32_t H, int32_t W, int32_t r, void* stream) {
    CCV_REQUIRE(x && y && n > 0 && c > 0 && H > 0 && W > 0 && r > 0, CCV_EINVAL, "ccv_pixel_unshuffle_rows: bad args");
    CCV_REQUIRE(H % r == 0 && W % r == 0, CCV_ESHAPE, "ccv_pixel_unshuffle_rows: H and W must be multiples of r");
    hipLaunchKernelGGL(pixel_unshuffle_rows_kernel, grid1d_pose((int64_t)n * c * H * W), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n, c, H, W, r);
    CCV_LAUNCH_CHECK("ccv_pixel_unshuffle_rows");
    return CCV_OK;
}

extern "C" int ccv_avgpool2_rows(const float* x, float* y, int32_t n, int32_t H, int32_t W, int32_t C, void* stream) {
    CCV_REQUIRE(x && y && n > 0 && H > 0 && W > 0 && C > 0, CCV_EINVAL, "ccv_avgpool2_rows: bad args");
    CCV_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, CCV_ESHAPE, "ccv_avgpool2_rows: H, W must be even and C a multiple of 4");
    hipLaunchKernelGGL(avgpool2_rows_kernel, grid1d_pose((int64_t)n * (H / 2) * (W / 2) * (C / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), n, H, W, C / 4);
    CCV_LAUNCH_CHECK("ccv_avgpool2_rows");
    return CCV_OK;
}

extern "C" int ccv_attn_small_fwd(const CcvAttn* pp, int32_t head_dim, void* stream) {
    CCV_REQUIRE(pp != nullptr, CCV_EINVAL, "ccv_attn_small_fwd: null params");
    const CcvAttn& p = *pp;
    CCV_REQUIRE(p.q && p.k && p.v && p.o && p.B > 0 && p.H > 0 && p.inner > 0, CCV_EINVAL, "ccv_attn_small_fwd: bad args");
    CCV_REQUIRE(p.Lq == p.Lk && p.Lq > 0 && p.Lq <= SMALL_T, CCV_ESHAPE, "ccv_attn_small_fwd: self-attention over 1..16 tokens (got %d, %d)", p.Lq, p.Lk);
    CCV_REQUIRE(head_dim > 0 && head_dim <= SMALL_D && head_dim % 8 == 0, CCV_ESHAPE, "ccv_attn_small_fwd: head_dim=%d must be a multiple of 8, <= 256", head_dim);
    CCV_REQUIRE(p.q_ls % 8 == 0 && p.k_ls % 8 == 0 && p.v_ls % 8 == 0 && !p.mask_bits && !p.k2 && !p.kreg, CCV_ESHAPE,
                "ccv_attn_small_fwd: 16-byte aligned token strides, no mask / second context / register tokens");
    const long items = (long)p.B * p.H;
    hipLaunchKernelGGL(attn_small_kernel, dim3((unsigned)((items + 1) / 2)), dim3(128), 0, static_cast<hipStream_t>(stream), p, head_dim);
    CCV_LAUNCH_CHECK("ccv_attn_small_fwd");
    return CCV_OK;
}

extern "C" int ccv_conv3d_small(const float* x, const float* w, const float* bias, const float* add, float* y, int32_t B, int32_t Cin,
                                int32_t Cout, int32_t T, int32_t H, int32_t W, void* stream) {
    CCV_REQUIRE(x && w && y && B > 0 && T > 0 && H > 0 && W > 0, CCV_EINVAL, "ccv_conv3d_small: bad args");
    CCV_REQUIRE(Cin > 0 && Cin <= 8 && Cout > 0 && Cout <= 8, CCV_ESHAPE, "ccv_conv3d_small: 1..8 channels (got %d -> %d)", Cin, Cout);
    hipLaunchKernelGGL(conv3d_small_kernel, grid1d_pose((int64_t)B * T * H * W), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, bias, add, y,
                       B, Cin, Cout, T, H, W);
    CCV_LAUNCH_CHECK("ccv_conv3d_small");
    return CCV_OK;
}

extern "C" int ccv_cross_norm(const float* x, const float* ref, float* y, int32_t n_slices, int64_t len_x, int32_t slices_per_ref,
                              int64_t len_ref, float eps, void* stream) {
    CCV_REQUIRE(x && ref && y && n_slices > 0 && slices_per_ref > 0, CCV_EINVAL, "ccv_cross_norm: bad args");
    CCV_REQUIRE(len_x > 1 && len_ref > 1, CCV_ESHAPE, "ccv_cross_norm: the unbiased std needs slices of at least 2 elements (got %ld, %ld)",
                (long)len_x, (long)len_ref);
    CCV_REQUIRE(n_slices % slices_per_ref == 0, CCV_ESHAPE, "ccv_cross_norm: %d slices do not divide into groups of %d per reference slice",
                n_slices, slices_per_ref);
    hipLaunchKernelGGL(cross_norm_kernel, dim3((unsigned)n_slices), dim3(256), 0, static_cast<hipStream_t>(stream), x, ref, y, len_x, len_ref,
                       slices_per_ref, eps);
    CCV_LAUNCH_CHECK("ccv_cross_norm");
    return CCV_OK;
}
