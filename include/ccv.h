/* ccv.h -- C ABI of libccv_hip.so, the MI355X (gfx950) kernel library behind the
 * CamContextI2V DDIM denoising hot path.
 *
 * Boundary contract (last row):
 *   - flat extern "C" functions, plain pointers + sizes, no torch / C++ types;
 *   - every pointer is a DEVICE pointer unless the name says otherwise; the caller
 *     allocates every input, output and workspace; the library holds no tensors;
 *   - every launch is asynchronous on the `stream` argument (a hipStream_t passed as
 *     void*; NULL = the default stream); nothing synchronises, allocates or copies
 *     synchronously, so every entry point may be captured into a hipGraph;
 *   - return value: 0 = ok, negative = argument/shape error (CCV_E*), positive = the
 *     hipError_t of the failed launch; ccv_last_error() returns a thread-local message;
 *   - stateless and re-entrant: any number of host threads may call in on different streams (the Python host samples two
 *     clips at a time that way).  The one piece of device-side state, the work-queue counters of the persistent sparse
 *     attention kernel, is caller-owned when CcvAttn.queue_counters is given (otherwise a rotating pool of 64 rows);
 *     one process per GPU.
 *
 * Each entry point names the reference call site(s) it replaces (paths relative to
 * /root/reference/CamContextI2V).  The reference has no native code: every function
 * here stands in for a third-party kernel the reference reaches through torch,
 * xformers or F.scaled_dot_product_attention.
 *
 * Layouts: activations are token-major ("NHWC"): a tensor [b, t, h, w, C] is a row-major
 * matrix [rows = b*t*h*w, C].  The residual stream is fp16 (UNet; fp32 is supported too), GEMM/attention operands bf16
 * (raw uint16 bit patterns), statistics and accumulation fp32.
 */
#ifndef CCV_H_
#define CCV_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCV_VERSION 100 /* 0.1.0 */

enum {
    CCV_OK = 0,
    CCV_EINVAL = -1,  /* bad argument (null pointer, negative size, unknown enum) */
    CCV_ESHAPE = -2,  /* shape not supported by the kernel (alignment, head dim) */
};

int ccv_version(void);
/* Scheduling hint (the one process-wide setting of the library): how many independent launch streams the caller keeps busy at
 * a time (1 = default).  With two or more, ccv_gemm's planner takes the larger tile already at one workgroup per CU instead of
 * 1.5 (the other stream's kernels fill the rest of the chip): measured on MI355X +2 % frames/s with two clips in flight, -2 %
 * with one.  Read when a launch is planned, so set it before capturing graphs.  Returns the previous value. */
int ccv_set_streams_in_flight(int32_t n);
const char* ccv_last_error(void);

/* ------------------------------------------------------------------------------------
 * ccv_gemm: C = epilogue( sum_tap gather_tap(A) @ W_tap^T )      bf16 MFMA, fp32 accumulate
 *
 * Replaces: every nn.Linear of the path (lvdm/modules/attention.py:58-62,277,301,348,378,
 * 434,454; model/modules/epipolar.py:55-63; model/camcontexti2v.py:152), nn.Conv2d 3x3 /
 * 1x1 of ResBlock, Downsample, Upsample and the UNet in/out convs
 * (lvdm/modules/networks/openaimodel3d.py:68,96,151-187,386,564), nn.Conv3d (3,1,1) of
 * TemporalConvBlock (:255-266), the time-embedding add (:219-228), the residual adds
 * (:230, attention.py:250-252,320,428) and GEGLU (attention.py:431-438).
 *
 * A  : [src_rows, lda] bf16 or fp32 (a_f32), row-major.  Row m of the implicit im2col
 *      matrix is gathered per tap:
 *        gather 0 (linear)  : source row m, taps = 1
 *        gather 1 (conv3x3) : m = (img, oy, ox); tap (ky,kx) reads source pixel
 *                             (oy*stride+ky-1, ox*stride+kx-1) of image img, zero outside;
 *                             with `upsample` the source image (src_h x src_w) is first
 *                             nearest-upsampled 2x (F.interpolate, openaimodel3d.py:103)
 *        gather 2 (tconv3)  : m = (clip, frame, pixel); tap kt reads frame+kt-1, zero outside
 *        gather 3 (segments): A is `taps` stacked operands [taps][hw rows][lda]; tap t reads row m of segment t.
 *                             Several linear maps into the same output run as ONE GEMM over K' = taps*K with one
 *                             epilogue: x += attn1.to_out(o1) + pluker_projection(n + P) + epipolar.to_out(o2) of a
 *                             camera-conditioned temporal block (model/modules/modified_forwards.py:519-529) reads and
 *                             writes the fp32 stream once instead of three times.
 * W  : [N, taps*K] bf16 row-major, k index = tap*K + c  (prepared once from the checkpoint)
 * out: v = alpha*acc + bias[n] + bias2[(m / rows_per_batch)*ldb2 + n]; v = act(v);
 *      geglu: W rows are interleaved in 16-row blocks (value block, gate block); the output
 *             has N/2 columns = value * gelu_erf(gate);
 *      v += residual[m][n] (fp32, or fp16 with res_f16); store bf16, fp32 or fp16 (out_f32 = 0 / 1 / 2) at C[m*ldc + n].
 * Constraints: K % 64 == 0, N % 16 == 0 (N % 32 for geglu), lda/ldc/ldr % 8 == 0; the activation operand (all rows a gather may
 * touch) and the packed weights each span less than 2 GiB: the LDS-DMA kernels address them with 32-bit byte offsets from their base
 * (buffer descriptors; rows that do not exist -- padding, rows past M / N -- are zero-filled by the hardware's range check).
 * ------------------------------------------------------------------------------------ */
typedef struct CcvGemm {
    const void* A;
    const uint16_t* W;
    void* C;
    const float* bias;     /* [N] or NULL */
    const float* bias2;    /* [M / rows_per_batch, N] or NULL */
    const void* residual;  /* [M, ldr] fp32 (fp16 when res_f16) or NULL (may alias C when the output has the same type) */
    int32_t M, N, K, taps;
    int32_t lda, ldc, ldr, ldb2; /* ldb2: row stride of bias2 (>= N) */
    int32_t a_f32;          /* 0: A is bf16, 1: A is fp32 (converted on load) */
    int32_t gather;         /* 0 linear, 1 conv3x3, 2 tconv3, 3 stacked segments */
    int32_t out_h, out_w;   /* gather 1: output image size */
    int32_t src_h, src_w;   /* gather 1: stored source image size (before upsample) */
    int32_t stride;         /* gather 1: 1 or 2 */
    int32_t upsample;       /* gather 1: 0/1 */
    int32_t no_lead_pad;    /* gather 1: 0 = one pixel of zero padding on every side (padding=1); 1 = zero padding only
                             * after the last row / column: F.pad(x, (0,1,0,1)) + stride-2 conv of the first-stage
                             * encoder's Downsample (lvdm/modules/networks/ae_modules.py:106-110) */
    int32_t frames, hw;     /* gather 2: frames per clip, pixels per frame; gather 3: hw = rows between segments (>= M) */
    int32_t rows_per_batch; /* bias2 row = m / rows_per_batch */
    int32_t act;            /* 0 none, 1 SiLU, 2 GELU(erf), 3 ReLU */
    int32_t geglu;          /* 0/1 */
    int32_t out_f32;        /* 0: C is bf16, 1: C is fp32, 2: C is fp16 */
    float alpha;
    void* ws;               /* optional split-K workspace (ccv_gemm_ws_bytes) or NULL */
    int64_t ws_bytes;
    int32_t split_k;        /* set by the library; callers leave it 0 */
    /* GroupNorm statistics of the OUTPUT, produced in the epilogue (the GroupNorm(32) that follows a ResBlock / temporal
     * convolution / transformer, openaimodel3d.py:175-182,210-236,255-266, attention.py:273,343, lvdm/basics.py:78-91): NULL, or
     * [M / gn_rows][gn_slots][32][2] fp32 = per instance (gn_rows consecutive output rows) and slot (an output tile, or a row band of
     * the split-K reduce pass) the sums and sums of squares of every group's channels as stored (rounded to the output type, residual
     * added), in the layout ccv_groupnorm_apply_parts() reads.  gn_slots must be what ccv_gemm_gn_slots() returns for this problem. */
    float* gn_partial;
    int32_t gn_rows, gn_slots;
    int32_t res_f16;        /* 0: residual is fp32, 1: residual is fp16 (the residual stream's hand-off format between blocks:
                             * what the reference carries under torch.autocast, main/trainer.py:193; arithmetic stays fp32) */
    int32_t tile_order;     /* set by the library; callers leave it 0 (0: an XCD walks output tiles N-fastest, 1: M-fastest) */
    /* LayerNorm prologue: NULL, or gamma / beta [K] fp32 of the LayerNorm in front of this projection (attention.py:248-253): A is then
     * the fp16 residual stream [M, lda] itself and every row is normalised over its K = C channels (fp32 statistics, eps = ln_eps)
     * on its way into the MFMA operand registers.  Only problems ccv_gemm_ln_fusable() admits (the A-stationary kernel: K = 320, the
     * 32x32-latent blocks; bf16 output without bias, or the GEGLU up-projection); for the rest run ccv_layernorm first. */
    const float* ln_gamma;
    const float* ln_beta;
    float ln_eps;
} CcvGemm;
/* Workspace the library would like for this problem (0 = none).  Long-K, few-tile problems (the 4x4 and 8x8
 * latent layers) are split along K over extra workgroups when the workspace is provided; without it the
 * call still succeeds, unsplit. */
int64_t ccv_gemm_ws_bytes(const CcvGemm* p);
int ccv_gemm(const CcvGemm* p, void* stream);
/* Which kernel ccv_gemm would run for this problem when the workspace is provided (introspection for tests and
 * the tuning tools; no device work): *tile = index of the LDS-ring tile configuration (0: 128x320, 1: 64x320,
 * 2: 128x160, 3: 64x160 4-deep, 4: 64x160 8-deep, 5: 128x320 2-deep at two workgroups per CU, 6: 128x160 2-deep, 7: 64x320 2-deep; the last two are tuning
 * candidates the planner does not pick), -1 for the 128x128-family kernels (two stages of 64-deep slabs) on the tile they
 * choose by workgroup count, -2 for that kernel on a 128x160 tile, -4 for the A-stationary kernel (K = 320 linear layers over
 * 24576 .. 65536 rows: one workgroup per CU keeps its 128 activation rows in registers, weights stream in 64-column strips);
 * -5 for the row-vector kernel (M <= 4 rows, linear gather: the timestep / frame-stride MLPs of openaimodel3d.py:583-592 -- a
 * matrix-vector product bound by reading W once, one wave per 4 output columns);
 * *split = split-K factor. */
int ccv_gemm_plan(const CcvGemm* p, int32_t* tile, int32_t* split);
/* 1 when ccv_gemm can run this problem with its LayerNorm prologue (p->ln_gamma set), 0 when the caller must normalise first. */
int32_t ccv_gemm_ln_fusable(const CcvGemm* p);
/* Slots per instance the epilogue statistics of this problem would take (see gn_partial), 0 when the kernel ccv_gemm would run
 * cannot produce them (GEGLU / activation epilogues, fp32 activations, the A-stationary kernel, tiles without a statistics
 * instance, tile rows not dividing the instance, more than 512 slots): the caller then runs ccv_groupnorm as usual. */
int32_t ccv_gemm_gn_slots(const CcvGemm* p, int32_t rows_per_instance);

/* ------------------------------------------------------------------------------------
 * ccv_ff_fused: one launch for a whole gated feed-forward block of a transformer,
 *     out = x + Linear_2( value * gelu_erf(gate) ) ,  [value | gate] = Linear_1( LayerNorm(x) )
 * Replaces `x = self.ff(self.norm3(x)) + x` (lvdm/modules/attention.py:253) = nn.LayerNorm (:234) + GEGLU (:431-438) +
 * FeedForward (:441-458) -- today ccv_layernorm + two ccv_gemm launches with the [M, 4C] bf16 hidden tensor written to and
 * re-read from HBM in between (84 MB each way per block at 32x32 latents).  Here one 4-wave workgroup owns 128 token rows:
 * the rows are normalised once into MFMA operand registers, the hidden units are produced 32 at a time (v_mfma_f32_32x32x16_bf16
 * against a 64-column strip of W1 streamed through LDS), gated in registers, and -- the accumulator layout of one product being the
 * operand layout of the next -- multiplied straight into the resident [128, C] output accumulators against the matching
 * 32-column chunk of W2: the hidden activation never exists in memory, not even in LDS.
 *   x      : [M, ldx] fp16 residual stream (LayerNorm input and residual), M % 128 == 0
 *   w1, b1 : [8C, C] bf16 / [8C] fp32, value / gate rows interleaved in 16-row blocks (the layout ccv_gemm's geglu takes)
 *   w2p    : [C, 4C] bf16 with the columns of every 16-column group in the order 0-3, 8-11, 4-7, 12-15 (the k order in which a
 *            32x32 accumulator hands its rows to the next MFMA); b2 [C] fp32
 *   out    : [M, ldo] fp16 (out_kind 2; may alias x) or bf16 (out_kind 0)
 * Only C = 320 (the 32x32-latent blocks) is built: ccv_ff_fusable() says whether a problem is taken (0: run the three launches).
 * ------------------------------------------------------------------------------------ */
typedef struct CcvFF {
    const void* x;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    const uint16_t* w1; const float* b1;
    const uint16_t* w2p; const float* b2;
    void* out;
    int32_t M, C, ldx, ldo;
    int32_t out_kind;       /* 0: bf16, 2: fp16 */
} CcvFF;
int32_t ccv_ff_fusable(const CcvFF* p);
int ccv_ff_fused(const CcvFF* p, void* stream);

/* ------------------------------------------------------------------------------------
 * ccv_attn_fwd: O = softmax(Q K^T * scale [+ mask]) V, head dim 64, bf16 MFMA, fp32 online
 * softmax; optionally a second key/value set attended separately and added with a gate.
 *
 * Replaces: xformers.ops.memory_efficient_attention (lvdm/modules/attention.py:177,189 incl.
 * the gated image branch :205-209), the einsum attention (:105-129) used by the temporal
 * transformers, and F.scaled_dot_product_attention with the boolean epipolar mask and the
 * always-visible register tokens (model/modules/epipolar.py:86-99).
 *
 * Addressing (elements): X(batch i, token l, head h, d) =
 *     X + (i / inner)*x_bso + (i % inner)*x_bsi + l*x_ls + h*64 + d
 * so that a "batch" can be a frame of a clip (spatial), a pixel of a clip (temporal, token
 * stride = pixels*C) or a clip (epipolar), and K/V shared by all frames use x_bsi = 0.
 *
 * mask_bits : NULL or [mask_nb, Lq, mask_words] uint32; bit j of word w set => key 32w+j visible.
 * tile_flags: NULL or [mask_nb, ceil(Lq/128), ceil(Lk/64)] uint8; 0 => no visible key in that
 *             (128 query x 64 key) tile (the tile is skipped; exact).
 * kreg/vreg : NULL or [nreg, H*64] bf16 register-token keys/values, visible to every query.
 * k2/v2     : NULL or second context (image tokens), length Lk2:  O += gate2 * attn(Q,K2,V2).
 * ------------------------------------------------------------------------------------ */
typedef struct CcvAttn {
    const uint16_t* q; const uint16_t* k; const uint16_t* v; uint16_t* o;
    int64_t q_bso, q_bsi, q_ls;
    int64_t k_bso, k_bsi, k_ls;
    int64_t v_bso, v_bsi, v_ls;
    int64_t o_bso, o_bsi, o_ls;
    int32_t B, inner, H, Lq, Lk;
    float scale;
    const uint16_t* k2; const uint16_t* v2;
    int64_t k2_bso, k2_bsi, k2_ls;
    int64_t v2_bso, v2_bsi, v2_ls;
    int32_t Lk2;
    float gate2;
    const uint32_t* mask_bits; int64_t mask_bs; int32_t mask_words;
    int32_t mask_nb;  /* masks exist for mask_nb batches; batch i uses mask i % mask_nb (CFG halves share) */
    const uint8_t* tile_flags; int64_t flags_bs; int32_t flags_ktiles;
    const uint32_t* wave_bits; int64_t wave_bs; int32_t wave_words; /* NULL or [mask_nb, ceil(Lq/64), wave_words] uint32: bit j of
                              * word w set => the 64-query group needs key block 32*(32w+j) .. +31 (any visible key) */
    const int32_t* group_order; int64_t order_bs; /* NULL or [mask_nb, ceil(Lq/64)] int32 from ccv_attn_group_order: the 64-query
                              * groups by decreasing number of needed key blocks; the persistent sparse kernel hands them
                              * out in this order (longest first) so that the last waves to finish hold the cheapest groups */
    const uint16_t* kreg; const uint16_t* vreg; int32_t nreg;
    int32_t perm_hw, perm_w; /* token order of q/k/v/o rows and of the mask: 0 = as stored; otherwise the kernel walks each
                              * frame (perm_hw tokens, perm_w wide) in 4x8-pixel patches, the patches in 2x2 quads when the frame
                              * has even numbers of patch rows and columns, else row-major (index -> row map: ccv_patch_row);
                              * the mask must have been built with the same values */
    int32_t variant;  /* 0: default (LDS-DMA kernel, 64 queries per wave; unmasked two-context calls run both softmaxes in it);
                         1: first-generation kernel, V^T via ds_read_b64_tr_b16; 2: same, V transposed while staging;
                         3: as 0 but always a persistent sparse kernel when wave_bits is given (0 picks it from 1024
                            64-query groups upwards and the tiled masked kernel below that);
                         4 / 5: as 3, the workgroup-shared sparse kernel with 8 / 4 waves (256 / 128 queries) per workgroup;
                         6: as 3, the per-wave sparse kernel (every wave streams its own K / V blocks);
                         7 / 8: as 0, the unmasked tiled kernels with 32 / 64 queries per wave (four / two workgroups per CU) whatever the
                            library's default is (bit-identical results; tests) */
    uint32_t* queue_counters; /* NULL or 8 uint32 that the caller ZEROED on `stream` before the call: the work-queue counters of
                              * the persistent sparse kernel (wave_bits path).  With caller-owned counters the call keeps no
                              * state in the library, so launches may overlap freely on different streams (two clips in
                              * flight, two graphs replayed concurrently).  NULL: a row of the library's rotating pool of 64
                              * counter rows, reset by a one-block launch in front of the kernel. */
    const int32_t* wg_order; int64_t wg_order_bs; int32_t wg_merge; /* NULL or [mask_nb, ceil(ceil(Lq/64) / wg_merge)] int32 from
                              * ccv_attn_group_order_merged: the items of the workgroup-shared sparse kernel (wg_merge consecutive
                              * 64-query groups whose K / V blocks one workgroup stages once: 4 for its 8-wave form, 2 for the
                              * 4-wave form) by decreasing size of the union of their needed key blocks.  NULL or another
                              * wg_merge than the kernel's: items are handed out in index order (same result, longer tail). */
    void* split_ws; int64_t split_ws_bytes; /* NULL or >= ccv_attn_split_ws_bytes(p) bytes whose first 64 KiB (the counters: the whole
                              * buffer may simply be zeroed) the caller ZEROED once before its first use on `stream`: workspace of the
                              * workgroup-shared sparse kernel's key-split items (the shortest items come in 2-4 parts so that the
                              * persistent workgroups finish together; the part that finishes last merges them and resets its counter, so
                              * one buffer serves every later call on the same stream).  NULL / too small: no item is split. */
    int32_t split_all_parts; /* 0: only the queue's tail is split (launches with more items than resident workgroups).  2 .. 4: a launch
                              * that leaves workgroup slots empty may also split EVERY item into up to that many parts (measured
                              * slower at the benchmark's sizes; kept for tests and A/B runs) */
} CcvAttn;
int ccv_attn_fwd(const CcvAttn* p, void* stream);
/* Bytes of CcvAttn.split_ws this call would use on the current device (0: the call does not take the workgroup-shared sparse kernel, or
 * its items divide well enough over the chip).  Replaces nothing in the reference (scheduling aid of epipolar.py:75-102's attention). */
int64_t ccv_attn_split_ws_bytes(const CcvAttn* p);
/* Self-attention over Lq = Lk <= 16 tokens with an arbitrary head width (multiple of 8, <= 256): the temporal blocks of
 * CameraPoseEncoder (model/modules/camera_pose_encoder.py:15-158; heads of 40 / 80 / 160 channels).  Uses q/k/v/o with
 * their strides, B, inner, H, Lq, scale of CcvAttn; the head h of a token starts at column h * head_dim.  No masks. */
int ccv_attn_small_fwd(const CcvAttn* p, int32_t head_dim, void* stream);

/* ------------------------------------------------------------------------------------
 * GroupNorm(32 groups) (+SiLU) over token-major activations, fp32 statistics.
 * Replaces GroupNormSpecific / nn.GroupNorm + nn.SiLU (lvdm/basics.py:78-91,
 * openaimodel3d.py:151-153,175-177,255-266,561-563; attention.py:273,343).
 *   x  : [instances*rows_per_instance, C]; x_f32 = element kind: 0 bf16, 1 fp32, 2 fp16 (the residual stream's hand-off format)
 *   y  : same shape, bf16
 *   ws : workspace, ccv_groupnorm_ws_bytes(instances, C) bytes
 * An instance is the set of rows that share statistics: one frame (h*w rows) for 4-D
 * inputs, one clip (t*h*w rows) for the 5-D inputs of TemporalConvBlock/TemporalTransformer.
 * Constraint: C % 64 == 0, C <= 4096.
 * ------------------------------------------------------------------------------------ */
int64_t ccv_groupnorm_ws_bytes(int32_t instances, int32_t C);
int ccv_groupnorm(const void* x, int32_t x_f32, uint16_t* y, const float* gamma, const float* beta,
                  int32_t instances, int32_t rows_per_instance, int32_t C, float eps, int32_t silu,
                  void* ws, void* stream);

/* The two halves of ccv_groupnorm as separate calls, for statistics that span more rows than this process holds -- a clip whose
 * frames are sharded over GPUs (GroupNorm over (t, h, w) in TemporalConvBlock / TemporalTransformer, openaimodel3d.py:255-266,
 * attention.py:343): ccv_groupnorm_stats fills ws (the ccv_groupnorm workspace) with per-chunk (sum, sum of squares) partials
 * laid out [instances][ccv_groupnorm_chunks(...)][32 groups][2]; the caller reduces them over chunks and ranks, writes the totals
 * into chunk 0 (zeros elsewhere) and calls ccv_groupnorm_apply with inv_count = 1 / (elements per group over ALL ranks). */
int32_t ccv_groupnorm_chunks(int32_t instances, int32_t rows_per_instance, int32_t C);
/* 1 when ccv_groupnorm runs this problem as a single launch (small slices: 8x8 / 4x4 latents): epilogue statistics save nothing there. */
int32_t ccv_groupnorm_single_launch(int32_t instances, int32_t rows_per_instance, int32_t C, int32_t x_kind);
int ccv_groupnorm_stats(const void* x, int32_t x_f32, int32_t instances, int32_t rows_per_instance, int32_t C, void* ws, void* stream);
int ccv_groupnorm_apply(const void* x, int32_t x_f32, uint16_t* y, const float* gamma, const float* beta, int32_t instances,
                        int32_t rows_per_instance, int32_t C, float eps, int32_t silu, const void* ws, float inv_count, void* stream);
/* ... and on statistics produced elsewhere (ccv_gemm's gn_partial): `parts` (<= 512) slots of [32][2] fp32 per instance, summed in slot order. */
int ccv_groupnorm_apply_parts(const void* x, int32_t x_f32, uint16_t* y, const float* gamma, const float* beta, int32_t instances,
                              int32_t rows_per_instance, int32_t C, float eps, int32_t silu, const void* partial, int32_t parts, void* stream);

/* LayerNorm over the last dim, fp32 (x_kind 1) or fp16 (x_kind 2) in -> bf16 out, fp32 statistics; optional second output
 * y2[r] = y[r] + addend[r % addend_rows] (bf16 [addend_rows, C]) used for the Pluecker-feature add `normed_x + pluker_embedding_features`
 * (model/modules/modified_forwards.py:508-515).  Replaces nn.LayerNorm (attention.py:232-234).
 * Constraint: C % 64 == 0, C <= 2048. */
int ccv_layernorm(const void* x, int32_t x_kind, uint16_t* y, const float* gamma, const float* beta,
                  int32_t rows, int32_t C, float eps,
                  const uint16_t* addend, int32_t addend_rows, uint16_t* y2, void* stream);
/* LayerNorm fp32 in (row stride ldx) -> fp32 out [rows, C], any C, one wave per row: the output norms of the
 * once-per-clip modules (MultiLatentEpipolarAdaptor model/modules/adaptors.py:180: 4 channels; Resampler
 * lvdm/modules/encoders/resampler.py:162: 1024), whose results are returned in fp32. */
int ccv_layernorm_small(const float* x, float* y, const float* gamma, const float* beta, int64_t rows, int32_t C,
                        int64_t ldx, float eps, void* stream);

/* ------------------------------------------------------------------------------------
 * Layout / elementwise helpers on the path.
 * ------------------------------------------------------------------------------------ */
/* cat([x, c_concat], dim=1) + 'b c t h w -> (b t h w) c' (lvdm/models/ddpm3d.py:1270,
 * openaimodel3d.py:587): x [b,c1,t,h,w], x2 [b,c2,t,h,w] (or NULL, c2=0) fp32 ->
 * out [b*t*h*w, ldo] fp32, channels c1+c2..ldo-1 zero filled. */
int ccv_pack_nchw_to_rows(const float* x, int32_t c1, const float* x2, int32_t c2,
                          float* out, int32_t ldo, int32_t b, int32_t t, int32_t hw, void* stream);
/* '(b t) c h w -> b c t h w' of the first c columns (openaimodel3d.py:623): in [rows, ldi] fp32. */
int ccv_unpack_rows_to_nchw(const float* in, int32_t ldi, float* out, int32_t c,
                            int32_t b, int32_t t, int32_t hw, void* stream);
/* torch.cat([h, skip], dim=1) on token-major rows of the residual stream (openaimodel3d.py:617), kind 1 = fp32 / 2 = fp16 (a, b
 * and out alike); out_bf16 (or NULL) receives the same rows rounded to bf16, the form the ResBlock's 1x1 skip convolution
 * consumes as a GEMM operand. */
int ccv_concat_rows(const void* a, int32_t ca, const void* b, int32_t cb, void* out, uint16_t* out_bf16,
                    int64_t rows, int32_t kind, void* stream);
/* Once-per-clip camera feeders (model/base.py:112-174, model/modules/camera_pose_encoder.py:361-376).
 * ccv_ray_condition: K [B,V,3,3], c2w [B,V,4,4] fp32 -> out [B,6,V,H,W] fp32, (o x d | d) when plucker != 0 else (o | d).
 * ccv_pixel_unshuffle_rows: x [n,c,H,W] fp32 -> token rows [(n H/r W/r), c r^2] bf16 (torch.nn.PixelUnshuffle order).
 * ccv_avgpool2_rows: token rows [(n H W), C] fp32 -> [(n H/2 W/2), C] fp32 (nn.AvgPool2d(2)). */
int ccv_ray_condition(const float* K, const float* c2w, float* out, int32_t B, int32_t V, int32_t H, int32_t W,
                      int32_t plucker, void* stream);
int ccv_pixel_unshuffle_rows(const float* x, uint16_t* y, int32_t n, int32_t c, int32_t H, int32_t W, int32_t r, void* stream);
int ccv_avgpool2_rows(const float* x, float* y, int32_t n, int32_t H, int32_t W, int32_t C, void* stream);
/* 3x3x3 convolution (padding 1) over [B, Cin, T, H, W] fp32 with Cin, Cout <= 8, + bias, + optional image `add`
 * [B, Cout, H, W] broadcast over T: the latent projection + residual after the context-frame adaptor
 * (nn.Conv3d(4, 4, 3, 1, 1), model/camcontexti2v.py:81-84, 368-373). */
int ccv_conv3d_small(const float* x, const float* w, const float* bias, const float* add, float* y, int32_t B, int32_t Cin,
                     int32_t Cout, int32_t T, int32_t H, int32_t W, void* stream);
/* CrossNormalization over the last three dims (model/modules/utils.py:30-45; used after the context-frame adaptor when
 * use_cross_normalization is set, model/camcontexti2v.py:354-364): x = n_slices contiguous slices of len_x fp32, slice s
 * takes the mean and unbiased std of reference slice s / slices_per_ref (len_ref fp32 each):
 * y = (x - mean_x) * (std_ref / (std_x + eps)) + mean_ref (the reference hard-codes eps = 1e-5 here).  y may alias x. */
int ccv_cross_norm(const float* x, const float* ref, float* y, int32_t n_slices, int64_t len_x, int32_t slices_per_ref,
                   int64_t len_ref, float eps, void* stream);
/* Row softmax fp32 [rows, ldx] -> bf16 [rows, ldy] over L columns: the single-head, 512-wide attention of the first-stage
 * decoder (lvdm/modules/networks/ae_modules.py:66-70) runs as GEMM (QK^T, alpha = C^-1/2) -> this -> GEMM (P V). */
int ccv_softmax_rows(const float* x, uint16_t* y, int32_t rows, int32_t L, int64_t ldx, int64_t ldy, void* stream);
/* fp32 (x_kind 1) or fp16 (x_kind 2) -> bf16 (contexts, pose features, stream rows that feed a convolution); with an optional
 * 'b c t h w -> (b t h w) c' transpose (ccv_nchw_to_rows_bf16). */
int ccv_cast_bf16(const void* x, int32_t x_kind, uint16_t* y, int64_t n, void* stream);
int ccv_nchw_to_rows_bf16(const float* x, uint16_t* y, int32_t b, int32_t c, int32_t t, int32_t hw, void* stream);
/* timestep_embedding (lvdm/models/utils_diffusion.py:8-28): t [n] fp32 -> [n, dim] bf16 (cos | sin). */
int ccv_timestep_embedding(const float* t, uint16_t* out, int32_t n, int32_t dim, void* stream);
/* out = SiLU(a + b) as bf16 (b may be NULL): the `emb` path of openaimodel3d.py:168-170,598. */
int ccv_add_silu_bf16(const float* a, const float* b, uint16_t* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * ccv_ddim_cfg_step: classifier-free guidance + std rescale + DDIM update, fp32.
 * Replaces lvdm/models/samplers/ddim.py:267,282-283,305-346 and
 * lvdm/models/utils_diffusion.py:147-157.
 *   e = e_uc + scale*(e_c - e_uc)                       (e_uc NULL => e = e_c)
 *   e = gr * e*std(e_c)/std(e) + (1-gr)*e               (per sample, unbiased std)
 *   x0 = (x - sqrt(1-a_t) e)/sqrt(a_t);  x_prev = sqrt(a_prev) x0 + sqrt(max(1-a_prev-s^2,0)) e + s z
 * coef: DEVICE pointer to 4 floats {a_t, a_prev, sigma_t, sqrt(1-a_t)} (device resident so a
 * captured graph can be replayed for every step).  ws: n_samples*4 floats.  noise may be NULL.
 * ------------------------------------------------------------------------------------ */
int ccv_ddim_cfg_step(const float* x, const float* e_c, const float* e_uc, const float* noise,
                      float* x_prev, float* pred_x0, const float* coef,
                      float scale, float guidance_rescale,
                      int32_t n_samples, int64_t per_sample, float* ws, void* stream);
/* Camera guidance, the optional third forward of the sampler (lvdm/models/samplers/ddim.py:268-280):
 *   model_output = e_uc + s (e_c - e_uc) + (camera_cfg - 1) w(t) (e_c - e_nc),  e_nc = conditional prediction without camera,
 * is the plain guidance formula applied to  out = e_uc + coeff * w(t) * (e_c - e_nc)  with coeff = (camera_cfg - 1)/(1 - s);
 * this call forms `out` (then handed to ccv_ddim_cfg_step as e_uc).  t: DEVICE int64 [n_samples] timesteps for the
 * 'cosine' scheduler, w = cos((1 - t/999) pi/2); NULL for 'constant' (w = 1).  out may alias e_uc. */
int ccv_camera_cfg_fold(const float* e_uc, const float* e_c, const float* e_nc, const int64_t* t, float coeff, float* out,
                        int32_t n_samples, int64_t per_sample, void* stream);

/* ------------------------------------------------------------------------------------
 * Epipolar mask preparation (once per clip).
 * ccv_pack_mask: bool mask [B, Lq, Lk] (1 byte per element, as handed over in
 * camera_condition["sample_locs_dict"], model/camcontexti2v.py:552) -> bit-packed rows
 * [B, Lq, words] + tile flags [B, ceil(Lq/128), ceil(Lk/64)] + per-64-query-group key-block bitmaps
 * wave_bits [B, ceil(Lq/64), ceil(ceil(Lk/32)/32)] (flags and wave_bits must be zeroed by the caller; either may be NULL).
 * ccv_epipolar_mask_bits: the same packed form straight from the fundamental matrices
 * F [B, T, T, 3, 3] (model/camcontexti2v.py:200-239), never materialising the bool tensor.
 * Patch order (perm_hw = H*W, perm_w = W, needs H % 4 == 0 and W % 8 == 0): rows and bit columns are emitted in the
 * order frame -> 4x8-pixel patch -> pixel, so that a 32-query x 32-key MFMA block covers two compact patches; an
 * epipolar line then crosses ~28 % of the blocks instead of ~42 % in raster order (32x32 latents).  Attention
 * calls using such a mask pass the same perm_hw / perm_w and read/write q, k, v, o rows through the same map.
 * ------------------------------------------------------------------------------------ */
int ccv_pack_mask(const uint8_t* mask, uint32_t* bits, uint8_t* flags, uint32_t* wave_bits,
                  int32_t B, int32_t Lq, int32_t Lk, int32_t perm_hw, int32_t perm_w, void* stream);
int ccv_epipolar_mask_bits(const float* F, uint32_t* bits, uint8_t* flags, uint32_t* wave_bits,
                           int32_t B, int32_t T, int32_t H, int32_t W, int32_t downsample, int32_t patch_order, void* stream);
/* Same with Tq query frames and Tk key frames, F [B, Tq, Tk, 3, 3]: rows (Tq H W) x bit columns (Tk H W) -- the
 * target-frame x context-frame mask of the adaptor (compute_conditional_epipolar_mask, model/camcontexti2v.py:493-521). */
int ccv_epipolar_mask_bits_rect(const float* F, uint32_t* bits, uint8_t* flags, uint32_t* wave_bits,
                                int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t W, int32_t downsample, int32_t patch_order,
                                void* stream);

/* Host-side view of the sparse attention kernel's per-XCD work queues (no device work; tests): the kernel gives each of the 8
 * XCDs a longest-first queue of (batch-head slice, rank) items -- the slices s with s % 8 == xq in full plus an equal share of
 * one leftover slice -- so that an XCD's L2 holds the K/V of ~1-3 slices instead of all of them.  Returns the length of queue
 * `xq`; for 0 <= idx < length also *bh / *rank of item idx (rank >= ngroups: padding, skipped by the kernel).  Reference
 * call site of the attention itself: model/modules/epipolar.py:75-102. */
int64_t ccv_attn_sparse_queue_item(int32_t nbh, int32_t ngroups, int32_t xq, int64_t idx, int32_t* bh, int32_t* rank);
/* Schedule of the sparse attention kernel (once per clip, after the mask was packed on the same stream):
 * order[b][r] = index of the 64-query group with the r-th largest popcount of its wave_bits row (ties: lower index first).
 * wave_bits [B, ngroups, wave_words], order [B, ngroups] int32; ngroups <= 8192. */
int ccv_attn_group_order(const uint32_t* wave_bits, int32_t B, int32_t ngroups, int32_t wave_words, int32_t* order, void* stream);
/* The same for items of `merge` consecutive 64-query groups (CcvAttn.wg_order): order [B, ceil(ngroups / merge)] = the items by
 * decreasing popcount of the OR of their wave_bits rows.  merge in 1 .. 8. */
int ccv_attn_group_order_merged(const uint32_t* wave_bits, int32_t B, int32_t ngroups, int32_t wave_words, int32_t merge, int32_t* order,
                                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCV_H_ */
