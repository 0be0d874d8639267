"""Tensor-level wrappers over the C ABI (include/ccv.h).

torch is used for device memory and streams only: every wrapper checks device /
dtype / layout, allocates the output with torch.empty and launches the HIP kernel
on torch's current stream.  There is no CPU path: a CPU tensor raises.
"""
import ctypes as C
import math
import threading

import torch

from . import lib as _lib_mod
from .lib import CcvAttn, CcvError, CcvFF, CcvGemm, check, lib

BF16 = torch.bfloat16 if _lib_mod.OPERANDS == "bf16" else torch.float16     # the MFMA operand type: element kind 0 of the C ABI (lib.OPERANDS)
F32 = torch.float32
F16 = torch.float16      # the residual stream's hand-off format between UNet blocks (arithmetic stays fp32 inside the kernels)
_KIND = {BF16: 0, F32: 1, F16: 2}      # element-kind codes of the C ABI (include/ccv.h)


def _kind(t, allowed, what):
    if t.dtype not in allowed:
        raise CcvError(f"{what}: dtype {t.dtype} not supported (expected one of {[str(d) for d in allowed]})")
    return _KIND[t.dtype]

GATHER_LINEAR, GATHER_CONV3X3, GATHER_TCONV3, GATHER_SEGMENTS = 0, 1, 2, 3
ACT_NONE, ACT_SILU, ACT_GELU, ACT_RELU = 0, 1, 2, 3

# 0: V^T fragments through ds_read_b64_tr_b16; 1: V transposed while staging (fallback)
ATTN_VARIANT = 0
SPARSE_VARIANT = None   # tests / A-B aid: 6 = the per-wave sparse kernel, 4 / 5 = the workgroup-shared kernel with 8 / 4 waves, for every call that carries wave_bits


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise CcvError("camc2v_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _untag(t):
    """An op is about to write into the caller-supplied tensor t: GroupNorm statistics riding on it (``_ccv_gn``, tag_stats) describe
    the OLD contents -- the kernels write through raw pointers, so torch's ``_version`` never moves -- and are dropped here."""
    if t is not None:
        t.__dict__.pop("_ccv_gn", None)
    return t


def _rows(t):
    """[..., C] contiguous -> (rows, C)."""
    if not t.is_contiguous():
        raise CcvError("expected a contiguous tensor")
    return t.numel() // t.shape[-1], t.shape[-1]


# ---------------------------------------------------------------------------------------
# GEMM family
# ---------------------------------------------------------------------------------------
STREAMS_IN_FLIGHT = 1


def set_streams_in_flight(n):
    """Planner hint: how many independent launch streams the caller keeps busy (see include/ccv.h).  Returns the previous value."""
    global STREAMS_IN_FLIGHT
    STREAMS_IN_FLIGHT = max(1, int(n))
    return lib().ccv_set_streams_in_flight(int(n))


# LayerNorm in the consuming GEMM's prologue: "1" always, "0" never, default "auto" = when ONE launch stream is busy.  Measured in the
# model on one box (profiles/r03_ab_switches.txt): one clip at a time +0.8 % (25 launches and 1 GB of traffic per step less); with two
# clips in flight -1.1 % -- the separate LayerNorm is a streaming kernel that runs beside the other clip's GEMMs for free, while the
# longer prologue holds a whole CU's LDS.
_FUSE_LN_MODE = __import__("os").environ.get("CCV_LN_FUSE", "auto")


def _fuse_ln():
    return _FUSE_LN_MODE == "1" or (_FUSE_LN_MODE != "0" and STREAMS_IN_FLIGHT < 2)


class LazyLN:
    """LayerNorm(x) as a GEMM operand that need not exist in memory: ``gemm(LazyLN(...), w)`` runs the norm inside the GEMM's prologue
    where the kernel for that problem has one (ccv_gemm_ln_fusable: the A-stationary kernel of the 32x32-latent blocks, x the fp16
    stream) and otherwise materialises the bf16 rows with ``layernorm`` first (once: the result is kept for further consumers)."""

    def __init__(self, x, gamma, beta, eps):
        self.x, self.gamma, self.beta, self.eps = x, gamma, beta, float(eps)
        self.shape, self.device, self.dtype = x.shape, x.device, BF16
        self._t = None

    def tensor(self):
        if self._t is None:
            self._t = layernorm(self.x, self.gamma, self.beta, eps=self.eps)
        return self._t


def gemm(a, w, *, n_out=None, k=None, taps=1, lda=None, m=None, bias=None, bias2=None, ldb2=0, rows_per_batch=0,
         residual=None, act=ACT_NONE, geglu=False, out_f32=False, out=None, gather=GATHER_LINEAR,
         conv=None, tconv=None, seg_rows=None, alpha=1.0, debug_ws=None, gn_rows=None, out_dtype=None):
    """out[m, n] = epilogue(sum_tap gather(a) @ w_tap^T).  See include/ccv.h (ccv_gemm).
    gn_rows: the output feeds a GroupNorm(32) whose instances are `gn_rows` consecutive rows; the call then returns
    (out, stats) with stats = GroupNorm statistics produced by the epilogue (hand them to ``groupnorm(..., stats=)``) or None when
    the kernel this problem runs on cannot produce them (``groupnorm`` then computes its own).

    out_dtype: torch.bfloat16 (default), torch.float32 (= out_f32=True) or torch.float16 (the stream's hand-off format);
    residual: fp32 or fp16 [M, >= N].

    a: [rows, lda] bf16 or fp32 (2-D, last dim contiguous); w: [N, taps*K] bf16.
    conv = (out_h, out_w, src_h, src_w, stride, upsample[, no_lead_pad]); tconv = (frames, hw);
    seg_rows (GATHER_SEGMENTS): a = `taps` stacked operands, `seg_rows` rows apart; tap t multiplies rows [t*seg_rows, +M).
    """
    _untag(out)
    ln = None
    if isinstance(a, LazyLN):       # try the LayerNorm prologue (decided below, once the problem is described); else the norm runs first
        ln = a
        a = ln.x if (_fuse_ln() and ln._t is None and ln.x.dtype == F16 and gather == GATHER_LINEAR and taps == 1) else ln.tensor()
    _dev(a, w, bias, bias2, residual, out)
    if a.dim() != 2 or a.stride(1) != 1:
        raise CcvError("gemm: A must be 2-D with a contiguous last dim")
    if w.dtype != BF16 or not w.is_contiguous():
        raise CcvError("gemm: W must be contiguous bf16")
    N = w.shape[0]
    K = k if k is not None else w.shape[1] // taps
    if w.shape[1] != taps * K:
        raise CcvError(f"gemm: W has {w.shape[1]} columns, expected taps*K = {taps}*{K}")
    M = m if m is not None else a.shape[0]
    n_cols = N // 2 if geglu else N
    if out_dtype is None:
        out_dtype = F32 if out_f32 else BF16
    elif out_dtype not in _KIND:
        raise CcvError(f"gemm: out_dtype {out_dtype} not supported")
    if out is None:
        out = torch.empty((M, n_cols), dtype=out_dtype, device=a.device)
    p = CcvGemm()
    p.A, p.W, p.C = _ptr(a), _ptr(w), _ptr(out)
    p.bias, p.bias2, p.residual = _ptr(bias), _ptr(bias2), _ptr(residual)
    p.M, p.N, p.K, p.taps = M, N, K, taps
    p.lda = lda if lda is not None else a.stride(0)
    p.ldc = out.stride(0)
    p.ldr = residual.stride(0) if residual is not None else 0
    p.ldb2 = ldb2 if bias2 is not None else 0
    if a.dtype == F32:
        p.a_f32 = 1
    elif a.dtype == BF16 or (ln is not None and a is ln.x):      # (the fp16 stream under a LayerNorm prologue: two bytes per element too)
        p.a_f32 = 0
    else:
        raise CcvError(f"gemm: unsupported A dtype {a.dtype}")
    if bias is not None and (bias.dtype != F32 or bias.numel() < N):
        raise CcvError("gemm: bias must be fp32 [N]")
    if residual is not None:
        p.res_f16 = int(_kind(residual, (F32, F16), "gemm: residual") == 2)
        if residual.stride(1) != 1:
            raise CcvError("gemm: residual must have a contiguous last dim")
    if out.dtype != out_dtype:
        raise CcvError("gemm: out dtype mismatch")
    p.gather = gather
    # the kernels trust M and the gather geometry: check them against the tensors here, on the host
    if out.dim() != 2 or out.shape[0] < M or out.shape[1] < n_cols or out.stride(1) != 1:
        raise CcvError(f"gemm: out {tuple(out.shape)} cannot hold [{M}, {n_cols}]")
    if residual is not None and (residual.dim() != 2 or residual.shape[0] < M or residual.shape[1] < n_cols):
        raise CcvError(f"gemm: residual {tuple(residual.shape)} is smaller than the output [{M}, {n_cols}]")
    if a.shape[1] < K:
        raise CcvError(f"gemm: A has {a.shape[1]} columns, K = {K}")
    if gather == GATHER_CONV3X3:
        p.out_h, p.out_w, p.src_h, p.src_w, p.stride, p.upsample = conv[:6]
        p.no_lead_pad = int(conv[6]) if len(conv) > 6 else 0
        pix = p.out_h * p.out_w
        if pix <= 0 or M % pix or a.shape[0] < (M // pix) * p.src_h * p.src_w:
            raise CcvError(f"gemm: conv3x3 output rows M={M} = images x {p.out_h}x{p.out_w} need {(M // max(pix, 1)) * p.src_h * p.src_w} "
                           f"source rows ({p.src_h}x{p.src_w} per image), A has {a.shape[0]} (pass m= for strided / upsampling convolutions)")
    elif gather == GATHER_SEGMENTS:
        if seg_rows is None or seg_rows < M or a.shape[0] < (taps - 1) * seg_rows + M or a.dtype != BF16:
            raise CcvError(f"gemm: segment gather needs bf16 A with {taps} segments of >= M = {M} rows, seg_rows = {seg_rows}, A has {a.shape[0]} rows")
        p.hw = seg_rows
    else:
        if a.shape[0] < M:
            raise CcvError(f"gemm: A has {a.shape[0]} rows, M = {M}")
        if gather == GATHER_TCONV3:
            p.frames, p.hw = tconv
            if p.frames <= 0 or p.hw <= 0 or M % (p.frames * p.hw):
                raise CcvError(f"gemm: tconv rows M={M} are not whole clips of {p.frames} frames x {p.hw} pixels")
    if bias2 is not None:   # [batches, >= N] fp32 rows ldb2 apart (a column slice of a wider table is fine)
        nb2 = (M + max(rows_per_batch, 1) - 1) // max(rows_per_batch, 1)
        ok = rows_per_batch > 0 and bias2.dtype == F32
        if ok and bias2.dim() == 2:
            ok = bias2.shape[0] >= nb2 and bias2.shape[1] >= N and bias2.stride(1) == 1 and (nb2 == 1 or bias2.stride(0) == ldb2)
        elif ok:
            ok = bias2.is_contiguous() and bias2.numel() >= (nb2 - 1) * ldb2 + N
        if not ok:
            raise CcvError(f"gemm: bias2 must hold {nb2} fp32 rows of >= {N} columns, ldb2 = {ldb2} apart")
    p.rows_per_batch = rows_per_batch
    # (fp16-operand build: BF16 is F16, a two-byte output is kind 2 -- the same bits as kind 0 -- except where the C side asks for kind 0)
    p.act, p.geglu, p.out_f32, p.alpha = act, int(geglu), (0 if geglu else {BF16: 0, F32: 1, F16: 2}[out_dtype]), alpha
    if ln is not None and a is ln.x:
        p.ln_gamma, p.ln_beta, p.ln_eps = _ptr(ln.gamma), _ptr(ln.beta), ln.eps
        if not lib().ccv_gemm_ln_fusable(C.byref(p)):          # no prologue for this problem: normalise first
            a = ln.tensor()
            p.A, p.lda, p.ln_gamma, p.ln_beta = _ptr(a), (lda if lda is not None else a.stride(0)), None, None
    stats = None
    if gn_rows and M % gn_rows == 0 and not lib().ccv_groupnorm_single_launch(M // gn_rows, gn_rows, n_cols, _KIND[out_dtype]):
        slots = lib().ccv_gemm_gn_slots(C.byref(p), gn_rows)
        if slots > 0:
            part = torch.empty((M // gn_rows, slots, 64), dtype=F32, device=a.device)
            p.gn_partial, p.gn_rows, p.gn_slots = _ptr(part), gn_rows, slots
            stats = (part, gn_rows)
    ws_bytes = lib().ccv_gemm_ws_bytes(C.byref(p))
    if ws_bytes > 0:  # split-K workspace for long-K / few-tile problems
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
        p.ws, p.ws_bytes = _ptr(ws), ws_bytes
    if debug_ws is not None:   # diagnosis: the A-stationary kernel stamps its phases into this buffer (tools/astat_stamps.py)
        p.ws, p.ws_bytes = _ptr(debug_ws), debug_ws.numel() * debug_ws.element_size()
    global LAST_GEMM_PLAN
    if TRACK_GEMM_PLAN:
        tile, split = C.c_int32(0), C.c_int32(0)
        check(lib().ccv_gemm_plan(C.byref(p), C.byref(tile), C.byref(split)), "ccv_gemm_plan")
        LAST_GEMM_PLAN = (tile.value, split.value)
    probe = GEMM_PROBE
    if probe is not None:      # measurement aid (bench.py): HIP events on the launch stream around this GEMM's launch(es)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib().ccv_gemm(C.byref(p), _stream()), "ccv_gemm")
    if probe is not None:
        e1.record()
        probe.append((e0, e1, 2.0 * M * N * K * taps))
    return (out, stats) if gn_rows else out


def _ff_params(x, gamma, beta, eps, w1, b1, w2p, b2, out):
    p = CcvFF()
    p.x, p.ln_gamma, p.ln_beta, p.ln_eps = _ptr(x), _ptr(gamma), _ptr(beta), float(eps)
    p.w1, p.b1, p.w2p, p.b2, p.out = _ptr(w1), _ptr(b1), _ptr(w2p), _ptr(b2), _ptr(out)
    p.M, p.C, p.ldx, p.ldo = x.shape[0], x.shape[1], x.stride(0), out.stride(0)
    p.out_kind = _KIND[out.dtype]
    return p


def ff_fusable(x, w1, w2p):
    """Whether ``ff_fused`` takes this feed-forward (csrc/ccv_fused.hip: the fp16 stream of the 32x32-latent blocks, C = 320)."""
    if w2p is None or x.dtype != F16 or x.dim() != 2 or x.stride(1) != 1 or not x.is_cuda:
        return False
    C_ = x.shape[1]
    if tuple(w1.shape) != (8 * C_, C_) or tuple(w2p.shape) != (C_, 4 * C_):
        return False
    return bool(lib().ccv_ff_fusable(C.byref(_ff_params(x, x, x, 0.0, w1, x, w2p, x, x))))


FF_PROBE = None     # a list: ff_fused() appends (start event, end event, flops) per call (eager mode only; bench.py's gemm_family)


def ff_fused(x, gamma, beta, eps, w1, b1, w2p, b2, *, out=None, out_dtype=None):
    """out = x + Linear_2(value * gelu(gate)), [value | gate] = Linear_1(LayerNorm(x)) in ONE launch (include/ccv.h, ccv_ff_fused;
    reference lvdm/modules/attention.py:253,431-458).  x: the fp16 stream [M, C]; w1 / b1: GEGLU-interleaved (pack.interleave_geglu);
    w2p: pack.permute_k16_for_acc_operand(W2).  out: fp16 (default: a new tensor; may be x itself) or bf16."""
    _untag(out)
    _dev(x, gamma, beta, w1, b1, w2p, b2, out)
    if x.dtype != F16 or x.dim() != 2 or x.stride(1) != 1:
        raise CcvError("ff_fused: x must be the fp16 stream [M, C] with a contiguous last dim")
    M, C_ = x.shape
    if w1.dtype != BF16 or w2p.dtype != BF16 or not w1.is_contiguous() or not w2p.is_contiguous() or \
            tuple(w1.shape) != (8 * C_, C_) or tuple(w2p.shape) != (C_, 4 * C_):
        raise CcvError(f"ff_fused: w1 must be contiguous bf16 [8C, C] and w2p [C, 4C] for C = {C_}")
    for t, n in ((gamma, C_), (beta, C_), (b1, 8 * C_), (b2, C_)):
        if t.dtype != F32 or not t.is_contiguous() or t.numel() < n:
            raise CcvError("ff_fused: gamma / beta / b1 / b2 must be contiguous fp32 vectors of C / C / 8C / C elements")
    if out is None:
        out = torch.empty((M, C_), dtype=out_dtype or F16, device=x.device)
    if out.dtype not in (F16, BF16) or out.dim() != 2 or out.shape[0] < M or out.shape[1] < C_ or out.stride(1) != 1:
        raise CcvError(f"ff_fused: out must be fp16 / bf16 [>= {M}, >= {C_}]")
    p = _ff_params(x, gamma, beta, eps, w1, b1, w2p, b2, out)
    probe = FF_PROBE
    if probe is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib().ccv_ff_fused(C.byref(p), _stream()), "ccv_ff_fused")
    if probe is not None:
        e1.record()
        probe.append((e0, e1, 2.0 * M * C_ * 12 * C_))
    return out


# tests / tuning tools: when TRACK_GEMM_PLAN is set, LAST_GEMM_PLAN = (ring tile index or -1, split-K) of the last call
TRACK_GEMM_PLAN = False
LAST_GEMM_PLAN = None
GEMM_PROBE = None     # a list: gemm() appends (start event, end event, 2 M N K taps) per call (eager mode only; bench.py's gemm_family)


# ---------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------
_ctr_tls = threading.local()


class queue_counter_arena:
    """One zero-filled int32 buffer for the work-queue counters of all sparse-attention launches issued by this thread inside the
    ``with`` block (a UNet forward: one fill launch instead of one per attention call).  Rows are handed out in order and never
    reused inside the block, so launches may overlap freely; calls beyond ``rows`` (or on another device) allocate their own."""

    def __init__(self, device, rows=32):
        self.device, self.rows = device, rows

    def __enter__(self):
        self.prev = getattr(_ctr_tls, "arena", None)
        _ctr_tls.arena = [torch.zeros(8 * self.rows, dtype=torch.int32, device=self.device), 0, self.rows]
        return self

    def __exit__(self, *exc):
        _ctr_tls.arena = self.prev
        return False


def in_queue_counter_arena():
    return getattr(_ctr_tls, "arena", None) is not None


def _queue_counters(device):
    a = getattr(_ctr_tls, "arena", None)
    if a is not None and a[1] < a[2] and a[0].device == device:
        row = a[0][8 * a[1]: 8 * a[1] + 8]
        a[1] += 1
        return row
    return torch.zeros(8, dtype=torch.int32, device=device)


SPARSE_SPLIT = True   # False: no key-split items (tests / A-B aid; split=False per call does the same)
_SPLIT_WS = {}        # (device, stream) -> zeroed uint8 workspace of the sparse kernel's key-split items (grown on demand)


def _split_workspace(device, nbytes):
    """The calls of one stream run one after the other and every launch leaves the counters at zero again, so one buffer per stream
    serves them all; another stream (a second clip in flight) gets its own."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _SPLIT_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
        _SPLIT_WS[key] = ws
    return ws


SPARSE_PROBE = None   # a list: attention() appends (start event, end event, Lq, H, B) per sparse-kernel launch (eager mode only)


def attention(q, k, v, *, B, inner, H, Lq, Lk, q_str, k_str, v_str, out=None, o_str=None, scale=None,
              k2=None, v2=None, k2_str=None, v2_str=None, Lk2=0, gate2=1.0,
              mask_bits=None, mask_nb=1, tile_flags=None, wave_bits=None, group_order=None, kreg=None, vreg=None, variant=None,
              perm=None, split=True, split_all=0):
    """Fused attention, head dim 64.  *_str = (batch_outer, batch_inner, token) strides in elements;
    q/k/v are bf16 tensors whose data_ptr() is the element (batch 0, token 0, head 0, d 0).
    Returns bf16 [B*Lq, H*64] unless `out`/`o_str` are given."""
    _untag(out)
    _dev(q, k, v, k2, v2, mask_bits, tile_flags, kreg, vreg, out)
    for t in (q, k, v, k2, v2, kreg, vreg):
        if t is not None and t.dtype != BF16:
            raise CcvError("attention: q/k/v must be bf16")
    if out is None:
        out = torch.empty((B * Lq, H * 64), dtype=BF16, device=q.device)
        o_str = ((Lq * H * 64) * inner, Lq * H * 64, H * 64)
    p = CcvAttn()
    p.q, p.k, p.v, p.o = _ptr(q), _ptr(k), _ptr(v), _ptr(out)
    p.q_bso, p.q_bsi, p.q_ls = q_str
    p.k_bso, p.k_bsi, p.k_ls = k_str
    p.v_bso, p.v_bsi, p.v_ls = v_str
    p.o_bso, p.o_bsi, p.o_ls = o_str
    p.B, p.inner, p.H, p.Lq, p.Lk = B, inner, H, Lq, Lk
    p.scale = scale if scale is not None else 1.0 / math.sqrt(64.0)
    if k2 is not None:
        p.k2, p.v2 = _ptr(k2), _ptr(v2)
        p.k2_bso, p.k2_bsi, p.k2_ls = k2_str
        p.v2_bso, p.v2_bsi, p.v2_ls = v2_str
        p.Lk2, p.gate2 = Lk2, gate2
    if mask_bits is not None:
        if mask_bits.dtype != torch.int32 or not mask_bits.is_contiguous():
            raise CcvError("attention: mask_bits must be contiguous int32 [nb, Lq, words]")
        p.mask_bits = _ptr(mask_bits)
        p.mask_words = mask_bits.shape[-1]
        p.mask_bs = mask_bits.shape[-2] * mask_bits.shape[-1]
        p.mask_nb = mask_nb
        if tile_flags is not None:
            p.tile_flags = _ptr(tile_flags)
            p.flags_ktiles = tile_flags.shape[-1]
            p.flags_bs = tile_flags.shape[-2] * tile_flags.shape[-1]
        if wave_bits is not None:
            # the persistent sparse kernel's work-queue counters: caller-owned and zeroed here, so that launches of several
            # streams / replays of several graphs may overlap (no state kept in the library)
            ctr = _queue_counters(q.device)
            p.queue_counters = _ptr(ctr)
            p.wave_bits = _ptr(wave_bits)
            p.wave_words = wave_bits.shape[-1]
            p.wave_bs = wave_bits.shape[-2] * wave_bits.shape[-1]
            if group_order is not None:
                g64 = wave_bits.shape[-2]
                n_wg = (g64 + WG_MERGE - 1) // WG_MERGE
                if group_order.dtype != torch.int32 or not group_order.is_contiguous() or group_order.shape[-1] not in (g64, g64 + n_wg):
                    raise CcvError("attention: group_order must be contiguous int32 [mask_nb, ceil(Lq/64)] (+ the merged order, attn_group_order)")
                p.group_order = _ptr(group_order)
                p.order_bs = group_order.shape[-1]
                if group_order.shape[-1] == g64 + n_wg:   # the workgroup-shared kernel's items ride behind the 64-query groups' order
                    p.wg_order = C.c_void_p(group_order.data_ptr() + 4 * g64)
                    p.wg_order_bs = group_order.shape[-1]
                    p.wg_merge = WG_MERGE
    if kreg is not None:
        p.kreg, p.vreg, p.nreg = _ptr(kreg), _ptr(vreg), kreg.shape[0]
    if perm is not None:   # (frame tokens, frame width): rows and mask are in 4x8-patch order
        p.perm_hw, p.perm_w = perm
    p.variant = (SPARSE_VARIANT if (SPARSE_VARIANT is not None and wave_bits is not None and mask_bits is not None and k2 is None) else ATTN_VARIANT) if variant is None else variant
    # measurement aid (bench.py): HIP events on the launch stream around the launches ccv_attn_fwd routes to the persistent
    # sparse kernel (same rule as csrc/ccv_attn.hip: block bitmap given and variant 3 or >= 1024 64-query groups)
    probe = SPARSE_PROBE if (wave_bits is not None and mask_bits is not None
                             and (p.variant >= 3 or ((Lq + 63) // 64) * H * B >= 1024)) else None
    p.split_all_parts = int(split_all)
    if wave_bits is not None and mask_bits is not None and split and SPARSE_SPLIT:
        need = lib().ccv_attn_split_ws_bytes(C.byref(p))
        if need > 0:      # key-split tail of the workgroup-shared sparse kernel: one zeroed, self-cleaning workspace per stream
            ws = _split_workspace(q.device, need)
            p.split_ws, p.split_ws_bytes = _ptr(ws), ws.numel()
    if probe is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib().ccv_attn_fwd(C.byref(p), _stream()), "ccv_attn_fwd")
    if probe is not None:
        e1.record()
        probe.append((e0, e1, Lq, H, B))
    return out


def groupnorm(x, gamma, beta, *, instances, eps, silu, stats=None):
    """x [rows, C] fp32|fp16|bf16 -> bf16; statistics per (instance, group of C/32 channels).  stats: what ``gemm(..., gn_rows=)``
    returned for the GEMM that produced x (the statistics pass is skipped), or None; a tensor tagged by ``tag_stats`` carries them."""
    _dev(x, gamma, beta)
    if stats is None:
        stats = getattr(x, "_ccv_gn", None)
        if stats is not None and (stats[1] * instances != x.shape[0] or stats[2] != x._version):
            stats = None      # produced for another instance size, or the tensor was written since
        elif stats is not None:
            stats = stats[:2]
    rows, Cc = _rows(x)
    if rows % instances:
        raise CcvError("groupnorm: rows not divisible by instances")
    xk = _kind(x, (F32, F16, BF16), "groupnorm")
    y = torch.empty((rows, Cc), dtype=BF16, device=x.device)
    if stats is not None:
        part, gn_rows = stats
        if gn_rows * instances != rows or part.shape[0] != instances or part.shape[2] != 64 or part.dtype != F32 or not part.is_contiguous():
            raise CcvError(f"groupnorm: stats were produced for instances of {gn_rows} rows, x has {rows} rows in {instances} instances")
        check(lib().ccv_groupnorm_apply_parts(_ptr(x), xk, _ptr(y), _ptr(gamma), _ptr(beta), instances, gn_rows, Cc, eps,
                                              int(silu), _ptr(part), part.shape[1], _stream()), "ccv_groupnorm_apply_parts")
        return y
    ws = torch.empty(lib().ccv_groupnorm_ws_bytes(instances, Cc), dtype=torch.uint8, device=x.device)
    check(lib().ccv_groupnorm(_ptr(x), xk, _ptr(y), _ptr(gamma), _ptr(beta), instances,
                              rows // instances, Cc, eps, int(silu), _ptr(ws), _stream()), "ccv_groupnorm")
    return y


def tag_stats(out, stats):
    """Let ``out`` carry the GroupNorm statistics its producing GEMM emitted (``gemm(..., gn_rows=)`` -> (out, stats)): the next
    ``groupnorm(out, ...)`` with matching instances skips its statistics pass.  The tag dies with any in-place write to ``out``: torch
    writes move ``_version``; this package's own kernels write through raw pointers, so every op that takes a caller-supplied output
    (``gemm(out=)``, ``attention(out=)``, ``ff_fused(out=)``, ``layernorm(out2=)``) drops the tag explicitly (``_untag``)."""
    if stats is not None:
        out._ccv_gn = (stats[0], stats[1], out._version)
    return out


def groupnorm_sharded(x, gamma, beta, *, instances, eps, silu, reduce_sums, total_rows_per_instance):
    """GroupNorm whose statistics span rows held by several processes: x [rows, C] are THIS process's rows of every instance;
    ``reduce_sums(t)`` sums the fp32 tensor t [instances, 64] in place over the processes (an all_reduce);
    ``total_rows_per_instance`` = rows of an instance over all processes.  Same kernels as ``groupnorm``, split in two."""
    _dev(x, gamma, beta)
    rows, Cc = _rows(x)
    if rows % instances:
        raise CcvError("groupnorm_sharded: rows not divisible by instances")
    rpi = rows // instances
    nchunk = lib().ccv_groupnorm_chunks(instances, rpi, Cc)
    ws = torch.empty(lib().ccv_groupnorm_ws_bytes(instances, Cc) // 4, dtype=F32, device=x.device)
    xf = _kind(x, (F32, F16, BF16), "groupnorm_sharded")
    check(lib().ccv_groupnorm_stats(_ptr(x), xf, instances, rpi, Cc, _ptr(ws), _stream()), "ccv_groupnorm_stats")
    part = ws[:instances * nchunk * 64].view(instances, nchunk, 64)
    sums = part.sum(1)
    reduce_sums(sums)
    part.zero_()
    part[:, 0] = sums
    y = torch.empty((rows, Cc), dtype=BF16, device=x.device)
    inv_count = 1.0 / (float(total_rows_per_instance) * (Cc // 32))
    check(lib().ccv_groupnorm_apply(_ptr(x), xf, _ptr(y), _ptr(gamma), _ptr(beta), instances, rpi, Cc, eps, int(silu), _ptr(ws), inv_count,
                                    _stream()), "ccv_groupnorm_apply")
    return y


def groupnorm_partial_sums(x, instances):
    """This process's (sum, sum of squares) per (instance, group) of x [rows, C]: fp32 [instances, 64] (the statistics half of
    ``groupnorm``; see ``groupnorm_apply_sums``)."""
    _dev(x)
    rows, Cc = _rows(x)
    if rows % instances:
        raise CcvError("groupnorm_partial_sums: rows not divisible by instances")
    rpi = rows // instances
    nchunk = lib().ccv_groupnorm_chunks(instances, rpi, Cc)
    ws = torch.empty(lib().ccv_groupnorm_ws_bytes(instances, Cc) // 4, dtype=F32, device=x.device)
    check(lib().ccv_groupnorm_stats(_ptr(x), _kind(x, (F32, F16, BF16), "groupnorm_partial_sums"), instances, rpi, Cc, _ptr(ws), _stream()),
          "ccv_groupnorm_stats")
    return ws[:instances * nchunk * 64].view(instances, nchunk, 64).sum(1)


def groupnorm_apply_sums(x, gamma, beta, sums, *, instances, total_rows_per_instance, eps, silu):
    """The normalise half of ``groupnorm`` on given statistics: sums fp32 [instances, 64] = (sum, sum of squares) per group over ALL
    ``total_rows_per_instance`` rows of an instance (which may live on other processes: x holds any subset of them)."""
    _dev(x, gamma, beta, sums)
    rows, Cc = _rows(x)
    if rows % instances or tuple(sums.shape) != (instances, 64) or sums.dtype != F32:
        raise CcvError("groupnorm_apply_sums: sums must be fp32 [instances, 64] and rows divisible by instances")
    rpi = rows // instances
    nchunk = lib().ccv_groupnorm_chunks(instances, rpi, Cc)
    ws = torch.zeros(lib().ccv_groupnorm_ws_bytes(instances, Cc) // 4, dtype=F32, device=x.device)
    ws[:instances * nchunk * 64].view(instances, nchunk, 64)[:, 0] = sums
    y = torch.empty((rows, Cc), dtype=BF16, device=x.device)
    inv_count = 1.0 / (float(total_rows_per_instance) * (Cc // 32))
    check(lib().ccv_groupnorm_apply(_ptr(x), _kind(x, (F32, F16, BF16), "groupnorm_apply_sums"), _ptr(y), _ptr(gamma), _ptr(beta), instances, rpi, Cc,
                                    eps, int(silu), _ptr(ws), inv_count, _stream()), "ccv_groupnorm_apply")
    return y


def layernorm(x, gamma, beta, *, eps=1e-5, addend=None, out2=None):
    """x [rows, C] fp32|fp16 -> bf16 (and y + addend[r % addend_rows] when addend is given; `out2`: where that second output goes)."""
    _untag(out2)
    _dev(x, gamma, beta, addend)
    xk = _kind(x, (F32, F16), "layernorm (the residual stream)")
    rows, Cc = _rows(x)
    y = torch.empty((rows, Cc), dtype=BF16, device=x.device)
    y2 = None
    arows = 0
    if addend is not None:
        if addend.dtype != BF16 or not addend.is_contiguous() or addend.shape[-1] != Cc:
            raise CcvError("layernorm: addend must be contiguous bf16 [rows', C]")
        arows = addend.numel() // Cc
        y2 = torch.empty_like(y) if out2 is None else out2
        if y2.dtype != BF16 or not y2.is_contiguous() or tuple(y2.shape) != tuple(y.shape):
            raise CcvError("layernorm: out2 must be contiguous bf16 shaped like the output")
    check(lib().ccv_layernorm(_ptr(x), xk, _ptr(y), _ptr(gamma), _ptr(beta), rows, Cc, eps, _ptr(addend), arows,
                              _ptr(y2), _stream()), "ccv_layernorm")
    return (y, y2) if addend is not None else y


# ---------------------------------------------------------------------------------------
# layout / elementwise
# ---------------------------------------------------------------------------------------
def pack_nchw_to_rows(x, x2=None, ldo=64):
    """cat([x, x2], 1) as token-major fp32 rows [(b t h w), ldo], zero padded."""
    _dev(x, x2)
    x = x.contiguous().float()
    b, c1, t, h, w = x.shape
    c2 = 0
    if x2 is not None:
        x2 = x2.contiguous().float()
        c2 = x2.shape[1]
    out = torch.empty((b * t * h * w, ldo), dtype=F32, device=x.device)
    check(lib().ccv_pack_nchw_to_rows(_ptr(x), c1, _ptr(x2), c2, _ptr(out), ldo, b, t, h * w, _stream()),
          "ccv_pack_nchw_to_rows")
    return out


def unpack_rows_to_nchw(rows, c, b, t, h, w):
    _dev(rows)
    out = torch.empty((b, c, t, h, w), dtype=F32, device=rows.device)
    check(lib().ccv_unpack_rows_to_nchw(_ptr(rows), rows.stride(0), _ptr(out), c, b, t, h * w, _stream()),
          "ccv_unpack_rows_to_nchw")
    return out


def concat_rows(a, b, with_bf16=False):
    """[rows, ca] ++ [rows, cb], both fp32 or both fp16; with_bf16 also returns the bf16 rounding of the result (out, out16)."""
    _dev(a, b)
    rows, ca = _rows(a)
    rows_b, cb = _rows(b)
    kind = _kind(a, (F32, F16), "concat_rows")
    if rows != rows_b or b.dtype != a.dtype:
        raise CcvError("concat_rows: inputs of one dtype with equal row counts expected")
    out = torch.empty((rows, ca + cb), dtype=a.dtype, device=a.device)
    out16 = torch.empty((rows, ca + cb), dtype=BF16, device=a.device) if with_bf16 else None
    check(lib().ccv_concat_rows(_ptr(a), ca, _ptr(b), cb, _ptr(out), _ptr(out16), rows, kind, _stream()), "ccv_concat_rows")
    return (out, out16) if with_bf16 else out


def attention_small(q, k, v, *, B, inner, H, T, head_dim, q_str, k_str, v_str, out=None, o_str=None, scale=None):
    """Self-attention over T <= 16 tokens, any head width (multiple of 8, <= 256); strides as in ``attention``."""
    _untag(out)
    _dev(q, k, v, out)
    for t in (q, k, v):
        if t.dtype != BF16:
            raise CcvError("attention_small: q/k/v must be bf16")
    if out is None:
        out = torch.empty((B * T, H * head_dim), dtype=BF16, device=q.device)
        o_str = ((T * H * head_dim) * inner, T * H * head_dim, H * head_dim)
    p = CcvAttn()
    p.q, p.k, p.v, p.o = _ptr(q), _ptr(k), _ptr(v), _ptr(out)
    p.q_bso, p.q_bsi, p.q_ls = q_str
    p.k_bso, p.k_bsi, p.k_ls = k_str
    p.v_bso, p.v_bsi, p.v_ls = v_str
    p.o_bso, p.o_bsi, p.o_ls = o_str
    p.B, p.inner, p.H, p.Lq, p.Lk = B, inner, H, T, T
    p.scale = scale if scale is not None else 1.0 / math.sqrt(float(head_dim))
    check(lib().ccv_attn_small_fwd(C.byref(p), head_dim, _stream()), "ccv_attn_small_fwd")
    return out


def ray_condition(K, c2w, H, W, plucker=True):
    """K [B,V,3,3], c2w [B,V,4,4] -> fp32 [B, 6, V, H, W] (reference model/base.py:112-174)."""
    _dev(K, c2w)
    K, c2w = K.float().contiguous(), c2w.float().contiguous()
    B, V = K.shape[:2]
    out = torch.empty((B, 6, V, H, W), dtype=F32, device=K.device)
    check(lib().ccv_ray_condition(_ptr(K), _ptr(c2w), _ptr(out), B, V, H, W, int(plucker), _stream()), "ccv_ray_condition")
    return out


def pixel_unshuffle_rows(x, r):
    """x fp32 [n, c, H, W] -> bf16 token rows [(n H/r W/r), c r^2] in torch.nn.PixelUnshuffle channel order."""
    _dev(x)
    x = x.float().contiguous()
    n, c, H, W = x.shape
    y = torch.empty((n * (H // r) * (W // r), c * r * r), dtype=BF16, device=x.device)
    check(lib().ccv_pixel_unshuffle_rows(_ptr(x), _ptr(y), n, c, H, W, r, _stream()), "ccv_pixel_unshuffle_rows")
    return y


def avgpool2_rows(x, n, H, W):
    """fp32 token rows [(n H W), C] -> [(n H/2 W/2), C] (nn.AvgPool2d(2))."""
    _dev(x)
    if x.dtype != F32 or not x.is_contiguous() or x.shape[0] != n * H * W:
        raise CcvError("avgpool2_rows: contiguous fp32 rows [(n H W), C] expected")
    y = torch.empty((n * (H // 2) * (W // 2), x.shape[1]), dtype=F32, device=x.device)
    check(lib().ccv_avgpool2_rows(_ptr(x), _ptr(y), n, H, W, x.shape[1], _stream()), "ccv_avgpool2_rows")
    return y


def conv3d_small(x, weight, bias=None, add=None):
    """x fp32 [B, Cin, T, H, W], weight [Cout, Cin, 3, 3, 3] (Cin, Cout <= 8), padding 1, + bias, + add [B, Cout, H, W]
    broadcast over T."""
    _dev(x, weight, bias, add)
    x, weight = x.float().contiguous(), weight.float().contiguous()
    B, Cin, T, H, W = x.shape
    Cout = weight.shape[0]
    if tuple(weight.shape) != (Cout, Cin, 3, 3, 3):
        raise CcvError("conv3d_small: weight must be [Cout, Cin, 3, 3, 3]")
    bias = bias.float().contiguous() if bias is not None else None
    add = add.float().contiguous() if add is not None else None
    if add is not None and tuple(add.shape) != (B, Cout, H, W):
        raise CcvError("conv3d_small: add must be [B, Cout, H, W]")
    y = torch.empty((B, Cout, T, H, W), dtype=F32, device=x.device)
    check(lib().ccv_conv3d_small(_ptr(x), _ptr(weight), _ptr(bias), _ptr(add), _ptr(y), B, Cin, Cout, T, H, W, _stream()), "ccv_conv3d_small")
    return y


def cross_norm(x, ref, n_slices, slices_per_ref=1, eps=1e-5):
    """CrossNormalization over contiguous slices (model/modules/utils.py:30-45): x fp32 = n_slices equal slices, slice s is
    moved to the mean / unbiased std of ref's slice s // slices_per_ref.  Returns a new fp32 tensor shaped like x."""
    _dev(x, ref)
    x, ref = x.float().contiguous(), ref.float().contiguous()
    if n_slices <= 0 or x.numel() % n_slices or n_slices % slices_per_ref or ref.numel() % (n_slices // slices_per_ref):
        raise CcvError(f"cross_norm: {x.numel()} / {ref.numel()} elements do not split into {n_slices} slices, {slices_per_ref} per reference")
    y = torch.empty_like(x)
    check(lib().ccv_cross_norm(_ptr(x), _ptr(ref), _ptr(y), n_slices, x.numel() // n_slices, slices_per_ref,
                               ref.numel() // (n_slices // slices_per_ref), float(eps), _stream()), "ccv_cross_norm")
    return y


def layernorm_small(x, gamma, beta, *, eps=1e-5):
    """x fp32 [rows, >= C] (first C = gamma.numel() columns used) -> fp32 [rows, C]; for small once-per-clip tensors."""
    _dev(x, gamma, beta)
    if x.dtype != F32 or x.dim() != 2 or x.stride(1) != 1:
        raise CcvError("layernorm_small: fp32 [rows, C'] with a contiguous last dim expected")
    Cc = gamma.numel()
    y = torch.empty((x.shape[0], Cc), dtype=F32, device=x.device)
    check(lib().ccv_layernorm_small(_ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), x.shape[0], Cc, x.stride(0), eps, _stream()),
          "ccv_layernorm_small")
    return y


def softmax_rows(x):
    """fp32 [rows, L] (last dim contiguous) -> bf16 softmax over the last axis."""
    _dev(x)
    if x.dtype != F32 or x.dim() != 2 or x.stride(1) != 1:
        raise CcvError("softmax_rows: fp32 [rows, L] with a contiguous last dim expected")
    rows, L = x.shape
    y = torch.empty((rows, L), dtype=BF16, device=x.device)
    check(lib().ccv_softmax_rows(_ptr(x), _ptr(y), rows, L, x.stride(0), L, _stream()), "ccv_softmax_rows")
    return y


def cast_bf16(x):
    _dev(x)
    x = x.contiguous()
    kind = _kind(x, (F32, F16), "cast_bf16")
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(lib().ccv_cast_bf16(_ptr(x), kind, _ptr(y), x.numel(), _stream()), "ccv_cast_bf16")
    return y


def nchw_to_rows_bf16(x):
    """[b, c, t, h, w] fp32 -> [(b t h w), c] bf16."""
    _dev(x)
    x = x.contiguous().float()
    b, c, t, h, w = x.shape
    y = torch.empty((b * t * h * w, c), dtype=BF16, device=x.device)
    check(lib().ccv_nchw_to_rows_bf16(_ptr(x), _ptr(y), b, c, t, h * w, _stream()), "ccv_nchw_to_rows_bf16")
    return y


def timestep_embedding(t, dim):
    """t [n] (any numeric dtype) -> [n, dim] bf16, cos | sin."""
    _dev(t)
    tf = t.to(F32).contiguous()
    out = torch.empty((tf.numel(), dim), dtype=BF16, device=t.device)
    check(lib().ccv_timestep_embedding(_ptr(tf), _ptr(out), tf.numel(), dim, _stream()), "ccv_timestep_embedding")
    return out


def add_silu_bf16(a, b=None):
    _dev(a, b)
    for t in (a, b):
        if t is not None and (t.dtype != F32 or not t.is_contiguous()):
            raise CcvError("add_silu_bf16: contiguous fp32 tensors expected")
    if b is not None and b.shape != a.shape:
        raise CcvError(f"add_silu_bf16: shapes differ ({tuple(a.shape)} vs {tuple(b.shape)}); the kernel reads a.numel() elements of both")
    out = torch.empty(a.shape, dtype=BF16, device=a.device)
    check(lib().ccv_add_silu_bf16(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "ccv_add_silu_bf16")
    return out


def ddim_cfg_step(x, e_c, e_uc, noise, coef, scale, guidance_rescale, want_x0=True):
    """Fused CFG + rescale + DDIM update.  coef: device fp32 [4] = (a_t, a_prev, sigma_t, sqrt(1-a_t))."""
    _dev(x, e_c, e_uc, noise, coef)
    for t in (x, e_c, e_uc, noise):
        if t is not None and (t.dtype != F32 or not t.is_contiguous()):
            raise CcvError("ddim_cfg_step: contiguous fp32 tensors expected")
    n = x.shape[0]
    per = x.numel() // n
    x_prev = torch.empty_like(x)
    x0 = torch.empty_like(x) if want_x0 else None
    ws = torch.empty((n, 4), dtype=F32, device=x.device)
    check(lib().ccv_ddim_cfg_step(_ptr(x), _ptr(e_c), _ptr(e_uc), _ptr(noise), _ptr(x_prev), _ptr(x0), _ptr(coef),
                                  float(scale), float(guidance_rescale), n, per, _ptr(ws), _stream()),
          "ccv_ddim_cfg_step")
    return x_prev, x0


def camera_cfg_fold(e_uc, e_c, e_nc, coeff, t=None):
    """e_uc + coeff * w * (e_c - e_nc): the camera-guidance term of the sampler folded into the unconditional prediction
    (see ccv_camera_cfg_fold).  t: device int64 [n] timesteps => w = cos((1 - t/999) pi/2) ('cosine'); None => w = 1."""
    _dev(e_uc, e_c, e_nc, t)
    for v in (e_uc, e_c, e_nc):
        if v.dtype != F32 or not v.is_contiguous() or v.shape != e_uc.shape:
            raise CcvError("camera_cfg_fold: three contiguous fp32 tensors of one shape expected")
    n = e_uc.shape[0]
    if t is not None and (t.dtype != torch.long or t.numel() != n or not t.is_contiguous()):
        raise CcvError("camera_cfg_fold: t must be a contiguous int64 tensor with one timestep per sample")
    out = torch.empty_like(e_uc)
    check(lib().ccv_camera_cfg_fold(_ptr(e_uc), _ptr(e_c), _ptr(e_nc), _ptr(t), float(coeff), _ptr(out), n, e_uc.numel() // n, _stream()),
          "ccv_camera_cfg_fold")
    return out


class MaskPack(tuple):
    """(bits, flags) with the per-64-query-group key-block bitmap and the longest-first group schedule riding along as
    ``.wave_bits`` / ``.group_order``; unpacks as a pair so ``bits, flags = pack_mask(...)`` call sites keep working."""

    def __new__(cls, bits, flags, wave_bits, group_order=None):
        self = super().__new__(cls, (bits, flags))
        self.wave_bits = wave_bits
        self.group_order = group_order
        return self


WG_MERGE = 2   # 64-query groups per item of the workgroup-shared sparse kernel's default form (4 waves x 32 queries)


def attn_group_order(wave_bits, merged=True):
    """wave_bits int32 [B, groups, words] -> int32 [B, groups (+ ceil(groups / WG_MERGE))]: the 64-query groups by decreasing popcount
    (per-wave sparse kernel's schedule) and, behind them in the same rows, the items of WG_MERGE consecutive groups by decreasing size
    of the union of their key blocks (workgroup-shared kernel's schedule; ops.attention splits the two)."""
    _dev(wave_bits)
    B, groups, words = wave_bits.shape
    order = torch.empty((B, groups), dtype=torch.int32, device=wave_bits.device)
    check(lib().ccv_attn_group_order(_ptr(wave_bits), B, groups, words, _ptr(order), _stream()), "ccv_attn_group_order")
    if not merged:
        return order
    n_wg = (groups + WG_MERGE - 1) // WG_MERGE
    worder = torch.empty((B, n_wg), dtype=torch.int32, device=wave_bits.device)
    check(lib().ccv_attn_group_order_merged(_ptr(wave_bits), B, groups, words, WG_MERGE, _ptr(worder), _stream()), "ccv_attn_group_order_merged")
    return torch.cat([order, worder], dim=1).contiguous()


def patch_order_ok(H, W):
    """4x8-pixel patch order is defined for feature maps with H % 4 == 0 and W % 8 == 0."""
    return H % 4 == 0 and W % 8 == 0


def pack_mask(mask, perm=None):
    """bool [B, Lq, Lk] -> (bits int32 [B, Lq, ceil(Lk/32)], flags uint8 [B, ceil(Lq/128), ceil(Lk/64)]).
    perm = (frame tokens, frame width) emits rows and bit columns in 4x8-patch order (see include/ccv.h)."""
    _dev(mask)
    if mask.dtype != torch.bool or mask.dim() != 3:
        raise CcvError("pack_mask: bool [B, Lq, Lk] expected")
    mask = mask.contiguous()
    B, Lq, Lk = mask.shape
    bits = torch.empty((B, Lq, (Lk + 31) // 32), dtype=torch.int32, device=mask.device)
    flags = torch.zeros((B, (Lq + 127) // 128, (Lk + 63) // 64), dtype=torch.uint8, device=mask.device)
    wbits = torch.zeros((B, (Lq + 63) // 64, ((Lk + 31) // 32 + 31) // 32), dtype=torch.int32, device=mask.device)
    hw, w = perm if perm is not None else (0, 0)
    check(lib().ccv_pack_mask(_ptr(mask), _ptr(bits), _ptr(flags), _ptr(wbits), B, Lq, Lk, hw, w, _stream()), "ccv_pack_mask")
    return MaskPack(bits, flags, wbits, attn_group_order(wbits))


def epipolar_mask_bits(F, T, H, W, downsample, patch_order=False):
    """F [B, T, Tk, 3, 3] fp32 -> packed epipolar mask (bits, flags) for an HxW feature map: T query frames x Tk key frames
    (Tk = T for the UNet's temporal blocks; target frames x context frames for the adaptor)."""
    _dev(F)
    F = F.contiguous().float()
    B, Tk = F.shape[0], F.shape[2]
    if F.shape[1] != T:
        raise CcvError(f"epipolar_mask_bits: F has {F.shape[1]} query frames, expected {T}")
    L, Lk = T * H * W, Tk * H * W
    bits = torch.empty((B, L, (Lk + 31) // 32), dtype=torch.int32, device=F.device)
    flags = torch.zeros((B, (L + 127) // 128, (Lk + 63) // 64), dtype=torch.uint8, device=F.device)
    wbits = torch.zeros((B, (L + 63) // 64, ((Lk + 31) // 32 + 31) // 32), dtype=torch.int32, device=F.device)
    check(lib().ccv_epipolar_mask_bits_rect(_ptr(F), _ptr(bits), _ptr(flags), _ptr(wbits), B, T, Tk, H, W, downsample, int(patch_order),
                                            _stream()), "ccv_epipolar_mask_bits_rect")
    return MaskPack(bits, flags, wbits, attn_group_order(wbits))
