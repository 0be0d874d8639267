"""ctypes binding of libccv_hip.so (the C ABI declared in include/ccv.h).

The product path has no CPU fallback: if the shared library is missing, cannot be
loaded, or an entry point is absent, importing this module's ``lib()`` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CCV_OPERANDS=f16: the build whose MFMA operand type (element kind 0 of the C ABI) is IEEE half instead of bf16 (round 4's numerics
# experiment, csrc/ccv_common.h: ccv_opnd_t); the Python side then types operand tensors torch.float16 (ops.BF16)
OPERANDS = os.environ.get("CCV_OPERANDS", "bf16")
if OPERANDS not in ("bf16", "f16"):
    raise ValueError(f"CCV_OPERANDS must be bf16 or f16, got {OPERANDS!r}")
LIB_NAME = "libccv_hip.so" if OPERANDS == "bf16" else "libccv_hip_f16.so"
LIB_PATH = os.environ.get("CCV_HIP_LIB") or os.path.join(_HERE, LIB_NAME)    # CCV_HIP_LIB: another build of the library (A/B runs)

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class CcvGemm(C.Structure):
    _fields_ = [
        ("A", vp), ("W", vp), ("C", vp), ("bias", vp), ("bias2", vp), ("residual", vp),
        ("M", i32), ("N", i32), ("K", i32), ("taps", i32),
        ("lda", i32), ("ldc", i32), ("ldr", i32), ("ldb2", i32),
        ("a_f32", i32), ("gather", i32),
        ("out_h", i32), ("out_w", i32), ("src_h", i32), ("src_w", i32),
        ("stride", i32), ("upsample", i32), ("no_lead_pad", i32), ("frames", i32), ("hw", i32),
        ("rows_per_batch", i32), ("act", i32), ("geglu", i32), ("out_f32", i32),
        ("alpha", f32),
        ("ws", vp), ("ws_bytes", i64), ("split_k", i32),
        ("gn_partial", vp), ("gn_rows", i32), ("gn_slots", i32),
        ("res_f16", i32), ("tile_order", i32),
        ("ln_gamma", vp), ("ln_beta", vp), ("ln_eps", f32),
    ]


class CcvFF(C.Structure):
    _fields_ = [
        ("x", vp), ("ln_gamma", vp), ("ln_beta", vp), ("ln_eps", f32),
        ("w1", vp), ("b1", vp), ("w2p", vp), ("b2", vp), ("out", vp),
        ("M", i32), ("C", i32), ("ldx", i32), ("ldo", i32), ("out_kind", i32),
    ]


class CcvAttn(C.Structure):
    _fields_ = [
        ("q", vp), ("k", vp), ("v", vp), ("o", vp),
        ("q_bso", i64), ("q_bsi", i64), ("q_ls", i64),
        ("k_bso", i64), ("k_bsi", i64), ("k_ls", i64),
        ("v_bso", i64), ("v_bsi", i64), ("v_ls", i64),
        ("o_bso", i64), ("o_bsi", i64), ("o_ls", i64),
        ("B", i32), ("inner", i32), ("H", i32), ("Lq", i32), ("Lk", i32),
        ("scale", f32),
        ("k2", vp), ("v2", vp),
        ("k2_bso", i64), ("k2_bsi", i64), ("k2_ls", i64),
        ("v2_bso", i64), ("v2_bsi", i64), ("v2_ls", i64),
        ("Lk2", i32), ("gate2", f32),
        ("mask_bits", vp), ("mask_bs", i64), ("mask_words", i32), ("mask_nb", i32),
        ("tile_flags", vp), ("flags_bs", i64), ("flags_ktiles", i32),
        ("wave_bits", vp), ("wave_bs", i64), ("wave_words", i32),
        ("group_order", vp), ("order_bs", i64),
        ("kreg", vp), ("vreg", vp), ("nreg", i32),
        ("perm_hw", i32), ("perm_w", i32),
        ("variant", i32),
        ("queue_counters", vp),
        ("wg_order", vp), ("wg_order_bs", i64), ("wg_merge", i32),
        ("split_ws", vp), ("split_ws_bytes", i64), ("split_all_parts", i32),
    ]


# name -> (restype, argtypes); every symbol include/ccv.h declares
SIGNATURES = {
    "ccv_version": (i32, []),
    "ccv_set_streams_in_flight": (i32, [i32]),
    "ccv_last_error": (C.c_char_p, []),
    "ccv_gemm": (i32, [C.POINTER(CcvGemm), vp]),
    "ccv_gemm_ws_bytes": (i64, [C.POINTER(CcvGemm)]),
    "ccv_gemm_plan": (i32, [C.POINTER(CcvGemm), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ccv_gemm_ln_fusable": (i32, [C.POINTER(CcvGemm)]),
    "ccv_gemm_gn_slots": (i32, [C.POINTER(CcvGemm), i32]),
    "ccv_ff_fusable": (i32, [C.POINTER(CcvFF)]),
    "ccv_ff_fused": (i32, [C.POINTER(CcvFF), vp]),
    "ccv_groupnorm_apply_parts": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, f32, i32, vp, i32, vp]),
    "ccv_attn_fwd": (i32, [C.POINTER(CcvAttn), vp]),
    "ccv_attn_split_ws_bytes": (i64, [C.POINTER(CcvAttn)]),
    "ccv_groupnorm_ws_bytes": (i64, [i32, i32]),
    "ccv_groupnorm_chunks": (i32, [i32, i32, i32]),
    "ccv_groupnorm_single_launch": (i32, [i32, i32, i32, i32]),
    "ccv_groupnorm_stats": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "ccv_groupnorm_apply": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, f32, i32, vp, f32, vp]),
    "ccv_groupnorm": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp]),
    "ccv_layernorm": (i32, [vp, i32, vp, vp, vp, i32, i32, f32, vp, i32, vp, vp]),
    "ccv_pack_nchw_to_rows": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    "ccv_unpack_rows_to_nchw": (i32, [vp, i32, vp, i32, i32, i32, i32, vp]),
    "ccv_concat_rows": (i32, [vp, i32, vp, i32, vp, vp, i64, i32, vp]),
    "ccv_attn_small_fwd": (i32, [C.POINTER(CcvAttn), i32, vp]),
    "ccv_ray_condition": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ccv_pixel_unshuffle_rows": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ccv_conv3d_small": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ccv_cross_norm": (i32, [vp, vp, vp, i32, i64, i32, i64, f32, vp]),
    "ccv_epipolar_mask_bits_rect": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ccv_avgpool2_rows": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ccv_layernorm_small": (i32, [vp, vp, vp, vp, i64, i32, i64, f32, vp]),
    "ccv_softmax_rows": (i32, [vp, vp, i32, i32, i64, i64, vp]),
    "ccv_cast_bf16": (i32, [vp, i32, vp, i64, vp]),
    "ccv_nchw_to_rows_bf16": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ccv_timestep_embedding": (i32, [vp, vp, i32, i32, vp]),
    "ccv_add_silu_bf16": (i32, [vp, vp, vp, i64, vp]),
    "ccv_ddim_cfg_step": (i32, [vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, i64, vp, vp]),
    "ccv_camera_cfg_fold": (i32, [vp, vp, vp, vp, f32, vp, i32, i64, vp]),
    "ccv_pack_mask": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ccv_epipolar_mask_bits": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ccv_attn_group_order": (i32, [vp, i32, i32, i32, vp, vp]),
    "ccv_attn_group_order_merged": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "ccv_attn_sparse_queue_item": (i64, [i32, i32, i32, i64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}

_lib = None


class CcvError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CcvError(
                f"{LIB_PATH} is missing: build it with `python -m camc2v_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback on the product path.")
        # torch first: its wheel carries its own libamdhip64, and device pointers / streams handed to the kernels come
        # from that runtime; loading libccv_hip.so before torch would bind it to /opt/rocm's copy (a second runtime
        # instance in the process: launches then fail with "no ROCm-capable device is detected")
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.ccv_version() < 100:
            raise CcvError("libccv_hip.so is older than this package")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().ccv_last_error()
        raise CcvError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
