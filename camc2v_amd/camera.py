"""Once-per-clip camera geometry that feeds the hot path.

Pose algebra (a few hundred 4x4 / 3x3 products per clip) runs as small batched torch linalg on
the device; the expensive part -- turning the pairwise fundamental matrices into the
(T*H*W)^2 epipolar visibility masks -- is the HIP kernel ``ccv_epipolar_mask_bits``, which writes
the bit-packed rows and tile flags the attention kernel consumes and never materialises the
268 MB boolean tensor the reference builds (model/camcontexti2v.py:200-271).

Reference anchors: model/base.py:176-199 (get_relative_pose), model/camcontexti2v.py:172-198
(pairs, fundamental matrix), :273-278 (perturbation), :525-572 (assembly).
"""
import torch

from . import ops, rng


def relative_c2w(w2c, cond_frame_index, trace_scale_factor=1.0):
    """w2c [b,t,4,4] -> c2w relative to the conditioning frame ('left' mode), translation scaled."""
    c2w = torch.linalg.inv(w2c.float())
    b = c2w.shape[0]
    first = c2w[torch.arange(b, device=c2w.device), cond_frame_index].unsqueeze(1)
    rel = torch.linalg.inv(first) @ c2w
    rel[:, :, :3, 3] = rel[:, :, :3, 3] * trace_scale_factor
    return rel


def pairwise_fundamental(K, rel_c2w, perturb_zero_translation=True, generator=None, noise=None):
    """K [b,t,3,3] (pixel intrinsics), rel_c2w [b,t,4,4] -> F [b,t1,t2,3,3].

    Pair (t1,t2) uses inv(RT[t2]) @ RT[t1]; E = t x R (column-wise cross product); F = K^-T E K^-1.
    Exactly-zero translations (same-frame pairs) are replaced by 1e-6 * N(0,1) as the reference does when
    ``add_small_perturbation_on_zero_T`` is set (``noise`` overrides the draw for reproducible tests)."""
    pairs = torch.linalg.inv(rel_c2w)[:, None] @ rel_c2w[:, :, None]
    R, t = pairs[..., :3, :3], pairs[..., :3, 3:4]
    if perturb_zero_translation:
        if noise is None:
            noise = torch.randn(t.shape, device=t.device, dtype=t.dtype, generator=generator) if generator is not None else rng.randn(t.shape, device=t.device, dtype=t.dtype)
        zero = (t.abs() < 1e-6).all(dim=-2, keepdim=True)
        t = torch.where(zero, noise.to(t) * 1e-6, t)
    E = torch.cross(t.expand_as(R), R, dim=-2)
    K_inv = torch.linalg.inv(K.float().unsqueeze(1))
    return K_inv.transpose(-1, -2) @ E @ K_inv


def conditional_fundamental(K, w2c, w2c_context, cond_frame_index=None):
    """Fundamental matrices between the T target frames and the context frames (reference
    compute_conditional_epipolar_mask, model/camcontexti2v.py:493-516): K [b,t,3,3] target intrinsics, w2c [b,t,4,4],
    w2c_context [b,n,4,4]; with cond_frame_index the conditioning frame's pose is prepended to the context poses.
    Pair (t, c) uses inv(c2w_ctx[c]) @ c2w[t]; E = t x R; F = K_t^-T E K_t^-1.  Returns F [b, t, n(+1), 3, 3]."""
    c2w = torch.linalg.inv(w2c.float())
    ctx = torch.linalg.inv(w2c_context.float())
    if cond_frame_index is not None:
        b = c2w.shape[0]
        ctx = torch.cat([c2w[torch.arange(b, device=c2w.device), cond_frame_index].unsqueeze(1), ctx], 1)
    rel = torch.linalg.inv(ctx)[:, None] @ c2w[:, :, None]                 # [b, t, c, 4, 4]
    R, t = rel[..., :3, :3], rel[..., :3, 3:4]
    E = torch.cross(t.expand_as(R), R, dim=-2)
    K_inv = torch.linalg.inv(K.float())[:, :, None]
    return K_inv.transpose(-1, -2) @ E @ K_inv


def epipolar_masks_packed(F, T, H_px, W_px, attention_resolution=(8, 4, 2, 1), patch_order=True):
    """F [b,T,T,3,3] -> {8*ds: (bits int32 [b, L, L/32], flags uint8 [b, L/128, L/64], perm, wave_bits int32 [b, L/64, L/1024],
    group_order int32 [b, L/64])},
    L = T*(H_px/8ds)*(W_px/8ds).
    perm = (H*W, W) when the mask is emitted in 4x8-patch token order (feature maps with H % 4 == 0, W % 8 == 0;
    fewer non-empty 32x32 attention blocks), else None (raster order)."""
    out = {}
    for ds in attention_resolution:
        d = int(8 * ds)
        H, W = H_px // d, W_px // d
        po = bool(patch_order and ops.patch_order_ok(H, W))
        mp = ops.epipolar_mask_bits(F, T, H, W, d, patch_order=po)
        out[d] = (mp[0], mp[1], (H * W, W) if po else None, mp.wave_bits, mp.group_order)
    return out


def camera_condition(K, w2c, cond_frame_index, H_px, W_px, pluker_features=None, add_type="add_to_main_branch",
                     attention_resolution=(8, 4, 2, 1), trace_scale_factor=1.0, perturb=True, generator=None, noise=None):
    """The dict the UNet takes as ``camera_condition`` (reference model/camcontexti2v.py:565-570), with the
    masks in packed form under ``sample_locs_packed`` (``sample_locs_dict`` stays available for bool masks)."""
    rel = relative_c2w(w2c, cond_frame_index, trace_scale_factor)
    F = pairwise_fundamental(K, rel, perturb, generator, noise)
    return {
        "pluker_embedding_features": pluker_features,
        "sample_locs_dict": None,
        "sample_locs_packed": epipolar_masks_packed(F, w2c.shape[1], H_px, W_px, attention_resolution),
        "cond_frame_index": cond_frame_index,
        "add_type": add_type,
        "fundamental": F,
        "relative_c2w": rel,
    }


def synthetic_trajectory(b, t, device, yaw_step=0.02, dx=0.05, dz=0.02):
    """Benchmark camera of SURVEY.md section 8(d): yaw 0.02 rad/frame + translation (0.05, 0, 0.02)/frame.
    Returns w2c [b,t,4,4]."""
    f = torch.arange(t, dtype=torch.float32)
    c2w = torch.eye(4).repeat(t, 1, 1)
    c2w[:, 0, 0], c2w[:, 0, 2] = torch.cos(yaw_step * f), torch.sin(yaw_step * f)
    c2w[:, 2, 0], c2w[:, 2, 2] = -torch.sin(yaw_step * f), torch.cos(yaw_step * f)
    c2w[:, 0, 3], c2w[:, 2, 3] = dx * f, dz * f
    return torch.linalg.inv(c2w).unsqueeze(0).repeat(b, 1, 1, 1).to(device)
