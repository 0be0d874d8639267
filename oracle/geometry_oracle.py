"""fp32 CPU restatement of the once-per-clip camera geometry that feeds the hot
path: relative poses, pairwise fundamental matrices, boolean epipolar masks.
TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference anchors (relative to /root/reference/CamContextI2V):
  get_relative_pose                         model/base.py:176-199
  get_relative_c2w_RT_pairs                 model/camcontexti2v.py:172-184
  add_small_perturbation                    model/camcontexti2v.py:273-278
  get_fundamental_matrix                    model/camcontexti2v.py:186-198
  get_epipolar_mask                         model/camcontexti2v.py:200-271
  pix2coord                                 model/modules/epipolar.py:32-34
  get_batch_input_camera_condition_process  model/camcontexti2v.py:525-572
"""
from math import sqrt

import numpy as np
import torch


def relative_c2w(w2c, cond_frame_index, trace_scale_factor=1.0):
    """w2c [b,t,4,4] -> c2w relative to the conditioning frame ('left' mode), with the
    translation scaled by trace_scale_factor (camcontexti2v.py:531-537)."""
    c2w = torch.linalg.inv(w2c.float())
    b = c2w.shape[0]
    first = c2w[torch.arange(b), cond_frame_index].unsqueeze(1)
    rel = torch.linalg.inv(first) @ c2w
    rel[:, :, :3, 3] = rel[:, :, :3, 3] * trace_scale_factor
    return rel


def relative_pairs(rel_c2w):
    """[b,t,4,4] -> [b,t1,t2,4,4] = inv(RT[t2]) @ RT[t1] (camcontexti2v.py:181-184)."""
    inv = torch.linalg.inv(rel_c2w)[:, None]          # b 1 t 4 4
    return inv @ rel_c2w[:, :, None]                   # b t t 4 4


def perturb_zero_translation(t, noise, epsilon=1e-6):
    """camcontexti2v.py:273-278 with the randn draw injected: t [..,3,1]; noise same shape ~N(0,1)."""
    zero = (t.abs() < epsilon).all(dim=-2, keepdim=True)
    return torch.where(zero, noise * epsilon, t)


def fundamental_matrix(K, R, t):
    """E = t x R (cross product column-wise along dim -2), F = K^-T E K^-1 (camcontexti2v.py:195-198)."""
    E = torch.cross(t.expand_as(R), R, dim=-2)
    K_inv = torch.linalg.inv(K)
    return K_inv.transpose(-1, -2) @ E @ K_inv


def pixel_grid(H, W, downsample):
    """Homogeneous pixel-centre coordinates (x, y, 1) of an HxW feature map, row-major
    (p = row*W + col); x = col*d + d/2 - 0.5 (epipolar.py:32-34, camcontexti2v.py:217-227)."""
    y = torch.arange(H, dtype=torch.float32) * downsample + downsample / 2.0 - 0.5
    x = torch.arange(W, dtype=torch.float32) * downsample + downsample / 2.0 - 0.5
    gy, gx = torch.meshgrid(y, x, indexing="ij")
    return torch.stack([gx, gy, torch.ones_like(gx)], dim=2).reshape(-1, 3)


def epipolar_mask(F, H, W, downsample):
    """F [B,T1,T2,3,3] -> bool [B, T1*H*W, T2*H*W] (camcontexti2v.py:229-239, :269).

    Row index = query (t1, p1), column index = key (t2, p2).  A key pixel is visible when
    its centre lies within d*sqrt(2)/2 of the query pixel's epipolar line in frame t2."""
    grid = pixel_grid(H, W, downsample)                      # HW 3
    lines = F @ grid.transpose(-1, -2)                       # B T1 T2 3 HW
    norm = torch.norm(lines[..., :2, :], dim=-2, keepdim=True)
    lines = lines / norm
    dist = (lines.transpose(-1, -2) @ grid.transpose(-1, -2)).abs()   # B T1 T2 HW HW
    mask = dist < (downsample * sqrt(2) / 2)
    B, T1, T2, P, _ = mask.shape
    return mask.permute(0, 1, 3, 2, 4).reshape(B, T1 * P, T2 * P)


def camera_masks(K, w2c, cond_frame_index, H_px, W_px, resolutions=(8, 4, 2, 1),
                 trace_scale_factor=1.0, perturb_noise=None):
    """The geometry half of get_batch_input_camera_condition_process (camcontexti2v.py:531-554).

    K [b,t,3,3] pixel intrinsics, w2c [b,t,4,4].  Returns (F [b,t,t,3,3], {8*ds: mask}).
    perturb_noise: N(0,1) tensor [b,t,t,3,1] standing in for torch.randn_like (None => no
    perturbation, i.e. add_small_perturbation_on_zero_T False)."""
    rel = relative_c2w(w2c, cond_frame_index, trace_scale_factor)
    pairs = relative_pairs(rel)
    R = pairs[..., :3, :3]
    t = pairs[..., :3, 3:4]
    if perturb_noise is not None:
        t = perturb_zero_translation(t, perturb_noise)
    F = fundamental_matrix(K.float().unsqueeze(1), R, t)
    T = w2c.shape[1]
    masks = {}
    for ds in resolutions:
        d = int(8 * ds)
        masks[d] = epipolar_mask(F, H_px // d, W_px // d, d)
    del T
    return F, masks


def conditional_fundamental(K, w2c, w2c_context, cond_frame_index=None):
    """Target-frame x context-frame fundamental matrices (compute_conditional_epipolar_mask, camcontexti2v.py:493-516):
    context poses = [conditioning frame's pose ; w2c_context]; pair (t, c) = inv(c2w_ctx[c]) @ c2w[t]; F = K_t^-T (t x R) K_t^-1.
    Returns [b, t, c, 3, 3]."""
    c2w = torch.linalg.inv(w2c.float())
    ctx = torch.linalg.inv(w2c_context.float())
    if cond_frame_index is not None:
        ctx = torch.cat([c2w[torch.arange(c2w.shape[0]), cond_frame_index].unsqueeze(1), ctx], 1)
    rel = torch.linalg.inv(ctx)[:, None] @ c2w[:, :, None]
    return fundamental_matrix(K.float()[:, :, None], rel[..., :3, :3], rel[..., :3, 3:4])


def pack_mask_bits(mask):
    """bool [..., L] -> uint8 [..., ceil(L/8)] little-endian bit order (bit j of byte i = column 8i+j)."""
    return np.packbits(mask.numpy().astype(np.uint8), axis=-1, bitorder="little")


def synthetic_trajectory(b, t, yaw_step=0.02, dx=0.05, dz=0.02):
    """SURVEY.md section 8(d) synthetic camera: c2w_f = yaw(0.02 f) + translation (0.05 f, 0, 0.02 f).
    Returns w2c [b,t,4,4] fp32."""
    c2w = torch.eye(4).repeat(b, t, 1, 1)
    for f in range(t):
        a = yaw_step * f
        c2w[:, f, 0, 0] = np.cos(a)
        c2w[:, f, 0, 2] = np.sin(a)
        c2w[:, f, 2, 0] = -np.sin(a)
        c2w[:, f, 2, 2] = np.cos(a)
        c2w[:, f, 0, 3] = dx * f
        c2w[:, f, 2, 3] = dz * f
    return torch.linalg.inv(c2w)
