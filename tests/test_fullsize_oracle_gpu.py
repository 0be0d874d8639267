"""Parity of the headline workload itself against the oracle / reference-held data at FULL size (BASELINE.json
configs[1]: 1.5 B-parameter CamContextI2V UNet, 1 x 16 x 256 x 256 clip, camera + 2 context frames):

  * one CFG step's UNet work -- the conditional half (context 77 + 768 tokens) and the unconditional half (77 + 256),
    both camera conditioned (Pluecker rows, register tokens, the 16384 / 4096 / 1024 / 256-token epipolar masks) -- run
    as the product runs it (`apply_model_pair`, native packed masks) against `oracle.unet_oracle.unet_forward` on the
    host with the same weights, inputs and fundamental matrices.  Stated tolerance: rel-L2 <= 2.5e-2 per half.
  * the HIP epipolar-mask kernel at the headline size (32x32 latents, L = 16384) against the per-query-row popcounts
    the REFERENCE produced for the same F (tests/golden/geometry.npz, written by oracle/gen_golden.py).
"""
import os
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REL_L2, MAX_REL = 2.5e-2, 8e-2


def test_full_size_camera_cfg_step_vs_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bench
    from camc2v_amd import configs
    from oracle import geometry_oracle, unet_oracle
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    model = bench.build_model(dev)
    cond, uncond, fs, x_T, _ = bench.synthetic_inputs(model, dev)
    t = torch.full((1,), 439, dtype=torch.long, device=dev)
    uc = dict(uncond, camera_condition=dict(cond["camera_condition"], is_uc=True))
    e_c, e_uc = model.apply_model_pair(x_T, t, cond, uc, fs=fs, enable_camera_condition=True)
    e_c, e_uc = e_c.float().cpu(), e_uc.float().cpu()

    # ---- the same two forwards on the host ---------------------------------------------------------------------------
    unet = model.model.diffusion_model
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    cam = cond["camera_condition"]
    F = cam["fundamental"].float().cpu()
    masks = {d: geometry_oracle.epipolar_mask(F, 256 // d, 256 // d, d) for d in (8, 16, 32, 64)}
    cam_cpu = dict(pluker_embedding_features=[f.float().cpu() for f in cam["pluker_embedding_features"]],
                   sample_locs_dict=masks, add_type=cam["add_type"])
    x = torch.cat([x_T, cond["c_concat"][0]], 1).float().cpu()
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    for got, ctx, what in ((e_c, cond["c_crossattn"][0], "conditional"), (e_uc, uncond["c_crossattn"][0], "unconditional")):
        t0 = time.perf_counter()
        ref = unet_oracle.unet_forward(sd, configs.UNET_256, x, t.cpu(), ctx.float().cpu(), fs.cpu(), cam_cpu, origin_h=256)
        dt = time.perf_counter() - t0
        l2 = ((got - ref).norm() / ref.norm()).item()
        mx = ((got - ref).abs().max() / ref.abs().max()).item()
        print(f"[parity] full-size camera forward, {what} half (ctx {ctx.shape[1]}) vs fp32 oracle: rel_l2={l2:.3e} "
              f"max_rel={mx:.3e} (oracle {dt:.1f} s)")
        assert torch.isfinite(got).all() and l2 <= REL_L2 and mx <= MAX_REL, (what, l2, mx)
    del model
    torch.cuda.empty_cache()


def _row_popcounts(bits):
    """int32 [L, words] (device) -> int64 numpy [L]."""
    b = bits.contiguous().cpu().numpy().view(np.uint8)
    return np.unpackbits(b, axis=-1).sum(-1).astype(np.int64)


def test_hip_mask_rows_vs_reference_popcounts_full_size(golden_dir):
    """`ccv_epipolar_mask_bits` on the reference's own F at 256 x 256 px, every attention resolution including the
    32x32-latent mask of the headline config: per-query-row popcounts against what the reference's get_epipolar_mask
    produced (reference-held fixture).  The GPU evaluates the same fp32 formula with its own rounding of the 3-term dot
    products, so single bits within an ulp of the threshold may flip; budget: <= 1e-4 of the set bits in total, and the
    raster-order and patch-order emissions must hold exactly the same bits per row."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from camc2v_amd import ops
    fx = np.load(os.path.join(golden_dir, "geometry.npz"))
    F = torch.from_numpy(fx["F256"]).cuda()
    for d in (8, 16, 32, 64):
        H = 256 // d
        ref = fx[f"mask256_d{d}_popcount_rows"].astype(np.int64)[0]
        bits = ops.epipolar_mask_bits(F, 16, H, H, d)[0]
        got = _row_popcounts(bits[0])
        rows_off = int((got != ref).sum())
        flips = int(np.abs(got - ref).sum())
        print(f"[parity] HIP mask d={d} (L={16 * H * H}): {rows_off} of {ref.size} rows differ from the reference popcounts, "
              f"|delta| = {flips} of {int(ref.sum())} set bits")
        assert flips <= 1e-4 * ref.sum() + 1, (d, flips)
        if ops.patch_order_ok(H, H):
            # patch order permutes rows (and bit columns) within each frame: frame f, pixel (r, c) sits at
            # f*HW + patch*32 + (r%4)*8 + c%8 with patch = (r//4)*(W//8) + c//8
            pbits = ops.epipolar_mask_bits(F, 16, H, H, d, patch_order=True)[0]
            gp = _row_popcounts(pbits[0])
            r, c = np.meshgrid(np.arange(H), np.arange(H), indexing="ij")
            pos = ((r // 4) * (H // 8) + c // 8) * 32 + (r % 4) * 8 + c % 8
            perm = (np.arange(16)[:, None] * H * H + pos.reshape(-1)[None]).reshape(-1)
            assert np.array_equal(gp[perm], got), f"d={d}: patch-order rows hold different bits than raster-order rows"
