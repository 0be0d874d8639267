"""Parity of the headline workload itself against the oracle / reference-held data at FULL size (BASELINE.json
configs[1]: 1.5 B-parameter CamContextI2V UNet, 1 x 16 x 256 x 256 clip, camera + 2 context frames):

  * one CFG step's UNet work -- the conditional half (context 77 + 768 tokens) and the unconditional half (77 + 256),
    both camera conditioned (Pluecker rows, register tokens, the 16384 / 4096 / 1024 / 256-token epipolar masks) -- run
    as the product runs it (`apply_model_pair`, native packed masks) against `oracle.unet_oracle.unet_forward` on the
    host with the same weights, inputs and fundamental matrices.  Stated tolerance: rel-L2 <= 2.0e-2 per half (measured 1.55 ... 1.60e-2).
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

REL_L2, MAX_REL = 2.0e-2, 4e-2       # round 4: tightened to 1.25x / 1.6x the measured 1.60e-2 / 2.5e-2 (were 2.5e-2 / 8e-2)
TRAJ_REL_L2 = 4e-2          # x_t after each of the first three steps of the 25-step schedule (eta = 1, injected noise): the stated trajectory
                          # tolerance of tests/test_trajectory_gpu.py (guidance 7.5 amplifies the per-forward error of 1.6e-2); measured 2.0e-2 / 2.5e-2 / ...



class _FullSize:
    """The benchmark model and clip (bench.build_model / bench.synthetic_inputs) next to the fp32 oracle on the host: built once
    for the tests of this module (1.5 B parameters)."""

    def __init__(self):
        import bench
        from camc2v_amd import configs
        from oracle import geometry_oracle, unet_oracle
        self.dev = torch.device("cuda:0")
        torch.set_grad_enabled(False)
        self.model = bench.build_model(self.dev)
        self.cond, self.uncond, self.fs, self.x_T, _ = bench.synthetic_inputs(self.model, self.dev)
        unet = self.model.model.diffusion_model
        self.sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
        cam = self.cond["camera_condition"]
        F = cam["fundamental"].float().cpu()
        masks = {d: geometry_oracle.epipolar_mask(F, 256 // d, 256 // d, d) for d in (8, 16, 32, 64)}
        self.cam_cpu = dict(pluker_embedding_features=[f.float().cpu() for f in cam["pluker_embedding_features"]],
                            sample_locs_dict=masks, add_type=cam["add_type"])
        self.cfg = configs.UNET_256
        self.unet_forward = unet_oracle.unet_forward
        torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))

    def hip_pair(self, x, t):
        uc = dict(self.uncond, camera_condition=dict(self.cond["camera_condition"], is_uc=True))
        e_c, e_uc = self.model.apply_model_pair(x, t, self.cond, uc, fs=self.fs, enable_camera_condition=True)
        return e_c.float().cpu(), e_uc.float().cpu()

    def oracle_pair(self, x_cpu, t_cpu):
        """(e_c, e_uc) of the oracle for latents x [1, 4, 16, 32, 32] at timestep t: two fp32 forwards on the host cores."""
        xin = torch.cat([x_cpu, self.cond["c_concat"][0].float().cpu()], 1)
        out = []
        for ctx in (self.cond["c_crossattn"][0], self.uncond["c_crossattn"][0]):
            out.append(self.unet_forward(self.sd, self.cfg, xin, t_cpu, ctx.float().cpu(), self.fs.cpu(), self.cam_cpu, origin_h=256))
        return out


@pytest.fixture(scope="module")
def full():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    f = _FullSize()
    yield f
    del f.model
    torch.cuda.empty_cache()


def _check_pair(full, got, ref, what):
    for g, r, half in zip(got, ref, ("conditional (ctx 845)", "unconditional (ctx 333)")):
        l2 = ((g - r).norm() / r.norm()).item()
        mx = ((g - r).abs().max() / r.abs().max()).item()
        print(f"[parity] full-size camera forward {what}, {half} half vs fp32 oracle: rel_l2={l2:.3e} max_rel={mx:.3e}")
        assert torch.isfinite(g).all() and l2 <= REL_L2 and mx <= MAX_REL, (what, half, l2, mx)


@pytest.mark.parametrize("timestep", [439, 39])
def test_full_size_camera_cfg_step_vs_oracle(full, timestep):
    """One CFG step's UNet work in the middle (t = 439) and at the end (t = 39, the last step) of the 25-step schedule; the first
    step (t = 999) is the first step of the trajectory test below."""
    t = torch.full((1,), timestep, dtype=torch.long, device=full.dev)
    t0 = time.perf_counter()
    ref = full.oracle_pair(full.x_T.float().cpu(), t.cpu())
    print(f"[parity] oracle: two forwards in {time.perf_counter() - t0:.1f} s")
    _check_pair(full, full.hip_pair(full.x_T, t), ref, f"t={timestep}")


def test_full_size_three_step_trajectory_vs_ddim_oracle(full):
    """The first three steps (t = 999, 959, 919) of the headline 25-step schedule -- CFG 7.5, guidance_rescale 0.7, eta = 1 with the
    N(0,1) draws injected -- through the product's sampler (`DDIMSampler.p_sample_ddim`: batched cond+uncond forward, fused
    guidance + rescale + update) against `oracle.ddim_oracle.cfg_ddim_update` around oracle forwards, each side evolving its own
    latents.  Stated tolerance: rel-L2 of x_t <= 4e-2 after every step (measured 2.0e-2, 2.5e-2, ...); the noise predictions of the first step (t = 999, the
    step where |eps| and the timestep embedding are largest) are held to the single-step tolerance."""
    from camc2v_amd.sampler import DDIMSampler
    from oracle import ddim_oracle
    dev = full.dev
    sampler = DDIMSampler(full.model)
    sampler.make_schedule(25, ddim_discretize="uniform_trailing", ddim_eta=1.0, verbose=False)
    tab = ddim_oracle.ddim_tables(25, 1.0)
    steps = np.flip(tab["timesteps"])
    assert [int(s) for s in steps[:3]] == [999, 959, 919]
    g = torch.Generator().manual_seed(20230211)
    zs = [torch.randn(full.x_T.shape, generator=g) for _ in range(3)]
    x_hip, x_ref = full.x_T.float().contiguous(), full.x_T.float().cpu()
    for i in range(3):
        index = 24 - i
        t = torch.full((1,), int(steps[i]), dtype=torch.long, device=dev)
        if i == 0:
            got0 = full.hip_pair(x_hip, t)
        uc = dict(full.uncond)
        x_hip, _ = sampler.p_sample_ddim(x_hip, full.cond, t, index=index, unconditional_guidance_scale=7.5,
                                         unconditional_conditioning=uc, guidance_rescale=0.7, noise=zs[i].to(dev), fs=full.fs,
                                         enable_camera_condition=True)
        e_c, e_uc = full.oracle_pair(x_ref, t.cpu())
        if i == 0:
            _check_pair(full, got0, (e_c, e_uc), "t=999")
        x_ref, _, _ = ddim_oracle.cfg_ddim_update(x_ref, e_c, e_uc, zs[i], tab["alphas"][index], tab["alphas_prev"][index],
                                                  tab["sigmas"][index], tab["sqrt_one_minus_alphas"][index], 7.5, 0.7)
        got = x_hip.float().cpu()
        l2 = ((got - x_ref).norm() / x_ref.norm()).item()
        print(f"[parity] full-size trajectory: x after step {i + 1}/25 (t = {int(steps[i])}) vs ddim_oracle: rel_l2={l2:.3e}")
        assert torch.isfinite(got).all() and l2 <= TRAJ_REL_L2, (i, l2)


def _row_popcounts(bits):
    """int32 [L, words] (device) -> int64 numpy [L]."""
    b = bits.contiguous().cpu().numpy().view(np.uint8)
    return np.unpackbits(b, axis=-1).sum(-1).astype(np.int64)


def test_hip_mask_bits_equal_the_reference_masks_full_size(golden_dir):
    """`ccv_epipolar_mask_bits` on the reference's own F at 256 x 256 px, every attention resolution including the
    32x32-latent mask of the headline config (L = 16384), against what the reference's get_epipolar_mask produced for the
    same F (reference-held fixtures: per-row popcounts in geometry.npz; packed positions at d = 16 / 32 / 64 and the SHA-256
    of the packed mask at every resolution in geometry_bits.npz, oracle/gen_golden_geometry_bits.py).  Index work: the bar is
    bit-exact -- 0 flipped bits, identical positions; the patch-order emission must hold the same bits per row."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import hashlib
    from camc2v_amd import ops
    fx = np.load(os.path.join(golden_dir, "geometry.npz"))
    fb = np.load(os.path.join(golden_dir, "geometry_bits.npz"))
    F = torch.from_numpy(fx["F256"]).cuda()
    for d in (8, 16, 32, 64):
        H = 256 // d
        ref = fx[f"mask256_d{d}_popcount_rows"].astype(np.int64)[0]
        bits = ops.epipolar_mask_bits(F, 16, H, H, d)[0]
        got = _row_popcounts(bits[0])
        flips = int(np.abs(got - ref).sum())
        raw = np.ascontiguousarray(bits[0].contiguous().cpu().numpy()).view(np.uint8)     # little-endian words = packbits(little) bytes
        digest = hashlib.sha256(raw.tobytes()).digest()
        same = digest == fb[f"mask256_d{d}_sha256"].tobytes()
        print(f"[parity] HIP mask d={d} (L={16 * H * H}): |popcount delta| = {flips} of {int(ref.sum())} set bits, "
              f"SHA-256 of the packed positions {'==' if same else '!='} the reference's")
        assert flips == 0, (d, flips)
        assert same, f"d={d}: packed mask positions differ from the reference's"
        if d >= 16:
            assert np.array_equal(raw, fb[f"mask256_d{d}_bits"][0]), f"d={d}: positions"
        if ops.patch_order_ok(H, H):
            # patch order permutes rows (and bit columns) within each frame: frame f, pixel (r, c) sits at
            # f*HW + patch*32 + (r%4)*8 + c%8; patch = (r//4)*(W//8) + c//8, or -- even numbers of patch rows and columns --
            # 4 * quad + 2 * (patch row % 2) + patch column % 2 with the 2x2 quads of patches row-major (csrc/ccv_common.h: ccv_patch_row)
            pbits = ops.epipolar_mask_bits(F, 16, H, H, d, patch_order=True)[0]
            gp = _row_popcounts(pbits[0])
            r, c = np.meshgrid(np.arange(H), np.arange(H), indexing="ij")
            pr, pc = r // 4, c // 8
            if (H // 8) % 2 == 0 and (H // 4) % 2 == 0:
                patch = ((pr // 2) * (H // 16) + pc // 2) * 4 + (pr % 2) * 2 + pc % 2
            else:
                patch = pr * (H // 8) + pc
            pos = patch * 32 + (r % 4) * 8 + c % 8
            perm = (np.arange(16)[:, None] * H * H + pos.reshape(-1)[None]).reshape(-1)
            assert np.array_equal(gp[perm], got), f"d={d}: patch-order rows hold different bits than raster-order rows"
