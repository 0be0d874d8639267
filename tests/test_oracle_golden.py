"""Pin the CPU oracle against fixtures produced by RUNNING THE REFERENCE
(oracle/gen_golden.py, build container only).  CPU-only, no reference needed."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ddim_oracle, geometry_oracle, unet_oracle
from oracle.golden_inputs import SEED, SMALL_CFG, checksum, small_inputs

torch.set_grad_enabled(False)


def _unbits(bits, L):
    return torch.from_numpy(np.unpackbits(bits, axis=-1, bitorder="little")[..., :L].astype(bool))


@pytest.fixture(scope="module")
def small(golden_dir):
    fx = dict(np.load(os.path.join(golden_dir, "unet_small.npz")))
    man = json.load(open(os.path.join(golden_dir, "unet_small_manifest.json")))
    sd = unet_oracle.seeded_state_dict(man, SEED)
    inp = small_inputs()
    assert checksum(inp["ctx_pf"]) == pytest.approx(float(fx["ctx_pf_checksum"]), abs=1e-6)
    assert checksum(inp["ctx_rep"]) == pytest.approx(float(fx["ctx_rep_checksum"]), abs=1e-6)
    for f, c in zip(inp["feats"], fx["feat_checksum"]):
        assert checksum(f) == pytest.approx(float(c), abs=1e-6)
    assert np.array_equal(inp["x"].numpy(), fx["x"])
    L = {8: 16 * 64, 16: 16 * 16, 32: 16 * 4, 64: 16}
    masks = {d: _unbits(fx[f"mask_d{d}_bits"], L[d]) for d in L}
    cam = dict(pluker_embedding_features=inp["feats"], sample_locs_dict=masks, add_type="add_to_main_branch")
    return fx, sd, inp, cam


def _close(a, b, tol=2e-4):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= tol * max(ref, 1e-6), f"max abs err {err:.3e} vs ref absmax {ref:.3e}"


def test_full_manifest_matches_survey(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "unet_full_manifest.json")))
    assert len(man) == 1660
    assert sum(int(np.prod(s)) for s in man.values()) == 1500881876
    assert man["input_blocks.1.0.temopral_conv.conv1.2.weight"] == [320, 320, 3, 1, 1]
    assert man["init_attn.0.proj_in.weight"] == [512, 320, 1]
    assert man["input_blocks.1.2.transformer_blocks.0.epipolar.epipolar_attn.register_tokens"] == [1, 4, 320]


def test_topology_matches_reference_probe():
    from oracle.golden_inputs import FULL_CFG
    topo = unet_oracle.unet_topology(FULL_CFG)
    assert topo["input_ds"] == [1, 1, 1, 1, 2, 2, 2, 4, 4, 4, 8, 8]
    assert topo["output_ds"] == [8, 8, 8, 4, 4, 4, 2, 2, 2, 1, 1, 1]
    n_sp = sum(l[0] == "spatial" for blk in topo["input"] + [topo["middle"]] + topo["output"] for l in blk)
    assert n_sp == 16


def test_unet_no_camera_per_frame_context(small):
    fx, sd, inp, _ = small
    y = unet_oracle.unet_forward(sd, SMALL_CFG, inp["x"], inp["t"], inp["ctx_pf"], inp["fs"], None)
    _close(y, fx["y_nocam_pf"])


def test_unet_camera_repeat_context(small):
    fx, sd, inp, cam = small
    y = unet_oracle.unet_forward(sd, SMALL_CFG, inp["x"], inp["t"], inp["ctx_rep"], inp["fs"], cam)
    _close(y, fx["y_cam_rep"])


def test_unet_camera_per_frame_context(small):
    fx, sd, inp, cam = small
    y = unet_oracle.unet_forward(sd, SMALL_CFG, inp["x"], inp["t"], inp["ctx_pf"], inp["fs"], cam)
    _close(y, fx["y_cam_pf"])


def test_unet_camera_other_add_type_and_nomask(small):
    fx, sd, inp, cam = small
    one = lambda t: t[:1]
    cam1 = dict(cam, pluker_embedding_features=[one(f) for f in inp["feats"]],
                sample_locs_dict={d: one(m) for d, m in cam["sample_locs_dict"].items()})
    y = unet_oracle.unet_forward(sd, SMALL_CFG, one(inp["x"]), one(inp["t"]), one(inp["ctx_rep"]), one(inp["fs"]),
                                 dict(cam1, add_type="add_into_temporal_attn"))
    _close(y, fx["y_cam_other_addtype"])
    y = unet_oracle.unet_forward(sd, SMALL_CFG, one(inp["x"]), one(inp["t"]), one(inp["ctx_rep"]), one(inp["fs"]),
                                 dict(cam1, sample_locs_dict=None))
    _close(y, fx["y_cam_nomask"])


def test_unet_default_fs(small):
    fx, sd, inp, _ = small
    y = unet_oracle.unet_forward(sd, SMALL_CFG, inp["x"][:1], inp["t"][:1], inp["ctx_pf"][:1], None, None)
    _close(y, fx["y_default_fs"])


def test_unet_medium_fixture(golden_dir):
    """model_channels 128, 16x16 latents (well conditioned): the tight-tolerance fixture of the GPU tests."""
    from oracle.golden_inputs import MEDIUM_CFG, medium_inputs
    fx = np.load(os.path.join(golden_dir, "unet_medium.npz"))
    man = json.load(open(os.path.join(golden_dir, "unet_medium_manifest.json")))
    sd = unet_oracle.seeded_state_dict(man, SEED)
    inp = medium_inputs()
    assert checksum(inp["ctx_rep"]) == pytest.approx(float(fx["ctx_rep_checksum"]), abs=1e-6)
    F = torch.from_numpy(fx["F128"])
    masks = {d: geometry_oracle.epipolar_mask(F, 128 // d, 128 // d, d) for d in (8, 16, 32, 64)}
    assert [int(masks[d].sum()) for d in (8, 16, 32, 64)] == list(fx["mask_popcount"])
    cam = dict(pluker_embedding_features=inp["feats"], sample_locs_dict=masks, add_type="add_to_main_branch")
    y = unet_oracle.unet_forward(sd, MEDIUM_CFG, inp["x"], inp["t"], inp["ctx_rep"], inp["fs"], cam, origin_h=128)
    _close(y, fx["y_cam_rep"])
    y = unet_oracle.unet_forward(sd, MEDIUM_CFG, inp["x"], inp["t"], inp["ctx_pf"], inp["fs"], None)
    _close(y, fx["y_nocam_pf"])


def test_ddim_tables(golden_dir):
    fx = np.load(os.path.join(golden_dir, "ddim.npz"))
    assert np.allclose(ddim_oracle.alphas_cumprod().astype(np.float32), fx["alphas_cumprod"], rtol=1e-6)
    for eta in (0.0, 1.0):
        tab = ddim_oracle.ddim_tables(25, eta)
        tag = f"eta{int(eta)}"
        assert np.array_equal(tab["timesteps"], fx[f"timesteps_{tag}"])
        assert list(tab["timesteps"][:3]) == [39, 79, 119] and tab["timesteps"][-1] == 999
        np.testing.assert_allclose(tab["alphas"].numpy(), fx[f"ddim_alphas_{tag}"], rtol=1e-6)
        np.testing.assert_allclose(tab["alphas_prev"].numpy(), fx[f"ddim_alphas_prev_{tag}"], rtol=1e-6)
        np.testing.assert_allclose(tab["sigmas"].numpy(), fx[f"ddim_sigmas_{tag}"], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(tab["sqrt_one_minus_alphas"].numpy(), fx[f"ddim_sqrt_one_minus_alphas_{tag}"], rtol=1e-6)
    assert np.array_equal(ddim_oracle.ddim_timesteps("uniform", 50), fx["timesteps_uniform50"])


def test_cfg_ddim_step(golden_dir):
    fx = np.load(os.path.join(golden_dir, "ddim.npz"))
    tab = ddim_oracle.ddim_tables(25, 1.0)
    x, e_c, e_uc = (torch.from_numpy(fx[k]) for k in ("step_x", "step_e_c", "step_e_uc"))
    for index in (24, 7, 0):
        z = torch.from_numpy(fx[f"step{index}_noise"])
        x_prev, pred_x0, _ = ddim_oracle.cfg_ddim_update(
            x, e_c, e_uc, z, tab["alphas"][index], tab["alphas_prev"][index], tab["sigmas"][index],
            tab["sqrt_one_minus_alphas"][index], 7.5, 0.7)
        _close(x_prev, fx[f"step{index}_x_prev"], 1e-5)
        _close(pred_x0, fx[f"step{index}_pred_x0"], 1e-5)
    z = torch.from_numpy(fx["noguid_noise"])
    x_prev, _, _ = ddim_oracle.cfg_ddim_update(x, e_c, None, z, tab["alphas"][3], tab["alphas_prev"][3],
                                               tab["sigmas"][3], tab["sqrt_one_minus_alphas"][3], 1.0, 0.0)
    _close(x_prev, fx["noguid_x_prev"], 1e-5)


def test_cfg_ddim_step_camera_guidance(golden_dir):
    """Camera guidance (third forward, camera_cfg != 1; constant and cosine weight) against the reference sampler run by
    oracle/gen_golden_camcfg.py."""
    fx = np.load(os.path.join(golden_dir, "ddim_camera_cfg.npz"))
    tab = ddim_oracle.ddim_tables(25, 1.0)
    x, e_c, e_uc, e_nc = (torch.from_numpy(fx[k]) for k in ("x", "e_c", "e_uc", "e_nc"))
    for scheduler, camera_cfg, index in (("constant", 2.0, 20), ("cosine", 1.5, 20), ("cosine", 3.0, 2)):
        tag = f"{scheduler}_{camera_cfg:g}_{index}"
        z, t = torch.from_numpy(fx[f"{tag}_noise"]), torch.from_numpy(fx[f"{tag}_t"])
        nb = z.shape[0]
        x_prev, pred_x0, _ = ddim_oracle.cfg_ddim_update(
            x[:nb], e_c[:nb], e_uc[:nb], z, tab["alphas"][index], tab["alphas_prev"][index], tab["sigmas"][index],
            tab["sqrt_one_minus_alphas"][index], 7.5, 0.7, e_nc=e_nc[:nb], camera_cfg=camera_cfg,
            camera_weight=ddim_oracle.camera_cfg_weight(t, scheduler))
        _close(x_prev, fx[f"{tag}_x_prev"], 1e-5)
        _close(pred_x0, fx[f"{tag}_pred_x0"], 1e-5)
    # without enable_camera_condition the term is off: plain guidance
    x_prev, _, _ = ddim_oracle.cfg_ddim_update(x, e_c, e_uc, torch.from_numpy(fx["disabled_noise"]), tab["alphas"][2],
                                               tab["alphas_prev"][2], tab["sigmas"][2], tab["sqrt_one_minus_alphas"][2], 7.5, 0.7)
    _close(x_prev, fx["disabled_x_prev"], 1e-5)


def test_ddim_three_step_trajectory(small):
    fx, sd, inp, cam = small
    cc = torch.from_numpy(fx["traj_c_concat"])
    assert np.array_equal(inp["c_concat"].numpy(), fx["traj_c_concat"])

    def cond(x, t):
        return unet_oracle.unet_forward(sd, SMALL_CFG, torch.cat([x, cc], 1), t, inp["ctx_rep"], inp["fs"], cam)

    def uncond(x, t):  # the sampler copies the camera dict into the uncond branch (ddim.py:258-260)
        return unet_oracle.unet_forward(sd, SMALL_CFG, torch.cat([x, cc], 1), t, inp["ctx_pf"], inp["fs"], cam)

    noises = list(torch.from_numpy(fx["traj_noises"]))
    x0, _ = ddim_oracle.ddim_sample(cond, uncond, torch.from_numpy(fx["traj_x_T"]), 3, 1.0, 7.5, 0.7, noises)
    _close(x0, fx["traj_x0"], 5e-4)


def test_geometry_pose_chain(golden_dir):
    """w2c -> relative c2w -> pairs -> F.  The pose chain is inverse/matmul on 4x4s whose
    last-ulp rounding depends on tensor strides (LAPACK/BLAS paths), so it is pinned to 1e-4
    relative; the mask test below starts from the reference's own F and is bit exact."""
    fx = np.load(os.path.join(golden_dir, "geometry.npz"))
    w2c = torch.from_numpy(fx["w2c"])
    assert torch.allclose(geometry_oracle.synthetic_trajectory(1, 16), w2c)
    noise = torch.from_numpy(fx["perturb_noise"])
    idx = torch.zeros(1, dtype=torch.long)
    rel = geometry_oracle.relative_c2w(w2c, idx)
    np.testing.assert_allclose(rel.numpy(), fx["rel64"], rtol=1e-5, atol=1e-6)
    for px in (64, 256):
        F, masks = geometry_oracle.camera_masks(torch.from_numpy(fx[f"K{px}"]), w2c, idx, px, px, perturb_noise=noise)
        np.testing.assert_allclose(F.numpy(), fx[f"F{px}"], rtol=1e-4, atol=1e-8)
        if px == 256:  # tolerance-budgeted: threshold flips caused by the 1-ulp F differences
            for d, m in masks.items():
                ref = fx[f"mask256_d{d}_popcount_rows"].astype(np.int64)
                got = m.sum(-1).numpy().astype(np.int64)
                assert np.abs(got - ref).sum() <= 2e-5 * ref.sum(), d


def test_geometry_masks_bit_exact(golden_dir):
    """F (the reference's own, post-perturbation) -> boolean masks: bit exact."""
    fx = np.load(os.path.join(golden_dir, "geometry.npz"))
    F64, F256 = torch.from_numpy(fx["F64"]), torch.from_numpy(fx["F256"])
    for d in (8, 16, 32, 64):
        m = geometry_oracle.epipolar_mask(F64, 64 // d, 64 // d, d)
        assert list(m.shape) == list(fx[f"mask64_d{d}_shape"])
        nflip = int(np.unpackbits(geometry_oracle.pack_mask_bits(m) ^ fx[f"mask64_d{d}_bits"]).sum())
        assert nflip == 0, f"d={d}: {nflip} mask bits differ"
        m = geometry_oracle.epipolar_mask(F256, 256 // d, 256 // d, d)
        assert np.array_equal(m.sum(-1).to(torch.int32).numpy(), fx[f"mask256_d{d}_popcount_rows"]), d


def test_geometry_mask_positions_full_size(golden_dir):
    """The 256 x 256 px masks, positions and not only row counts (tests/golden/geometry_bits.npz: the reference's packed
    masks at d = 16 / 32 / 64 and the SHA-256 of the packed mask at every resolution, the 16384^2 one included)."""
    import hashlib
    fx = np.load(os.path.join(golden_dir, "geometry.npz"))
    fb = np.load(os.path.join(golden_dir, "geometry_bits.npz"))
    F256 = torch.from_numpy(fx["F256"])
    for d in (8, 16, 32, 64):
        bits = geometry_oracle.pack_mask_bits(geometry_oracle.epipolar_mask(F256, 256 // d, 256 // d, d))
        assert hashlib.sha256(bits.tobytes()).digest() == fb[f"mask256_d{d}_sha256"].tobytes(), d
        if d >= 16:
            assert np.array_equal(bits, fb[f"mask256_d{d}_bits"]), d


def test_vae_decode_oracle_vs_reference_fixture(golden_dir):
    """First-stage decoder restatement (oracle/vae_oracle.py) against the reference's AutoencoderKL.decode run by
    oracle/gen_golden_vae.py: full decode, the activation after the middle block, and the clip-level entry."""
    from oracle import vae_oracle as vo
    fx = dict(np.load(os.path.join(golden_dir, "vae_small.npz")))
    man = json.load(open(os.path.join(golden_dir, "vae_small_manifest.json")))
    sd = unet_oracle.seeded_state_dict(man, int(fx["seed"]), std=float(fx["std"]))
    z = torch.from_numpy(fx["z"])
    _close(vo.decode(sd, vo.SMALL_DDCONFIG, z), fx["y"])
    zq = torch.nn.functional.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = torch.nn.functional.conv2d(zq, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = vo.resnet_block(sd, "decoder.mid.block_2", vo.attn_block(sd, "decoder.mid.attn_1", vo.resnet_block(sd, "decoder.mid.block_1", h)))
    _close(h, fx["mid"])
    _close(vo.decode_first_stage(sd, vo.SMALL_DDCONFIG, torch.from_numpy(fx["z5"])), fx["y5"])
    # encoder side: posterior parameters and a sample with the fixture's noise draw
    mom = vo.encode_moments(sd, vo.SMALL_DDCONFIG, torch.from_numpy(fx["img"]))
    _close(mom, fx["moments"])
    _close(vo.posterior_sample(mom, torch.from_numpy(fx["enc_noise"])), fx["enc_sample"])


def test_cross_normalization_oracle_vs_reference_fixture(golden_dir):
    """CrossNormalization restatement against the reference's module run by oracle/gen_golden_crossnorm.py: the two call
    forms after the adaptor (per-frame and whole-clip statistics) and the self-referenced form."""
    from oracle import adaptor_oracle as ao
    fx = dict(np.load(os.path.join(golden_dir, "crossnorm_small.npz")))
    lat, z_cond = torch.from_numpy(fx["lat"]), torch.from_numpy(fx["z_cond"])
    B, T, D, H, W = 2, 4, 4, 8, 8
    x_st = lat.reshape(B, T, H, W, D).permute(0, 1, 4, 2, 3)
    for got, key in ((ao.cross_normalization(x_st, z_cond[:, None]), "y_st"), (ao.cross_normalization(lat[:, None], z_cond), "y_tok"),
                     (ao.cross_normalization(x_st), "y_self"),
                     (ao.cross_normalize_adaptor_output(lat, z_cond, T, H, W, "spatio_temporal"), "y_st")):
        assert (got - torch.from_numpy(fx[key])).abs().max().item() < 1e-6, key
    tok = ao.cross_normalize_adaptor_output(lat, z_cond, T, H, W, "token")
    want = torch.from_numpy(fx["y_tok"]).reshape(B, T, H, W, D).permute(0, 1, 4, 2, 3)
    assert (tok - want).abs().max().item() < 1e-6


def test_adaptor_oracle_vs_reference_fixture(golden_dir):
    """MultiLatentEpipolarAdaptor restatement (oracle/adaptor_oracle.py) against the reference's module run by
    oracle/gen_golden_adaptor.py (masked with a registers-only row, and unmasked)."""
    from oracle import adaptor_oracle as ao
    fx = dict(np.load(os.path.join(golden_dir, "adaptor_small.npz")))
    man = json.load(open(os.path.join(golden_dir, "adaptor_small_manifest.json")))
    sd = unet_oracle.seeded_state_dict(man, int(fx["seed"]), std=float(fx["std"]))
    x = torch.from_numpy(fx["x"])
    mask = _unbits(fx["mask"], x.shape[1])
    _close(ao.adaptor_forward(sd, ao.SMALL_CFG, x, mask), fx["y"])
    _close(ao.adaptor_forward(sd, ao.SMALL_CFG, x, None), fx["y_nomask"])


def test_resampler_oracle_vs_reference_fixture(golden_dir):
    """Resampler restatement (oracle/resampler_oracle.py) against the reference's module run by
    oracle/gen_golden_resampler.py."""
    from oracle import resampler_oracle as ro
    fx = dict(np.load(os.path.join(golden_dir, "resampler_small.npz")))
    man = json.load(open(os.path.join(golden_dir, "resampler_small_manifest.json")))
    sd = unet_oracle.seeded_state_dict(man, int(fx["seed"]), std=float(fx["std"]))
    _close(ro.resampler_forward(sd, ro.SMALL_CFG, torch.from_numpy(fx["x"])), fx["y"])


def test_ray_condition_oracle_vs_reference_fixture(golden_dir):
    """ray_condition restatement against the reference's method (oracle/gen_golden_pose.py), both embeddings."""
    from oracle import pose_oracle as po
    fx = dict(np.load(os.path.join(golden_dir, "pose_small.npz")))
    K, c2w = torch.from_numpy(fx["K"]), torch.from_numpy(fx["c2w"])
    H, W = fx["plucker"].shape[-2:]
    _close(po.ray_condition(K, c2w, H, W, plucker=True), fx["plucker"], 1e-5)
    _close(po.ray_condition(K, c2w, H, W, plucker=False), fx["ray"], 1e-5)


def test_conditional_epipolar_mask_oracle_vs_reference_fixture(golden_dir):
    """Target x context fundamental matrices and mask (the adaptor's mask) against the reference's sub-functions."""
    fx = dict(np.load(os.path.join(golden_dir, "pose_small.npz")))
    F = geometry_oracle.conditional_fundamental(torch.from_numpy(fx["cond_K"]), torch.from_numpy(fx["cond_w2c"]),
                                                torch.from_numpy(fx["cond_w2c_ctx"]), torch.from_numpy(fx["cond_index"]))
    _close(F, fx["cond_F"], 1e-4)
    shape = tuple(int(v) for v in fx["cond_mask_shape"])
    ref = _unbits(fx["cond_mask"], shape[-1])
    assert torch.equal(geometry_oracle.epipolar_mask(torch.from_numpy(fx["cond_F"]), 8, 8, 8), ref)


def test_oracle_sampler_follows_reference_25_step_trajectory(golden_dir):
    """The oracle UNet + oracle sampler against the first three steps of the REFERENCE's 25-step trajectories
    (tests/golden/traj_medium.npz, oracle/gen_golden_traj.py): camera + CFG 7.5 + rescale, and the plain CFG-off run."""
    from oracle.golden_inputs import MEDIUM_CFG, medium_inputs
    fx = np.load(os.path.join(golden_dir, "traj_medium.npz"))
    med = np.load(os.path.join(golden_dir, "unet_medium.npz"))
    man = json.load(open(os.path.join(golden_dir, "unet_medium_manifest.json")))
    sd = unet_oracle.seeded_state_dict(man, SEED)
    inp = medium_inputs()
    F = torch.from_numpy(med["F128"])
    masks = {d: geometry_oracle.epipolar_mask(F, 128 // d, 128 // d, d) for d in (8, 16, 32, 64)}
    cam = dict(pluker_embedding_features=inp["feats"], sample_locs_dict=masks, add_type="add_to_main_branch")
    keep = [int(i) for i in fx["keep_steps"]]
    assert keep[:3] == [0, 1, 2]

    def eps(ctx, camera, state):
        return lambda x, t: unet_oracle.unet_forward(state, MEDIUM_CFG, torch.cat([x, inp["c_concat"]], 1), t, ctx, inp["fs"],
                                                     camera, origin_h=128)

    class _Stop(Exception):
        pass

    def run(tag, seed_key, apply_c, apply_uc, scale, rescale):
        torch.manual_seed(int(fx[seed_key]))
        zs = [torch.randn(1, 4, 16, 16, 16) for _ in range(25)]
        assert np.allclose([checksum(z) for z in zs], fx[f"{tag}_noise_checksum"], atol=1e-6)
        seen = []

        def counted(fn):
            def wrapped(x, t):
                if len(seen) == 3 and fn is apply_c:
                    raise _Stop
                if fn is apply_c:
                    seen.append(x)
                return fn(x, t)
            return wrapped

        try:
            ddim_oracle.ddim_sample(counted(apply_c), counted(apply_uc) if apply_uc else None, inp["x_T"], 25, 1.0, scale, rescale, zs)
        except _Stop:
            pass
        # seen[i] is the latent entering step i = the latent after step i-1
        for i in (1, 2):
            _close(seen[i], fx[f"{tag}_x_steps"][i - 1], 5e-4)

    run("cam", "noise_seed_cam", eps(inp["ctx_rep"], cam, sd), eps(inp["ctx_pf"], cam, sd), 7.5, 0.7)
    sd_plain = {k: v for k, v in sd.items() if ".pluker_projection." not in k and ".epipolar." not in k}
    assert len(sd_plain) == int(fx["dc_num_keys"])
    run("dc", "noise_seed_dc", eps(inp["ctx_pf"], None, sd_plain), None, 1.0, 0.0)


def test_per_op_oracle_vs_reference_module_fixtures(golden_dir):
    """every op of the hot path, restated in oracle/unet_oracle.py, against the output of the REFERENCE'S OWN
    module on the same seeded weights and inputs (tests/golden/ops_medium.npz, oracle/gen_golden_ops.py): GroupNorm32, LayerNorm,
    ResBlock (+ TemporalConvBlock), TemporalConvBlock, Down / Upsample, SpatialTransformer, self / cross attention with 77, 77 + 16 and
    77 + 768 context tokens, GEGLU feed-forward, the camera-patched TemporalTransformer (both add_types, Pluecker features, epipolar
    mask), Epipolar at Lq = 1024 and 256 with register tokens, masked and unmasked."""
    from oracle import ops_fixture
    fx, sd, inp, masks = ops_fixture.load(golden_dir)
    got = ops_fixture.oracle_outputs(sd, inp, masks)
    ys = [k for k in fx if not k.startswith(("checksum_", "mask_")) and k not in ("F64", "perturb_noise")]
    assert sorted(ys) == sorted(got), (sorted(ys), sorted(got))
    for k in ys:
        a, b = got[k], torch.from_numpy(fx[k])
        assert tuple(a.shape) == tuple(b.shape), k
        err = ((a - b).norm() / b.norm()).item()
        assert err <= 2e-5, f"{k}: rel-L2 {err:.3e}"
        _close(a, b)


def test_zero_terminal_snr_betas_vs_reference(golden_dir):
    """rescale_betas_zero_snr (lvdm/models/utils_diffusion.py:112-144): the product's restatement against the reference's betas."""
    from camc2v_amd.sampler import rescale_zero_terminal_snr
    from oracle import sampler_cases
    fx = np.load(os.path.join(golden_dir, "ddim_branches.npz"))
    got = rescale_zero_terminal_snr(sampler_cases.schedule())
    assert np.allclose(got, fx["betas_zero_snr"], rtol=1e-12, atol=1e-15)
    assert np.allclose(got, 1.0 - (1.0 - sampler_cases.schedule(zero_snr=True)), rtol=1e-12, atol=1e-15)
    assert abs(np.cumprod(1.0 - got)[-1]) < 1e-12                  # zero terminal SNR
