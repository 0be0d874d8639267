"""Camera feeders on the HIP kernels: ray_condition vs the reference fixture; the pose
encoder vs the oracle restatement (parity of that module is unpinned: diffusers is absent, see oracle/pose_oracle.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    assert torch.isfinite(a).all()
    return ((a - b).norm() / b.norm()).item(), ((a - b).abs().max() / b.abs().max()).item()


def test_ray_condition_vs_reference_fixture(golden_dir):
    from camc2v_amd import ops
    fx = dict(np.load(os.path.join(golden_dir, "pose_small.npz")))
    K, c2w = torch.from_numpy(fx["K"]).cuda(), torch.from_numpy(fx["c2w"]).cuda()
    H, W = fx["plucker"].shape[-2:]
    for mode in ("plucker", "ray"):
        y = ops.ray_condition(K, c2w, H, W, plucker=(mode == "plucker"))
        l2, mx = _rel(y, fx[mode])
        print(f"[parity] ray_condition {mode}: rel_l2={l2:.2e} max_rel={mx:.2e}")
        assert y.shape == fx[mode].shape and l2 < 1e-5 and mx < 1e-5


def test_pose_elementwise_kernels():
    from camc2v_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 6, 16, 24, generator=g)
    ref = F.pixel_unshuffle(x, 8).permute(0, 2, 3, 1).reshape(-1, 384).to(torch.bfloat16)
    assert torch.equal(ops.pixel_unshuffle_rows(x.cuda(), 8).cpu(), ref)
    r = torch.randn(2 * 6 * 8, 64, generator=g)
    ref = F.avg_pool2d(r.reshape(2, 6, 8, 64).permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).reshape(-1, 64)
    l2, mx = _rel(ops.avgpool2_rows(r.cuda(), 2, 6, 8), ref)
    assert l2 < 1e-6
    a, w = torch.randn(100, 64, generator=g).to(torch.bfloat16), (torch.randn(32, 64, generator=g) * 0.1).to(torch.bfloat16)
    l2, mx = _rel(ops.gemm(a.cuda(), w.cuda(), act=ops.ACT_RELU), F.relu(a.float() @ w.float().t()))
    assert l2 < 5e-3


@pytest.mark.parametrize("T,D,H", [(16, 40, 8), (16, 160, 8), (5, 8, 2), (16, 256, 1)])
def test_attention_small(T, D, H):
    """Temporal self-attention with arbitrary head width, frame-major rows [(b f hw), 3C] (fused QKV)."""
    from camc2v_amd import ops
    g = torch.Generator().manual_seed(T * 1000 + D)
    b, hw, C = 2, 6, H * D
    qkv = torch.randn(b * T * hw, 3 * C, generator=g).to(torch.bfloat16)
    ld = 3 * C
    st = (T * hw * ld, ld, hw * ld)
    qd = qkv.cuda()
    o = ops.attention_small(qd, qd[:, C:], qd[:, 2 * C:], B=b * hw, inner=hw, H=H, T=T, head_dim=D, q_str=st, k_str=st, v_str=st,
                            out=torch.empty(b * T * hw, C, dtype=torch.bfloat16, device="cuda"), o_str=(T * hw * C, C, hw * C))
    x = qkv.float().reshape(b, T, hw, 3, H, D).permute(3, 0, 2, 4, 1, 5)        # [3, b, hw, H, T, D]
    ref = torch.softmax(x[0] @ x[1].transpose(-1, -2) * D ** -0.5, -1) @ x[2]     # [b, hw, H, T, D]
    ref = ref.permute(0, 3, 1, 2, 4).reshape(b * T * hw, C)
    l2, mx = _rel(o, ref)
    print(f"[parity] attention_small T={T} D={D}: rel_l2={l2:.2e} max_rel={mx:.2e}")
    assert l2 < 8e-3 and mx < 2e-2


def _encoder_case(cfg, b, f, H, W, seed):
    from oracle import pose_oracle as po, unet_oracle
    from utils.utils import instantiate_from_config
    man = po.pose_encoder_manifest(cfg)
    sd = unet_oracle.seeded_state_dict({k: v for k, v in man.items() if not k.endswith(".pe")}, seed, std=0.05)
    m = instantiate_from_config({"target": "model.modules.camera_pose_encoder.CameraPoseEncoder", "params": dict(cfg)})
    missing = m.load_state_dict(sd, strict=False)
    assert all(k.endswith(".pe") for k in missing.missing_keys) and not missing.unexpected_keys
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(b, 6, f, H, W, generator=g)
    with torch.no_grad():
        ref = po.pose_encoder_forward(sd, cfg, x)
        with unet_oracle.operand_rounding(torch.bfloat16):
            emu = po.pose_encoder_forward(sd, cfg, x)
    got = m(x.cuda())
    assert len(got) == 4
    for i, (y, r, e) in enumerate(zip(got, ref, emu)):
        assert y.shape == r.shape and y.dtype == torch.float32
        l2, mx = _rel(y, r)
        floor, _ = _rel(e, r)
        print(f"[parity] pose feature level {i} {tuple(y.shape)}: HIP rel_l2={l2:.3e}, emulated bf16 oracle rel_l2={floor:.3e}")
        assert l2 <= 1.6 * floor + 2e-3 and l2 <= 2.5e-2
    return m, x


def test_pose_encoder_small_vs_oracle():
    from oracle import pose_oracle as po
    _encoder_case(po.SMALL_CFG, b=2, f=6, H=64, W=128, seed=3)


def test_pose_encoder_shipped_size_vs_oracle():
    from oracle import pose_oracle as po
    m, x = _encoder_case(po.FULL_CFG, b=1, f=16, H=256, W=256, seed=4)
    xd = x.cuda()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        m(xd)
    e1.record()
    torch.cuda.synchronize()
    print(f"[perf] CameraPoseEncoder b=1, 16 frames, 256x256: {e0.elapsed_time(e1) / 3:.2f} ms per call")


def test_model_level_pose_features():
    """CameraControlLVDM.build_feeders + pose_features: poses -> [b, C_i, f, h_i, w_i] for the UNet."""
    from camc2v_amd import camera
    from oracle import pose_oracle as po
    from oracle.golden_inputs import SMALL_CFG
    from utils.utils import instantiate_from_config
    model = instantiate_from_config({"target": "model.camcontexti2v.CamContextI2V", "params": dict(
        unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": SMALL_CFG},
        conditioning_key="hybrid", channels=4, image_size=[8, 8], temporal_length=16, scale_factor=0.18215,
        pose_encoder_config={"target": "model.modules.camera_pose_encoder.CameraPoseEncoder", "params": dict(po.SMALL_CFG)},
        epipolar_config=dict(origin_h=64, origin_w=64, is_3d_full_attn=False, num_register_tokens=4, attention_resolution=[8, 4, 2, 1],
                             compression_factor=1))})
    assert model.scale_factor == pytest.approx(0.18215)
    assert model.build_feeders() == ["pose_encoder"]
    model = model.cuda()
    dev = torch.device("cuda:0")
    K = torch.tensor([[32.0, 0, 32], [0, 32, 32], [0, 0, 1]], device=dev).repeat(1, 16, 1, 1)
    w2c = camera.synthetic_trajectory(1, 16, dev)
    feats = model.pose_features(K, w2c, torch.zeros(1, dtype=torch.long, device=dev), 64, 64)
    assert [tuple(f.shape) for f in feats] == [(1, 64, 16, 8, 8), (1, 128, 16, 4, 4), (1, 128, 16, 2, 2), (1, 128, 16, 1, 1)]
    assert all(torch.isfinite(f).all() for f in feats)


def test_conditional_mask_and_conv3d(golden_dir):
    """Rectangular (target x context) epipolar mask bits vs the reference fixture, and the 3x3x3 latent projection."""
    from camc2v_amd import camera, ops
    fx = dict(np.load(os.path.join(golden_dir, "pose_small.npz")))
    dev = torch.device("cuda:0")
    F = camera.conditional_fundamental(torch.from_numpy(fx["cond_K"]).to(dev), torch.from_numpy(fx["cond_w2c"]).to(dev),
                                       torch.from_numpy(fx["cond_w2c_ctx"]).to(dev), torch.from_numpy(fx["cond_index"]).to(dev))
    l2, mx = _rel(F, fx["cond_F"])
    assert l2 < 1e-4
    shape = tuple(int(v) for v in fx["cond_mask_shape"])                       # [1, 256, 192]
    mp = ops.epipolar_mask_bits(torch.from_numpy(fx["cond_F"]).to(dev), 4, 8, 8, 8)
    got = np.unpackbits(mp[0].cpu().numpy().view(np.uint8), axis=-1, bitorder="little")[..., :shape[-1]]
    ref = np.unpackbits(fx["cond_mask"], axis=-1, bitorder="little")[..., :shape[-1]]
    assert got.shape == ref.shape and np.array_equal(got, ref)
    g = torch.Generator().manual_seed(8)
    x, w, b = torch.randn(2, 4, 5, 6, 7, generator=g), torch.randn(4, 4, 3, 3, 3, generator=g) * 0.2, torch.randn(4, generator=g)
    add = torch.randn(2, 4, 6, 7, generator=g)
    ref = torch.nn.functional.conv3d(x, w, b, padding=1) + add[:, :, None]
    l2, mx = _rel(ops.conv3d_small(x.to(dev), w.to(dev), b.to(dev), add=add.to(dev)), ref)
    assert l2 < 1e-6


@pytest.mark.parametrize("cross_norm", [None, "spatio_temporal", "token"])
def test_context_concat_vs_oracle(cross_norm):
    """CamContextI2V.context_concat: adaptor over [conditioning ; context] latents with the conditional epipolar mask,
    (optional cross normalisation,) Conv3d latent projection, + conditioning latent -- against the oracle pieces composed
    the same way."""
    from camc2v_amd import camera
    from oracle import adaptor_oracle as ao, geometry_oracle as go, unet_oracle
    from oracle.golden_inputs import SMALL_CFG
    from utils.utils import instantiate_from_config
    acfg = dict(ao.FULL_CFG, query_dim=128, num_queries=64, video_length=4, depth=2)
    model = instantiate_from_config({"target": "model.camcontexti2v.CamContextI2V", "params": dict(
        unet_config={"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": SMALL_CFG},
        conditioning_key="hybrid", channels=4, image_size=[8, 8], temporal_length=4, scale_factor=0.18215,
        multi_cond_strategy="token_concat_latent_epipolar", use_zero_conv_latent_input=True,
        use_cross_normalization=cross_norm is not None, cross_normalization_mode=cross_norm or "spatio_temporal",
        multi_latent_adaptor={"target": "model.modules.adaptors.MultiLatentEpipolarAdaptor", "params": acfg})})
    assert model.build_feeders() == ["multi_cond_latent_adaptor"]
    man = {k: list(v.shape) for k, v in model.multi_cond_latent_adaptor.state_dict().items()}
    sd = unet_oracle.seeded_state_dict(man, 12, std=0.05)
    model.multi_cond_latent_adaptor.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(13)
    wproj, bproj = torch.randn(4, 4, 3, 3, 3, generator=g) * 0.1, torch.randn(4, generator=g) * 0.1
    with torch.no_grad():
        model.multi_cond_in_projection.weight.copy_(wproj)
        model.multi_cond_in_projection.bias.copy_(bproj)
    model = model.cuda()
    dev = torch.device("cuda:0")
    T, n = 4, 2
    K = torch.tensor([[32.0, 0, 32], [0, 32, 32], [0, 0, 1]]).repeat(1, T, 1, 1)
    w2c = go.synthetic_trajectory(1, T)
    w2c_ctx = go.synthetic_trajectory(1, 8)[:, [5, 7]]
    cond_idx = torch.zeros(1, dtype=torch.long)
    z_cond, z_ctx = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, n, 8, 8, generator=g)
    with torch.no_grad():
        F = go.conditional_fundamental(K, w2c, w2c_ctx, cond_idx)
        mask = go.epipolar_mask(F, 8, 8, 8)
        tokens = torch.cat([z_cond[:, :, None], z_ctx], 2).permute(0, 2, 3, 4, 1).reshape(1, -1, 4)
        lat = ao.adaptor_forward(sd, acfg, tokens, mask)
        if cross_norm is None:
            x = lat.reshape(1, T, 8, 8, 4).permute(0, 4, 1, 2, 3)
        else:
            x = ao.cross_normalize_adaptor_output(lat, z_cond, T, 8, 8, cross_norm).permute(0, 2, 1, 3, 4)   # B T D H W -> B D T H W
        ref = torch.nn.functional.conv3d(x, wproj, bproj, padding=1) + z_cond[:, :, None]
    got = model.context_concat(z_cond.to(dev), z_ctx.to(dev), K.to(dev), w2c.to(dev), w2c_ctx.to(dev), cond_idx.to(dev))
    assert got.shape == (1, 4, T, 8, 8)
    l2, mx = _rel(got, ref)
    print(f"[parity] context_concat (cross_norm={cross_norm}) vs oracle composition: rel_l2={l2:.3e} max_rel={mx:.3e}")
    assert l2 < 1.5e-2
