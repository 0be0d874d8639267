"""Per-op parity of the HIP path against the REFERENCE'S OWN modules (tests/golden/ops_medium.npz written by
oracle/gen_golden_ops.py, which runs the reference's ResBlock / TemporalConvBlock / SpatialTransformer / CrossAttention / FeedForward /
camera-patched TemporalTransformer / Epipolar / norms on seeded weights and inputs).  Every case goes through the module-level entry
point of camc2v_amd/unet.py that the UNet itself calls (forward_rows / run / forward) and is held to a rel-L2 bound next to the
max-norm bound: bf16 MFMA operands, fp32 accumulation, fp16 stream."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

REL_L2 = 6e-3        # bf16 operand rounding through one block (measured 1.5e-3 ... 4e-3); the whole-network fixtures state 2.5e-2
MAX_REL = 2.5e-2


def _check(got, ref, what, rel=REL_L2, mx=MAX_REL):
    got, ref = got.float().cpu(), torch.as_tensor(ref).float()
    assert tuple(got.shape) == tuple(ref.shape), f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), what
    e2 = ((got - ref).norm() / ref.norm()).item()
    em = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"[per-op] {what}: rel-L2 {e2:.2e}, max/absmax {em:.2e}")
    assert e2 <= rel, f"{what}: rel-L2 {e2:.3e} > {rel}"
    assert em <= mx, f"{what}: max err / absmax {em:.3e} > {mx}"


@pytest.fixture(scope="module")
def net(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from camc2v_amd import unet as U
    from oracle import ops_fixture
    from oracle.golden_inputs import MEDIUM_CFG, OPS_NAMES, OPS_PX
    from utils.utils import instantiate_from_config
    fx, sd, inp, masks = ops_fixture.load(golden_dir)
    unet = instantiate_from_config({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": MEDIUM_CFG})
    unet.enable_camera_conditioning(dict(origin_h=OPS_PX, origin_w=OPS_PX, is_3d_full_attn=False, num_register_tokens=4,
                                         attention_resolution=[8, 4, 2, 1], compression_factor=1))
    unet.epipolar_origin_h = OPS_PX
    unet.load_state_dict(sd, strict=True)
    unet = unet.cuda().eval()
    mods = dict(unet.named_modules())
    return dict(U=U, unet=unet, m={k: mods[v] for k, v in OPS_NAMES.items()}, fx=fx, inp=inp, masks=masks)


def rows4(x, dtype):     # [(b t), C, h, w] -> [(b t h w), C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous().cuda().to(dtype)


def back4(r, n, h, w):   # rows -> [(b t), C, h, w]
    return r.float().reshape(n, h, w, -1).permute(0, 3, 1, 2)


def rows5(x, dtype):     # [b, C, t, h, w] -> [(b t h w), C]
    return x.permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1]).contiguous().cuda().to(dtype)


def back5(r, b, t, h, w):
    return r.float().reshape(b, t, h, w, -1).permute(0, 4, 1, 2, 3)


def test_norms(net):
    """a15: GroupNorm32 (fp32 statistics, eps 1e-5) and LayerNorm on the fp16 stream; bf16 outputs."""
    from camc2v_amd import ops
    U, m, fx, inp = net["U"], net["m"], net["fx"], net["inp"]
    gn = m["gn"]
    x = rows4(inp["gn_x"], U.STREAM)
    y = ops.groupnorm(x, gn.weight.float().contiguous(), gn.bias.float().contiguous(), instances=inp["Tf"], eps=1e-5, silu=False)
    _check(back4(y, inp["Tf"], 8, 8), fx["gn_y"], "GroupNorm32", 4e-3, 1.5e-2)
    ln = m["ln"]
    xl = inp["ln_x"].reshape(-1, 128).contiguous().cuda().to(U.STREAM)
    yl = ops.layernorm(xl, ln.weight.float().contiguous(), ln.bias.float().contiguous(), eps=ln.eps)
    _check(yl.reshape(inp["Tf"], 64, 128), fx["ln_y"], "LayerNorm", 4e-3, 1.5e-2)


def test_resblock_and_temporal_conv(net):
    """a6 + a7: ResBlock 128 -> 256 (GN+SiLU+conv3x3, + emb, GN+SiLU+conv3x3, 1x1 skip) with its TemporalConvBlock; a7 alone."""
    U, m, fx, inp = net["U"], net["m"], net["fx"], net["inp"]
    T = inp["T"]
    blk = m["res"]
    lin = blk.emb_layers[1]
    emb_all = F.linear(F.silu(inp["res_emb"].cuda()), lin.weight.float(), lin.bias.float()).contiguous()     # [b = 1, Cout] fp32
    keep = blk.emb_slice
    blk.emb_slice = (0, lin.out_features)
    try:
        y = blk.forward_rows(rows4(inp["res_x"], U.STREAM), emb_all, U.Geom(1, T, 4, 4))
    finally:
        blk.emb_slice = keep
    _check(back4(y, T, 4, 4), fx["res_y"], "ResBlock + TemporalConvBlock")
    y = m["tconv"].forward_rows(rows5(inp["tconv_x"], U.STREAM), U.Geom(1, T, 4, 4))
    _check(back5(y, 1, T, 4, 4), fx["tconv_y"], "TemporalConvBlock")


def test_down_and_upsample(net):
    """a8: 3x3 stride-2 convolution; nearest 2x upsampling fused into the 3x3 convolution's gather."""
    U, m, fx, inp = net["U"], net["m"], net["fx"], net["inp"]
    n = inp["Tf"]
    y, g = m["down"].forward_rows(rows4(inp["down_x"], U.STREAM), U.Geom(1, n, 8, 8))
    assert (g.h, g.w) == (4, 4)
    _check(back4(y, n, 4, 4), fx["down_y"], "Downsample")
    y, g = m["up"].forward_rows(rows4(inp["up_x"], U.STREAM), U.Geom(1, n, 4, 4))
    assert (g.h, g.w) == (8, 8)
    _check(back4(y, n, 8, 8), fx["up_y"], "Upsample")


def test_spatial_transformer(net):
    """a9: GroupNorm -> proj_in -> [LN, self attention, LN, text + gated image cross attention, LN, GEGLU] -> proj_out -> + x.
    The fixture's frames carry their own 77 + 16 context tokens each: handed over as Tf one-frame clips (context rule 77 + 16 t)."""
    U, unet, m, fx, inp = net["U"], net["unet"], net["m"], net["fx"], net["inp"]
    n = inp["Tf"]
    st = m["st"]
    blocks = [blk for mod in unet.modules() if isinstance(mod, U.SpatialTransformer) for blk in mod.transformer_blocks]
    nclips, per_block = unet._context_groups(inp["st_ctx"].cuda(), 1)
    assert nclips == n
    per_block = list(per_block)
    groups = [per_block[blocks.index(b)] for b in st.transformer_blocks]
    y = st.forward_rows(rows4(inp["st_x"], U.STREAM), U.Geom(n, 1, 8, 8), groups)
    _check(back4(y, n, 8, 8), fx["st_y"], "SpatialTransformer")


def test_attention_and_feed_forward_modules(net):
    """a10 / a14: CrossAttention.forward (self; text only; text + 16 image tokens; text + 768 image tokens) and FeedForward.forward."""
    m, fx, inp = net["m"], net["fx"], net["inp"]
    x = inp["xattn_x"].cuda()
    _check(m["sattn"](x), fx["sattn_y"], "self attention module")
    for L in (77, 93, 845):
        _check(m["xattn"](x, context=inp[f"xattn_ctx{L}"].cuda()), fx[f"xattn_y{L}"], f"cross attention module, context {L}")
    _check(m["ff"](inp["ff_x"].cuda()), fx["ff_y"], "GEGLU feed-forward module")


@pytest.mark.parametrize("add_type", ["add_to_main_branch", "add_into_temporal_attn"])
def test_temporal_transformer_camera_block(net, add_type):
    """a11 / a12: GroupNorm(clip) -> proj_in -> [LN; Pluecker projection + epipolar attention + frame attention; LN, frame attention; LN,
    GEGLU] -> proj_out -> + x, with Pluecker features and the epipolar mask (C = 256, 4 heads, L = 16 x 4 x 4)."""
    U, unet, m, fx, inp, masks = net["U"], net["unet"], net["m"], net["fx"], net["inp"], net["masks"]
    T = inp["T"]
    cc = dict(pluker_embedding_features=[inp["tt_p"].cuda()], sample_locs_dict={16: masks[16].cuda()}, add_type=add_type)
    cam = unet._camera_inputs(cc, 1, T, 4, 4)
    lvl = dict(rows=cam["rows"][0], mask=cam["masks"][16], add_type=add_type)
    y = m["tt"].forward_rows(rows5(inp["tt_x"], U.STREAM), U.Geom(1, T, 4, 4), lvl)
    _check(back5(y, 1, T, 4, 4), fx[f"tt_y_{add_type}"], f"TemporalTransformer camera block ({add_type})")


@pytest.mark.parametrize("name,hw,key", [("epi1024", 8, 8), ("epi256", 4, 16), ("epi256_nomask", 4, None)])
def test_epipolar_module(net, name, hw, key):
    """a13: Epipolar (to_q / to_k / to_v over [4 register tokens ; T*h*w tokens], boolean epipolar mask, to_out) at Lq = 1024 (patch-ordered
    bits, 2 heads) and Lq = 256 (raster order, 4 heads), and unmasked."""
    from camc2v_amd import ops
    U, m, fx, inp, masks = net["U"], net["m"], net["fx"], net["inp"], net["masks"]
    mod = m[name.split("_")[0]]
    x = inp[name.split("_")[0] + "_x"]                       # [B, T, C, H, W]
    B, T, C, H, W = x.shape
    src = x.permute(0, 1, 3, 4, 2).reshape(-1, C).contiguous().cuda().to(torch.bfloat16)
    packed = None
    if key is not None:
        perm = (H * W, W) if ops.patch_order_ok(H, W) else None
        mp = ops.pack_mask(masks[key].cuda(), perm)
        packed = (mp[0], mp[1], 1, perm, mp.wave_bits, mp.group_order)
    stream = torch.zeros(B * T * H * W, C, dtype=torch.float32, device="cuda")
    mod.run(src, stream, U.Geom(B, T, H, W), packed)
    y = stream.reshape(B, T, H * W, C).permute(0, 2, 1, 3).reshape(B * H * W, T, C)
    # the operand itself is rounded to bf16 here (the fixture's input is fp32): bound from the bf16-input emulation
    _check(y, fx[name + "_y" if key is not None else "epi256_y_nomask"], f"Epipolar {name}", 8e-3, 3e-2)
