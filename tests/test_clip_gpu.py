"""OpenCLIP text / image embedders on the HIP kernels against the CPU oracle (oracle/clip_oracle.py; PARITY UNPINNED against
open_clip itself, see there).  Stated tolerance (bf16 GEMM / attention operands against fp32): rel-L2 <= 1.5e-2 of the output
tokens, printed per case."""
import gzip

import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL = 1.5e-2
SMALL = dict(embed_dim=64, vision=dict(image_size=56, layers=2, width=320, head_width=80, patch_size=14),
             text=dict(context_length=77, vocab_size=600, width=128, heads=2, layers=3))
# the real widths / head counts / sequence lengths of ViT-H-14 with few layers (random weights; depth is repetition)
WIDE = dict(embed_dim=1024, vision=dict(image_size=224, layers=2, width=1280, head_width=80, patch_size=14),
            text=dict(context_length=77, vocab_size=2048, width=1024, heads=16, layers=3))


def _seed_(module, std, seed):
    g = torch.Generator().manual_seed(seed)
    for name, p in module.named_parameters():
        p.copy_(torch.randn(p.shape, generator=g) * std)
        if name.endswith(("ln_1.weight", "ln_2.weight", "ln_pre.weight", "ln_final.weight", "ln_post.weight")):
            p.add_(1.0)


def _rel(got, want, what):
    got, want = got.float().cpu(), want.float()
    assert torch.isfinite(got).all()
    l2 = ((got - want).norm() / want.norm()).item()
    print(f"[parity] {what}: rel_l2={l2:.3e}")
    assert l2 <= TOL, f"{what}: rel-L2 {l2:.3e} > {TOL}"


@pytest.mark.parametrize("cfg,std", [(SMALL, 0.05), (WIDE, 0.02)], ids=["small", "vit_h_widths"])
def test_text_embedder_vs_oracle(cfg, std):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from camc2v_amd import clip
    from oracle import clip_oracle as co
    t = clip.FrozenOpenCLIPEmbedder(layer="penultimate", cfg=cfg)
    _seed_(t, std, 1)
    sd = {k: v.clone() for k, v in t.model.state_dict().items()}
    tokens = torch.randint(0, cfg["text"]["vocab_size"], (3, 77), generator=torch.Generator().manual_seed(2))
    want = co.text_tokens(sd, tokens, heads=cfg["text"]["heads"], layer_idx=1)
    t = t.to("cuda:0")
    got = t.encode(tokens)
    assert got.shape == (3, 77, cfg["text"]["width"]) and got.dtype == torch.float32
    _rel(got, want, f"text tower width {cfg['text']['width']} x {cfg['text']['heads']} heads, penultimate layer")
    # causality: a change of the last token leaves every earlier position untouched
    tokens2 = tokens.clone()
    tokens2[:, -1] = (tokens2[:, -1] + 1) % cfg["text"]["vocab_size"]
    got2 = t.encode(tokens2)
    assert torch.equal(got2[:, :-1], got[:, :-1]) and not torch.equal(got2[:, -1], got[:, -1])
    last = clip.FrozenOpenCLIPEmbedder(layer="last", cfg=cfg).to("cuda:0")
    last.load_state_dict({k: v for k, v in t.state_dict().items()})
    _rel(last.encode(tokens), co.text_tokens(sd, tokens, heads=cfg["text"]["heads"], layer_idx=0), "text tower, last layer")


@pytest.mark.parametrize("cfg,std", [(SMALL, 0.05), (WIDE, 0.02)], ids=["small", "vit_h_widths"])
def test_image_embedder_vs_oracle(cfg, std):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from camc2v_amd import clip
    from oracle import clip_oracle as co
    v = clip.FrozenOpenCLIPImageEmbedderV2(cfg=cfg)
    _seed_(v, std, 3)
    sd = {k: p.clone() for k, p in v.model.state_dict().items()}
    side = cfg["vision"]["image_size"]
    frames = torch.rand(2, 3, side + 32, side + 32, generator=torch.Generator().manual_seed(4)) * 2 - 1     # [-1, 1], resized down like 256 -> 224
    want = co.vision_tokens(sd, co.preprocess(frames, side), heads=cfg["vision"]["width"] // 80, patch=14)
    v = v.to("cuda:0")
    got = v(frames.to("cuda:0"))
    L = (side // 14) ** 2 + 1
    assert got.shape == (2, L, cfg["vision"]["width"]) and got.dtype == torch.float32
    _rel(got, want, f"vision tower width {cfg['vision']['width']}, {L} tokens, heads of 80")
    assert torch.equal(v.encode(frames.to("cuda:0")), got)


def test_captions_and_frames_through_the_model_hooks(tmp_path, monkeypatch):
    """`get_learned_conditioning` (ddpm3d.py:600-611) and the image-token hook with the embedders attached: strings -> tokenizer
    (synthetic merges file via CCV_CLIP_BPE) -> text tower; frames -> preprocess -> vision tower -> Resampler."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from camc2v_amd import clip
    from camc2v_amd.models import DynamiCrafter
    from oracle import clip_oracle as co
    from oracle.golden_inputs import SMALL_CFG
    path = tmp_path / "merges.txt.gz"
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version: test\nc a\nca m\ncam e\ncame r\ncamer a</w>\n")
    monkeypatch.setenv("CCV_CLIP_BPE", str(path))
    cfg = dict(SMALL, text=dict(SMALL["text"], vocab_size=512 + 5 + 2))
    model = DynamiCrafter({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": dict(SMALL_CFG)}, linear_start=0.00085,
                          linear_end=0.012, channels=4, image_size=[8, 8], temporal_length=16,
                          cond_stage_config={"target": "lvdm.modules.encoders.condition.FrozenOpenCLIPEmbedder", "params": {"layer": "penultimate", "cfg": cfg}},
                          img_cond_stage_config={"target": "lvdm.modules.encoders.condition.FrozenOpenCLIPImageEmbedderV2", "params": {"cfg": cfg}},
                          image_proj_stage_config={"target": "lvdm.modules.encoders.resampler.Resampler", "params": dict(
                              dim=128, depth=1, dim_head=64, heads=2, num_queries=4, embedding_dim=320, output_dim=128, ff_mult=2, video_length=16)})
    assert model.build_feeders() == ["image_proj_model"]                      # the ~1 B-parameter embedders only on request
    assert sorted(model.build_feeders(encoders=True)) == ["cond_stage_model", "embedder"]
    _seed_(model.cond_stage_model, 0.05, 5)
    _seed_(model.embedder, 0.05, 6)
    sd_t = {k: p.clone() for k, p in model.cond_stage_model.model.state_dict().items()}
    sd_v = {k: p.clone() for k, p in model.embedder.model.state_dict().items()}
    model = model.to("cuda:0").eval()
    emb = model.get_learned_conditioning(["a camera", ""])
    tok = clip.SimpleTokenizer(str(path))(["a camera", ""])
    assert tok[0, :4].tolist() == [tok[0, 0].item(), tok[0, 1].item(), 512 + 4, 518] and tok[1, :2].tolist() == [517, 518]
    _rel(emb, co.text_tokens(sd_t, tok, heads=2, layer_idx=1), "get_learned_conditioning(strings)")
    frames = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(7)) * 2 - 1
    tokens = model._image_tokens({}, "image_clip_tokens", frames.to("cuda:0"))
    _rel(tokens, co.vision_tokens(sd_v, co.preprocess(frames, 56), heads=4, patch=14), "model._image_tokens(frames)")
    ctx = model._project_image_tokens(tokens)
    assert ctx.shape == (2, 64, 128) and torch.isfinite(ctx).all()
