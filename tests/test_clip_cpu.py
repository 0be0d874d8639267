"""OpenCLIP ViT-H/14 embedders, CPU side: checkpoint key layout, the oracle against a tower run
through torch's own nn.MultiheadAttention / nn.LayerNorm modules (what open_clip's blocks are assembled from), the tokenizer
on a synthetic merges file, the preprocessing against an independent statement.  PARITY UNPINNED against open_clip itself
(absent here, no reference-held fixtures): see oracle/clip_oracle.py."""
import gzip
import os

import pytest
import torch

SMALL = dict(embed_dim=64, vision=dict(image_size=56, layers=2, width=320, head_width=80, patch_size=14),
             text=dict(context_length=77, vocab_size=600, width=128, heads=2, layers=3))


def test_yaml_targets_build_the_vit_h_14_checkpoint_layout():
    """cond_stage_config / img_cond_stage_config of configs/models/camcontexti2v_256.yaml:97-108 through the plugin mechanism:
    parameter names and counts of open_clip's ViT-H-14 with `del model.visual` / `del model.transformer` applied."""
    from utils.utils import instantiate_from_config
    with torch.device("meta"):
        t = instantiate_from_config({"target": "lvdm.modules.encoders.condition.FrozenOpenCLIPEmbedder", "params": {"freeze": True, "layer": "penultimate", "version": None}})
        v = instantiate_from_config({"target": "lvdm.modules.encoders.condition.FrozenOpenCLIPImageEmbedderV2", "params": {"freeze": True, "version": None}})
    ts, vs = t.state_dict(), v.state_dict()
    assert len(ts) == 6 + 24 * 12 and sum(p.numel() for p in t.parameters()) == 354_032_641      # the published text-tower size
    assert tuple(ts["model.token_embedding.weight"].shape) == (49408, 1024) and tuple(ts["model.positional_embedding"].shape) == (77, 1024)
    assert tuple(ts["model.transformer.resblocks.23.attn.in_proj_weight"].shape) == (3072, 1024)
    assert tuple(ts["model.transformer.resblocks.0.mlp.c_fc.weight"].shape) == (4096, 1024)
    assert "model.attn_mask" not in ts and t.layer_idx == 1 and not any(p.requires_grad for p in t.parameters())
    visual = {k: tuple(x.shape) for k, x in vs.items() if k.startswith("model.visual.")}
    assert len(visual) == 8 + 32 * 12 and sum(x.numel() for k, x in vs.items() if k.startswith("model.visual.")) == 632_076_800
    assert visual["model.visual.conv1.weight"] == (1280, 3, 14, 14) and visual["model.visual.positional_embedding"] == (257, 1280)
    assert visual["model.visual.proj"] == (1280, 1024) and visual["model.visual.transformer.resblocks.31.attn.in_proj_weight"] == (3840, 1280)
    assert "model.transformer.resblocks.0.ln_1.weight" not in vs and "model.token_embedding.weight" in vs
    assert v.model.visual.input_patchnorm is False and v.model.visual.grid_size == (16, 16) and v.model.visual.patch_size == (14, 14)


def _torch_tower(tr, x, mask=None, n=None):
    """open_clip's Transformer.forward on torch's own modules (LND layout)."""
    x = x.permute(1, 0, 2)
    for blk in list(tr.resblocks)[:n]:
        h = blk.ln_1(x)
        x = x + blk.attn(h, h, h, need_weights=False, attn_mask=mask)[0]
        x = x + blk.mlp(blk.ln_2(x))
    return x.permute(1, 0, 2)


def test_oracle_equals_torch_multihead_attention_towers():
    from camc2v_amd import clip
    from oracle import clip_oracle as co
    torch.manual_seed(0)
    with torch.no_grad():
        t = clip.FrozenOpenCLIPEmbedder(layer="penultimate", cfg=SMALL)
        v = clip.FrozenOpenCLIPImageEmbedderV2(cfg=SMALL)
        for m in (t, v):
            for p in m.parameters():
                p.normal_(0.0, 0.05)
        tokens = torch.randint(0, 600, (2, 77))
        m = t.model
        x = m.token_embedding(tokens) + m.positional_embedding
        want = m.ln_final(_torch_tower(m.transformer, x, m.attn_mask, n=2))          # penultimate: 2 of the 3 blocks
        got = co.text_tokens({k: p for k, p in m.state_dict().items()}, tokens, heads=2, layer_idx=1)
        assert torch.allclose(got, want, atol=2e-5, rtol=1e-4)
        img = torch.randn(2, 3, 56, 56)
        vis = v.model.visual
        y = vis.conv1(img).reshape(2, 320, -1).permute(0, 2, 1)
        y = torch.cat([vis.class_embedding.expand(2, 1, 320), y], 1) + vis.positional_embedding
        want = _torch_tower(vis.transformer, vis.ln_pre(y))
        got = co.vision_tokens(v.model.state_dict(), img, heads=4, patch=14)
        assert got.shape == (2, 17, 320) and torch.allclose(got, want, atol=2e-5, rtol=1e-4)


def test_product_path_refuses_the_cpu():
    from camc2v_amd import clip
    from camc2v_amd.lib import CcvError
    t = clip.FrozenOpenCLIPEmbedder(layer="penultimate", cfg=SMALL)
    with pytest.raises(CcvError):
        t(torch.zeros(1, 77, dtype=torch.long))
    with pytest.raises(CcvError, match="CCV_CLIP_BPE"):
        t(["a caption"])
    v = clip.FrozenOpenCLIPImageEmbedderV2(cfg=SMALL)
    with pytest.raises(CcvError):
        v(torch.zeros(1, 3, 64, 64))


def test_tokenizer_on_a_synthetic_merges_file(tmp_path):
    """Byte-level BPE mechanics (vocabulary layout, merge ranks, </w>, start / end tokens, padding, truncation)."""
    from camc2v_amd.clip import SimpleTokenizer, _bytes_to_unicode
    path = tmp_path / "merges.txt.gz"
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version: test\nh e\nl l\nhe ll\nhell o</w>\nw o\n")
    tok = SimpleTokenizer(str(path))
    b2u = _bytes_to_unicode()
    base = list(b2u.values())
    assert len(tok.encoder) == 512 + 5 + 2 and tok.sot == 517 and tok.eot == 518
    assert tok.encode("Hello") == [512 + 3]                                       # h e l l o -> he ll o</w> -> hell o</w> -> "hello</w>"
    ids = tok.encode("hello  wow!")
    assert ids == [515, tok.encoder["wo"], tok.encoder["w</w>"], tok.encoder["!</w>"]]
    assert tok.encoder["w</w>"] == 256 + base.index("w")
    out = tok(["hello", "hello " * 100])
    assert out.shape == (2, 77) and out[0, :3].tolist() == [517, 515, 518] and int(out[0, 3:].abs().sum()) == 0
    assert out[1, 0] == 517 and out[1, -1] == 518 and (out[1, 1:-1] == 515).all()  # truncated, the last id stays <end_of_text>


def test_preprocess_matches_the_independent_statement():
    from camc2v_amd.clip import clip_preprocess
    from oracle import clip_oracle as co
    g = torch.Generator().manual_seed(3)
    for hw, size in (((256, 256), 224), ((64, 80), 56), ((32, 32), 56)):           # down, anisotropic down, up (no blur)
        x = torch.rand(2, 3, *hw, generator=g) * 2 - 1
        got, want = clip_preprocess(x, size), co.preprocess(x, size)
        assert got.shape == (2, 3, size, size) and torch.allclose(got, want, atol=5e-4)   # fp32 against the fp64 statement
    x = torch.zeros(1, 3, 224, 224)                                               # mid-grey -> (0.5 - mean) / std, no resampling
    got = clip_preprocess(x, 224)[0, :, 0, 0]
    assert torch.allclose(got, (0.5 - torch.tensor([0.48145466, 0.4578275, 0.40821073])) / torch.tensor([0.26862954, 0.26130258, 0.27577711]), atol=1e-6)
