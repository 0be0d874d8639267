"""OpenCLIP ViT-H/14 text and image towers on the HIP kernels: the conditioning encoders that
run once per clip in front of the DDIM path.

Reference call sites: ``FrozenOpenCLIPEmbedder`` (lvdm/modules/encoders/condition.py:174-235: token embedding + positional
embedding -> all but the last ``layer_idx`` residual blocks under the causal mask -> ``ln_final``; the yaml uses
``layer: penultimate``) and ``FrozenOpenCLIPImageEmbedderV2`` (:295-372: kornia resize to 224 + CLIP normalisation -> conv1
patches -> class token + positional embedding -> ``ln_pre`` -> transformer; returns all 257 tokens, no ``ln_post`` / ``proj``).
The towers themselves live in the third-party dependency ``open_clip_torch==2.22.0`` (reference requirements.txt:5), which is
absent from this image and from /root/reference: the architecture is restated from its published model definition
(``ViT-H-14``: text width 1024 / 16 heads / 24 layers / vocabulary 49408 / context 77; vision width 1280 / head width 80 /
32 layers / patch 14 / image 224; pre-LayerNorm residual blocks ``x + attn(ln_1 x)``, ``x + mlp(ln_2 x)`` with
``nn.MultiheadAttention`` and an erf-GELU MLP) with ITS parameter names, so the ``cond_stage_model.model.*`` and
``embedder.model.visual.*`` slices of a reference checkpoint load strictly.  PARITY UNPINNED: see oracle/clip_oracle.py.

Execution: fp32 residual stream in token-major rows, bf16 GEMM operands; LayerNorm -> fused QKV projection (bias in the
epilogue) -> attention -> output projection with the residual add in the epilogue -> LayerNorm -> c_fc + GELU -> c_proj +
residual.  The text tower's heads are 64 wide: ``ccv_attn_fwd`` with the causal mask as packed bits.  The vision tower's heads
are 80 wide, which the d = 64 attention kernels do not cover: per (image, head) Q K^T GEMM (heads zero-padded to 128 columns by
the packed projection weights, scale and a -inf bias on the padded key columns in the epilogue) -> ``ccv_softmax_rows`` ->
P V GEMM, like the first-stage model's attention (vae.py).  Nothing here is on the 25-step path; it runs once per clip.
"""
import gzip
import html
import os
from collections import OrderedDict
from functools import lru_cache

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, pack
from .lib import CcvError
from .unet import _Prepared, _dev_f32

VIT_H_14 = dict(embed_dim=1024,
                vision=dict(image_size=224, layers=32, width=1280, head_width=80, patch_size=14),
                text=dict(context_length=77, vocab_size=49408, width=1024, heads=16, layers=24))
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ---------------------------------------------------------------------------------------------------------------------------
# parameter containers with open_clip's names
# ---------------------------------------------------------------------------------------------------------------------------
class ResidualAttentionBlock(nn.Module):
    """ln_1, attn.{in_proj_weight, in_proj_bias, out_proj.{weight, bias}}, ln_2, mlp.{c_fc, c_proj}."""

    def __init__(self, width, heads, mlp_ratio=4.0):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_2 = nn.LayerNorm(width)
        hidden = int(width * mlp_ratio)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, hidden)), ("gelu", nn.GELU()), ("c_proj", nn.Linear(hidden, width))]))


class Transformer(nn.Module, _Prepared):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.grad_checkpointing = False
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads) for _ in range(layers)])
        self._register_load_state_dict_pre_hook(lambda *a, **k: self.invalidate())

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate()
        return out

    def _pack(self):
        C, H = self.width, self.heads
        d = C // H
        if d != 64 and not (d % 16 == 0 and d <= 128):
            raise CcvError(f"head width {d}: supported are 64 (attention kernel) and multiples of 16 up to 128 (GEMM path)")
        dp = 64 if d == 64 else 128          # head width of q / k in the packed projection (zero columns beyond d)
        layers = []
        for blk in self.resblocks:
            w, b = blk.attn.in_proj_weight.detach().float(), blk.attn.in_proj_bias.detach().float()
            if d == 64:
                w_qkv, b_qkv = pack.pack_linear(w), pack.pad_bias(b)
            else:                            # [q heads padded to dp | k heads padded to dp | v heads d wide]
                wq, wk, wv = w[:C].view(H, d, C), w[C:2 * C].view(H, d, C), w[2 * C:]
                bq, bk, bv = b[:C].view(H, d), b[C:2 * C].view(H, d), b[2 * C:]
                zw, zb = wq.new_zeros(H, dp - d, C), bq.new_zeros(H, dp - d)
                w_qkv = pack.pack_linear(torch.cat([torch.cat([wq, zw], 1).reshape(H * dp, C), torch.cat([wk, zw], 1).reshape(H * dp, C), wv], 0))
                b_qkv = pack.pad_bias(torch.cat([torch.cat([bq, zb], 1).reshape(-1), torch.cat([bk, zb], 1).reshape(-1), bv], 0))
            layers.append(dict(
                g1=_dev_f32(blk.ln_1.weight), b1=_dev_f32(blk.ln_1.bias), g2=_dev_f32(blk.ln_2.weight), b2=_dev_f32(blk.ln_2.bias),
                w_qkv=w_qkv, b_qkv=b_qkv, w_o=pack.pack_linear(blk.attn.out_proj.weight), b_o=pack.pad_bias(blk.attn.out_proj.bias),
                w_fc=pack.pack_linear(blk.mlp.c_fc.weight), b_fc=pack.pad_bias(blk.mlp.c_fc.bias),
                w_pr=pack.pack_linear(blk.mlp.c_proj.weight), b_pr=pack.pad_bias(blk.mlp.c_proj.bias),
                eps1=blk.ln_1.eps, eps2=blk.ln_2.eps))
        return dict(layers=layers, d=d, dp=dp)

    # ---- attention forms ---------------------------------------------------------------------------------------------
    def _attend64(self, qkv, B, L, mask):
        C, H = self.width, self.heads
        ld = 3 * C
        s = (L * ld, 0, ld)
        kw = dict(mask_bits=mask[0], tile_flags=mask[1], mask_nb=1) if mask is not None else {}
        return ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=L, Lk=L, q_str=s, k_str=s, v_str=s, scale=64 ** -0.5, **kw)

    def _attend_wide(self, qkv, B, L, d, dp):
        """heads of d != 64 columns, no mask: per (sample, head) S = Q K^T (K = dp, scale, -inf on the padded key columns) ->
        softmax rows -> P V^T."""
        C, H = self.width, self.heads
        Lp = (L + 63) // 64 * 64                               # key length padded for the GEMMs (K of P V, N of Q K^T)
        dev = qkv.device
        key_bias = torch.zeros(Lp, dtype=torch.float32, device=dev)
        key_bias[L:] = -1e30
        out = torch.empty((B * L, C), dtype=ops.BF16, device=dev)
        q_all = qkv[:, :H * dp]
        k_all = qkv[:, H * dp:2 * H * dp].view(B, L, H, dp)
        v_all = qkv[:, 2 * H * dp:2 * H * dp + C].view(B, L, H, d)
        k_pad = torch.zeros((B, H, Lp, dp), dtype=ops.BF16, device=dev)
        k_pad[:, :, :L] = k_all.permute(0, 2, 1, 3)
        vt_pad = torch.zeros((B, H, d, Lp), dtype=ops.BF16, device=dev)
        vt_pad[:, :, :, :L] = v_all.permute(0, 2, 3, 1)
        for b in range(B):
            rows = slice(b * L, (b + 1) * L)
            for h in range(H):
                s = ops.gemm(q_all[rows, h * dp:(h + 1) * dp], k_pad[b, h], k=dp, bias=key_bias, out_f32=True, alpha=float(d) ** -0.5)
                ops.gemm(ops.softmax_rows(s), vt_pad[b, h], out=out[rows, h * d:(h + 1) * d])
        return out

    @torch.no_grad()
    def forward_rows(self, x, B, L, mask=None, n_layers=None):
        """x fp32 [(B L), width] (updated in place and returned) through the first ``n_layers`` blocks."""
        pk = self._pk()
        d, dp = pk["d"], pk["dp"]
        if mask is not None and d != 64:
            raise CcvError("the masked form needs 64-wide heads")
        for lp in pk["layers"][:self.layers if n_layers is None else n_layers]:
            qkv = ops.gemm(ops.layernorm(x, lp["g1"], lp["b1"], eps=lp["eps1"]), lp["w_qkv"], bias=lp["b_qkv"])
            o = self._attend64(qkv, B, L, mask) if d == 64 else self._attend_wide(qkv, B, L, d, dp)
            ops.gemm(o, lp["w_o"], bias=lp["b_o"], residual=x, out_f32=True, out=x)
            h = ops.gemm(ops.layernorm(x, lp["g2"], lp["b2"], eps=lp["eps2"]), lp["w_fc"], bias=lp["b_fc"], act=ops.ACT_GELU)
            ops.gemm(h, lp["w_pr"], bias=lp["b_pr"], residual=x, out_f32=True, out=x)
        return x


class VisionTransformer(nn.Module, _Prepared):
    """open_clip VisionTransformer parameter layout (conv1 without bias, class / positional embeddings, ln_pre, transformer,
    ln_post, proj) and the attributes the reference wrapper reads (``input_patchnorm``, ``grid_size``, ``patch_size``,
    ``patch_dropout``)."""

    def __init__(self, image_size=224, patch_size=14, width=1280, layers=32, head_width=80, output_dim=1024):
        super().__init__()
        self.image_size, self.patch_size = (image_size, image_size), (patch_size, patch_size)
        self.grid_size = (image_size // patch_size, image_size // patch_size)
        self.output_dim, self.width = output_dim, width
        self.input_patchnorm = False
        scale = width ** -0.5
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid_size[0] * self.grid_size[1] + 1, width))
        self.patch_dropout = nn.Identity()
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, width // head_width)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self._register_load_state_dict_pre_hook(lambda *a, **k: self.invalidate())

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate()
        return out

    def _pack(self):
        return dict(w_patch=pack.pack_linear(self.conv1.weight.detach().float().reshape(self.width, -1)),    # [width, 3*p*p] zero padded to a multiple of 64
                    g_pre=_dev_f32(self.ln_pre.weight), b_pre=_dev_f32(self.ln_pre.bias))

    @torch.no_grad()
    def tokens(self, img):
        """img [B, 3, H, W] already resized / normalised -> fp32 [B, grid^2 + 1, width]: the transformer's output tokens
        (condition.py:344-372; no ln_post / proj)."""
        if not img.is_cuda:
            raise CcvError("VisionTransformer: the product path runs on the GPU only (see oracle/clip_oracle.py)")
        pk = self._pk()
        B, p, (gh, gw), C = img.shape[0], self.patch_size[0], self.grid_size, self.width
        if img.shape[-2:] != (gh * p, gw * p):
            raise CcvError(f"VisionTransformer: image {tuple(img.shape[-2:])}, expected {(gh * p, gw * p)}")
        patches = img.float().reshape(B, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, 3 * p * p)
        kp = pk["w_patch"].shape[1]
        a = torch.zeros((B * gh * gw, kp), dtype=ops.BF16, device=img.device)
        a[:, :3 * p * p] = patches
        emb = ops.gemm(a, pk["w_patch"], out_f32=True)[:, :C]                                   # conv1 as a GEMM over the patches
        L = gh * gw + 1
        x = torch.empty((B, L, C), dtype=torch.float32, device=img.device)
        x[:, 0] = self.class_embedding.detach().float()
        x[:, 1:] = emb.reshape(B, gh * gw, C)
        x += self.positional_embedding.detach().float()
        x = ops.layernorm_small(x.reshape(B * L, C).contiguous(), pk["g_pre"], pk["b_pre"], eps=self.ln_pre.eps)
        return self.transformer.forward_rows(x.contiguous(), B, L).reshape(B, L, C)


class CLIPModel(nn.Module):
    """The slice of open_clip's ``CLIP`` the two wrappers keep: text attributes directly on the model (token_embedding,
    positional_embedding, transformer, ln_final, text_projection, logit_scale; ``attn_mask`` is a non-persistent buffer) and
    ``visual``."""

    def __init__(self, cfg=None, text=True, visual=True):
        super().__init__()
        cfg = cfg or VIT_H_14
        t = cfg["text"]
        self.context_length, self.vocab_size = t["context_length"], t["vocab_size"]
        self.token_embedding = nn.Embedding(t["vocab_size"], t["width"])
        self.positional_embedding = nn.Parameter(torch.empty(t["context_length"], t["width"]).normal_(std=0.01))
        self.ln_final = nn.LayerNorm(t["width"])
        self.text_projection = nn.Parameter(torch.empty(t["width"], cfg["embed_dim"]).normal_(std=t["width"] ** -0.5))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.6592)
        mask = torch.empty(t["context_length"], t["context_length"]).fill_(float("-inf")).triu_(1)
        self.register_buffer("attn_mask", mask, persistent=False)
        if text:                # FrozenOpenCLIPImageEmbedderV2 deletes it (condition.py:307)
            self.transformer = Transformer(t["width"], t["layers"], t["heads"])
        if visual:              # FrozenOpenCLIPEmbedder deletes it (condition.py:189)
            v = cfg["vision"]
            self.visual = VisionTransformer(v["image_size"], v["patch_size"], v["width"], v["layers"], v["head_width"], cfg["embed_dim"])


# ---------------------------------------------------------------------------------------------------------------------------
# tokenizer (open_clip.tokenize: byte-level BPE of CLIP; the merges file is third-party data and is not shipped)
# ---------------------------------------------------------------------------------------------------------------------------
@lru_cache()
def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class SimpleTokenizer:
    """CLIP's byte-pair tokenizer over a merges file in the format of ``bpe_simple_vocab_16e6.txt.gz`` (first line a header;
    the released file contributes its first 48894 merges): lower-cased, whitespace-collapsed text -> regex pre-tokens -> bytes
    mapped to printable code points -> greedy lowest-rank merges; ids = [<start_of_text>] + tokens + [<end_of_text>],
    zero padded / truncated to the context length (the last id stays <end_of_text>)."""

    def __init__(self, bpe_path, max_merges=49152 - 256 - 2):
        import regex
        opener = gzip.open if str(bpe_path).endswith(".gz") else open
        with opener(bpe_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(m.split()) for m in lines[1:1 + max_merges] if len(m.split()) == 2]
        self.byte_encoder = _bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<start_of_text>", "<end_of_text>"]
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {"<start_of_text>": "<start_of_text>", "<end_of_text>": "<end_of_text>"}
        self.pat = regex.compile(r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
        self.sot, self.eot = self.encoder["<start_of_text>"], self.encoder["<end_of_text>"]

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = set(zip(word[:-1], word[1:]))
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text):
        text = " ".join(html.unescape(html.unescape(text)).strip().split()).lower()
        ids = []
        for token in self.pat.findall(text):
            token = "".join(self.byte_encoder[b] for b in token.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(token).split(" "))
        return ids

    def __call__(self, texts, context_length=77):
        texts = [texts] if isinstance(texts, str) else list(texts)
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > context_length:
                ids = ids[:context_length]
                ids[-1] = self.eot
            out[i, :len(ids)] = torch.tensor(ids)
        return out


def _default_tokenizer():
    path = os.environ.get("CCV_CLIP_BPE")
    if not path or not os.path.exists(path):
        raise CcvError("tokenising text needs CLIP's merges file (open_clip's bpe_simple_vocab_16e6.txt.gz, third-party data that is "
                       "not shipped here): point CCV_CLIP_BPE at it, or call the embedder with token ids [b, 77] / pass 'caption_emb' "
                       "in the batch")
    return SimpleTokenizer(path)


# ---------------------------------------------------------------------------------------------------------------------------
# the two yaml targets
# ---------------------------------------------------------------------------------------------------------------------------
class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class FrozenOpenCLIPEmbedder(AbstractEncoder):
    """Text encoder (condition.py:174-235).  ``version`` names pretrained weights open_clip would download: there is no network
    and no open_clip here, so the parameters start random and come from the model checkpoint (``cond_stage_model.model.*``)."""
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, layer="last", cfg=None):
        super().__init__()
        if layer not in self.LAYERS:
            raise ValueError(f"layer must be one of {self.LAYERS}")
        if cfg is None and arch != "ViT-H-14":
            raise NotImplementedError(f"arch {arch!r}: only ViT-H-14 is described here (pass cfg= for another geometry)")
        self.model = CLIPModel(cfg, text=True, visual=False)
        self.device, self.max_length, self.layer = device, max_length, layer
        self.layer_idx = 0 if layer == "last" else 1
        self._tokenizer = None
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p in self.parameters():
            p.requires_grad = False

    def tokenize(self, text):
        if self._tokenizer is None:
            self._tokenizer = _default_tokenizer()
        return self._tokenizer(text, self.model.context_length)

    @torch.no_grad()
    def encode_with_transformer(self, tokens):
        """token ids [B, 77] -> fp32 [B, 77, width] (condition.py:212-230)."""
        m = self.model
        dev = m.positional_embedding.device
        if dev.type != "cuda":
            raise CcvError("FrozenOpenCLIPEmbedder: the product path runs on the GPU only (see oracle/clip_oracle.py)")
        tokens = tokens.to(dev)
        B, L = tokens.shape
        x = (m.token_embedding.weight.detach().float()[tokens] + m.positional_embedding.detach().float()[:L]).reshape(B * L, -1).contiguous()
        tr = m.transformer
        mask = tr.__dict__.get("_causal")
        if mask is None or mask[0].shape[1] != L or mask[0].device != dev:
            mask = tr.__dict__["_causal"] = ops.pack_mask(torch.ones(L, L, dtype=torch.bool, device=dev).tril_()[None])
        x = tr.forward_rows(x, B, L, mask=mask, n_layers=tr.layers - self.layer_idx)
        return ops.layernorm_small(x, _dev_f32(m.ln_final.weight), _dev_f32(m.ln_final.bias), eps=m.ln_final.eps).reshape(B, L, -1)

    def forward(self, text):
        tokens = text if torch.is_tensor(text) else self.tokenize(text)
        return self.encode_with_transformer(tokens)

    def encode(self, text):
        return self(text)


def clip_preprocess(x, size=224, antialias=True):
    """[-1, 1] images [B, 3, H, W] -> CLIP-normalised [B, 3, size, size] (condition.py:327-335: kornia.geometry.resize bicubic,
    align_corners=True, antialias -> (x + 1) / 2 -> kornia.enhance.normalize).  kornia is absent here; its documented resize is
    restated: when downscaling, a separable Gaussian blur (sigma = max((factor - 1) / 2, 0.001), kernel int(max(4 sigma, 3)) made
    odd, reflect border) in front of the bicubic interpolation.  Plain torch tensor ops on the GPU (plumbing, once per clip)."""
    x = x.float()
    H, W = x.shape[-2:]
    fy, fx = H / size, W / size
    if antialias and max(fy, fx) > 1:
        sig = (max((fy - 1) / 2, 0.001), max((fx - 1) / 2, 0.001))
        ks = [int(max(2.0 * 2 * s, 3)) for s in sig]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]

        def kernel(k, s):
            t = torch.arange(k, dtype=torch.float32, device=x.device) - k // 2
            g = torch.exp(-t * t / (2 * s * s))
            return g / g.sum()
        ky, kx = kernel(ks[0], sig[0]), kernel(ks[1], sig[1])
        xp = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
        C = x.shape[1]
        xp = F.conv2d(xp, kx.view(1, 1, 1, -1).expand(C, 1, 1, -1), groups=C)
        x = F.conv2d(xp, ky.view(1, 1, -1, 1).expand(C, 1, -1, 1), groups=C)
    x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=True)
    x = (x + 1.0) / 2.0
    mean = torch.tensor(CLIP_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std


class FrozenOpenCLIPImageEmbedderV2(AbstractEncoder):
    """Image encoder (condition.py:295-372): frames [b, 3, H, W] in [-1, 1] -> tokens [b, 257, 1280] for the Resampler."""

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", freeze=True, layer="pooled", antialias=True, cfg=None):
        super().__init__()
        if cfg is None and arch != "ViT-H-14":
            raise NotImplementedError(f"arch {arch!r}: only ViT-H-14 is described here (pass cfg= for another geometry)")
        if layer == "penultimate":
            raise NotImplementedError()
        self.model = CLIPModel(cfg, text=False, visual=True)
        self.device, self.layer, self.antialias = device, layer, antialias
        self.register_buffer("mean", torch.tensor(CLIP_MEAN), persistent=False)
        self.register_buffer("std", torch.tensor(CLIP_STD), persistent=False)
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for p in self.model.parameters():
            p.requires_grad = False

    def preprocess(self, x):
        return clip_preprocess(x, self.model.visual.image_size[0], self.antialias)

    @torch.no_grad()
    def encode_with_vision_transformer(self, x):
        return self.model.visual.tokens(self.preprocess(x))

    def forward(self, image, no_dropout=False):
        return self.encode_with_vision_transformer(image)

    def encode(self, image):
        return self(image)


__all__ = ["FrozenOpenCLIPEmbedder", "FrozenOpenCLIPImageEmbedderV2", "AbstractEncoder", "CLIPModel", "Transformer", "VisionTransformer",
           "ResidualAttentionBlock", "SimpleTokenizer", "clip_preprocess", "VIT_H_14"]
