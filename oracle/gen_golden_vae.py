#!/usr/bin/env python
"""Generate tests/golden/vae_small.npz + manifests by RUNNING THE REFERENCE's first-stage decoder (build container
only).  TEST INFRASTRUCTURE, companion of oracle/gen_golden.py (same stand-ins: cv2, pytorch_lightning,
torchvision.utils.make_grid are imported at file top by the reference but never touched on this path).

What is run: the reference's ``AutoencoderKL`` (lvdm/models/autoencoder.py) built from the yaml's ddconfig at reduced
width, seeded weights loaded through its own ``load_state_dict``, ``decode`` on seeded latents; plus the key->shape
manifest of the full-size config for the checkpoint-layout test.  Fixtures hold tensors and scalars only.

Usage:  python oracle/gen_golden_vae.py [--out tests/golden]
"""
import argparse
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/CamContextI2V"


def _load(name):
    spec = importlib.util.spec_from_file_location(f"_ccv_oracle_{name}", os.path.join(HERE, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    args = ap.parse_args()
    repo_root = os.path.dirname(HERE)
    assert all(os.path.abspath(p or os.getcwd()) != repo_root for p in sys.path), \
        "run as `python oracle/gen_golden_vae.py`: the repo root on sys.path would shadow the reference's packages"
    gg = _load("gen_golden")
    gg._install_shims()
    vo = _load("vae_oracle")
    seeded_state_dict = _load("unet_oracle").seeded_state_dict
    torch.set_grad_enabled(False)

    from lvdm.models.autoencoder import AutoencoderKL
    assert sys.modules["lvdm.models.autoencoder"].__file__.startswith(REF)

    def build(ddconfig):
        return AutoencoderKL(ddconfig=dict(ddconfig), lossconfig={"target": "torch.nn.Identity"}, embed_dim=4).eval()

    full = build(vo.FULL_DDCONFIG)
    with open(os.path.join(args.out, "vae_full_manifest.json"), "w") as f:
        json.dump(gg.manifest_of(full), f, indent=0, sort_keys=True)
    del full

    small = build(vo.SMALL_DDCONFIG)
    man = gg.manifest_of(small)
    with open(os.path.join(args.out, "vae_small_manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    sd = seeded_state_dict(man, gg.SEED + 11, std=0.03)
    small.load_state_dict(sd, strict=True)

    g = torch.Generator().manual_seed(gg.SEED + 12)
    z = torch.randn(3, 4, 8, 8, generator=g)              # 3 frames of 8x8 latents -> 64x64 pixels
    y = small.decode(z)
    zq = small.post_quant_conv(z)
    mid = small.decoder.mid.block_2(small.decoder.mid.attn_1(small.decoder.mid.block_1(small.decoder.conv_in(zq), None)), None)
    # the clip-level entry of the sampler side: b c t h w latents, 1/scale_factor, frames as batch (ddpm3d.py:648-666)
    z5 = torch.randn(1, 4, 3, 8, 8, generator=g) * vo.SCALE_FACTOR
    frames = (z5.permute(0, 2, 1, 3, 4).reshape(3, 4, 8, 8)) * (1.0 / vo.SCALE_FACTOR)
    y5 = small.decode(frames).reshape(1, 3, 3, 64, 64).permute(0, 2, 1, 3, 4)

    # encoder side: 2 images -> posterior parameters, and a sample with a supplied noise draw
    img = torch.randn(2, 3, 64, 64, generator=g)
    post = small.encode(img)
    enc_noise = torch.randn(2, 4, 8, 8, generator=g)
    enc_sample = post.sample(enc_noise)
    err_e = (vo.encode_moments(sd, vo.SMALL_DDCONFIG, img) - post.parameters).abs().max().item()
    err_s = (vo.posterior_sample(post.parameters, enc_noise) - enc_sample).abs().max().item()
    assert err_e < 2e-4 * post.parameters.abs().max().item() and err_s < 1e-5, (err_e, err_s)

    # the restatement must agree before anything is written
    err = (vo.decode(sd, vo.SMALL_DDCONFIG, z) - y).abs().max().item()
    err5 = (vo.decode_first_stage(sd, vo.SMALL_DDCONFIG, z5) - y5).abs().max().item()
    assert err < 2e-4 * y.abs().max().item() and err5 < 2e-4 * y5.abs().max().item(), (err, err5)
    np.savez_compressed(os.path.join(args.out, "vae_small.npz"), z=z.numpy(), y=y.numpy(), mid=mid.numpy(),
                        z5=z5.numpy(), y5=y5.numpy(), img=img.numpy(), moments=post.parameters.numpy(), enc_noise=enc_noise.numpy(),
                        enc_sample=enc_sample.numpy(), seed=np.int64(gg.SEED + 11), std=np.float32(0.03))
    print(f"vae_small: y absmax {y.abs().max().item():.3f}, oracle max abs err {err:.2e} / {err5:.2e}; "
          f"{len(man)} tensors ({sum(int(np.prod(s)) for s in man.values()) / 1e6:.1f} M params)")


if __name__ == "__main__":
    main()
