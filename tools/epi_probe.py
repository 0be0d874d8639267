import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import camera, ops
dev = torch.device("cuda:0")
patch = len(sys.argv) > 1 and sys.argv[1] == "patch"
b, T, px, d, H = 1, 16, 256, 8, 5
K = torch.tensor([[128.0, 0, 128], [0, 128, 128], [0, 0, 1]], device=dev).repeat(b, T, 1, 1)
w2c = camera.synthetic_trajectory(b, T, dev)
cam = camera.camera_condition(K, w2c, torch.zeros(b, dtype=torch.long, device=dev), px, px)
hh = px // d
mp = ops.epipolar_mask_bits(cam["fundamental"], T, hh, hh, d, patch_order=patch)
bits, flags = mp
L = bits.shape[1]; C = H * 64
qkv = torch.randn(2 * L, 3 * C, device=dev).to(torch.bfloat16)
kreg, vreg = torch.randn(4, C, device=dev).to(torch.bfloat16), torch.randn(4, C, device=dev).to(torch.bfloat16)
ld = 3 * C; s = (L * ld, 0, ld)
for _ in range(3):
    ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=2, inner=1, H=H, Lq=L, Lk=L, q_str=s, k_str=s, v_str=s,
                  mask_bits=bits, mask_nb=1, tile_flags=flags, wave_bits=mp.wave_bits, kreg=kreg, vreg=vreg, perm=(hh * hh, hh) if patch else None)
torch.cuda.synchronize()
