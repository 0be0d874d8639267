import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops
dev = torch.device("cuda:0")
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L, H, B = 4096, 10, 2
C = H * 64
qkv = torch.randn(B * L, 3 * C, device=dev).to(torch.bfloat16)
ld = 3 * C
s = (L * ld, 0, ld)
for _ in range(3):
    ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B=B, inner=1, H=H, Lq=L, Lk=L, q_str=s, k_str=s, v_str=s, variant=variant)
torch.cuda.synchronize()
