import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops, pack
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
if which == "conv":      # 16x16 latents, 1280 -> 640: gemm_dma_kernel<4,4,1>, 90 K-slabs
    h, cin, cout = 16, 1280, 640
    M = 32 * h * h
    a = torch.randn(M, cin, device=dev).to(torch.bfloat16)
    w = pack.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02)
    fn = lambda: ops.gemm(a, w, k=cin, taps=9, gather=ops.GATHER_CONV3X3, conv=(h, h, h, h, 1, 0))
elif which == "lin":     # (8192, 5120, 640) plain
    a = torch.randn(8192, 640, device=dev).to(torch.bfloat16)
    w = (torch.randn(5120, 640, device=dev) * 0.05).to(torch.bfloat16)
    fn = lambda: ops.gemm(a, w)
elif which == "wide":    # 32x32 latents 960 -> 320 (wide kernel)
    h, cin, cout = 32, 960, 320
    M = 32 * h * h
    a = torch.randn(M, cin, device=dev).to(torch.bfloat16)
    w = pack.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.02)
    fn = lambda: ops.gemm(a, w, k=cin, taps=9, gather=ops.GATHER_CONV3X3, conv=(h, h, h, h, 1, 0))
for _ in range(5):
    fn()
torch.cuda.synchronize()
