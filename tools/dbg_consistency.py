import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camc2v_amd import ops, pack
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
def rel(a, b): return ((a.float()-b.float()).norm()/b.float().norm()).item()
torch.manual_seed(0)
# GroupNorm: doubling the batch must not change per-instance results
for (rows_pi, C, inst) in ((64, 64, 32), (16, 128, 32), (4, 256, 32), (1, 256, 32), (1024, 64, 2), (256, 128, 2), (64, 256, 2), (16, 256, 2)):
    for f32 in (True, False):
        x = (torch.randn(inst * rows_pi, C, device=dev) * 2 + 0.3).to(torch.float32 if f32 else torch.bfloat16)
        g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        y1 = ops.groupnorm(x, g, b, instances=inst, eps=1e-5, silu=True)
        y2 = ops.groupnorm(torch.cat([x, x]), g, b, instances=2 * inst, eps=1e-5, silu=True)
        y1b = ops.groupnorm(x, g, b, instances=inst, eps=1e-5, silu=True)
        ref = torch.nn.functional.silu(torch.nn.functional.group_norm(x.float().reshape(inst, rows_pi, C).permute(0, 2, 1), 32, g, b, 1e-5)).permute(0, 2, 1).reshape(-1, C)
        print(f"GN rows/inst={rows_pi:5d} C={C:4d} inst={inst:3d} f32={f32}: vs ref {rel(y1, ref):.2e}  doubled {rel(y2[:len(y1)], y1):.2e} {rel(y2[len(y1):], y1):.2e} rerun {rel(y1b, y1):.2e}")
# GEMM: doubling M
for (M, N, K, taps) in ((128, 256, 256, 9), (256, 256, 256, 9), (32, 256, 256, 9), (64, 256, 512, 9), (512, 128, 128, 9), (128, 256, 256, 3), (128, 256, 1024, 1), (128, 512, 256, 1)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, taps * K, device=dev) * 0.05).to(torch.bfloat16)
    kw = {}
    if taps == 9:
        h = 4; frames = M // 16
        kw = dict(gather=ops.GATHER_CONV3X3, conv=(4, 4, 4, 4, 1, 0))
    elif taps == 3:
        kw = dict(gather=ops.GATHER_TCONV3, tconv=(16, M // 32))
    y1 = ops.gemm(a, w, k=K, taps=taps, out_f32=True, **kw)
    y2 = ops.gemm(torch.cat([a, a]), w, k=K, taps=taps, out_f32=True, **kw)
    print(f"GEMM M={M} N={N} K={K} taps={taps}: doubled {rel(y2[:M], y1):.2e} {rel(y2[M:], y1):.2e}")
