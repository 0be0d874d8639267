import os, sys, torch
sys.path.insert(0, os.getcwd())
from camc2v_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(64, 64, device=dev)
y = torch.empty(64, 64, device=dev, dtype=torch.bfloat16)
def chain(n):
    for _ in range(n):
        ops.cast_bf16(x)
for n in (100, 400):
    chain(3); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): chain(2)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): chain(n)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"graph of {n} tiny kernels: {e0.elapsed_time(e1) / n * 1e3:.2f} us per kernel")
# torch native tiny op
def chain2(n):
    for _ in range(n): x.add_(1.0)
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): chain2(2)
torch.cuda.current_stream().wait_stream(side)
with torch.cuda.graph(g): chain2(200)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"graph of 200 torch add_: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per kernel")
