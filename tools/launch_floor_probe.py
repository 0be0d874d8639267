import torch, time
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
big = torch.zeros(32768 * 320, device=dev, dtype=torch.bfloat16)
def chain(n, t):
    for _ in range(n):
        t.add_(1)
for name, t in (("64 floats", x), ("21 MB bf16", big)):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(10, t)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain(1000, t)
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5): g.replay()
        e1.record(s); s.synchronize()
        print(f"{name}: graph of 1000 dependent launches: {e0.elapsed_time(e1) / 5:.3f} us per launch", flush=True)
        e0.record(s)
        for _ in range(5): chain(1000, t)
        e1.record(s); s.synchronize()
        print(f"{name}: eager: {e0.elapsed_time(e1) / 5:.3f} us per launch", flush=True)
