import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
def run(body, n=32, reps=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): body()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n * 1e6
for rows in (4800, 2400, 1024, 320):
    D = 256
    xs = [torch.randn(rows, D, device=dev) for _ in range(8)]
    dys = [torch.randn(rows, D, device=dev) for _ in range(8)]
    g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    y, mean, rstd = kn.layernorm_fwd(xs[0], g, b)
    dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    def f():
        for i in range(32): kn.layernorm_fwd(xs[i % 8], g, b)
    def bw():
        for i in range(32): kn.layernorm_bwd(dys[i % 8], xs[i % 8], g, mean, rstd, dg, db)
    print("rows %5d: LN fwd %.2f us, LN bwd %.2f us" % (rows, run(f), run(bw)))
