"""Under-filled products (150-300 tiles of 64 x 64) split along K, partial sums by atomic adds onto C (accumulating calls:
C holds the earlier contributions, no zero-fill needed).  usage: python tools/probe/splitk_small.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
torch.manual_seed(0)

def run(body, n, reps=20):
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

for M, N, K, tb in ((2400, 256, 512, False), (2400, 256, 256, False), (4800, 256, 256, False), (4800, 256, 256, True),
                    (2400, 512, 256, True), (1024, 256, 1024, False), (2400, 256, 1024, False)):
    As = [torch.randn(M, K, device=dev) for _ in range(4)]
    B = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
    Cs = [torch.zeros(M, N, device=dev) for _ in range(4)]
    line = "M=%d N=%d K=%d %s:" % (M, N, K, "NT" if tb else "NN")
    for sk in (1, 2, 4):
        if K // sk < 128:
            continue
        def f():
            for i in range(16):
                kn.gemm(As[i % 4], B, Cs[i % 4], trans_b=tb, accumulate=1 if sk == 1 else 0, split_k=sk)
        line += "  split_k=%d %.2f us" % (sk, run(f, 16))
    print(line)
