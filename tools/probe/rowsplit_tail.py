"""The 4800 / 4864-row x 256-column x K = 1024 products (FFN2 forward, dX of FFN1) run as 256 tiles of 64 x 64 (one round) plus
a remainder of 44 / 48 tiles that takes a second, mostly empty round.  Here: the remainder split 2 / 4 / 8 ways along K
(atomic accumulation onto zeroed rows) inside the same grouped launch.  usage: python tools/probe/rowsplit_tail.py"""
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

for M, K, tb in ((4864, 1024, True), (4800, 1024, True), (4800, 1024, False)):
    N = 256
    As = [torch.randn(M, K, device=dev) for _ in range(4)]
    B = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    Cs = [torch.zeros(M, N, device=dev) for _ in range(4)]
    ref = torch.empty(M, N, device=dev)
    kn.gemm(As[0], B, ref, trans_b=tb, bias=bias, residual=res)
    def base():
        for i in range(16):
            kn.gemm(As[i % 4], B, Cs[i % 4], trans_b=tb, bias=bias, residual=res)
    line = "M=%d K=%d %s: row split as today %.2f us" % (M, K, "NT" if tb else "NN", run(base, 16))
    cut = 4096
    for sk in (2, 4, 8):
        def split():
            for i in range(16):
                A, C = As[i % 4], Cs[i % 4]
                with kn.phase():
                    kn.gemm(A[:cut], B, C[:cut], trans_b=tb, bias=bias, residual=res[:cut], row0=-1)
                    kn.gemm(A[cut:], B, C[cut:], trans_b=tb, bias=bias, residual=res[cut:], split_k=sk, row0=cut)
        t = run(split, 16)
        # correctness of one call (the timed loop accumulates onto old values)
        C = torch.zeros(M, N, device=dev)
        with kn.phase():
            kn.gemm(As[0][:cut], B, C[:cut], trans_b=tb, bias=bias, residual=res[:cut], row0=-1)
            kn.gemm(As[0][cut:], B, C[cut:], trans_b=tb, bias=bias, residual=res[cut:], split_k=sk, row0=cut)
        err = float((C - ref).abs().max() / ref.abs().max())
        line += " | tail split_k=%d %.2f us (rel err %.1e)" % (sk, t, err)
    print(line)
