"""FFN1-class products (x 256 -> 1024, PReLU + dropout epilogue, pre-activation as second output) by row count: is the third,
37 % full round of the 1,216-tile launch worth restructuring the block for?  usage: python tools/probe/ffn1_rows.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn
from mesm_amd._lib import ACT_PRELU
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

W = torch.randn(1024, 256, device=dev) * 0.05
b = torch.randn(1024, device=dev)
slope = torch.full((1,), 0.25, device=dev)
for M in (768, 2048, 4096, 4608, 4864, 6144, 8192):
    xs = [torch.randn(M, 256, device=dev) for _ in range(4)]
    zs = [torch.empty(M, 1024, device=dev) for _ in range(4)]
    hs = [torch.empty(M, 1024, device=dev) for _ in range(4)]
    def f():
        for i in range(16):
            kn.gemm(xs[i % 4], W, hs[i % 4], trans_b=True, bias=b, e_act=ACT_PRELU, slope=slope, e_drop=(0.1, 5), pre_out=zs[i % 4])
    t = run(f, 16)
    tiles = ((M + 63) // 64) * 16
    print("M=%5d  %5d tiles  %6.2f us  %5.1f TF  %.3f us per 512 tiles" % (M, tiles, t, 2.0 * M * 1024 * 256 / t / 1e6, t / (tiles / 512.0)))
