"""per-stage time of the k-split 64 x 64 kernel in the split modes: one exact round of tiles (4096 x 256), growing K"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")


def t_us(M, N, K, mode, tb=True):
    kn.gemm_switches(bf16x=mode)
    A = torch.randn(M, K, device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev) * 0.06
    Cs = [torch.zeros(M, N, device=dev) for _ in range(4)]
    def body():
        for i in range(16):
            kn.gemm(A, B, Cs[i % 4], trans_b=tb)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body()
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 160 * 1e6


for M in (4096, 8192):
    for K in (256, 512, 1024, 2048, 4096, 8192):
        r = {m: t_us(M, 256, K, m) for m in (0, 6, 2)}
        print("M=%d N=256 K=%5d  f32 %7.2f us  bf16x6 %7.2f us (%5.1f TF)  f16x3 %7.2f us (%5.1f TF)   stages/wave %3d" % (
            M, K, r[0], r[6], 2.0 * M * 256 * K / r[6] / 1e6, r[2], 2.0 * M * 256 * K / r[2] / 1e6, K // 128), flush=True)
kn.gemm_switches(bf16x=6)
