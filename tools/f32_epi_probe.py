"""f32 GEMM time with bias + residual epilogues on the step's shapes (probe; run with MESM_LIB_PATH variants)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from mesm_amd import kernels as kn
import px_check as P
dev = torch.device("cuda:0")
for (M, N, K) in [(4800, 256, 1024), (4800, 256, 256), (2400, 256, 256), (1024, 256, 256), (320, 256, 256), (4800, 1024, 256)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev)
    Cs = [torch.zeros(M, N, device=dev) for _ in range(4)]
    Rs = [torch.randn(M, N, device=dev) for _ in range(4)]
    bias = torch.randn(N, device=dev)
    for mode in ("plain", "bias", "bias+res"):
        def body():
            for i in range(16):
                kn.gemm(A, B, Cs[i % 4], trans_b=True, bias=bias if mode != "plain" else None, residual=Rs[i % 4] if mode == "bias+res" else None)
        print(M, N, K, mode, "%.2f us" % P.timed(body, 16), flush=True)
