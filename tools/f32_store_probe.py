"""f32 GEMM time of large-output shapes (probe: how much of a launch is its store phase; run with MESM_LIB_PATH variants)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from mesm_amd import kernels as kn
import px_check as P
dev = torch.device("cuda:0")
for (M, N, K, ta, tb, pre) in [(4800, 1024, 256, False, True, False), (4800, 1024, 256, False, True, True), (4800, 256, 256, False, True, False),
                               (4800, 768, 256, False, True, False), (2400, 2818, 256, False, False, False), (1024, 5003, 256, False, True, False)]:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    Cs = [torch.zeros(M, N, device=dev) for _ in range(4)]
    Ps = [torch.zeros(M, N, device=dev) for _ in range(4)] if pre else None
    bias = torch.randn(N, device=dev)
    def body():
        for i in range(16):
            kn.gemm(A, B, Cs[i % 4], trans_a=ta, trans_b=tb, bias=bias, pre_out=Ps[i % 4] if pre else None,
                    e_act=kn.ACT_RELU if pre else kn.ACT_NONE)
    print(M, N, K, "pre_out" if pre else "", "%.2f us" % P.timed(body, 16), flush=True)
