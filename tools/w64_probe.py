"""Where does a stage of the split-bf16 64 x 64 k-split kernel go?  Times gemm_wstage64_kernel<.., 6> (forced tile 4) on the
step's large shapes; run under MESM_LIB_PATH = probe builds (tools/build_variant.sh <name> -DMESM_W64_NO_SPLIT | _NO_MMA |
_NO_LOAD: wrong results, timing only).  usage: python tools/w64_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from mesm_amd import kernels as kn
import px_check as P
dev = torch.device("cuda:0")
kn.gemm_switches(tile=4, bf16x=int(os.environ.get("W64_BF", "6")))
for (M, N, K, ta, tb, split) in [(4800, 256, 1024, False, True, 1), (4800, 1024, 256, False, True, 1), (4800, 256, 256, False, False, 1),
                                 (1024, 256, 4800, True, False, 4), (256, 1024, 4800, True, False, 4), (2400, 256, 2818, False, True, 1),
                                 (2400, 2818, 256, False, False, 1), (4096, 256, 1024, False, True, 1), (8192, 256, 1024, False, True, 1)]:
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev)
    Cs = [torch.zeros(M, N, device=dev) for _ in range(4)]
    def body():
        for i in range(16):
            kn.gemm(A, B, Cs[i % 4], trans_a=ta, trans_b=tb, split_k=split, accumulate=2 if split > 1 else 0)
    us = P.timed(body, 16)
    print("%5d x %4d x %4d %s%s/s%d  %7.2f us  %6.1f TF" % (M, N, K, "T" if ta else "N", "T" if tb else "N", split, us, 2.0 * M * N * K / us / 1e6), flush=True)
