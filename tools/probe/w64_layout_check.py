"""forced k-split 64 x 64 kernel on every layout pair against fp64 (debugging aid for its load addressing)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
kn.gemm_switches(tile=4, bf16x=int(os.environ.get("W64_BF", "6")))
for (M, N, K) in [(256, 256, 256), (2400, 256, 256), (320, 192, 1024), (256, 256, 96)]:
    for ta in (False, True):
        for tb in (False, True):
            g = torch.Generator().manual_seed(M + N + K)
            A = torch.randn((K, M) if ta else (M, K), generator=g).to(dev)
            B = torch.randn((N, K) if tb else (K, N), generator=g).to(dev)
            C = torch.zeros(M, N, device=dev)
            kn.gemm(A, B, C, trans_a=ta, trans_b=tb)
            ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
            err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
            bad = (C.double() - ref).abs() > 1e-4 * ref.abs().max()
            rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
            print("%4d x %4d x %4d %s%s  err %.2e  bad rows %s cols %s" % (M, N, K, "T" if ta else "N", "T" if tb else "N", err,
                  (rows[:4], len(rows)), (cols[:4], len(cols))), flush=True)
