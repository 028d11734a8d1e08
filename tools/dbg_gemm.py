import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K) in [(2400, 256, 256), (64, 64, 64), (64, 64, 128), (32, 32, 64), (96, 32, 64)]:
    A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
    C = torch.zeros(M, N, device=dev)
    kn.gemm(A, B, C, trans_a=True)
    ref = (A.t().double() @ B.double())
    err = (C.double() - ref).abs()
    t = err.view(M // 32, 32, N // 32, 32).amax(dim=(1, 3))
    bad = (t > 1e-3).nonzero()
    print(M, N, K, "bad tiles:", bad.shape[0], "of", t.numel(), bad[:10].tolist())
    if bad.shape[0]:
        i, j = bad[0].tolist()
        e = err[i*32:(i+1)*32, j*32:(j+1)*32]
        print(" bad rows in first bad tile:", (e.amax(1) > 1e-3).nonzero().flatten().tolist())
        print(" bad cols in first bad tile:", (e.amax(0) > 1e-3).nonzero().flatten().tolist())
