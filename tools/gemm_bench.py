"""Micro-benchmark of mesm_gemm_f32 on the hot-path shapes (back-to-back launches, HIP events),
with torch.mm (rocBLAS/hipBLASLt fp32) beside it as an on-box reference point."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn

dev = torch.device("cuda:0")
SHAPES = [  # name, M, N, K, trans_a, trans_b, split
    ("fwd d->d      ", 2400, 256, 256, False, True, 1),
    ("fwd d->F      ", 2400, 1024, 256, False, True, 1),
    ("fwd F->d      ", 2400, 256, 1024, False, True, 1),
    ("fwd words d->d", 1024, 256, 256, False, True, 1),
    ("fwd dec d->d  ", 320, 256, 256, False, True, 1),
    ("fwd tiny      ", 32, 256, 256, False, True, 1),
    ("fwd Dv->d     ", 2400, 256, 2818, False, True, 1),
    ("fwd MLM head  ", 1024, 5003, 256, False, True, 1),
    ("dX d<-d       ", 2400, 256, 256, False, False, 1),
    ("dX d<-F       ", 2400, 256, 1024, False, False, 1),
    ("dX F<-d       ", 2400, 1024, 256, False, False, 1),
    ("dW dxd s16    ", 256, 256, 2400, True, False, 16),
    ("dW Fxd s4     ", 1024, 256, 2400, True, False, 4),
    ("dW dxF s4     ", 256, 1024, 2400, True, False, 4),
    ("dW dxDv s1    ", 256, 2818, 2400, True, False, 1),
]
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us
for name, M, N, K, ta, tb, split in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.zeros(M, N, device=dev)
    us = timeit(lambda: kn.gemm(A, B, C, trans_a=ta, trans_b=tb, split_k=split))
    Am = A.t() if ta else A
    Bm = B.t() if tb else B
    us_t = timeit(lambda: torch.mm(Am, Bm, out=C))
    fl = 2.0 * M * N * K
    print("%s M=%5d N=%5d K=%5d  mesm %8.1f us %6.1f TF | torch.mm %8.1f us %6.1f TF" % (
        name, M, N, K, us, fl / us / 1e6, us_t, fl / us_t / 1e6), flush=True)
