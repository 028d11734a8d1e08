import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M, N, K = 256, 1024, 2400
A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
for split, acc in [(4, 2), (1, 1), (1, 0), (2, 2), (8, 2), (16, 2)]:
    C = torch.zeros(M, N, device=dev)
    us = timeit(lambda: kn.gemm(A, B, C, trans_a=True, split_k=split, accumulate=acc))
    print("dW dxF ldc=1024 split %2d acc %d: %8.1f us" % (split, acc, us), flush=True)
for pad in (16, 64):
    Cb = torch.zeros(M, N + pad, device=dev); C = Cb[:, :N]
    us = timeit(lambda: kn.gemm(A, B, C, trans_a=True, split_k=4, accumulate=2))
    print("dW dxF ldc=%d split 4 atomic: %8.1f us" % (N + pad, us), flush=True)
# the d x d case at several splits
M, N = 256, 256
A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
for split, acc in [(16, 2), (8, 2), (4, 2), (1, 1)]:
    C = torch.zeros(M, N, device=dev)
    us = timeit(lambda: kn.gemm(A, B, C, trans_a=True, split_k=split, accumulate=acc))
    print("dW dxd split %2d acc %d: %8.1f us" % (split, acc, us), flush=True)
# raw launch-rate floor of the python binding
x = torch.randn(64, 256, device=dev); g = torch.ones(256, device=dev); b = torch.zeros(256, device=dev)
print("layernorm_fwd tiny (host floor): %.1f us" % timeit(lambda: kn.layernorm_fwd(x, g, b)))
