import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import kernels as kn
from mesm_amd._lib import lib
dev = torch.device("cuda:0")
def gen(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)
for (M, N) in ((4800, 256), (2433, 258)):
  for K in (256, 262, 1024):
    for ta in (False, True):
        for tb in (False, True):
            A = gen((K, M) if ta else (M, K), 1); B = gen((N, K) if tb else (K, N), 2, 0.1)
            ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
            errs = []
            for mode in (0, 1):
                lib().mesm_gemm_set_pipe(mode)
                C = torch.zeros(M, N, device=dev)
                kn.gemm(A, B, C, trans_a=ta, trans_b=tb)
                torch.cuda.synchronize()
                d = (C.double() - ref).abs()
                d = torch.nan_to_num(d, nan=1e30, posinf=1e30)
                errs.append(float(d.max() / ref.abs().max()))
                if mode == 1 and errs[-1] > 1e-5:
                    bad = (d > 1e-3 * ref.abs().max())
                    rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
                    print("   bad frac %.4f rows %d..%d (%d) cols %d..%d (%d) nonfinite %d" % (float(bad.float().mean()), int(rows.min()), int(rows.max()), len(rows), int(cols.min()), int(cols.max()), len(cols), int((~torch.isfinite(C)).sum())))
            print("M=%d N=%d K=%5d ta%d tb%d  phased %.2e  pipelined %.2e" % (M, N, K, ta, tb, errs[0], errs[1]))
lib().mesm_gemm_set_pipe(0)
