"""The one-round 160 x 128 tile GEMM kernel (MESM_GEMM_TILE=7) against the default dispatch on the FFN shapes:
results (identical inputs, every epilogue term the step uses there) and time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g)).to(dev)

def run(force, fn, reps=50):
    if force: os.environ["MESM_GEMM_TILE"] = str(force)
    else: os.environ.pop("MESM_GEMM_TILE", None)
    out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    os.environ.pop("MESM_GEMM_TILE", None)
    return out, e0.elapsed_time(e1) / reps * 1e3

for M in (4800, 4864, 2400):
    for N, K in ((1024, 256), (1024, 512)):
        x = rnd(M, K); W = rnd(N, K) * 0.05; Wt = W.t().contiguous(); b = rnd(N); z = rnd(M, N); res = rnd(M, N)
        slope = torch.tensor([0.25], device=dev)
        cases = {
            "NT bias,pre_out,prelu,drop": lambda: (lambda C, P: (kn.gemm(x, W, C, trans_b=True, bias=b, e_act=kn.ACT_PRELU, slope=slope, e_drop=(0.1, 7), pre_out=P), P)[0:2])(torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)),
            "NN aux,prelu-grad,drop": lambda: (kn.gemm(x, Wt, torch.empty(M, N, device=dev), aux=z, e_actgrad=kn.ACT_PRELU, slope=slope, e_drop=(0.1, 9)),),
            "NT bias,residual": lambda: (kn.gemm(x, W, torch.empty(M, N, device=dev), trans_b=True, bias=b, residual=res),),
        }
        for name, fn in cases.items():
            o0, t0 = run(0, fn)
            o7, t7 = run(7, fn)
            o8, t8 = run(8, fn)
            err = max(float((a - c).abs().max()) / max(float(a.abs().max()), 1e-6) for a, c in zip(o0 + o0, o7 + o8))
            fl = 2.0 * M * N * K
            print("%5d x %4d x %4d %-28s default %6.2f us (%5.1f TF)  one round %6.2f us (%5.1f TF)  two halves %6.2f us (%5.1f TF)  max rel diff %.1e" %
                  (M, N, K, name, t0, fl / t0 / 1e6, t7, fl / t7 / 1e6, t8, fl / t8 / 1e6, err))
