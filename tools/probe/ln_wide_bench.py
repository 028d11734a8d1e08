"""The input LayerNorm of the feature projections (2400 x 2818 QVHighlights, 8192 x 4098 TACoS): forward with / without
the fused dropout, parameter-gradient backward; us per launch and effective TB/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
def run(body, n=8, reps=10):
    body(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): body()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / n * 1e3
for rows, D in ((2400, 2818), (8192, 4098), (2400, 2816), (8192, 4096)):
    xs = [torch.randn(rows, D, device=dev) for _ in range(4)]
    g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    for drop in ((0.0, 0), (0.5, 11)):
        def f():
            for i in range(8): kn.layernorm_fwd(xs[i % 4], g, b, drop=drop)
        t = run(f)
        print("%5d x %4d fwd drop %.1f: %7.2f us  %.2f TB/s" % (rows, D, drop[0], t, 2 * rows * D * 4 / t / 1e6))
