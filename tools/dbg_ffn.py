import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn, ops
dev = torch.device("cuda:0")
def rel(a, b):
    return float((a.double() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-6)
g = torch.Generator().manual_seed(9)
for (M, K, Fd) in [(600, 256, 1024), (2400, 256, 1024), (320, 256, 1024), (600, 32, 64)]:
    x = torch.randn(M, K, generator=g).to(dev).requires_grad_(True)
    w1 = (torch.randn(Fd, K, generator=g) * 0.1).to(dev).requires_grad_(True)
    b1 = (torch.randn(Fd, generator=g) * 0.1).to(dev).requires_grad_(True)
    w2 = (torch.randn(K, Fd, generator=g) * 0.1).to(dev).requires_grad_(True)
    b2 = (torch.randn(K, generator=g) * 0.1).to(dev).requires_grad_(True)
    slope = torch.tensor([0.25], device=dev, requires_grad=True)
    y = ops.ffn(x, x, w1, b1, slope, w2, b2)
    gy = torch.randn(M, K, generator=g).to(dev)
    y.backward(gy)
    got = [t.grad.clone() for t in (x, w1, b1, w2, b2, slope)]
    for t in (x, w1, b1, w2, b2, slope): t.grad = None
    xd, w1d, b1d, w2d, b2d, sd = [t.detach().double().requires_grad_(True) for t in (x, w1, b1, w2, b2, slope)]
    z = xd @ w1d.t() + b1d
    ref = xd + (torch.where(z > 0, z, sd * z) @ w2d.t() + b2d)
    ref.backward(gy.double())
    print(M, K, Fd, "y", "%.1e" % rel(y, ref), " ".join("%s %.1e" % (n, rel(a, t.grad)) for n, a, t in zip("x w1 b1 w2 b2 slope".split(), got, (xd, w1d, b1d, w2d, b2d, sd))))
