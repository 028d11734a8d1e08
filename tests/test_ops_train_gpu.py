"""Train-mode (dropout on) blocks of mesm_amd.ops against plain-torch autograd compositions that
use the SAME counter-based masks, materialised with kn.dropout (mask = dropout(ones) / scale)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev():
    return torch.device("cuda:0")


def gen(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


def rel(a, b):
    a, b = a.double(), b.double()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)


def mask_of(shape, p, seed):
    from mesm_amd import kernels as kn
    return kn.dropout(torch.ones(*shape, device=dev()), p, seed).double()  # 0 or 1/(1-p)


@pytest.mark.parametrize("rows,D", [(2400, 256), (300, 2818)])
def test_layernorm_with_fused_dropout(rows, D):
    from mesm_amd import ops
    x = gen((rows, D), 1).requires_grad_()
    g = (gen((D,), 2) * 0.2 + 1.0).requires_grad_()
    b = (gen((D,), 3) * 0.1).requires_grad_()
    drop = (0.5, 4242)
    y = ops.layer_norm(x, g, b, drop=drop)
    m = mask_of((rows, D), *drop)
    xd, gd, bd = (t.detach().double().requires_grad_() for t in (x, g, b))
    ref = torch.nn.functional.layer_norm(xd, (D,), gd, bd, 1e-5) * m
    assert rel(y, ref) < TOL
    dy = gen((rows, D), 5)
    y.backward(dy)
    ref.backward(dy.double())
    assert rel(x.grad, xd.grad) < TOL and rel(g.grad, gd.grad) < TOL and rel(b.grad, bd.grad) < TOL
    # parameter-gradient-only path (input without gradient)
    x2 = x.detach()
    g2 = g.detach().clone().requires_grad_()
    b2 = b.detach().clone().requires_grad_()
    ops.layer_norm(x2, g2, b2, drop=drop).backward(dy)
    assert rel(g2.grad, gd.grad) < TOL and rel(b2.grad, bd.grad) < TOL


def test_ffn_block_with_dropout():
    from mesm_amd import ops
    rows, d, F = 1200, 256, 1024
    x = gen((rows, d), 10).requires_grad_()
    res = gen((rows, d), 11).requires_grad_()
    w1 = (gen((F, d), 12) * 0.05).requires_grad_()
    b1 = (gen((F,), 13) * 0.1).requires_grad_()
    w2 = (gen((d, F), 14) * 0.05).requires_grad_()
    b2 = (gen((d,), 15) * 0.1).requires_grad_()
    slope = torch.full((1,), 0.25, device=dev()).requires_grad_()
    mid, outd = (0.1, 77), (0.1, 78)
    y = ops.ffn(x, res, w1, b1, slope, w2, b2, mid_drop=mid, out_drop=outd)
    mm, mo = mask_of((rows, F), *mid), mask_of((rows, d), *outd)
    dd = [t.detach().double().requires_grad_() for t in (x, res, w1, b1, slope, w2, b2)]
    xd, rd, w1d, b1d, sd, w2d, b2d = dd
    z = xd @ w1d.t() + b1d
    h = torch.where(z > 0, z, sd * z) * mm
    ref = rd + (h @ w2d.t() + b2d) * mo
    assert rel(y, ref) < TOL
    dy = gen((rows, d), 16)
    y.backward(dy)
    ref.backward(dy.double())
    for got, want, name in zip((x, res, w1, b1, slope, w2, b2), dd, "x res w1 b1 slope w2 b2".split()):
        assert rel(got.grad, want.grad) < 2e-4, name
