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


@pytest.mark.parametrize("rows,D", [(2400, 256), (77, 300), (300, 2818)])
def test_layernorm_backward_second_output_is_the_masked_dx(rows, D):
    """mesm_layernorm_bwd2: dx2 == dropout(dx; p, seed) element for element."""
    from mesm_amd import kernels as kn
    x = gen((rows, D), 11, 2.0)
    g = gen((D,), 12) * 0.2 + 1.0
    b = gen((D,), 13) * 0.1
    dy = gen((rows, D), 14)
    _, mean, rstd = kn.layernorm_fwd(x, g, b)
    dg1, db1 = torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
    dg2, db2 = torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
    dx_ref = kn.layernorm_bwd(dy, x, g, mean, rstd, dg1, db1)
    dx, dxm = kn.layernorm_bwd(dy, x, g, mean, rstd, dg2, db2, drop2=(0.1, 777))
    assert torch.equal(dx, dx_ref)
    assert torch.equal(dxm, kn.dropout(dx_ref, 0.1, 777))
    assert rel(dg2, dg1) < 1e-5 and rel(db2, db1) < 1e-5


@pytest.mark.parametrize("block", ["linear", "ffn", "mha"])
def test_post_norm_blocks_take_the_masked_gradient_from_the_layernorm(block, monkeypatch):
    """y = LayerNorm(res + dropout(block(x))): with the DropSink hand-over the block's backward launches no
    mask kernel of its own, and every gradient equals the hand-over-free path (split-K atomics make the
    weight gradients order-dependent in the last bits, hence a tolerance)."""
    from mesm_amd import kernels as kn, ops
    N, L, d = 8, 75, 256
    x0 = gen((N, L, d), 1)
    w = gen((d, d), 2, 0.05); bb = gen((d,), 3, 0.05)
    w1 = gen((4 * d, d), 4, 0.05); b1 = gen((4 * d,), 5, 0.05); w2 = gen((d, 4 * d), 6, 0.05); b2 = gen((d,), 7, 0.05)
    slope = torch.tensor([0.25], device=dev())
    w_in = gen((3 * d, d), 8, 0.05); b_in = gen((3 * d,), 9, 0.05)
    gam = gen((d,), 10) * 0.2 + 1.0; bet = gen((d,), 11) * 0.1
    up = gen((N, L, d), 12)

    def run(use_sink):
        calls = {"n": 0}
        orig = kn.dropout

        def counting(*a, **k):
            calls["n"] += 1
            return orig(*a, **k)
        monkeypatch.setattr(kn, "dropout", counting)
        if not use_sink:
            monkeypatch.setattr(ops, "_sink_for", lambda out_drop: None)
        leaves = [t.clone().requires_grad_() for t in (x0, w, bb, w1, b1, w2, b2, slope, w_in, b_in, gam, bet)]
        x, w_, bb_, w1_, b1_, w2_, b2_, sl_, wi_, bi_, g_, be_ = leaves
        od = (0.1, 4321)
        if block == "linear":
            y = ops.linear(x, w_, bb_, residual=x, out_drop=od)
        elif block == "ffn":
            y = ops.ffn(x, x, w1_, b1_, sl_, w2_, b2_, mid_drop=(0.1, 99), out_drop=od)
        else:
            y = ops.mha(x, None, x, None, x, wi_, bi_, w_, bb_, 8, attn_drop=(0.1, 5), out_drop=od, self_attn=True)
        z = ops.layer_norm(y, g_, be_)
        (z * up).sum().backward()
        torch.cuda.synchronize()
        monkeypatch.undo()
        return calls["n"], [t.grad.clone() if t.grad is not None else None for t in leaves]

    n_sink, g_sink = run(True)
    n_plain, g_plain = run(False)
    assert n_sink == 0 and n_plain == 1, (n_sink, n_plain)
    for a, b_ in zip(g_sink, g_plain):
        assert (a is None) == (b_ is None)
        if a is not None:
            assert rel(a, b_) < 1e-5
