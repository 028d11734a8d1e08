"""Per-kernel numerics on the MI355X: each HIP kernel against a plain PyTorch reference
(fp64 on the same device where a floating-point reference is needed).

Tolerance for fp32 kernels: 1e-4 relative to the output scale (the north-star budget).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    a = a.double()
    b = b.double()
    scale = max(b.abs().max().item(), 1e-6)
    return (a - b).abs().max().item() / scale


def gen(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


# --------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [(2400, 256, 256), (75, 33, 17), (320, 1024, 256), (300, 256, 2818),
               (130, 70, 50), (256, 256, 2400), (64, 64, 16), (1, 5, 3), (1024, 5003, 256)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_plain(M, N, K, ta, tb):
    from mesm_amd import kernels as kn
    A = gen((K, M) if ta else (M, K), 1)
    B = gen((N, K) if tb else (K, N), 2)
    C = torch.full((M, N), float("nan"), device=dev())
    kn.gemm(A, B, C, trans_a=ta, trans_b=tb)
    ref = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
    assert rel_err(C, ref) < TOL


def test_gemm_strided_views_and_bias_residual():
    from mesm_amd import kernels as kn
    M, N, K = 150, 96, 256
    Wfull = gen((3 * N, K), 3)
    W = Wfull[N:2 * N]  # a row slice of in_proj_weight
    X = gen((M, K), 4)
    X2 = gen((M, K), 5)
    bias = gen((N,), 6)
    res = gen((M, N), 7)
    Cbig = torch.zeros(M, 2 * N, device=dev())
    C = Cbig[:, N:]  # strided output
    kn.gemm(X, W, C, trans_b=True, A2=X2, bias=bias, residual=res, out_scale=0.5)
    ref = ((X + X2).double() @ W.t().double()) * 0.5 + bias.double() + res.double()
    assert rel_err(C, ref) < TOL
    assert Cbig[:, :N].abs().max().item() == 0.0


@pytest.mark.parametrize("split", [2, 4, 16])
def test_gemm_split_k_and_accumulate(split):
    from mesm_amd import kernels as kn
    M, N, K = 256, 300, 2400
    A = gen((K, M), 8)
    B = gen((K, N), 9)
    base = gen((M, N), 10)
    C = base.clone()
    bias = gen((N,), 11)
    kn.gemm(A, B, C, trans_a=True, split_k=split, bias=bias)
    ref = A.t().double() @ B.double() + base.double() + bias.double()
    assert rel_err(C, ref) < TOL
    C2 = base.clone()
    kn.gemm(A, B, C2, trans_a=True, accumulate=1)
    assert rel_err(C2, A.t().double() @ B.double() + base.double()) < TOL


def test_gemm_activations_and_grad_epilogue():
    from mesm_amd import kernels as kn
    M, N, K = 200, 160, 96
    X = gen((M, K), 12)
    W = gen((N, K), 13)
    b = gen((N,), 14)
    slope = torch.tensor([0.25], device=dev())
    z_ref = X.double() @ W.t().double() + b.double()
    C = torch.empty(M, N, device=dev())
    kn.gemm(X, W, C, trans_b=True, bias=b, e_act=kn.ACT_RELU)
    assert rel_err(C, z_ref.clamp(min=0)) < TOL
    kn.gemm(X, W, C, trans_b=True, bias=b, e_act=kn.ACT_PRELU, slope=slope)
    assert rel_err(C, torch.where(z_ref > 0, z_ref, 0.25 * z_ref)) < TOL
    # prologue PReLU on A:  prelu(Z) @ W2^T
    Z = gen((M, N), 15)
    W2 = gen((K, N), 16)
    C2 = torch.empty(M, K, device=dev())
    kn.gemm(Z, W2, C2, trans_b=True, a_act=kn.ACT_PRELU, slope=slope)
    H = torch.where(Z > 0, Z, 0.25 * Z).double()
    assert rel_err(C2, H @ W2.t().double()) < TOL
    # same on B (dW2 = dY^T @ prelu(Z))
    dY = gen((M, K), 17)
    dW = torch.empty(K, N, device=dev())
    colsum = torch.zeros(K, device=dev())
    kn.gemm(dY, Z, dW, trans_a=True, b_act=kn.ACT_PRELU, slope=slope, colsum=colsum)
    assert rel_err(dW, dY.t().double() @ H) < TOL
    assert rel_err(colsum, dY.double().sum(0)) < TOL
    # backward epilogue: dZ = (dY @ W2) * prelu'(Z), dslope = sum(dH * min(Z,0))
    dZ = torch.empty(M, N, device=dev())
    dslope = torch.zeros(1, device=dev())
    kn.gemm(dY, W2, dZ, aux=Z, e_actgrad=kn.ACT_PRELU, slope=slope, dslope=dslope)
    dH = dY.double() @ W2.double()
    assert rel_err(dZ, torch.where(Z > 0, dH, 0.25 * dH)) < TOL
    ds_ref = (dH * Z.double().clamp(max=0)).sum()
    assert abs(dslope.item() - ds_ref.item()) / max(abs(ds_ref.item()), 1.0) < TOL
    # relu grad epilogue uses the activation output
    Y = Z.clamp(min=0)
    kn.gemm(dY, W2, dZ, aux=Y, e_actgrad=kn.ACT_RELU)
    assert rel_err(dZ, torch.where(Y > 0, dH, torch.zeros_like(dH))) < TOL


def test_gemm_dropout_matches_materialised_mask():
    from mesm_amd import kernels as kn
    M, N, K = 190, 130, 300
    X = gen((M, K), 18)
    W = gen((N, K), 19)
    Xd = kn.dropout(X, 0.5, 77)
    keep = (Xd != 0).float().mean().item()
    assert 0.45 < keep < 0.55
    assert torch.allclose(Xd[Xd != 0], (X * 2.0)[Xd != 0])
    C = torch.empty(M, N, device=dev())
    kn.gemm(X, W, C, trans_b=True, a_drop=(0.5, 77))
    assert rel_err(C, Xd.double() @ W.t().double()) < TOL
    # operand B (dW = dY^T @ dropout(X)): same mask, logical index m*K + k
    dY = gen((M, N), 20)
    dW = torch.empty(N, K, device=dev())
    kn.gemm(dY, X, dW, trans_a=True, b_drop=(0.5, 77))
    assert rel_err(dW, dY.t().double() @ Xd.double()) < TOL
    # epilogue dropout (dX = dropout_mask * (dY @ W))
    dX = torch.empty(M, K, device=dev())
    kn.gemm(dY, W, dX, e_drop=(0.5, 77))
    full = (dY.double() @ W.double())
    mask = (Xd != 0).double() * 2.0
    assert rel_err(dX, full * mask) < TOL


@pytest.mark.parametrize("tile", ["0", "1", "2", "3", "4", "6", "32", "64"])
@pytest.mark.parametrize("M,N,K", [(2400, 256, 256), (190, 132, 300), (130, 70, 262), (66, 129, 35)])
def test_gemm_operand_dropout_uses_the_stored_index(tile, M, N, K, monkeypatch):
    """The mask an epilogue wrote on Y (index row*N + col) is replayed when dY is an operand:
    as A of dX = drop(dY) @ W and as A^T of dW = drop(dY)^T @ X (+ the bias column sums)."""
    from mesm_amd import kernels as kn
    monkeypatch.setenv("MESM_GEMM_TILE", tile)
    dY = gen((M, N), 40)
    W = gen((N, K), 41)
    X = gen((M, K), 42)
    dYd = kn.dropout(dY, 0.1, 1234)
    dX = torch.empty(M, K, device=dev())
    kn.gemm(dY, W, dX, a_drop=(0.1, 1234))
    assert rel_err(dX, dYd.double() @ W.double()) < TOL
    dW = torch.zeros(N, K, device=dev())
    db = torch.zeros(N, device=dev())
    kn.gemm(dY, X, dW, trans_a=True, a_drop=(0.1, 1234), colsum=db, split_k=4, accumulate=2)
    assert rel_err(dW, dYd.t().double() @ X.double()) < TOL
    assert rel_err(db, dYd.double().sum(0)) < TOL
    # and operand B stored (N, K) (trans_b): index n*K + k
    Wd = kn.dropout(W, 0.1, 99)
    C = torch.empty(M, N, device=dev())
    kn.gemm(X, W, C, trans_b=True, b_drop=(0.1, 99))
    assert rel_err(C, X.double() @ Wd.t().double()) < TOL
    # prelu + dropout on A together
    slope = torch.tensor([0.1], device=dev())
    Z = gen((M, K), 21)
    Hd = kn.dropout(torch.where(Z > 0, Z, 0.1 * Z).contiguous(), 0.1, 5)
    kn.gemm(Z, W, C, trans_b=True, a_act=kn.ACT_PRELU, slope=slope, a_drop=(0.1, 5))
    assert rel_err(C, Hd.double() @ W.t().double()) < TOL


# --------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,D", [(2400, 256), (7, 32), (1056, 512), (64, 300), (300, 2818),
                                    (33, 4098), (5, 1024)])
def test_layernorm(rows, D):
    from mesm_amd import kernels as kn
    x = gen((rows, D), 30, 2.0) + 0.5
    g = gen((D,), 31) * 0.2 + 1.0
    b = gen((D,), 32) * 0.1
    y, mean, rstd = kn.layernorm_fwd(x, g, b)
    xd = x.double().requires_grad_(True)
    gd = g.double().requires_grad_(True)
    bd = b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (D,), gd, bd, 1e-5)
    assert rel_err(y, ref) < TOL
    dy = gen((rows, D), 33)
    ref.backward(dy.double())
    dg = torch.zeros(D, device=dev())
    db = torch.zeros(D, device=dev())
    dx = kn.layernorm_bwd(dy, x, g, mean, rstd, dg, db)
    assert rel_err(dx, xd.grad) < TOL
    assert rel_err(dg, gd.grad) < TOL
    assert rel_err(db, bd.grad) < TOL
    # accumulate into an existing dx
    base = gen((rows, D), 34)
    dx2 = base.clone()
    kn.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dx=dx2, accumulate_dx=True)
    assert rel_err(dx2, xd.grad + base.double()) < TOL
    # parameter gradients only (input without gradient)
    dg2 = torch.zeros(D, device=dev())
    db2 = torch.zeros(D, device=dev())
    assert kn.layernorm_bwd(dy, x, g, mean, rstd, dg2, db2, need_dx=False) is None
    assert rel_err(dg2, gd.grad) < TOL
    assert rel_err(db2, bd.grad) < TOL


@pytest.mark.parametrize("M,K,J", [(320, 256, 2), (640, 256, 2), (320, 256, 1), (37, 300, 3), (1, 7, 4), (2400, 512, 1)])
@pytest.mark.parametrize("relu_mask", [False, True])
def test_skinny_linear_backward_in_one_launch(M, K, J, relu_mask):
    """dX, dW, db of a 1-4 feature Linear (the heads' last layers) from mesm_skinny_linear_bwd against fp64;
    dW / db accumulate into what the views already hold."""
    from mesm_amd import kernels as kn
    x = gen((M, K), 60)
    if relu_mask:
        x = torch.relu(x)
    w = gen((J, K), 61, 0.2)
    dz = gen((M, J), 62)
    dw0, db0 = gen((J, K), 63), gen((J,), 64)
    dw, db = dw0.clone(), db0.clone()
    dx = kn.skinny_linear_bwd(dz, x, w, dw, db, relu_mask=relu_mask)
    ref_dx = dz.double() @ w.double()
    if relu_mask:
        ref_dx = ref_dx * (x > 0)
    assert rel_err(dx, ref_dx) < TOL
    assert rel_err(dw, dw0.double() + dz.double().t() @ x.double()) < TOL
    assert rel_err(db, db0.double() + dz.double().sum(0)) < TOL
    # parameter gradients only
    dw2 = torch.zeros_like(dw0)
    assert kn.skinny_linear_bwd(dz, x, w, dw2, None, need_dx=False) is None
    assert rel_err(dw2, dz.double().t() @ x.double()) < TOL


def test_linear_block_backward_is_the_same_through_either_route():
    """ops.linear with 2 output features behind a ReLU layer: the one-launch backward against the two-GEMM route."""
    from mesm_amd import ops
    outs = {}
    for skinny in (True, False):
        ops.SKINNY_BWD = skinny
        try:
            x = gen((32, 10, 256), 70).requires_grad_(True)
            w1, b1 = gen((256, 256), 71, 0.06).requires_grad_(True), gen((256,), 72, 0.1).requires_grad_(True)
            w2, b2 = gen((2, 256), 73, 0.06).requires_grad_(True), gen((2,), 74, 0.1).requires_grad_(True)
            h = ops.linear(x, w1, b1, relu=True)
            y = ops.linear(h, w2, b2)
            y.backward(gen((32, 10, 2), 75))
            outs[skinny] = [t.grad.clone() for t in (x, w1, b1, w2, b2)]
        finally:
            ops.SKINNY_BWD = True
    for a, b in zip(outs[True], outs[False]):
        assert rel_err(a, b) < 1e-5


def test_reference_point_init_matches_sigmoid_expand_and_its_autograd():
    from mesm_amd import ops
    p = gen((10, 2), 80).requires_grad_(True)
    ref = ops.ref_init(p, 32)
    pd = p.detach().double().requires_grad_(True)
    want = torch.sigmoid(pd)[None].expand(32, 10, 2)
    assert ref.shape == (32, 10, 2) and rel_err(ref, want) < 1e-6
    g = gen((32, 10, 2), 81)
    ref.backward(g)
    want.backward(g.double())
    assert rel_err(p.grad, pd.grad) < 1e-5


@pytest.mark.parametrize("n,nq,D", [(32, 10, 256), (3, 7, 32), (512, 10, 64)])
def test_fused_reference_point_init_equals_ref_init_then_query_sine(n, nq, D):
    """ops.ref_init_sine (one launch each way, three aliases of ref) against ref_init | query_sine and the autograd
    engine's fan-in adds: bit-identical forward, parameter gradient to rounding."""
    from mesm_amd import ops
    p = gen((nq, 2), 80).requires_grad_(True)
    buf = torch.empty((2, n, nq, 2), device=dev())
    ra, rb, rc, qs, qsb = ops.ref_init_sine(p, n, D, ops.Slot(buf, 0))
    p2 = p.detach().clone().requires_grad_(True)
    ref = ops.ref_init(p2, n)
    qs2 = ops.query_sine(ref, D)
    assert torch.equal(ra, ref) and torch.equal(qs, qs2)
    assert ra.data_ptr() == buf[0].data_ptr() and torch.equal(buf[0], ref)
    ga, gb, gc, gq = gen((n, nq, 2), 81), gen((n, nq, 2), 82), gen((n, nq, 2), 83), gen((n, nq, D), 84)
    gq2 = gen((n, nq, D), 85)
    torch.autograd.backward([ra, rb, rc, qs, qsb], [ga, gb, gc, gq, gq2])
    torch.autograd.backward([ref, qs2], [ga + gb + gc, gq + gq2])
    assert rel_err(p.grad, p2.grad) < 1e-5
    # only some consumers deliver a gradient
    p3 = p.detach().clone().requires_grad_(True)
    ra, rb, rc, qs, qsb = ops.ref_init_sine(p3, n, D)
    torch.autograd.backward([rb], [gb])
    p4 = p.detach().clone().requires_grad_(True)
    ops.ref_init(p4, n).backward(gb)
    assert rel_err(p3.grad, p4.grad) < 1e-5


@pytest.mark.parametrize("n,nq,D,prev_grad", [(32, 10, 256, True), (32, 10, 256, False), (5, 3, 32, True)])
def test_fused_reference_point_step_equals_ref_update_query_sine_qsine_scale(n, nq, D, prev_grad):
    """ops.ref_step (the layer boundary of the decoder as one launch each way) against ref_update | query_sine(detached) |
    qsine_scale(detached): bit-identical forward and gradients."""
    from mesm_amd import ops
    ins = [gen((n, nq, 2), 90, 0.5), torch.sigmoid(gen((n, nq, 2), 91)), gen((n, nq, D), 92), gen((n, nq, 1), 93)]
    outs = []
    for fused in (True, False):
        delta, prev, scale, anchor = [t.clone().requires_grad_(i != 1 or prev_grad) for i, t in enumerate(ins)]
        if fused:
            buf = torch.empty((3, n, nq, 2), device=dev())
            new_ref, qs, qsc = ops.ref_step(delta, prev, scale, anchor, D, ops.Slot(buf, 2))
            assert new_ref.data_ptr() == buf[2].data_ptr() and not qs.requires_grad
        else:
            new_ref = ops.ref_update(delta, prev)
            qs = ops.query_sine(new_ref.detach(), D)
            qsc = ops.qsine_scale(qs, scale, anchor, new_ref.detach())
        torch.autograd.backward([new_ref, qsc], [gen((n, nq, 2), 94), gen((n, nq, D), 95)])
        outs.append([new_ref, qs, qsc, delta.grad, prev.grad, scale.grad, anchor.grad])
    for a, b in zip(*outs):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    # one of the two output gradients absent
    delta, prev, scale, anchor = [t.clone().requires_grad_(True) for t in ins]
    new_ref, qs, qsc = ops.ref_step(delta, prev, scale, anchor, D)
    new_ref.backward(gen((n, nq, 2), 94))
    assert torch.equal(delta.grad, outs[1][3]) and scale.grad is None and anchor.grad is None
    delta, prev, scale, anchor = [t.clone().requires_grad_(True) for t in ins]
    new_ref, qs, qsc = ops.ref_step(delta, prev, scale, anchor, D)
    qsc.backward(gen((n, nq, D), 95))
    assert torch.equal(scale.grad, outs[1][5]) and torch.equal(anchor.grad, outs[1][6]) and delta.grad is None


def test_stacked_outputs_written_in_place_equal_torch_stack():
    """ops.stacked over slots that LayerNorm kernels wrote (the decoder's per-layer outputs): the values and gradients of
    torch.stack without its copy."""
    from mesm_amd import ops
    xs = [gen((32, 10, 256), 100 + i) for i in range(2)]
    gamma, beta = gen((256,), 110).requires_grad_(True), gen((256,), 111).requires_grad_(True)
    g = gen((2, 32, 10, 256), 112)
    res = []
    for in_place in (True, False):
        xi = [x.clone().requires_grad_(True) for x in xs]
        gm, bt = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
        if in_place:
            buf = torch.empty((2, 32, 10, 256), device=dev())
            ys = [ops.run(ops.layer_norm_call(x, gm, bt, slot=ops.Slot(buf, k))) for k, x in enumerate(xi)]
            st = ops.stacked(buf, ys)
            assert st.data_ptr() == buf.data_ptr()
        else:
            st = torch.stack([ops.layer_norm(x, gm, bt) for x in xi])
        st.backward(g)
        res.append([st.detach()] + [x.grad for x in xi] + [gm.grad, bt.grad])
    for a, b in zip(*res):
        assert rel_err(a, b) < 1e-6
    with pytest.raises(AssertionError):
        ops.stacked(torch.empty((2, 32, 10, 256), device=dev()), [x for x in xs])


# --------------------------------------------------------------------------- attention
def attn_reference(q, k, v, H, kpad, qpad, scale, quirk_B=None):
    """fp64 restatement of nn.MultiheadAttention's core incl. the reference's
    .repeat(nhead,1,1) attn_mask (transformer.py:528-530)."""
    B, Lq, E = q.shape
    Lk = k.shape[1]
    dk = E // H
    dv = v.shape[2] // H
    qh = q.double().view(B, Lq, H, dk).permute(0, 2, 1, 3).reshape(B * H, Lq, dk)
    kh = k.double().view(B, Lk, H, dk).permute(0, 2, 1, 3).reshape(B * H, Lk, dk)
    vh = v.double().view(B, Lk, H, dv).permute(0, 2, 1, 3).reshape(B * H, Lk, dv)
    s = torch.bmm(qh * scale, kh.transpose(1, 2))
    if qpad is not None:
        am = torch.matmul(qpad.float().unsqueeze(2), kpad.float().unsqueeze(1)).bool().repeat(H, 1, 1)
        s = s.masked_fill(am, float("-inf"))
    if kpad is not None:
        s = s.view(B, H, Lq, Lk).masked_fill(kpad.view(B, 1, 1, Lk), float("-inf")).view(B * H, Lq, Lk)
    p = torch.softmax(s, dim=-1)
    o = torch.bmm(p, vh).view(B, H, Lq, dv).permute(0, 2, 1, 3).reshape(B, Lq, H * dv)
    return o, p


def make_pad(B, L, seed, min_valid=1):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(min_valid, L + 1, (B,), generator=g)
    lens[0] = L
    return (torch.arange(L)[None, :] >= lens[:, None]).to(dev())


ATTN_CASES = [
    # B, H, Lq, Lk, dk, dv, kpad, quirk
    (32, 8, 75, 32, 32, 32, True, True),
    (32, 8, 75, 33, 32, 32, True, True),
    (4, 8, 76, 76, 32, 32, True, False),
    (32, 8, 10, 75, 64, 32, True, False),
    (32, 8, 10, 10, 32, 32, False, False),
    (6, 4, 20, 12, 8, 8, True, True),
    (3, 4, 7, 300, 8, 8, True, True),
    (2, 8, 130, 513, 32, 32, True, False),
    (5, 4, 9, 20, 16, 8, True, False),
    # matrix-core kernels (dk = dv = 32, Lk <= 128): MLM shape, one query row, ragged last blocks, the Lk limit
    (32, 8, 33, 75, 32, 32, True, True),
    (6, 8, 1, 75, 32, 32, True, False),
    (3, 4, 100, 128, 32, 32, True, True),
    (3, 4, 65, 129, 32, 32, True, True),
    (2, 2, 40, 2, 32, 32, False, False),
    (4, 8, 75, 36, 32, 32, True, True),
    (4, 8, 20, 67, 32, 32, True, False),
    # long query ranges: matrix-core backward (Lq >= 256, Lk <= 128), two-pass forward (Lk > 128, Lq >= 128)
    (4, 8, 512, 17, 32, 32, True, True),
    (2, 8, 300, 128, 32, 32, True, False),
    (2, 4, 257, 40, 32, 32, True, True),
    # 16 x 16-block kernels (attention_blk.hip): 64-wide heads forward (Lq > 16) and backward, the LDS limit of the
    # staged head (96 + 112 rows), just beyond it (lane-per-key fallback)
    (4, 4, 40, 75, 64, 32, True, False),
    (3, 4, 50, 60, 64, 32, True, True),
    (2, 4, 96, 112, 32, 32, True, True),
    (2, 4, 112, 112, 32, 32, True, False),
    (2, 4, 140, 200, 32, 32, True, True),
    (2, 8, 513, 513, 32, 32, True, True),   # TACoS encoder: J / I workgroups streaming the other side; the 1-row tail block of
                                            # each side rides with the side's last full workgroup
    (1, 4, 512, 512, 32, 32, True, True),   # ... no tail
    (2, 4, 272, 397, 32, 32, True, True),   # ... a full 16-row tail block (272 = 2 x 128 + 16) beside a side without one
    (1, 2, 140, 258, 32, 32, False, False),  # ... tails of 12 and 2 rows, no masks
    (1, 2, 300, 130, 32, 32, False, False),
]


@pytest.mark.parametrize("B,H,Lq,Lk,dk,dv,use_kpad,quirk", ATTN_CASES)
def test_attention_fwd_bwd(B, H, Lq, Lk, dk, dv, use_kpad, quirk):
    from mesm_amd import kernels as kn
    q = gen((B, Lq, H * dk), 40)
    k = gen((B, Lk, H * dk), 41)
    v = gen((B, Lk, H * dv), 42)
    kpad = make_pad(B, Lk, 43) if use_kpad else None
    qpad = make_pad(B, Lq, 44) if quirk else None
    scale = dk ** -0.5
    o, lse = kn.attn_fwd(q, k, v, H, kpad=kpad, qpad=qpad, scale=scale)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, _ = attn_reference(qd, kd, vd, H, kpad, qpad, scale)
    assert rel_err(o, ref) < TOL
    do = gen((B, Lq, H * dv), 45)
    ref.backward(do.double())
    dq, dk_, dv_ = kn.attn_bwd(do, q, k, v, o, lse, H, kpad=kpad, qpad=qpad, scale=scale)
    assert rel_err(dq, qd.grad) < TOL
    assert rel_err(dk_, kd.grad) < TOL
    assert rel_err(dv_, vd.grad) < TOL


def test_two_long_attention_backwards_in_one_phase():
    """with Lk > 128 and a head too large to stage, mesm_attn_bwd alone takes the dq-WRITING long block kernel
    (attn_bwd_adds_dq == False, so callers hand in uninitialised dq); two such problems in ONE launch phase used to fall
    into the lane-per-key GROUP kernel, which ADDS into dq.  dq starts as NaN here: any add shows."""
    from mesm_amd import kernels as kn
    B, H, Lq, Lk, dk = 2, 4, 150, 256, 32
    probs = []
    for i in range(2):
        q, k, v = gen((B, Lq, H * dk), 70 + i), gen((B, Lk, H * dk), 72 + i), gen((B, Lk, H * dk), 74 + i)
        o, lse = kn.attn_fwd(q, k, v, H)
        do = gen((B, Lq, H * dk), 76 + i)
        adds = kn.attn_bwd_adds_dq(B, H, Lq, Lk, dk, dk)
        dq = torch.zeros_like(q) if adds else torch.full_like(q, float("nan"))
        probs.append((q, k, v, o, lse, do, dq, torch.empty_like(k), torch.empty_like(v)))
    with kn.phase():
        for q, k, v, o, lse, do, dq, dk_, dv_ in probs:
            kn.attn_bwd_into(do, q, k, v, o, lse, H, dq, dk_, dv_)
    for q, k, v, o, lse, do, dq, dk_, dv_ in probs:
        qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
        ref, _ = attn_reference(qd, kd, vd, H, None, None, dk ** -0.5)
        ref.backward(do.double())
        assert rel_err(dq, qd.grad) < TOL
        assert rel_err(dk_, kd.grad) < TOL
        assert rel_err(dv_, vd.grad) < TOL


def test_attention_dropout_is_consistent_between_fwd_and_bwd():
    from mesm_amd import kernels as kn
    B, H, Lq, Lk, dk = 4, 8, 30, 90, 32
    q = gen((B, Lq, H * dk), 50)
    k = gen((B, Lk, H * dk), 51)
    v = gen((B, Lk, H * dk), 52)
    p, seed = 0.1, 1234
    o, lse = kn.attn_fwd(q, k, v, H, drop=(p, seed))
    # materialise the mask with the same counter hash: element index ((b*H+h)*Lq+i)*Lk+j
    ones = torch.ones(B * H * Lq * Lk, device=dev())
    mask = kn.dropout(ones, p, seed).view(B * H, Lq, Lk).double()
    assert 0.85 < (mask != 0).double().mean().item() < 0.95
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    _, probs = attn_reference(qd, kd, vd, H, None, None, dk ** -0.5)
    vh = vd.view(B, Lk, H, dk).permute(0, 2, 1, 3).reshape(B * H, Lk, dk)
    ref = torch.bmm(probs * mask, vh).view(B, H, Lq, dk).permute(0, 2, 1, 3).reshape(B, Lq, H * dk)
    assert rel_err(o, ref) < TOL
    do = gen((B, Lq, H * dk), 53)
    ref.backward(do.double())
    dq, dk_, dv_ = kn.attn_bwd(do, q, k, v, o, lse, H, drop=(p, seed))
    assert rel_err(dq, qd.grad) < TOL
    assert rel_err(dk_, kd.grad) < TOL
    assert rel_err(dv_, vd.grad) < TOL


# --------------------------------------------------------------------------- position encodings
def sine_pos_reference(mask, D):
    # restatement of position_encoding.py:61-70 on the CPU in fp32
    x = mask.cpu().cumsum(1, dtype=torch.float32)
    x = x / (x[:, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(D, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / D)
    pos = x[:, :, None] / dim_t
    return torch.stack((pos[:, :, 0::2].sin(), pos[:, :, 1::2].cos()), dim=3).flatten(2)


@pytest.mark.parametrize("B,L,D", [(32, 75, 256), (3, 600, 256), (4, 13, 32)])
def test_sine_pos(B, L, D):
    from mesm_amd import kernels as kn
    mask = ~make_pad(B, L, 60)
    out = kn.sine_pos(mask, D)
    ref = sine_pos_reference(mask, D)
    assert (out.cpu() - ref).abs().max().item() < 2e-5


def query_sine_reference(ref_pts, D):
    scale = 2 * math.pi
    half = D // 2
    dim_t = torch.arange(half, dtype=ref_pts.dtype, device=ref_pts.device)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="trunc") / half)
    outs = []
    for c in range(2):
        e = ref_pts[..., c] * scale
        pos = e[..., None] / dim_t
        outs.append(torch.stack((pos[..., 0::2].sin(), pos[..., 1::2].cos()), dim=-1).flatten(-2))
    return torch.cat(outs, dim=-1)


@pytest.mark.parametrize("R,D", [(320, 256), (20, 32)])
def test_query_sine(R, D):
    from mesm_amd import kernels as kn
    ref_pts = torch.sigmoid(gen((R, 2), 61))
    out = kn.query_sine_fwd(ref_pts, D)
    rd = ref_pts.double().requires_grad_(True)
    ref = query_sine_reference(rd, D)
    assert (out.double() - ref).abs().max().item() < 2e-5
    dout = gen((R, D), 62)
    ref.backward(dout.double())
    dref = kn.query_sine_bwd(ref_pts, dout)
    assert rel_err(dref, rd.grad) < TOL


# --------------------------------------------------------------------------- act / bias backward
@pytest.mark.parametrize("rows,cols", [(2400, 1024), (37, 50), (320, 256)])
def test_act_bias_bwd(rows, cols):
    from mesm_amd import kernels as kn
    dy = gen((rows, cols), 70)
    z = gen((rows, cols), 71)
    slope = torch.tensor([0.3], device=dev())
    db = torch.zeros(cols, device=dev())
    ds = torch.zeros(1, device=dev())
    dz = kn.act_bias_bwd(dy, z, kn.ACT_PRELU, dbias=db, slope=slope, dslope=ds)
    ref = torch.where(z > 0, dy, 0.3 * dy).double()
    assert rel_err(dz, ref) < TOL
    assert rel_err(db, ref.sum(0)) < TOL
    ds_ref = (dy.double() * z.double().clamp(max=0)).sum().item()
    assert abs(ds.item() - ds_ref) / max(abs(ds_ref), 1.0) < TOL
    y = z.clamp(min=0)
    db.zero_()
    dz = kn.act_bias_bwd(dy, y, kn.ACT_RELU, dbias=db)
    ref = torch.where(y > 0, dy, torch.zeros_like(dy)).double()
    assert rel_err(dz, ref) < TOL
    assert rel_err(db, ref.sum(0)) < TOL
    db.zero_()
    kn.act_bias_bwd(dy, None, kn.ACT_NONE, dbias=db)
    assert rel_err(db, dy.double().sum(0)) < TOL


# --------------------------------------------------------------------------- losses
@pytest.mark.parametrize("R,C", [(1024, 5003), (64, 1112), (10, 53)])
def test_nll_smooth(R, C):
    from mesm_amd import kernels as kn
    logit = gen((R, C), 80, 3.0)
    g = torch.Generator().manual_seed(81)
    label = torch.randint(0, C, (R,), generator=g).to(dev())
    mask = (torch.rand(R, generator=g) > 0.3).to(dev())
    row_loss, row_lse, correct = kn.nll_smooth_fwd(logit, label, mask)
    ld = logit.double().requires_grad_(True)
    lp = ld.log_softmax(-1)
    nll = -lp.gather(-1, label[:, None]).squeeze(-1)
    smooth = -lp.sum(-1)
    ref = ((1 - 0.1) * nll + 0.1 / C * smooth).masked_fill(~mask, 0)
    assert rel_err(row_loss, ref) < TOL
    assert torch.equal(correct.bool(), logit.argmax(-1) == label)
    w = gen((R,), 82).abs() * mask
    (ref * w.double()).sum().backward()
    dl = kn.nll_smooth_bwd(logit, label, row_lse, w)
    assert rel_err(dl, ld.grad) < TOL


def saliency_reference(s_pos, s_neg, label, vmask, pos_idx, neg_idx, rank_coef, margin):
    """fp64 restatement of criterion.py:139-221."""
    vm = vmask.double()
    N = s_pos.shape[0]
    loss_neg = (-torch.log(1.0 - torch.sigmoid(s_neg)) * vm).sum(1).mean()
    sc = torch.cat([s_pos, s_neg], 1)
    lab = torch.cat([label, torch.zeros_like(label)], 1)
    vm2 = vm.repeat(1, 2)
    sc = vm2 * sc + (1.0 - vm2) * -1e3
    total = 0.0
    for r in range(1, 12):
        pos = lab >= r
        if pos.sum() == 0:
            continue
        bd = pos.sum(1) > 0
        cur = sc / 0.5
        lg = cur - cur.max(1, keepdim=True)[0]
        lp = lg - torch.log(torch.exp(lg).sum(1, keepdim=True) + 1e-6)
        mlp = (pos * lp * vm2).sum(1) / (pos.sum(1) + 1e-6)
        total = total + (-mlp * bd).mean()
    total = total / rank_coef + loss_neg
    if pos_idx is not None:
        P = pos_idx.shape[1]
        bi = torch.arange(N, device=s_pos.device)
        ps = torch.stack([s_pos[bi, pos_idx[:, c]] for c in range(P)], 1)
        ns = torch.stack([s_pos[bi, neg_idx[:, c]] for c in range(P)], 1)
        total = total + torch.clamp(margin + ns - ps, min=0).sum() / (N * P) * 2
    return total


@pytest.mark.parametrize("N,L,qvh", [(32, 75, True), (5, 40, False), (3, 600, False)])
def test_saliency_loss(N, L, qvh):
    from mesm_amd import kernels as kn
    s_pos = gen((N, L), 90)
    s_neg = gen((N, L), 91)
    vmask = ~make_pad(N, L, 92, min_valid=12)
    g = torch.Generator().manual_seed(93)
    if qvh:
        label = torch.randint(0, 13, (N, L), generator=g).double().to(dev())
        label = label * (torch.rand(N, L, generator=g) > 0.6).to(dev())
    else:
        label = (torch.rand(N, L, generator=g) > 0.7).double().to(dev())
    label = label * vmask
    label[0] = 0  # a row without positives
    pos_idx = torch.randint(0, 12, (N, 2), generator=g).to(dev())
    neg_idx = torch.randint(0, 12, (N, 2), generator=g).to(dev())
    out = kn.saliency_loss_fwd(s_pos, s_neg, label, vmask, pos_idx, neg_idx, 12.0, 0.2)
    sp = s_pos.double().requires_grad_(True)
    sn = s_neg.double().requires_grad_(True)
    ref = saliency_reference(sp, sn, label, vmask, pos_idx, neg_idx, 12.0, 0.2)
    assert abs(out.item() - ref.item()) / max(abs(ref.item()), 1.0) < TOL
    ref.backward()
    gs = torch.tensor([0.7], device=dev())
    ds_pos, ds_neg = kn.saliency_loss_bwd(s_pos, s_neg, label, vmask, pos_idx, neg_idx, 12.0, 0.2, gs)
    assert rel_err(ds_pos, 0.7 * sp.grad) < TOL
    assert rel_err(ds_neg, 0.7 * sn.grad) < TOL
    # without the triplet term
    out2 = kn.saliency_loss_fwd(s_pos, s_neg, label, vmask, None, None, 1.0, 0.2)
    ref2 = saliency_reference(s_pos.double(), s_neg.double(), label, vmask, None, None, 1.0, 0.2)
    assert abs(out2.item() - ref2.item()) / max(abs(ref2.item()), 1.0) < TOL


@pytest.mark.parametrize("N,Q,tmax", [(32, 10, 5), (9, 6, 9), (7, 7, 7), (12, 40, 20), (5, 64, 64), (4, 3, 64), (33, 33, 2)])
def test_match_against_scipy(N, Q, tmax):
    """any (Q x T) block like the reference's scipy call (matcher.py:108-117): T < Q, T == Q, T > Q (targets left
    unmatched: -1), up to 64 x 64"""
    from scipy.optimize import linear_sum_assignment
    from mesm_amd import kernels as kn
    g = torch.Generator().manual_seed(95 + Q)
    logits = gen((N, Q, 2), 96)
    spans = torch.sigmoid(gen((N, Q, 2), 97))
    sizes = [1 + (i % tmax) for i in range(N)]
    sizes[-1] = tmax
    off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32)
    T = int(off[-1])
    st = torch.rand(T, generator=g) * 0.6
    ed = st + 0.05 + torch.rand(T, generator=g) * 0.3
    xx = torch.stack([st, ed], 1).to(dev())
    cxw = torch.stack([(st + ed) * 0.5, ed - st], 1).to(dev())
    mq, cost = kn.match(logits, spans, cxw, xx, off.to(dev()), tmax, 10.0, 1.0, 4.0, want_cost=True)
    # cost against the reference formula (matcher.py:70-105) in fp32 torch
    prob = logits.flatten(0, 1).softmax(-1)
    osp = spans.flatten(0, 1)
    c_span = torch.cdist(osp, cxw, p=1)
    x1 = osp[:, 0] - 0.5 * osp[:, 1]
    x2 = osp[:, 0] + 0.5 * osp[:, 1]
    inter = (torch.min(x2[:, None], xx[:, 1]) - torch.max(x1[:, None], xx[:, 0])).clamp(min=0)
    union = (x2 - x1)[:, None] + (xx[:, 1] - xx[:, 0]) - inter
    enc = (torch.max(x2[:, None], xx[:, 1]) - torch.min(x1[:, None], xx[:, 0])).clamp(min=0)
    giou = inter / union - (enc - union) / enc
    Cref = (10.0 * c_span + 1.0 * (-giou) + 4.0 * (-prob[:, :1])).view(N, Q, T).cpu()
    mq = mq.cpu()
    for b in range(N):
        cb = Cref[b, :, off[b]:off[b + 1]]
        assert (cost[b, :, :sizes[b]].cpu() - cb).abs().max().item() < 1e-5
        # the kernel's own fp32 costs decide the assignment (bit-exact against scipy on the same numbers)
        qi, ti = linear_sum_assignment(cost[b, :, :sizes[b]].cpu().double().numpy())
        want = torch.full((sizes[b],), -1, dtype=torch.int32)
        want[torch.as_tensor(ti)] = torch.as_tensor(qi, dtype=torch.int32)
        assert torch.equal(mq[off[b]:off[b + 1]], want), (b, mq[off[b]:off[b + 1]], want)
        assert int((want >= 0).sum()) == min(sizes[b], Q)


def _gemm_fuzz_case(rng, kn):
    import random
    M = rng.choice([1, 7, 32, 33, 64, 80, 130, 256, 264, 600, 1024])
    N = rng.choice([2, 32, 48, 64, 130, 256, 302, 1024])
    K = rng.choice([3, 24, 64, 70, 128, 130, 256, 600, 1000])
    ta, tb = rng.random() < 0.5, rng.random() < 0.5
    A = gen((K, M) if ta else (M, K), rng.randrange(10 ** 6))
    B = gen((N, K) if tb else (K, N), rng.randrange(10 ** 6))
    kw, desc = {}, []
    Aeff, Beff = (A.t() if ta else A).double(), (B.t() if tb else B).double()
    r = rng.random()
    if r < 0.25:
        A2 = gen(tuple(A.shape), rng.randrange(10 ** 6))
        kw["A2"] = A2
        Aeff = Aeff + (A2.t() if ta else A2).double()
        desc.append("A2")
    elif r < 0.5:
        B2 = gen(tuple(B.shape), rng.randrange(10 ** 6))
        kw["B2"] = B2
        Beff = Beff + (B2.t() if tb else B2).double()
        desc.append("B2")
    ref = Aeff @ Beff
    base = gen((M, N), rng.randrange(10 ** 6))
    C = base.clone()
    mode = rng.choice(["store", "rmw", "split"])
    if mode == "rmw":
        kw["accumulate"] = 1
        ref = ref + base.double()
    elif mode == "split":
        kw["split_k"] = rng.choice([2, 3, 4, 16])
        ref = ref + base.double()
    if rng.random() < 0.5:
        bias = gen((N,), rng.randrange(10 ** 6))
        kw["bias"] = bias
        ref = ref + bias.double()
        desc.append("bias")
    if rng.random() < 0.3:
        res = gen((M, N), rng.randrange(10 ** 6))
        kw["residual"] = res
        ref = ref + res.double()
        desc.append("res")
    colsum = None
    if rng.random() < 0.4:
        colsum = torch.zeros(M, device=dev())
        kw["colsum"] = colsum
        desc.append("colsum")
    kn.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
    tag = "M%d N%d K%d ta%d tb%d %s %s" % (M, N, K, ta, tb, mode, "+".join(desc))
    assert rel_err(C, ref) < TOL, tag
    if colsum is not None:
        assert rel_err(colsum, Aeff.sum(1)) < TOL, tag + " [colsum]"


def test_gemm_fuzz():
    """Random shapes x layouts x addends x accumulate modes x side outputs (all tile configs,
    K tails, split-K chunk tails)."""
    import random
    from mesm_amd import kernels as kn
    rng = random.Random(1234)
    for _ in range(300):
        _gemm_fuzz_case(rng, kn)


def test_gemm_group_matches_individual_launches():
    """mesm_gemm_group: independent problems of mixed shapes, layouts and fusions (small ones share
    launches, large / addend ones fall through) give exactly what the single launches give."""
    import random
    from mesm_amd import kernels as kn
    rng = random.Random(99)
    slope = torch.tensor([0.25], device=dev())
    for rep in range(6):
        calls = []
        for k in range(rng.choice([2, 5, 11])):
            M = rng.choice([32, 33, 320, 1024, 2400, 4800]); N = rng.choice([4, 130, 256, 512, 1024]); K = rng.choice([64, 70, 256, 320, 1024])
            ta, tb = rng.random() < 0.4, rng.random() < 0.5
            A = gen((K, M) if ta else (M, K), rng.randrange(10 ** 6))
            B = gen((N, K) if tb else (K, N), rng.randrange(10 ** 6), 0.1)
            kw = dict(trans_a=ta, trans_b=tb)
            r = rng.random()
            if r < 0.2: kw["bias"] = gen((N,), 5)
            elif r < 0.4: kw.update(residual=gen((M, N), 6), e_drop=(0.1, 9))
            elif r < 0.55: kw.update(aux=gen((M, N), 7), e_actgrad=kn.ACT_PRELU, slope=slope, dslope=torch.zeros(1, device=dev()))
            elif r < 0.7 and ta: kw.update(split_k=4, accumulate=2, colsum=torch.zeros(M, device=dev()))
            elif r < 0.8 and not ta: kw["A2"] = gen((M, K), 8)
            calls.append((A, B, kw))
        outs1, outs2, side1, side2 = [], [], [], []
        for A, B, kw in calls:
            kw1 = {k_: (v.clone() if k_ in ("colsum", "dslope") else v) for k_, v in kw.items()}
            M = A.shape[1] if kw["trans_a"] else A.shape[0]
            N = B.shape[0] if kw["trans_b"] else B.shape[1]
            C = torch.zeros(M, N, device=dev())
            kn.gemm(A, B, C, **kw1)
            outs1.append(C); side1.append([kw1.get("colsum"), kw1.get("dslope")])
        with kn.gemm_group():
            for A, B, kw in calls:
                kw2 = {k_: (v.clone() if k_ in ("colsum", "dslope") else v) for k_, v in kw.items()}
                M = A.shape[1] if kw["trans_a"] else A.shape[0]
                N = B.shape[0] if kw["trans_b"] else B.shape[1]
                C = torch.zeros(M, N, device=dev())
                kn.gemm(A, B, C, **kw2)
                outs2.append(C); side2.append([kw2.get("colsum"), kw2.get("dslope")])
        torch.cuda.synchronize()
        for a, b in zip(outs1, outs2):
            assert rel_err(b, a) < 1e-5
        for (A, B, kw), sa, sb in zip(calls, side1, side2):
            for name, x, y in zip(("colsum", "dslope"), sa, sb):
                if x is None:
                    continue
                if name == "dslope":
                    # a cancelling sum over M x N terms: the two routes may differ in arithmetic (a product that rides
                    # in the 64 x 64 split-bf16 launch against the f32 MFMA kernel it gets alone), so the bound is
                    # relative to the sum of the terms' magnitudes, not to the sum
                    prod = (A.t() if kw["trans_a"] else A).double() @ (B.t() if kw["trans_b"] else B).double()
                    scale = (prod * kw["aux"].double())[kw["aux"] < 0].abs().sum().item()
                    assert abs(float(y) - float(x)) < 2e-6 * scale, (float(x), float(y), scale)
                else:
                    assert rel_err(y, x) < 1e-4


@pytest.mark.parametrize("tile", ["0", "2", "3", "4", "6"])
def test_gemm_reference_widths_on_the_lds_dma_kernels(tile, monkeypatch):
    """Dv = 2818 and the 5003-word vocabulary: rows 8- / 4-byte aligned, K % 4 = 2 / 3, outer extents that
    are not multiples of 4 -- forward, input-gradient and weight-gradient forms, plus strided views."""
    from mesm_amd import kernels as kn
    monkeypatch.setenv("MESM_GEMM_TILE", tile)
    for (rows, wide, d) in [(300, 2818, 256), (200, 5003, 64)]:
        X = gen((rows, wide), 1); W = gen((d, wide), 2, 0.1); b = gen((d,), 3)
        Y = torch.empty(rows, d, device=dev())
        kn.gemm(X, W, Y, trans_b=True, bias=b, e_act=kn.ACT_RELU)                       # K = wide
        assert rel_err(Y, (X.double() @ W.double().t() + b.double()).clamp(min=0)) < TOL
        dY = gen((rows, d), 4)
        dX = torch.empty(rows, wide, device=dev())
        kn.gemm(dY, W, dX)                                                               # N = wide (outer-contiguous B)
        assert rel_err(dX, dY.double() @ W.double()) < TOL
        dW = torch.zeros(d, wide, device=dev()); db = torch.zeros(d, device=dev())
        kn.gemm(dY, X, dW, trans_a=True, colsum=db, split_k=4, accumulate=2)             # dW = dY^T X
        assert rel_err(dW, dY.double().t() @ X.double()) < TOL
        assert rel_err(db, dY.double().sum(0)) < TOL
        dWt = torch.zeros(wide, d, device=dev()); dbt = torch.zeros(wide, device=dev())
        kn.gemm(X, dY, dWt, trans_a=True, colsum=dbt, split_k=2, accumulate=2)           # M = wide (outer-contiguous A)
        assert rel_err(dWt, X.double().t() @ dY.double()) < TOL
        assert rel_err(dbt, X.double().sum(0)) < TOL
        Xv = X[:, 1:wide - 2]                                                            # odd start, K = wide - 3
        kn.gemm(Xv, W[:, 1:wide - 2], Y, trans_b=True)
        assert rel_err(Y, Xv.double() @ W[:, 1:wide - 2].double().t()) < TOL


@pytest.mark.parametrize("tile", ["0", "2", "3", "4", "6"])
def test_gemm_tiny_reduce_ranges_on_the_lds_dma_kernels(tile, monkeypatch):
    """K of 4..9 with outer extents that are not multiples of 4: the MFMA loop may be EMPTY (gemm_kmain = 0)
    and everything comes from the scalar tail, alone or split over k-slices, with bias column sums."""
    from mesm_amd import kernels as kn
    monkeypatch.setenv("MESM_GEMM_TILE", tile)
    seed = 0
    for K in (4, 5, 6, 7, 9):
        for (M, N) in ((33, 130), (66, 67), (130, 34)):
            for ta in (False, True):
                for tb in (False, True):
                    seed += 1
                    A = gen((K, M) if ta else (M, K), seed)
                    B = gen((N, K) if tb else (K, N), seed + 1000)
                    ref = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
                    C = torch.empty(M, N, device=dev())
                    kn.gemm(A, B, C, trans_a=ta, trans_b=tb)
                    assert rel_err(C, ref) < TOL, (K, M, N, ta, tb)
                    C2 = torch.zeros(M, N, device=dev()); cs = torch.zeros(M, device=dev())
                    kn.gemm(A, B, C2, trans_a=ta, trans_b=tb, split_k=2, accumulate=2, colsum=cs)
                    assert rel_err(C2, ref) < TOL, (K, M, N, ta, tb, "split")
                    assert rel_err(cs, (A.t() if ta else A).double().sum(1)) < TOL, (K, M, N, ta, tb, "colsum")


@pytest.mark.parametrize("tile", ["1", "2", "3", "4", "6"])
def test_gemm_fuzz_forced_kernel(tile, monkeypatch):
    """The same fuzz with every launch forced onto one of the small-problem / LDS-DMA kernels
    (1 = register fragments, 2 = wave-private LDS-DMA k-split, 3 = 64x64 LDS-DMA ring, 4 = 64x64 k-split,
    6 = tall 96x32 k-split); launches a
    kernel cannot take (addend operands, unaligned rows) fall through to the auto dispatch."""
    import random
    from mesm_amd import kernels as kn
    monkeypatch.setenv("MESM_GEMM_TILE", tile)
    rng = random.Random(4321 + int(tile))
    for _ in range(150):
        _gemm_fuzz_case(rng, kn)


def test_gemm_fuzz_activations():
    """Random shapes with the activation prologue / epilogue / gradient-epilogue features."""
    import random
    from mesm_amd import kernels as kn
    rng = random.Random(77)
    slope = torch.tensor([0.2], device=dev())
    for it in range(200):
        M = rng.choice([7, 24, 32, 80, 130, 256, 264, 600, 608])
        N = rng.choice([32, 64, 130, 256, 1024])
        K = rng.choice([24, 64, 70, 130, 256, 1024])
        A = gen((M, K), rng.randrange(10 ** 6))
        tb = rng.random() < 0.5
        B = gen((N, K) if tb else (K, N), rng.randrange(10 ** 6), 0.1)
        Bm = (B.t() if tb else B).double()
        feat = rng.choice(["e_relu", "e_prelu", "a_prelu", "grad_prelu", "grad_relu", "b_prelu"])
        C = torch.empty(M, N, device=dev())
        tag = "%s M%d N%d K%d tb%d" % (feat, M, N, K, tb)
        if feat == "e_relu":
            kn.gemm(A, B, C, trans_b=tb, e_act=kn.ACT_RELU)
            ref = (A.double() @ Bm).clamp(min=0)
        elif feat == "e_prelu":
            kn.gemm(A, B, C, trans_b=tb, e_act=kn.ACT_PRELU, slope=slope)
            z = A.double() @ Bm
            ref = torch.where(z > 0, z, 0.2 * z)
        elif feat == "a_prelu":
            kn.gemm(A, B, C, trans_b=tb, a_act=kn.ACT_PRELU, slope=slope)
            ref = torch.where(A > 0, A, 0.2 * A).double() @ Bm
        elif feat == "b_prelu":
            At = gen((K, M), rng.randrange(10 ** 6))
            Bn = gen((K, N), rng.randrange(10 ** 6))
            kn.gemm(At, Bn, C, trans_a=True, b_act=kn.ACT_PRELU, slope=slope)
            ref = At.t().double() @ torch.where(Bn > 0, Bn, 0.2 * Bn).double()
        else:
            aux = gen((M, N), rng.randrange(10 ** 6))
            ds = torch.zeros(1, device=dev())
            full = A.double() @ Bm
            if feat == "grad_prelu":
                kn.gemm(A, B, C, trans_b=tb, aux=aux, e_actgrad=kn.ACT_PRELU, slope=slope, dslope=ds)
                ref = torch.where(aux > 0, full, 0.2 * full)
                ds_ref = float((full * aux.double().clamp(max=0)).sum())
                assert abs(ds.item() - ds_ref) < 2e-4 * max(abs(ds_ref), full.abs().sum().item() * 1e-3), tag
            else:
                kn.gemm(A, B, C, trans_b=tb, aux=aux, e_actgrad=kn.ACT_RELU)
                ref = torch.where(aux > 0, full, torch.zeros_like(full))
        assert rel_err(C, ref) < TOL, tag


# --------------------------------------------------------------------------- round-2 fusions
@pytest.mark.parametrize("B,H,Lq,Lk,dh,dv", [(32, 8, 10, 75, 32, 32), (4, 8, 40, 75, 32, 32), (3, 4, 7, 130, 8, 8),
                                             (2, 2, 20, 33, 16, 16)])
def test_attention_split_heads_equal_the_interleaved_copy(B, H, Lq, Lk, dh, dv):
    """q2 / k2 (MesmAttnArgs): head h sees [q_h || q2_h], [k_h || k2_h] -- the decoder's per-head
    [content || position] concatenation (transformer.py:778-784) -- forward and all five gradients equal the
    kernel run on the materialised (B, L, 2d) tensors."""
    from mesm_amd import kernels as kn
    d = H * dh
    qc, qs = gen((B, Lq, d), 1), gen((B, Lq, d), 2)
    kc, kp = gen((B, Lk, d), 3), gen((B, Lk, d), 4)
    v, do = gen((B, Lk, H * dv), 5), gen((B, Lq, H * dv), 6)
    kpad = (torch.arange(Lk, device=dev())[None, :] >= torch.tensor([Lk - (i % 3) for i in range(B)], device=dev())[:, None])
    cat = lambda a, b, L: torch.cat([a.view(B, L, H, dh), b.view(B, L, H, dh)], -1).reshape(B, L, 2 * d)
    q_i, k_i = cat(qc, qs, Lq), cat(kc, kp, Lk)
    drop = (0.1, 77)
    o_ref, lse_ref = kn.attn_fwd(q_i, k_i, v, H, kpad=kpad, drop=drop)
    o, lse = kn.attn_fwd(qc, kc, v, H, kpad=kpad, drop=drop, q2=qs, k2=kp)
    assert torch.equal(o, o_ref) and torch.equal(lse, lse_ref)
    dq_i, dk_i, dv_ref = kn.attn_bwd(do, q_i, k_i, v, o_ref, lse_ref, H, kpad=kpad, drop=drop)
    dq, dk, dvv, dq2, dk2 = kn.attn_bwd(do, qc, kc, v, o, lse, H, kpad=kpad, drop=drop, q2=qs, k2=kp)
    assert rel_err(dvv, dv_ref) < 1e-6
    assert rel_err(cat(dq, dq2, Lq), dq_i) < 1e-5 and rel_err(cat(dk, dk2, Lk), dk_i) < 1e-5


@pytest.mark.parametrize("rows,D", [(4800, 256), (33, 32), (70, 300)])
def test_layernorm_second_output_and_gradient_joins(rows, D):
    """mesm_layernorm_fwd2 / _bwd3: y2 = y + add; dy <- dy + dyb on load; dx <- dx + addend on store; with the
    dropout-masked copy of dx as second gradient output."""
    from mesm_amd import kernels as kn
    x, g, b, add = gen((rows, D), 1), gen((D,), 2), gen((D,), 3), gen((rows, D), 4)
    y, mean, rstd, y2 = kn.layernorm_fwd(x, g, b, add=add)
    y0, _, _ = kn.layernorm_fwd(x, g, b)
    assert torch.equal(y, y0) and rel_err(y2, y0 + add) < 1e-6
    dy, dyb, addend = gen((rows, D), 5), gen((rows, D), 6), gen((rows, D), 7)
    dg_ref, db_ref = torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
    dx_ref = kn.layernorm_bwd(dy + dyb, x, g, mean, rstd, dg_ref, db_ref)
    dg, db = torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
    dx, dxm = kn.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dyb=dyb, addend=addend, drop2=(0.1, 9))
    assert rel_err(dx, dx_ref + addend) < 1e-5
    assert rel_err(dg, dg_ref) < 1e-4 and rel_err(db, db_ref) < 1e-4
    assert rel_err(dxm, kn.dropout(dx, 0.1, 9)) < 1e-6


def test_gemm_row_split_keeps_the_dropout_mask_and_every_epilogue_term():
    """kernels.gemm issues 4800-row x 256 outputs with K >= 1024 as a 4096-row launch (one full round of 64 x 64
    tiles) plus a remainder; the epilogue-dropout index carries the row offset (MesmGemmArgs.e_drop_row0), so the
    result -- mask included -- equals the single launch."""
    from mesm_amd import kernels as kn
    M, N, K = 4800, 256, 1024
    A, W = gen((M, K), 1, 0.1), gen((N, K), 2, 0.1)
    bias, res = gen((N,), 3), gen((M, N), 4)
    outs = []
    for split in (True, False):
        kn._SPLIT_ROWS = split
        try:
            C = torch.empty(M, N, device=dev())
            kn.gemm(A, W, C, trans_b=True, bias=bias, residual=res, e_drop=(0.1, 321))
            outs.append(C)
        finally:
            kn._SPLIT_ROWS = True
    assert rel_err(outs[0], outs[1]) < 1e-6
    assert float((outs[0] == res).float().mean()) > 0.05  # dropped elements (only the residual survives) exist


@pytest.mark.parametrize("M,tb", [(4800, True), (4864, False)])
def test_gemm_remainder_rows_split_along_k_onto_pool_zeroed_rows(M, tb):
    """kn.rows_out hands out a C whose remainder rows [4096, M) the step's one fill launch has cleared (ZeroPool.tail_zeroed:
    a step's requests are the ranges the next step's fill clears); gemm() then runs the remainder's tiles as four k-slices
    that meet by atomic adds -- bias and residual on the first slice, the same dropout mask on every slice -- and the result
    equals the unsplit launch; a first step (nothing cleared yet) and a no-grad call get plain tensors."""
    from mesm_amd import kernels as kn
    N, K = 256, 1024
    A = gen((M, K), 1, 0.1)
    W = gen((N, K), 2, 0.1) if tb else gen((K, N), 2, 0.1)
    bias, res = gen((N,), 3), gen((M, N), 4)
    like = torch.empty(M, N, device=dev())
    pool, saved = kn.ZeroPool(), kn.zero_pool
    kn.zero_pool = pool
    try:
        got = []
        for step in range(3):
            pool.begin(dev())
            z = kn.zeros((300,), dev())  # (the pool's ordinary users share the fill launch)
            C = kn.rows_out(like)
            pooled = pool.handed.get(C.data_ptr()) == (4096, M)
            assert pooled == (step > 0), step
            if pooled:
                assert float(C[4096:].abs().max()) == 0.0 and float(z.abs().max()) == 0.0
            C[:4096].fill_(7.0)  # (the rows above the cut may hold anything)
            kn.gemm(A, W, C, trans_b=tb, bias=bias, residual=res, e_drop=(0.1, 321))
            z += 1.0  # dirty the pool: the next step's fill has to clear it again
            got.append(C.clone())
            assert not pool.handed  # (the product took its registration: valid for one GEMM)
        pool.idle()  # (a forward without a fill, MESM._begin under no_grad: plain tensors)
        assert kn.rows_out(like).data_ptr() not in pool.handed and not pool.handed
        ref = torch.empty(M, N, device=dev())
        kn.gemm(A, W, ref, trans_b=tb, bias=bias, residual=res, e_drop=(0.1, 321))
        for c in got:
            assert rel_err(c, ref) < 1e-6
        assert torch.equal(got[1][:4096], ref[:4096])  # (only the remainder's summation order is free)
    finally:
        kn.zero_pool = saved


def test_layernorm_group_matches_individual_launches():
    """mesm_layernorm_{fwd,bwd}_group: the LayerNorms of one launch phase (different row counts, fused dropout,
    second output, gradient joins, masked second gradient) in shared launches give bit for bit what the plain
    launches give; a wide problem (D = 2818) in the same phase falls through to its own launch, the 512-wide ones
    share a launch of their own class."""
    from mesm_amd import kernels as kn
    D = 256
    specs = [dict(rows=32), dict(rows=1024, add=True), dict(rows=4800, drop=(0.5, 11)), dict(rows=75, dyb=True, addend=True),
             dict(rows=2400, drop2=(0.1, 5)), dict(rows=7, add=True, dyb=True), dict(rows=300, D=2818),
             dict(rows=33), dict(rows=640, addend=True, drop2=(0.1, 77)), dict(rows=1), dict(rows=129),
             # the second class (256 < D <= 512: two chunks per lane) and parameter-gradient-only members
             dict(rows=1024, D=512, drop=(0.5, 3), no_dx=True), dict(rows=32, D=512, no_dx=True), dict(rows=1, D=512),
             dict(rows=1, D=512, drop=(0.1, 9)), dict(rows=77, D=384, add=True, dyb=True), dict(rows=50, no_dx=True),
             dict(rows=3000, D=512, no_dx=True),
             # twins: the same wide x normalised once under two dropout masks (forward) / one pass over x for both parameter
             # gradients (backward)
             dict(rows=300, D=2818, drop=(0.5, 21), no_dx=True, twin_of=6), dict(rows=2400, D=2818, drop=(0.5, 5), no_dx=True),
             dict(rows=2400, D=2818, drop=(0.5, 6), no_dx=True, twin_of=19)]
    probs = []
    for i, sp in enumerate(specs):
        d = sp.get("D", D)
        x = gen((sp["rows"], d), 100 + i)
        g, b = gen((d,), 200 + i) + 1.0, gen((d,), 300 + i)
        if "twin_of" in sp:  # (same input and parameters as that problem: what the grouped entry points look for)
            x, g, b = probs[sp["twin_of"]][1], probs[sp["twin_of"]][2], probs[sp["twin_of"]][3]
        add = gen((sp["rows"], d), 400 + i) if sp.get("add") else None
        dy = gen((sp["rows"], d), 500 + i)
        dyb = gen((sp["rows"], d), 600 + i) if sp.get("dyb") else None
        addend = gen((sp["rows"], d), 700 + i) if sp.get("addend") else None
        probs.append((sp, x, g, b, add, dy, dyb, addend))

    def run_all(grouped):
        import contextlib
        outs = []
        ctx = kn.phase() if grouped else contextlib.nullcontext()
        fw = []
        with ctx:
            for sp, x, g, b, add, dy, dyb, addend in probs:
                fw.append(kn.layernorm_fwd(x, g, b, 1e-5, sp.get("drop", (0.0, 0)), add=add))
        ctx = kn.phase() if grouped else contextlib.nullcontext()
        shared = {}
        with ctx:
            for i, ((sp, x, g, b, add, dy, dyb, addend), f) in enumerate(zip(probs, fw)):
                dg, db = torch.zeros_like(g), torch.zeros_like(b)
                if "twin_of" in sp and sp.get("no_dx") and probs[sp["twin_of"]][0].get("no_dx"):
                    dg, db = shared[sp["twin_of"]]  # (one parameter: both gradients accumulate into the same views)
                shared[i] = (dg, db)
                r = kn.layernorm_bwd(dy, x, g, f[1], f[2], dg, db, drop=sp.get("drop", (0.0, 0)), drop2=sp.get("drop2"),
                                     dyb=dyb, addend=addend, need_dx=not sp.get("no_dx"))
                outs.append(list(f) + ([] if r is None else list(r) if isinstance(r, tuple) else [r]) + [dg, db])
        torch.cuda.synchronize()
        return outs

    ref, got = run_all(False), run_all(True)
    for (sp, *_), a, b in zip(probs, ref, got):
        for k, (x, y) in enumerate(zip(a, b)):
            if k >= len(a) - 2:  # dgamma / dbeta: float atomics, the order of the adders is free
                assert rel_err(y, x) < 1e-5, (sp, k)
            else:
                assert torch.equal(x, y), (sp, k)


def test_attention_group_matches_individual_launches():
    """mesm_attn_{fwd,bwd}_group: the attention cores of one launch phase (enhance 75 x 32, SS-MESM 1 x 75 and 4 x 300,
    MLM 32 x 20, t2v 75 x 33; quirk masks, dropout, stacked passes) in shared launches against the plain launches."""
    from mesm_amd import kernels as kn
    H, d = 8, 256
    cfgs = [(64, 75, 32, True, 32), (32, 1, 75, True, 0), (32, 32, 20, True, 0), (64, 75, 33, True, 32), (8, 4, 300, True, 0),
            (16, 76, 76, False, 0), (4, 130, 17, False, 0)]
    probs = []
    for i, (B, Lq, Lk, quirk, group) in enumerate(cfgs):
        q, k, v = gen((B, Lq, d), 10 + i), gen((B, Lk, d), 20 + i), gen((B, Lk, d), 30 + i)
        g = torch.Generator().manual_seed(40 + i)
        kpad = torch.rand(B, Lk, generator=g) < 0.2
        kpad[:, 0] = False
        qpad = (torch.rand(B, Lq, generator=g) < 0.2) if quirk else None
        do = gen((B, Lq, d), 50 + i)
        probs.append((q, k, v, kpad.to(dev()), qpad.to(dev()) if qpad is not None else None, group, do, (0.1, 1000 + i)))

    def run_all(grouped):
        import contextlib
        ctx = kn.phase() if grouped else contextlib.nullcontext()
        fw, outs = [], []
        with ctx:
            for q, k, v, kpad, qpad, group, do, drop in probs:
                fw.append(kn.attn_fwd(q, k, v, H, kpad=kpad, qpad=qpad, drop=drop, group=group))
        ctx = kn.phase() if grouped else contextlib.nullcontext()
        with ctx:
            for (q, k, v, kpad, qpad, group, do, drop), (o, lse) in zip(probs, fw):
                dq, dk, dv = torch.zeros_like(q), torch.empty_like(k), torch.empty_like(v)
                kn.attn_bwd_into(do, q, k, v, o, lse, H, dq, dk, dv, kpad=kpad, qpad=qpad, drop=drop, group=group)
                outs.append([o, lse, dq, dk, dv])
        torch.cuda.synchronize()
        return outs

    ref, got = run_all(False), run_all(True)
    for cfg, a, b in zip(cfgs, ref, got):
        for k, (x, y) in enumerate(zip(a, b)):
            ok = ~torch.isnan(x)  # rows whose keys are all masked are NaN in both
            assert torch.equal(torch.isnan(x), torch.isnan(y)), (cfg, k)
            # forward: the same kernel body; backward: 4-wave workgroups for every key-tile count (2 alone) and
            # float atomics into dq when there are several key tiles
            assert rel_err(y[ok], x[ok]) < (1e-6 if k < 2 else 2e-5), (cfg, k)


def test_slope_gradient_partials_ride_in_the_next_gemm_launch():
    """gemm.hip side_reduce: the PReLU slope-gradient partials of a dz GEMM are reduced by the NEXT GEMM launch on the
    stream (wave 0 of its first workgroup), not by a launch of their own -- pending after the producing phase, complete
    after the carrier, for every kernel that can carry (k-split 32 x 32 / 64 x 64, 64 x 64 ring, grouped); what nobody
    carried is completed by gemm_flush_side."""
    from mesm_amd import kernels as kn
    slope = torch.tensor([0.25], device=dev())
    for (M, N, K), carrier in (((320, 256, 256), (64, 64, 64)), ((4800, 1024, 256), (4096, 256, 1024)),
                               ((1024, 1024, 256), (4800, 256, 256)), ((320, 256, 256), None)):
        dY, W, Z = gen((M, K), 1), gen((K, N), 2, 0.1), gen((M, N), 3)
        ds = torch.zeros(1, device=dev())
        ref = float(((dY.double() @ W.double()) * Z.double().clamp(max=0)).sum())
        kn.defer_side(+1)
        try:
            with kn.phase():
                kn.gemm(dY, W, torch.empty(M, N, device=dev()), aux=Z, e_actgrad=kn.ACT_PRELU, slope=slope, dslope=ds)
            torch.cuda.synchronize()
            assert ds.item() == 0.0  # produced, not yet reduced
            if carrier is not None:
                m, n, k = carrier
                with kn.phase():
                    kn.gemm(gen((m, k), 4), gen((k, n), 5), torch.empty(m, n, device=dev()))
                    kn.gemm(gen((64, 32), 6), gen((32, 64), 7), torch.empty(64, 64, device=dev()))
            else:
                kn.gemm_flush_side()
            torch.cuda.synchronize()
            assert abs(ds.item() - ref) < 2e-4 * max(abs(ref), 1.0), (M, N, K, carrier, ds.item(), ref)
        finally:
            kn.defer_side(-1)
            kn.gemm_flush_side()


@pytest.mark.gpu
def test_grouped_64x64_split_launch_against_fp64():
    """The production form of a GEMM call: one gemm_wstage64_group_kernel launch (split-bf16 products) carrying a
    large product AND the small ones that ride along -- every layout pair, partial tiles in both extents, reduce ranges
    with a tail, split-K with column sums, bias / residual epilogues -- against fp64."""
    from mesm_amd import kernels as kn
    for ta in (False, True):
        for tb in (False, True):
            probs = []
            for (M, N, K, extra) in [(2433, 258, 262, "bias"), (1056, 256, 1030, "res"), (33, 256, 256, None), (1, 256, 70, "bias"),
                                     (320, 512, 64, None)]:
                A = gen((K, M) if ta else (M, K), M + 3 * K)
                B = gen((N, K) if tb else (K, N), N + 7 * K, 0.1)
                kw = dict(trans_a=ta, trans_b=tb)
                ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
                if extra == "bias":
                    kw["bias"] = gen((N,), 5)
                    ref = ref + kw["bias"].double()
                elif extra == "res":
                    kw["residual"] = gen((M, N), 6)
                    ref = ref + kw["residual"].double()
                probs.append((A, B, kw, ref, torch.zeros(M, N, device=dev())))
            if ta:  # a split-K weight-gradient form with its column sums rides along too
                A = gen((4800, 256), 11); B = gen((4800, 192), 12, 0.1)
                cs = torch.zeros(256, device=dev())
                probs.append((A, B, dict(trans_a=True, trans_b=False, split_k=4, accumulate=2, colsum=cs),
                              A.double().t() @ B.double(), torch.zeros(256, 192, device=dev())))
            with kn.gemm_group():
                for A, B, kw, ref, C in probs:
                    kn.gemm(A, B, C, **kw)
            torch.cuda.synchronize()
            for A, B, kw, ref, C in probs:
                err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
                assert err < 2e-6, (ta, tb, tuple(C.shape), A.shape, err)
                if "colsum" in kw:
                    assert rel_err(kw["colsum"], A.sum(0)) < 1e-5



@pytest.mark.gpu
def test_split_bf16_products_keep_non_finite_operands_non_finite():
    """the three-term split computes x - hi(x), so an operand of +/-inf turns into NaN inside the split (inf - inf)
    where the f32 matrix instruction and torch give +/-inf; a NaN operand stays NaN.  What is promised (and documented in
    DESIGN.md section 4): the result is NON-FINITE exactly where the f32 product is non-finite, finite and accurate
    everywhere else -- a diverged activation cannot come out of a GEMM looking healthy."""
    from mesm_amd import kernels as kn
    M, N, K = 4800, 256, 256   # takes the split-bf16 64 x 64 kernel
    A = gen((M, K), 901)
    B = gen((N, K), 902, 0.1)
    A[7, 13] = float("inf")
    A[100, 200] = float("-inf")
    A[4000, 0] = float("nan")
    B[5, 77] = float("inf")
    C = torch.zeros(M, N, device=dev())
    kn.gemm(A, B, C, trans_b=True)
    torch.cuda.synchronize()
    ref = A.double() @ B.double().t()
    bad_ref = ~torch.isfinite(ref)
    bad = ~torch.isfinite(C)
    assert torch.equal(bad, bad_ref), (int(bad.sum()), int(bad_ref.sum()))
    assert int(bad.sum()) >= 3 * N + M - 3   # rows 7, 100, 4000 and column 5
    ok = ~bad_ref
    assert float((C.double() - ref)[ok].abs().max()) / float(ref[ok].abs().max()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 5, 8])
def test_add_n_sums_up_to_eight_tensors_in_one_launch(k):
    from mesm_amd import kernels as kn
    ts = [gen((32, 10, 256), 950 + i) for i in range(k)]
    out = kn.add_n(ts)
    want = torch.stack(ts).double().sum(0)
    assert rel_err(out, want) < 1e-6


@pytest.mark.gpu
def test_fork_sums_the_consumers_gradients_like_autograd_does():
    """ops.fork: n aliases of a tensor whose gradients meet in ONE launch (mesm_add_n) -- same values as the autograd
    engine's pairwise adds, None gradients (an alias nobody used) skipped, fewer than three consumers left to autograd."""
    from mesm_amd import ops
    x = gen((32, 10, 256), 960).requires_grad_(True)
    ws = [gen((32, 10, 256), 961 + i) for i in range(5)]
    a = ops.fork(x, 6)
    assert len(a) == 6 and all(t.data_ptr() == x.data_ptr() for t in a)
    loss = sum((a[i] * ws[i]).sum() for i in range(5))   # the sixth alias stays unused
    loss.backward()
    x2 = x.detach().clone().requires_grad_(True)
    sum((x2 * ws[i]).sum() for i in range(5)).backward()
    assert rel_err(x.grad, x2.grad) < 1e-6
    y = gen((4, 8), 970).requires_grad_(True)
    assert all(t is y for t in ops.fork(y, 2))            # nothing to gain: no node


def test_glue_group_matches_individual_launches():
    """mesm_glue_group: the assembly problems of one launch phase (stacked copies, token mixes, gathers and their
    backward forms, the tiled sum) as workgroup ranges of one grid give bit for bit what the plain launches give."""
    from mesm_amd import kernels as kn
    N, L, D = 6, 11, 64
    g = torch.Generator().manual_seed(5)
    x = gen((N, L, D), 1)
    m1 = (torch.rand(N, L, generator=g) < 0.3).to(dev())
    m2 = (torch.rand(N, L, generator=g) < 0.2).to(dev())
    t1, t2 = gen((D,), 2), gen((D,), 3)
    idx = torch.randperm(N * L, generator=g)[:40].to(dev())
    valid = (torch.rand(40, generator=g) < 0.8).to(dev())
    inv = torch.full((N * L,), -1, dtype=torch.int64)
    inv[idx.cpu()] = torch.arange(40)
    inv = inv.to(dev())
    dy = gen((N, L, D), 4)
    dg = gen((40, D), 5)
    perm = torch.randperm(N, generator=g).to(dev())
    pad = (torch.rand(N, L, generator=g) < 0.5).to(dev())
    d2 = gen((2 * N, L, D), 6)
    b = gen((N, L, D), 7)

    def run(grouped):
        import contextlib
        with (kn.phase() if grouped else contextlib.nullcontext()), (kn.glue_deferred() if grouped else contextlib.nullcontext()):
            outs = list(kn.stack_rows([x, pad, b], [1, 1, 0], perm))
            outs.append(kn.token_mix_fwd(x, m1, t1, m2, t2))
            outs.append(kn.token_mix_fwd(b, m1, t1))
            y, rn = kn.gather_rows_fwd(x.view(-1, D), idx, valid, True)
            outs += [y, rn]
            outs.append(kn.gather_rows_fwd(b.view(-1, D), idx)[0])
            dx, dt1, dt2 = torch.empty_like(dy), torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
            kn.token_mix_bwd(dy, m1, m2, dx, dt1, dt2)
            outs += [dx, dt1, dt2]
            outs.append(kn.unstack_rows(d2, perm, N))
            outs.append(kn.unstack_rows(d2, None, N))
            outs.append(kn.add_tile(x, b, 2))
        # (second phase: the normalised gather's backward reads what the first phase wrote)
        with (kn.phase() if grouped else contextlib.nullcontext()), (kn.glue_deferred() if grouped else contextlib.nullcontext()):
            outs.append(kn.gather_rows_bwd(dg, y, rn, inv, valid, N * L, True))
            outs.append(kn.gather_rows_bwd(dg, None, None, inv, None, N * L, False))
        torch.cuda.synchronize()
        return outs

    ref, got = run(False), run(True)
    assert len(ref) == len(got) == 16
    for k, (a, c) in enumerate(zip(ref, got)):
        if k in (9, 10):  # the token gradients: float atomics, the order of the adders is free
            assert rel_err(c, a) < 1e-5, k
        else:
            assert torch.equal(a, c), k
    assert torch.equal(got[13].view(2, N, L, D)[1], x + b)


@pytest.mark.parametrize("M,N,K,tb", [(320, 256, 1024, False), (1024, 256, 5003, True), (64, 256, 1030, False)])
def test_split_k_takes_the_linear_epilogues(M, N, K, tb):
    """Split-K (partial sums by atomic adds onto a zeroed C) with every epilogue term that is linear in the partial sums:
    bias and residual (first slice), dropout (the same mask on every slice), the ReLU gradient (a 0 / 1 factor from aux) --
    what kernels.deep_out relies on; the result equals the unsplit launch.  A nonlinear epilogue is refused."""
    from mesm_amd import kernels as kn
    from mesm_amd._lib import ACT_RELU, ACT_PRELU, MesmError
    A = gen((M, K), 1, 0.1)
    W = gen((N, K), 2, 0.1) if tb else gen((K, N), 2, 0.1)
    bias, res, aux = gen((N,), 3), gen((M, N), 4), gen((M, N), 5)
    for kw in (dict(bias=bias, residual=res, e_drop=(0.2, 77)), dict(aux=aux, e_actgrad=ACT_RELU),
               dict(aux=aux, e_actgrad=ACT_RELU, e_drop=(0.5, 3), residual=res)):
        ref = torch.empty(M, N, device=dev())
        kn.gemm(A, W, ref, trans_b=tb, **kw)
        for sk in (2, 4):
            C = torch.zeros(M, N, device=dev())
            kn.gemm(A, W, C, trans_b=tb, split_k=sk, **kw)
            assert rel_err(C, ref) < 3e-6, (sorted(kw), sk)  # (another summation order over up to 5003 terms)
    with pytest.raises(MesmError):
        kn.gemm(A, W, torch.zeros(M, N, device=dev()), trans_b=tb, split_k=2, e_act=ACT_RELU)
    with pytest.raises(MesmError):
        kn.gemm(A, W, torch.zeros(M, N, device=dev()), trans_b=tb, split_k=2, aux=aux, e_actgrad=ACT_PRELU,
                slope=torch.full((1,), 0.25, device=dev()))


@pytest.mark.parametrize("B,L,D", [(64, 75, 256), (5, 3, 64), (2, 9, 132)])
def test_prepend_and_split_token_kernels(B, L, D):
    """mesm_prepend_* / mesm_split_token_* (a wave per row, four rows per workgroup; the shared token's gradient summed over
    the batch by one wave) against their torch statements: values bit-exact (copies and one add), gradients too where
    nothing is summed, the shared token's batch sums within rounding."""
    from mesm_amd import kernels as kn
    tok, ptok = gen((D,), 1), gen((D,), 2)
    x, pos = gen((B, L, D), 3), gen((B, L, D), 4)
    pad = (torch.rand(B, L, generator=torch.Generator().manual_seed(5)) < 0.3).to(dev())
    xo, po, xp, pado = kn.prepend_fwd(tok, x, ptok, pos, pad, True)
    assert torch.equal(xo, torch.cat([tok.expand(B, 1, D), x], 1)) and torch.equal(po, torch.cat([ptok.expand(B, 1, D), pos], 1))
    assert torch.equal(xp, xo + po) and torch.equal(pado, torch.cat([torch.ones(B, 1, dtype=pad.dtype, device=dev()), pad], 1))
    rows = gen((B, D), 6)  # a token per batch row
    xo2 = kn.prepend_fwd(rows, x, pad=pad, first_pad=False)[0]
    assert torch.equal(xo2, torch.cat([rows[:, None], x], 1))
    dxo, dxp, dpo = gen((B, L + 1, D), 7), gen((B, L + 1, D), 8), gen((B, L + 1, D), 9)
    dx, dtok, dptok = torch.empty(B, L, D, device=dev()), torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
    kn.prepend_bwd(dxo, dxp, dpo, dx, dtok, dptok, B, L, D, False)
    assert torch.equal(dx, (dxo + dxp)[:, 1:])
    assert rel_err(dtok, (dxo + dxp)[:, 0].double().sum(0).float()) < 1e-6
    assert rel_err(dptok, (dxp + dpo)[:, 0].double().sum(0).float()) < 1e-6
    dx2, drows = torch.empty(B, L, D, device=dev()), torch.empty(B, D, device=dev())
    kn.prepend_bwd(dxo, None, None, dx2, drows, None, B, L, D, True)
    assert torch.equal(dx2, dxo[:, 1:]) and torch.equal(drows, dxo[:, 0])
    mem = gen((B, L + 1, D), 10)
    Bd = B // 2
    g, loc, dec = kn.split_token_fwd(mem, Bd)
    assert torch.equal(g, mem[:, 0]) and torch.equal(loc, mem[:, 1:]) and (Bd == 0 or torch.equal(dec, mem[:Bd, 1:]))
    dg, dloc, ddec = gen((B, D), 11), gen((B, L, D), 12), gen((Bd, L, D), 13)
    dmem = kn.split_token_bwd(dg, dloc, ddec, B, L, D, Bd, dev())
    want = torch.cat([dg[:, None], dloc], 1)
    want[:Bd, 1:] += ddec
    assert torch.equal(dmem, want)
