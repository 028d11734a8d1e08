"""Fused criterion kernels (csrc/criterion.hip) against plain-torch autograd statements of the
same reference formulas (tests/ref_losses.py) on the MI355X: values and gradients."""
import pytest
import torch

import ref_losses as R

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev():
    return torch.device("cuda:0")


def gen(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


def close(a, b, tol=TOL):
    a, b = a.double().cpu(), b.double().cpu()
    scale = max(b.abs().max().item(), 1e-6)
    return (a - b).abs().max().item() / scale < tol


def _targets(N, seed, tmax=5):
    g = torch.Generator().manual_seed(seed)
    sizes = [1 + (i % tmax) for i in range(N)]
    off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32)
    T = int(off[-1])
    st = torch.rand(T, generator=g) * 0.6
    ed = st + 0.05 + torch.rand(T, generator=g) * 0.3
    xx = torch.stack([st, ed], 1).to(dev())
    cxw = torch.stack([(st + ed) * 0.5, ed - st], 1).to(dev())
    pair_of_t = torch.repeat_interleave(torch.arange(N), torch.tensor(sizes)).to(dev())
    return sizes, off.to(dev()), xx, cxw, pair_of_t


@pytest.mark.parametrize("N,Q,tmax", [(32, 10, 5), (3, 10, 1), (70, 12, 4), (16, 32, 16),
                                      (9, 6, 9), (7, 7, 7), (12, 40, 20), (5, 64, 64), (4, 3, 64)])
def test_set_loss_matches_match_kernel_and_torch(N, Q, tmax):
    """the last rows: more targets than queries (targets left unmatched, the denominators count the matched
    pairs), square blocks, and the largest extents the kernel takes -- the reference's scipy call has no shape
    rule (matcher.py:108-117)"""
    from mesm_amd import kernels as kn
    sizes, off, xx, cxw, pair_of_t = _targets(N, 5 + N, tmax)
    logits = gen((N, Q, 2), 96 + N).requires_grad_()
    spans = torch.sigmoid(gen((N, Q, 2), 97 + N)).requires_grad_()
    out4 = torch.zeros(4, device=dev())
    mq = kn.set_loss_fwd(logits.detach(), spans.detach(), cxw, xx, off, tmax, 10.0, 1.0, 4.0, 0.1, out4)
    mq_ref = kn.match(logits.detach(), spans.detach(), cxw, xx, off, tmax, 10.0, 1.0, 4.0)
    assert torch.equal(mq.cpu(), mq_ref.cpu())  # integer result: bit-exact
    ls, lg, ll, ce = R.set_losses(logits, spans, cxw, xx, pair_of_t, mq_ref, 0.1)
    ref = torch.stack([ls, lg, ll, ce]).detach()
    assert close(out4, ref), (out4, ref)
    g3 = torch.tensor([0.7, -1.3, 2.1], device=dev())
    (g3[0] * ls + g3[1] * lg + g3[2] * ll).backward()
    dl, ds = kn.set_loss_bwd(logits.detach(), spans.detach(), cxw, xx, off, mq, 0.1, g3)
    assert close(dl, logits.grad) and close(ds, spans.grad)


@pytest.mark.parametrize("N,Lv,Le,D", [(32, 75, 33, 256), (5, 20, 9, 32), (16, 512, 17, 256)])
def test_rec_ss_value_and_gradients(N, Lv, Le, D):
    from mesm_amd import kernels as kn
    pv = gen((N, Lv, D), 1).requires_grad_()
    ew = gen((N, Le, D), 2).requires_grad_()
    g = torch.Generator().manual_seed(3)
    cmask = torch.rand(N, Lv, generator=g) < 0.3
    cmask[:, 0] = True
    wmask = torch.rand(N, Le, generator=g) < 0.7
    wmask[:, 0] = True
    pos = torch.rand(N, N, generator=g) < 0.2
    pos |= torch.eye(N, dtype=torch.bool)
    pos[N - 1] = False  # a row without positives
    cmask, wmask, pos = cmask.to(dev()), wmask.to(dev()), pos.to(dev())
    ref = R.rec_ss(pv, cmask, ew, wmask, pos, 0.5)
    ref.backward()
    out = torch.zeros(1, device=dev())
    pos8 = pos.to(torch.uint8).contiguous()
    saved = kn.rec_ss_fwd(pv.detach(), cmask, ew.detach(), wmask, pos8, 0.5, out)
    assert close(out, ref.detach().reshape(1)), (out, ref)
    gs = torch.ones(1, device=dev())
    dpv, dew = kn.rec_ss_bwd(saved, pos8, cmask, wmask, Lv, Le, 0.5, gs)
    assert close(dpv, pv.grad, 2e-4) and close(dew, ew.grad, 2e-4)


def test_rec_fw_reduce_and_rowgrad():
    from mesm_amd import kernels as kn
    N, Lw, C = 32, 32, 5003
    logit = gen((N, Lw, C), 4).requires_grad_()
    label = torch.randint(0, C, (N, Lw), generator=torch.Generator().manual_seed(5)).to(dev())
    lens = torch.tensor([4 + (5 * i) % (Lw - 3) for i in range(N)])
    mask = (torch.arange(Lw)[None] < lens[:, None]).to(dev())
    loss_ref, acc_ref = R.rec_fw(logit, label, mask)
    loss_ref.backward()
    row_loss, row_lse, correct = kn.nll_smooth_fwd(logit.detach().view(-1, C), label.view(-1), mask.view(-1), 0.1)
    out2 = torch.zeros(2, device=dev())
    kn.rec_fw_reduce(row_loss, correct, mask, out2)
    assert close(out2[:1], loss_ref.detach().reshape(1)) and close(out2[1:], acc_ref.reshape(1))
    g = torch.ones(1, device=dev())
    rg = kn.rec_fw_rowgrad(mask, g)
    dl = kn.nll_smooth_bwd(logit.detach().view(-1, C), label.view(-1), row_lse, rg, 0.1)
    assert close(dl.view(N, Lw, C), logit.grad)


def test_rowdot():
    from mesm_amd import ops
    N, L, D = 32, 75, 256
    a = gen((N, L, D), 6).requires_grad_()
    b = gen((N, D), 7).requires_grad_()
    s = ops.rowdot(a, b, 1.0 / 16.0)
    ref = torch.sum(a.double() * b.double().unsqueeze(1), dim=-1) / 16.0
    assert close(s, ref)
    w = gen((N, L), 8)
    (s * w).sum().backward()
    ga, gb = a.grad.clone(), b.grad.clone()
    a.grad = b.grad = None
    (ref.float() * w).sum().backward()
    assert close(ga, a.grad) and close(gb, b.grad)


@pytest.mark.parametrize("normalize", [True, False])
def test_text_prep(normalize):
    from mesm_amd import kernels as kn
    N, Lw, D = 32, 32, 512
    x = gen((N, Lw, D), 9)
    lens = torch.tensor([4 + (5 * i) % (Lw - 3) for i in range(N)])
    x = x * (torch.arange(Lw)[None] < lens[:, None]).to(dev()).unsqueeze(-1)
    w, m, s = kn.text_prep(x, normalize)
    rw, rm, rs = R.post_process_text(x, normalize)
    assert torch.equal(m.cpu(), rm.cpu())
    assert close(w, rw) and close(s, rs)


def test_weighted_sum_and_scale_vec():
    from mesm_amd import kernels as kn
    v = gen((13,), 10)
    v[3] = float("nan")  # a logged-only slot (weight 0) may be non-finite
    w = gen((13,), 11)
    w[3] = 0.0
    t = kn.weighted_sum(v, w)
    keep = w != 0
    assert close(t.reshape(1), (v[keep].double() * w[keep].double()).sum().reshape(1))
    g = torch.tensor([2.5], device=dev())
    assert close(kn.scale_vec(g, w), 2.5 * w)
