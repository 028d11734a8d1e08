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


@pytest.mark.parametrize("nv,blocks", [(None, "set sal fw ss"), (21, "set sal fw ss"), (None, "sal"), (None, "set ss"),
                                       (5, "fw"), (None, "salwide fw")])
def test_one_launch_backward_equals_the_blocks_own_launches(nv, blocks):
    """mesm_criterion_bwd (every block's gradient kernel as a workgroup range of one grid, d total x weight and the NLL's
    row weights folded in) against the separate launches it replaces: same device functions, bit-exact."""
    from mesm_amd import kernels as kn
    N, Q, tmax, Lw, C, Lv, Le, D = 32, 10, 5, 32, 1503, 75, 33, 64
    L = 400 if "salwide" in blocks else 75  # (2L > 256: the 20-element saliency rows)
    n_valid = None if nv is None else torch.tensor([nv], dtype=torch.int32, device=dev())
    wv = gen((12,), 40).abs() + 0.1
    g = torch.tensor([0.37], device=dev())
    gv = kn.scale_vec(g, wv)
    want, kw = {}, {}
    if "set" in blocks:
        sizes, off, xx, cxw, _ = _targets(N, 41, tmax)
        lay = []
        for l in range(3):
            logits, spans = gen((N, Q, 2), 50 + l), torch.sigmoid(gen((N, Q, 2), 60 + l))
            mq = kn.set_loss_fwd(logits, spans, cxw, xx, off, tmax, 10.0, 1.0, 4.0, 0.1, torch.zeros(4, device=dev()),
                                 n_valid=n_valid)
            want["set%d" % l] = kn.set_loss_bwd(logits, spans, cxw, xx, off, mq, 0.1, gv[4 * l:4 * l + 3], n_valid=n_valid)
            lay.append((logits, spans, mq, torch.empty_like(logits), torch.empty_like(spans), 4 * l))
        kw["set_losses"] = dict(Q=Q, eos_coef=0.1, tgt_cxw=cxw, tgt_xx=xx, tgt_off=off, layers=lay)
    if "sal" in blocks:
        gg = torch.Generator().manual_seed(42)
        sp, sn = gen((N, L), 43), gen((N, L), 44)
        label = torch.randint(0, 5, (N, L), generator=gg).double().to(dev())
        vmask = (torch.rand(N, L, generator=gg) < 0.8).to(dev())
        pos_idx = torch.randint(0, L, (N, 2), generator=gg).to(dev())
        neg_idx = torch.randint(0, L, (N, 2), generator=gg).to(dev())
        want["sal"] = kn.saliency_loss_bwd(sp, sn, label, vmask, pos_idx, neg_idx, 12.0, 0.2, gv[10:11], n_valid=n_valid)
        kw["sal"] = dict(s_pos=sp, s_neg=sn, label=label, vmask=vmask, pos_idx=pos_idx, neg_idx=neg_idx, rank_coef=12.0,
                         margin=0.2, ds_pos=torch.empty_like(sp), ds_neg=torch.empty_like(sn), slot=10)
    if "fw" in blocks:
        logit = gen((N, Lw, C), 45)
        label = torch.randint(0, C, (N, Lw), generator=torch.Generator().manual_seed(46)).to(dev()).view(-1)
        lens = torch.tensor([1 + (5 * i) % Lw for i in range(N)])
        mask = (torch.arange(Lw)[None] < lens[:, None]).to(dev())
        _, row_lse, _ = kn.nll_smooth_fwd(logit.view(-1, C), label, mask.view(-1), 0.1)
        rg = kn.rec_fw_rowgrad(mask, gv[9:10], n_valid=n_valid)
        want["fw"] = kn.nll_smooth_bwd(logit.view(-1, C), label, row_lse, rg, 0.1).view(N, Lw, C)
        kw["recfw"] = dict(logit=logit, label=label, row_lse=row_lse, mask=mask, eps=0.1, dlogit=torch.empty_like(logit),
                           slot=9)
    if "ss" in blocks:
        gg = torch.Generator().manual_seed(47)
        pv, ew = gen((N, Lv, D), 48), gen((N, Le, D), 49)
        cmask = torch.rand(N, Lv, generator=gg) < 0.3
        cmask[:, 0] = True
        wmask = torch.rand(N, Le, generator=gg) < 0.7
        wmask[:, 0] = True
        pos = (torch.rand(N, N, generator=gg) < 0.2) | torch.eye(N, dtype=torch.bool)
        cmask, wmask, pos8 = cmask.to(dev()), wmask.to(dev()), pos.to(torch.uint8).to(dev())
        saved = kn.rec_ss_fwd(pv, cmask, ew, wmask, pos8, 0.5, torch.zeros(1, device=dev()), n_valid=n_valid)
        want["ss"] = kn.rec_ss_bwd(saved, pos8, cmask, wmask, Lv, Le, 0.5, gv[11:12], n_valid=n_valid)
        kw["recss"] = dict(saved=saved, pos=pos8, cmask=cmask, wmask=wmask, Lv=Lv, Le=Le, tau=0.5,
                           dpv=torch.empty(N, Lv, D, device=dev()), dew=torch.empty(N, Le, D, device=dev()), slot=11)
    kn.criterion_bwd(g, wv, N, n_valid=n_valid, **kw)
    torch.cuda.synchronize()
    if "set" in blocks:
        for l, lay_l in enumerate(kw["set_losses"]["layers"]):
            assert torch.equal(lay_l[3], want["set%d" % l][0]) and torch.equal(lay_l[4], want["set%d" % l][1])
    if "sal" in blocks:
        assert torch.equal(kw["sal"]["ds_pos"], want["sal"][0]) and torch.equal(kw["sal"]["ds_neg"], want["sal"][1])
    if "fw" in blocks:
        assert torch.equal(kw["recfw"]["dlogit"], want["fw"])
    if "ss" in blocks:
        assert torch.equal(kw["recss"]["dpv"], want["ss"][0]) and torch.equal(kw["recss"]["dew"], want["ss"][1])


@pytest.mark.parametrize("nv,blocks,C", [(None, "set sal fw ss", 1503), (21, "set sal fw ss", 5003), (None, "sal", 0),
                                         (None, "set ss", 0), (5, "fw", 6001), (None, "salwide fw", 1022)])
def test_three_launch_forward_equals_the_blocks_own_launches(nv, blocks, C):
    """mesm_criterion_fwd (first stages of every block in one grid of 1,024-thread workgroups, rec_ss' similarity rows,
    one finishing workgroup) against the separate launches it replaces: loss vector, total and everything the backward
    reads, bit-exact.  C covers the three row forms of the NLL (<= 2048, <= 5120, any); "salwide" the saliency rows that
    stay a launch of their own."""
    from mesm_amd import kernels as kn
    N, Q, tmax, Lw, Lv, Le, D = 32, 10, 5, 30, 75, 33, 64
    L = 400 if "salwide" in blocks else 75
    n_valid = None if nv is None else torch.tensor([nv], dtype=torch.int32, device=dev())
    wv = gen((16,), 40).abs() + 0.1
    wv[3] = 0.0  # (class_error: a logged value, not part of the total)
    lv_a = torch.full((16,), 7.0, device=dev())
    lv_b = lv_a.clone()
    kw, want = {}, {}
    if "set" in blocks:
        sizes, off, xx, cxw, _ = _targets(N, 41, tmax)
        lay = [(gen((N, Q, 2), 50 + l), torch.sigmoid(gen((N, Q, 2), 60 + l)), 4 * l) for l in range(3)]
        want["match"] = kn.set_loss_fwd_layers([(lg, sp, lv_a[s:s + 4]) for lg, sp, s in lay], cxw, xx, off, tmax, 10.0, 1.0,
                                               4.0, 0.1, n_valid=n_valid)
        kw["set_losses"] = dict(Q=Q, Tmax=tmax, w_span=10.0, w_giou=1.0, w_class=4.0, eos_coef=0.1, tgt_cxw=cxw, tgt_xx=xx,
                                tgt_off=off, layers=lay)
    if "sal" in blocks:
        gg = torch.Generator().manual_seed(42)
        sp, sn = gen((N, L), 43), gen((N, L), 44)
        label = torch.randint(0, 5, (N, L), generator=gg).double().to(dev())
        vmask = (torch.rand(N, L, generator=gg) < 0.8).to(dev())
        pos_idx = torch.randint(0, L, (N, 2), generator=gg).to(dev())
        neg_idx = torch.randint(0, L, (N, 2), generator=gg).to(dev())
        kn.saliency_loss_fwd(sp, sn, label, vmask, pos_idx, neg_idx, 12.0, 0.2, out=lv_a[12:13], n_valid=n_valid)
        kw["sal"] = dict(s_pos=sp, s_neg=sn, label=label, vmask=vmask, pos_idx=pos_idx, neg_idx=neg_idx, rank_coef=12.0,
                         margin=0.2, slot=12)
    if "fw" in blocks:
        logit = gen((N, Lw, C), 45)
        label = torch.randint(0, C, (N, Lw), generator=torch.Generator().manual_seed(46)).to(dev()).view(-1)
        lens = torch.tensor([1 + (5 * i) % Lw for i in range(N)])
        mask = (torch.arange(Lw)[None] < lens[:, None]).to(dev())
        row_loss, row_lse, correct = kn.nll_smooth_fwd(logit.view(-1, C), label, mask.view(-1), 0.1)
        kn.rec_fw_reduce(row_loss, correct, mask, lv_a[13:15], n_valid=n_valid)
        want["row_lse"] = row_lse
        kw["recfw"] = dict(logit=logit, label=label, mask=mask, eps=0.1, slot=13)
    if "ss" in blocks:
        gg = torch.Generator().manual_seed(47)
        pv, ew = gen((N, Lv, D), 48), gen((N, Le, D), 49)
        cmask = torch.rand(N, Lv, generator=gg) < 0.3
        cmask[:, 0] = True
        wmask = torch.rand(N, Le, generator=gg) < 0.7
        wmask[:, 0] = True
        pos = (torch.rand(N, N, generator=gg) < 0.2) | torch.eye(N, dtype=torch.bool)
        cmask, wmask, pos8 = cmask.to(dev()), wmask.to(dev()), pos.to(torch.uint8).to(dev())
        want["recss"] = kn.rec_ss_fwd(pv, cmask, ew, wmask, pos8, 0.5, lv_a[15:16], n_valid=n_valid)
        kw["recss"] = dict(pv=pv, cmask=cmask, ew=ew, wmask=wmask, pos=pos8, tau=0.5, slot=15)
    total_a = kn.weighted_sum(lv_a, wv)
    total_b, out = kn.criterion_fwd(lv_b, wv, N, n_valid=n_valid, **kw)
    torch.cuda.synchronize()
    assert torch.equal(lv_a, lv_b), (lv_a, lv_b)
    assert torch.equal(total_a, total_b) and total_b.shape == total_a.shape
    rows = N if nv is None else nv
    if "set" in blocks:
        T = int(off[rows])  # (the padding pairs' targets are not matched)
        for a, b in zip(want["match"], out["match"]):
            assert torch.equal(a[:T], b[:T])
    if "fw" in blocks:
        assert torch.equal(want["row_lse"], out["row_lse"])
    if "ss" in blocks:
        cn, wn, stats, sim = want["recss"]
        assert torch.equal(cn, out["recss"][0]) and torch.equal(wn, out["recss"][1])
        assert torch.equal(stats[:N], out["recss"][2][:N])
        assert torch.equal(stats.view(-1)[4 * N:4 * N + rows], out["recss"][2].view(-1)[4 * N:4 * N + rows])  # (row terms)
        assert torch.equal(sim[:rows, :rows], out["recss"][3][:rows, :rows])
