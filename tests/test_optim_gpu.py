"""Flat AdamW + fused gradient clipping (mesm_amd/optim.py, csrc/optim.hip) against
nn.utils.clip_grad_norm_ + torch.optim.AdamW on the same gradients (train.py:68-72)."""
import argparse
import copy

import pytest
import torch

from golden_io import Fixture

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _setup():
    from mesm_amd import build_criterion, build_model, synthetic
    fx = Fixture("qvh_tiny")
    args = argparse.Namespace(**fx.cfg)
    args.device = "cuda:0"
    model = build_model(args)
    model.load_state_dict(fx.sd)
    crit = build_criterion(args)
    model.eval()
    batch = synthetic.to_device(fx.batch, dev())
    return fx, model, crit, batch


def _backward(fx, model, crit, batch):
    out = model(**batch, dataset_name=fx.cfg["dataset_name"], is_training=True,
                neg_index=fx.neg_index, masked_words=fx.masked_words)
    _, total = crit(out, batch, True)
    model.zero_grad(set_to_none=True)
    total.backward()
    return float(total)


@pytest.mark.parametrize("grad_clip", [0.1, 0.0])
def test_flat_adamw_matches_torch(grad_clip):
    from mesm_amd.optim import FlatAdamW
    fx, model, crit, batch = _setup()
    ref_params = {n: p.detach().clone().requires_grad_() for n, p in model.named_parameters()}
    ref_opt = torch.optim.AdamW([{"params": list(ref_params.values())}], lr=1e-3, weight_decay=1e-2)
    opt = FlatAdamW(model, lr=1e-3, weight_decay=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, 2, gamma=0.1)
    ref_sched = torch.optim.lr_scheduler.StepLR(ref_opt, 2, gamma=0.1)
    for step in range(4):
        _backward(fx, model, crit, batch)
        for n, p in model.named_parameters():
            ref_params[n].grad = None if p.grad is None else p.grad.detach().clone()
        if grad_clip > 0:
            want_norm = torch.nn.utils.clip_grad_norm_([p for p in ref_params.values() if p.grad is not None], grad_clip)
        ref_opt.step()
        opt.step(grad_clip=grad_clip)
        if grad_clip > 0:
            assert abs(float(opt.last_norm) - float(want_norm)) < 1e-5 * max(1.0, float(want_norm))
        worst = 0.0
        for n, p in model.named_parameters():
            a, b = p.detach().double(), ref_params[n].detach().double()
            worst = max(worst, float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6))
        assert worst < 2e-6, (step, worst)
        sched.step(); ref_sched.step()  # the learning rate drops after two steps
    # parameters without a gradient were left untouched (no decay), like torch skips p.grad is None
    for n, p in model.named_parameters():
        if p.grad is None:
            assert torch.equal(p.detach().cpu(), fx.sd[n].cpu()), n
    # checkpoint round trip in torch.optim.AdamW's format
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4.0
    opt.load_state_dict(sd)


def test_flat_clip_grad_norm():
    from mesm_amd.optim import clip_grad_norm_
    fx, model, crit, batch = _setup()
    _backward(fx, model, crit, batch)
    shadows = []
    for p in model.parameters():
        if p.grad is not None:
            s_ = torch.zeros_like(p).requires_grad_()
            s_.grad = p.grad.detach().clone()
            shadows.append(s_)
    want = torch.nn.utils.clip_grad_norm_(shadows, 0.1)
    got = clip_grad_norm_(model, 0.1)
    assert abs(float(got) - float(want)) < 1e-5 * max(1.0, float(want))
    k = 0
    for p in model.parameters():
        if p.grad is not None:
            assert torch.allclose(p.grad, shadows[k].grad, rtol=1e-5, atol=1e-9)
            k += 1
