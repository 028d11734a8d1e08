"""The CPU oracle against the real reference on the switches no shipped config flips (tests/golden/variants,
tools/gen_golden_variants.py): FW-MESM / SS-MESM off (model.py:175-184, 264-299, 307-352), aux_loss off
(model.py:304, 340), other layer counts and input-projection depths (model.py:49-62)."""
import pytest
import torch

from golden_io import VARIANTS, Fixture
from oracle import mesm_oracle as O

TOL = 2e-5


def close(a, b, tol=TOL):
    scale = max(float(b.abs().max()), 1.0)
    return float((a.double() - b.double()).abs().max()) / scale < tol


@pytest.fixture(scope="module", params=VARIANTS)
def step(request):
    fx = Fixture(request.param)
    out, losses, total, grads, idx = O.train_step(fx.sd, fx.cfg, fx.batch, fx.neg_index, fx.masked_words)
    return fx, out, losses, total, grads, idx


def compare_outputs(out, fx, close_fn):
    """every tensor the reference returned (the key set itself is part of the contract)"""
    ref_keys = {k for k in fx.out if not k.startswith("aux")}
    got_keys = {k for k, v in out.items() if torch.is_tensor(v)}
    assert got_keys == ref_keys, got_keys ^ ref_keys
    for k in sorted(ref_keys):
        r = fx.out[k]
        g = out[k].detach().cpu()
        if r.dtype == torch.bool or g.dtype == torch.bool:
            assert torch.equal(g.bool(), r.bool()), k
        else:
            assert close_fn(g, r), k
    n_aux = len({k.split(".")[0] for k in fx.out if k.startswith("aux")})
    assert len(out.get("aux_outputs", [])) == n_aux
    for i, a in enumerate(out.get("aux_outputs", [])):
        for k, v in a.items():
            assert close_fn(v.detach().cpu(), fx.out["aux%d.%s" % (i, k)]), (i, k)


def test_outputs_match_reference(step):
    fx, out, *_ = step
    compare_outputs(out, fx, close)


def test_losses_match_reference(step):
    fx, _, losses, total, _, _ = step
    assert set(losses) == set(fx.losses) - {"total"}, set(losses) ^ set(fx.losses)
    for k, v in fx.losses.items():
        got = float(total.detach()) if k == "total" else float(losses[k].detach())
        assert abs(got - v) < 1e-4 * max(1.0, abs(v)), (k, got, v)


def test_matching_is_bit_exact(step):
    fx, _, _, _, _, idx = step
    layers = ["main"] + ["aux%d" % i for i in range(len(idx) - 1)]
    for layer, ind in zip(layers, idx):
        got = set()
        for b, (q, t) in enumerate(ind):
            for qq, tt in zip(q.tolist(), t.tolist()):
                got.add((b, qq, tt))
        assert got == fx.matched_pairs(layer), layer


def test_gradients_match_reference(step):
    fx, _, _, _, grads, _ = step
    assert set(grads) == set(fx.grads), set(grads) ^ set(fx.grads)
    for k, g in fx.grads.items():
        assert close(grads[k], g, 1e-4), k
