"""The HIP path against the real reference on the switches no shipped config flips (tests/golden/variants,
tools/gen_golden_variants.py): FW-MESM / SS-MESM off, aux_loss off, other layer counts and input-projection
depths.  Same bar as the shipped-config fixtures: outputs / losses 1e-4, matcher bit-exact, gradients 5e-4."""
import pytest
import torch

from golden_io import VARIANTS, Fixture
from test_model_gpu import TOL, build, dev, rel, run_step
from test_variants_cpu import compare_outputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=VARIANTS)
def variant_step(request):
    fx = Fixture(request.param)
    args, model, crit = build(fx.cfg, fx.sd)
    out, losses, total = run_step(model, crit, fx.batch, fx.cfg, fx.neg_index, fx.masked_words)
    return fx, model, crit, out, losses, total


def test_outputs_and_key_set(variant_step):
    fx, _, _, out, _, _ = variant_step
    compare_outputs(out, fx, lambda a, b: rel(a, b) < TOL)


def test_losses(variant_step):
    fx, _, _, _, losses, total = variant_step
    assert set(losses) == set(fx.losses) - {"total"}, set(losses) ^ set(fx.losses)
    for k, v in fx.losses.items():
        got = float(total.detach()) if k == "total" else float(losses[k].detach())
        assert abs(got - v) < TOL * max(1.0, abs(v)), (k, got, v)


def test_matched_indices_bit_exact(variant_step):
    fx, _, crit, _, _, _ = variant_step
    qvh = fx.cfg["dataset_name"] == "qvhighlights"
    layers = ["main"] + ["aux%d" % i for i in range(len(crit.last_match) - 1)]
    assert len(layers) == 1 + len({k.split(".")[0] for k in fx.match if k.startswith("aux")})
    for layer, mq in zip(layers, crit.last_match):
        mq = mq.cpu().tolist()
        got, k = set(), 0
        for b, s in enumerate(fx.match["%s.sizes" % layer].tolist()):
            for t in range(s):
                got.add((b, mq[k], t if qvh else 0))
                k += 1
        assert got == fx.matched_pairs(layer), layer


def test_gradients(variant_step):
    fx, model, _, _, _, _ = variant_step
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(grads) == set(fx.grads), sorted(set(grads) ^ set(fx.grads))
    worst = max((rel(grads[k], g), k) for k, g in fx.grads.items())
    assert worst[0] < 5 * TOL, worst


def test_graph_replay_of_a_variant_equals_eager():
    """the captured step (arena plan, static buffers) on a model without the MLM / SS branches"""
    from mesm_amd import synthetic
    from mesm_amd.graphed import GraphedStep
    for name in ("variants/qvh_plain", "variants/cha_ss_only", "variants/qvh_no_aux", "variants/qvh_txt_pos"):
        fx = Fixture(name)
        args, model, crit = build(fx.cfg, fx.sd)
        model.eval()  # dropout off; is_training=True keeps the MLM branch where the config has one
        batch = synthetic.to_device(fx.batch, dev())
        g = GraphedStep(model, crit, batch, fx.cfg["dataset_name"], warmup=1)
        g.set_draws(fx.neg_index, fx.masked_words)
        total = float(g.run(redraw=False))
        torch.cuda.synchronize()
        assert abs(total - fx.losses["total"]) < TOL * max(1.0, abs(fx.losses["total"])), (name, total)
        worst = max((rel(p.grad, fx.grads[n]), n) for n, p in model.named_parameters() if n in fx.grads)
        assert worst[0] < 5 * TOL, (name, worst)


@pytest.mark.parametrize("name", VARIANTS)
def test_train_mode_step_is_finite_and_touches_the_same_parameters(name):
    """dropout on (no reference values to compare with): the step runs, stays finite, and the set of parameters
    that receive a gradient is the reference's"""
    fx = Fixture(name)
    args, model, crit = build(fx.cfg, fx.sd)
    out, losses, total = run_step(model, crit, fx.batch, fx.cfg, fx.neg_index, fx.masked_words, train=True)
    assert torch.isfinite(total)
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(grads) == set(fx.grads), sorted(set(grads) ^ set(fx.grads))
    assert all(bool(torch.isfinite(g).all()) for g in grads.values())
