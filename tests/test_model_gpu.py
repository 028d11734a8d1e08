"""End-to-end parity of the HIP build on the MI355X:

  * against the golden vectors produced by the REAL reference (tests/golden/*.npz): outputs,
    losses, matched indices (bit-exact), parameter gradients;
  * against the CPU oracle on freshly seeded inputs at the reference hidden size (d = 256);
  * train-mode (dropout on) consistency of the fused autograd blocks.

Tolerance: 1e-4 relative to the tensor scale for fp32 values (north-star budget); integer
results (matched query indices) must be identical.
"""
import argparse

import pytest
import torch

from golden_io import CASES, EvalFixture, Fixture

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-3)


def build(cfg, sd):
    from mesm_amd import build_criterion, build_model
    args = argparse.Namespace(**cfg)
    args.device = "cuda:0"
    model = build_model(args)
    model.load_state_dict(sd)
    crit = build_criterion(args)
    return args, model, crit


def run_step(model, crit, batch_cpu, cfg, neg_index, masked_words, train=False):
    from mesm_amd import synthetic
    batch = synthetic.to_device(batch_cpu, dev())
    model.train(train)
    out = model(**batch, dataset_name=cfg["dataset_name"], is_training=True, neg_index=neg_index,
                masked_words=masked_words)
    losses, total = crit(out, batch, True)
    model.zero_grad()
    total.backward()
    torch.cuda.synchronize()
    return out, losses, total


@pytest.fixture(scope="module", params=CASES)
def golden_step(request):
    fx = Fixture(request.param)
    args, model, crit = build(fx.cfg, fx.sd)
    out, losses, total = run_step(model, crit, fx.batch, fx.cfg, fx.neg_index, fx.masked_words)
    return fx, model, crit, out, losses, total


def test_outputs_match_reference_golden(golden_step):
    fx, _, _, out, _, _ = golden_step
    for k in ("pred_logits", "pred_spans", "saliency_scores", "neg_saliency_scores",
              "recfw_words_logit", "recon_feat", "projed_recon_feat", "projed_video_feat",
              "expanded_words_feat", "enhanced_video_feat", "projed_words_feat"):
        assert rel(out[k], fx.out[k]) < TOL, k
    assert rel(out["aux_outputs"][0]["pred_logits"], fx.out["aux0.pred_logits"]) < TOL
    assert rel(out["aux_outputs"][0]["pred_spans"], fx.out["aux0.pred_spans"]) < TOL
    assert torch.equal(out["expanded_words_mask"].cpu(), fx.out["expanded_words_mask"].bool())
    assert torch.equal(out["words_mask"].cpu(), fx.out["words_mask"].bool())


def test_losses_match_reference_golden(golden_step):
    fx, _, _, _, losses, total = golden_step
    for k, v in fx.losses.items():
        got = float(total) if k == "total" else float(losses[k])
        assert abs(got - v) < TOL * max(1.0, abs(v)), (k, got, v)


def test_matched_indices_bit_exact(golden_step):
    fx, _, crit, _, _, _ = golden_step
    qvh = fx.cfg["dataset_name"] == "qvhighlights"
    for layer, mq in zip(["main", "aux0"], crit.last_match):
        mq = mq.cpu().tolist()
        sizes = fx.match["%s.sizes" % layer].tolist()
        got, k = set(), 0
        for b, s in enumerate(sizes):
            for t in range(s):
                got.add((b, mq[k], t if qvh else 0))
                k += 1
        assert got == fx.matched_pairs(layer), layer


def test_matcher_module_returns_reference_format(golden_step):
    fx, _, crit, out, _, _ = golden_step
    from mesm_amd import synthetic
    batch = synthetic.to_device(fx.batch, dev())
    res = crit.matcher({k: v for k, v in out.items() if k != "aux_outputs"}, batch)
    if fx.cfg["dataset_name"] == "qvhighlights":
        q = torch.cat([a for a, _ in res])
        t = torch.cat([b for _, b in res])
        assert torch.equal(q, fx.match["main.q"]) and torch.equal(t, fx.match["main.t"])
    else:
        assert torch.equal(res[:, 0], fx.match["main.q"]) and torch.equal(res[:, 1], fx.match["main.t"])


def test_gradients_match_reference_golden(golden_step):
    fx, model, _, _, _, _ = golden_step
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(grads) == set(fx.grads), sorted(set(grads) ^ set(fx.grads))
    worst = max((rel(grads[k], g), k) for k, g in fx.grads.items())
    assert worst[0] < 5 * TOL, worst
    gb = model.gradbuf()
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert p.grad.data_ptr() == p._mesm_gview.data_ptr(), n  # lives in the flat buffer


def test_second_step_after_zero_grad_gives_same_gradients(golden_step):
    fx, model, crit, _, _, _ = golden_step
    g1 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    # train.py order: forward, criterion, optimizer.zero_grad() (set_to_none), backward
    from mesm_amd import synthetic
    batch = synthetic.to_device(fx.batch, dev())
    out = model(**batch, dataset_name=fx.cfg["dataset_name"], is_training=True,
                neg_index=fx.neg_index, masked_words=fx.masked_words)
    _, total = crit(out, batch, True)
    model.zero_grad(set_to_none=True)
    total.backward()
    for n, p in model.named_parameters():
        if n in g1:
            assert rel(p.grad, g1[n]) < 1e-5, n
        else:
            assert p.grad is None, n
    # and accumulation without zeroing doubles them
    out = model(**batch, dataset_name=fx.cfg["dataset_name"], is_training=True,
                neg_index=fx.neg_index, masked_words=fx.masked_words)
    _, total = crit(out, batch, True)
    total.backward()
    for n, p in model.named_parameters():
        if n in g1:
            assert rel(p.grad, 2 * g1[n]) < 1e-5, n


# ----------------------------------------------------------------------------- inference call
@pytest.mark.parametrize("case", CASES)
def test_eval_mode_forward_matches_reference_golden(case):
    """eval.py:63,102: model.eval(), torch.no_grad(), model(..., is_training=False) (MLM branch off) and
    criterion(outputs, batch, is_training=False) (rec_fw loss off) against the real reference."""
    from mesm_amd import synthetic
    fx, ev = Fixture(case), EvalFixture(case)
    _, model, crit = build(fx.cfg, fx.sd)
    model.eval()
    crit.eval()
    batch = synthetic.to_device(fx.batch, dev())
    with torch.no_grad():
        out = model(**batch, dataset_name=fx.cfg["dataset_name"], is_training=False, neg_index=ev.neg_index)
        losses, total = crit(out, batch, is_training=False)
    assert sorted(out.keys()) == ev.keys
    for k, v in ev.out.items():
        got = out["aux_outputs"][0][k[5:]] if k.startswith("aux0.") else out[k]
        if v.dtype == torch.bool:
            assert torch.equal(got.cpu(), v), k
        else:
            assert rel(got, v) < TOL, k
    assert set(losses) | {"total"} == set(ev.losses)
    for k, v in ev.losses.items():
        got = float(total) if k == "total" else float(losses[k])
        assert abs(got - v) < TOL * max(1.0, abs(v)), (k, got, v)
    assert not total.requires_grad


# ----------------------------------------------------------------------------- oracle at d=256
@pytest.mark.parametrize("dataset,groups,Lv,Lw,ragged", [
    ("qvhighlights", [2, 1, 3, 2], 75, 32, True),
    ("charades", [2, 2, 1], 75, 16, False),
])
def test_against_cpu_oracle_at_reference_width(dataset, groups, Lv, Lw, ragged):
    from mesm_amd import build_criterion, build_model, synthetic
    from oracle import mesm_oracle as O
    over = dict(dataset_name=dataset, v_feat_dim=130, t_feat_dim=64, vocab_size=301, share_MLP=True,
                set_cost_class=4, loss_label_coef=4, rank_coef=12 if dataset == "qvhighlights" else 1,
                use_triplet=dataset == "qvhighlights", loss_recfw_coef=0.5, loss_recss_coef=0.1,
                max_video_l=Lv, max_words_l=Lw, device="cuda:0")
    args = synthetic.make_args(None, **over)
    torch.manual_seed(5)
    model = build_model(args)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("_token") or "masked_sent_token" in n_:
                p.normal_(0, 0.5)
    crit = build_criterion(args)
    batch = synthetic.make_batch(dataset, groups, Lv, Lw, 130, 64, 302, seed=3, ragged=ragged)
    neg, masked = synthetic.host_draws(batch, seed=3)
    out, losses, total = run_step(model, crit, batch, vars(args), neg, masked)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = dict(vars(args))
    o_out, o_losses, o_total, o_grads, o_idx = O.train_step(sd, cfg, batch, neg, masked)
    for k in ("pred_logits", "pred_spans", "saliency_scores", "neg_saliency_scores", "recfw_words_logit"):
        assert rel(out[k], o_out[k]) < TOL, k
    for k, v in o_losses.items():
        assert abs(float(losses[k]) - float(v)) < TOL * max(1.0, abs(float(v))), k
    assert abs(float(total) - float(o_total)) < TOL * max(1.0, abs(float(o_total)))
    # matched indices
    mq = crit.last_match[0].cpu().tolist()
    want = []
    for q, t in o_idx[0]:
        order = torch.argsort(t)
        want += q[order].tolist()
    assert mq == want
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(grads) == set(o_grads)
    # Gradients at this width cross ~10^7 PReLU/ReLU kinks: a pre-activation that is zero to
    # within fp32 rounding (|z| ~ 1e-7; a handful per 600k-element FFN call, measured) takes the
    # other branch under a different summation order and changes ONE rank-1 contribution of the
    # weight gradients upstream.  That is rounding, not a defect (tools/dbg_model.py shows the
    # flips), so the bound here is 3e-3 in relative L2 and 5e-3 in max norm; the golden fixtures
    # above keep the tight bound.
    def l2(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        # floor: gradients that are analytically zero (softmax-invariant key biases) are noise
        return float((a - b).norm()) / max(float(b.norm()), 1e-3 * b.numel() ** 0.5)
    # A single flipped unit changes ONE output row of a weight gradient (and one bias element):
    # the max-norm bound is applied with the two worst rows set aside, which then have to stay
    # inside a loose bound of their own (tools/dbg_grad.py prints the row profile: one row at
    # ~1e-2, every other row at ~1e-7).
    def rel_rows(a, b, drop=2):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        scale = max(float(b.abs().max()), 1e-3)  # same floor as rel()
        d = (a - b).abs().reshape(a.shape[0], -1).max(1)[0] if a.dim() >= 1 else (a - b).abs().reshape(1)
        d = torch.sort(d.reshape(-1), descending=True)[0]
        rest = float(d[drop]) if d.numel() > drop else 0.0
        return rest / scale, float(d[0]) / scale
    worst = max((rel_rows(grads[k], g)[0], k) for k, g in o_grads.items())
    assert worst[0] < 5e-3, worst
    loose = max((rel_rows(grads[k], g)[1], k) for k, g in o_grads.items())
    assert loose[0] < 5e-2, loose
    worst2 = max((l2(grads[k], g), k) for k, g in o_grads.items())
    assert worst2[0] < 1e-2, worst2


# ----------------------------------------------------------------------------- full benchmark size
@pytest.mark.parametrize("workload", ["C1", "C2", "C3a", "C3b", "C5"])
def test_full_size_workload_against_cpu_oracle(workload):
    """Every BASELINE.json configuration at its full size: C1 Charades VGG+GloVe (N=2,
    Dv=4098, Dt=300), C2 Charades C+SF bs=32, C3a = the bench.py workload (QVHighlights C+SF, 32 pairs,
    Lv=75, Dv=2818, C=5003), C3b = 8 groups x 4 queries, C5 TACoS Lv=512 bs=16 (multi-tile attention
    backward, 4098-wide LayerNorm / GEMM, TwoMLP), with dropout off: losses / logits within 1e-4 of the
    CPU oracle, matched indices bit-exact, and gradient L2 error small on every parameter."""
    from mesm_amd import build_criterion, build_model, synthetic
    from oracle import mesm_oracle as O
    args = synthetic.make_args(workload, device="cuda:0")
    torch.manual_seed(1234)
    model = build_model(args)
    crit = build_criterion(args)
    batch = synthetic.workload_batch(workload, seed=0)
    neg, masked = synthetic.host_draws(batch, seed=0)
    out, losses, total = run_step(model, crit, batch, vars(args), neg, masked)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    o_out, o_losses, o_total, o_grads, o_idx = O.train_step(sd, dict(vars(args)), batch, neg, masked)
    for k in ("pred_logits", "pred_spans", "saliency_scores", "neg_saliency_scores", "recfw_words_logit"):
        assert rel(out[k], o_out[k]) < TOL, k
    for k, v in o_losses.items():
        assert abs(float(losses[k]) - float(v)) < TOL * max(1.0, abs(float(v))), k
    assert abs(float(total) - float(o_total)) < TOL * max(1.0, abs(float(o_total)))
    mq = crit.last_match[0].cpu().tolist()
    want = []
    for q, t in o_idx[0]:
        want += q[torch.argsort(t)].tolist()
    assert mq == want
    worst = 0.0
    for n, p in model.named_parameters():
        if n in o_grads:
            a, b = p.grad.detach().double().cpu(), o_grads[n].double()
            worst = max(worst, float((a - b).norm()) / max(float(b.norm()), 1e-3 * b.numel() ** 0.5))
    assert worst < 1e-2, worst


def test_full_width_gradients_are_tight_without_activation_kinks():
    """Control run for the loose full-width gradient bound above: the same C3a step with
    every PReLU slope = 1 and every ReLU bypassed (mesm_amd.testing.no_relu() / the oracle's NO_RELU)
    has no activation kink to flip, and then EVERY parameter gradient must sit within 5e-4 (max norm) of the
    oracle's.  If this held only with the loose bound there would be a defect in a backward kernel."""
    from mesm_amd import build_criterion, build_model, synthetic, testing
    from oracle import mesm_oracle as O
    args = synthetic.make_args("C3a", device="cuda:0")
    torch.manual_seed(1234)
    model = build_model(args)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("activation.weight"):
                p.fill_(1.0)
    crit = build_criterion(args)
    batch = synthetic.workload_batch("C3a", seed=0)
    neg, masked = synthetic.host_draws(batch, seed=0)
    O.NO_RELU = True
    try:
        with testing.no_relu():
            out, losses, total = run_step(model, crit, batch, vars(args), neg, masked)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o_out, o_losses, o_total, o_grads, o_idx = O.train_step(sd, dict(vars(args)), batch, neg, masked)
    finally:
        O.NO_RELU = False
    assert abs(float(total) - float(o_total)) < TOL * max(1.0, abs(float(o_total)))
    worst = max((rel(p.grad, o_grads[n]), n) for n, p in model.named_parameters() if n in o_grads)
    assert worst[0] < 5e-4, worst


def test_graph_replay_equals_eager_step(deterministic_forward):
    """One captured HIP graph per step (GraphedStep) reproduces the eager step: same loss and same
    flat gradient buffer (dropout off so that both are deterministic functions of the inputs)."""
    from mesm_amd import build_criterion, build_model, synthetic
    from mesm_amd.graphed import GraphedStep
    args = synthetic.make_args("C3a", device="cuda:0")
    torch.manual_seed(7)
    model = build_model(args)
    crit = build_criterion(args)
    args.dropout = 0.0
    batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=1), dev())
    for m in model.modules():
        if hasattr(m, "p"):
            m.p = 0.0
    g = GraphedStep(model, crit, batch, args.dataset_name, warmup=1)
    total_g = float(g.run(redraw=False))
    flat_g = model.gradbuf().flat.clone()
    out = model(**batch, dataset_name=args.dataset_name, is_training=True, plan=g.plan)
    _, total = crit(out, batch, True)
    model.zero_grad(set_to_none=True)
    total.backward()
    torch.cuda.synchronize()
    assert abs(float(total) - total_g) < 1e-5 * max(1.0, abs(total_g))
    flat_e = model.gradbuf().flat
    # (fixture deterministic_forward: the forward's K-split products off; with them on the run-to-run spread of a step is
    # what test_run_to_run_spread_of_a_replayed_step below measures and bounds, nowhere else)
    assert float((flat_e - flat_g).norm()) / max(float(flat_e.norm()), 1e-6) < 1e-4


@pytest.mark.parametrize("fwd_atomics", [False, True])
def test_run_to_run_spread_of_a_replayed_step(fwd_atomics):
    """The only run-to-run freedom of a step on identical inputs, draws and dropout masks is the order of float atomic adds.
    With the forward's K-split products off (MESM_GEMM_FWD_ATOMICS=0) they are all in the backward: the loss is
    bit-reproducible and the gradients agree to rounding.  With them on (default: 1.6 % faster) activations differ in the last
    bit and a ReLU / PReLU kink flips now and then: the loss still agrees to 1e-6, the gradients to a few 1e-5 of their norm."""
    from mesm_amd import build_criterion, build_model, kernels as kn, synthetic
    from mesm_amd.graphed import GraphedStep
    saved = kn._FWD_ATOMICS
    kn._FWD_ATOMICS = fwd_atomics
    try:
        args = synthetic.make_args("C3a", device="cuda:0")
        torch.manual_seed(11)
        model = build_model(args)
        crit = build_criterion(args)
        model.train()
        batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=3), dev())
        g = GraphedStep(model, crit, batch, args.dataset_name)
        gb = model.gradbuf()

        def replay():
            g.counter.fill_(5)  # (the same dropout masks every time)
            t = float(g.run(redraw=False))
            torch.cuda.synchronize()
            return t, gb.flat.clone()

        t0, f0 = replay()
        for _ in range(6):
            t, f = replay()
            rel = float((f - f0).norm()) / float(f0.norm())
            if fwd_atomics:
                assert abs(t - t0) < 1e-6 * max(1.0, abs(t0)) and rel < 1e-3, (t, t0, rel)
            else:
                assert t == t0 and rel < 2e-6, (t, t0, rel)
    finally:
        kn._FWD_ATOMICS = saved


def test_a_second_forward_before_the_first_backward_leaves_the_first_steps_activations_alone():
    """forward activations live in the step's zero pool (K-split products onto pool-zeroed outputs), and the pool
    is recycled when the next forward begins.  A second grad-enabled forward BEFORE the first one's backward (deferred
    backward, teacher / student, two models in one process: the pool is process-global) must not clear or re-issue what the
    first backward still reads: ZeroPool lets go of its buffers instead (kernels.ZeroPool.fwd_live)."""
    from mesm_amd import build_criterion, build_model, kernels as kn, synthetic
    args = synthetic.make_args("C3a", device="cuda:0")
    torch.manual_seed(5)
    model = build_model(args)
    crit = build_criterion(args)
    torch.manual_seed(6)
    other = build_model(args)  # a second model in the same process
    for mm in (model, other):
        for m in mm.modules():
            if hasattr(m, "p"):
                m.p = 0.0
    b1 = synthetic.to_device(synthetic.workload_batch("C3a", seed=1), dev())
    b2 = synthetic.to_device(synthetic.workload_batch("C3a", seed=2), dev())
    d1, d2 = synthetic.host_draws(synthetic.workload_batch("C3a", seed=1), 1), synthetic.host_draws(synthetic.workload_batch("C3a", seed=2), 2)
    name = args.dataset_name

    def fwd(m, b, d):
        out = m(**b, dataset_name=name, is_training=True, neg_index=d[0], masked_words=d[1])
        return crit(out, b, True)[1]

    for _ in range(2):  # (the pool learns its sizes on the first step)
        loss = fwd(model, b1, d1)
        model.zero_grad(set_to_none=True)
        loss.backward()
    torch.cuda.synchronize()
    ref = model.gradbuf().flat.clone()
    assert kn.zero_pool.buf is not None and not kn.zero_pool.fwd_live
    for second in (model, other):
        let_go = kn.zero_pool.let_go
        loss = fwd(model, b1, d1)
        assert kn.zero_pool.fwd_live  # pool-backed forward outputs are waiting for their backward
        loss_b = fwd(second, b2, d2)  # ... and another forward begins
        assert kn.zero_pool.let_go == let_go + 1
        model.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
        got = model.gradbuf().flat.clone()
        assert float((got - ref).norm()) / float(ref.norm()) < 1e-3  # (run-to-run bound of the default setting)
        second.zero_grad(set_to_none=True)
        loss_b.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(second.gradbuf().flat).all()
