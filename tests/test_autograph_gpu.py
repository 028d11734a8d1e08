"""Graph replay behind the UNCHANGED caller (mesm_amd/autograph.py): the reference's own loop body, train.py:64-72,

    outputs = model(**batch, dataset_name=opt.dataset_name, is_training=True)
    loss_dict, loss = criterion(outputs, batch, is_training=True)
    optimizer.zero_grad()
    loss.backward()
    nn.utils.clip_grad_norm_(model.parameters(), opt.grad_clip)
    optimizer.step()

run literally, with torch's own AdamW and clip_grad_norm_: first visit of a shape eager, every later one three graph
replays; loss and gradients of a replayed step equal the eager step on the same batch and draws (1e-4 of the gradient norm with the
forward deterministic, the bound of the one-graph replay test), parameters really update, gradients accumulate when zero_grad is skipped,
a second forward in flight is refused, other shapes get their own graphs, padded buckets serve changing pair counts."""
import os

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _no_dropout(model):
    for m in model.modules():
        if hasattr(m, "p"):
            m.p = 0.0


def _build(workload, seed=7, dropout=False, **over):
    from mesm_amd import build_criterion, build_model, synthetic
    args = synthetic.make_args(workload, device="cuda:0", **over)
    torch.manual_seed(seed)
    model = build_model(args)
    crit = build_criterion(args)
    if not dropout:
        _no_dropout(model)
    model.train()
    crit.train()
    model.autograph(True)
    return args, model, crit


def _reference_loop_body(model, criterion, optimizer, batch, opt, before_step=None):
    """train.py:64-72, verbatim but for the names"""
    outputs = model(**batch, dataset_name=opt.dataset_name, is_training=True)
    loss_dict, loss = criterion(outputs, batch, is_training=True)
    optimizer.zero_grad()
    loss.backward()
    if before_step is not None:
        before_step(outputs, loss_dict, loss)
    if opt.grad_clip > 0:
        nn.utils.clip_grad_norm_(model.parameters(), opt.grad_clip)
    optimizer.step()
    return outputs, loss_dict, loss


def _eager_on(model, crit, batch, name, neg, mw):
    """the eager step on `batch` with the given host draws (explicit draws keep the call out of autograph)"""
    kw = dict(neg_index=torch.as_tensor(neg))
    if mw is not None:
        kw["masked_words"] = torch.as_tensor(mw)
    out = model(**batch, dataset_name=name, is_training=True, **kw)
    losses, total = crit(out, batch, True)
    model.zero_grad(set_to_none=True)
    total.backward()
    torch.cuda.synchronize()
    return out, losses, float(total.detach()), model.gradbuf().flat.clone()


@pytest.mark.parametrize("workload,factory", [("C3a", "torch"), ("C2", "torch"), ("C3a", "this build")])
def test_reference_loop_body_runs_on_graph_replays_and_equals_eager(workload, factory, deterministic_forward):
    from mesm_amd import build_optimizer, synthetic
    from mesm_amd.autograph import AutoOutputs
    args, model, crit = _build(workload)
    opt = argparse_like(args, grad_clip=0.1, lr=1e-4, weight_decay=1e-4, lr_drop=400, gamma=0.1)
    if factory == "torch":   # the reference's own class (runner.py:348-352)
        optimizer = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
    else:                    # what train.py:115 gets when all three factories of its lines 17-18 come from this build
        optimizer, _ = build_optimizer(opt, model)
    auto = model._auto
    name = args.dataset_name
    seen = {}

    def check(batch):
        def hook(outputs, loss_dict, loss):
            seen["replayed"] = outputs._auto_step is not None
            if not seen["replayed"]:
                return
            torch.cuda.synchronize()
            step = outputs._auto_step
            flat_g = model.gradbuf().flat.clone()
            had = [p.grad is not None for p in model.gradbuf().params]
            out_g = {k: v.clone() for k, v in outputs.items() if torch.is_tensor(v)}
            ld_g = {k: float(v) for k, v in loss_dict.items()}
            total_g = float(loss.detach())
            neg, mw = step._draws
            out_e, ld_e, total_e, flat_e = _eager_on(model, crit, batch, name, neg, mw)
            assert abs(total_e - total_g) < 1e-5 * max(1.0, abs(total_e)), (total_e, total_g)
            assert set(ld_e) == set(ld_g)
            for k in ld_e:
                assert abs(float(ld_e[k]) - ld_g[k]) < 1e-5 * max(1.0, abs(ld_g[k])), k
            assert set(out_e) == set(outputs)
            for k, v in out_g.items():
                d = (out_e[k].double() - v.double()).abs().max().item() if v.dtype.is_floating_point else float((out_e[k] != v).sum())
                assert d <= 1e-4 * max(1.0, float(out_e[k].double().abs().max()) if v.dtype.is_floating_point else 1.0), (k, d)
            assert float((flat_e - flat_g).norm()) / max(float(flat_e.norm()), 1e-6) < 1e-4
            assert had == [p.grad is not None for p in model.gradbuf().params]
            model.gradbuf().flat.copy_(flat_g)  # the step goes on with the replayed gradients
        return hook

    p0 = model.flat_params().clone()
    for i in range(4):
        batch = synthetic.to_device(synthetic.workload_batch(workload, seed=10 + i), dev())
        outputs, loss_dict, loss = _reference_loop_body(model, crit, optimizer, batch, opt, before_step=check(batch))
        assert isinstance(outputs, AutoOutputs) and isinstance(outputs, dict)
        assert seen["replayed"] == (i >= 1), i
        float(loss)  # train.py:75
        assert all(torch.isfinite(v).all() for v in loss_dict.values())
        # train.py:76-77 reads every entry with float(): they share ONE host fetch (criterion.LossEntry) and equal the device values
        for k, v in loss_dict.items():
            assert float(v) == v.item() == float(v.clone()), k
            assert type(v * 2.0) is torch.Tensor
    assert (auto.eager, auto.captures, auto.replays) == (1, 1, 3)
    assert float((model.flat_params() - p0).abs().max()) > 0  # the optimizer's updates went through the views


def argparse_like(args, **kw):
    import argparse
    d = dict(vars(args))
    d.update(kw)
    return argparse.Namespace(**d)


def test_gradients_accumulate_when_zero_grad_is_skipped_and_stale_steps_are_refused(deterministic_forward):
    from mesm_amd import synthetic
    args, model, crit = _build("C3a")
    name = args.dataset_name
    b = [synthetic.to_device(synthetic.workload_batch("C3a", seed=30 + i), dev()) for i in range(3)]

    def step(batch, zero):
        out = model(**batch, dataset_name=name, is_training=True)
        _, loss = crit(out, batch, is_training=True)
        if zero:
            model.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
        return out, loss

    step(b[0], True)                       # eager visit
    out1, _ = step(b[1], True)             # capture + replay
    assert out1._auto_step is not None
    g1 = model.gradbuf().flat.clone()
    out2, _ = step(b[2], False)            # accumulates on top of g1
    assert out2._auto_step is not None
    g12 = model.gradbuf().flat.clone()
    model.zero_grad(set_to_none=True)
    neg, mw = out2._auto_step._draws
    _, _, _, g2 = _eager_on(model, crit, b[2], name, neg, mw)
    assert float((g12 - g1 - g2).norm()) / float(g2.norm()) < 2e-4
    # two forwards, then the first one's criterion / backward: refused, not silently wrong
    o_a = model(**b[1], dataset_name=name, is_training=True)
    _, loss_a = crit(o_a, b[1], is_training=True)
    o_b = model(**b[2], dataset_name=name, is_training=True)
    with pytest.raises(RuntimeError, match="ONE forward"):
        loss_a.backward()
    with pytest.raises(RuntimeError, match="ONE forward"):
        crit(o_a, b[1], is_training=True)
    _, loss_b = crit(o_b, b[2], is_training=True)
    loss_b.backward()  # the latest one is fine
    # targets that are not the forward's batch
    o_c = model(**b[1], dataset_name=name, is_training=True)
    with pytest.raises(RuntimeError, match="targets"):
        crit(o_c, b[2], is_training=True)


def test_no_grad_eval_and_switched_off_calls_stay_eager():
    from mesm_amd import synthetic
    args, model, crit = _build("C3a")
    name = args.dataset_name
    batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=40), dev())
    auto = model._auto
    for _ in range(2):
        with torch.no_grad():
            out = model(**batch, dataset_name=name, is_training=False)
            crit(out, batch, is_training=False)
    assert (auto.captures, auto.replays) == (0, 0)
    model.autograph(False)
    for _ in range(3):
        out = model(**batch, dataset_name=name, is_training=True)
        _, loss = crit(out, batch, is_training=True)
        model.zero_grad(set_to_none=True)
        loss.backward()
        assert out._auto_step is None
    assert (auto.captures, auto.replays) == (0, 0)


def test_other_shapes_get_their_own_graphs_and_dropout_draws_fresh_masks():
    from mesm_amd import synthetic
    args, model, crit = _build("C3b", dropout=True)
    name = args.dataset_name
    auto = model._auto
    ba = synthetic.to_device(synthetic.workload_batch("C3b", seed=50), dev())
    bb = synthetic.to_device(synthetic.workload_batch("C3a", seed=51), dev())  # same model dims, other grouping / pairs
    losses = []
    for batch in (ba, bb, ba, bb, ba, ba):
        out = model(**batch, dataset_name=name, is_training=True)
        _, loss = crit(out, batch, is_training=True)
        model.zero_grad(set_to_none=True)
        loss.backward()
        losses.append((out._auto_step is not None, float(loss)))
    assert [r for r, _ in losses] == [False, False, True, True, True, True]
    assert auto.captures == 2
    assert losses[4][1] != losses[5][1]  # same batch, new dropout masks and draws per replay


def test_padded_buckets_serve_changing_pair_counts(deterministic_forward):
    """model.autograph(pad=..., pairs=8): batches of 41 / 43 / 46 pairs replay the 48-pair graph; outputs come back with
    the caller's pair count and the step equals the eager step on the unpadded batch"""
    from mesm_amd import synthetic
    args, model, crit = _build("C3a")
    name = args.dataset_name
    model.autograph(True, pad=(75, 32), pairs=8)
    auto = model._auto

    def mk(groups, seed):
        w = synthetic.WORKLOADS["C3a"]
        return synthetic.to_device(synthetic.make_batch(w["dataset_name"], groups, w["Lv"], w["Lw"], w["v_feat_dim"], w["t_feat_dim"],
                                                        w["vocab_size"] + 1, seed=seed, ragged=True), dev())
    shapes = [[3, 4, 2, 5, 1, 3, 4, 2, 5, 1, 4, 3, 4], [4, 4, 2, 5, 1, 3, 4, 2, 5, 3, 4, 3, 3], [5, 4, 2, 5, 1, 3, 4, 2, 5, 3, 4, 3, 5]]
    assert [sum(s) for s in shapes] == [41, 43, 46]
    for i, groups in enumerate(shapes):
        batch = mk(groups, 60 + i)
        out = model(**batch, dataset_name=name, is_training=True)
        _, loss = crit(out, batch, is_training=True)
        model.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
        assert out["pred_spans"].shape[0] == sum(groups)
        if i == 0:
            assert out._auto_step is None
            continue
        assert out._auto_step is not None
        flat_g = model.gradbuf().flat.clone()
        neg, mw = out._auto_step._draws
        n = sum(groups)
        _, _, total_e, flat_e = _eager_on(model, crit, batch, name, neg[:n], None if mw is None else mw[:n])
        assert abs(total_e - float(loss)) < 1e-5 * max(1.0, abs(total_e))
        assert float((flat_e - flat_g).norm()) / max(float(flat_e.norm()), 1e-6) < 1e-4
    assert auto.captures == 1 and auto.replays == 2


@pytest.mark.parametrize("workload", ["C3a", "C2"])
def test_host_side_kept_by_the_collate_serves_the_plans_without_a_fetch(workload, deterministic_forward):
    """batching.attach_host_side at the end of the collate: `batch["_host"]` survives prepare_batch_input and the **batch call,
    the replayed forward builds its plans from it (no transfer back, no synchronisation) and equals the eager step"""
    from mesm_amd import batching, synthetic
    args, model, crit = _build(workload)
    name = args.dataset_name
    auto = model._auto
    for i in range(3):
        cpu = batching.attach_host_side(synthetic.workload_batch(workload, seed=70 + i, ragged=True))
        batch = synthetic.to_device(cpu, dev())
        assert isinstance(batch["_host"], dict) and not batch["_host"]["video_mask"].is_cuda
        if "norm_span" not in batch:  # prepare_batch_input's derived targets (dataset/base.py:380-384)
            batch["norm_moment"] = batch["moment"] / batch["duration"].unsqueeze(1)
            batch["norm_span"] = batching.span_xx_to_cxw(batch["norm_moment"])
        out = model(**batch, dataset_name=name, is_training=True)
        _, loss = crit(out, batch, is_training=True)
        model.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
        if i == 0:
            continue
        assert out._auto_step is not None
        flat_g = model.gradbuf().flat.clone()
        neg, mw = out._auto_step._draws
        plain = {k: v for k, v in batch.items() if k != "_host"}
        _, _, total_e, flat_e = _eager_on(model, crit, plain, name, neg, mw)
        assert abs(total_e - float(loss)) < 1e-5 * max(1.0, abs(total_e))
        assert float((flat_e - flat_g).norm()) / max(float(flat_e.norm()), 1e-6) < 1e-4
    assert auto.replays == 2 and auto.host_side == 2
    # a `_host` that does not mirror the batch is not trusted: the fetch path serves the step
    cpu = batching.attach_host_side(synthetic.workload_batch(workload, seed=80, ragged=True))
    batch = synthetic.to_device(cpu, dev())
    del batch["_host"]["video_mask"]
    if "norm_span" not in batch:
        batch["norm_moment"] = batch["moment"] / batch["duration"].unsqueeze(1)
        batch["norm_span"] = batching.span_xx_to_cxw(batch["norm_moment"])
    out = model(**batch, dataset_name=name, is_training=True)
    assert out._auto_step is not None and auto.host_side == 2


def test_prepare_batch_input_copies_page_locked_batches_directly_and_equals_the_packed_path():
    """batching.prepare_batch_input: a DataLoader(pin_memory=True) batch goes to the device by asynchronous per-tensor copies (the
    reference's own way, dataset/base.py:358-384), a pageable one through one packed transfer: same tensors either way, words_weight
    stays on the host, `_host` untouched"""
    from mesm_amd import batching, synthetic
    cpu = batching.attach_host_side(synthetic.workload_batch("C3b", seed=5, ragged=True))
    pin = lambda v: v.pin_memory() if torch.is_tensor(v) else ([{kk: vv.pin_memory() for kk, vv in d.items()} for d in v]
                                                               if isinstance(v, list) and v and isinstance(v[0], dict) else v)
    pinned = {k: (pin(v) if k != "_host" else v) for k, v in cpu.items()}
    a = batching.prepare_batch_input(dict(cpu), dev())
    b = batching.prepare_batch_input(dict(pinned), dev(), non_blocking=True)
    torch.cuda.synchronize()
    assert set(a) == set(b)
    for k in a:
        if k == "_host":
            assert a[k] is cpu["_host"] and b[k] is cpu["_host"]
        elif torch.is_tensor(a[k]):
            assert a[k].device == b[k].device and torch.equal(a[k], b[k]), k
            assert a[k].is_cuda == (k != "words_weight"), k
        elif isinstance(a[k], list) and a[k] and isinstance(a[k][0], dict):
            assert all(torch.equal(x[kk], y[kk]) and x[kk].is_cuda for x, y in zip(a[k], b[k]) for kk in x), k
