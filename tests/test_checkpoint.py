"""Checkpoint interop against tests/golden/resume_tiny.ckpt, a file the REAL reference loop wrote
(tools/gen_golden_io.py: train.py:185-192's dict after two optimizer steps) and resume_tiny.npz (the reference's
parameters after its THIRD step)."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from golden_io import GOLDEN

CKPT = os.path.join(GOLDEN, "resume_tiny.ckpt")
Z = np.load(os.path.join(GOLDEN, "resume_tiny.npz"))
CFG = json.loads(bytes(Z["cfg.json"]).decode())


def _args(device):
    a = argparse.Namespace(**{k: v for k, v in CFG.items() if k not in ("groups", "Lv", "Lw", "batch_seed")})
    a.device = device
    return a


def test_reference_checkpoint_loads_strictly_on_the_host():
    from mesm_amd import build_model
    from mesm_amd.checkpoint import load_checkpoint, state_dict_without_module
    ck = torch.load(CKPT, map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "epoch", "opt"}
    model = build_model(_args("cpu"))
    assert load_checkpoint(CKPT, model) is None
    sd = model.state_dict()
    assert set(sd) == set(ck["model"])
    for k, v in ck["model"].items():
        assert torch.equal(sd[k], v), k
    assert list(state_dict_without_module(model, "text_encoder")) == list(ck["model"])
    # the optimizer state is indexed by the trainable parameters in named_parameters() order, like torch's
    n_train = sum(1 for p in model.parameters() if p.requires_grad)
    assert ck["optimizer"]["param_groups"][0]["params"] == list(range(n_train))


@pytest.mark.gpu
def test_resume_all_continues_like_the_reference():
    """load model + AdamW + StepLR state, run the third step on the HIP path with the fused clip + AdamW update:
    loss and every parameter equal the reference's (train.py:64-72, 117-125)."""
    from mesm_amd import build_criterion, build_model, build_optimizer, synthetic
    from mesm_amd.checkpoint import load_checkpoint, save_checkpoint
    args = _args("cuda:0")
    model = build_model(args)
    crit = build_criterion(args)
    opt, sched = build_optimizer(args, model)
    start = load_checkpoint(CKPT, model, opt, sched, resume_all=True)
    assert start == 2
    assert abs(opt.param_groups[0]["lr"] - float(Z["lr_after"])) < 1e-12  # StepLR state: the drop has happened
    model.eval()
    batch = synthetic.make_batch("qvhighlights", CFG["groups"], CFG["Lv"], CFG["Lw"], CFG["v_feat_dim"],
                                 CFG["t_feat_dim"], CFG["vocab_size"] + 1, seed=CFG["batch_seed"])
    batch = synthetic.to_device(batch, torch.device("cuda:0"))
    out = model(**batch, dataset_name="qvhighlights", is_training=True, neg_index=torch.from_numpy(Z["neg2"]),
                masked_words=torch.from_numpy(Z["mw2"]).bool())
    _, total = crit(out, batch, True)
    opt.zero_grad()
    total.backward()
    opt.step(grad_clip=args.grad_clip)
    torch.cuda.synchronize()
    assert abs(float(total) - float(Z["losses"][2])) < 1e-4 * abs(float(Z["losses"][2]))
    sd = model.state_dict()
    # AdamW moves every element by ~lr in the direction of g / sqrt(v): the key biases (softmax-invariant, their
    # gradient is rounding noise) may end anywhere within one step of lr = 1e-4; everything else within 2e-4
    for k in sd:
        a, b = sd[k].double().cpu(), torch.from_numpy(Z["after." + k]).double()
        err = float((a - b).abs().max())
        key_bias = k.endswith("in_proj_bias") or k.endswith(("kcontent_proj.bias", "kpos_proj.bias"))
        assert err < (0.5 * 1e-4 if key_bias else 2e-4 * max(float(b.abs().max()), 1e-3)), (k, err)
    # and what this build saves has the reference's layout
    import tempfile
    path = os.path.join(tempfile.mkdtemp(), "x.ckpt")
    ck = save_checkpoint(path, model, opt, sched, 2, args)
    ref = torch.load(CKPT, map_location="cpu", weights_only=False)
    assert list(ck["model"]) == list(ref["model"])
    back = torch.load(path, map_location="cpu", weights_only=False)
    assert set(back["optimizer"]["state"][0]) == set(ref["optimizer"]["state"][0])
    assert float(back["optimizer"]["state"][0]["step"]) == 3.0
