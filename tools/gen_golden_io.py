"""Fixtures for the rows either side of the hot path, produced by the REAL reference
(/root/reference, build container only; imported, never copied; only data is stored):

  collate.npz     dataset/base.py:288-355 `collate` (Charades / TACoS form) and dataset/qvhighlights.py:214-284
                  `collate` (QVHighlights form) + `prepare_batch_input` (base.py:358-384) on seeded per-group
                  samples of the shape Dataset.__getitem__ returns (base.py:164-223)
  mr_results.json eval.py:52-117 `compute_mr_results` driven with a stub model that returns seeded outputs, a list
                  loader and a stub criterion, then `post_processing_mr_nms` (eval.py:476-485): the submission
                  rows incl. `pred_saliency_scores` (.half() then per-length tolist) and the NMS'd windows
  resume_tiny.ckpt / resume_tiny.npz
                  a checkpoint in train.py:185-192's format written after two optimizer steps of the reference
                  loop (train.py:64-72: zero_grad, backward, clip_grad_norm_, AdamW.step; StepLR) and the
                  parameters after the THIRD step, for the resume test

    python tools/gen_golden_io.py
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)
for name in ("ftfy", "nltk", "h5py"):
    sys.modules.setdefault(name, types.ModuleType(name))

import tqdm as _tqdm  # noqa: E402

_real_tqdm = _tqdm.tqdm
_tqdm.tqdm = lambda x, **k: x

import runner  # noqa: E402
import dataset.base as ds_base  # noqa: E402
import dataset.qvhighlights as ds_qvh  # noqa: E402
import eval as ref_eval  # noqa: E402
import model.model as ref_model_mod  # noqa: E402
import utils.post_processing as ref_pp  # noqa: E402

ref_eval.tqdm = lambda x, **k: x
ref_pp.tqdm = lambda x, **k: x

from mesm_amd import synthetic  # noqa: E402
from io_cases import StubCriterion, StubModel, group_samples, mr_inputs  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


# ----------------------------------------------------------------------------- collate
def flatten(prefix, d, blob, meta):
    for k, v in d.items():
        if torch.is_tensor(v):
            blob[prefix + k] = v.numpy()
        elif isinstance(v, list) and v and isinstance(v[0], dict):
            key = list(v[0].keys())[0]
            blob[prefix + k + ".sizes"] = np.array([len(x[key]) for x in v])
            blob[prefix + k + ".cat"] = torch.cat([x[key] for x in v]).numpy()
            meta[prefix + k] = key
        elif v is None:
            meta[prefix + k] = None
        else:
            meta[prefix + k] = v


def collate_cases():
    blob, meta = {}, {}
    for kind, fn, seed in (("base", ds_base.collate, 41), ("qvh", ds_qvh.collate, 42)):
        samples = group_samples(kind, seed)
        out = fn(group_samples(kind, seed))
        flatten(kind + ".out.", out, blob, meta)
        prepared = dict(out)
        r = ds_base.prepare_batch_input(prepared, torch.device("cpu"))  # mutates its argument
        if r is not None:
            prepared = r
        flatten(kind + ".prep.", prepared, blob, meta)
        meta[kind + ".seed"] = seed
    blob["meta.json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "collate.npz"), **blob)
    print("collate:", sorted(k for k in blob if k.startswith("qvh.prep"))[:6], "...")


# ----------------------------------------------------------------------------- eval rows + NMS
def mr_cases():
    from io_cases import MR_CASES
    import copy
    res = {}
    for name, c in MR_CASES.items():
        loader, outs = mr_inputs(c)
        opt = argparse.Namespace(device="cpu", pin_memory=False, dataset_name=name, span_loss_type="l1",
                                 sort_results=True, clip_len=c["clip_len"], max_ts_val=150, max_video_l=c["Lv"])
        mr_res, meters = ref_eval.compute_mr_results(StubModel(outs), loader, opt, criterion=StubCriterion())
        rows = copy.deepcopy(mr_res)
        nms = ref_eval.post_processing_mr_nms(copy.deepcopy(mr_res), nms_thd=c["nms_thd"], max_before_nms=10,
                                              max_after_nms=5)
        res[name] = {"mr_res": rows, "after_nms": nms, "loss_meters": {k: v.avg for k, v in meters.items()}}
    with open(os.path.join(OUT, "mr_results.json"), "w") as f:
        json.dump(res, f)
    print("mr_results:", {k: len(v["mr_res"]) for k, v in res.items()},
          os.path.getsize(os.path.join(OUT, "mr_results.json")) // 1024, "KB")


# ----------------------------------------------------------------------------- checkpoint / resume
def resume_case():
    import utils.model_utils as mu
    cfg = dict(hidden_dim=16, nheads=2, dim_feedforward=32, num_queries=5, max_video_l=12, max_words_l=6,
               dataset_name="qvhighlights", v_feat_dim=10, t_feat_dim=8, vocab_size=19, share_MLP=True,
               set_cost_class=4, loss_label_coef=4, rank_coef=12, use_triplet=True, loss_recfw_coef=0.5,
               loss_recss_coef=0.1, num_recss_layers=2)
    args = synthetic.make_args(None, **cfg)
    args.lr, args.weight_decay, args.lr_drop, args.gamma, args.grad_clip = 1e-3, 1e-2, 2, 0.1, 0.1
    torch.manual_seed(61)
    np.random.seed(61)
    net = runner.build_model(args)
    crit = runner.build_criterion(args)
    optimizer, sched = runner.build_optimizer(args, net)
    with torch.no_grad():
        for n_, p in net.named_parameters():
            if n_.endswith("_token"):
                p.normal_(0, 0.5)
    net.eval()  # dropout off (the optimizer arithmetic is what is pinned); MLM branch on via is_training=True
    groups, Lv, Lw = [2, 1, 2], 12, 6
    batch = synthetic.make_batch("qvhighlights", groups, Lv, Lw, 10, 8, 20, seed=61, ragged=False)
    draws = []  # what the reference's own sample_outclass_neg / _mask_words drew at every step
    orig_neg = ref_model_mod.sample_outclass_neg

    def neg_wrap(num_clips):
        r = orig_neg(num_clips)
        draws.append([r.clone(), None])
        return r

    ref_model_mod.sample_outclass_neg = neg_wrap
    orig_mask = net._mask_words

    def mask_wrap(*a, **kw):
        out = orig_mask(*a, **kw)
        draws[-1][1] = out[1].clone()
        return out

    net._mask_words = mask_wrap
    state = {"step": 0}

    def train_step():
        out = net(**batch, dataset_name=args.dataset_name, is_training=True)
        _, loss = crit(out, batch, True)
        optimizer.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), args.grad_clip)
        optimizer.step()
        state["step"] += 1
        return float(loss)

    try:
        losses = [train_step(), train_step()]
        sched.step()  # epoch boundary (train.py:130): lr_drop = 2 -> the drop happens one epoch later
        sched.step()
        ckpt = {"model": mu.state_dict_without_module(net, "text_encoder"), "optimizer": optimizer.state_dict(),
                "lr_scheduler": sched.state_dict(), "epoch": 1, "opt": args}
        torch.save(ckpt, os.path.join(OUT, "resume_tiny.ckpt"))
        losses.append(train_step())
    finally:
        ref_model_mod.sample_outclass_neg = orig_neg
    blob = {"after." + k: v.detach().numpy() for k, v in net.state_dict().items()}
    blob["losses"] = np.array(losses)
    blob["lr_after"] = np.array(optimizer.param_groups[0]["lr"])
    for s, (neg, mw) in enumerate(draws):
        blob["neg%d" % s], blob["mw%d" % s] = neg.numpy(), mw.numpy()
    cfgj = dict(vars(args))
    cfgj.update(groups=groups, Lv=Lv, Lw=Lw, batch_seed=61)
    blob["cfg.json"] = np.frombuffer(json.dumps(cfgj).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "resume_tiny.npz"), **blob)
    print("resume_tiny: losses", losses, "lr", optimizer.param_groups[0]["lr"],
          os.path.getsize(os.path.join(OUT, "resume_tiny.ckpt")) // 1024, "KB")


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["collate", "mr", "resume"]
    if "collate" in which:
        collate_cases()
    if "mr" in which:
        mr_cases()
    if "resume" in which:
        resume_case()
