"""Generate tests/golden/*.npz by running the REAL reference (/root/reference, read-only,
build container only) on seeded synthetic batches.

The reference is imported, never copied: only inputs, weights and outputs (data) are stored.
    python tools/gen_golden.py            # regenerates every fixture

For each case: build MESM + Criterion through the reference's own runner.build_model /
build_criterion (runner.py:255-345), put the model in eval() (dropout off) but call it with
is_training=True (MLM branch on — the two switches are independent, model.py:307), record
the two host-RNG draws (negative index, masked words) by wrapping the reference functions,
run criterion + backward, and dump everything needed to replay the step elsewhere.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
for name in ("ftfy", "nltk", "h5py"):
    sys.modules.setdefault(name, types.ModuleType(name))

import runner  # noqa: E402  (the reference's factory module)
import model.model as ref_model_mod  # noqa: E402

from mesm_amd import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

TINY = dict(hidden_dim=32, nheads=4, dim_feedforward=64, num_queries=5, max_video_l=20, max_words_l=8)

CASES = {
    "charades_tiny": dict(dataset_name="charades", groups=[2, 1, 2], Lv=20, Lw=8, v_feat_dim=18,
                          t_feat_dim=12, vocab_size=29, share_MLP=True, set_cost_class=4,
                          loss_label_coef=4, rank_coef=1, use_triplet=False, loss_recfw_coef=0.1,
                          loss_recss_coef=0.1, ragged=True, seed=11),
    "qvh_tiny": dict(dataset_name="qvhighlights", groups=[3, 1, 2], Lv=20, Lw=8, v_feat_dim=18,
                     t_feat_dim=12, vocab_size=29, share_MLP=True, set_cost_class=4,
                     loss_label_coef=4, rank_coef=12, use_triplet=True, loss_recfw_coef=0.5,
                     loss_recss_coef=0.1, ragged=True, seed=12),
    "tacos_tiny": dict(dataset_name="tacos", groups=[3, 2], Lv=70, Lw=8, v_feat_dim=18,
                       t_feat_dim=12, vocab_size=29, share_MLP=False, set_cost_class=6,
                       loss_label_coef=6, rank_coef=1, use_triplet=True, loss_recfw_coef=0.1,
                       loss_recss_coef=0.1, ragged=False, seed=13, max_video_l=70),
}


def flatten_targets(batch):
    """npz-friendly copy of the batch (lists of dicts -> concatenated arrays + sizes)."""
    flat = {}
    for k, v in batch.items():
        if v is None:
            continue
        if isinstance(v, torch.Tensor):
            flat["batch." + k] = v.numpy()
        elif isinstance(v, list):
            key = list(v[0].keys())[0]
            flat["batch.%s.sizes" % k] = np.array([len(d[key]) for d in v])
            flat["batch.%s.cat" % k] = torch.cat([d[key] for d in v]).numpy()
    return flat


def run_case(name, spec, tiny=None, out_dir=None):
    spec = dict(spec)
    ragged, seed = spec.pop("ragged"), spec.pop("seed")
    groups, Lv, Lw = spec.pop("groups"), spec.pop("Lv"), spec.pop("Lw")
    over = dict(TINY if tiny is None else tiny)
    over.update(spec)
    args = synthetic.make_args(None, **over)
    torch.manual_seed(seed)
    np.random.seed(seed)
    net = runner.build_model(args)
    crit = runner.build_criterion(args)
    # perturb the zero-initialised tokens / PReLU slopes so every path carries signal
    with torch.no_grad():
        for n_, p in net.named_parameters():
            if n_.endswith("masked_token") or n_.endswith("unknown_token") or n_.endswith("masked_sent_token"):
                p.normal_(0, 0.5)
            if n_.endswith("activation.weight"):
                p.uniform_(0.1, 0.4)
            if "LayerNorm" in n_ or ".norm" in n_:
                p.add_(torch.randn_like(p) * 0.1)
    net.eval()
    batch = synthetic.make_batch(args.dataset_name, groups, Lv, Lw, args.v_feat_dim, args.t_feat_dim,
                                 args.vocab_size + 1, seed=seed, ragged=ragged)

    rec = {}
    orig_neg = ref_model_mod.sample_outclass_neg

    def neg_wrap(num_clips):
        r = orig_neg(num_clips)
        rec["neg_index"] = r.clone()
        return r

    ref_model_mod.sample_outclass_neg = neg_wrap
    orig_mask = net._mask_words

    def mask_wrap(*a, **kw):
        out = orig_mask(*a, **kw)
        rec["masked_words"] = out[1].clone()
        return out

    net._mask_words = mask_wrap
    try:
        outputs = net(**batch, dataset_name=args.dataset_name, is_training=True)
        losses, total = crit(outputs, batch, True)
        net.zero_grad()
        total.backward()
        # matcher indices of the final layer and of the aux layer, for the record
        with torch.no_grad():
            idx_main = crit.matcher({k: v for k, v in outputs.items() if k != "aux_outputs"}, batch)
            idx_aux = [crit.matcher(a, batch) for a in outputs.get("aux_outputs", [])]
    finally:
        ref_model_mod.sample_outclass_neg = orig_neg

    def idx_to_arrays(idx):
        if isinstance(idx, list):  # qvh: list of (query_idx, target_idx)
            q = torch.cat([a for a, _ in idx])
            t = torch.cat([b for _, b in idx])
            sizes = torch.tensor([len(a) for a, _ in idx])
            return q.numpy(), t.numpy(), sizes.numpy()
        q = idx[:, 0]
        return q.numpy(), idx[:, 1].numpy(), np.ones(len(q), dtype=np.int64)

    blob = {}
    for k, v in net.state_dict().items():
        blob["sd." + k] = v.detach().numpy()
    for k, p in net.named_parameters():
        if p.grad is not None:
            blob["grad." + k] = p.grad.detach().numpy()
    blob.update(flatten_targets(batch))
    blob["neg_index"] = rec["neg_index"].numpy()
    if "masked_words" in rec:  # drawn by the MLM branch only (rec_fw)
        blob["masked_words"] = rec["masked_words"].numpy()
    for k, v in outputs.items():
        if isinstance(v, torch.Tensor):
            blob["out." + k] = v.detach().numpy()
    for i, a in enumerate(outputs.get("aux_outputs", [])):
        for k, v in a.items():
            blob["out.aux%d.%s" % (i, k)] = v.detach().numpy()
    for k, v in losses.items():
        blob["loss." + k] = np.asarray(v.detach().numpy())
    blob["loss.total"] = np.asarray(total.detach().numpy())
    q, t, s = idx_to_arrays(idx_main)
    blob["match.main.q"], blob["match.main.t"], blob["match.main.sizes"] = q, t, s
    for i, ia in enumerate(idx_aux):
        q, t, s = idx_to_arrays(ia)
        blob["match.aux%d.q" % i], blob["match.aux%d.t" % i], blob["match.aux%d.sizes" % i] = q, t, s
    cfg = dict(vars(args))
    cfg.update(groups=groups, Lv=Lv, Lw=Lw)
    blob["cfg.json"] = np.frombuffer(__import__("json").dumps(cfg).encode(), dtype=np.uint8)
    out_dir = out_dir or OUT
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **blob)
    print("%s: total loss %.6f, %d tensors, %.1f KB" % (name, float(total), len(blob),
                                                       os.path.getsize(path) / 1024))


def span_doctest_vectors():
    """Known answers the reference ships as doctests (utils/span_utils.py:12-19, 31-38, 54-60,
    105-109), evaluated by the reference itself."""
    import utils.span_utils as su
    xx = torch.tensor([[0, 1], [0.2, 0.4]])
    a = torch.tensor([[0, 0.2], [0.5, 1.0]])
    b = torch.tensor([[0, 0.3], [0., 1.0]])
    iou, union = su.temporal_iou(a, b)
    np.savez(os.path.join(OUT, "span_doctests.npz"), xx=xx.numpy(), cxw=su.span_xx_to_cxw(xx).numpy(),
             back=su.span_cxw_to_xx(su.span_xx_to_cxw(xx)).numpy(), a=a.numpy(), b=b.numpy(),
             iou=iou.numpy(), union=union.numpy(), giou=su.generalized_temporal_iou(a, b).numpy())


if __name__ == "__main__":
    torch.set_num_threads(4)
    for name, spec in CASES.items():
        run_case(name, spec)
    span_doctest_vectors()
