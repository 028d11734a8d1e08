"""Second batch of fixtures from the REAL reference (/root/reference, build container only):

  <case>_eval.npz   the inference call of eval.py:63,102 on the same weights / batch as <case>.npz:
                    model.eval(), torch.no_grad(), model(..., is_training=False) and
                    criterion(outputs, batch, is_training=False)  (MLM branch off, rec_fw loss off)
  draws.npz         the two host-RNG draws of the forward, taken from the reference's own functions
                    right after re-seeding torch / numpy: sample_outclass_neg
                    (utils/data_utils.py:113-124) and MESM._mask_words (model/model.py:361-384)

Only data is stored (inputs, seeds, outputs); the reference is imported, never copied.
    python tools/gen_golden_r2.py
"""
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
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, REF)
for name in ("ftfy", "nltk", "h5py"):
    sys.modules.setdefault(name, types.ModuleType(name))

import argparse  # noqa: E402

import runner  # noqa: E402  (the reference's factory module)
import model.model as ref_model_mod  # noqa: E402
from utils.data_utils import sample_outclass_neg  # noqa: E402

from golden_io import CASES, Fixture  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def eval_case(name):
    fx = Fixture(name)
    args = argparse.Namespace(**{k: v for k, v in fx.cfg.items() if k not in ("groups", "Lv", "Lw")})
    net = runner.build_model(args)
    net.load_state_dict(fx.sd)
    crit = runner.build_criterion(args)
    net.eval()
    crit.eval()
    rec = {}
    orig_neg = ref_model_mod.sample_outclass_neg

    def neg_wrap(num_clips):
        r = orig_neg(num_clips)
        rec["neg_index"] = r.clone()
        return r

    ref_model_mod.sample_outclass_neg = neg_wrap
    torch.manual_seed(77)
    try:
        with torch.no_grad():
            outputs = net(**fx.batch, dataset_name=args.dataset_name, is_training=False)
            losses, total = crit(outputs, fx.batch, is_training=False)
    finally:
        ref_model_mod.sample_outclass_neg = orig_neg
    blob = {"neg_index": rec["neg_index"].numpy()}
    for k, v in outputs.items():
        if isinstance(v, torch.Tensor):
            blob["out." + k] = v.detach().numpy()
    for i, a in enumerate(outputs["aux_outputs"]):
        for k, v in a.items():
            blob["out.aux%d.%s" % (i, k)] = v.detach().numpy()
    for k, v in losses.items():
        blob["loss." + k] = np.asarray(v.detach().numpy())
    blob["loss.total"] = np.asarray(total.detach().numpy())
    blob["keys.json"] = np.frombuffer(json.dumps(sorted(outputs.keys())).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + "_eval.npz")
    np.savez_compressed(path, **blob)
    print("%s_eval: total %.6f, keys %s" % (name, float(total), sorted(outputs.keys())))


def draws():
    """seed -> (neg_index, masked_words) from the reference's own functions."""
    fx = Fixture("qvh_tiny")
    args = argparse.Namespace(**{k: v for k, v in fx.cfg.items() if k not in ("groups", "Lv", "Lw")})
    net = runner.build_model(args)
    blob = {}
    cases = [
        dict(groups=[3, 1, 2], Lw=8, seed=5),
        dict(groups=[1] * 32, Lw=32, seed=6),
        dict(groups=[4] * 8, Lw=32, seed=7),
        dict(groups=[2, 7, 1, 1, 5], Lw=16, seed=8),
    ]
    for ci, c in enumerate(cases):
        N = sum(c["groups"])
        Lw = c["Lw"]
        wlen = [max(1, min(Lw, 1 + ((5 * i + ci) % Lw))) for i in range(N)]  # includes 1-word rows (skipped)
        wmask = torch.arange(Lw)[None, :] < torch.tensor(wlen)[:, None]
        weight = (1 + (torch.arange(N)[:, None] * 3 + torch.arange(Lw)[None, :]) % 4).long() * wmask
        num_clips = torch.tensor(c["groups"])
        src = torch.zeros(N, Lw, args.hidden_dim)
        for use_weight in (True, False):
            tag = "c%d.%s." % (ci, "w" if use_weight else "u")
            torch.manual_seed(c["seed"])
            np.random.seed(c["seed"])
            neg = sample_outclass_neg(num_clips)
            with torch.no_grad():
                _, masked = net._mask_words(src, wmask, net.masked_token, proj=True,
                                            weight=weight if use_weight else None)
            blob[tag + "neg_index"] = neg.numpy()
            blob[tag + "masked_words"] = masked.numpy()
        blob["c%d.groups" % ci] = np.array(c["groups"])
        blob["c%d.words_mask" % ci] = wmask.numpy()
        blob["c%d.words_weight" % ci] = weight.numpy()
        blob["c%d.seed" % ci] = np.array(c["seed"])
    blob["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "draws.npz"), **blob)
    print("draws: %d cases" % len(cases))


def clip_state_dict(width=128, layers=3, ctx=77, vocab=200, embed=64, seed=31):
    """A random CLIP text-tower state dict in the form the released checkpoints have after
    runner.build_CLIP_text_encoder's conversion: fp16 Linear / attention / projection tensors, fp32
    embeddings and LayerNorm parameters (text_encoder.py:373-394)."""
    import model.text_encoder as te
    torch.manual_seed(seed)
    enc = te.CLIPTextEncoder(embed, ctx, vocab, width, width // 64, layers)
    with torch.no_grad():
        for n, p in enc.named_parameters():  # the default init is tiny (std 0.01-0.09): make every term count
            if "ln_" in n:
                p.add_(torch.randn_like(p) * 0.2)
            elif n.endswith("bias"):
                p.normal_(0, 0.1)
            elif "embedding" in n:
                p.mul_(20.0)
            else:
                p.mul_(2.0)
    te.convert_weights(enc)
    return {k: v.clone() for k, v in enc.state_dict().items()}


def clip_cases():
    """clip_text_tiny.npz: CLIPTextEncoder.forward (text_encoder.py:340-354) + MESM.CLIP_encode_text
    (model.py:103-134) of the real reference on CPU in fp16, random weights at a small width; and
    qvh_clip_tiny.npz: a whole training step of the reference MESM built by runner.build_model with
    tokenizer_type='CLIP' (the shipped C+SF_C.json form), in the Fixture layout of gen_golden.py.
    The reference moves the encoder to "cuda" when its inputs are on the CPU (model.py:104-106, quirk Q10);
    here the wrapper is called with a non-CPU `device` argument, which skips those moves and nothing else."""
    import tempfile
    from mesm_amd import synthetic
    from gen_golden import TINY, flatten_targets
    sd_clip = clip_state_dict()
    tmp = os.path.join(tempfile.mkdtemp(), "clip_tiny.pth")
    torch.save(sd_clip, tmp)
    spec = dict(dataset_name="qvhighlights", v_feat_dim=18, t_feat_dim=128, vocab_size=197, share_MLP=True,
                set_cost_class=4, loss_label_coef=4, rank_coef=12, use_triplet=True, loss_recfw_coef=0.5,
                loss_recss_coef=0.1, tokenizer_type="CLIP", load_vocab_pkl=False, text_model_path=tmp)
    over = dict(TINY)
    over.update(spec)
    args = synthetic.make_args(None, **over)
    groups, Lv, Lw, seed = [3, 1, 2], 20, 8, 14
    torch.manual_seed(seed)
    np.random.seed(seed)
    net = runner.build_model(args)
    crit = runner.build_criterion(args)
    with torch.no_grad():
        for n_, p in net.named_parameters():
            if n_.startswith("text_encoder"):
                continue
            if n_.endswith("masked_token") or n_.endswith("unknown_token") or n_.endswith("masked_sent_token"):
                p.normal_(0, 0.5)
            if n_.endswith("activation.weight"):
                p.uniform_(0.1, 0.4)
            if "LayerNorm" in n_ or ".norm" in n_:
                p.add_(torch.randn_like(p) * 0.1)
    net.eval()
    feat = synthetic.make_batch(args.dataset_name, groups, Lv, Lw, args.v_feat_dim, args.t_feat_dim,
                                args.vocab_size + 3, seed=seed, ragged=True)
    batch = synthetic.with_clip_tokens(feat, vocab=200, seed=seed)
    orig = net.CLIP_encode_text
    net.CLIP_encode_text = lambda ids, mask, device: orig(ids, mask, torch.device("cuda"))

    # (1) the encoder alone and the wrapper
    with torch.no_grad():
        hid = net.text_encoder(batch["words_id"])["last_hidden_state"]
        wf, sf, wid, wmask = net.CLIP_encode_text(batch["words_id"], batch["words_mask"], None)
    blob = {"sd." + k: v.numpy() for k, v in sd_clip.items()}
    blob.update(ids=batch["words_id"].numpy(), mask=batch["words_mask"].numpy(), hidden=hid.numpy(),
                words_feat=wf.numpy(), sentence_feat=sf.numpy(), words_id_cut=wid.numpy(),
                words_mask_cut=wmask.numpy(), max_words_l=np.array(Lw))
    np.savez_compressed(os.path.join(OUT, "clip_text_tiny.npz"), **blob)
    print("clip_text_tiny: hidden", tuple(hid.shape), hid.dtype, "abs max %.3f" % float(hid.float().abs().max()))

    # (2) the whole step
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
        with torch.no_grad():
            idx_main = crit.matcher({k: v for k, v in outputs.items() if k != "aux_outputs"}, batch)
            idx_aux = [crit.matcher(a, batch) for a in outputs["aux_outputs"]]
    finally:
        ref_model_mod.sample_outclass_neg = orig_neg

    def idx_to_arrays(idx):
        q = torch.cat([a for a, _ in idx])
        t = torch.cat([b for _, b in idx])
        return q.numpy(), t.numpy(), np.array([len(a) for a, _ in idx])

    blob = {}
    for k, v in net.state_dict().items():
        blob["sd." + k] = v.detach().numpy()
    for k, p in net.named_parameters():
        if p.grad is not None:
            blob["grad." + k] = p.grad.detach().numpy()
    blob.update(flatten_targets(batch))
    blob["neg_index"] = rec["neg_index"].numpy()
    blob["masked_words"] = rec["masked_words"].numpy()
    for k, v in outputs.items():
        if isinstance(v, torch.Tensor):
            blob["out." + k] = v.detach().numpy()
    for i, a in enumerate(outputs["aux_outputs"]):
        for k, v in a.items():
            blob["out.aux%d.%s" % (i, k)] = v.detach().numpy()
    for k, v in losses.items():
        blob["loss." + k] = np.asarray(v.detach().numpy())
    blob["loss.total"] = np.asarray(total.detach().numpy())
    q, t, s_ = idx_to_arrays(idx_main)
    blob["match.main.q"], blob["match.main.t"], blob["match.main.sizes"] = q, t, s_
    for i, ia in enumerate(idx_aux):
        q, t, s_ = idx_to_arrays(ia)
        blob["match.aux%d.q" % i], blob["match.aux%d.t" % i], blob["match.aux%d.sizes" % i] = q, t, s_
    cfg = dict(vars(args))
    cfg.update(groups=groups, Lv=Lv, Lw=Lw, text_model_path=None)
    blob["cfg.json"] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    path = os.path.join(OUT, "qvh_clip_tiny.npz")
    np.savez_compressed(path, **blob)
    print("qvh_clip_tiny: total %.6f, %.1f KB, MLM classes %d" % (float(total), os.path.getsize(path) / 1024,
                                                                  outputs["recfw_words_logit"].shape[-1]))


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["eval", "draws", "clip"]
    if "eval" in which:
        for name in CASES:
            eval_case(name)
    if "draws" in which:
        draws()
    if "clip" in which:
        clip_cases()
