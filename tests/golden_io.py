"""Load the fixtures written by tools/gen_golden.py (outputs of the real reference)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["charades_tiny", "qvh_tiny", "tacos_tiny"]
CLIP_CASES = ["qvh_clip_tiny"]  # reference built with tokenizer_type="CLIP" (fp16 text tower, token ids in)
# tools/gen_golden_variants.py: the switches no shipped config flips (FW-MESM / SS-MESM off, no auxiliary losses,
# other layer counts and projection depths), run through the real reference at a very small width
VARIANTS = ["variants/" + n for n in ("qvh_plain", "qvh_fw_only", "qvh_ss_only", "cha_plain", "cha_ss_only",
                                      "qvh_no_aux", "qvh_depths", "cha_proj1", "qvh_txt_pos",
                                      "cha_txt_pos_fw_only")]


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.cfg = json.loads(bytes(z["cfg.json"]).decode())
        self.sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("sd.")}
        self.grads = {k[5:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("grad.")}
        self.out = {k[4:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("out.")}
        self.losses = {k[5:]: float(z[k]) for k in z.files if k.startswith("loss.")}
        self.match = {k[6:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("match.")}
        self.neg_index = torch.from_numpy(z["neg_index"].copy())
        self.masked_words = torch.from_numpy(z["masked_words"].copy()).bool() if "masked_words" in z.files else None
        batch = {}
        for k in z.files:
            if not k.startswith("batch."):
                continue
            parts = k.split(".")
            if len(parts) == 2:
                batch[parts[1]] = torch.from_numpy(z[k].copy())
        for key, field in (("norm_moment", "moments"), ("norm_span", "spans")):
            if "batch.%s.sizes" % key in z.files:
                sizes = z["batch.%s.sizes" % key].tolist()
                cat = torch.from_numpy(z["batch.%s.cat" % key].copy())
                batch[key] = [{field: c} for c in torch.split(cat, sizes)]
        batch.setdefault("words_mask", None)
        self.batch = batch

    def matched_pairs(self, layer="main"):
        """Set of (pair, query, target) triples the reference matcher produced."""
        q, t, sizes = (self.match["%s.%s" % (layer, f)] for f in ("q", "t", "sizes"))
        res, k = set(), 0
        for b, s in enumerate(sizes.tolist()):
            for _ in range(s):
                res.add((b, int(q[k]), int(t[k]) if self.cfg["dataset_name"] == "qvhighlights" else 0))
                k += 1
        return res


class EvalFixture:
    """<case>_eval.npz (tools/gen_golden_r2.py): the reference's inference call (eval.py:63,102) on the
    weights / batch of <case>.npz."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + "_eval.npz"))
        self.out = {k[4:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("out.")}
        self.losses = {k[5:]: float(z[k]) for k in z.files if k.startswith("loss.")}
        self.neg_index = torch.from_numpy(z["neg_index"].copy())
        self.keys = json.loads(bytes(z["keys.json"]).decode())


def draw_cases():
    """draws.npz: [(groups, words_mask, words_weight, seed, {tag: (neg_index, masked_words)})]."""
    z = np.load(os.path.join(GOLDEN, "draws.npz"))
    res = []
    for ci in range(int(z["n_cases"])):
        got = {}
        for tag in ("w", "u"):
            got[tag] = (torch.from_numpy(z["c%d.%s.neg_index" % (ci, tag)].copy()),
                        torch.from_numpy(z["c%d.%s.masked_words" % (ci, tag)].copy()))
        res.append((z["c%d.groups" % ci].tolist(), torch.from_numpy(z["c%d.words_mask" % ci].copy()),
                    torch.from_numpy(z["c%d.words_weight" % ci].copy()), int(z["c%d.seed" % ci]), got))
    return res
