"""Load the fixtures written by tools/gen_golden.py (outputs of the real reference)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["charades_tiny", "qvh_tiny", "tacos_tiny"]


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
        self.masked_words = torch.from_numpy(z["masked_words"].copy()).bool()
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
        batch["words_mask"] = None
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
