"""Bisect a failing fuzz case by configuration: the case with one setting changed at a time, worst gradient distance
from the fp64 oracle.  usage: fuzz_vary.py <case> <seed>"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import fuzz_parity as F
from oracle import mesm_oracle as O
case, seed = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed)
for c in range(case + 1):
    tag, spec0 = F.draw(rng, c)
print(tag)
VARS = [("base", {}, {}), ("ragged=0", {"ragged": False}, {}), ("Lw=8", {"Lw": 8}, {"max_words_l": 8}), ("Lw=16", {"Lw": 16}, {"max_words_l": 16}),
        ("Lv=33", {"Lv": 33}, {"max_video_l": 33}), ("groups=1111", {"groups": [1, 1, 1, 1]}, {}), ("groups=2222", {"groups": [2, 2, 2, 2]}, {}),
        ("enc_layers=1", {}, {"enc_layers": 1}), ("t2v_layers=1", {}, {"t2v_layers": 1}), ("rec_ss=0", {}, {"rec_ss": False}),
        ("rec_fw=0", {}, {"rec_fw": False}), ("ff=32", {}, {"dim_feedforward": 32}), ("d=64,h=2", {}, {"hidden_dim": 64, "nheads": 2, "dim_feedforward": 128}),
        ("seed+1", {"seed": spec0["seed"] + 1}, {}), ("dec_layers=1", {}, {"dec_layers": 1}), ("share_MLP flip", {}, {"share_MLP": not spec0["over"]["share_MLP"]})]
for name, sv, ov in VARS:
    spec = dict(spec0, **sv); spec["over"] = dict(spec0["over"], **ov)
    try:
        args, model, crit, batch, neg, masked = F.build(spec)
        out, losses, total, grads = F.hip_step(model, crit, batch, spec["dataset"], neg, masked)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o64 = O.train_step64(sd, dict(vars(args)), batch, neg, masked)
        rows = sorted(((F.l2(grads[k], g.float()), k) for k, g in o64[3].items() if k in grads), reverse=True)
        print("%-16s worst %.1e %s; %.1e %s" % (name, rows[0][0], rows[0][1], rows[2][0], rows[2][1]), flush=True)
    except Exception as e:
        print("%-16s ERROR %s" % (name, e))
