"""Localise a backward discrepancy of one fuzz case: the same configuration with one loss weight at a time (the others 0),
worst gradient distances from the fp64 oracle.  usage: fuzz_localize.py <case> <seed>"""
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
COEFS = ["loss_span_coef", "loss_giou_coef", "loss_label_coef", "loss_saliency_coef", "loss_recfw_coef", "loss_recss_coef"]
from mesm_amd import synthetic
base = vars(synthetic.make_args(None, **spec0["over"]))
for keep in COEFS + ["all"]:
    spec = dict(spec0); over = dict(spec0["over"])
    if keep != "all":
        for c in COEFS:
            over[c] = base[c] if c == keep else 0.0
    spec["over"] = over
    args, model, crit, batch, neg, masked = F.build(spec)
    out, losses, total, grads = F.hip_step(model, crit, batch, spec["dataset"], neg, masked)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    o64 = O.train_step64(sd, dict(vars(args)), batch, neg, masked)
    rows = sorted(((F.l2(grads[k], g.float()), k) for k, g in o64[3].items() if k in grads), reverse=True)
    print("only %-20s total %.6f / %.6f  worst: %s" % (keep, float(total), float(o64[2]), "; ".join("%.1e %s" % r for r in rows[:3])), flush=True)

# every parameter's distance in the run with only the loss named in argv[3]
if len(sys.argv) > 3:
    keep = sys.argv[3]
    spec = dict(spec0); over = dict(spec0["over"])
    for c in COEFS:
        over[c] = base[c] if c == keep else 0.0
    spec["over"] = over
    args, model, crit, batch, neg, masked = F.build(spec)
    out, losses, total, grads = F.hip_step(model, crit, batch, spec["dataset"], neg, masked)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    o64 = O.train_step64(sd, dict(vars(args)), batch, neg, masked)
    print("parameters in model order, only", keep)
    for k, p in model.named_parameters():
        if k in o64[3] and k in grads:
            e = F.l2(grads[k], o64[3][k].float())
            print("  %s %.2e  %s (norm %.2e)" % ("**" if e > 1e-4 else "  ", e, k, float(o64[3][k].norm())))
    print("neg_index", neg.tolist())
