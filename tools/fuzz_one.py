"""Re-run ONE case of a fuzz sweep (configurations are drawn in order up to it) and print every gradient whose L2
distance from the fp64 oracle exceeds 1e-4, largest first.  usage: fuzz_one.py <case> <seed>"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import fuzz_parity as F
from oracle import mesm_oracle as O
case, seed = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed)
for c in range(case + 1):
    tag, spec = F.draw(rng, c)
print(tag)
args, model, crit, batch, neg, masked = F.build(spec)
if os.environ.get("FUZZ_KEEPALIVE"):  # hold every tensor any kernel wrapper sees until the step is over (lifetime bugs vanish)
    from mesm_amd import kernels as kn
    keep = []
    def wrap(fn):
        def w(*a, **k):
            keep.append((a, k))
            r = fn(*a, **k)
            keep.append(r)
            return r
        return w
    which = os.environ["FUZZ_KEEPALIVE"].split(",")
    for name in dir(kn):
        f = getattr(kn, name)
        if callable(f) and not name.startswith("_") and not isinstance(f, type) and (which == ["all"] or name in which):
            setattr(kn, name, wrap(f))
out, losses, total, grads = F.hip_step(model, crit, batch, spec["dataset"], neg, masked)
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
o64 = O.train_step64(sd, dict(vars(args)), batch, neg, masked)
print("total %.7f oracle64 %.7f" % (float(total), float(o64[2])))
rows = sorted(((F.l2(grads[k], g.float()), k) for k, g in o64[3].items()), reverse=True)
for e, k in rows[:12]:
    if e > 1e-4:
        print("  %.3e  %s  (norm %.3e)" % (e, k, float(o64[3][k].norm())))
for k, v in o64[0].items():
    if torch.is_tensor(v) and v.dtype != torch.bool and k in out:
        e = F.rel(out[k], v.float())
        if e > 1e-5:
            print("  out %s %.2e" % (k, e))
