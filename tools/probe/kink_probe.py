"""Does a fuzz mismatch come from an activation kink on the DEVICE side?  Every PReLU pre-activation z of the HIP forward
(the second output of the FFN's first GEMM) against the fp64 oracle's: elements whose SIGN differs.
usage: kink_probe.py <case> <seed>"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import fuzz_parity as F
from mesm_amd import kernels as kn
from oracle import mesm_oracle as O
case, seed = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed)
for c in range(case + 1):
    tag, spec = F.draw(rng, c)
print(tag)
args, model, crit, batch, neg, masked = F.build(spec)
zs_hip = []
orig = kn.gemm
def spy(A, B, C, **kw):
    r = orig(A, B, C, **kw)
    if kw.get("pre_out") is not None:
        zs_hip.append(kw["pre_out"])
    return r
kn.gemm = spy
out, losses, total, grads = F.hip_step(model, crit, batch, spec["dataset"], neg, masked)
kn.gemm = orig
torch.cuda.synchronize()
zs_or = []
po = O._prelu
def ospy(x, slope):
    zs_or.append(x.detach().clone()); return po(x, slope)
O._prelu = ospy
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
o64 = O.train_step64(sd, dict(vars(args)), batch, neg, masked)
O._prelu = po
print("HIP FFN pre-activations: %d, oracle PReLU calls: %d" % (len(zs_hip), len(zs_or)))
for i, zh in enumerate(zs_hip):
    zh2 = zh.detach().cpu().double().reshape(-1, zh.shape[-1])
    # the stacked (positive + negative) pass of the HIP path holds two oracle calls back to back: every row block of the
    # oracle calls' extent gets its own best match
    cands = [zo.reshape(-1, zo.shape[-1]) for zo in zs_or if zo.shape[-1] == zh2.shape[1]]
    sizes = sorted({c.shape[0] for c in cands if c.shape[0] <= zh2.shape[0] and zh2.shape[0] % c.shape[0] == 0})
    if not sizes:
        print("  hip z #%d %s: no oracle call of a matching extent" % (i, tuple(zh.shape)))
        continue
    n = sizes[0] if zh2.shape[0] // sizes[0] <= 2 else sizes[-1]
    for off in range(0, zh2.shape[0], n):
        seg = zh2[off:off + n]
        best = min(((float((seg - c).abs().max()), j) for j, c in enumerate(cands) if c.shape[0] == n), default=None)
        if best is None or best[0] > 1e-3:
            print("  hip z #%d rows [%d, %d): no oracle match (%s)" % (i, off, off + n, best))
            continue
        zo2 = cands[best[1]]
        flips = ((seg > 0) != (zo2 > 0)).nonzero()
        print("  hip z #%d %s rows [%d, %d): max |diff| %.2e, sign flips %d" % (i, tuple(zh.shape), off, off + n, best[0], flips.shape[0]))
        for r, c_ in flips[:4].tolist():
            print("      [%d, %d]: hip %.3e  oracle64 %.3e" % (r + off, c_, float(seg[r, c_]), float(zo2[r, c_])))
