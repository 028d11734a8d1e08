"""Audit every GEMM of one fuzz-case step against an fp64 torch evaluation of the same call (operands kept alive until
the step is over): finds the launch whose result is off.  usage: gemm_audit.py <case> <seed> [only_loss_coef]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import fuzz_parity as F
from mesm_amd import kernels as kn, synthetic
from mesm_amd._lib import ACT_NONE, ACT_PRELU, ACT_RELU
case, seed = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed)
for c in range(case + 1):
    tag, spec = F.draw(rng, c)
print(tag)
if len(sys.argv) > 3:
    base = vars(synthetic.make_args(None, **spec["over"]))
    over = dict(spec["over"])
    for c in ["loss_span_coef", "loss_giou_coef", "loss_label_coef", "loss_saliency_coef", "loss_recfw_coef", "loss_recss_coef"]:
        over[c] = base[c] if c == sys.argv[3] else 0.0
    spec = dict(spec, over=over)
args, model, crit, batch, neg, masked = F.build(spec)
calls = []
orig = kn.gemm


def spy(A, B, C, **kw):
    pre = {}
    if kw.get("accumulate", 0) in (1, 2) or kw.get("split_k", 1) > 1:
        pre["C0"] = C.detach().clone()
    calls.append((A, B, C, dict(kw), pre))
    return orig(A, B, C, **kw)


kn.gemm = spy
import mesm_amd.ops as ops
out, losses, total, grads = F.hip_step(model, crit, batch, spec["dataset"], neg, masked)
kn.gemm = orig
torch.cuda.synchronize()
print("gemm calls:", len(calls))
worst = []
for i, (A, B, C, kw, pre) in enumerate(calls):
    if kw.get("a_drop", (0, 0))[0] or kw.get("b_drop", (0, 0))[0] or kw.get("e_drop", (0, 0))[0]:
        continue
    a = A.double()
    if kw.get("A2") is not None: a = a + kw["A2"].double()
    b = B.double()
    if kw.get("B2") is not None: b = b + kw["B2"].double()
    sl = float(kw["slope"]) if kw.get("slope") is not None else 0.0
    def act(x, k):
        if k == ACT_RELU: return x.clamp(min=0)
        if k == ACT_PRELU: return torch.where(x > 0, x, sl * x)
        return x
    a, b = act(a, kw.get("a_act", ACT_NONE)), act(b, kw.get("b_act", ACT_NONE))
    if kw.get("trans_a"): a = a.t()
    if kw.get("trans_b"): b = b.t()
    r = (a @ b) * kw.get("out_scale", 1.0)
    if kw.get("bias") is not None: r = r + kw["bias"].double()
    r = act(r, kw.get("e_act", ACT_NONE))
    ag = kw.get("e_actgrad", ACT_NONE)
    if ag != ACT_NONE:
        z = kw["aux"].double()
        r = torch.where(z > 0, r, (sl if ag == ACT_PRELU else 0.0) * r)
    if kw.get("residual") is not None: r = r + kw["residual"].double()
    if "C0" in pre:
        continue  # accumulating calls (weight gradients): audited through the parameter gradients
    e = float((C.double() - r).abs().max() / r.abs().max().clamp_min(1e-30))
    worst.append((e, i, tuple(A.shape), tuple(B.shape), {k: (v if not torch.is_tensor(v) else "T") for k, v in kw.items() if k in ("trans_a", "trans_b", "e_act", "e_actgrad", "a_act", "b_act")}))
worst.sort(reverse=True)
for w in worst[:12]:
    print("  %.2e  call %d A%s B%s %s" % w)
