"""saliency loss kernels against the oracle's autograd on the inputs of a dumped fuzz case (scores random or given)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from mesm_amd import kernels as kn, synthetic
from oracle import mesm_oracle as O
import fuzz_parity as F, random
case, seed = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(seed)
for c in range(case + 1):
    tag, spec = F.draw(rng, c)
print(tag)
args, model, crit, batch, neg, masked = F.build(spec)
cfg = dict(vars(args))
dev = torch.device("cuda:0")
N, L = batch["video_mask"].shape
for scale in (1.0, 4.0, 12.0):
    g = torch.Generator().manual_seed(3)
    sp = (torch.randn(N, L, generator=g) * scale).double().requires_grad_(True)
    sn = (torch.randn(N, L, generator=g) * scale).double().requires_grad_(True)
    out = {"saliency_scores": sp, "neg_saliency_scores": sn}
    t64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    loss = O.loss_saliency(out, t64, cfg)["loss_saliency"]
    loss.backward()
    label = (batch["saliency_label"] if "saliency_label" in batch else batch["clip_mask"]).to(torch.float64).contiguous().to(dev)
    vm = batch["video_mask"].contiguous().to(dev)
    pi = batch["pos_idx"].contiguous().to(dev) if cfg["use_triplet"] else None
    ni = batch["neg_idx"].contiguous().to(dev) if cfg["use_triplet"] else None
    spd, snd = sp.detach().float().to(dev).contiguous(), sn.detach().float().to(dev).contiguous()
    lv = kn.saliency_loss_fwd(spd, snd, label, vm, pi, ni, float(cfg["rank_coef"]), float(cfg["saliency_margin"]))
    gs = torch.ones(1, device=dev)
    dsp, dsn = kn.saliency_loss_bwd(spd, snd, label, vm, pi, ni, float(cfg["rank_coef"]), float(cfg["saliency_margin"]), gs)
    e1 = float((dsp.cpu().double() - sp.grad).abs().max() / sp.grad.abs().max())
    e2 = float((dsn.cpu().double() - sn.grad).abs().max() / sn.grad.abs().max())
    print("scale %.0f: loss %.7f vs %.7f; ds_pos max rel err %.2e, ds_neg %.2e" % (scale, float(lv), float(loss), e1, e2))
    bad = ((dsp.cpu().double() - sp.grad).abs() > 1e-4 * sp.grad.abs().max()).nonzero()
    for b in bad[:6].tolist():
        n_, l_ = b
        print("   pos[%d,%d]: kernel %.6f oracle %.6f label %.1f vmask %d" % (n_, l_, float(dsp[n_, l_]), float(sp.grad[n_, l_]), float(label[n_, l_]), int(vm[n_, l_])))
    bad = ((dsn.cpu().double() - sn.grad).abs() > 1e-4 * sn.grad.abs().max()).nonzero()
    for b in bad[:6].tolist():
        n_, l_ = b
        print("   neg[%d,%d]: kernel %.6f oracle %.6f vmask %d s %.3f" % (n_, l_, float(dsn[n_, l_]), float(sn.grad[n_, l_]), int(vm[n_, l_]), float(sn[n_, l_])))
print("pos_idx", batch.get("pos_idx"), "neg_idx", batch.get("neg_idx"))

# ---- the same with the ACTUAL scores of the case's forward
model.eval()
b = synthetic.to_device(batch, dev)
outm = model(**b, dataset_name=spec["dataset"], is_training=True, neg_index=neg, masked_words=masked)
sp = outm["saliency_scores"].detach().cpu().double().requires_grad_(True)
sn = outm["neg_saliency_scores"].detach().cpu().double().requires_grad_(True)
print("actual scores: pos range [%.2f, %.2f], neg range [%.2f, %.2f]" % (float(sp.min()), float(sp.max()), float(sn.min()), float(sn.max())))
t64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
loss = O.loss_saliency({"saliency_scores": sp, "neg_saliency_scores": sn}, t64, cfg)["loss_saliency"]
loss.backward()
spd, snd = sp.detach().float().to(dev).contiguous(), sn.detach().float().to(dev).contiguous()
dsp, dsn = kn.saliency_loss_bwd(spd, snd, label, vm, pi, ni, float(cfg["rank_coef"]), float(cfg["saliency_margin"]), torch.ones(1, device=dev))
print("actual: ds_pos L2 rel %.2e  ds_neg L2 rel %.2e" % (float((dsp.cpu().double() - sp.grad).norm() / sp.grad.norm()), float((dsn.cpu().double() - sn.grad).norm() / sn.grad.norm())))
d = (dsp.cpu().double() - sp.grad).abs()
for idx in d.flatten().argsort(descending=True)[:6].tolist():
    n_, l_ = divmod(idx, L)
    print("   pos[%d,%d]: kernel %.7f oracle %.7f score %.4f label %.1f vmask %d" % (n_, l_, float(dsp[n_, l_]), float(sp.grad[n_, l_]), float(sp[n_, l_]), float(label[n_, l_]), int(vm[n_, l_])))
d = (dsn.cpu().double() - sn.grad).abs()
for idx in d.flatten().argsort(descending=True)[:6].tolist():
    n_, l_ = divmod(idx, L)
    print("   neg[%d,%d]: kernel %.7f oracle %.7f score %.4f vmask %d" % (n_, l_, float(dsn[n_, l_]), float(sn.grad[n_, l_]), float(sn[n_, l_]), int(vm[n_, l_])))
