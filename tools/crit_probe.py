"""Where the merged criterion launches spend their time: each block alone and together, at the headline extents
(32 pairs, 10 queries, 2 decoder layers, Lv 75, Lw 32, C 5003, D 256).  Usage: python tools/crit_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
def R(*s): return torch.randn(*s, generator=g).to(dev)
N, Q, tmax, Lw, C, Lv, Le, D, L = 32, 10, 5, 32, 5003, 75, 33, 256, 75
sizes = [1 + (i % tmax) for i in range(N)]
off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32)
T = int(off[-1]); st = torch.rand(T, generator=g) * 0.6; ed = st + 0.05 + torch.rand(T, generator=g) * 0.3
xx = torch.stack([st, ed], 1).to(dev); cxw = torch.stack([(st + ed) * 0.5, ed - st], 1).to(dev); off = off.to(dev)
lay = [(R(N, Q, 2), torch.sigmoid(R(N, Q, 2)), 4 * l) for l in range(2)]
setb = dict(Q=Q, Tmax=tmax, w_span=10.0, w_giou=1.0, w_class=4.0, eos_coef=0.1, tgt_cxw=cxw, tgt_xx=xx, tgt_off=off, layers=lay)
sp, sn = R(N, L), R(N, L)
label = torch.randint(0, 5, (N, L), generator=g).double().to(dev)
vmask = (torch.rand(N, L, generator=g) < 0.8).to(dev)
pos_idx = torch.randint(0, L, (N, 2), generator=g).to(dev); neg_idx = torch.randint(0, L, (N, 2), generator=g).to(dev)
salb = dict(s_pos=sp, s_neg=sn, label=label, vmask=vmask, pos_idx=pos_idx, neg_idx=neg_idx, rank_coef=12.0, margin=0.2, slot=8)
logit = R(N, Lw, C); wl = torch.randint(0, C, (N * Lw,), generator=g).to(dev)
mask = (torch.arange(Lw)[None] < torch.tensor([4 + (5 * i) % (Lw - 3) for i in range(N)])[:, None]).to(dev)
fwb = dict(logit=logit, label=wl, mask=mask, eps=0.1, slot=9)
pv, ew = R(N, Lv, D), R(N, Le, D)
cmask = (torch.rand(N, Lv, generator=g) < 0.3); cmask[:, 0] = True
wmask = (torch.rand(N, Le, generator=g) < 0.7); wmask[:, 0] = True
pos8 = ((torch.rand(N, N, generator=g) < 0.2) | torch.eye(N, dtype=torch.bool)).to(torch.uint8).to(dev)
ssb = dict(pv=pv, cmask=cmask.to(dev), ew=ew, wmask=wmask.to(dev), pos=pos8, tau=0.5, slot=11)
wv = torch.rand(12, generator=g).to(dev) + 0.1
lv = torch.zeros(12, device=dev)

def timed(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

print("forward (three launches each: grid + ss rows [when rec_ss is on] + finishing workgroup), us per call, eager launches")
for name, kw in [("nothing but the tail", dict(sal=salb)), ("set losses", dict(set_losses=setb)), ("saliency", dict(sal=salb)),
                 ("rec_fw", dict(recfw=fwb)), ("rec_ss", dict(recss=ssb)),
                 ("all", dict(set_losses=setb, sal=salb, recfw=fwb, recss=ssb))]:
    print("  %-22s %7.1f" % (name, timed(lambda: kn.criterion_fwd(lv, wv, N, **kw))))
print("separate launches, us per call")
o4 = torch.zeros(4, device=dev)
print("  set_loss_fwd_layers    %7.1f" % timed(lambda: kn.set_loss_fwd_layers([(a, b, o4) for a, b, _ in lay], cxw, xx, off, tmax, 10.0, 1.0, 4.0, 0.1)))
print("  saliency_loss_fwd      %7.1f" % timed(lambda: kn.saliency_loss_fwd(sp, sn, label, vmask, pos_idx, neg_idx, 12.0, 0.2)))
print("  nll_smooth_fwd         %7.1f" % timed(lambda: kn.nll_smooth_fwd(logit.view(-1, C), wl, mask.view(-1), 0.1)))
print("  rec_ss_fwd (3)         %7.1f" % timed(lambda: kn.rec_ss_fwd(pv, ssb["cmask"], ew, ssb["wmask"], pos8, 0.5, o4[:1])))
total, out = kn.criterion_fwd(lv, wv, N, set_losses=setb, sal=salb, recfw=fwb, recss=ssb)
gt = torch.ones(1, device=dev)
bl = dict(set_losses=dict(Q=Q, eos_coef=0.1, tgt_cxw=cxw, tgt_xx=xx, tgt_off=off,
                          layers=[(a, b, m, torch.empty_like(a), torch.empty_like(b), s) for (a, b, s), m in zip(lay, out["match"])]),
          sal=dict(salb, ds_pos=torch.empty_like(sp), ds_neg=torch.empty_like(sn)),
          recfw=dict(logit=logit, label=wl, row_lse=out["row_lse"], mask=mask, eps=0.1, dlogit=torch.empty_like(logit), slot=9),
          recss=dict(saved=out["recss"], pos=pos8, cmask=ssb["cmask"], wmask=ssb["wmask"], Lv=Lv, Le=Le, tau=0.5,
                     dpv=torch.empty_like(pv), dew=torch.empty_like(ew), slot=11))
print("backward (one launch), us per call")
for name, keys in [("set losses", ["set_losses"]), ("saliency", ["sal"]), ("rec_fw", ["recfw"]), ("rec_ss", ["recss"]),
                   ("all", list(bl))]:
    print("  %-22s %7.1f" % (name, timed(lambda: kn.criterion_bwd(gt, wv, N, **{k: bl[k] for k in keys}))))
