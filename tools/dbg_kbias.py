import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, Lq, Lk, dh, dv = 2, 1, 10, 76, 32, 32
d = H * dh
def run():
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    qc, qs, kc, kp, v, do = r(B, Lq, d), r(B, Lq, d), r(B, Lk, d), r(B, Lk, d), r(B, Lk, H * dv), r(B, Lq, H * dv)
    kpad = (torch.arange(Lk, device=dev)[None, :] >= torch.tensor([Lk - 5, Lk], device=dev)[:, None])
    o, lse = kn.attn_fwd(qc, kc, v, H, kpad=kpad, q2=qs, k2=kp)
    dq, dk, dvv, dq2, dk2 = kn.attn_bwd(do, qc, kc, v, o, lse, H, kpad=kpad, q2=qs, k2=kp)
    # fp64 reference
    q64 = torch.cat([qc, qs], -1).double(); k64 = torch.cat([kc, kp], -1).double().requires_grad_()
    s = (q64 @ k64.transpose(1, 2)) * (2 * dh) ** -0.5
    s = s.masked_fill(kpad[:, None, :], float("-inf"))
    o64 = torch.softmax(s, -1) @ v.double()
    o64.backward(do.double())
    ref = k64.grad[..., dh:]
    print("dk2 rel err %.2e   colsum |ours| %.3e  |ref| %.3e   |dk2| %.3e" % (
        float((dk2.double() - ref).abs().max() / ref.abs().max()), float(dk2.double().sum((0, 1)).norm()),
        float(ref.sum((0, 1)).norm()), float(ref.norm())))
    print("row sums of dk2 over keys per batch:", dk2.double().sum(1).norm(dim=-1).tolist(), "ref", ref.sum(1).norm(dim=-1).tolist())
run()
