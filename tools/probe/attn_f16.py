"""attention cores under MESM_ATTN_F16X=0|1 (f32 matrix instruction | three fp16 products over two-term split operands): error against
fp64 on operands of several magnitudes (gradients of 1e-7, activations of 1e-3 ... 1e3) and time per launch.
usage: MESM_ATTN_F16X=1 python tools/probe/attn_f16.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")


def ref_attn(q, k, v, H, kpad, scale):
    B, Lq, _ = q.shape; Lk = k.shape[1]
    qh = q.view(B, Lq, H, -1).transpose(1, 2); kh = k.view(B, Lk, H, -1).transpose(1, 2); vh = v.view(B, Lk, H, -1).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if kpad is not None:
        s = s.masked_fill(kpad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    return (p @ vh).transpose(1, 2).reshape(B, Lq, -1)


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-300))


def timed(fn, n=16):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 / n * 1e6


print("MESM_ATTN_F16X =", os.environ.get("MESM_ATTN_F16X", "(default)"))
for name, B, H, Lq, Lk, qs, ks, vs, gs in [("513x513", 4, 8, 513, 513, 1, 1, 1, 1), ("513x513 dO 1e-7", 4, 8, 513, 513, 1, 1, 1, 1e-7),
                                            ("513x513 q 1e-3 k 1e3", 4, 8, 513, 513, 1e-3, 1e3, 1, 1), ("513x513 v 1e4 dO 1e-5", 4, 8, 513, 513, 1, 1, 1e4, 1e-5),
                                            ("76x76", 8, 8, 76, 76, 1, 1, 1, 1), ("76x76 dO 1e-7", 8, 8, 76, 76, 1, 1, 1, 1e-7),
                                            ("75x33", 8, 8, 75, 33, 1, 1, 1, 1), ("33x75", 8, 8, 33, 75, 1, 1, 1, 1e-6),
                                            ("300x130", 2, 4, 300, 130, 1, 1, 1, 1e-4)]:
    g = torch.Generator().manual_seed(Lq + Lk)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    q, k, v, do = r(B, Lq, H * 32) * qs, r(B, Lk, H * 32) * ks, r(B, Lk, H * 32) * vs, r(B, Lq, H * 32) * gs
    lens = torch.randint(Lk // 2, Lk + 1, (B,), generator=g); lens[0] = Lk
    kpad = (torch.arange(Lk)[None, :] >= lens[:, None]).to(dev)
    scale = 32 ** -0.5
    o, lse = kn.attn_fwd(q, k, v, H, kpad=kpad, scale=scale)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = ref_attn(qd, kd, vd, H, kpad, scale)
    ref.backward(do.double())
    dq, dk_, dv_ = kn.attn_bwd(do, q, k, v, o, lse, H, kpad=kpad, scale=scale)
    print("%-24s o %.1e  dq %.1e  dk %.1e  dv %.1e" % (name, rel(o, ref.detach()), rel(dq, qd.grad), rel(dk_, kd.grad), rel(dv_, vd.grad)), flush=True)
for name, B, H, Lq, Lk in [("tacos enc 513x513", 32, 8, 513, 513), ("enc 76x76", 64, 8, 76, 76), ("T2V 75x33", 64, 8, 75, 33), ("V2T 33x75", 64, 8, 33, 75)]:
    q = torch.randn(B, Lq, H * 32, device=dev); k = torch.randn(B, Lk, H * 32, device=dev)
    v = torch.randn(B, Lk, H * 32, device=dev); do = torch.randn(B, Lq, H * 32, device=dev)
    o, lse = kn.attn_fwd(q, k, v, H, drop=(0.1, 5))
    dq = torch.zeros_like(q); dkk = torch.empty_like(k); dvv = torch.empty_like(v)
    tf = timed(lambda: kn.attn_fwd(q, k, v, H, drop=(0.1, 5)))
    tb = timed(lambda: kn.attn_bwd_into(do, q, k, v, o, lse, H, dq, dkk, dvv, drop=(0.1, 5)))
    print("%s  B%d H%d: fwd %6.2f us  bwd %6.2f us" % (name, B, H, tf, tb), flush=True)
