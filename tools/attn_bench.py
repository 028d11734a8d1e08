"""Attention core timing (forward / backward) at the step's shapes, captured chains of 16 launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
SHAPES = [("T2V   Lq75 Lk33", 64, 8, 75, 33, 32, 32), ("enc   Lq76 Lk76", 64, 8, 76, 76, 32, 32),
          ("V2T   Lq33 Lk75", 64, 8, 33, 75, 32, 32), ("dec sa Lq10 Lk10", 32, 8, 10, 10, 32, 32),
          ("dec ca Lq10 Lk75", 32, 8, 10, 75, 64, 32), ("tacos enc 513x513", 32, 8, 513, 513, 32, 32),
          ("tacos t2v 512x17", 32, 8, 512, 17, 32, 32)]


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


for name, B, H, Lq, Lk, dk, dv in SHAPES:
    q = torch.randn(B, Lq, H * dk, device=dev); k = torch.randn(B, Lk, H * dk, device=dev)
    v = torch.randn(B, Lk, H * dv, device=dev); do = torch.randn(B, Lq, H * dv, device=dev)
    o, lse = kn.attn_fwd(q, k, v, H, drop=(0.1, 5))
    dq = torch.zeros_like(q); dkk = torch.empty_like(k); dvv = torch.empty_like(v)
    tf = timed(lambda: kn.attn_fwd(q, k, v, H, drop=(0.1, 5)))
    tb = timed(lambda: kn.attn_bwd_into(do, q, k, v, o, lse, H, dq, dkk, dvv, drop=(0.1, 5)))
    print("%s  B%d H%d: fwd %6.2f us  bwd %6.2f us" % (name, B, H, tf, tb), flush=True)
