"""Accuracy and speed of the split GEMM products (MESM_GEMM_BF16X = 6: three bf16 terms, six products | 2: two fp16 terms,
three products) against exact f32 MFMA,
per shape: relative error against an fp64 product and time per launch in a captured chain.
usage: python tools/bf16x_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
SHAPES = [(4800, 1024, 256, False, True), (4800, 256, 1024, False, True), (4800, 256, 256, False, False),
          (1024, 256, 4800, True, False), (2400, 256, 2818, False, True), (1024, 5003, 256, False, True)]


def run(M, N, K, ta, tb, mode):
    if mode: kn.gemm_switches(bf16x=int(str(mode)))
    else: kn.gemm_switches(bf16x=6)
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g).to(dev)
    B = (torch.randn((N, K) if tb else (K, N), generator=g) * 0.06).to(dev)
    C = torch.zeros(M, N, device=dev)
    split = 4 if ta else 1
    kw = dict(trans_a=ta, trans_b=tb, split_k=split, accumulate=2 if split > 1 else 0)
    kn.gemm(A, B, C, **kw)
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
    rms = ((C.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    sets = [(torch.randn_like(A), torch.randn_like(B), torch.zeros_like(C)) for _ in range(4)]
    def body():
        for i in range(16):
            a, b, c = sets[i % 4]
            kn.gemm(a, b, c, **kw)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): body()
    gr.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): gr.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 160 * 1e6
    return err, rms, us


for M, N, K, ta, tb in SHAPES:
    for mode in (0, 6, 2):
        err, rms, us = run(M, N, K, ta, tb, mode)
        print("M=%5d N=%5d K=%5d %s%s  %-8s max rel err %.2e  rms rel err %.2e  %7.2f us  %6.1f TF" % (
            M, N, K, "T" if ta else "N", "T" if tb else "N", {0: "f32", 6: "bf16x6", 2: "f16x3"}[mode], err, rms, us,
            2.0 * M * N * K / us / 1e6), flush=True)
kn.gemm_switches(bf16x=6)
