"""16-byte LDS-DMA from rows that are only 8- / 4-byte aligned (ld = 2818 / 5003): correctness and time per
forced kernel.  (The experiment that removed the alignment rule from wstage_ok: 78 -> 51 us at Dv = 2818.)
usage: unaligned_probe.py [M N ld K]   (operands are [:, :K] views of ld-wide matrices)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, KF, K = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (2400, 256, 2818, 2816)
Af = torch.randn(M, KF, device=dev); Bf = torch.randn(N, KF, device=dev)
A, B = Af[:, :K], Bf[:, :K]
ref = (A.double() @ B.double().t()).float()
relax = ""
if True:
    for tile in ("0", "3", "4", "2"):
        kn.gemm_switches(tile=int(tile))
        C = torch.zeros(M, N, device=dev)
        kn.gemm(A, B, C, trans_b=True)
        torch.cuda.synchronize()
        err = (C - ref).abs().max().item() / ref.abs().max().item()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            kn.gemm(A, B, C, trans_b=True)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(16): kn.gemm(A, B, C, trans_b=True)
        g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 160 * 1e6
        print("relax=%s tile=%s: rel err %.2e  %.1f us  %.1f TF" % (relax or "0", tile, err, us, 2.0 * M * N * K / us / 1e6), flush=True)
