"""Sweep one GEMM shape over forced kernels and split-k factors (captured chain of 32 launches over 8
rotating operand sets, like gemm_bench4).  usage: gemm_sweep.py M N K ta tb "splits" "tiles" [acc]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")


def run(M, N, K, ta, tb, split, tile, acc=0, NSET=8, NL=32):
    kn.gemm_switches(tile=int(str(tile)))
    sets = [(torch.randn((K, M) if ta else (M, K), device=dev), torch.randn((N, K) if tb else (K, N), device=dev),
             torch.zeros(M, N, device=dev)) for _ in range(NSET)]

    def body():
        for i in range(NL):
            A, B, C = sets[i % NSET]
            kn.gemm(A, B, C, trans_a=ta, trans_b=tb, split_k=split, accumulate=acc)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 / NL * 1e6


if __name__ == "__main__":
    for spec in sys.argv[1:]:
        M, N, K, ta, tb, splits, tiles, acc = spec.split(":")
        M, N, K, ta, tb, acc = int(M), int(N), int(K), int(ta), int(tb), int(acc)
        for tile in tiles.split(","):
            for sp in splits.split(","):
                us = run(M, N, K, bool(ta), bool(tb), int(sp), int(tile), acc)
                print("M=%5d N=%5d K=%5d %s%s split=%2d tile=%s: %7.2f us  %6.1f TF" % (
                    M, N, K, "T" if ta else "N", "T" if tb else "N", int(sp), tile, us, 2.0 * M * N * K / us / 1e6), flush=True)
