"""one-launch timings of small / medium GEMM shapes in a given tree: python tools/probe/launch_ab.py <tree dir> (same-device A/B of
kernel prologue changes; the tree's own libmesm_gfx950.so is loaded)"""
import os, sys, time
tree = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else ".")
sys.path.insert(0, tree)
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")


def t_us(M, N, K, tb=True, group=0):
    A = torch.randn(M, K, device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev) * 0.06
    Cs = [torch.zeros(M, N, device=dev) for _ in range(4)]
    A2 = torch.randn(320, K, device=dev); C2 = [torch.zeros(320, N, device=dev) for _ in range(4)]
    def body():
        for i in range(16):
            if group:
                with kn.gemm_group():
                    kn.gemm(A, B, Cs[i % 4], trans_b=tb)
                    kn.gemm(A2, B, C2[i % 4], trans_b=tb)
            else:
                kn.gemm(A, B, Cs[i % 4], trans_b=tb)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 160 * 1e6)
    return best


out = []
for (M, N, K, grp) in [(4096, 256, 256, 0), (4800, 256, 256, 0), (4800, 1024, 256, 0), (4096, 256, 1024, 0), (320, 256, 256, 0),
                       (320, 1024, 256, 0), (4800, 256, 256, 1), (2400, 256, 256, 1)]:
    out.append("%dx%dx%d%s %.2f" % (M, N, K, "+g" if grp else "", t_us(M, N, K, group=grp)))
print(os.path.basename(tree) or ".", " | ".join(out))
