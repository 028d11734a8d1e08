import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
def run(M, N, K, **extra):
    NS, NL = 8, 16
    sets = []
    for _ in range(NS):
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); C = torch.zeros(M, N, device=dev)
        kw = dict(extra)
        if kw.pop("aux", False): kw["aux"] = torch.randn(M, N, device=dev)
        if kw.pop("dsl", False): kw["dslope"] = torch.zeros(1, device=dev)
        if "e_actgrad" in kw: kw["slope"] = torch.full((1,), 0.25, device=dev)
        sets.append((A, B, C, kw))
    def body():
        for i in range(NL):
            A, B, C, kw = sets[i % NS]; kn.gemm(A, B, C, **kw)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body()
    for _ in range(2): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 / NL * 1e6
for M in (4800, 1024):
    print(M, "plain %.1f | aux+actgrad no dslope %.1f | with dslope %.1f | +e_drop %.1f | e_drop only %.1f" % (
        run(M, 1024, 256), run(M, 1024, 256, aux=True, e_actgrad=kn.ACT_PRELU),
        run(M, 1024, 256, aux=True, e_actgrad=kn.ACT_PRELU, dsl=True),
        run(M, 1024, 256, aux=True, e_actgrad=kn.ACT_PRELU, dsl=True, e_drop=(0.1, 3)),
        run(M, 1024, 256, e_drop=(0.1, 3))))
