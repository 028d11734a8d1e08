import os, sys
sys.path.insert(0, "/root/repo")
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rnd = lambda *s: (torch.randn(*s, generator=g)).to(dev)
def run(force, fn, reps=50):
    if force: kn.gemm_switches(tile=int(str(force)))
    else: kn.gemm_switches(tile=0)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    kn.gemm_switches(tile=0)
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K, tb) in ((4800, 256, 256, True), (4800, 256, 256, False), (2400, 512, 256, True), (4864, 512, 256, True), (2400, 256, 256, True),
                      (2048, 512, 256, True), (4800, 512, 256, True), (1024, 1024, 256, True), (1024, 256, 1024, True), (2400, 256, 512, False)):
    x = rnd(M, K); W = rnd(N, K) if tb else rnd(K, N); b = rnd(N); C = torch.empty(M, N, device=dev)
    fn = lambda: kn.gemm(x, W, C, trans_b=tb, bias=b)
    ts = {f: run(f, fn) for f in (0, 2, 3, 4)}
    fl = 2.0 * M * N * K
    print("%5d x %4d x %4d %s  " % (M, N, K, "NT" if tb else "NN") + "  ".join("%s %6.2f us (%4.1f TF)" % ({0: "auto", 2: "k32", 3: "ring64", 4: "k64"}[f], t, fl / t / 1e6) for f, t in ts.items()))
