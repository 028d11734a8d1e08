"""Probe: per-node cost of a 500-GEMM linear graph alone, next to a second busy queue, and with a fork inside,
under whatever HIP runtime knobs the environment sets (see tools/queue_probe.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn

dev = torch.device("cuda:0")
M, N, K = 320, 256, 256
A = [torch.randn(M, K, device=dev) for _ in range(8)]
W = [torch.randn(N, K, device=dev) for _ in range(8)]
C = [torch.zeros(M, N, device=dev) for _ in range(16)]


def g(i, off=0):
    kn.gemm(A[i % 8], W[i % 8], C[off + i % 8], trans_b=True)


def capture(body):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body()
    return gr


def timefn(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


NM = 500
g1 = capture(lambda: [g(i) for i in range(NM)])
res = ["alone %.3f" % timefn(g1.replay)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for NS in (1, 150):
    g2 = capture(lambda: [g(q, 8) for q in range(NS)])

    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            g1.replay()
        with torch.cuda.stream(s2):
            g2.replay()
        cur.wait_stream(s1); cur.wait_stream(s2)
    res.append("+side%d %.3f" % (NS, timefn(both)))


def forked():
    cur = torch.cuda.current_stream()
    s = torch.cuda.Stream()
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        g(0, 8)
    for i in range(NM):
        g(i)
    cur.wait_stream(s)


res.append("fork-in-graph %.3f" % timefn(capture(forked).replay))
print(os.environ.get("PROBE_TAG", "default"), " | ".join(res), flush=True)
