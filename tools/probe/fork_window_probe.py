"""Probe: the fixed cost of a parallel branch in a captured graph -- is it per graph, per node, or only for the
nodes between fork and join?  Main chain: NM small GEMMs; side branch: NS GEMMs forked after main node f and
joined after main node j."""
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


def timeit(gr, reps=30):
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def chain(NM, forks):
    """forks: list of (f, j, NS)"""
    def body():
        cur = torch.cuda.current_stream()
        sides = {}
        for i in range(NM):
            for (f, j, ns) in forks:
                if i == f:
                    s = torch.cuda.Stream()
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        for q in range(ns):
                            g(q, 8)
                    sides[(f, j)] = s
            g(i)
            for (f, j, ns) in forks:
                if i == j:
                    cur.wait_stream(sides[(f, j)])
    return body


for NM in (500,):
    base = timeit(capture(chain(NM, [])))
    print("main %4d nodes, no fork: %.3f ms" % (NM, base), flush=True)
    for forks in ([(0, NM - 1, 1)], [(0, NM - 1, 100)], [(NM // 2, NM // 2 + 50, 1)], [(NM // 2, NM // 2 + 50, 40)],
                  [(10, 60, 20), (NM // 2, NM // 2 + 50, 20)], [(0, 10, 1)], [(NM - 12, NM - 1, 1)]):
        t = timeit(capture(chain(NM, forks)))
        print("  forks %-40s %.3f ms  (%+.3f)" % (forks, t, t - base), flush=True)

# ---- two LINEAR graphs replayed on two streams: does each keep the fast path, and do they overlap?
print("two linear graphs on two streams:")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for NM, NS in ((500, 1), (500, 20), (500, 150), (500, 500), (250, 250)):
    g1 = capture(chain(NM, []))

    def side_body():
        for q in range(NS):
            g(q, 8)
    g2 = capture(side_body)
    t1, t2 = timeit(g1), timeit(g2)

    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            g1.replay()
        with torch.cuda.stream(s2):
            g2.replay()
        cur.wait_stream(s1); cur.wait_stream(s2)
    for _ in range(5):
        both()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        both()
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / 30 * 1e3
    print("  main %d alone %.3f ms, side %d alone %.3f ms, both concurrently %.3f ms" % (NM, t1, NS, t2, tb), flush=True)

# ---- side work as PLAIN launches (no graph) on a second stream next to the main graph
print("main graph + plain side launches on a second stream:")
g1 = capture(chain(500, []))
for NS in (1, 20, 150):
    def both2():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            g1.replay()
        with torch.cuda.stream(s2):
            for q in range(NS):
                g(q, 8)
        cur.wait_stream(s1); cur.wait_stream(s2)
    for _ in range(5):
        both2()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        both2()
    torch.cuda.synchronize()
    print("  main 500 + %d plain side launches: %.3f ms" % (NS, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
# ---- the main graph replayed on a side stream alone (no second queue busy)
def only():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    with torch.cuda.stream(s1):
        g1.replay()
    cur.wait_stream(s1)
for _ in range(5):
    only()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    only()
torch.cuda.synchronize()
print("  main 500 replayed on a non-default stream, nothing else: %.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3))
