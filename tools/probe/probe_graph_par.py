"""Probe: do parallel branches of a captured HIP graph run concurrently on gfx950 / ROCm 7.2, and
what does a fork/join pair cost?  Workload: 64 launches of the 2400x256x256 forward GEMM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn

dev = torch.device("cuda:0")
M, N, K = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (2400, 256, 256))]
NL = int(os.environ.get('NL', '64'))
NB = 64  # operand sets (rotated)
A = [torch.randn(M, K, device=dev) for _ in range(NB)]
W = [torch.randn(N, K, device=dev) for _ in range(NB)]
C = [torch.zeros(M, N, device=dev) for _ in range(NB)]


def g(i):
    kn.gemm(A[i % NB], W[i % NB], C[i % NB], trans_b=True)


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


def timeit(gr, reps=50):
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def serial():
    for i in range(NL):
        g(i)


def branches(nb):
    sides = [torch.cuda.Stream() for _ in range(nb - 1)]

    def body():
        cur = torch.cuda.current_stream()
        for s in sides:
            s.wait_stream(cur)
        per = NL // nb
        for b in range(nb):
            st = cur if b == 0 else sides[b - 1]
            with torch.cuda.stream(st):
                for i in range(b * per, (b + 1) * per):
                    g(i)
        for s in sides:
            cur.wait_stream(s)
    return body


def forkjoin_each():
    side = torch.cuda.Stream()

    def body():
        cur = torch.cuda.current_stream()
        for i in range(0, NL, 2):
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                g(i)
            g(i + 1)
            cur.wait_stream(side)
    return body


def fork_only():
    """main chain of 32, each step forks one side kernel; single join at the end (dW pattern)."""
    side = torch.cuda.Stream()

    def body():
        cur = torch.cuda.current_stream()
        for i in range(0, NL, 2):
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                g(i)
            g(i + 1)
        cur.wait_stream(side)
    return body


def long_main_short_side(frac):
    """main chain of NL launches, side chain of frac*NL forked after the first main launch and
    enqueued after the whole main chain (the masked-word-branch pattern)"""
    side = torch.cuda.Stream()
    ns = int(NL * frac)

    def body():
        cur = torch.cuda.current_stream()
        g(0)
        ev = torch.cuda.Event()
        ev.record(cur)
        for i in range(1, NL):
            g(i)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            for i in range(ns):
                g(NL + i)
        cur.wait_stream(side)
    return body


def early_event_late_enqueue():
    """main: g0, record ev, g1..g31; THEN side (waits ev): g32..g63; join.  Same dependencies as two
    branches forked after g0, but the side work is enqueued after the main work in host order."""
    side = torch.cuda.Stream()

    def body():
        cur = torch.cuda.current_stream()
        g(0)
        ev = torch.cuda.Event()
        ev.record(cur)
        for i in range(1, NL // 2):
            g(i)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            for i in range(NL // 2, NL):
                g(i)
        cur.wait_stream(side)
    return body


def interleaved_enqueue():
    """same graph, but host enqueue alternates between the two streams"""
    side = torch.cuda.Stream()

    def body():
        cur = torch.cuda.current_stream()
        g(0)
        side.wait_stream(cur)
        for i in range(1, NL // 2):
            g(i)
            with torch.cuda.stream(side):
                g(NL // 2 + i)
        cur.wait_stream(side)
    return body


def branches_with(kind):
    """2 branches whose every step also enqueues an ATen op of a given kind"""
    side = torch.cuda.Stream()
    buf = [torch.zeros(M, N, device=dev) for _ in range(2)]

    def extra(b):
        if kind == "zeros":
            torch.zeros(M, N, device=dev)
        elif kind == "zero_":
            buf[b].zero_()
        elif kind == "copy":
            buf[b].copy_(C[b])
        elif kind == "add":
            buf[b].add_(C[b])
        elif kind == "empty":
            torch.empty(M, N, device=dev)

    def body():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        for i in range(NL // 2):
            g(i); extra(0)
        with torch.cuda.stream(side):
            for i in range(NL // 2, NL):
                g(i); extra(1)
        cur.wait_stream(side)
    return body


print("shape", M, N, K)
print("serial 64           : %8.1f us" % timeit(capture(serial)))
for nb in (2, 4, 8):
    print("%d branches          : %8.1f us" % (nb, timeit(capture(branches(nb)))))
print("fork+join each pair : %8.1f us" % timeit(capture(forkjoin_each())))
print("fork each, join end : %8.1f us" % timeit(capture(fork_only())))
for kind in ("empty", "add", "zero_", "zeros", "copy"):
    print("2 branches + %-6s per step: %8.1f us" % (kind, timeit(capture(branches_with(kind)))))
print("main NL + side 0.2 NL (serial would be 1.2x): %8.1f us" % timeit(capture(long_main_short_side(0.2))))
print("early event, late enqueue: %8.1f us" % timeit(capture(early_event_late_enqueue())))
print("interleaved enqueue      : %8.1f us" % timeit(capture(interleaved_enqueue())))
