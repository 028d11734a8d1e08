"""Accuracy and speed of the split-bf16 PLANE GEMM (mesm_gemm_px: operands as hi / mid / lo bf16 planes written once by
mesm_split_planes, six products on v_mfma_f32_32x32x16_bf16, f32 accumulate) against the exact-f32 dispatch, on the
shapes of the step's census (every layout pair, split-K weight gradients, the unaligned 2818 / 5003 wide ones):
relative error against an fp64 product, time per launch in a captured chain of 16, and the split kernel's own time.
usage: python tools/px_check.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
# (M, N, K, trans_a, trans_b, split_k)
SHAPES = [(4800, 1024, 256, False, True, 1), (4800, 1024, 256, False, False, 1), (4800, 256, 1024, False, True, 1),
          (4800, 256, 1024, False, False, 1), (4800, 256, 256, False, True, 1), (4800, 256, 256, False, False, 1),
          (1024, 256, 4800, True, False, 4), (256, 1024, 4800, True, False, 4), (256, 256, 4800, True, False, 4),
          (2400, 256, 2818, False, True, 1), (2400, 2818, 256, False, False, 1), (256, 2818, 2400, True, False, 4),
          (1024, 5003, 256, False, True, 1), (1024, 256, 5003, False, False, 1), (5003, 256, 1024, True, False, 2),
          (2400, 512, 256, False, True, 1), (512, 256, 2400, True, False, 2), (4864, 256, 512, False, False, 1),
          (100, 70, 45, False, True, 1), (33, 129, 257, True, False, 2), (65, 64, 31, False, False, 1)]


def timed(body, n_inner):
    """median over 20 replays of a captured chain of n_inner launches (event-timed per replay), us per launch"""
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): body()
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n_inner)
    ts.sort()
    timed.spread = (ts[0], ts[-1])
    return ts[len(ts) // 2]


def run(M, N, K, ta, tb, split, px, tile=0):
    from mesm_amd._lib import lib
    lib().mesm_gemm_px_set_tile(tile)
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g).to(dev)
    B = (torch.randn((N, K) if tb else (K, N), generator=g) * 0.06).to(dev)
    C = torch.zeros(M, N, device=dev)
    kw = dict(trans_a=ta, trans_b=tb, split_k=split, accumulate=2 if split > 1 else 0)
    if px:
        pa, pb = kn.split_planes(A), kn.split_planes(B)
        assert torch.equal(pa.float(), A) and torch.equal(pb.float(), B), "hi + mid + lo must reproduce the operand exactly"
        kw.update(a_planes=pa, b_planes=pb)
    kn.gemm(A, B, C, **kw)
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
    rms = ((C.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    sets = []
    for _ in range(4):
        a, b, c = torch.randn_like(A), torch.randn_like(B), torch.zeros_like(C)
        k2 = dict(kw)
        if px:
            k2.update(a_planes=kn.split_planes(a), b_planes=kn.split_planes(b))
        sets.append((a, b, c, k2))

    def body():
        for i in range(16):
            a, b, c, k2 = sets[i % 4]
            kn.gemm(a, b, c, **k2)
    us = timed(body, 16)
    us_split = None
    if px:
        out = kn.PlaneSet(A.shape[0], A.shape[1], dev)

        def sbody():
            for i in range(16):
                kn.split_planes(sets[i % 4][0], out)
        us_split = timed(sbody, 16)
    return err, rms, us, us_split


if __name__ == "__main__":
    shapes = SHAPES[:4] + SHAPES[-3:] if len(sys.argv) > 1 and sys.argv[1] == "quick" else SHAPES
    worst = 0.0
    for M, N, K, ta, tb, split in shapes:
        e0, r0, t0, _ = run(M, N, K, ta, tb, split, False)
        e1, r1, t1, ts = run(M, N, K, ta, tb, split, True, 64)
        e2, r2, t2, _ = run(M, N, K, ta, tb, split, True, 96) if not ta else (e1, r1, t1, None)
        worst = max(worst, e1 / max(e0, 1e-9), e2 / max(e0, 1e-9))
        print("M=%5d N=%5d K=%5d %s%s/s%d  f32 err %.2e %7.2f us %6.1f TF | px64 err %.2e %7.2f us %6.1f TF (x%.2f) | px96 err %.2e %7.2f us %6.1f TF (x%.2f)  split(A) %.2f us" % (
            M, N, K, "T" if ta else "N", "T" if tb else "N", split, e0, t0, 2.0 * M * N * K / t0 / 1e6,
            e1, t1, 2.0 * M * N * K / t1 / 1e6, t0 / t1, e2, t2, 2.0 * M * N * K / t2 / 1e6, t0 / t2, ts), flush=True)
    from mesm_amd._lib import lib
    lib().mesm_gemm_px_set_tile(0)
    print("worst px / f32 max-error ratio: %.2f" % worst)
