"""A/B of the two matrix-instruction shapes of the split-bf16 k-split 64 x 64 kernel (single-problem launches):
v_mfma_f32_32x32x16_bf16 (production) vs v_mfma_f32_16x16x32_bf16 (MESM_GEMM_MF16 / mesm_gemm_set_pipe): error against
fp64 on every layout pair (with tails, split-K + column sums, epilogues) and time per launch on the step's large shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import kernels as kn
from mesm_amd._lib import lib
dev = torch.device("cuda:0")


def gen(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def run(M, N, K, ta, tb, mf, **kw):
    lib().mesm_gemm_set_pipe(mf)
    A = gen((K, M) if ta else (M, K), 1 + M + K)
    B = gen((N, K) if tb else (K, N), 2 + N + K, 0.1)
    C = torch.zeros(M, N, device=dev)
    kn.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
    torch.cuda.synchronize()
    return A, B, C


print("== correctness (max |err| / max |ref| against fp64), phased | pipelined")
for (M, N, K) in [(4800, 256, 256), (2433, 258, 262), (4100, 130, 1030)]:
    for ta in (False, True):
        for tb in (False, True):
            errs = []
            for mf in (0, 1):
                bias, res = gen((N,), 5), gen((M, N), 6)
                A, B, C = run(M, N, K, ta, tb, mf, bias=bias, residual=res, e_act=kn.ACT_RELU)
                ref = torch.relu((A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double()) + bias.double()) + res.double()
                errs.append(float((C.double() - ref).abs().max() / ref.abs().max()))
            print("%5dx%4dx%4d ta%d tb%d  %.2e | %.2e" % (M, N, K, ta, tb, errs[0], errs[1]))
for mf in (0, 1):
    cs = torch.zeros(256, device=dev)
    A, B, C = run(256, 1024, 4864, True, False, mf, split_k=4, accumulate=2, colsum=cs)
    ref = A.double().t() @ B.double()
    print("split-K 256x1024x4864 TN/s4 mf%d: C %.2e colsum %.2e" % (mf, float((C.double() - ref).abs().max() / ref.abs().max()),
                                                                   float((cs.double() - A.double().sum(0)).abs().max() / A.double().sum(0).abs().max())))

print("== time per launch (us), 7 interleaved rounds of 400 launches back to back, phased | pipelined")
shapes = [(8192, 256, 1024, False, True, {}), (4096, 256, 1024, False, True, {}), (4800, 256, 256, False, True, {}),
          (4800, 1024, 256, False, True, {}), (4800, 256, 1024, False, True, {}), (2400, 256, 2818, False, True, {}),
          (4800, 256, 1024, False, False, {}), (256, 1024, 4864, True, False, dict(split_k=4, accumulate=2)),
          (1024, 256, 4864, True, False, dict(split_k=4, accumulate=2))]
for (M, N, K, ta, tb, kw) in shapes:
    A = gen((K, M) if ta else (M, K), 11)
    B = gen((N, K) if tb else (K, N), 12, 0.1)
    C = torch.zeros(M, N, device=dev)
    ts = {0: [], 1: []}
    for rnd in range(7):
        for mf in (0, 1):
            lib().mesm_gemm_set_pipe(mf)
            for _ in range(30):
                kn.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(400):
                kn.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
            e1.record(); torch.cuda.synchronize()
            ts[mf].append(e0.elapsed_time(e1) / 400 * 1e3)
    gf = 2.0 * M * N * K / 1e9
    m0, m1 = sorted(ts[0])[3], sorted(ts[1])[3]
    print("%5dx%4dx%4d %s%s%s  median %7.2f | %7.2f  (min %6.2f | %6.2f)  %.0f | %.0f TF  ratio %.3f" % (
        M, N, K, "T" if ta else "N", "T" if tb else "N", "/s4" if kw else "", m0, m1, min(ts[0]), min(ts[1]), gf / m0 * 1e3, gf / m1 * 1e3, m0 / m1))
lib().mesm_gemm_set_pipe(0)
