"""The two-term fp16 split of the f32 GEMM products (MESM_GEMM_BF16X=2: hi = f16(2^e x), lo = f16(2^e x - hi), three
products on v_mfma_f32_32x32x16_f16, gemm_ws.hpp `HalfFrag` / `HalfScale`) against fp64 -- above all the part fp16 cannot
take for granted: RANGE.  The scale 2^e is owned by each wave and follows the data (a stage whose fragment maximum leaves
[2^3, 2^15) re-picks it and rescales the accumulators), so the checks are operands far outside fp16's range, operands whose
magnitude changes by many orders along the reduce axis, along the rows, and per element, and non-finite operands.

Error measure: |C - ref| against  sum_k |a_ik| |b_kj|  (the bound an f32 product itself is held to), and the plain
max-error / max-|ref| figure the other GEMM tests use.  The exact-f32 MFMA kernel runs beside it on the same operands:
the split form must stay within 2x of its error (review ask) or under an absolute 4e-7 of the |a||b| sum."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def gen(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


class mode:
    def __init__(self, m):
        self.m = m

    def __enter__(self):
        from mesm_amd import kernels as kn
        self.prev = kn.gemm_mode()
        kn.gemm_switches(bf16x=self.m)

    def __exit__(self, *a):
        from mesm_amd import kernels as kn
        kn.gemm_switches(bf16x=self.prev)


def product(A, B, ta, tb, m, **kw):
    from mesm_amd import kernels as kn
    M = A.shape[1] if ta else A.shape[0]
    N = B.shape[0] if tb else B.shape[1]
    C = torch.zeros(M, N, device=dev())
    with mode(m):
        kn.gemm(A, B, C, trans_a=ta, trans_b=tb, **kw)
    torch.cuda.synchronize()
    return C


def errors(C, A, B, ta, tb):
    a = (A.t() if ta else A).double()
    b = (B.t() if tb else B).double()
    ref = a @ b
    bound = a.abs() @ b.abs()
    e = (C.double() - ref).abs()
    ok = torch.isfinite(ref) & (bound > 0)
    return float((e[ok] / bound[ok]).max()), float(e[ok].max() / ref[ok].abs().max())


SHAPES = [(4800, 256, 256, False, True), (4800, 1024, 256, False, True), (4800, 256, 1024, False, False),
          (1024, 256, 4800, True, False), (2400, 256, 2818, False, True), (1024, 5003, 256, False, True),
          (2433, 258, 262, False, False), (2433, 258, 262, True, True)]


@pytest.mark.parametrize("M,N,K,ta,tb", SHAPES)
def test_f16x3_against_fp64_and_the_exact_kernel(M, N, K, ta, tb):
    A = gen((K, M) if ta else (M, K), M + K)
    B = gen((N, K) if tb else (K, N), N + K, 0.06)
    kw = dict(split_k=4, accumulate=2) if ta else {}
    e2, m2 = errors(product(A, B, ta, tb, 2, **kw), A, B, ta, tb)
    e0, m0 = errors(product(A, B, ta, tb, 0, **kw), A, B, ta, tb)
    assert m2 < 2e-6, (m2, m0)
    assert e2 <= max(2 * e0, 4e-7), (e2, e0)


def _scaled(shape, seed, kind, axis_k, lo=-30, hi=30):
    """randn times 10^u: u per reduce index (`k`), per outer index (`o`), per element (`e`), or one `c`onstant"""
    x = gen(shape, seed).double()
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    n_k, n_o = shape[axis_k], shape[1 - axis_k]
    if kind == "k":
        # magnitudes drift along the reduce axis in blocks of 32 (a stage), so the waves must re-pick their scale
        u = torch.empty(n_k).uniform_(lo, hi, generator=g)
        u = u.view(-1)[(torch.arange(n_k) // 32) * 32 % n_k]
        s = u.view(-1, 1) if axis_k == 0 else u.view(1, -1)
    elif kind == "o":
        u = torch.empty(n_o).uniform_(lo, hi, generator=g)
        s = u.view(1, -1) if axis_k == 0 else u.view(-1, 1)
    elif kind == "e":
        s = torch.empty(shape).uniform_(lo, hi, generator=g)
    else:
        s = torch.full((1, 1), float(lo))
    return (x * (10.0 ** s.double().to(dev()))).float()


@pytest.mark.parametrize("kind_a,kind_b,lo,hi", [
    ("c", "c", -15, -15),     # both operands ~1e-15: far below fp16's smallest subnormal (6e-8); products ~1e-30
    ("c", "c", 15, 15),       # ~1e15 each: far above fp16's 65504 (product 1e30, inside f32)
    ("k", "c", -12, 12),      # A's magnitude jumps by up to 24 orders between stages
    ("c", "k", -12, 12),
    ("k", "k", -8, 8),
    ("o", "c", -15, 15),      # rows of very different magnitude in one tile
    ("o", "o", -9, 9),
    ("e", "c", -3, 3),        # every element its own magnitude (6 orders inside a fragment)
    ("e", "e", -2, 2),
])
@pytest.mark.parametrize("ta,tb", [(False, True), (True, False)])
def test_f16x3_follows_the_dynamic_range_of_f32_operands(kind_a, kind_b, lo, hi, ta, tb):
    M, N, K = (1024, 256, 4800) if ta else (4800, 256, 1024)
    A = _scaled((K, M) if ta else (M, K), 11, kind_a, 0 if ta else 1, lo, hi)
    B = _scaled((N, K) if tb else (K, N), 12, kind_b, 1 if tb else 0, lo if kind_b != "c" or kind_a == "c" else 0,
                hi if kind_b != "c" or kind_a == "c" else 0)
    kw = dict(split_k=4, accumulate=2) if ta else {}
    C2 = product(A, B, ta, tb, 2, **kw)
    C0 = product(A, B, ta, tb, 0, **kw)
    a = (A.t() if ta else A).double()
    b = (B.t() if tb else B).double()
    ref = a @ b
    # per TILE ROW the error is held against that row's own |a||b| sums when magnitudes vary by row only; with magnitudes
    # varying inside a fragment the promise is relative to the fragment maximum, i.e. to the row-block's largest sums
    bound = a.abs() @ b.abs()
    fin = torch.isfinite(ref) & torch.isfinite(bound) & (bound > 1e-33) & (bound < 1e36)
    assert torch.isfinite(C2[fin]).all()
    e2 = (C2.double() - ref).abs()
    e0 = (C0.double() - ref).abs()
    if kind_a in ("c",) and kind_b in ("c",):
        r2 = float((e2[fin] / bound[fin]).max()); r0 = float((e0[fin] / bound[fin]).max())
        assert r2 <= max(2 * r0, 4e-7), (r2, r0)
    else:
        # normwise per 64-row block of the output (what a wave's scale can promise)
        Mo = ref.shape[0] // 64 * 64
        blk = lambda t: t[:Mo].view(-1, 64, t.shape[1])
        fb = blk(fin.double())
        num2 = (blk(e2) * fb).amax(dim=(1, 2)); num0 = (blk(e0) * fb).amax(dim=(1, 2))
        den = (blk(torch.where(fin, bound, torch.zeros_like(bound)))).amax(dim=(1, 2)).clamp_min(1e-300)
        r2 = float((num2 / den).max()); r0 = float((num0 / den).max())
        assert r2 <= max(2 * r0, 4e-7), (r2, r0)


def test_f16x3_keeps_non_finite_operands_non_finite_and_the_rest_exact():
    M, N, K = 4800, 256, 256
    A = gen((M, K), 901)
    B = gen((N, K), 902, 0.1)
    A[7, 13] = float("inf")
    A[100, 200] = float("-inf")
    A[4000, 0] = float("nan")
    B[5, 77] = float("inf")
    C = product(A, B, False, True, 2)
    ref = A.double() @ B.double().t()
    bad_ref = ~torch.isfinite(ref)
    bad = ~torch.isfinite(C)
    assert torch.equal(bad, bad_ref), (int(bad.sum()), int(bad_ref.sum()))
    ok = ~bad_ref
    assert float((C.double() - ref)[ok].abs().max()) / float(ref[ok].abs().max()) < 2e-6


def test_f16x3_zero_blocks_and_denormals():
    """all-zero fragments (padded pairs) must not move the scale; f32 denormal operands give what f32 gives (~0)"""
    M, N, K = 4800, 256, 1024
    A = gen((M, K), 5)
    A[:, 256:768] = 0.0
    A[1000:3000] = 0.0
    B = gen((N, K), 6, 0.05)
    C = product(A, B, False, True, 2)
    ref = A.double() @ B.double().t()
    assert float((C.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert float(C[1000:3000].abs().max()) == 0.0
    A2 = gen((M, K), 7) * 1e-42
    C2 = product(A2, B, False, True, 2)
    assert torch.isfinite(C2).all() and float(C2.abs().max()) < 1e-38


def test_f16x3_grouped_launch_every_layout_and_epilogue():
    from mesm_amd import kernels as kn
    with mode(2):
        for ta in (False, True):
            for tb in (False, True):
                probs = []
                for (M, N, K, extra) in [(2433, 258, 262, "bias"), (1056, 256, 1030, "res"), (33, 256, 256, None),
                                         (1, 256, 70, "bias"), (320, 512, 64, None)]:
                    A = gen((K, M) if ta else (M, K), M + 3 * K)
                    B = gen((N, K) if tb else (K, N), N + 7 * K, 0.1)
                    kw = dict(trans_a=ta, trans_b=tb)
                    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
                    if extra == "bias":
                        kw["bias"] = gen((N,), 5)
                        ref = ref + kw["bias"].double()
                    elif extra == "res":
                        kw["residual"] = gen((M, N), 6)
                        ref = ref + kw["residual"].double()
                    probs.append((A, B, kw, ref, torch.zeros(M, N, device=dev())))
                if ta:
                    A = gen((4800, 256), 11); B = gen((4800, 192), 12, 0.1)
                    cs = torch.zeros(256, device=dev())
                    probs.append((A, B, dict(trans_a=True, trans_b=False, split_k=4, accumulate=2, colsum=cs),
                                  A.double().t() @ B.double(), torch.zeros(256, 192, device=dev())))
                with kn.gemm_group():
                    for A, B, kw, ref, C in probs:
                        kn.gemm(A, B, C, **kw)
                torch.cuda.synchronize()
                for A, B, kw, ref, C in probs:
                    err = ((C.double() - ref).abs().max() / ref.abs().max()).item()
                    assert err < 2e-6, (ta, tb, tuple(C.shape), A.shape, err)
