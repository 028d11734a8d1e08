"""LayerNorm without a launch of its own (round-5 review, item 3): the PRODUCING GEMM's epilogue accumulates the row statistics
of its output, the CONSUMING GEMM runs on the raw tensor with weights pre-scaled by gamma and normalises in its epilogue

    LN(x) W^T + b = rstd_r (x W'^T - mu_r c) + d,    W' = W diag(gamma),  c = W gamma,  d = W beta + b

against GEMM -> LayerNorm launch -> GEMM, on the step's two sites: out-projection (K = 256) -> norm -> FFN1 (N = 1024) and
FFN2 (K = 1024) -> norm -> the next layer's projection (N = 256), 4800 rows.  The probe FAVOURS the fused form: it does not
write the normalised tensor at all (the real step needs it again as the residual of the next block and in the backward), the
statistics buffer is zeroed outside the timed chain (the step's fill launch would do it), and only the forward is built.

Needs the probe build of the library:
    tools/build_variant.sh lnprobe -DMESM_LN_PROBE && MESM_LIB_PATH=mesm_amd/variants/libmesm_lnprobe.so python tools/probe/ln_stats.py
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn
from mesm_amd._lib import GemmArgs, LAYOUT_REDUCE_CONTIG, check, lib, stream_ptr

dev = torch.device("cuda:0")
kn.gemm_switches(bf16x=2)


def raw_gemm(A, W, C, bias=None, residual=None, mode=0, stats=None, cvec=None, D=0):
    """C = A W^T (+ bias) (+ residual) through mesm_gemm_f32; mode 1: + row statistics of C into stats; mode 2: the normalising
    epilogue with stats / cvec / bias = d"""
    g = GemmArgs()
    M, K = A.shape
    N = W.shape[0]
    g.A, g.B, g.C = A.data_ptr(), W.data_ptr(), C.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.a_layout = g.b_layout = LAYOUT_REDUCE_CONTIG
    g.lda, g.ldb, g.ldc = A.stride(0), W.stride(0), C.stride(0)
    if bias is not None:
        g.bias = bias.data_ptr()
    if residual is not None:
        g.residual, g.ldr = residual.data_ptr(), residual.stride(0)
    g.out_scale = 1.0
    g.reserved0 = mode
    if mode:
        g.dslope_ws = stats.data_ptr()
    if mode == 2:
        g.aux, g.ldaux = cvec.data_ptr(), D
    check(lib().mesm_gemm_f32(ctypes.byref(g), stream_ptr()), "mesm_gemm_f32")
    return C


def graph_time(body, reps=16):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            body()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (10 * reps) * 1e6)
    return best


def site(rows, K1, D, N2, name):
    gen = torch.Generator().manual_seed(rows + K1 + N2)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(dev)
    A1, W1, b1, res = r(rows, K1), r(D, K1, sc=0.06), r(D, sc=0.1), r(rows, D)
    gamma, beta = 1.0 + r(D, sc=0.1), r(D, sc=0.1)
    W2, b2 = r(N2, D, sc=0.06), r(N2, sc=0.1)
    x, y = torch.empty(rows, D, device=dev), torch.empty(rows, N2, device=dev)
    x2, y2 = torch.empty(rows, D, device=dev), torch.empty(rows, N2, device=dev)
    W2p = (W2 * gamma[None, :]).contiguous()
    cvec = (W2 @ gamma).contiguous()
    dvec = (W2 @ beta + b2).contiguous()
    stats = torch.zeros(rows, 2, device=dev)

    def base():
        raw_gemm(A1, W1, x, bias=b1, residual=res)
        xn = kn.layernorm_fwd(x, gamma, beta)[0]
        raw_gemm(xn, W2, y, bias=b2)

    def fused():
        raw_gemm(A1, W1, x2, bias=b1, residual=res, mode=1, stats=stats)
        raw_gemm(x2, W2p, y2, bias=dvec, mode=2, stats=stats, cvec=cvec, D=D)

    base(); stats.zero_(); fused(); torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm((A1.double() @ W1.double().t() + b1.double() + res.double()), (D,), gamma.double(),
                                         beta.double()) @ W2.double().t() + b2.double()
    e_base = float((y.double() - ref).abs().max() / ref.abs().max())
    e_fused = float((y2.double() - ref).abs().max() / ref.abs().max())

    def prod_plain(): raw_gemm(A1, W1, x, bias=b1, residual=res)
    def prod_stats(): raw_gemm(A1, W1, x2, bias=b1, residual=res, mode=1, stats=stats)   # (stats keep growing: timing only)
    def ln_only(): kn.layernorm_fwd(x, gamma, beta)
    def cons_plain(): raw_gemm(x, W2, y, bias=b2)
    def cons_norm(): raw_gemm(x2, W2p, y2, bias=dvec, mode=2, stats=stats, cvec=cvec, D=D)
    t = {k: graph_time(f) for k, f in (("base", base), ("fused", fused), ("producer", prod_plain), ("producer+stats", prod_stats),
                                       ("LayerNorm", ln_only), ("consumer", cons_plain), ("consumer+norm", cons_norm))}
    print("%s  rows %d: GEMM(K=%d) -> LN(%d) -> GEMM(N=%d)" % (name, rows, K1, D, N2))
    print("   chain:  GEMM + LN + GEMM %.2f us   fused pair %.2f us   (%+.2f us)      max err vs fp64: %.1e | %.1e"
          % (t["base"], t["fused"], t["fused"] - t["base"], e_base, e_fused))
    print("   alone:  producer %.2f -> with row statistics %.2f (%+.2f);  LayerNorm launch %.2f;  consumer %.2f -> normalising %.2f (%+.2f)"
          % (t["producer"], t["producer+stats"], t["producer+stats"] - t["producer"], t["LayerNorm"], t["consumer"],
             t["consumer+norm"], t["consumer+norm"] - t["consumer"]), flush=True)


site(4800, 256, 256, 1024, "out-projection -> norm -> FFN1")
site(4800, 1024, 256, 256, "FFN2 -> norm -> next projection")
site(2400, 256, 256, 1024, "out-projection -> norm -> FFN1")
site(320, 256, 256, 1024, "decoder: out-projection -> norm -> FFN1")
