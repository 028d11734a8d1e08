"""Forced-kernel check of the tall-tile GEMM (MESM_GEMM_TILE=5 / 6) against the default dispatch on the same
arguments, then timings of the step's tall shapes.  usage: wtall_check.py [check] [time]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
kn._SPLIT_ROWS = False


def one(M, N, K, ta, tb, tile, **kw):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g).to(dev)
    B = torch.randn((N, K) if tb else (K, N), generator=g).to(dev)
    extra = {}
    if kw.get("bias"): extra["bias"] = torch.randn(N, generator=g).to(dev)
    if kw.get("residual"): extra["residual"] = torch.randn(M, N, generator=g).to(dev)
    if kw.get("e_act"): extra["e_act"] = kw["e_act"]
    if kw.get("e_drop"): extra["e_drop"] = (0.1, 1234)
    if kw.get("a_act"): extra["a_act"] = kw["a_act"]
    if kw.get("pre_out"): pass
    slope = torch.tensor([0.25], device=dev)
    if kw.get("actgrad"):
        extra.update(aux=torch.randn(M, N, generator=g).to(dev), e_actgrad=kw["actgrad"], slope=slope)
    outs = []
    for t in (0, tile):
        os.environ["MESM_GEMM_TILE"] = str(t)
        C = torch.full((M, N), 0.5, device=dev)
        ex = dict(extra)
        ds = None
        if kw.get("actgrad") == kn.ACT_PRELU:
            ds = torch.zeros(1, device=dev); ex["dslope"] = ds
        pre = None
        if kw.get("pre_out"):
            pre = torch.zeros(M, N, device=dev); ex["pre_out"] = pre
        kn.gemm(A, B, C, trans_a=ta, trans_b=tb, accumulate=kw.get("acc", 0), **ex)
        torch.cuda.synchronize()
        outs.append((C, ds, pre))
    os.environ["MESM_GEMM_TILE"] = "0"
    ref = outs[0][0]
    err = float((outs[1][0] - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
    msg = "M=%d N=%d K=%d %s%s tile=%d %s: rel %.2e" % (M, N, K, "T" if ta else "N", "T" if tb else "N", tile, kw, err)
    if outs[0][1] is not None:
        de = abs(float(outs[1][1]) - float(outs[0][1])) / max(abs(float(outs[0][1])), 1e-6)
        msg += " dslope rel %.2e" % de
        assert de < 1e-4, msg
    if outs[0][2] is not None:
        pe = float((outs[1][2] - outs[0][2]).abs().max()) / max(float(outs[0][2].abs().max()), 1e-6)
        msg += " pre rel %.2e" % pe
        assert pe < 1e-5, msg
    print(msg, flush=True)
    assert err < 1e-5, msg


def check():
    for tile in (5, 6):
        for (M, N, K) in [(4800, 256, 256), (2400, 256, 512), (4864, 256, 1024), (163, 45, 70), (100, 33, 37),
                          (321, 96, 129), (160, 32, 32), (96, 32, 4)]:
            for ta, tb in [(0, 1), (0, 0), (1, 0), (1, 1)]:
                one(M, N, K, bool(ta), bool(tb), tile)
        one(4800, 256, 256, False, True, tile, bias=True, residual=True, e_act=kn.ACT_RELU)
        one(4800, 256, 256, False, True, tile, bias=True, e_drop=True, pre_out=True, e_act=kn.ACT_RELU)
        one(2400, 256, 512, False, False, tile, actgrad=kn.ACT_RELU)
        one(2400, 256, 512, False, False, tile, actgrad=kn.ACT_PRELU, residual=True)
        one(1203, 130, 515, False, True, tile, bias=True, residual=True, acc=1, a_act=kn.ACT_RELU)
    print("wtall check ok")


def times():
    from gemm_sweep import run
    for (M, N, K, ta, tb) in [(4800, 256, 256, 0, 1), (4800, 256, 256, 0, 0), (4864, 256, 256, 0, 1),
                              (4800, 256, 1024, 0, 1), (4800, 256, 1024, 0, 0), (4864, 256, 512, 0, 1),
                              (2400, 256, 256, 0, 1), (2400, 256, 512, 0, 1), (2400, 256, 512, 0, 0),
                              (2400, 256, 2818, 0, 1), (4800, 512, 256, 0, 1), (2400, 512, 256, 0, 1),
                              (4800, 1024, 256, 0, 1)]:
        for tile in (0, 5, 6):
            kn._SPLIT_ROWS = tile == 0
            us = run(M, N, K, bool(ta), bool(tb), 1, tile)
            print("M=%5d N=%5d K=%5d %s%s tile=%s: %7.2f us  %6.1f TF" % (
                M, N, K, "T" if ta else "N", "T" if tb else "N", tile, us, 2.0 * M * N * K / us / 1e6), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    if "check" in what: check()
    if "time" in what: times()
