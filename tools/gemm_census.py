"""Census of the GEMM launches of one training step (shape, layouts, fusions), each config timed
alone in a captured graph chain (8 rotating operand sets), and the projected total."""
import collections, os, sys, time
os.environ.setdefault("MESM_AUTOGRAPH", "0")  # this tool looks at the EAGER step (autograph.py would replay graphs behind these calls)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic
from mesm_amd import kernels as kn

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "C3a"
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)
census = collections.Counter()
orig = kn.gemm


def spy(A, B, C, **kw):
    ta, tb = kw.get("trans_a", False), kw.get("trans_b", False)
    M = A.shape[1] if ta else A.shape[0]
    K = A.shape[0] if ta else A.shape[1]
    N = B.shape[0] if tb else B.shape[1]
    flags = []
    for k in ("A2", "B2", "bias", "residual", "aux", "colsum"):
        if kw.get(k) is not None:
            flags.append(k)
    for k in ("a_act", "b_act", "e_act", "e_actgrad"):
        if kw.get(k, 0):
            flags.append(k)
    for k in ("a_drop", "b_drop", "e_drop"):
        if kw.get(k, (0, 0))[0] > 0:
            flags.append(k)
    census[(M, N, K, ta, tb, kw.get("split_k", 1), kw.get("accumulate", 0), tuple(flags))] += 1
    return orig(A, B, C, **kw)


kn.gemm = spy
import mesm_amd.ops as ops_mod
out = model(**batch, dataset_name=args.dataset_name, is_training=True)
losses, total = crit(out, batch, True)
total.backward()
torch.cuda.synchronize()
kn.gemm = orig


def time_cfg(M, N, K, ta, tb, split, acc, flags):
    NS, NL = 8, 16
    sets = []
    for _ in range(NS):
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        B = torch.randn((N, K) if tb else (K, N), device=dev)
        C = torch.zeros(M, N, device=dev)
        kw = dict(trans_a=ta, trans_b=tb, split_k=split, accumulate=acc)
        if "A2" in flags: kw["A2"] = torch.randn_like(A)
        if "B2" in flags: kw["B2"] = torch.randn_like(B)
        if "bias" in flags: kw["bias"] = torch.randn(N, device=dev)
        if "residual" in flags: kw["residual"] = torch.randn(M, N, device=dev)
        if "aux" in flags: kw["aux"] = torch.randn(M, N, device=dev)
        if "colsum" in flags: kw["colsum"] = torch.zeros(M, device=dev)
        sl = torch.full((1,), 0.25, device=dev)
        if "a_act" in flags: kw.update(a_act=kn.ACT_PRELU, slope=sl)
        if "b_act" in flags: kw.update(b_act=kn.ACT_PRELU, slope=sl)
        if "e_act" in flags: kw.update(e_act=kn.ACT_RELU)
        if "e_actgrad" in flags: kw.update(e_actgrad=kn.ACT_PRELU, slope=sl, dslope=torch.zeros(1, device=dev))
        if "a_drop" in flags: kw["a_drop"] = (0.1, 7)
        if "b_drop" in flags: kw["b_drop"] = (0.1, 7)
        if "e_drop" in flags: kw["e_drop"] = (0.1, 7)
        sets.append((A, B, C, kw))

    def body():
        for i in range(NL):
            A, B, C, kw = sets[i % NS]
            kn.gemm(A, B, C, **kw)
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


rows = []
for cfg, n in census.items():
    us = time_cfg(*cfg)
    M, N, K = cfg[:3]
    rows.append((n * us, n, us, cfg))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows); nl = sum(r[1] for r in rows); fl = sum(2.0 * r[3][0] * r[3][1] * r[3][2] * r[1] for r in rows)
print("GEMM launches/step %d, projected %.3f ms, %.1f GFLOP -> %.1f TFLOP/s" % (nl, tot / 1e3, fl / 1e9, fl / tot / 1e6))
for t, n, us, (M, N, K, ta, tb, split, acc, flags) in rows[:45]:
    print("%7.1f us = %3d x %6.2f us  M=%5d N=%5d K=%5d %s%s s%-2d acc%d %5.1f TF  %s" % (
        t, n, us, M, N, K, "T" if ta else "N", "T" if tb else "N", split, acc, 2.0 * M * N * K / us / 1e6, ",".join(flags)))
