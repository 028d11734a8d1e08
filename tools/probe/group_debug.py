import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); 
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
def gen(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed); return (torch.randn(*shape, generator=g) * scale).to(dev)
def rel_err(a, b): return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
rng = random.Random(99)
slope = torch.tensor([0.25], device=dev)
for rep in range(6):
    calls = []
    for k in range(rng.choice([2, 5, 11])):
        M = rng.choice([32, 33, 320, 1024, 2400, 4800]); N = rng.choice([4, 130, 256, 512, 1024]); K = rng.choice([64, 70, 256, 320, 1024])
        ta, tb = rng.random() < 0.4, rng.random() < 0.5
        A = gen((K, M) if ta else (M, K), rng.randrange(10 ** 6))
        B = gen((N, K) if tb else (K, N), rng.randrange(10 ** 6), 0.1)
        kw = dict(trans_a=ta, trans_b=tb)
        r = rng.random()
        if r < 0.2: kw["bias"] = gen((N,), 5)
        elif r < 0.4: kw.update(residual=gen((M, N), 6), e_drop=(0.1, 9))
        elif r < 0.55: kw.update(aux=gen((M, N), 7), e_actgrad=kn.ACT_PRELU, slope=slope, dslope=torch.zeros(1, device=dev))
        elif r < 0.7 and ta: kw.update(split_k=4, accumulate=2, colsum=torch.zeros(M, device=dev))
        elif r < 0.8 and not ta: kw["A2"] = gen((M, K), 8)
        calls.append((A, B, kw))
    outs1, outs2, side1, side2, refs = [], [], [], [], []
    for A, B, kw in calls:
        kw1 = {k_: (v.clone() if k_ in ("colsum", "dslope") else v) for k_, v in kw.items()}
        M = A.shape[1] if kw["trans_a"] else A.shape[0]; N = B.shape[0] if kw["trans_b"] else B.shape[1]
        C = torch.zeros(M, N, device=dev); kn.gemm(A, B, C, **kw1)
        outs1.append(C); side1.append([kw1.get("colsum"), kw1.get("dslope")])
    with kn.gemm_group():
        for A, B, kw in calls:
            kw2 = {k_: (v.clone() if k_ in ("colsum", "dslope") else v) for k_, v in kw.items()}
            M = A.shape[1] if kw["trans_a"] else A.shape[0]; N = B.shape[0] if kw["trans_b"] else B.shape[1]
            C = torch.zeros(M, N, device=dev); kn.gemm(A, B, C, **kw2)
            outs2.append(C); side2.append([kw2.get("colsum"), kw2.get("dslope")])
    torch.cuda.synchronize()
    for i, ((A, B, kw), a, b, sa, sb) in enumerate(zip(calls, outs1, outs2, side1, side2)):
        M, N = a.shape; K = A.shape[0] if kw["trans_a"] else A.shape[1]
        e = rel_err(b, a)
        es = [rel_err(y, x) for x, y in zip(sa, sb) if x is not None]
        flag = "  <<<<" if e >= 1e-5 or any(v >= 1e-4 for v in es) else ""
        print("rep %d #%d %dx%dx%d %s%s %s  C err %.2e side %s%s" % (rep, i, M, N, K, "T" if kw["trans_a"] else "N", "T" if kw["trans_b"] else "N",
              [k for k in kw if k not in ("trans_a", "trans_b")], e, ["%.2e" % v for v in es], flag))
