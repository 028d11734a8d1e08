"""which problem of tests/test_kernels_gpu.py::test_gemm_group_matches_individual_launches differs under the persistent launch"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from mesm_amd import kernels as kn
from test_kernels_gpu import gen, dev, rel_err  # (needs a build with gemm_pk.hip linked in and its entry points bound)
rng = random.Random(99)
slope = torch.tensor([0.25], device=dev())
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
        elif r < 0.55: kw.update(aux=gen((M, N), 7), e_actgrad=kn.ACT_PRELU, slope=slope, dslope=torch.zeros(1, device=dev()))
        elif r < 0.7 and ta: kw.update(split_k=4, accumulate=2, colsum=torch.zeros(M, device=dev()))
        elif r < 0.8 and not ta: kw["A2"] = gen((M, K), 8)
        calls.append((A, B, kw))
    for mode in ((0, 512, 8, 3),) + ((1, 512, 8, 3), (1, 512, 100, 3), (1, 512, 1, 0), (1, 64, 1, 3)) * 6:
        kn.gemm_pk(on=mode[0], grid=mode[1], cut_min=mode[2], c0=mode[3])
        outs = []
        with kn.gemm_group():
            for A, B, kw in calls:
                kw2 = {k_: (v.clone() if k_ in ("colsum", "dslope") else v) for k_, v in kw.items()}
                M = A.shape[1] if kw["trans_a"] else A.shape[0]
                N = B.shape[0] if kw["trans_b"] else B.shape[1]
                C = torch.zeros(M, N, device=dev())
                kn.gemm(A, B, C, **kw2)
                outs.append((C, kw2))
        torch.cuda.synchronize()
        st = kn.gemm_pk_status()
        if mode[0] == 0:
            base = outs
            continue
        for i, ((C, kw2), (C0, kw0)) in enumerate(zip(outs, base)):
            e = rel_err(C, C0)
            extra = ""
            for nm in ("colsum", "dslope"):
                if kw2.get(nm) is not None:
                    extra += " %s %.2e" % (nm, rel_err(kw2[nm], kw0[nm]))
            if e > 1e-5 or "e-0" in extra and False:
                A, B, kw = calls[i]
                print("rep %d mode %s problem %d: C %s A %s keys %s err %.3e%s status %d" % (rep, mode, i, tuple(C.shape), tuple(A.shape), sorted(kw), e, extra, st))
                bad = ((C - C0).abs() > 1e-3 * C0.abs().max()).nonzero()
                print("   bad elements %d, first %s last %s; rows %s" % (len(bad), bad[0].tolist(), bad[-1].tolist(), sorted(set((bad[:, 0] // 64).tolist()))[:20]))
    print("rep %d done (status %d)" % (rep, kn.gemm_pk_status()))
