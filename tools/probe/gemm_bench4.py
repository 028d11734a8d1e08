"""GEMM timing the way the step runs it: a captured HIP graph of 32 launches of one shape over 8
rotating operand sets (so operands are not L2-hot from the previous launch), replayed 20 x.
Columns: auto dispatch | frag kernel | staged 32 / 64 / 128 tiles.   us per launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
SHAPES = [
    ("FFN1 4800     ", 4800, 1024, 256, False, True, 1),
    ("FFN2 4800     ", 4800, 256, 1024, False, True, 1),
    ("dz1 4800      ", 4800, 1024, 256, False, False, 1),
    ("dx 4800 F->d  ", 4800, 256, 1024, False, False, 1),
    ("dW Fxd 4800 s4", 1024, 256, 4800, True, False, 4),
    ("dW dxF 4800 s4", 256, 1024, 4800, True, False, 4),
    ("fwd d->d      ", 2400, 256, 256, False, True, 1),
    ("fwd 2N d->d   ", 4800, 256, 256, False, True, 1),
    ("fwd d->F      ", 2400, 1024, 256, False, True, 1),
    ("fwd F->d      ", 2400, 256, 1024, False, True, 1),
    ("fwd words d->d", 1024, 256, 256, False, True, 1),
    ("fwd dec d->d  ", 320, 256, 256, False, True, 1),
    ("fwd tiny      ", 32, 256, 256, False, True, 1),
    ("fwd Dv->d     ", 2400, 256, 2818, False, True, 1),
    ("fwd MLM head  ", 1024, 5003, 256, False, True, 1),
    ("dX d<-d       ", 2400, 256, 256, False, False, 1),
    ("dX d<-F       ", 2400, 256, 1024, False, False, 1),
    ("dX F<-d       ", 2400, 1024, 256, False, False, 1),
    ("dW dxd s16    ", 256, 256, 2400, True, False, 16),
    ("dW dxd s4     ", 256, 256, 2400, True, False, 4),
    ("dW Fxd s4     ", 1024, 256, 2400, True, False, 4),
    ("dW dxF s4     ", 256, 1024, 2400, True, False, 4),
    ("dW dxDv s1    ", 256, 2818, 2400, True, False, 1),
]
only = sys.argv[1:] 
NSET, NL = 8, 32
for name, M, N, K, ta, tb, split in SHAPES:
    if only and not any(o in name for o in only):
        continue
    sets = [(torch.randn((K, M) if ta else (M, K), device=dev), torch.randn((N, K) if tb else (K, N), device=dev),
             torch.zeros(M, N, device=dev)) for _ in range(NSET)]
    res = []
    for tile in ("0", "2", "4", "3", "32"):
        kn.gemm_switches(tile=int(tile))
        def body():
            for i in range(NL):
                A, B, C = sets[i % NSET]
                kn.gemm(A, B, C, trans_a=ta, trans_b=tb, split_k=split)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        REPS = int(os.environ.get("REPS", "20"))
        t0 = time.perf_counter()
        for _ in range(REPS):
            g.replay()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / REPS / NL * 1e6)
    kn.gemm_switches(tile=int("0"))
    print("%s M=%5d N=%5d K=%5d s%-2d auto %7.2f | wstage %7.2f | wstage64 %7.2f | lds64 %7.2f | t32 %7.2f us  (best %5.1f TF)" % (
        name, M, N, K, split, res[0], res[1], res[2], res[3], res[4], 2.0 * M * N * K / min(res) / 1e6), flush=True)
