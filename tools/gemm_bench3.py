"""Device-bound GEMM timing: record launches on the library's tape, replay them from C++ back to
back with an event pair per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
SHAPES = [
    ("fwd d->d      ", 2400, 256, 256, False, True, 1),
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
    ("dW Fxd s4     ", 1024, 256, 2400, True, False, 4),
    ("dW dxF s4     ", 256, 1024, 2400, True, False, 4),
    ("dW dxDv s1    ", 256, 2818, 2400, True, False, 1),
]
keep = []
for name, M, N, K, ta, tb, split in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.zeros(M, N, device=dev)
    keep.append((A, B, C))
    res = []
    for tile in ("0", "32", "64", "128"):
        os.environ["MESM_GEMM_TILE"] = tile
        kn.gemm_tape(True)
        kn.gemm(A, B, C, trans_a=ta, trans_b=tb, split_k=split)
        kn.gemm_tape(False)
        kn.gemm_tape_replay(5)
        r = kn.gemm_tape_replay(100)
        res.append(r["ms"] / r["launches"] * 1e3)
    os.environ["MESM_GEMM_TILE"] = "0"
    print("%s M=%5d N=%5d K=%5d  auto %7.2f | t32 %7.2f | t64 %7.2f | t128 %7.2f us  (best %5.1f TF)" % (
        name, M, N, K, res[0], res[1], res[2], res[3], 2.0 * M * N * K / min(res) / 1e6), flush=True)
# all shapes interleaved (code of different instantiations alternates)
kn.gemm_tape(True)
for (name, M, N, K, ta, tb, split), (A, B, C) in zip(SHAPES, keep):
    kn.gemm(A, B, C, trans_a=ta, trans_b=tb, split_k=split)
kn.gemm_tape(False)
kn.gemm_tape_replay(2)
r = kn.gemm_tape_replay(20)
print("interleaved: avg %.2f us per launch" % (r["ms"] / r["launches"] * 1e3))
# wall-clock of 100 identical launches without events (C++ replay has events; use torch events around it)
