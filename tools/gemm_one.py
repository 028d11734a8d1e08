"""Run ONE gemm shape N times (for rocprofv3 --pmc runs).  usage: gemm_one.py M N K ta tb split tile reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
M, N, K, ta, tb, split, tile, reps = [int(x) for x in sys.argv[1:9]]
os.environ["MESM_GEMM_TILE"] = str(tile)
dev = torch.device("cuda:0")
A = torch.randn((K, M) if ta else (M, K), device=dev)
B = torch.randn((N, K) if tb else (K, N), device=dev)
C = torch.zeros(M, N, device=dev)
for _ in range(reps):
    kn.gemm(A, B, C, trans_a=bool(ta), trans_b=bool(tb), split_k=split)
torch.cuda.synchronize()
