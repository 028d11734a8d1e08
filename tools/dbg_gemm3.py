import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
M, N, K = 32, 64, 64
A = (torch.arange(K, device=dev)[:, None] * 100 + torch.arange(M, device=dev)[None, :]).float()  # A[k][m] = 100k+m
B = torch.eye(K, N, device=dev)
C = torch.zeros(M, N, device=dev)
kn.gemm(A, B, C, trans_a=True)          # C[m][n] = A[n][m] = 100 n + m
torch.set_printoptions(linewidth=200)
print("C[24:32, 52:64] (expect 100*n + m):")
print(C[24:32, 52:64].long())
# now B carries the pattern, A = identity  -> tests the B operand path:  C[m][n] = B[m][n]
M, N, K = 64, 32, 64
A = torch.eye(K, M, device=dev)
B = (torch.arange(K, device=dev)[:, None] * 100 + torch.arange(N, device=dev)[None, :]).float()
C = torch.zeros(M, N, device=dev)
kn.gemm(A, B, C, trans_a=True)
print("C[52:64, 24:32] (expect 100*m + n):")
print(C[52:64, 24:32].long())
