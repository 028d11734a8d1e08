import sys, torch
sys.path.insert(0, "/root/repo")
from mesm_amd import kernels as kn
from mesm_amd._lib import lib
dev = torch.device("cuda:0")
lib().mesm_gemm_px_set_ring(int(sys.argv[1]))
for (M, N, K, ta, tb) in [(64, 64, 32, False, True), (64, 64, 64, False, True), (64, 64, 96, False, True), (64,64,128,False,True), (128, 128, 256, False, True), (64, 64, 64, False, False), (64, 64, 64, True, False), (200, 130, 77, False, True)]:
    g = torch.Generator().manual_seed(1)
    A = torch.randn((K, M) if ta else (M, K), generator=g).to(dev)
    B = torch.randn((N, K) if tb else (K, N), generator=g).to(dev)
    C = torch.zeros(M, N, device=dev)
    kn.gemm(A, B, C, trans_a=ta, trans_b=tb, a_planes=kn.split_planes(A), b_planes=kn.split_planes(B))
    torch.cuda.synchronize()
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    err = (C.double() - ref).abs().max().item() / ref.abs().max().item()
    print(M, N, K, ta, tb, "err %.2e" % err, flush=True)
