"""Why does the FFN1 forward GEMM (2400x1024x256) take 60 us inside the step and 29 us alone?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
M, N, K = 2400, 1024, 256
NS = 8
xs = [torch.randn(M, K, device=dev) for _ in range(NS)]
ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(NS)]
bs = [torch.randn(N, device=dev) for _ in range(NS)]
g = torch.ones(K, device=dev); b = torch.zeros(K, device=dev)
zs = [torch.empty(M, N, device=dev) for _ in range(NS)]

def run(body, n=32, reps=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): body()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n * 1e6

def plain():
    for i in range(32): kn.gemm(xs[i % NS], ws[i % NS], zs[i % NS], trans_b=True)
def bias():
    for i in range(32): kn.gemm(xs[i % NS], ws[i % NS], zs[i % NS], trans_b=True, bias=bs[i % NS])
def fresh_out():
    for i in range(32):
        z = torch.empty(M, N, device=dev)
        kn.gemm(xs[i % NS], ws[i % NS], z, trans_b=True, bias=bs[i % NS])
def ln_only():
    for i in range(32): kn.layernorm_fwd(xs[i % NS], g, b)
def ln_then():
    for i in range(32):
        y, _, _ = kn.layernorm_fwd(xs[i % NS], g, b)
        kn.gemm(y, ws[i % NS], zs[i % NS], trans_b=True, bias=bs[i % NS])
def ffn_pair():
    for i in range(32):
        kn.gemm(xs[i % NS], ws[i % NS], zs[i % NS], trans_b=True, bias=bs[i % NS])
        kn.gemm(zs[i % NS], ws[(i + 1) % NS].t().contiguous() if False else ws2[i % NS], ys[i % NS], trans_b=True, bias=bs2[i % NS])
ws2 = [torch.randn(K, N, device=dev) * 0.05 for _ in range(NS)]
bs2 = [torch.randn(K, device=dev) for _ in range(NS)]
ys = [torch.empty(M, K, device=dev) for _ in range(NS)]
print("plain            %.2f us" % run(plain))
print("bias             %.2f us" % run(bias))
print("fresh output     %.2f us" % run(fresh_out))
t_ln = run(ln_only)
print("LN only          %.2f us" % t_ln)
print("LN -> FFN1       %.2f us (pair)" % run(ln_then))
print("FFN1 -> FFN2     %.2f us (pair)" % run(ffn_pair))
for tile in ("64", "32", "2", "1"):
    os.environ["MESM_GEMM_TILE"] = tile
    print("tile %s: bias %.2f us ; LN->FFN1 pair %.2f us" % (tile, run(bias), run(ln_then)))
