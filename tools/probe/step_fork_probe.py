"""Probe: what does a parallel branch cost inside the captured training step?  The real C3a step is captured
with a side stream forked at its start and joined at its end; the side branch runs n small independent GEMMs
(320 x 256 x 256, ~4.5 us each when serial).  n = -1: no fork at all.  usage: step_fork_probe.py "-1,0,1,50,150" """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import build_criterion, build_model, synthetic, kernels as kn
from mesm_amd.graphed import GraphedStep

dev = torch.device("cuda:0")
wl = "C3a"
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)
M, N, K = [int(x) for x in os.environ.get("SIDE_SHAPE", "320,256,256").split(",")]
A = [torch.randn(M, K, device=dev) for _ in range(8)]
W = [torch.randn(N, K, device=dev) for _ in range(8)]
C = [torch.zeros(M, N, device=dev) for _ in range(8)]


class Forked(GraphedStep):
    n_side = -1

    def _step_body(self):
        if self.n_side < 0:
            return super()._step_body()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for i in range(self.n_side):
                kn.gemm(A[i % 8], W[i % 8], C[i % 8], trans_b=True)
        r = super()._step_body()
        cur.wait_stream(side)
        return r


for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "-1,0,1,50,150").split(",")]:
    Forked.n_side = n
    gs = Forked(model, crit, batch, args.dataset_name)
    for _ in range(5):
        gs.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        gs.run()
    torch.cuda.synchronize()
    print("side branch of %3d GEMMs (%dx%dx%d): %.3f ms/step" % (n, M, N, K, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
    del gs
