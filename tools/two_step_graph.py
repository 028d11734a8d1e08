"""What does the boundary between two graph replays cost on the device?  The headline step as one HIP graph per step
against the same two steps captured back to back in ONE graph (same draws for both: a timing probe, not a training loop).
usage: python tools/two_step_graph.py [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic, kernels as kn
from mesm_amd.graphed import GraphedStep, capture_stream
wl = sys.argv[1] if len(sys.argv) > 1 else "C3a"
dev = torch.device("cuda:0")
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(1234)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)
g = GraphedStep(model, crit, batch, args.dataset_name)
side = capture_stream(dev)
kn.set_seed_offset(g.counter)
graphs = {}
for n in (1, 2, 4):
    model.zero_grad(set_to_none=True)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side, capture_error_mode="thread_local"):
        for _ in range(n):
            g.counter.add_(1)
            g._step_body()
    graphs[n] = gr
kn.set_seed_offset(None)

def timed(fn, iters):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(iters): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e3)
    return best

for _ in range(40): g.graph.replay()
torch.cuda.synchronize()
base = timed(g.graph.replay, 60)
print("the step's own graph, replayed back to back      %.3f ms/step" % base)
for n, gr in graphs.items():
    t = timed(gr.replay, 60 // n)
    print("%d step(s) per graph                               %.3f ms/step  (%+.1f us per step)" % (n, t / n, (t / n - base) * 1e3))
