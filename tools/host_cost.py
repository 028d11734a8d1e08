"""Host-side cost of one replayed step: redraw (host RNG draws + arena upload) and graph.replay() (enqueue only), next
to the GPU time of the step.  A step is host-bound when the first two add up to more than the third.
usage: python tools/host_cost.py [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic
from mesm_amd.graphed import GraphedStep
wl = sys.argv[1] if len(sys.argv) > 1 else "C3a"
dev = torch.device("cuda:0")
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(1234)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)
g = GraphedStep(model, crit, batch, args.dataset_name)
for _ in range(5): g.run()
torch.cuda.synchronize()
N = 30
tr = tp = 0.0
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter(); g.redraw(); b = time.perf_counter(); g.graph.replay(); c = time.perf_counter()
    tr += b - a; tp += c - b
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
# the same with the device idle at the start of every step: redraw() then never waits for the arena mirror it wrote
# two uploads ago (back-pressure of the loop above: that wait is as long as the device step and is not host work)
ir = ip = 0.0
for _ in range(N):
    torch.cuda.synchronize()
    a = time.perf_counter(); g.redraw(); b = time.perf_counter(); g.graph.replay(); c = time.perf_counter()
    ir += b - a; ip += c - b
torch.cuda.synchronize()
# GPU time alone: replays back to back without the host work in between
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): g.graph.replay()
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / N * 1e3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): g.graph.replay()
e1.record(); torch.cuda.synchronize()
print("host work with the device idle at the start of each step: redraw %.3f ms + replay call %.3f ms" % (ir / N * 1e3, ip / N * 1e3))
print("wall %.3f ms/step with redraw; host: redraw %.3f ms + replay call %.3f ms; replay-only wall %.3f ms; event-timed %.3f ms"
      % (wall, tr / N * 1e3, tp / N * 1e3, gpu, e0.elapsed_time(e1) / N))
