"""Run-to-run spread of one training step on identical inputs and draws: the order of float atomic adds (split-K weight gradients,
and since round 5 the K-split remainder rows / deep products of the forward) is the only freedom.  Relative difference of the
loss and of the flat gradient between replays of one graph, and between an eager step and a replay.  usage: run_to_run.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mesm_amd import build_criterion, build_model, synthetic
from mesm_amd.graphed import GraphedStep
wl = sys.argv[1] if len(sys.argv) > 1 else "C3a"
dev = torch.device("cuda:0")
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(1234); np.random.seed(1)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)
g = GraphedStep(model, crit, batch, args.dataset_name)
gb = model.gradbuf()
def replay():
    g.counter.fill_(7)  # same dropout masks every time
    t = g.run(redraw=False)
    torch.cuda.synchronize()
    return float(t), gb.flat.clone()
t0, f0 = replay()
le, ge = [], []
for _ in range(20):
    t, f = replay()
    le.append(abs(t - t0) / max(1.0, abs(t0))); ge.append(float((f - f0).norm()) / float(f0.norm()))
print("%s, 20 replays against the first: loss rel diff max %.2e, gradient rel diff (norm) max %.2e median %.2e"
      % (wl, max(le), max(ge), sorted(ge)[10]))
