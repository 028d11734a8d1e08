"""Where does GraphedStep.load_batch spend its time?  (host plan build vs H2D copies vs waits)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import build_criterion, build_model, synthetic
from mesm_amd.graphed import GraphedStep

dev = torch.device("cuda:0")
args = synthetic.make_args("C3a", device=str(dev))
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args)
cpu = synthetic.workload_batch("C3a", seed=0)
batch = synthetic.to_device(cpu, dev)
g = GraphedStep(model, crit, batch, args.dataset_name, warmup=1, caps="auto")
for _ in range(3):
    g.load_batch(cpu); g.run()
torch.cuda.synchronize()


def t(f, n=10, sync=True):
    if sync: torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    if sync: torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("replay only                 %.2f ms" % t(lambda: g.run(redraw=False)))
print("redraw + replay             %.2f ms" % t(lambda: g.run(redraw=True)))
print("load_batch (idle GPU)       %.2f ms" % t(lambda: g.load_batch(cpu)))
print("load_batch + replay         %.2f ms" % t(lambda: (g.load_batch(cpu), g.run(redraw=False))))
import cProfile, pstats
torch.cuda.synchronize()
cProfile.run("for _ in range(5): g.load_batch(cpu); g.run(redraw=False)\ntorch.cuda.synchronize()", "/tmp/lb.prof")
pstats.Stats("/tmp/lb.prof").sort_stats("tottime").print_stats(14)
