"""cProfile of GraphedStep.redraw() (the host work of a replayed step) with the GPU idle and busy."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
t0 = time.perf_counter()
for _ in range(30): g.redraw()
torch.cuda.synchronize()
print("redraw alone (GPU idle): %.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    g.redraw(); g.graph.replay()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime"); st.print_stats(22)
