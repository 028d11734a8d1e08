"""Step time with the SS-MESM / FW-MESM branches switched off (how much work sits in them)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic
from mesm_amd.graphed import GraphedStep
dev = torch.device("cuda:0")
def run(**over):
    args = synthetic.make_args("C3a", device=str(dev))
    for k, v in over.items(): setattr(args, k, v)
    torch.manual_seed(1)
    model = build_model(args); crit = build_criterion(args); model.train()
    batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=0), dev)
    g = GraphedStep(model, crit, batch, args.dataset_name, warmup=1)
    for _ in range(3): g.run(redraw=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.run(redraw=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3
full = run()
print("full %.3f ms" % full)
print("no rec_ss %.3f ms" % run(rec_ss=False))
print("no rec_fw %.3f ms" % run(rec_fw=False))
print("neither   %.3f ms" % run(rec_ss=False, rec_fw=False))
