import os, sys, traceback, collections
sys.path.insert(0, "/root/repo")
import torch
from mesm_amd import build_criterion, build_model, synthetic, gradbuf, kernels as kn
dev = torch.device("cuda:0")
args = synthetic.make_args("C3a", device=str(dev))
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=0), dev)
def step():
    model.gradbuf().zero()
    out = model(**batch, dataset_name=args.dataset_name, is_training=True)
    losses, total = crit(out, batch, True)
    total.backward()
for _ in range(2): step()
log = collections.Counter()
orig_z, orig_zl = torch.zeros, torch.zeros_like
def where():
    fr = [f for f in traceback.extract_stack() if "mesm_amd" in f.filename]
    return " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[-3:])
def z(*a, **k):
    t = orig_z(*a, **k)
    if t.is_cuda: log[("zeros", tuple(t.shape), where())] += 1
    return t
def zl(x, *a, **k):
    if x.is_cuda: log[("zeros_like", tuple(x.shape), where())] += 1
    return orig_zl(x, *a, **k)
torch.zeros, torch.zeros_like = z, zl
step(); torch.cuda.synchronize()
for k, n in sorted(log.items(), key=lambda kv: -kv[1]): print(n, k)
