"""Who zero-fills small tensors during one training step (eager AND under capture): patches the fill entry points
and prints the innermost mesm_amd frames.  usage: dbg_fills.py"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic

dev = torch.device("cuda:0")
args = synthetic.make_args("C3a", device=str(dev))
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=0), dev)
seen = collections.Counter()
ON = [False]


def where():
    fr = [f for f in traceback.extract_stack()[:-2] if "mesm_amd" in f.filename]
    return " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in fr[-3:][::-1])


def wrap(mod, name, tag, shape_of):
    orig = getattr(mod, name)

    def f(*a, **k):
        if ON[0]:
            try:
                seen[(tag, shape_of(a, k), where())] += 1
            except Exception:
                pass
        return orig(*a, **k)
    setattr(mod, name, f)


wrap(torch, "zeros", "zeros", lambda a, k: tuple(a[0]) if isinstance(a[0], (tuple, list, torch.Size)) else tuple(a))
wrap(torch, "zeros_like", "zeros_like", lambda a, k: tuple(a[0].shape))
wrap(torch.Tensor, "zero_", "zero_", lambda a, k: tuple(a[0].shape))
wrap(torch.Tensor, "fill_", "fill_", lambda a, k: tuple(a[0].shape))
wrap(torch.Tensor, "copy_", "copy_", lambda a, k: tuple(a[0].shape))
wrap(torch.Tensor, "new_zeros", "new_zeros", lambda a, k: tuple(a[1]) if isinstance(a[1], (tuple, list, torch.Size)) else tuple(a[1:]))


def step():
    model.gradbuf().zero()
    out = model(**batch, dataset_name=args.dataset_name, is_training=True)
    losses, total = crit(out, batch, True)
    total.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
ON[0] = True
step()
ON[0] = False
torch.cuda.synchronize()
for (tag, shape, w), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("%3d x %-10s %-18s %s" % (n, tag, shape, w))
