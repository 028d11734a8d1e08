"""Which tensors of the step have several consumers in the autograd graph (= gradient fan-in adds by the engine):
producer node, output index, shape, consumers."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "C3a"
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)
from mesm_amd import ops
ops.FANIN_DEBUG = {}
out = model(**batch, dataset_name=args.dataset_name, is_training=True)
losses, total = crit(out, batch, True)
seen, edges, stack = set(), collections.defaultdict(list), [total.grad_fn]
while stack:
    fn = stack.pop()
    if fn is None or id(fn) in seen:
        continue
    seen.add(id(fn))
    for nxt, idx in fn.next_functions:
        if nxt is not None:
            edges[(nxt, idx)].append(fn.name())
            stack.append(nxt)
rows = []
for (fn, idx), cons in edges.items():
    if len(cons) >= 2 and "AccumulateGrad" not in fn.name():
        try:
            shp = tuple(fn._input_metadata[idx].shape)
        except Exception:
            shp = "?"
        n = 1
        for d in (shp if shp != "?" else ()):
            n *= d
        rows.append((n, fn.name(), idx, shp, cons))
for n, name, idx, shp, cons in sorted(rows, key=lambda r: -r[0]):
    print("%-28s out %d %-18s <- %d consumers: %s" % (name, idx, shp, len(cons), ", ".join(c.replace("Backward", "") for c in cons)))
acc = [(fn, cons) for (fn, idx), cons in edges.items() if "AccumulateGrad" in fn.name() and len(cons) >= 2]
print("parameters with several gradient contributions through autograd:", [(tuple(fn.variable.shape), len(c)) for fn, c in acc])

# which block inputs received a gradient from several blocks (lockstep / stand-alone blocks only: the assembly and
# criterion functions are not instrumented, their contributions come on top)
total.backward()
torch.cuda.synchronize()
print("block inputs with gradients from several blocks:")
for (ptr, shp), names in sorted(ops.FANIN_DEBUG.items(), key=lambda kv: -len(kv[1])):
    if len(names) >= 2 or (len(shp) == 3 and shp[0] * shp[1] * shp[2] > 500000):
        print("  %-18s x%d  %s" % (shp, len(names), ", ".join(names)))

