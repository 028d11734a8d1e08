"""which single-block autograd nodes of a step receive MATERIALISED zero gradients (autograd fills one tensor per output
that nothing reached: an element-wise launch each) -- python tools/grad_target_debug.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import build_criterion, build_model, synthetic, ops
dev = torch.device("cuda:0")
seen = collections.Counter()
orig_make = ops.make_fn


def make_fn(block):
    Fn = orig_make(block)
    ob = Fn.backward

    def backward(ctx, *gs):
        for i, g in enumerate(gs):
            if torch.is_tensor(g) and g.numel() <= (1 << 20) and float(g.abs().sum()) == 0.0:
                seen[(block.__name__, i, tuple(g.shape))] += 1
        return ob(ctx, *gs)
    Fn.backward = staticmethod(backward)
    return Fn


ops.make_fn = make_fn
ops._FN_CACHE.clear()
import mesm_amd.layers as _layers, mesm_amd.model as _model, mesm_amd.criterion as _crit
for mod in (ops, _layers, _model, _crit):
    for name, obj in list(vars(mod).items()):
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
            ob = obj.backward

            def backward(ctx, *gs, _ob=ob, _n=name):
                for i, g in enumerate(gs):
                    if torch.is_tensor(g) and g.numel() <= (1 << 20) and float(g.abs().sum()) == 0.0:
                        seen[(_n, i, tuple(g.shape))] += 1
                return _ob(ctx, *gs)
            obj.backward = staticmethod(backward)
args = synthetic.make_args("C3a", device=str(dev))
torch.manual_seed(1234)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=0), dev)
out = model(**batch, dataset_name=args.dataset_name, is_training=True)
losses, total = crit(out, batch, True)
total.backward()
torch.cuda.synchronize()
for k, v in sorted(seen.items()):
    print(v, k)
