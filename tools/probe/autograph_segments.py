"""host segments of one replayed model(...) call of the unchanged-caller path, device idle at the start of every call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mesm_amd import build_criterion, build_model, synthetic, graphed, autograph, hostplan
from mesm_amd.criterion import TargetPlan

dev = torch.device("cuda:0")
args = synthetic.make_args("C3a", device="cuda:0")
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args)
model.train(); model.autograph(True)
batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=0), dev)
T = {}


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        T[label] = T.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)


def step():
    out = model(**batch, dataset_name=args.dataset_name, is_training=True)
    _, loss = crit(out, batch, is_training=True)
    model.zero_grad(set_to_none=True)
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
auto = model._auto
st = next(iter(auto.steps.values()))[0]
wrap(auto, "fetch", "fetch (pack + D2H + sync + unpack)")
wrap(st.spec, "plan_arrays", "  plan_arrays")
wrap(TargetPlan, "arrays", "  TargetPlan.arrays")
wrap(st.spec, "words_mask", "  words_mask")
wrap(st, "_host_arrays", "host_arrays (all)")
wrap(st.arena, "check", "arena.check")
wrap(st.arena, "upload", "arena.upload")
wrap(st, "load_batch", "load_batch (all)")
wrap(st, "forward_replay", "forward_replay")
wrap(auto, "forward", "AutoGraph.forward (all)")
n = 30
for _ in range(n):
    torch.cuda.synchronize()
    step()
torch.cuda.synchronize()
for k, v in T.items():
    print("%-40s %.3f ms" % (k, v / n * 1e3))
