"""Where do the kernel launches of one training step come from?  Eager step under torch.profiler:
forward kernels grouped by model region (record_function scopes set by MESM_SCOPES=1), backward
kernels grouped by autograd node.  Usage: python tools/count_kernels.py [workload]"""
import collections, os, sys
os.environ.setdefault("MESM_AUTOGRAPH", "0")  # this tool looks at the EAGER step (autograph.py would replay graphs behind these calls)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MESM_SCOPES"] = "1"
import torch
from torch.profiler import profile, ProfilerActivity
from mesm_amd import build_criterion, build_model, synthetic

wl = sys.argv[1] if len(sys.argv) > 1 else "C3a"
dev = torch.device("cuda:0")
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)


def step():
    with torch.profiler.record_function("FWD"):
        out = model(**batch, dataset_name=args.dataset_name, is_training=True)
    with torch.profiler.record_function("CRIT"):
        losses, total = crit(out, batch, True)
    model.zero_grad(set_to_none=True)
    with torch.profiler.record_function("BWD"):
        total.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()

evs = prof.events()
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
# top-level-ish scopes: user scopes and autograd nodes
scopes = [e for e in cpu if e.name in ("FWD", "CRIT", "BWD") or e.name.startswith("R:") or
          e.name.startswith("autograd::engine::evaluate_function")]
scopes.sort(key=lambda e: (e.time_range.start, -e.time_range.end))
kern = collections.Counter()
ktime = collections.Counter()
for e in cpu:
    if not e.kernels:
        continue
    # innermost enclosing scope
    best = None
    for s in scopes:
        if s.thread == e.thread and s.time_range.start <= e.time_range.start and e.time_range.end <= s.time_range.end:
            if best is None or s.time_range.start >= best.time_range.start:
                best = s
    name = best.name.replace("autograd::engine::evaluate_function: ", "bwd:") if best else "?"
    # only count launches attributed to leaf ops (avoid double counting parents)
    if any(c.kernels for c in e.cpu_children):
        continue
    kern[name] += len(e.kernels)
    ktime[name] += sum(k.duration for k in e.kernels)
# backward kernels by the model REGION whose forward created the autograd node (sequence numbers link the two)
seq_scope = {}
regions = [e for e in cpu if e.name.startswith("R:") or e.name == "CRIT"]
for e in cpu:
    sq = getattr(e, "sequence_nr", -1)
    if sq is None or sq < 0 or e.name.startswith("autograd::engine"):
        continue
    best = None
    for s in regions:
        if s.thread == e.thread and s.time_range.start <= e.time_range.start and e.time_range.end <= s.time_range.end:
            if best is None or s.time_range.start >= best.time_range.start:
                best = s
    if best is not None and sq not in seq_scope:
        seq_scope[sq] = best.name
rk, rt = collections.Counter(), collections.Counter()
for e in cpu:
    if not e.name.startswith("autograd::engine::evaluate_function"):
        continue
    reg = seq_scope.get(getattr(e, "sequence_nr", -1), "?")
    stack = [e]
    while stack:
        x = stack.pop()
        if x.kernels and not any(c.kernels for c in x.cpu_children):
            rk["bwd@" + reg] += len(x.kernels)
            rt["bwd@" + reg] += sum(k.duration for k in x.kernels)
        stack.extend(x.cpu_children)
print("backward kernels by forward region:")
for n, c in rk.most_common(20):
    print("%-60s %5d  %8.1f us" % (n, c, rt[n]))
tot = sum(kern.values())
print("total kernels in step: %d" % tot)
for n, c in kern.most_common(70):
    print("%-60s %5d  %8.1f us" % (n[:60], c, ktime[n]))
