"""Which Python lines launch the ATen (non-library) kernels of one eager training step: torch profiler
with stacks, grouped by op and innermost mesm_amd frame.  usage: aten_origin.py [workload]"""
import collections, os, sys
os.environ.setdefault("MESM_AUTOGRAPH", "0")  # this tool looks at the EAGER step (autograph.py would replay graphs behind these calls)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from mesm_amd import build_criterion, build_model, synthetic

dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "C3a"
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)


plan_holder = {}


def step():
    model.gradbuf().zero()
    out = model(**batch, dataset_name=args.dataset_name, is_training=True)
    losses, total = crit(out, batch, True)
    total.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
cfg = torch._C._profiler._ExperimentalConfig(verbose=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=cfg) as prof:
    step()
    torch.cuda.synchronize()
rows = collections.Counter(); times = collections.Counter()
# forward call site of every autograd node, by sequence number (the engine-side fan-in adds have no python frame)
fwd_site = {}
for e in prof.events():
    sn = getattr(e, "sequence_nr", -1)
    if sn is not None and sn >= 0 and e.stack and "Backward" not in e.name:
        fr = [f for f in e.stack if "mesm_amd" in f and "ops.py" not in f and "kernels.py" not in f]
        if fr and sn not in fwd_site:
            # innermost model / layers frame(s): the stack lists the innermost frame first
            fwd_site[sn] = " < ".join(f.split("mesm_amd/")[-1].split(":")[0] for f in fr[:2])
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    dt = getattr(e, "self_device_time_total", 0) or 0
    if dt <= 0:
        continue
    anc, frames = e, []
    while anc is not None and not frames:
        frames = [f for f in (anc.stack or []) if "mesm_amd" in f]
        anc = anc.cpu_parent
    where = frames[0].split("mesm_amd/")[-1] if frames else "<autograd engine / no python frame>"
    # for engine-side ops name the autograd node when the profiler recorded one as a parent
    par = e.cpu_parent
    node = ""
    while par is not None:
        if "Backward" in par.name or "AccumulateGrad" in par.name:
            node = par.name.split("autograd::engine::evaluate_function: ")[-1]
            sn = getattr(par, "sequence_nr", -1)
            if not frames and sn in fwd_site:
                where = "bwd of " + fwd_site[sn]
            break
        par = par.cpu_parent
    shp = str([tuple(x) for x in (e.input_shapes or []) if x])[:60]
    rows[(e.name, where, node, shp)] += 1
    times[(e.name, where, node, shp)] += dt
tot = sum(rows.values())
print("ATen ops with device work in one step: %d, device time %.1f us" % (tot, sum(times.values())))
for k, n in sorted(rows.items(), key=lambda kv: -times[kv[0]]):
    print("%4d x %-22s %7.1f us  %-44s %-26s %s" % (n, k[0], times[k], k[1][:44], k[2][:26], k[3]))
