"""Host-side profile of the EAGER step (no HIP graph: what every first batch of a shape bucket and every capture warm-up
pays): cProfile over 10 steps of C3a, top functions by own time and by cumulative time.
usage: python tools/eager_profile.py [workload] [n]"""
import cProfile, os, pstats, sys, time
os.environ.setdefault("MESM_AUTOGRAPH", "0")  # this tool looks at the EAGER step (autograph.py would replay graphs behind these calls)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic
wl = sys.argv[1] if len(sys.argv) > 1 else "C3a"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(1234)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)


def step():
    out = model(**batch, dataset_name=args.dataset_name, is_training=True)
    losses, total = crit(out, batch, True)
    model.zero_grad(set_to_none=True)
    total.backward()
    return total


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print("eager: %.2f ms/step" % ((time.perf_counter() - t0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
