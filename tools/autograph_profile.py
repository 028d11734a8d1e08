"""Where the host time of the unchanged-caller path (mesm_amd/autograph.py) goes: cProfile over replayed loop bodies
(train.py:64-72) + wall time of each call of the sequence with a device synchronisation behind it.
usage: python tools/autograph_profile.py [steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from mesm_amd import build_criterion, build_model, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
args = synthetic.make_args("C3a", device="cuda:0")
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args)
model.train(); model.autograph(True)
opt = torch.optim.AdamW(model.parameters(), lr=0.0, weight_decay=1e-4)
batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=0), dev)


def body(sync=False, t=None):
    def mark(k, t0):
        if sync:
            torch.cuda.synchronize()
        if t is not None:
            t[k] = t.get(k, 0.0) + time.perf_counter() - t0
        return time.perf_counter()
    t0 = time.perf_counter()
    out = model(**batch, dataset_name=args.dataset_name, is_training=True); t0 = mark("model()", t0)
    ld, loss = crit(out, batch, is_training=True); t0 = mark("criterion()", t0)
    opt.zero_grad(); t0 = mark("zero_grad()", t0)
    loss.backward(); t0 = mark("backward()", t0)
    nn.utils.clip_grad_norm_(model.parameters(), 0.1); t0 = mark("clip_grad_norm_", t0)
    opt.step(); t0 = mark("optimizer.step()", t0)


for _ in range(4):
    body()
torch.cuda.synchronize()
for sync in (False, True):
    t = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        body(sync, t)
    torch.cuda.synchronize()
    print("%s: %.3f ms/step   " % ("each call followed by a device sync" if sync else "host time per call (asynchronous)",
                                   (time.perf_counter() - t0) / steps * 1e3)
          + "  ".join("%s %.3f" % (k, v / steps * 1e3) for k, v in t.items()))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    body()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative")
st.print_stats(45)
