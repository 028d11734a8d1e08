"""Host-side profile of the loader-like stream of bench.py (StepCache over batches from host memory)."""
import cProfile, pstats, random, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic
from mesm_amd.graphed import StepCache
dev = torch.device("cuda:0")
pin = len(sys.argv) > 1 and sys.argv[1] == "pin"
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
args = synthetic.make_args("C3a", device=str(dev))
wl = synthetic.WORKLOADS["C3a"]
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
rng = random.Random(5)
sizes, probs = list(range(1, 10)), [0.18, 0.22, 0.20, 0.15, 0.10, 0.07, 0.04, 0.025, 0.015]
cache = StepCache(model, crit, args.dataset_name, pad=(wl["Lv"], wl["Lw"]), pairs=pairs, group_caps=(5, 9))
stream = []
for i in range(24):
    groups = [rng.choices(sizes, probs)[0] for _ in range(12)]
    hb = synthetic.make_batch(wl["dataset_name"], groups, wl["Lv"], wl["Lw"], wl["v_feat_dim"], wl["t_feat_dim"], wl["vocab_size"] + 1, seed=1000 + i, ragged=True)
    if pin:
        hb = {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in hb.items()}
    stream.append(hb)
for hb in stream: cache.run(hb, redraw=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for hb in stream: cache.run(hb, redraw=True)
torch.cuda.synchronize()
pr.disable()
print("pin=%s pairs=%d: %.2f ms/step, graphs %d" % (pin, pairs, (time.perf_counter() - t0) / len(stream) * 1e3, cache.captures))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
