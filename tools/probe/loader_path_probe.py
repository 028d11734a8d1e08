"""Where the loader-worker path's time goes (bench.py's loader-like stream): per-batch times of worker prepare, the pin
thread, load_prepared, replay; variants: workers 0 / 2 / 4, pin on / off."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic
from mesm_amd.graphed import StepCache
from mesm_amd.loader import prepared_loader
dev = torch.device("cuda:0")
args = synthetic.make_args("C3a", device=str(dev)); wl = synthetic.WORKLOADS["C3a"]
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
rng = random.Random(5)
sizes, probs = list(range(1, 10)), [0.18, 0.22, 0.20, 0.15, 0.10, 0.07, 0.04, 0.025, 0.015]
cache = StepCache(model, crit, args.dataset_name, pad=(wl["Lv"], wl["Lw"]), pairs=16, group_caps=(5, 9))
stream = []
for i in range(24):
    groups = [rng.choices(sizes, probs)[0] for _ in range(12)]
    stream.append(synthetic.make_batch(wl["dataset_name"], groups, wl["Lv"], wl["Lw"], wl["v_feat_dim"], wl["t_feat_dim"], wl["vocab_size"] + 1, seed=1000 + i, ragged=True))
for hb in stream: cache.run(hb, redraw=True)
torch.cuda.synchronize()
def t_inproc():
    t0 = time.perf_counter()
    for hb in stream: cache.run(hb, redraw=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / len(stream) * 1e3
print("in-process: %.2f ms/step" % t_inproc(), flush=True)
pipe = cache.pipeline(keep_raw=False)
t0 = time.perf_counter(); preps = [pipe.prepare(hb) for hb in stream]; print("prepare (this process): %.2f ms/batch" % ((time.perf_counter() - t0) / len(stream) * 1e3))
t0 = time.perf_counter()
for p in preps: cache.run_prepared(p)
torch.cuda.synchronize(); print("run_prepared on ready (pageable) preps: %.2f ms/step" % ((time.perf_counter() - t0) / len(stream) * 1e3))
pinned = [dict(p, big={k: v.pin_memory() for k, v in p["big"].items()}) for p in preps]
t0 = time.perf_counter()
for p in pinned: cache.run_prepared(p)
torch.cuda.synchronize(); print("run_prepared on ready PINNED preps: %.2f ms/step" % ((time.perf_counter() - t0) / len(stream) * 1e3))
for workers in (0, 2, 4):
    for pin in (False, True):
        ld = prepared_loader(stream, pipe, num_workers=workers, pin_memory=pin)
        for p in ld: cache.run_prepared(p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for p in ld: pass
        tl = (time.perf_counter() - t0) / len(stream) * 1e3
        t0 = time.perf_counter()
        for p in ld: cache.run_prepared(p)
        torch.cuda.synchronize()
        print("workers %d pin %d: loader alone %.2f ms/batch, loader + run_prepared %.2f ms/step" % (workers, pin, tl, (time.perf_counter() - t0) / len(stream) * 1e3), flush=True)
        del ld
