"""Every GEMM launch of one captured step timed ALONE in the grouping the step uses (mesm_gemm_tape_entry): time, flops,
TF, and the time lost against RATE (the big-GEMM rate, default 90 TF).  Sorted by lost time, then totals per kind.
usage: tape_profile.py [workload] [rate_tf]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic, kernels as kn
from mesm_amd._lib import lib
from mesm_amd.graphed import GraphedStep

wl = sys.argv[1] if len(sys.argv) > 1 else "C3a"
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
dev = torch.device("cuda:0")
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)
step = GraphedStep(model, crit, batch, args.dataset_name, warmup=1, instrument=True)
step.run(); torch.cuda.synchronize()
n = lib().mesm_gemm_tape_size()
rows = []
shapes = (ctypes.c_int32 * 256)()
for i in range(n):
    ms = ctypes.c_double(); k = ctypes.c_int32()
    for reps in (3, 30):
        kn.check(lib().mesm_gemm_tape_entry(kn.stream_ptr(), i, reps, ctypes.byref(ms), ctypes.byref(k), shapes), "tape_entry")
    probs = [(shapes[4 * j], shapes[4 * j + 1], shapes[4 * j + 2], shapes[4 * j + 3]) for j in range(k.value)]
    fl = sum(2.0 * M * N * K for M, N, K, _ in probs)
    rows.append((i, ms.value * 1e3, fl, probs))
tot_us = sum(r[1] for r in rows); tot_fl = sum(r[2] for r in rows)
print("%d launches, %.1f us back to back one at a time, %.1f GFLOP -> %.1f TF; at %.0f TF: %.1f us" %
      (n, tot_us, tot_fl / 1e9, tot_fl / tot_us / 1e6, rate, tot_fl / rate / 1e6))
def desc(p):
    M, N, K, f = p
    # '*': not taken by the grouped 32 x 32 launch of its call -- a kernel of its own
    return "%dx%dx%d%s%s%s%s" % (M, N, K, "T" if f & 256 else "N", "T" if f & 512 else "N", ("/s%d" % (f & 255)) if (f & 255) > 1 else "",
                               "" if f & 1024 else "*")
rows.sort(key=lambda r: -(r[1] - r[2] / rate / 1e6))
for i, us, fl, probs in rows:
    lost = us - fl / rate / 1e6
    print("#%3d %7.2f us %7.3f GF %5.1f TF lost %6.2f us  %s" % (i, us, fl / 1e9, fl / us / 1e6, lost, " ".join(desc(p) for p in probs)))
alone = [(2.0 * M * N * K / 1e9, desc((M, N, K, f))) for _, _, _, probs in rows for (M, N, K, f) in probs if not f & 1024]
alone.sort()
print("problems launched as kernels of their own: %d; below 1.5 GF: %s" % (len(alone), " ".join("%s(%.2f)" % (d, g) for g, d in alone if g < 1.5)))

