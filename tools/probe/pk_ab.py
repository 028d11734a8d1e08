"""A/B of the persistent grouped split-bf16 launch (csrc/gemm_pk.hip) on the GEMM calls of one captured step: every
call of the step's tape timed ALONE (mesm_gemm_tape_entry, its own grouping and buffers) with the persistent kernel off
(one workgroup per tile: gemm_wstage64_group_kernel) and on, at the given grids.
usage: pk_ab.py [workload] [grid:cut_min[:c0] ...]   (default: 512:8 512:2)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic, kernels as kn
from mesm_amd._lib import lib
from mesm_amd.graphed import GraphedStep

wl = sys.argv[1] if len(sys.argv) > 1 else "C3a"
grids = [x for x in sys.argv[2:]] or ["512:8", "512:2"]
dev = torch.device("cuda:0")
args = synthetic.make_args(wl, device=str(dev))
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch(wl, seed=0), dev)
step = GraphedStep(model, crit, batch, args.dataset_name, warmup=1, instrument=True)
step.run(); torch.cuda.synchronize()
n = lib().mesm_gemm_tape_size()
shapes = (ctypes.c_int32 * 256)()


def time_all():
    out = []
    for i in range(n):
        ms = ctypes.c_double(); k = ctypes.c_int32()
        for reps in (3, 40):
            kn.check(lib().mesm_gemm_tape_entry(kn.stream_ptr(), i, reps, ctypes.byref(ms), ctypes.byref(k), shapes), "tape_entry")
        probs = [(shapes[4 * j], shapes[4 * j + 1], shapes[4 * j + 2], shapes[4 * j + 3]) for j in range(k.value)]
        out.append((ms.value * 1e3, probs))
    return out


def desc(p):
    M, N, K, f = p
    return "%dx%dx%d%s%s%s" % (M, N, K, "T" if f & 256 else "N", "T" if f & 512 else "N", ("/s%d" % (f & 255)) if (f & 255) > 1 else "")


kn.gemm_pk(on=0)
base = time_all()
runs = {}
for g in grids:
    f = [int(x) for x in g.split(":")] + [3]
    kn.gemm_pk(on=1, grid=f[0], cut_min=f[1], c0=f[2])
    runs[g] = time_all()
assert kn.gemm_pk_status() == 0, "a persistent launch timed out waiting for a partial tile"
kn.gemm_pk(on=1, grid=512, cut_min=8, c0=3)
fl = [sum(2.0 * M * N * K for M, N, K, _ in probs) for _, probs in base]
print("%d calls, %.1f GFLOP; one at a time: per-tile launches %.1f us" % (n, sum(fl) / 1e9, sum(b[0] for b in base))
      + "".join("; persistent %s: %.1f us" % (g, sum(r[0] for r in runs[g])) for g in grids))
print("%4s %9s" % ("#", "per-tile") + "".join(" %9s" % ("pk" + g) for g in grids) + "   GF    call")
for i in range(n):
    tiles = sum(((M + 63) // 64) * ((N + 63) // 64) * max(f & 255, 1) for M, N, K, f in base[i][1])
    print("%4d %9.2f" % (i, base[i][0]) + "".join(" %9.2f" % runs[g][i][0] for g in grids)
          + "  %5.2f  tiles64=%d  %s" % (fl[i] / 1e9, tiles, " ".join(desc(p) for p in base[i][1])))
