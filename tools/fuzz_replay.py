"""Investigation of a SEQUENCE-dependent fuzz mismatch (round-3 review, case 14 of `fuzz_parity.py 200 777`).
Replays cases 0 .. last of a sweep in sequence and, for every case, answers three questions:

 (a) does the HIP path reproduce itself?   two steps in a row on the same model / batch, gradients compared (the only
     run-to-run freedom is the order of the float atomics of the split-K weight gradients: ~1e-7);
 (b) does the HIP path read memory it did not write?   MODE=poison wraps torch.empty / empty_like / new_empty of this
     process so that every CUDA buffer the product path allocates starts as NaN (floats) or 0x7f7f.. (integers): a kernel
     that accumulates into, or reads beyond what it wrote of, a torch.empty buffer then shows up as NaN / garbage instead
     of as "whatever the previous case left in the caching allocator";
 (c) which side moves?   the oracle in fp32 and in fp64 against the HIP gradients.

usage: fuzz_replay.py [last_case] [seed] ; env MODE=plain|poison (default: both, plain first), FUZZ_ONLY as in fuzz_parity"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import fuzz_parity as F
from oracle import mesm_oracle as O

_orig = dict(empty=torch.empty, empty_like=torch.empty_like, new_empty=torch.Tensor.new_empty)


def _poison(t):
    if t.is_cuda and t.numel():
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype == torch.bool:
            t.fill_(True)
        else:
            t.view(torch.uint8).fill_(0x7F) if t.is_contiguous() else None
    return t


def poison_on():
    torch.empty = lambda *a, **k: _poison(_orig["empty"](*a, **k))
    torch.empty_like = lambda *a, **k: _poison(_orig["empty_like"](*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: _poison(_orig["new_empty"](self, *a, **k))


def poison_off():
    torch.empty, torch.empty_like, torch.Tensor.new_empty = _orig["empty"], _orig["empty_like"], _orig["new_empty"]


def oracle64(sd, args, batch, neg, masked):
    try:
        return O.train_step64(sd, dict(vars(args)), batch, neg, masked)
    except Exception as e:  # noqa: BLE001
        print("   (fp64 oracle not available: %s: %s)" % (type(e).__name__, str(e)[:120]))
        return None


def worst(ga, gb):
    w = (0.0, "-")
    for k, g in gb.items():
        if k in ga:
            e = F.l2(ga[k], g)
            if not e <= w[0]:
                w = (e, k)
    return w


def replay(last, seed, mode):
    rng = random.Random(seed)
    bad = 0
    for case in range(last + 1):
        tag, spec = F.draw(rng, case)
        if F.ONLY and case not in F.ONLY:
            continue
        if mode == "poison":
            poison_on()
        try:
            args, model, crit, batch, neg, masked = F.build(spec)
            out, losses, total, grads = F.hip_step(model, crit, batch, spec["dataset"], neg, masked)
            g1 = {k: v.detach().clone() for k, v in grads.items()}
            t1 = float(total.detach())
            out, losses, total, grads = F.hip_step(model, crit, batch, spec["dataset"], neg, masked)
            g2 = {k: v.detach().clone() for k, v in grads.items()}
            t2 = float(total.detach())
        finally:
            poison_off()
        nan = sorted(k for k, v in g1.items() if not bool(torch.isfinite(v).all()))
        w12 = worst(g1, g2)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o32 = O.train_step(sd, dict(vars(args)), batch, neg, masked)
        o64 = oracle64(sd, args, batch, neg, masked)
        w32 = worst(g1, o32[3])
        line = "%s\n   [%s] total %.6f / %.6f (oracle32 %.6f%s)  hip-vs-hip worst grad L2 %.2e (%s)  hip-vs-oracle32 %.2e (%s)" % (
            tag, mode, t1, t2, float(o32[2]), "" if o64 is None else ", oracle64 %.6f" % float(o64[2]), w12[0], w12[1], w32[0], w32[1])
        w64 = None
        if o64 is not None:
            w64 = worst(g1, {k: v.float() for k, v in o64[3].items()})
            wo = worst({k: v for k, v in o32[3].items()}, {k: v.float() for k, v in o64[3].items()})
            line += "  hip-vs-oracle64 %.2e (%s)  oracle32-vs-oracle64 %.2e (%s)" % (w64[0], w64[1], wo[0], wo[1])
        if nan:
            line += "\n   NON-FINITE gradients: %s" % nan[:8]
        # the fp64 oracle is the referee of a gradient the fp32 oracle and the device disagree on (activation kinks)
        off = not w32[0] < 5e-3 and (w64 is None or not w64[0] < 5e-3)
        flag = bool(nan) or not w12[0] < 1e-4 or off or abs(t1 - t2) > 1e-5 * max(1.0, abs(t1))
        if not w32[0] < 5e-3 and not off:
            line += "\n   (fp32 oracle flipped an activation kink: the device agrees with the fp64 oracle)"
        if flag:
            bad += 1
            line += "\n   ^^^ FLAGGED"
            print("   dumped:", F.dump_case(case, spec, sd, batch, neg, masked, g1, o32[3],
                                             os.path.join(ROOT, "gpurun_out", "fuzz_replay_%s_case%d.pt" % (mode, case))))
        print(line, flush=True)
    print("mode %s: cases 0..%d, flagged %d" % (mode, last, bad), flush=True)
    return bad


if __name__ == "__main__":
    last = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
    modes = [os.environ["MODE"]] if os.environ.get("MODE") else ["plain", "poison"]
    tot = 0
    for m in modes:
        tot += replay(last, seed, m)
    sys.exit(1 if tot else 0)
