"""Random-configuration sweep of graph replay behind the unchanged caller (mesm_amd/autograph.py): for every drawn configuration
(the draws of tools/fuzz_parity.py: dataset, group sizes, lengths, widths / heads, layer counts, projection depth, the ablation
switches) the reference's loop body (train.py:64-72) runs on three batches of that shape with torch's own AdamW -- first visit
eager, then capture + replays -- and every replayed step is compared with the EAGER step on the same batch and host draws:
loss, every loss entry, the gradient buffer (relative L2), the set of parameters that got a gradient.  The forward's K-split
products are off (deterministic forward), dropout is off.  usage: fuzz_autograph.py [n_cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from torch import nn
import fuzz_parity as F
from mesm_amd import kernels as kn, synthetic

dev = torch.device("cuda:0")
kn._FWD_ATOMICS = False


def eager_on(model, crit, batch, name, neg, mw):
    kw = dict(neg_index=torch.as_tensor(neg))
    if mw is not None:
        kw["masked_words"] = torch.as_tensor(mw)
    out = model(**batch, dataset_name=name, is_training=True, **kw)
    losses, total = crit(out, batch, True)
    model.zero_grad(set_to_none=True)
    total.backward()
    torch.cuda.synchronize()
    return {k: float(v) for k, v in losses.items()}, float(total.detach()), model.gradbuf().flat.clone(), \
        [p.grad is not None for p in model.gradbuf().params]


def case(rng, i):
    tag, spec = F.draw(rng, i)
    try:
        args, model, crit, batch0, _, _ = F.build(spec)
        model.train(); crit.train()
        for m in model.modules():
            if hasattr(m, "p") and isinstance(m.p, float):
                m.p = 0.0
        model.autograph(True)
        name = spec["dataset"]
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
        over = spec["over"]
        errs, replayed = [], 0
        for step in range(3):
            b = synthetic.make_batch(name, spec["groups"], spec["Lv"], spec["Lw"], over["v_feat_dim"], over["t_feat_dim"],
                                     over["vocab_size"] + 1, seed=spec["seed"] + 17 * step, ragged=spec["ragged"])
            batch = synthetic.to_device(b, dev)
            out = model(**batch, dataset_name=name, is_training=True)
            ld, loss = crit(out, batch, is_training=True)
            opt.zero_grad()
            loss.backward()
            torch.cuda.synchronize()
            if out._auto_step is not None:
                replayed += 1
                flat_g = model.gradbuf().flat.clone()
                had_g = [p.grad is not None for p in model.gradbuf().params]
                ld_g, tot_g = {k: float(v) for k, v in ld.items()}, float(loss.detach())
                neg, mw = out._auto_step._draws
                ld_e, tot_e, flat_e, had_e = eager_on(model, crit, batch, name, neg, mw)
                if not abs(tot_e - tot_g) < 1e-5 * max(1.0, abs(tot_e)):
                    errs.append("step %d total %.7f vs eager %.7f" % (step, tot_g, tot_e))
                for k in ld_e:
                    if not abs(ld_e[k] - ld_g.get(k, float("nan"))) < 1e-5 * max(1.0, abs(ld_e[k])):
                        errs.append("step %d %s %.6g vs %.6g" % (step, k, ld_g.get(k, float("nan")), ld_e[k]))
                r = float((flat_e - flat_g).norm()) / max(float(flat_e.norm()), 1e-6)
                if not r < 1e-4:
                    errs.append("step %d gradient rel L2 %.2e" % (step, r))
                if had_e != had_g:
                    errs.append("step %d: other set of parameters with gradients" % step)
                model.gradbuf().flat.copy_(flat_g)
            nn.utils.clip_grad_norm_(model.parameters(), 0.1)
            opt.step()
        if replayed != 2:
            errs.append("replayed %d of 3 steps (captures %d, bad %s)" % (replayed, model._auto.captures, list(model._auto.bad)[:1]))
        return tag, "ok" if not errs else "MISMATCH " + "; ".join(errs[:5])
    except Exception as e:  # noqa: BLE001
        import traceback
        return tag, "ERROR %s: %s | %s" % (type(e).__name__, str(e)[:200], traceback.format_exc().splitlines()[-3].strip()[:160])


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for i in range(n):
        tag, status = case(rng, i)
        print("%s -> %s" % (tag, status), flush=True)
        bad += status != "ok"
    print("cases %d, not ok %d" % (n, bad))
