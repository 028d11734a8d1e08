"""40 optimizer steps of C3a (dropout off, fixed batch sequence and host draws) under the GEMM arithmetic of MESM_GEMM_BF16X: prints the loss
trajectory and a parameter checksum, to compare the two-term fp16 split (2) with exact f32 (0) and the bf16 split (6) over a short run.
usage: MESM_GEMM_BF16X=2 python tools/probe/train_modes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mesm_amd import build_criterion, build_model, build_optimizer, kernels as kn, synthetic
kn._FWD_ATOMICS = False
dev = torch.device("cuda:0")
args = synthetic.make_args("C3a", device="cuda:0", lr=1e-4, weight_decay=1e-4, lr_drop=400, gamma=0.1)
torch.manual_seed(3)
model = build_model(args); crit = build_criterion(args); model.train(); model.autograph(False)
for m in model.modules():
    if hasattr(m, "p") and isinstance(m.p, float):
        m.p = 0.0
opt, _ = build_optimizer(args, model)
losses = []
for step in range(40):
    cpu = synthetic.workload_batch("C3a", seed=100 + step % 8, ragged=True)
    neg, mw = synthetic.host_draws(cpu, seed=step)
    b = synthetic.to_device(cpu, dev)
    out = model(**b, dataset_name=args.dataset_name, is_training=True, neg_index=neg, masked_words=mw)
    _, loss = crit(out, b, True)
    opt.zero_grad(); loss.backward(); opt.step(grad_clip=0.1)
    losses.append(float(loss))
fp = model.flat_params().double()
print("mode", kn.gemm_mode(), "losses", " ".join("%.5f" % l for l in losses[::5] + losses[-1:]))
print("mode", kn.gemm_mode(), "param checksum %.10e  sumsq %.10e" % (float(fp.sum()), float((fp * fp).sum())))
np.save("gpurun_out/train_mode_%d.npy" % kn.gemm_mode(), np.array(losses))
