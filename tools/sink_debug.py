"""Which post-norm blocks take mask*dY from their LayerNorm (ops.DropSink) in one eager C3a step, and why
the others do not (T2V layers: the block output also feeds the FFN residual, so its gradient is a sum)."""
import sys, collections
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import build_criterion, build_model, synthetic, ops
dev = torch.device("cuda:0")
args = synthetic.make_args("C3a", device=str(dev))
torch.manual_seed(0)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=0), dev)
cnt = collections.Counter()
orig = ops._masked_dy
def dbg(sink, dy2, out_drop):
    if sink is None: why = "no sink"
    elif sink.dz is None: why = "sink never filled"
    elif sink.src.data_ptr() != dy2.data_ptr(): why = "different tensor"
    else: why = "hand-over"
    cnt[(why, tuple(dy2.shape))] += 1
    return orig(sink, dy2, out_drop)
ops._masked_dy = dbg
out = model(**batch, dataset_name=args.dataset_name, is_training=True)
losses, total = crit(out, batch, True)
total.backward()
torch.cuda.synchronize()
for k, v in sorted(cnt.items()): print(v, k)
