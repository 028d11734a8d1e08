"""Does the side-stream MLM branch overlap inside a captured graph?  forward only / forward+backward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic
from mesm_amd.criterion import TargetPlan
dev = torch.device("cuda:0")
def run(side, backward):
    args = synthetic.make_args("C3a", device=str(dev))
    torch.manual_seed(1)
    model = build_model(args); crit = build_criterion(args); model.train()
    model.side_streams = side
    batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=0), dev)
    wm = torch.ones(32, 32, dtype=torch.bool)
    from mesm_amd import kernels as kn
    wm = kn.text_prep(batch["words_id"], True)[1].cpu()
    plan = model.make_plan(batch["video_mask"], wm, batch["num_clips"], args.dataset_name, True,
                           words_weight=batch["words_weight"], clip_mask=batch.get("clip_mask"), device=dev)
    batch["_target_plan"] = TargetPlan(batch, crit.multi_clip, dev, crit.gamma)
    model.gradbuf().ensure(dev)
    def body():
        out = model(**batch, dataset_name=args.dataset_name, is_training=True, plan=plan)
        losses, total = crit(out, batch, True)
        if backward:
            model.zero_grad(set_to_none=True)
            total.backward()
        return total.detach()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body(); body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        t = body()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3
for bw in (False, True):
    a = run(False, bw); b = run(True, bw)
    print("backward=%s: single stream %.3f ms, MLM on side stream %.3f ms" % (bw, a, b))
