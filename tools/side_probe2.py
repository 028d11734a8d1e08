"""Bisect: a dummy chain of 60 GEMMs next to the real model forward inside one captured graph,
(a) appended on the main stream, (b) forked onto a side stream right after the first model kernels
(event recorded before the model forward, enqueued after it)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import build_criterion, build_model, synthetic
from mesm_amd import kernels as kn
from mesm_amd.criterion import TargetPlan
dev = torch.device("cuda:0")
args = synthetic.make_args("C3a", device=str(dev))
torch.manual_seed(1)
model = build_model(args); crit = build_criterion(args); model.train()
batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=0), dev)
wm = kn.text_prep(batch["words_id"], True)[1].cpu()
plan = model.make_plan(batch["video_mask"], wm, batch["num_clips"], args.dataset_name, True,
                       words_weight=batch["words_weight"], clip_mask=batch.get("clip_mask"), device=dev)
batch["_target_plan"] = TargetPlan(batch, crit.multi_clip, dev, crit.gamma)
model.gradbuf().ensure(dev)
A = [torch.randn(2400, 256, device=dev) for _ in range(8)]
W = [torch.randn(256, 256, device=dev) for _ in range(8)]
C = [torch.zeros(2400, 256, device=dev) for _ in range(8)]
side = torch.cuda.Stream()
def dummy(n=60):
    for i in range(n): kn.gemm(A[i % 8], W[i % 8], C[i % 8], trans_b=True)
def fwd():
    with torch.no_grad():
        return model(**batch, dataset_name=args.dataset_name, is_training=False, plan=plan)
def body(mode):
    cur = torch.cuda.current_stream()
    if mode == "none":
        fwd(); return
    if mode == "main":
        fwd(); dummy(); return
    if mode == "dummy_only":
        dummy(); return
    ev = torch.cuda.Event(); ev.record(cur)
    fwd()
    side.wait_event(ev)
    with torch.cuda.stream(side): dummy()
    cur.wait_stream(side)
from mesm_amd.sidecall import Branch, JoinGrad, side_call
br = Branch(dev)
def body_sidecall(record):
    cur = torch.cuda.current_stream()
    x = torch.ones(4, device=dev, requires_grad=True)
    br.mark_fork()
    xj = JoinGrad.apply(x, br, 0)
    fwd()
    def fn(t):
        if record:
            t2 = t * 2
        dummy()
        return C[0].sum() * t.sum() if not record else C[0].sum() * t2.sum()
    (y,) = side_call(br, fn, [(0, xj)])
    cur.wait_event(br.done_fwd)
    return y
def run(mode):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    fn = (lambda: body_sidecall(mode == "sidecall+op")) if mode.startswith("sidecall") else (lambda: body(mode))
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3
for mode in ("none", "dummy_only", "main", "side", "sidecall", "sidecall+op"):
    print("%-11s %.3f ms" % (mode, run(mode)))
