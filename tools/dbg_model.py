import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_model_gpu import run_step, rel
from mesm_amd import build_criterion, build_model, synthetic
from oracle import mesm_oracle as O
dataset, groups, Lv, Lw, ragged = "qvhighlights", [2, 1, 3, 2], 75, 32, True
over = dict(dataset_name=dataset, v_feat_dim=130, t_feat_dim=64, vocab_size=301, share_MLP=True,
            set_cost_class=4, loss_label_coef=4, rank_coef=12, use_triplet=True, loss_recfw_coef=0.5,
            loss_recss_coef=0.1, max_video_l=Lv, max_words_l=Lw, device="cuda:0")
args = synthetic.make_args(None, **over)
torch.manual_seed(5)
model = build_model(args)
with torch.no_grad():
    for n_, p in model.named_parameters():
        if n_.endswith("_token") or "masked_sent_token" in n_:
            p.normal_(0, 0.5)
crit = build_criterion(args)
batch = synthetic.make_batch(dataset, groups, Lv, Lw, 130, 64, 302, seed=3, ragged=ragged)
neg, masked = synthetic.host_draws(batch, seed=3)
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
o_out, o_losses, o_total, o_grads, o_idx = O.train_step(sd, dict(vars(args)), batch, neg, masked)
from mesm_amd import synthetic as syn
def step(fwd_tile, bwd_tile):
    b = syn.to_device(batch, torch.device("cuda:0"))
    model.eval()
    os.environ["MESM_GEMM_TILE"] = fwd_tile
    out = model(**b, dataset_name=dataset, is_training=True, neg_index=neg, masked_words=masked)
    losses, total = crit(out, b, True)
    model.zero_grad()
    torch.cuda.synchronize()
    os.environ["MESM_GEMM_TILE"] = bwd_tile
    total.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    worst = sorted(((rel(grads[k], g), k) for k, g in o_grads.items()), reverse=True)
    print("fwd", fwd_tile, "bwd", bwd_tile, ["%.1e %s" % (e, k) for e, k in worst[:2]], "total", float(total), float(o_total))
from mesm_amd import kernels as kn
zs = {}
orig = kn.gemm
def make(tag):
    def spy(A, B, C, **kw):
        r = orig(A, B, C, **kw)
        if C.shape[1] == 1024 and kw.get("bias") is not None and kw.get("trans_b"):
            zs.setdefault(tag, []).append(C.clone())
        return r
    return spy
for tag in ("64", "0"):
    kn.gemm = make(tag)
    step(tag, "64")
kn.gemm = orig
for i, (a, b) in enumerate(zip(zs["64"], zs["0"])):
    flip = (a > 0) != (b > 0)
    if flip.any():
        print("ffn call", i, "shape", tuple(a.shape), "sign flips", int(flip.sum()), "|z| at flips", a[flip].abs().tolist()[:5], b[flip].abs().tolist()[:5], "max|da|", float((a-b).abs().max()))
print("done", len(zs["64"]))
