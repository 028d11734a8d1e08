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
out, losses, total = run_step(model, crit, batch, vars(args), neg, masked)
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
o_out, o_losses, o_total, o_grads, o_idx = O.train_step(sd, dict(vars(args)), batch, neg, masked)
grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
worst = sorted(((rel(grads[k], g), k) for k, g in o_grads.items()), reverse=True)
for e, k in worst[:8]:
    print("%.2e %s" % (e, k))
k = sys.argv[1] if len(sys.argv) > 1 else worst[0][1]
a, b = grads[k].detach().cpu().double(), o_grads[k].double()
d = (a - b).abs()
print(k, tuple(a.shape), "max|b|", float(b.abs().max()))
if d.dim() == 2:
    rows = d.max(1)[0]
    top = torch.topk(rows, 5)
    print("row max diffs", top.values.tolist(), top.indices.tolist())
    cols = d.max(0)[0]
    top = torch.topk(cols, 5)
    print("col max diffs", top.values.tolist(), top.indices.tolist())
print("recfw logit rel", rel(out["recfw_words_logit"], o_out["recfw_words_logit"]))
