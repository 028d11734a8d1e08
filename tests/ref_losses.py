"""Plain-torch (autograd) statements of the loss blocks, as the previous unfused criterion had
them; test-only references for the fused kernels of mesm_amd/csrc/criterion.hip."""
import torch


def span_cxw_to_xx(s):
    return torch.stack([s[..., 0] - 0.5 * s[..., 1], s[..., 0] + 0.5 * s[..., 1]], dim=-1)


def generalized_temporal_iou(a, b):
    a, b = a.float(), b.float()
    inter = (torch.min(a[:, None, 1], b[:, 1]) - torch.max(a[:, None, 0], b[:, 0])).clamp(min=0)
    union = (a[:, 1] - a[:, 0])[:, None] + (b[:, 1] - b[:, 0]) - inter
    enc = (torch.max(a[:, None, 1], b[:, 1]) - torch.min(a[:, None, 0], b[:, 0])).clamp(min=0)
    return inter / union - (enc - union) / enc


def paired_giou(a, b):
    inter = (torch.min(a[:, 1], b[:, 1]) - torch.max(a[:, 0], b[:, 0])).clamp(min=0)
    union = (a[:, 1] - a[:, 0]) + (b[:, 1] - b[:, 0]) - inter
    enc = (torch.max(a[:, 1], b[:, 1]) - torch.min(a[:, 0], b[:, 0])).clamp(min=0)
    return inter / union - (enc - union) / enc


def set_losses(logits, spans, tgt_cxw, tgt_xx, pair_of_t, match_q, eos_coef):
    """criterion.py:71-137 given the matching -> (loss_span, loss_giou, loss_label, class_error)."""
    N, Q = logits.shape[:2]
    keep = match_q >= 0  # a pair with more targets than queries leaves targets unmatched (matcher.py:108-117)
    flat = (pair_of_t * Q + match_q.to(torch.int64))[keep]
    tgt_cxw, tgt_xx = tgt_cxw[keep], tgt_xx[keep]
    src = spans.reshape(-1, 2)[flat]
    loss_span = (src - tgt_cxw).abs().mean()
    loss_giou = (1 - paired_giou(span_cxw_to_xx(src), tgt_xx)).mean()
    cls = torch.ones(N * Q, dtype=torch.int64, device=logits.device)
    cls[flat] = 0
    w = torch.tensor([1.0, eos_coef], device=logits.device)
    logp = torch.log_softmax(logits.reshape(N * Q, 2), dim=-1)
    ce = -logp.gather(1, cls[:, None]).squeeze(1) * w[cls]
    picked = logits.detach().reshape(N * Q, 2)[flat]
    acc = (picked.argmax(-1) == 0).float().sum() * (100.0 / picked.shape[0])
    return loss_span, loss_giou, ce.mean(), 100 - acc


def rec_ss(pv, cmask, ew, wmask, pos, tau):
    """criterion.py:240-273."""
    cm = cmask.unsqueeze(-1)
    clip = (pv * cm).sum(dim=1) / cm.sum(dim=1)
    wm = wmask.unsqueeze(-1)
    wf = (ew * wm).sum(dim=1) / wm.sum(dim=1)
    sim = torch.nn.functional.normalize(clip, dim=-1) @ torch.nn.functional.normalize(wf, dim=-1).t()
    sim = sim / tau
    lg = sim - sim.max(dim=1, keepdim=True)[0]
    logp = lg - torch.log(torch.exp(lg).sum(1, keepdim=True) + 1e-6)
    loss = -(pos * logp).sum(1) / (pos.sum(1) + 1e-6)
    return loss.mean()


def rec_fw(logit, label, mask, eps=0.1):
    """criterion.py:291-306."""
    acc = (logit.max(dim=-1)[1] == label).float()
    mean_acc = (acc * mask).sum() / mask.sum()
    lp = logit.log_softmax(dim=-1)
    nll = -lp.gather(dim=-1, index=label.unsqueeze(-1)).squeeze(-1)
    smooth = -lp.sum(dim=-1)
    nll = (1 - eps) * nll + eps / lp.size(-1) * smooth
    nll = nll.masked_fill(mask == 0, 0)
    nll = nll.sum(dim=-1) / mask.sum(dim=-1)
    return nll.mean(), mean_acc


def post_process_text(x, normalize=True):
    """model.py:145-152."""
    F = torch.nn.functional
    w = F.normalize(x, dim=-1, p=2, eps=1e-5) if normalize else x
    m = w.sum(dim=-1) != 0
    s = w.sum(dim=1) / m.sum(dim=1).unsqueeze(-1)
    if normalize:
        s = F.normalize(s, dim=-1, p=2, eps=1e-5)
    return w, m, s
