"""Criterion + HungarianMatcher (model/criterion.py, model/matcher.py of the reference) with
the per-pair reductions on the gfx950 kernels and no host synchronisation inside forward:
matching cost + assignment run on the device (mesm_match), the saliency losses and the
masked-LM NLL are fused kernels, and every index the losses need is a device tensor.
"""
import torch
from torch import nn

from . import kernels as kn
from . import ops


def span_cxw_to_xx(s):
    return torch.stack([s[..., 0] - 0.5 * s[..., 1], s[..., 0] + 0.5 * s[..., 1]], dim=-1)


def span_xx_to_cxw(s):
    return torch.stack([s.sum(-1) * 0.5, s[..., 1] - s[..., 0]], dim=-1)


def generalized_temporal_iou(a, b):
    """utils/span_utils.py:92-121 (pairwise (len(a), len(b)) matrix)."""
    a, b = a.float(), b.float()
    inter = (torch.min(a[:, None, 1], b[:, 1]) - torch.max(a[:, None, 0], b[:, 0])).clamp(min=0)
    union = (a[:, 1] - a[:, 0])[:, None] + (b[:, 1] - b[:, 0]) - inter
    enc = (torch.max(a[:, None, 1], b[:, 1]) - torch.min(a[:, None, 0], b[:, 0])).clamp(min=0)
    return inter / union - (enc - union) / enc


def paired_giou(a, b):
    """diag(generalized_temporal_iou(a, b)) without the full matrix."""
    inter = (torch.min(a[:, 1], b[:, 1]) - torch.max(a[:, 0], b[:, 0])).clamp(min=0)
    union = (a[:, 1] - a[:, 0]) + (b[:, 1] - b[:, 0]) - inter
    enc = (torch.max(a[:, 1], b[:, 1]) - torch.min(a[:, 0], b[:, 0])).clamp(min=0)
    return inter / union - (enc - union) / enc


class TargetPlan:
    """Flattened targets + host-built index tensors (built once per batch, reused for the aux
    decoder layers): tgt_cxw / tgt_xx (sumT, 2), tgt_off (N+1) int32, pair_of_t (sumT) int64,
    Tmax, group_mask (N, N) bool (True where two pairs share a video group)."""

    def __init__(self, targets, multi_clip, device):
        if multi_clip:
            sizes = [len(t["spans"]) for t in targets["norm_span"]]
            self.tgt_cxw = torch.cat([t["spans"] for t in targets["norm_span"]]).to(device).float().contiguous()
            self.tgt_xx = torch.cat([t["moments"] for t in targets["norm_moment"]]).to(device).float().contiguous()
        else:
            self.tgt_cxw = targets["norm_span"].to(device).float().contiguous()
            self.tgt_xx = targets["norm_moment"].to(device).float().contiguous()
            sizes = [1] * self.tgt_cxw.shape[0]
        self.sizes = sizes
        self.N = len(sizes)
        self.Tmax = max(sizes)
        off = [0]
        for s in sizes:
            off.append(off[-1] + s)
        self.tgt_off = torch.tensor(off, dtype=torch.int32, device=device)
        self.pair_of_t = torch.repeat_interleave(torch.arange(self.N), torch.tensor(sizes)).to(device)
        nc = targets["num_clips"]
        groups = [int(g) for g in nc.tolist()]
        gid = torch.repeat_interleave(torch.arange(len(groups)), torch.tensor(groups))
        self.group_mask = (gid[:, None] == gid[None, :]).to(device)


class HungarianMatcher(nn.Module):
    """matcher.py:14-117."""

    def __init__(self, cost_class=1, cost_span=1, cost_giou=1, span_loss_type="l1", max_v_l=75,
                 multi_clip=False):
        super().__init__()
        assert cost_class != 0 or cost_span != 0 or cost_giou != 0, "all costs cant be 0"
        self.cost_class, self.cost_span, self.cost_giou = cost_class, cost_span, cost_giou
        self.span_loss_type = span_loss_type
        self.max_v_l = max_v_l
        self.foreground_label = 0
        self.multi_clip = multi_clip

    @torch.no_grad()
    def match_device(self, outputs, plan):
        """match_q (sumT) int32 on the device: the query assigned to every target."""
        return kn.match(outputs["pred_logits"].detach().contiguous(),
                        outputs["pred_spans"].detach().contiguous(), plan.tgt_cxw, plan.tgt_xx,
                        plan.tgt_off, plan.Tmax, self.cost_span, self.cost_giou, self.cost_class)

    @torch.no_grad()
    def forward(self, outputs, targets):
        """Reference-format result (host tensors; synchronises): list of (query_idx, target_idx)
        sorted by query for multi_clip, else an (N, 2) tensor [query, 0]."""
        plan = TargetPlan(targets, self.multi_clip, outputs["pred_spans"].device)
        mq = self.match_device(outputs, plan).cpu().to(torch.int64)
        if not self.multi_clip:
            return torch.stack([mq, torch.zeros_like(mq)], dim=1)
        res, start = [], 0
        for s in plan.sizes:
            q = mq[start:start + s]
            order = torch.argsort(q)
            res.append((q[order], order.to(torch.int64)))
            start += s
        return res


class Criterion(nn.Module):
    """criterion.py:9-367."""

    def __init__(self, matcher, weight_dict, losses, eos_coef, span_loss_type, max_video_l, rank_coef,
                 use_triplet, saliency_margin=1, multi_clip=False, gamma=0.9, recss_tau=0.5):
        super().__init__()
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.losses = losses
        self.span_loss_type = span_loss_type
        self.max_video_l = max_video_l
        self.saliency_margin = saliency_margin
        self.foreground_label, self.background_label = 0, 1
        self.eos_coef = eos_coef
        empty_weight = torch.ones(2)
        empty_weight[-1] = eos_coef
        self.register_buffer("empty_weight", empty_weight)
        self.rank_coef = rank_coef
        self.use_triplet = use_triplet
        self.rec_ss = "rec_ss" in losses
        self.rec_fw = "rec_fw" in losses
        self.multi_clip = multi_clip
        self.gamma = gamma
        self.recss_tau = recss_tau

    # -- span / label losses on the matched pairs (criterion.py:71-137)
    def loss_spans(self, outputs, plan, match_q):
        spans = outputs["pred_spans"]
        Q = spans.shape[1]
        flat = plan.pair_of_t * Q + match_q.to(torch.int64)
        src = spans.reshape(-1, 2)[flat]
        loss_span = (src - plan.tgt_cxw).abs().mean()
        loss_giou = (1 - paired_giou(span_cxw_to_xx(src), plan.tgt_xx)).mean()
        return {"loss_span": loss_span, "loss_giou": loss_giou}

    def loss_labels(self, outputs, plan, match_q, log=True):
        logits = outputs["pred_logits"]
        N, Q = logits.shape[:2]
        flat = plan.pair_of_t * Q + match_q.to(torch.int64)
        cls = torch.ones(N * Q, dtype=torch.int64, device=logits.device)
        cls.index_fill_(0, flat, self.foreground_label)  # no host scalar copy: graph-capturable
        logp = torch.log_softmax(logits.reshape(N * Q, 2), dim=-1)
        ce = -logp.gather(1, cls[:, None]).squeeze(1) * self.empty_weight[cls]
        losses = {"loss_label": ce.mean()}
        if log:
            picked = logits.detach().reshape(N * Q, 2)[flat]
            acc = (picked.argmax(-1) == self.foreground_label).float().sum() * (100.0 / picked.shape[0])
            losses["class_error"] = 100 - acc
        return losses

    def loss_saliency(self, outputs, targets):
        vmask = targets["video_mask"]
        if "saliency_label" in targets:
            label = targets["saliency_label"]
        else:
            label = targets["clip_mask"]
        label = label.to(torch.float64).contiguous()
        pos_idx = targets["pos_idx"].contiguous() if self.use_triplet else None
        neg_idx = targets["neg_idx"].contiguous() if self.use_triplet else None
        loss = ops.saliency_loss(outputs["saliency_scores"], outputs["neg_saliency_scores"], label,
                                 vmask.contiguous(), pos_idx, neg_idx, float(self.rank_coef),
                                 float(self.saliency_margin))
        return {"loss_saliency": loss}

    def loss_rec_ss(self, outputs, targets, plan):
        """criterion.py:223-274 (ablation 3)."""
        if self.multi_clip:
            idx = plan.pair_of_t[:, None].expand(-1, 1)
            lo = torch.full((plan.N, 1), float("inf"), device=idx.device).scatter_reduce(
                0, idx, plan.tgt_xx.min(1, keepdim=True)[0], "amin")
            hi = torch.full((plan.N, 1), float("-inf"), device=idx.device).scatter_reduce(
                0, idx, plan.tgt_xx.max(1, keepdim=True)[0], "amax")
            mom = torch.cat([lo, hi], dim=1)
        else:
            mom = plan.tgt_xx
        pos = (generalized_temporal_iou(mom, mom) >= self.gamma) & plan.group_mask
        cm = targets["clip_mask"].unsqueeze(-1)
        clip = (outputs["projed_video_feat"] * cm).sum(dim=1) / cm.sum(dim=1)
        wm = outputs["expanded_words_mask"].unsqueeze(-1)
        wf = (outputs["expanded_words_feat"] * wm).sum(dim=1) / wm.sum(dim=1)
        sim = torch.nn.functional.normalize(clip, dim=-1) @ torch.nn.functional.normalize(wf, dim=-1).t()
        sim = sim / self.recss_tau
        lg = sim - sim.max(dim=1, keepdim=True)[0]
        logp = lg - torch.log(torch.exp(lg).sum(1, keepdim=True) + 1e-6)
        loss = -(pos * logp).sum(1) / (pos.sum(1) + 1e-6)
        return {"loss_rec_ss": loss.mean()}

    def loss_rec_fw(self, outputs, targets):
        """criterion.py:276-306."""
        mask = outputs["words_mask"]
        row_loss, correct = ops.nll_smooth(outputs["recfw_words_logit"], targets["words_label"], mask, 0.1)
        acc = (correct.float() * mask).sum() / mask.sum()
        nll = row_loss.sum(dim=-1) / mask.sum(dim=-1)
        return {"loss_rec_fw": nll.mean(), "rec_fw_acc": acc}

    def forward(self, outputs, targets, is_training=True):
        device = outputs["pred_spans"].device
        plan = targets.get("_target_plan")
        if plan is None:
            plan = TargetPlan(targets, self.multi_clip, device)
        losses = {}
        mq = self.matcher.match_device(outputs, plan)
        self.last_match = [mq]
        for loss in self.losses:
            if loss == "span":
                losses.update(self.loss_spans(outputs, plan, mq))
            elif loss == "label":
                losses.update(self.loss_labels(outputs, plan, mq))
            elif loss == "saliency":
                losses.update(self.loss_saliency(outputs, targets))
            elif loss == "rec_fw":
                if is_training:
                    losses.update(self.loss_rec_fw(outputs, targets))
            elif loss == "rec_ss":
                losses.update(self.loss_rec_ss(outputs, targets, plan))
            else:
                raise AssertionError("do you really want to compute %s loss?" % loss)
        for i, aux in enumerate(outputs.get("aux_outputs", [])):
            amq = self.matcher.match_device(aux, plan)
            self.last_match.append(amq)
            for loss in self.losses:
                if loss == "span":
                    l = self.loss_spans(aux, plan, amq)
                elif loss == "label":
                    l = self.loss_labels(aux, plan, amq)
                else:
                    continue
                losses.update({k + "_%d" % i: v for k, v in l.items()})
        total = sum(losses[k] * self.weight_dict[k] for k in losses.keys() if k in self.weight_dict)
        return losses, total
