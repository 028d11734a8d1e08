"""Criterion + HungarianMatcher (model/criterion.py, model/matcher.py of the reference) as ONE
autograd block on the gfx950 kernels: every loss is one forward and one backward launch
(csrc/criterion.hip, csrc/losses.hip), the matching runs on the device inside the set-loss
kernel, all indices the losses need are device tensors, and nothing synchronises with the host.
"""
import torch
from torch import nn
from torch.autograd import Function

from . import kernels as kn
from .autograph import AutoOutputs


def span_cxw_to_xx(s):
    return torch.stack([s[..., 0] - 0.5 * s[..., 1], s[..., 0] + 0.5 * s[..., 1]], dim=-1)


def span_xx_to_cxw(s):
    return torch.stack([s.sum(-1) * 0.5, s[..., 1] - s[..., 0]], dim=-1)


def generalized_temporal_iou(a, b):
    """utils/span_utils.py:92-121 (pairwise (len(a), len(b)) matrix); host-side use only."""
    a, b = a.float(), b.float()
    inter = (torch.min(a[:, None, 1], b[:, 1]) - torch.max(a[:, None, 0], b[:, 0])).clamp(min=0)
    union = (a[:, 1] - a[:, 0])[:, None] + (b[:, 1] - b[:, 0]) - inter
    enc = (torch.max(a[:, None, 1], b[:, 1]) - torch.min(a[:, None, 0], b[:, 0])).clamp(min=0)
    return inter / union - (enc - union) / enc


class _LossPack:
    """host copy of a step's loss vector, fetched ONCE: the reference's loop reads every entry of the loss dict with float()
    (train.py:75-77: 14 device synchronisations per step); the entries handed out below share this pack, so the first
    float() / .item() costs one transfer + one event wait and the others are host reads"""
    __slots__ = ("vec", "vals")

    def __init__(self, vec):
        self.vec, self.vals = vec, None

    def host(self):
        if self.vals is None:
            v = self.vec
            if v.is_cuda:
                buf = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                buf.copy_(v, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(v.device))
                ev.synchronize()
                v = buf
            self.vals = v.tolist()
        return self.vals


class LossEntry(torch.Tensor):
    """one element of the loss vector: a tensor like any other (every torch operation returns a plain tensor), whose float() /
    .item() read the shared host copy"""
    __torch_function__ = torch._C._disabled_torch_function_impl

    @staticmethod
    def __new__(cls, t, pack, i):
        r = torch.Tensor._make_subclass(cls, t, False)
        r._pack, r._i = pack, i
        return r

    def item(self):
        return self._pack.host()[self._i]

    def __float__(self):
        return float(self._pack.host()[self._i])


def loss_entries(lv, index):
    """{name: LossEntry} over the 1-D, gradient-free loss vector `lv`; index = {name: position}"""
    pack = _LossPack(lv)
    return {n: LossEntry(lv[i], pack, i) for n, i in index.items()}


class TargetPlan:
    """Flattened targets + host-built index tensors (built once per batch, reused for the aux
    decoder layers): tgt_cxw / tgt_xx (sumT, 2), tgt_off (N+1) int32, Tmax, group_mask (N, N) bool (True where two
    pairs share a video group) and ss_pos (N, N) uint8, the positives of loss_rec_ss (criterion.py:224-238:
    block-diagonal gIoU of the merged moments >= gamma) -- target-only, so it is evaluated on the host in fp32.
    Everything reaches the device in ONE transfer (arena.Arena); `arrays()` is the host half, which
    graphed.GraphedStep.load_batch re-runs for every new batch."""

    @staticmethod
    def arrays(targets, multi_clip, gamma=0.9, T_cap=None, Tmax_cap=None):
        """-> ({name: np.ndarray}, meta).  T_cap / Tmax_cap: pad the flattened targets to T_cap rows and report
        Tmax_cap as the matcher's work-array extent (a captured HIP graph bakes both in, graphed.py); the kernels
        read the true counts from tgt_off, so padding changes no result."""
        import numpy as np
        if multi_clip:
            spans, moms = [t["spans"] for t in targets["norm_span"]], [t["moments"] for t in targets["norm_moment"]]
            sizes = [int(t.shape[0]) for t in spans]
            cxw = torch.cat(spans).float().cpu().numpy()
            xx = torch.cat(moms).float().cpu().numpy()
            if min(sizes) > 0 and [int(t.shape[0]) for t in moms] == sizes:
                # the merged moment of every pair, [min, max] over its windows (criterion.py:226-229): segment reductions
                # (32 pairs x 3 small tensor ops cost 0.4 ms on the host path of every replayed step)
                first = np.concatenate([[0], np.cumsum(sizes)[:-1]])
                mom = np.stack([np.minimum.reduceat(xx.min(1), first), np.maximum.reduceat(xx.max(1), first)], axis=1)
            else:
                mom = torch.stack([torch.stack([t.min(), t.max()]) for t in moms]).float().cpu().numpy()
        else:
            cxw = targets["norm_span"].float().cpu().numpy()
            xx = targets["norm_moment"].float().cpu().numpy()
            sizes = [1] * cxw.shape[0]
            mom = xx
        N, Tmax, sumT = len(sizes), max(sizes), cxw.shape[0]
        if Tmax_cap is not None:
            if Tmax > Tmax_cap:
                raise ValueError("TargetPlan: a pair has %d target windows > Tmax_cap %d" % (Tmax, Tmax_cap))
            Tmax = Tmax_cap
        if T_cap is not None:
            if sumT > T_cap:
                raise ValueError("TargetPlan: %d target windows > T_cap %d" % (sumT, T_cap))
            pad = T_cap - sumT
            cxw = np.concatenate([cxw, np.tile(np.float32([[0.5, 1.0]]), (pad, 1))])
            xx = np.concatenate([xx, np.tile(np.float32([[0.0, 1.0]]), (pad, 1))])
        groups = [int(g) for g in targets["num_clips"].tolist()]
        gid = np.repeat(np.arange(len(groups)), groups)
        gmask = gid[:, None] == gid[None, :]
        # generalized_temporal_iou(mom, mom) >= gamma in fp32, the arithmetic of utils/span_utils.py:92-121 operation for
        # operation (numpy float32 = torch float32 element-wise)
        a = np.ascontiguousarray(mom, dtype=np.float32)
        inter = np.maximum(np.minimum(a[:, None, 1], a[None, :, 1]) - np.maximum(a[:, None, 0], a[None, :, 0]), np.float32(0))
        union = (a[:, 1] - a[:, 0])[:, None] + (a[:, 1] - a[:, 0])[None, :] - inter
        enc = np.maximum(np.maximum(a[:, None, 1], a[None, :, 1]) - np.minimum(a[:, None, 0], a[None, :, 0]), np.float32(0))
        with np.errstate(divide="ignore", invalid="ignore"):
            giou = inter / union - (enc - union) / enc
        ss_pos = ((giou >= np.float32(gamma)) & gmask).astype(np.uint8)
        arr = {"tgt_cxw": np.ascontiguousarray(cxw, dtype=np.float32), "tgt_xx": np.ascontiguousarray(xx, dtype=np.float32),
               "tgt_off": np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32), "group_mask": gmask, "ss_pos": ss_pos}
        return arr, {"sizes": sizes, "N": N, "Tmax": Tmax, "sumT": sumT}

    def __init__(self, targets, multi_clip, device, gamma=0.9, T_cap=None, Tmax_cap=None):
        from .arena import Arena
        arr, meta = self.arrays(targets, multi_clip, gamma, T_cap, Tmax_cap)
        self.arena = Arena(arr, device)
        self.adopt(self.arena.views, meta)

    def adopt(self, views, meta):
        for k, v in views.items():
            setattr(self, k, v)
        self.sizes, self.N, self.Tmax, self.sumT = meta["sizes"], meta["N"], meta["Tmax"], meta["sumT"]


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
        """match_q (sumT) int32 on the device: the query assigned to every target (-1: none, T > Q only)."""
        return kn.match(outputs["pred_logits"].detach().contiguous(),
                        outputs["pred_spans"].detach().contiguous(), plan.tgt_cxw, plan.tgt_xx,
                        plan.tgt_off, plan.Tmax, self.cost_span, self.cost_giou, self.cost_class)

    def to_reference_format(self, mq, plan):
        """device match_q -> the reference's return value (host tensors; synchronises): list of
        (query_idx, target_idx) sorted by query for multi_clip, else an (N, 2) tensor [query, 0]."""
        mq = mq.cpu().to(torch.int64)
        if not self.multi_clip:
            return torch.stack([mq, torch.zeros_like(mq)], dim=1)
        res, start = [], 0
        for s in plan.sizes:
            q = mq[start:start + s]
            order = torch.argsort(q)
            order = order[q[order] >= 0]  # targets left unmatched (a pair with more targets than queries)
            res.append((q[order], order.to(torch.int64)))
            start += s
        return res

    @torch.no_grad()
    def forward(self, outputs, targets):
        plan = TargetPlan(targets, self.multi_clip, outputs["pred_spans"].device)
        return self.to_reference_format(self.match_device(outputs, plan), plan)


class _Spec:
    """Static description of one criterion call: which blocks run, where their values sit in the
    loss vector, and the non-differentiable side inputs."""
    pass


class CriterionFn(Function):
    """(total, loss_vector) = criterion(model outputs).  Forward: three launches (mesm_criterion_fwd: every block's first
    stage as workgroup ranges of one grid, rec_ss' similarity rows, one finishing workgroup with the weighted sum), the
    blocks' values written straight into their slots of the loss vector.  Backward: ONE launch, the blocks'
    gradient kernels as workgroup ranges of one grid (mesm_criterion_bwd)."""

    @staticmethod
    def forward(ctx, spec, *t):
        ctx.set_materialize_grads(False)  # (the engine would zero-fill a gradient for the loss vector: one launch)
        c, plan = spec.crit, spec.plan
        m = c.matcher
        dev = t[0].device
        lv = torch.empty(len(spec.names), device=dev, dtype=torch.float32)
        saved = {}
        blocks = {}
        N = None
        lay = []
        for li, (il, isp, slot, k) in enumerate(spec.set_layers):
            # k: this layer's index in the stacked (layers, N, Q, 2) decoder outputs (None: a tensor of its own)
            logits, spans = (t[il].contiguous(), t[isp].contiguous()) if k is None else (t[il][k], t[isp][k])
            lay.append((logits, spans, slot))
            N = logits.shape[0]
        mqs = []
        for i0 in range(0, len(lay), 8):  # (one launch holds 8 layers)
            part = dict(Q=lay[0][0].shape[1], Tmax=plan.Tmax, w_span=m.cost_span, w_giou=m.cost_giou, w_class=m.cost_class,
                        eos_coef=c.eos_coef, tgt_cxw=plan.tgt_cxw, tgt_xx=plan.tgt_xx, tgt_off=plan.tgt_off,
                        layers=lay[i0:i0 + 8])
            if i0 + 8 < len(lay):
                mqs += kn.set_loss_fwd_layers([(lg, sp, lv[slot:slot + 4]) for lg, sp, slot in part["layers"]], plan.tgt_cxw,
                                              plan.tgt_xx, plan.tgt_off, plan.Tmax, m.cost_span, m.cost_giou, m.cost_class,
                                              c.eos_coef, n_valid=spec.n_valid)
            else:
                blocks["set_losses"] = part
        if spec.sal is not None:
            ip, ineg, slot = spec.sal
            if ineg is None:  # both passes stacked in one (2N, L) tensor
                half = t[ip].shape[0] // 2
                sp, sn = t[ip][:half], t[ip][half:]
            else:
                sp, sn = t[ip].contiguous(), t[ineg].contiguous()
            N = sp.shape[0]
            blocks["sal"] = dict(s_pos=sp, s_neg=sn, label=spec.sal_label, vmask=spec.vmask, pos_idx=spec.pos_idx,
                                 neg_idx=spec.neg_idx, rank_coef=c.rank_coef, margin=c.saliency_margin, slot=slot)
            saved["sal"] = (sp, sn)
        if spec.recfw is not None:
            il, slot = spec.recfw
            logit = t[il].contiguous()
            N = logit.shape[0]
            blocks["recfw"] = dict(logit=logit, label=spec.words_label, mask=spec.words_mask, eps=0.1, slot=slot)
        if spec.recss is not None:
            ipv, iew, slot = spec.recss
            pv, ew = t[ipv].contiguous(), t[iew].contiguous()
            N = pv.shape[0]
            blocks["recss"] = dict(pv=pv, cmask=spec.clip_mask, ew=ew, wmask=spec.ewords_mask, pos=plan.ss_pos,
                                   tau=c.recss_tau, slot=slot)
            saved["recss_shape"] = (pv.shape[1], ew.shape[1])
        # every block's first stage in one grid, rec_ss' similarity rows, one finishing workgroup with the weighted sum
        total, out = kn.criterion_fwd(lv, spec.wv, N, n_valid=spec.n_valid, **blocks)
        mqs += out.get("match", [])
        for li, ((logits, spans, _), mq) in enumerate(zip(lay, mqs)):
            saved["set%d" % li] = (logits, spans, mq)
        if spec.recfw is not None:
            saved["recfw"] = (blocks["recfw"]["logit"], out["row_lse"])
        if spec.recss is not None:
            saved["recss"] = out["recss"]
        ctx.spec, ctx.saved, ctx.n_in = spec, saved, len(t)
        spec.matches = mqs
        ctx.mark_non_differentiable(lv)
        return total, lv

    @staticmethod
    def backward(ctx, g_total, _g_lv):
        spec, saved = ctx.spec, ctx.saved
        c, plan = spec.crit, spec.plan
        grads = [None] * ctx.n_in
        g = g_total.reshape(1).to(torch.float32).contiguous()
        N = None
        stacked = {}
        blocks = {}
        lay = []
        for li, (il, isp, slot, k) in enumerate(spec.set_layers):
            logits, spans, mq = saved["set%d" % li]
            N, Q = logits.shape[0], logits.shape[1]
            if k is None:
                grads[il], grads[isp] = torch.empty_like(logits), torch.empty_like(spans)
                lay.append((logits, spans, mq, grads[il], grads[isp], slot))
                continue
            # every layer writes its slice of ONE gradient of the stacked tensor: no select / stack backward
            # launches.  Layers of the stack the criterion does not read (aux_loss off) keep a zero slice.
            if il not in stacked:
                full = spec.stack_layers == len([1 for x in spec.set_layers if x[0] == il])
                stacked[il] = ((torch.empty_like if full else torch.zeros_like)(spec.stack_base[0]),
                               (torch.empty_like if full else torch.zeros_like)(spec.stack_base[1]))
                grads[il], grads[isp] = stacked[il]
            lay.append((logits, spans, mq, stacked[il][0][k], stacked[il][1][k], slot))
        for i0 in range(0, len(lay), 8):  # (one launch holds 8 layers)
            part = dict(Q=Q, eos_coef=c.eos_coef, tgt_cxw=plan.tgt_cxw, tgt_xx=plan.tgt_xx, tgt_off=plan.tgt_off,
                        layers=lay[i0:i0 + 8])
            if i0 + 8 < len(lay):
                kn.criterion_bwd(g, spec.wv, N, n_valid=spec.n_valid, set_losses=part)
            else:
                blocks["set_losses"] = part
        if spec.sal is not None:
            ip, ineg, slot = spec.sal
            sp, sn = saved["sal"]
            N = sp.shape[0]
            if ineg is None:
                ds = torch.empty(2 * N, sp.shape[1], device=sp.device, dtype=torch.float32)
                grads[ip] = ds
                dsp, dsn = ds[:N], ds[N:]
            else:
                dsp, dsn = torch.empty_like(sp), torch.empty_like(sn)
                grads[ip], grads[ineg] = dsp, dsn
            blocks["sal"] = dict(s_pos=sp, s_neg=sn, label=spec.sal_label, vmask=spec.vmask, pos_idx=spec.pos_idx,
                                 neg_idx=spec.neg_idx, rank_coef=c.rank_coef, margin=c.saliency_margin, ds_pos=dsp,
                                 ds_neg=dsn, slot=slot)
        if spec.recfw is not None:
            il, slot = spec.recfw
            logit, row_lse = saved["recfw"]
            N = logit.shape[0]
            grads[il] = torch.empty_like(logit)
            blocks["recfw"] = dict(logit=logit, label=spec.words_label, row_lse=row_lse, mask=spec.words_mask, eps=0.1,
                                   dlogit=grads[il], slot=slot)
        if spec.recss is not None:
            ipv, iew, slot = spec.recss
            Lv, Le = saved["recss_shape"]
            cn = saved["recss"][0]
            N, D = cn.shape
            grads[ipv] = torch.empty(N, Lv, D, device=cn.device, dtype=torch.float32)
            grads[iew] = torch.empty(N, Le, D, device=cn.device, dtype=torch.float32)
            blocks["recss"] = dict(saved=saved["recss"], pos=plan.ss_pos, cmask=spec.clip_mask, wmask=spec.ewords_mask,
                                   Lv=Lv, Le=Le, tau=c.recss_tau, dpv=grads[ipv], dew=grads[iew], slot=slot)
        if blocks:
            kn.criterion_bwd(g, spec.wv, N, n_valid=spec.n_valid, **blocks)
        return (None,) + tuple(grads)


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
        self._wv_cache = {}
        for loss in losses:
            if loss not in ("span", "label", "saliency", "rec_fw", "rec_ss"):
                raise AssertionError("do you really want to compute %s loss?" % loss)

    def _weights(self, names, device):
        key = (tuple(names), str(device))
        wv = self._wv_cache.get(key)
        if wv is None:
            wv = torch.tensor([float(self.weight_dict.get(n, 0.0)) for n in names], dtype=torch.float32,
                              device=device)
            self._wv_cache[key] = wv
        return wv

    def forward(self, outputs, targets, is_training=True):
        if isinstance(outputs, AutoOutputs) and outputs._mesm_model is not None:
            m = outputs._mesm_model()
            auto = getattr(m, "_auto", None)
            if auto is not None and outputs._auto_step is not None:
                return auto.criterion(self, outputs, targets, is_training)  # replayed forward: replay the criterion graph
            if auto is not None and not auto.busy:
                auto.note_criterion(self)  # eager step: the model learns which criterion its steps are captured with
            side = outputs._mesm_side
            if side is not None and torch.cuda.current_stream(side.device) != side:
                cur = torch.cuda.current_stream(side.device)  # stay on the stream the eager forward ran on (model.py)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    res = self._forward(outputs, targets, is_training)
                cur.wait_stream(side)
                return res
        return self._forward(outputs, targets, is_training)

    def _forward(self, outputs, targets, is_training=True):
        device = outputs["pred_spans"].device
        plan = targets.get("_target_plan")
        if plan is None:
            plan = TargetPlan(targets, self.multi_clip, device, self.gamma)
        spec = _Spec()
        spec.crit, spec.plan = self, plan
        # pairs padded to a captured capacity (batching.pad_pairs): the real count, a device scalar (int32, 1 element)
        spec.n_valid = targets.get("_n_valid")
        if spec.n_valid is None and targets.get("_n_real") is not None:  # eager call on a padded batch
            spec.n_valid = torch.tensor([int(targets["_n_real"])], dtype=torch.int32, device=device)
        names, tensors = [], []

        def add(t):
            tensors.append(t)
            return len(tensors) - 1

        # span + label blocks need each other's matching: they run in one kernel whenever either
        # is requested; a loss that was not requested simply gets no entry in the returned dict.
        want_span, want_label = "span" in self.losses, "label" in self.losses
        spec.set_layers = []
        layers = [(outputs["pred_logits"], outputs["pred_spans"], "")]
        layers += [(a["pred_logits"], a["pred_spans"], "_%d" % i)
                   for i, a in enumerate(outputs.get("aux_outputs", []))]
        hidden = set()
        spec.sal = spec.recfw = spec.recss = None
        # The model hands out per-layer views of the STACKED decoder outputs (model.py:246-258 slices `hs`) and tags
        # them with their base (`_mesm_stack`): when every layer here is such a view of one base pair, the block takes
        # the two stacked tensors and writes one stacked gradient instead of autograd's select / stack backward.
        base = None
        tags = [(getattr(lg, "_mesm_stack", None), getattr(sp_, "_mesm_stack", None)) for lg, sp_, _ in layers]
        if all(a is not None and b is not None and a[1] == b[1] for a, b in tags) \
                and all(a[0] is tags[0][0][0] and b[0] is tags[0][1][0] for a, b in tags) \
                and len({a[1] for a, _ in tags}) == len(tags):
            base = (tags[0][0][0], tags[0][1][0])
            if not (base[0].is_contiguous() and base[1].is_contiguous()):
                base = None
        spec.stack_base = base
        spec.stack_layers = base[0].shape[0] if base is not None else 0
        base_idx = None

        def set_block(logits, spans, suffix):
            nonlocal base_idx
            slot = len(names)
            block = ["loss_span", "loss_giou", "loss_label", "class_error"]
            names.extend(k + suffix for k in block)
            if not want_span:
                hidden.update({"loss_span" + suffix, "loss_giou" + suffix})
            if not want_label:
                hidden.update({"loss_label" + suffix, "class_error" + suffix})
            if base is not None:
                if base_idx is None:
                    base_idx = (add(base[0]), add(base[1]))
                spec.set_layers.append((base_idx[0], base_idx[1], slot, logits._mesm_stack[1]))
            else:
                spec.set_layers.append((add(logits), add(spans), slot, None))

        if want_span or want_label:
            set_block(*layers[0])
        for loss in self.losses:
            if loss == "saliency":
                vmask = targets["video_mask"]
                label = targets["saliency_label"] if "saliency_label" in targets else targets["clip_mask"]
                spec.sal_label = label.to(torch.float64).contiguous()
                spec.vmask = vmask.contiguous()
                spec.pos_idx = targets["pos_idx"].contiguous() if self.use_triplet else None
                spec.neg_idx = targets["neg_idx"].contiguous() if self.use_triplet else None
                sp_, sn_ = outputs["saliency_scores"], outputs["neg_saliency_scores"]
                ta, tb = getattr(sp_, "_mesm_stack", None), getattr(sn_, "_mesm_stack", None)
                if ta is not None and tb is not None and ta[0] is tb[0] and (ta[1], tb[1]) == (0, 1) \
                        and ta[0].is_contiguous():
                    spec.sal = (add(ta[0]), None, len(names))  # both passes as the halves of one tensor
                else:
                    spec.sal = (add(sp_), add(sn_), len(names))
                names.append("loss_saliency")
            elif loss == "rec_fw" and is_training:
                spec.words_label = targets["words_label"].contiguous().view(-1)
                spec.words_mask = outputs["words_mask"].contiguous()
                spec.recfw = (add(outputs["recfw_words_logit"]), len(names))
                names.extend(["loss_rec_fw", "rec_fw_acc"])
            elif loss == "rec_ss":
                spec.clip_mask = targets["clip_mask"].contiguous()
                spec.ewords_mask = outputs["expanded_words_mask"].contiguous()
                spec.recss = (add(outputs["projed_video_feat"]), add(outputs["expanded_words_feat"]),
                              len(names))
                names.append("loss_rec_ss")
        if want_span or want_label:
            for lg, sp, suffix in layers[1:]:
                set_block(lg, sp, suffix)
        spec.names = names
        wnames = [n if n not in hidden else "" for n in names]
        spec.wv = self._weights(wnames, device)
        total, lv = CriterionFn.apply(spec, *tensors)
        self.last_match = spec.matches
        losses = loss_entries(lv, {n: i for i, n in enumerate(names) if n not in hidden})
        return losses, total
