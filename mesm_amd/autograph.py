"""Graph replay behind the UNCHANGED caller (reference train.py:64-72):

    outputs = model(**batch, dataset_name=..., is_training=True)
    loss_dict, loss = criterion(outputs, batch, is_training=True)
    optimizer.zero_grad()
    loss.backward()
    clip_grad_norm_(model.parameters(), ...); optimizer.step()

The eager path issues ~260 launches from Python and is host-bound (17 ms per step where the device needs 3.5).  An
`AutoGraph` hangs off every `MESM` and turns that same call sequence into three HIP-graph replays
(graphed.SplitGraphedStep: forward | criterion | backward over one memory pool) once a batch shape has been seen:

  * first visit of a shape bucket: eager, exactly as before (it is also the step the criterion introduces itself in:
    `Criterion.forward` finds the model on the outputs it is handed);
  * second visit: the bucket is captured (warm-up step + three captures; the caller's gradients are put back as they
    were) and replayed; from then on `model(...)` = host plans + one arena upload + forward replay,
    `criterion(...)` = criterion replay, `loss.backward()` = ONE autograd node whose backward replays the third graph
    and leaves `param.grad` aliasing the flat gradient buffer like the eager backward does.  `optimizer.zero_grad()`
    in between, `clip_grad_norm_`, any `torch.optim` optimizer: unchanged.  Gradients ACCUMULATE when `param.grad` is
    not None at backward time (the caller skipped zero_grad), like autograd's.

What a replayed step cannot do, and says so instead of computing something else: the activations the backward reads are
static memory that the next forward overwrites, so ONE forward may be in flight -- a backward (or criterion call) that
belongs to an older forward raises; the outputs carry no autograd history of their own (they are consumed by this
build's criterion; anything else that wants gradients through them needs the eager path).  `MESM_AUTOGRAPH=0`, or
`model.autograph(False)`, keeps every step eager.

Buckets: key = (dataset, train/eval mode, pairs, clip / word extents, feature dims, group-size bucket, batch keys).
The reference's loaders emit another (pairs, Lv, Lw) almost every batch; `model.autograph(pad=(max_v_l, max_words_l),
pairs=8)` pads every batch like graphed.StepCache does (clips / words to fixed extents, the pair axis to a multiple of
`pairs` with the real count as a device scalar), so that an epoch replays from a handful of graphs.
"""
import os
import warnings
import weakref
from collections import OrderedDict

import torch

from . import kernels as kn


class AutoOutputs(dict):
    """the dict MESM.forward returns, remembering which forward made it (a plain dict to every reader)"""
    _mesm_model = None   # weakref to the model (eager and replayed outputs)
    _auto_step = None    # the SplitGraphedStep whose forward graph wrote these tensors (replayed outputs only)
    _auto_gen = -1
    _auto_batch = None
    _mesm_side = None    # the stream an eager forward ran on when that was not the caller's (eager outputs only)


class _Backward(torch.autograd.Function):
    """loss = f(token): the ONE node of a replayed step; its backward replays the third graph"""

    @staticmethod
    def forward(ctx, token, total, auto, step, gen):
        ctx.auto, ctx.step, ctx.gen = auto, step, gen
        return total.clone()

    @staticmethod
    def backward(ctx, g):
        ctx.auto.backward(ctx.step, ctx.gen, g)
        return None, None, None, None, None


_NP = {torch.float32: "float32", torch.float64: "float64", torch.int64: "int64", torch.int32: "int32",
       torch.bool: "bool", torch.uint8: "uint8", torch.int16: "int16", torch.float16: "float16", torch.int8: "int8"}


class _Fetch:
    """Everything small of a DEVICE batch (prepare_batch_input has moved it there, dataset/base.py:358-384) back on the host
    in ONE transfer: the index plans and the flattened targets are host arithmetic on masks, labels and target windows (a few
    KB), and one `.cpu()` per tensor is one stream synchronisation each -- 77 per C3a batch, 1.2 ms.  The tensors are packed
    by one concatenation kernel (largest element size first, so every piece stays aligned), copied into a pinned buffer and
    cut into numpy views.  The per-pair target lists travel as one tensor per entry; the word-validity mask of pre-extracted
    word features (2 MB of features otherwise) is formed on the device with hostplan.HostSpec.words_mask's arithmetic."""

    def __init__(self):
        self.pinned = None

    def __call__(self, batch, spec, big):
        import numpy as np
        items, lists = [], {}
        for k, v in batch.items():
            if torch.is_tensor(v):
                if v.is_cuda and v.dtype in _NP and v.numel() * v.element_size() <= big:
                    items.append((k, v))
            elif isinstance(v, list) and v and isinstance(v[0], dict):
                for kk in v[0]:
                    ts = [d[kk] for d in v]
                    if all(torch.is_tensor(t) and t.is_cuda for t in ts):
                        lists[(k, kk)] = [int(t.shape[0]) for t in ts]
                        items.append(((k, kk), torch.cat(ts)))
        w = batch["words_id"]
        if w.dim() == 3 and w.is_cuda:
            a = w
            if spec.normalize_txt:
                a = a / a.pow(2).sum(-1, keepdim=True).sqrt().clamp_min(1e-5)
            items.append(("_words_mask", a.sum(-1) != 0))
        items.sort(key=lambda kv: -kv[1].element_size())
        flat = torch.cat([t.contiguous().view(-1).view(torch.uint8) for _, t in items])
        n = flat.numel()
        if self.pinned is None or self.pinned.numel() < n:
            self.pinned = torch.empty(max(2 * n, 1 << 16), dtype=torch.uint8, pin_memory=True)
        self.pinned[:n].copy_(flat, non_blocking=True)
        torch.cuda.current_stream(flat.device).synchronize()
        raw = self.pinned.numpy()[:n].copy()  # (the pinned buffer is reused by the next batch)
        host, off = {}, 0
        for key, t in items:
            nb = t.numel() * t.element_size()
            arr = raw[off:off + nb].view(_NP[t.dtype]).reshape(tuple(t.shape))
            off += nb
            host[key] = torch.from_numpy(arr)
        out = {}
        for k, v in batch.items():
            if torch.is_tensor(v):
                out[k] = host.get(k, v)
            elif isinstance(v, list) and v and isinstance(v[0], dict) and all((k, kk) in lists for kk in v[0]):
                parts = {kk: host[(k, kk)].split(lists[(k, kk)]) for kk in v[0]}
                out[k] = [{kk: parts[kk][i] for kk in v[0]} for i in range(len(v))]
        if "_words_mask" in host:
            out["_words_mask"] = host["_words_mask"]
        return out


def _enabled_default():
    return os.environ.get("MESM_AUTOGRAPH", "1") != "0"


GROUP_CAPS = (1, 2, 3, 5, 9, 16, 32, 64)  # largest-video-group buckets (9 = the QVHighlights maximum, base.py:116-162)


class AutoGraph:
    def __init__(self, model):
        self.model = weakref.ref(model)
        self.enabled = _enabled_default()
        self.crit = None
        self.steps = OrderedDict()  # key -> [SplitGraphedStep]
        self.seen = {}
        self.bad = set()
        self.max_graphs = int(os.environ.get("MESM_AUTOGRAPH_MAX", "16"))
        self.pad = None
        self.pairs = None
        self.gen = 0          # forwards of this model (eager ones included): one may be in flight
        self.busy = False     # inside a capture: the model's forward is being called by the step itself
        self.token = None
        self.captures = self.replays = self.eager = 0
        self.fetch = _Fetch()
        self.spec = None
        self.host_side = 0    # forwards served from the collate's host copies (batch["_host"]) instead of a fetch

    # ------------------------------------------------------------------ configuration
    def configure(self, enabled=None, pad=None, pairs=None, max_graphs=None):
        if enabled is not None:
            self.enabled = bool(enabled)
        if pad is not None:
            self.pad = tuple(pad)
        if pairs is not None:
            self.pairs = int(pairs)
        if max_graphs is not None:
            self.max_graphs = int(max_graphs)
        return self

    def note_criterion(self, crit):
        if self.crit is None or self.crit() is not crit:
            self.crit = weakref.ref(crit)
            if self.steps:  # graphs captured with another criterion object: its weights / losses are baked in
                self.steps.clear()
                self.seen.clear()

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _with_host_side(batch, spec):
        """the batch with its small tensors replaced by the host copies the collate kept (batching.attach_host_side) -> a dict
        that serves both as the shaped batch (big device tensors) and as the host half (everything else), or None when
        `_host` is absent / does not mirror the batch"""
        hs = batch.get("_host")
        if not isinstance(hs, dict):
            return None
        work = {}
        for k, v in batch.items():
            if k == "_host":
                continue
            if torch.is_tensor(v) and v.is_cuda and k in hs:
                h = hs[k]
                if not torch.is_tensor(h) or h.is_cuda or tuple(h.shape) != tuple(v.shape) or h.dtype != v.dtype:
                    return None
                work[k] = h
            elif isinstance(v, list) and v and isinstance(v[0], dict) and k in hs:
                if len(hs[k]) != len(v):
                    return None
                work[k] = hs[k]
            elif torch.is_tensor(v) and v.is_cuda and v.numel() * v.element_size() <= (1 << 20):
                return None  # a small device tensor the collate did not keep: fetch instead
            else:
                work[k] = v
        w = batch["words_id"]
        if w.dim() == 3:
            m = hs.get("_words_mask_norm" if spec.normalize_txt else "_words_mask_raw")
            if m is None or tuple(m.shape) != tuple(w.shape[:2]):
                return None
            work["_words_mask"] = m
        return work

    def _shaped(self, batch):
        """the batch padded like StepCache pads it (device tensors stay on the device)"""
        if self.pad is not None and "_words_mask" in batch:
            from .graphed import StepCache
            batch = dict(batch)
            batch["_words_mask"] = StepCache._pad_dim1(batch["_words_mask"], self.pad[1])
        if self.pad is not None:
            from .graphed import StepCache
            Lv, Lw = self.pad
            if batch["video_feat"].shape[1] > Lv or batch["words_id"].shape[1] > Lw:
                return None
            b = dict(batch)
            for k in StepCache.CLIP_KEYS:
                if k in b:
                    b[k] = StepCache._pad_dim1(b[k], Lv)
            for k in StepCache.WORD_KEYS:
                if k in b:
                    b[k] = StepCache._pad_dim1(b[k], Lw)
            batch = b
        if self.pairs:
            from .batching import pad_pairs
            n = batch["video_feat"].shape[0]
            batch = pad_pairs(batch, (n + self.pairs - 1) // self.pairs * self.pairs)
        return batch

    @staticmethod
    def _key(model, batch, dataset_name, num_clips=None):
        n = batch["video_mask"].shape[0]
        gmax = int((batch["num_clips"] if num_clips is None else num_clips).max())
        gcap = next((c for c in GROUP_CAPS if c >= gmax), gmax)
        sig = tuple(sorted(k for k, v in batch.items() if not k.startswith("_")))
        return (dataset_name, bool(model.training), n, tuple(batch["video_feat"].shape[1:]),
                tuple(batch["words_id"].shape[1:]), gcap, "_n_real" in batch, sig), gcap

    def eligible(self, model, batch, kwargs):
        return (self.enabled and not self.busy and kwargs.get("is_training") is True and kwargs.get("plan") is None
                and torch.is_grad_enabled() and batch["video_feat"].is_cuda
                and kwargs.get("neg_index") is None and kwargs.get("masked_words") is None
                and not torch.cuda.is_current_stream_capturing())

    def eager_stream(self, dev):
        """the stream an EAGER grad-enabled forward of a caller on the default stream runs on (None: where the caller is)"""
        if not self.enabled or self.busy or not torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():
            return None
        if torch.cuda.current_stream(dev) != torch.cuda.default_stream(dev):
            return None
        from .graphed import capture_stream
        return capture_stream(dev)

    def forward(self, batch, dataset_name):
        """-> AutoOutputs of a replayed forward, or None: run this forward eagerly"""
        model = self.model()
        self.gen += 1
        if model.gradbuf().on_ready is not None:
            # a ddp.GradReducer is hooked into the gradient buffer: its collectives belong in the one-graph step that was
            # built WITH the reducer (GraphedStep / StepCache(reducer=...)), not in a capture made behind its back
            return None
        crit = self.crit() if self.crit is not None else None
        try:
            host = None
            if crit is not None and (self.spec is None or self.spec.dataset_name != dataset_name):
                from .hostplan import HostSpec
                self.spec = HostSpec.from_model(model, crit, dataset_name)
            work = self._with_host_side(batch, self.spec) if crit is not None else None
            if work is not None:
                # the collate kept the host copies: no transfer back, no synchronisation in front of the forward launch
                shaped = host = self._shaped(work)
                self.host_side += 1
            else:
                shaped = self._shaped({k: v for k, v in batch.items() if k != "_host"})
            if shaped is None:
                return None
            if crit is not None and host is None:
                from .graphed import GraphedStep
                host = self.fetch(shaped, self.spec, GraphedStep.BIG)
            key, gcap = self._key(model, shaped, dataset_name, None if host is None else host["num_clips"])
        except Exception:
            return None
        if key in self.bad:
            return None
        if crit is None or (key not in self.steps and self.seen.get(key, 0) < 1):
            self.seen[key] = self.seen.get(key, 0) + 1
            self.eager += 1
            return None
        step = None
        for s in self.steps.get(key, []):
            try:
                s.load_batch(shaped, redraw=True, host=host, defer_targets=True)
            except ValueError:
                continue
            step = s
            break
        if step is None:
            step = self._capture(key, gcap, shaped, dataset_name, crit)
            if step is None:
                return None
        else:
            self.steps.move_to_end(key)
        out = step.forward_replay()
        if step._pending_targets is not None:
            try:
                step.finish_targets()  # (the criterion's arrays, while the forward graph runs)
            except ValueError:
                return None  # targets beyond the captured capacities: this step runs eagerly
        self.replays += 1
        n = batch["video_feat"].shape[0]
        res = AutoOutputs(out if shaped["video_mask"].shape[0] == n else self._real_rows(out, n))
        res._mesm_model = self.model
        res._auto_step, res._auto_gen, res._auto_batch = step, self.gen, batch
        return res

    @staticmethod
    def _real_rows(out, n):
        """outputs of a pair-padded replay, cut back to the caller's pairs (views; every entry leads with the pair axis)"""
        cut = lambda t: t[:n] if torch.is_tensor(t) and t.dim() >= 1 else t
        return {k: ([{kk: cut(vv) for kk, vv in d.items()} for d in v] if isinstance(v, list) else cut(v))
                for k, v in out.items()}

    def _capture(self, key, gcap, batch, dataset_name, crit):
        from .graphed import SplitGraphedStep
        model = self.model()
        gb = model.gradbuf()
        while sum(len(v) for v in self.steps.values()) >= self.max_graphs and self.steps:
            self.steps.popitem(last=False)
        dev = batch["video_feat"].device
        clone = lambda v: v.clone() if torch.is_tensor(v) else v
        static = {}
        for k, v in batch.items():
            if k == "_n_real":
                continue
            if isinstance(v, list) and v and isinstance(v[0], dict):
                static[k] = [{kk: clone(vv) for kk, vv in d.items()} for d in v]
            else:
                static[k] = clone(v)
        if "_n_real" in batch:
            static["_n_real"] = batch["_n_real"]
            P = static["video_mask"].shape[0]
            for kk in ("video_feat", "words_id"):
                t = static[kk]
                if t.shape[0] < P:
                    static[kk] = torch.cat([t, t[:1].expand((P - t.shape[0],) + tuple(t.shape[1:]))]).contiguous()
        # the warm-up and capture steps run real backward passes: the caller's gradients are put back afterwards
        had = [p.grad is not None for p in gb.params]
        saved = gb.flat.clone() if (gb.flat is not None and any(had)) else None
        self.busy = True
        try:
            step = SplitGraphedStep(model, crit, static, dataset_name, warmup=1, caps="auto", group_cap=gcap if gcap > 1 else None)
        except Exception as e:  # a shape this build cannot capture: stay eager for it, say so once
            self.bad.add(key)
            warnings.warn("mesm_amd.autograph: capture failed for %r (%s: %s); this shape stays eager" % (key, type(e).__name__, e))
            step = None
        finally:
            self.busy = False
            if saved is not None:
                gb.flat.copy_(saved)
            for p, h in zip(gb.params, had):
                p.grad = p._mesm_gview if h else None
        if step is None:
            return None
        self.steps.setdefault(key, []).append(step)
        self.captures += 1
        return step

    # ------------------------------------------------------------------ criterion / backward
    def _check(self, step, gen, what):
        if gen != self.gen:
            raise RuntimeError(
                "mesm_amd.autograph: %s of a forward that is no longer the latest one of this model -- a replayed step keeps "
                "its activations in static graph memory, so only ONE forward may be in flight (forward -> criterion -> "
                "backward).  Set MESM_AUTOGRAPH=0 (or model.autograph(False)) for patterns that interleave forwards." % what)

    def criterion(self, crit, outputs, targets, is_training):
        step = outputs._auto_step
        self._check(step, outputs._auto_gen, "criterion call on the outputs")
        if self.crit is None or self.crit() is not crit:
            raise RuntimeError("mesm_amd.autograph: these outputs belong to a step captured with another criterion object; "
                               "call model.autograph(False) to mix criteria")
        if not is_training:
            raise RuntimeError("mesm_amd.autograph: criterion(..., is_training=False) on the outputs of a training forward")
        b = outputs._auto_batch
        if targets is not b and any(targets.get(k) is not b.get(k) for k in ("video_mask", "saliency_label", "norm_span", "words_label")):
            raise RuntimeError("mesm_amd.autograph: the criterion's targets are not the batch the forward was called with "
                               "(a replayed step reads its targets when the forward starts)")
        losses, total = step.criterion_replay()
        if self.token is None or self.token.device != total.device:
            self.token = torch.zeros((), device=total.device, requires_grad=True)
        return losses, _Backward.apply(self.token, total, self, step, outputs._auto_gen)

    def backward(self, step, gen, g):
        self._check(step, gen, "backward")
        model = self.model()
        gb = model.gradbuf()
        prev = gb.flat.clone() if any(p.grad is not None for p in gb.params) else None
        step.backward_replay(g)
        if prev is not None:
            gb.flat.add_(prev)  # the captured backward starts from a cleared buffer: accumulate like autograd does
        for p in step.grad_params:
            if p.grad is None:
                p.grad = p._mesm_gview
