"""A whole training step (forward + criterion + backward [+ gradient all-reduce]) captured in ONE
HIP graph, and a cache of such graphs keyed by batch shape.

The eager path issues ~700 kernel launches per step from Python and is host-bound; for a fixed
batch *shape* every launch argument is static, so the step is captured once
(torch.cuda.CUDAGraph records the HIP launches our C-ABI makes on the capture stream) and
replayed with a single host call.  What stays dynamic is moved to device memory:

  * the batch tensors are static buffers, refreshed with `load_batch()`;
  * every data-dependent host decision of the forward and of the criterion (group gathers, GT-clip
    gathers, flattened targets, rec_ss positives: model.Plan / criterion.TargetPlan) lives in static
    index tensors that `load_batch()` REBUILDS from the new batch on the host and copies over; their
    data-dependent extents (GT clips per pair Lc, group video length Lss, target windows) are padded
    to capture-time capacities, and a batch that does not fit raises ValueError (StepCache then
    captures another graph);
  * the host-RNG draws of the reference (negative query index, masked-LM word choice) are
    static index tensors, refreshed with `redraw()` before each replay;
  * dropout seeds: the kernels add a device-side counter (`seed_offset`) that the graph itself
    increments at the start of every replay, so each step gets fresh masks.

Gradients land in the model's flat gradient buffer (param.grad aliases it), exactly as in the
eager path, so optimizer / clip_grad_norm_ work unchanged after `run()`.  With a `reducer`
(ddp.GradReducer) the bucketed gradient all-reduce is captured INSIDE the graph on the process
group's side stream: bucket k reduces while the rest of backward runs.

The captured launches bake in the addresses of the parameters: they must not move afterwards.
`build_model` / `FlatAdamW.__init__` flatten the parameters eagerly (model.flat_params), and `run()`
raises if an address differs from capture time.
"""
import os

import torch

from . import kernels as kn
from .criterion import TargetPlan


def _round_up(x, m):
    return (x + m - 1) // m * m


class GraphedStep:
    def __init__(self, model, criterion, batch, dataset_name, warmup=3, instrument=False, reducer=None,
                 caps=None):
        """caps: None = exact extents of `batch` (benchmarks); "auto" = bucketed capacities so that other
        batches of the same (N, Lv, Lw, groups) replay; or a dict with any of Lc / Lss / T / Tmax."""
        self.model, self.crit = model, criterion
        self.dataset_name = dataset_name
        dev = batch["video_feat"].device
        self.dev = dev
        self.batch = batch  # static input buffers (device tensors)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.reducer = reducer
        model.train()
        model.flat_params()
        self._wm_cpu = self._words_mask_cpu()
        self._groups = [int(g) for g in batch["num_clips"].tolist()]
        self.caps = self._resolve_caps(caps, batch)
        self.plan = self._make_plan(batch, self._wm_cpu)
        self.tplan = self._make_tplan(batch)
        self.batch["_target_plan"] = self.tplan
        gb = model.gradbuf()
        gb.ensure(dev)
        kn.set_seed_offset(self.counter)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(warmup, 2 if reducer is not None else 1)):
                    self._step_body()  # (a reducer learns its bucket schedule on the first one)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            dot = os.environ.get("MESM_GRAPH_DOT")
            if dot:
                self.graph.enable_debug_mode()
            model.zero_grad(set_to_none=True)
            if instrument:  # record the GEMM launches of the captured step (bench roofline)
                kn.gemm_tape(True)
            with torch.cuda.graph(self.graph):
                self.counter.add_(1)
                self.total, self.losses = self._step_body()
            if dot:
                self.graph.debug_dump(dot)
        finally:
            kn.set_seed_offset(None)
            if instrument:
                kn.gemm_tape(False)
        self._ptrs = self._param_ptrs()
        self._pins, self._pin_turn = {}, {}

    # ------------------------------------------------------------------ capture-time capacities
    def _resolve_caps(self, caps, batch):
        if caps is None:
            return {}
        if caps == "auto":
            caps = {}
            N, Lv = batch["video_mask"].shape
            if self.model.rec_fw:
                caps["Lc"] = min(Lv, _round_up(int(batch["clip_mask"].sum(1).max()), 8))
            if self.model.rec_ss and self.dataset_name == "qvhighlights":
                vm = batch["video_mask"].cpu()
                lens = [int(c.sum()) for c in torch.split(vm, self._groups)]
                full = all(g == 1 for g in self._groups) and bool(vm.all())
                # 64 = one key tile of the attention kernels; a group cannot hold more than its pairs' clips
                caps["Lss"] = Lv if full else min(_round_up(max(lens), 64), max(self._groups) * Lv)
            if self.crit.multi_clip:
                Q = self.model.num_queries
                tmax = max(len(t["spans"]) for t in batch["norm_span"])
                caps["Tmax"] = max(tmax, min(5, Q))  # QVHighlights keeps <= 5 windows (qvhighlights.py:148-150)
                caps["T"] = N * caps["Tmax"]
        return dict(caps)

    def _words_mask_cpu(self):
        w = self.batch["words_id"]
        if w.dim() != 3:  # token ids (CLIP / GloVe encoders): the collate mask, cut like model.py:114-116
            return self.batch["words_mask"][:, :self.model.max_words_l].cpu()
        return kn.text_prep(w, self.model.normalize_txt)[1].cpu()

    def _make_plan(self, b, wm_cpu):
        return self.model.make_plan(b["video_mask"], wm_cpu, b["num_clips"], self.dataset_name, True,
                                    words_weight=b["words_weight"], clip_mask=b.get("clip_mask"),
                                    device=self.dev, Lc_cap=self.caps.get("Lc"), Lss_cap=self.caps.get("Lss"))

    def _make_tplan(self, b):
        return TargetPlan(b, self.crit.multi_clip, self.dev, self.crit.gamma, T_cap=self.caps.get("T"),
                          Tmax_cap=self.caps.get("Tmax"))

    def _param_ptrs(self):
        gb = self.model.gradbuf()
        return [p.data_ptr() for p in gb.params] + [gb.flat.data_ptr()]

    def _step_body(self):
        b = self.batch
        out = self.model(**b, dataset_name=self.dataset_name, is_training=True, plan=self.plan)
        losses, total = self.crit(out, b, True)
        self.model.zero_grad(set_to_none=True)
        total.backward()  # a hooked reducer launches its bucket collectives from inside and joins at the end
        return total.detach(), {k: v.detach() for k, v in losses.items()}

    # ------------------------------------------------------------------ new batch, same graph
    @staticmethod
    def _copy_plan(dst, src, what, check_only=False):
        """Copy every tensor attribute of the freshly built host plan `src` into the static tensors of
        the captured plan `dst`; shapes and every non-tensor attribute must be identical."""
        for k, v in vars(src).items():
            cur = getattr(dst, k, None)
            if torch.is_tensor(v):
                if not torch.is_tensor(cur) or cur.shape != v.shape or cur.dtype != v.dtype:
                    raise ValueError("GraphedStep.load_batch: %s.%s does not fit the captured graph (%s -> %s)"
                                     % (what, k, tuple(cur.shape) if torch.is_tensor(cur) else cur, tuple(v.shape)))
                if not check_only:
                    cur.copy_(v, non_blocking=True)
            elif k in ("sizes", "sumT"):  # per-batch bookkeeping the kernels read from tgt_off instead
                if not check_only:
                    setattr(dst, k, v)
            elif cur != v:
                raise ValueError("GraphedStep.load_batch: %s.%s changed (%r -> %r): needs its own graph"
                                 % (what, k, cur, v))

    def load_batch(self, batch):
        """Make the captured step run on `batch` (host or device tensors): same (N, Lv, Lw, Dv, Dt) and
        the same group sizes; GT-clip counts, group video lengths and target windows may differ within
        the capture-time capacities.  Raises ValueError (nothing is modified) when it does not fit."""
        groups = [int(g) for g in batch["num_clips"].tolist()]
        if groups != self._groups:
            raise ValueError("GraphedStep.load_batch: group sizes changed %s -> %s" % (self._groups, groups))
        for k, cur in self.batch.items():
            v = batch.get(k)
            if torch.is_tensor(cur) and torch.is_tensor(v) and cur.shape != v.shape:
                raise ValueError("GraphedStep.load_batch: %s changed shape %s -> %s"
                                 % (k, tuple(cur.shape), tuple(v.shape)))
        words = batch["words_id"]
        if words.dim() != 3:
            wm_cpu = batch["words_mask"][:, :self.model.max_words_l].cpu()
        elif words.is_cuda:
            wm_cpu = kn.text_prep(words, self.model.normalize_txt)[1].cpu()
        else:  # host batch: post_process_text's mask rule (model.py:145-152) without a device round trip
            w = torch.nn.functional.normalize(words, dim=-1, eps=1e-5) if self.model.normalize_txt else words
            wm_cpu = w.sum(-1) != 0
        drawn = self.plan  # keep the current host draws; redraw() replaces them
        plan = self.model.make_plan(batch["video_mask"], wm_cpu, batch["num_clips"], self.dataset_name, True,
                                    words_weight=batch["words_weight"], clip_mask=batch.get("clip_mask"),
                                    neg_index=drawn.neg_index, masked_words=getattr(drawn, "masked_words", None),
                                    device=self.dev, Lc_cap=self.caps.get("Lc"), Lss_cap=self.caps.get("Lss"))
        tplan = self._make_tplan(batch)
        self._copy_plan(self.plan, plan, "plan", check_only=True)
        self._copy_plan(self.tplan, tplan, "targets", check_only=True)
        # everything below only copies: the checks above and the two builders raise before any change
        self._copy_plan(self.plan, plan, "plan")
        self._copy_plan(self.tplan, tplan, "targets")
        for k, cur in self.batch.items():
            v = batch.get(k)
            if torch.is_tensor(cur) and torch.is_tensor(v):
                if cur.is_cuda:
                    cur.copy_(v, non_blocking=True)
                else:
                    self.batch[k] = v  # host-side inputs (words_weight) are only read by redraw()
        self._wm_cpu = wm_cpu

    def _pinned_h2d(self, name, host, dst):
        """host tensor -> static device tensor through one of two pinned staging buffers (a pageable
        source makes the copy synchronous with everything in flight; with pinned memory the host
        only waits for the copy that used the same staging buffer two redraws ago)."""
        slots = self._pins.setdefault(name, [])
        if not slots:
            for _ in range(2):
                slots.append([torch.empty(host.shape, dtype=host.dtype, pin_memory=True), None])
        i = self._pin_turn.get(name, 0)
        self._pin_turn[name] = i ^ 1
        buf, ev = slots[i]
        if ev is not None:
            ev.synchronize()
        buf.copy_(host)
        dst.copy_(buf, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        slots[i][1] = ev

    def redraw(self):
        """New negative-query indices and MLM word choices (host RNG, like the reference does on
        every forward), copied into the static index tensors the graph reads."""
        m = self.model
        self._pinned_h2d("neg", m.draw_neg_index(self._groups), self.plan.neg_index)
        if hasattr(self.plan, "masked_words"):
            mw = m.draw_masked_words(self._wm_cpu, self.batch["words_weight"]).bool()
            self._pinned_h2d("mw", mw, self.plan.masked_words)

    def run(self, redraw=True):
        if self._param_ptrs() != self._ptrs:
            raise RuntimeError(
                "GraphedStep: a parameter or the gradient buffer moved after capture (model.to(), a new "
                "optimizer flattening the parameters, ...): the graph still points at the old storage. "
                "Create the optimizer before the GraphedStep, or capture again.")
        if redraw:
            self.redraw()
        self.graph.replay()
        return self.total


class StepCache:
    """One captured graph per batch-shape bucket: `run(batch)` replays the graph whose
    key (N, Lv, Lw, feature dims, group sizes) and capacities fit the batch, capturing a new one when none
    does.  Ragged real batches therefore stay on the graph path; `captures` counts the graphs built."""

    def __init__(self, model, criterion, dataset_name, reducer=None, max_graphs=16):
        self.model, self.crit, self.dataset_name = model, criterion, dataset_name
        self.reducer = reducer
        self.steps = {}
        self.max_graphs = max_graphs
        self.captures = 0

    @staticmethod
    def key(batch):
        return (tuple(batch["video_feat"].shape), tuple(batch["words_id"].shape),
                tuple(int(g) for g in batch["num_clips"].tolist()))

    def run(self, batch, redraw=True):
        k = self.key(batch)
        for gs in self.steps.get(k, []):
            try:
                gs.load_batch(batch)
            except ValueError:
                continue
            return gs.run(redraw=redraw), gs
        if sum(len(v) for v in self.steps.values()) >= self.max_graphs:
            self.steps.pop(next(iter(self.steps)))
        from .synthetic import to_device
        static = to_device({kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in batch.items()},
                           next(self.model.parameters()).device)
        gs = GraphedStep(self.model, self.crit, static, self.dataset_name, warmup=1, reducer=self.reducer,
                         caps="auto")
        self.steps.setdefault(k, []).append(gs)
        self.captures += 1
        return gs.run(redraw=redraw), gs
