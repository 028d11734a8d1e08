"""A whole training step (forward + criterion + backward) captured in ONE HIP graph.

The eager path issues ~2,000 kernel launches per step from Python and is host-bound; for a
fixed batch *shape* every launch argument is static, so the step is captured once
(torch.cuda.CUDAGraph records the HIP launches our C-ABI makes on the capture stream) and
replayed with a single host call.  What stays dynamic is moved to device memory:

  * the batch tensors are static buffers, refreshed with `load_batch()` (same shapes);
  * the host-RNG draws of the reference (negative query index, masked-LM word choice) are
    static index tensors, refreshed with `redraw()` before each replay;
  * dropout seeds: the kernels add a device-side counter (`seed_offset`) that the graph itself
    increments at the start of every replay, so each step gets fresh masks.

Gradients land in the model's flat gradient buffer (param.grad aliases it), exactly as in the
eager path, so optimizer / clip_grad_norm_ / the DDP reducer work unchanged after `run()`.
"""
import os

import torch

from . import kernels as kn
from .criterion import TargetPlan


class GraphedStep:
    def __init__(self, model, criterion, batch, dataset_name, warmup=3, instrument=False):
        self.model, self.crit = model, criterion
        self.dataset_name = dataset_name
        dev = batch["video_feat"].device
        self.dev = dev
        self.batch = batch  # static input buffers (device tensors)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        model.train()
        self._wm_cpu = self._words_mask_cpu()
        self._groups = [int(g) for g in batch["num_clips"].tolist()]
        self.plan = self._make_plan()
        self.tplan = TargetPlan(batch, criterion.multi_clip, dev, criterion.gamma)
        self.batch["_target_plan"] = self.tplan
        gb = model.gradbuf()
        gb.ensure(dev)
        kn.set_seed_offset(self.counter)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._step_body()
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
        self.reducer = None
        self._pins, self._pin_turn = {}, {}

    def _words_mask_cpu(self):
        w = self.batch["words_id"]
        return kn.text_prep(w, self.model.normalize_txt)[1].cpu()

    def _make_plan(self):
        b = self.batch
        return self.model.make_plan(b["video_mask"], self._wm_cpu, b["num_clips"],
                                    self.dataset_name, True, words_weight=b["words_weight"],
                                    clip_mask=b.get("clip_mask"), device=self.dev)

    def _step_body(self):
        b = self.batch
        out = self.model(**b, dataset_name=self.dataset_name, is_training=True, plan=self.plan)
        losses, total = self.crit(out, b, True)
        self.model.zero_grad(set_to_none=True)
        total.backward()
        return total.detach(), {k: v.detach() for k, v in losses.items()}

    def load_batch(self, batch):
        """Copy a new batch of the SAME shapes (and the same group sizes / target counts, which
        the captured index plans depend on) into the static input buffers."""
        for k, v in batch.items():
            cur = self.batch.get(k)
            if torch.is_tensor(cur) and torch.is_tensor(v):
                if cur.shape != v.shape:
                    raise ValueError("GraphedStep.load_batch: %s changed shape %s -> %s"
                                     % (k, tuple(cur.shape), tuple(v.shape)))
                cur.copy_(v, non_blocking=True)

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
        if redraw:
            self.redraw()
        self.graph.replay()
        return self.total
