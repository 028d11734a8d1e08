"""Data-parallel gradient reduction over the flat gradient buffer (new functionality: the
reference is single-process).

One process per GPU; every rank runs the same model on its own video groups; gradients are
averaged with a few large all-reduces (RCCL over xGMI when the backend is "nccl") of
contiguous slices of the ONE flat fp32 buffer (gradbuf.py), launched asynchronously from
inside backward as soon as every parameter of a slice has received all its contributions,
so the collective overlaps the rest of backward.  `finish()` runs automatically at the end
of backward (autograd engine callback): it waits for the collectives and scales by 1/world.

Parameters that never receive a gradient (e.g. txt_position_embed.*, output_sent_proj.*) keep
.grad = None on every rank and their (zero) slices are reduced harmlessly.

The number of contributions per parameter (weights shared by the positive / negative / MLM
passes get several) is learnt during the first backward, which reduces everything at the end.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, gradbuf, process_group=None, n_buckets=6, hook=True):
        self.gb = gradbuf
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.n_buckets = n_buckets
        self.expected = None        # contributions per parameter, learnt on the first backward
        self.counts = {}
        self.works = []
        self.launched = set()
        self.callback_queued = False
        self.stale = False
        self.buckets = None         # list of (lo, hi, [param ids])
        self.bucket_left = None
        self.param_bucket = {}
        if hook:  # overlap mode: collectives are launched from inside backward
            gradbuf.on_ready = self._on_ready

    # bucket = contiguous slice of the flat buffer, roughly equal sizes
    def _make_buckets(self):
        gb = self.gb
        target = max(1, gb.numel // self.n_buckets)
        self.buckets, lo, ids = [], 0, []
        for p, off in zip(gb.params, gb.offsets):
            ids.append(id(p))
            end = off + (p.numel() + 3) // 4 * 4
            if end - lo >= target:
                self.buckets.append((lo, end, ids))
                lo, ids = end, []
        if ids:
            self.buckets.append((lo, gb.numel, ids))
        for b, (_, _, pids) in enumerate(self.buckets):
            for pid in pids:
                self.param_bucket[pid] = b

    def _on_ready(self, p):
        if self.world == 1:
            return
        if not self.callback_queued:
            self.callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
        pid = id(p)
        self.counts[pid] = self.counts.get(pid, 0) + 1
        if self.expected is None:
            return
        if self.counts[pid] > self.expected.get(pid, 0) and self.param_bucket[pid] in self.launched:
            self.stale = True  # a gradient arrived after its slice was sent: pattern changed
        if self.counts[pid] == self.expected.get(pid, -1):
            b = self.param_bucket[pid]
            self.bucket_left[b] -= 1
            if self.bucket_left[b] == 0:
                self._launch(b)

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self.launched.add(b)
        self.works.append(dist.all_reduce(self.gb.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg,
                                          async_op=True))

    def _reset_step(self):
        self.counts = {}
        self.works = []
        self.launched = set()
        self.callback_queued = False
        if self.expected is not None:
            self.bucket_left = [sum(1 for pid in pids if self.expected.get(pid, 0) > 0)
                                for _, _, pids in self.buckets]

    def finish(self):
        """Wait for the bucket collectives (launching the ones not yet started) and average."""
        if self.world == 1:
            return
        if self.buckets is None:
            self._make_buckets()
        for b in range(len(self.buckets)):
            if b not in self.launched:
                self._launch(b)
        for w in self.works:
            w.wait()
        if self.stale:
            raise RuntimeError("GradReducer: the per-parameter contribution pattern changed between "
                               "steps; create a new GradReducer (or call relearn()) after changing the "
                               "forward configuration")
        self.gb.flat.mul_(1.0 / self.world)
        if self.expected is None:
            self.expected = dict(self.counts)
        self._reset_step()


    def relearn(self):
        self.expected = None
        self.stale = False
        self._reset_step()


def init_process_group_from_env(device=None):
    """RANK / WORLD_SIZE / MASTER_* from the environment (torchrun contract).  backend 'nccl' is
    RCCL on ROCm; 'gloo' for the CPU tests."""
    import os
    if dist.is_initialized():
        return
    backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]),
                            world_size=int(os.environ["WORLD_SIZE"]), **kw)
