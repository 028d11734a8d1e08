"""Data-parallel training over the flat gradient buffer (new functionality: the reference is
single-process; insertion point train.py:68-72).

One process per GPU.  The path shards by VIDEO GROUP (`shard_groups`: rank r takes groups r::W of the
global batch, at least two per rank because the negatives come from another group of the local batch,
SURVEY Q5); negatives, the in-batch rec_ss contrast and the Q1 mask leakage stay per-rank, so the
parity target is: gradient == mean over ranks of the single-process gradient on each rank's shard.

The one exchange step is a sum-all-reduce (then x 1/W) of contiguous slices ("buckets") of the ONE flat
fp32 gradient buffer (gradbuf.py) -- RCCL over xGMI when the backend is "nccl".  In hook mode the
collectives are launched asynchronously from INSIDE backward, as soon as every parameter of a bucket has
received all its contributions, so they overlap the rest of backward on the process group's side
stream; `finish()` runs at the end of backward (autograd engine callback): it launches what is left,
waits, and scales.  Under HIP-graph capture (graphed.GraphedStep(reducer=...)) the same calls are
recorded into the step's graph: fork to the collective stream at the launch point, join at finish().

Every rank issues the bucket collectives in the SAME order -- highest bucket first (backward produces
gradients in reverse parameter order), bucket k only after bucket k+1 -- whatever order its own
gradients become ready in; the per-parameter contribution counts that define "ready" are learnt on the
first backward and agreed across ranks (element-wise MAX), so a rank that sees fewer contributions (no
out-of-vocabulary word in its shard, say) just launches later, never differently.  If a gradient still
arrives after its bucket was sent, the step's result would be wrong: all ranks agree on that through a
flag reduced with the last bucket and raise together.

Parameters that never receive a gradient (txt_position_embed.*, output_sent_proj.*) keep .grad = None
on every rank and their (zero) slices are reduced harmlessly.
"""
import os

import torch
import torch.distributed as dist


def shard_groups(batch, rank, world, min_groups=2):
    """The rows of `batch` (collate output, dataset/base.py:326-355 / qvhighlights.py:252-284) that
    belong to video groups rank, rank + world, ...: tensors whose first extent is the number of pairs
    are row-selected, per-pair lists are sub-listed, `num_clips` keeps the selected groups."""
    groups = [int(g) for g in batch["num_clips"].tolist()]
    G, N = len(groups), sum(groups)
    mine = list(range(rank, G, world))
    if len(mine) < min_groups:
        raise ValueError("shard_groups: rank %d of %d gets %d of %d video groups; negatives are drawn from "
                         "ANOTHER group of the local batch, so every rank needs >= %d"
                         % (rank, world, len(mine), G, min_groups))
    starts = [0]
    for g in groups:
        starts.append(starts[-1] + g)
    rows = [i for gi in mine for i in range(starts[gi], starts[gi + 1])]
    idx = torch.tensor(rows, dtype=torch.int64)
    out = {}
    for k, v in batch.items():
        if k == "num_clips":
            out[k] = torch.tensor([groups[gi] for gi in mine], dtype=v.dtype)
        elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == N:
            out[k] = v[idx.to(v.device)]
        elif isinstance(v, (list, tuple)) and len(v) == N:
            out[k] = [v[i] for i in rows]
        else:
            out[k] = v
    return out


class RcclComm:
    """This library's OWN RCCL communicator (csrc/ddp.hip: mesm_ddp_*): collectives issued as raw ncclAllReduce on
    HIP streams the library picks.  torch.distributed is only the rendezvous (rank 0's 128-byte unique id travels
    through its object broadcast); no process-group watchdog ever sees these collectives, so recording them into the
    step's HIP graph is safe (the torch process group aborted ~3 % of process starts doing that).
    Without an initialised process group: a 1-rank communicator (tests on one GPU)."""

    def __init__(self, device, process_group=None):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        L = _lib.lib()
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        torch.cuda.set_device(device)
        torch.zeros(1, device=device)  # the HIP context of this device exists before RCCL looks for it
        buf = (ctypes.c_uint8 * 128)()
        if self.rank == 0:
            self._check(L.mesm_ddp_unique_id(buf), "mesm_ddp_unique_id")
        if self.world > 1:
            box = [bytes(buf)]
            dist.broadcast_object_list(box, src=0, group=process_group, device=device)
            buf = (ctypes.c_uint8 * 128)(*box[0])
        h = ctypes.c_void_p()
        self._check(L.mesm_ddp_init(buf, self.rank, self.world, ctypes.byref(h)), "mesm_ddp_init")
        self.handle = h

    def _check(self, rc, what):
        if rc != 0:
            raise self._lib.MesmError("%s failed with status %d: %s"
                                      % (what, rc, self._lib.lib().mesm_ddp_last_error().decode(errors="replace")))

    def allreduce(self, t, side):
        """in-place sum of a contiguous fp32 tensor; side: on the communicator's own stream behind the current one"""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._check(self._lib.lib().mesm_ddp_allreduce(self.handle, self._ct.c_void_p(t.data_ptr()), t.numel(),
                                                       self._lib.stream_ptr(), 1 if side else 0), "mesm_ddp_allreduce")

    def wait(self):
        self._check(self._lib.lib().mesm_ddp_wait(self.handle, self._lib.stream_ptr()), "mesm_ddp_wait")

    def count(self):
        """ranks the communicator spans, as RCCL reports it (ncclCommCount)"""
        n = self._ct.c_int32(0)
        self._check(self._lib.lib().mesm_ddp_count(self.handle, self._ct.byref(n)), "mesm_ddp_count")
        return int(n.value)


def flat_checksum(flat):
    """(sum, sum of squares, sum of |x| weighted by position) of a flat buffer in fp64, as a 3-element tensor on its
    device: cheap, and sensitive to a permuted or partially reduced buffer"""
    x = flat.detach().double()
    w = torch.arange(1, x.numel() + 1, device=x.device, dtype=torch.float64) / x.numel()
    return torch.stack([x.sum(), (x * x).sum(), (x.abs() * w).sum()])


def ranks_agree(flat, process_group=None, rtol=0.0):
    """True iff every rank holds the same flat buffer (after a gradient all-reduce they must: same sum, same order of
    the RCCL ring on every rank): the checksums' MAX and MIN over ranks coincide.  -> (ok, relative spread).  Used by
    bench.py after its first data-parallel step and by the gloo tests."""
    c = flat_checksum(flat)
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return True, 0.0
    hi, lo = c.clone(), c.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=process_group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=process_group)
    spread = float(((hi - lo).abs() / hi.abs().clamp_min(1e-30)).max())
    return spread <= rtol, spread


class GradReducer:
    def __init__(self, gradbuf, process_group=None, n_buckets=6, hook=True, force=False, inline=False, comm=None,
                 fold_scale=False):
        """force: issue the collectives even in a 1-rank group (exercises the RCCL / capture path on one GPU)
        inline: blocking collectives (async_op=False), which this torch issues on the CURRENT stream: under graph
        capture the step stays ONE linear chain -- no second hardware queue, hence none of the ~1 us per kernel
        boundary that any concurrently active queue costs the main chain on this runtime, but
        no overlap with backward either.  Worth it when the wire time is shorter than that toll."""
        self.gb = gradbuf
        self.pg = process_group
        # comm: an RcclComm -- the bucket collectives go through this library's own communicator (capturable without a
        # watchdog); None: torch.distributed's process group.  fold_scale: the 1 / world factor is applied to the LOSS
        # gradient by the caller (`backward_scale()`; graphed.GraphedStep does) instead of a pass over the flat buffer.
        self.comm = comm
        self.fold_scale = fold_scale
        self.world = comm.world if comm is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.active = self.world > 1 or (force and (comm is not None or dist.is_initialized()))
        self.n_buckets = n_buckets
        self.expected = None        # contributions per parameter, learnt on the first backward
        self.counts = {}
        self.works = []
        self.next_bucket = -1       # buckets are launched from the last one down to 0, in this order only
        self.ready = set()
        self.callback_queued = False
        self.stale = False
        self.buckets = None         # list of (lo, hi, [param ids])
        self.bucket_left = None
        self.param_bucket = {}
        self.flag = None            # 1-element tensor reduced after the last bucket: "a late gradient somewhere"
        self.hook = hook
        self.inline = inline
        self.launch_log = []        # bucket indices in launch order (tests)
        if hook:  # overlap mode: collectives are launched from inside backward
            gradbuf.on_ready = self._on_ready

    # bucket = contiguous slice of the flat buffer, roughly equal sizes
    def _make_buckets(self):
        gb = self.gb
        target = max(1, gb.numel // self.n_buckets)
        self.buckets, lo, ids = [], 0, []
        for off, p in sorted(zip(gb.offsets, gb.params), key=lambda t: t[0]):  # (packs permute the layout)
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
        self.next_bucket = len(self.buckets) - 1

    def _on_ready(self, p):
        if not self.active:
            return
        if not self.callback_queued:
            self.callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
        pid = id(p)
        self.counts[pid] = self.counts.get(pid, 0) + 1
        if self.expected is None:
            return
        b = self.param_bucket[pid]
        if self.counts[pid] > self.expected.get(pid, 0) and b > self.next_bucket:
            self.stale = True  # a gradient arrived after its slice was sent: this step's mean is wrong
        if self.counts[pid] == self.expected.get(pid, -1):
            self.bucket_left[b] -= 1
            if self.bucket_left[b] == 0:
                self.ready.add(b)
                self._launch_ready()

    def _launch_ready(self):
        while self.next_bucket >= 0 and self.next_bucket in self.ready:
            self._launch(self.next_bucket)

    def _launch(self, b):
        assert b == self.next_bucket
        lo, hi, _ = self.buckets[b]
        self.next_bucket -= 1
        self.launch_log.append(b)
        if self.comm is not None:
            self.comm.allreduce(self.gb.flat[lo:hi], side=not self.inline)
            return
        w = dist.all_reduce(self.gb.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=not self.inline)
        if w is not None:
            self.works.append(w)

    def _reset_step(self):
        self.counts = {}
        self.works = []
        self.ready = set()
        self.next_bucket = len(self.buckets) - 1 if self.buckets is not None else -1
        self.callback_queued = False
        self.stale = False
        if self.expected is not None:
            self.bucket_left = [sum(1 for pid in pids if self.expected.get(pid, 0) > 0)
                                for _, _, pids in self.buckets]
            # a bucket none of whose parameters ever gets a gradient is ready from the start
            self.ready = {b for b, n in enumerate(self.bucket_left) if n == 0}

    def _agree_expected(self):
        """element-wise MAX over ranks of the learnt contribution counts (same parameter order everywhere)"""
        gb = self.gb
        dev = gb.flat.device
        v = torch.tensor([self.counts.get(id(p), 0) for p in gb.params], dtype=torch.int32, device=dev)
        if dist.is_initialized():
            dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.pg)
        self.expected = {id(p): int(c) for p, c in zip(gb.params, v.tolist())}

    def finish(self):
        """Launch the buckets not yet started (in order), wait for all of them and average."""
        if not self.active:
            return
        if self.buckets is None:
            self._make_buckets()
        while self.next_bucket >= 0:
            self._launch(self.next_bucket)
        capturing = self.gb.flat.is_cuda and torch.cuda.is_current_stream_capturing()
        if capturing and self.hook and self.expected is not None and self.stale:
            # no collective is needed to know it locally, and a graph captured now would replay a bucket
            # all-reduce that was issued before every gradient of that bucket had been written -- on every step
            self._reset_step()
            raise RuntimeError("GradReducer: while capturing, a gradient arrived after its bucket had been "
                               "all-reduced (the captured batch has another contribution pattern than the one "
                               "learnt); the captured graph would average incomplete gradients.  relearn() on a "
                               "batch of this kind, then capture again.")
        agree = self.hook and self.expected is not None and not capturing and dist.is_initialized()
        if agree:
            # late-gradient flag, agreed across ranks (MAX) so that every rank raises, or none does
            if self.flag is None:
                self.flag = torch.zeros(1, device=self.gb.flat.device, dtype=torch.float32)
            self.flag.fill_(1.0 if self.stale else 0.0)
            self.works.append(dist.all_reduce(self.flag, op=dist.ReduceOp.MAX, group=self.pg, async_op=True))
        for w in self.works:
            w.wait()
        if self.comm is not None:
            self.comm.wait()
        if not self.fold_scale:
            self.gb.flat.mul_(1.0 / self.world)
        if agree and float(self.flag) > 0:
            self._reset_step()
            raise RuntimeError("GradReducer: on some rank a gradient arrived after its bucket had been "
                               "all-reduced (the contribution pattern changed between steps); the reduced "
                               "gradients of this step are incomplete.  Call relearn() after changing the "
                               "forward configuration.")
        if self.hook and self.expected is None:
            self._agree_expected()
        self._reset_step()

    def backward_scale(self):
        """what to multiply the loss gradient with when fold_scale is on (1 / world), else 1"""
        return 1.0 / self.world if (self.fold_scale and self.active) else 1.0

    def relearn(self):
        self.expected = None
        self._reset_step()


def init_process_group_from_env(device=None):
    """RANK / WORLD_SIZE / MASTER_* from the environment (torchrun contract).  backend 'nccl' is
    RCCL on ROCm; 'gloo' for the CPU tests."""
    if dist.is_initialized():
        return
    backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = device
        # Collectives captured inside a HIP graph: the process group's flight recorder keeps querying the events
        # of the collectives it has seen -- also those recorded in a CAPTURING stream, which is an error
        # (hipErrorCapturedEvent) that terminates the process from the watchdog thread (seen in ~1 of 10 runs of
        # the 1-rank capture test; 0 of 46 with the recorder off).  It is a debugging aid; off unless asked for.
        os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "0")
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]),
                            world_size=int(os.environ["WORLD_SIZE"]), **kw)
