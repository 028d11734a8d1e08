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

from . import draws
from . import kernels as kn
from .criterion import TargetPlan


def _round_up(x, m):
    return (x + m - 1) // m * m


_CAPTURE_STREAMS = {}


def capture_stream(dev):
    """THE stream of this process on which steps are warmed up and captured (one per device).  Autograd pins an
    AccumulateGrad node to the stream of its first use for as long as the node lives, and the graph of a warm-up step
    can live on through reference cycles; a later capture on ANOTHER stream records the hand-over to that stream as a
    side branch of the HIP graph -- a second hardware queue at replay, +0.3 ms per step.  With one stream for every
    warm-up and capture of the process there is nothing to hand over, however many graphs are captured."""
    key = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    st = _CAPTURE_STREAMS.get(key)
    if st is None:
        st = _CAPTURE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


class _DrawRing:
    """The per-step host draws reach the device through a ring of pinned host buffers that the FIRST NODE of the
    captured step reads itself (kernels.step_begin), instead of a host-to-device copy between two replays: replay n
    reads slot n % SLOTS -- the device counts its own replays -- so the host publishes the draws of replay n into that
    slot right before launching it and waits, when it gets SLOTS - 1 replays ahead, for the replay that last used the
    slot."""
    SLOTS = 4

    def __init__(self, arena, names, device):
        self.arena = arena
        self.lo, self.hi = arena.span(names)
        self.nbytes = self.hi - self.lo
        self.ring = torch.zeros(self.SLOTS * self.nbytes, dtype=torch.uint8).pin_memory()
        self.host = self.ring.numpy()
        self.ctr = torch.zeros(1, dtype=torch.int32, device=device)
        self.dst = arena.dev[self.lo:self.hi]
        self.n = 0
        self.events = [None] * self.SLOTS

    def capture_node(self, seed_ctr):
        kn.step_begin(self.ring, self.nbytes, self.SLOTS, self.dst, self.ctr, seed_ctr)

    def publish(self, arrays):
        s = self.n % self.SLOTS
        if self.events[s] is not None:
            self.events[s].synchronize()  # the replay SLOTS launches ago has read this slot
        self.arena.pack(arrays, self.lo, self.hi, self.host[s * self.nbytes:(s + 1) * self.nbytes])
        return s

    def launched(self, s):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.events[s] = ev
        self.n += 1


class _PulledGraph:
    """graph.replay() that first publishes the owner's current draws for the replay's first node"""

    def __init__(self, graph, owner):
        self._g, self._o = graph, owner

    def replay(self):
        o = self._o
        s = o._pull.publish(o._arr)
        self._g.replay()
        o._pull.launched(s)

    def __getattr__(self, k):
        return getattr(self._g, k)


class GraphedStep:
    BIG = 1 << 20  # batch tensors above 1 MiB (video_feat, word features) are copied on their own
    DRAWN = ("p.neg_index", "p.masked_words")  # what redraw() changes: the host RNG draws of a step

    def __init__(self, model, criterion, batch, dataset_name, warmup=3, instrument=False, reducer=None,
                 caps=None, group_cap=None):
        """caps: None = exact extents of `batch` (benchmarks); "auto" = bucketed capacities so that other
        batches of the same (N, Lv, Lw, groups) replay; or a dict with any of Lc / Lss / T / Tmax."""
        from .arena import Arena
        from .hostplan import HostSpec
        self.model, self.crit = model, criterion
        self.dataset_name = dataset_name
        self.spec = HostSpec.from_model(model, criterion, dataset_name)
        dev = batch["video_feat"].device
        self.dev = dev
        self.batch = dict(batch)  # static inputs: big tensors as given, everything small re-homed in the arena
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self._seed = {}
        self.reducer = reducer
        # the step is captured in the model's CURRENT mode (train(): dropout on, fresh masks every replay;
        # eval(): dropout off, e.g. to reproduce a recorded step) -- like the reference's loop, which calls
        # model.train() itself (train.py:53)
        model.flat_params()
        self._groups = [int(g) for g in batch["num_clips"].tolist()]
        # batches padded to a pair capacity (batching.pad_pairs): the number of REAL pairs is a device scalar of the
        # captured step; None = the step was captured for exactly its pairs and stays that way
        self._n_real = batch.get("_n_real")
        self.batch.pop("_n_real", None)
        self.caps = self._resolve_caps(caps, batch, group_cap)
        # ONE device arena for: the forward's index plan, the criterion's target plan and the small batch tensors
        host = {k: v.detach().cpu() for k, v in batch.items() if torch.is_tensor(v)}
        for k in ("norm_span", "norm_moment"):
            if isinstance(batch.get(k), list):
                host[k] = [{kk: vv.detach().cpu() for kk, vv in d.items()} for d in batch[k]]
        arr, self._pmeta, self._tmeta, self._wm_cpu = self._host_arrays(host, None, None)
        self._arr = arr  # host mirror of the arena
        self._draws = (arr["p.neg_index"], arr.get("p.masked_words"))
        self.arena = Arena(arr, dev, first=self.DRAWN)  # the per-step draws sit together at the front
        # MESM_STEP_PULL=1: the draws are pulled from a pinned ring by the graph's first node instead of travelling as a
        # host-to-device copy between replays.  Measured equal (bench 3.964 vs 3.966 ms/step, profiles/r4c/
        # step_pull_ab.txt: the copy between two replays was never the 50 us tools/host_cost.py's 30-step loop
        # suggested), so the plain copy stays the default.
        self._pull = (_DrawRing(self.arena, self.DRAWN, dev)
                      if dev.type == "cuda" and os.environ.get("MESM_STEP_PULL", "0") == "1" else None)
        v = self.arena.views
        self.plan = model.plan_from({k[2:]: t for k, t in v.items() if k.startswith("p.")}, self._pmeta)
        self.tplan = TargetPlan.__new__(TargetPlan)
        self.tplan.adopt({k[2:]: t for k, t in v.items() if k.startswith("t.")}, self._tmeta)
        for k, t in v.items():
            if k.startswith("b."):
                self.batch[k[2:]] = t
        self.batch["_target_plan"] = self.tplan
        if self._n_real is not None:
            self.batch["_n_valid"] = v["p.n_valid"]
        gb = model.gradbuf()
        gb.ensure(dev)
        kn.set_seed_offset(self.counter)
        try:
            side = capture_stream(dev)
            # autograd graphs of earlier (eager) steps can live on through reference cycles and keep their
            # AccumulateGrad nodes -- pinned to the stream THEY ran on -- alive: collect them before the first warm-up
            import gc
            gc.collect()
            side.wait_stream(torch.cuda.current_stream())
            # A reducer learns its bucket schedule on its first step, so the first graph warms up with the
            # collectives live (every rank is there together).  Later graphs of a StepCache are captured whenever
            # a rank meets a new batch shape -- not at the same step on every rank -- so their warm-up steps must
            # not issue collectives the other ranks do not match: the gradient hooks are detached for them.
            learnt = reducer is not None and getattr(reducer, "expected", None) is not None
            saved_hook = gb.on_ready
            if learnt:
                gb.on_ready = None
            try:
                with torch.cuda.stream(side):
                    for _ in range(max(warmup, 2 if (reducer is not None and not learnt) else 1)):
                        self._step_body()
            finally:
                gb.on_ready = saved_hook
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            dot = os.environ.get("MESM_GRAPH_DOT")
            if dot:
                self.graph.enable_debug_mode()
            model.zero_grad(set_to_none=True)
            if instrument:  # record the GEMM launches of the captured step (bench roofline)
                kn.gemm_tape(True)
            # capture_error_mode="thread_local": the process group's watchdog THREAD polls the events of earlier
            # (eager, warm-up) collectives whenever it likes; under the default global mode such a query during
            # our capture is an error that kills the process (seen in ~1 of 3 runs of the 1-rank RCCL test)
            # captured on the SAME stream the warm-up steps ran on: autograd pins every AccumulateGrad node (kept
            # alive by gradbuf's hooks) to the stream of its first use, and a capture on another stream records the
            # hand-over to that stream as a side branch of the graph -- a second hardware queue at replay, which
            # costs every kernel boundary of the main chain (5.2 -> 5.5 ms per step)
            self._capture(side)
            if dot:
                self.graph.debug_dump(dot)
            if self._pull is not None:
                self.graph = _PulledGraph(self.graph, self)
        finally:
            kn.set_seed_offset(None)
            if instrument:
                kn.gemm_tape(False)
        self._ptrs = self._param_ptrs()

    def _capture(self, side):
        """the whole step -- forward, criterion, backward -- as ONE graph (SplitGraphedStep: three)"""
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            if self._pull is not None:
                self._pull.capture_node(self.counter)  # draws from the pinned ring + seed offset += 1: one node
            else:
                self.counter.add_(1)
            self.total, self.losses = self._step_body()

    # ------------------------------------------------------------------ host half (shared with the loader workers)
    def _resolve_caps(self, caps, batch, group_cap=None):
        return self.spec.resolve_caps(caps, batch, self._groups, group_cap)

    def _host_arrays(self, host, neg_index, masked_words, targets=True):
        """hostplan.HostSpec.host_arrays with this step's groups / capacities: {name: numpy array} of everything small
        the captured step reads, the two metas and the word mask"""
        return self.spec.host_arrays(host, self._groups, self.caps, self._n_real, neg_index, masked_words, self.BIG,
                                     targets=targets)

    def _param_ptrs(self):
        gb = self.model.gradbuf()
        return [p.data_ptr() for p in gb.params] + [gb.flat.data_ptr()]

    def _step_body(self):
        b = self.batch
        out = self.model(**b, dataset_name=self.dataset_name, is_training=True, plan=self.plan)
        losses, total = self.crit(out, b, True)
        self.model.zero_grad(set_to_none=True)
        # a hooked reducer launches its bucket collectives from inside and joins at the end; with fold_scale the
        # 1 / world of the gradient MEAN rides the loss gradient (one scalar) instead of a pass over the flat buffer
        sc = self.reducer.backward_scale() if self.reducer is not None else 1.0
        # the seed of the backward pass from a tensor made once (autograd's own ones_like is a fill launch per step)
        seed = self._seed.get(sc)
        if seed is None or seed.shape != total.shape:
            seed = self._seed[sc] = torch.full_like(total, sc)
        total.backward(seed)
        return total.detach(), {k: v.detach() for k, v in losses.items()}

    # ------------------------------------------------------------------ new batch, same graph
    def load_batch(self, batch, redraw=False, host=None, defer_targets=False):
        """Make the captured step run on `batch` (host tensors; device tensors are brought to the host first,
        the plans are host work): same (N, Lv, Lw, Dv, Dt) and the same group sizes; GT-clip counts, group video
        lengths and target windows may differ within the capture-time capacities.  Everything small -- index
        plans, targets, masks, labels -- goes up in ONE pinned transfer (arena.Arena), the feature tensors in one
        copy each.  redraw: also draw new negatives / MLM words here (then call run(redraw=False): one upload per
        step instead of two).  Raises ValueError (nothing is modified) when it does not fit."""
        groups = [int(g) for g in (batch if host is None else host)["num_clips"].tolist()]
        if groups != self._groups and (self.caps.get("M") is None or sum(groups) != sum(self._groups)):
            # exact-extent graphs (caps=None) are tied to their grouping; with capacities any grouping of the
            # same number of pairs fits as long as its largest group does (checked by the plan below)
            raise ValueError("GraphedStep.load_batch: group sizes changed %s -> %s" % (self._groups, groups))
        n_real = batch.get("_n_real")
        if (n_real is None) != (self._n_real is None):
            raise ValueError("GraphedStep.load_batch: the step was captured %s a device-side pair count"
                             % ("with" if self._n_real is not None else "without"))
        old_groups, self._groups = self._groups, groups
        old_real, self._n_real = self._n_real, n_real
        try:
            self._load_batch(batch, redraw, host, defer_targets)
        except ValueError:
            self._groups, self._n_real = old_groups, old_real
            raise

    def _load_batch(self, batch, redraw, host_given=None, defer_targets=False):
        for k, cur in self.batch.items():
            v = batch.get(k)
            if k == "num_clips" or k.startswith("_"):
                continue
            if torch.is_tensor(cur) and torch.is_tensor(v) and cur.shape != v.shape:
                if k in ("video_feat", "words_id") and cur.shape[1:] == v.shape[1:] and v.shape[0] <= cur.shape[0] \
                        and self._n_real is not None:
                    continue  # the real rows only: the rest of the static input keeps (finite) older rows
                raise ValueError("GraphedStep.load_batch: %s changed shape %s -> %s"
                                 % (k, tuple(cur.shape), tuple(v.shape)))
        if host_given is not None:
            # (autograph._Fetch: every small tensor of a device batch in ONE transfer, the word mask formed on the device)
            host = {k: v for k, v in host_given.items() if torch.is_tensor(v) or isinstance(v, list)}
        else:
            host = {k: (v.detach().cpu() if v.is_cuda and v.numel() * v.element_size() <= self.BIG else v)
                    for k, v in batch.items() if torch.is_tensor(v)}
            for k in ("video_mask", "clip_mask"):
                if k in host and host[k].is_cuda:
                    host[k] = host[k].cpu()
            if host["words_id"].is_cuda:
                host["words_id"] = host["words_id"].cpu()
            for k in ("norm_span", "norm_moment"):
                if isinstance(batch.get(k), list):
                    host[k] = [{kk: vv.detach().cpu() for kk, vv in d.items()} for d in batch[k]]
        # keep the current host draws unless asked to redraw (plan_arrays draws when given None)
        arr, pmeta, tmeta, wm = self._host_arrays(host, None if redraw else self._draws[0],
                                                  None if redraw else self._draws[1], targets=not defer_targets)
        if defer_targets:
            # the criterion's arrays are built and uploaded by finish_targets(), after the forward graph has been launched;
            # until then the arena image keeps the previous batch's (they are not part of this upload)
            arr.update({k: v for k, v in self._arr.items() if k.startswith("t.")})
            self._pending_targets = host
        for k in ("vid_identity", "has_vid_src"):
            if pmeta.get(k) != self._pmeta.get(k):
                raise ValueError("GraphedStep.load_batch: plan.%s changed (%r -> %r): needs its own graph"
                                 % (k, self._pmeta.get(k), pmeta.get(k)))
        self.arena.check(arr)
        # nothing above modified anything; from here on only copies
        self._arr = arr
        self._draws = (arr["p.neg_index"], arr.get("p.masked_words"))
        if defer_targets:
            self.arena.upload(arr, only=[k for k in arr if not k.startswith("t.")])
        else:
            self.arena.upload(arr)
            self.tplan.sizes, self.tplan.sumT = tmeta["sizes"], tmeta["sumT"]
        for k, cur in self.batch.items():
            v = batch.get(k)
            if k.startswith("_"):
                continue
            if k == "num_clips":
                self.batch[k] = v  # host-side bookkeeping only (the step reads the plans)
                continue
            if torch.is_tensor(cur) and torch.is_tensor(v) and ("b." + k) not in arr:
                if not cur.is_cuda:
                    self.batch[k] = v  # host-side inputs (words_weight) are only read by redraw()
                elif v.is_cuda:
                    cur[:v.shape[0]].copy_(v, non_blocking=True)
                else:
                    self._stage_big(k, cur, v)
        self._turn ^= 1
        self._wm_cpu = wm

    _pending_targets = None

    def finish_targets(self):
        """second half of load_batch(defer_targets=True): the criterion's flattened targets, built and uploaded while the
        forward graph runs.  Raises ValueError when they do not fit the captured capacities (the caller then runs the step
        eagerly: nothing but static graph memory has been touched)."""
        host, self._pending_targets = self._pending_targets, None
        tarr, tmeta = self.spec.target_arrays(host, self.caps)
        arr = dict(self._arr)
        arr.update(tarr)
        self.arena.check(arr)
        self._arr = arr
        self.arena.upload(arr, only=list(tarr))
        self.tplan.sizes, self.tplan.sumT = tmeta["sizes"], tmeta["sumT"]

    def load_prepared(self, prep):
        """load_batch for a batch whose host half was done elsewhere (loader.HostPipeline.prepare in a worker process):
        `prep` carries the arena arrays (placeholder draws), the metas, the word mask and the big feature tensors
        (pinned when they came through a DataLoader with pin_memory).  Here: compatibility checks, the two host-RNG
        draws of the reference's forward (in THIS process: its RNG stream), one arena upload, the feature copies.
        Raises ValueError (nothing modified) when the batch does not fit this graph."""
        if prep["spec"] != self.spec:
            raise ValueError("GraphedStep.load_prepared: prepared for another model configuration")
        groups = prep["groups"]
        if groups != self._groups and (self.caps.get("M") is None or sum(groups) != sum(self._groups)):
            raise ValueError("GraphedStep.load_prepared: group sizes changed %s -> %s" % (self._groups, groups))
        if (prep["n_real"] is None) != (self._n_real is None):
            raise ValueError("GraphedStep.load_prepared: pair-count mode differs from the captured step")
        if {k: prep["caps"].get(k) for k in self.caps} != self.caps:
            raise ValueError("GraphedStep.load_prepared: prepared with other capacities %r (graph: %r)" % (prep["caps"], self.caps))
        pmeta, tmeta = prep["pmeta"], prep["tmeta"]
        for k in ("vid_identity", "has_vid_src"):
            if pmeta.get(k) != self._pmeta.get(k):
                raise ValueError("GraphedStep.load_prepared: plan.%s changed (%r -> %r): needs its own graph"
                                 % (k, self._pmeta.get(k), pmeta.get(k)))
        arr = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in prep["arr"].items()}
        self.arena.check(arr)
        big = prep["big"]
        for k, v in big.items():
            cur = self.batch[k]
            # fewer rows than the static input only where the step reads the real pair count from the device (like
            # load_batch: without it the stale rows of the previous batch would take part in the step)
            if cur.shape[1:] != v.shape[1:] or v.shape[0] > cur.shape[0] or (v.shape[0] < cur.shape[0] and self._n_real is None):
                raise ValueError("GraphedStep.load_prepared: %s changed shape %s -> %s" % (k, tuple(cur.shape), tuple(v.shape)))
        # nothing above modified anything; from here on: draws, copies
        self._groups, self._n_real = groups, prep["n_real"]
        self._wm_cpu = prep["wm"]
        self.batch["words_weight"] = prep["words_weight"]
        self.batch["num_clips"] = prep["num_clips"]
        neg = draws.neg_index(self._groups, self._n_real)
        arr["p.neg_index"] = neg
        mw = None
        if "p.masked_words" in arr:
            mw = draws.masked_words(self._wm_cpu, prep["words_weight"])
            arr["p.masked_words"] = mw
        self._arr, self._draws = arr, (neg, mw)
        self.arena.upload(arr)
        self.tplan.sizes, self.tplan.sumT = tmeta["sizes"], tmeta["sumT"]
        for k, v in big.items():
            self._stage_big(k, self.batch[k], v)
        self._turn ^= 1

    _copy_stream = None
    _turn = 0

    def _stage_big(self, k, cur, v):
        """host feature tensor -> static graph input without stalling the compute stream: the (pageable, hence
        host-blocking) H2D runs on a copy stream into one of two device staging buffers while the previous
        replay is still executing; the compute stream then only does a device-to-device copy (27 MB: ~10 us)
        behind an event.  A staging buffer is rewritten only after the D2D copy that read it two batches ago."""
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.dev)
            self._stages, self._d2d_done = {}, {}
        j = self._turn
        st = self._stages.setdefault(k, [None, None])
        if st[j] is None:
            st[j] = torch.empty_like(cur)
        main, cs = torch.cuda.current_stream(), self._copy_stream
        n = v.shape[0]  # (fewer rows than the static input: a padded batch's real pairs)
        with torch.cuda.stream(cs):
            ev = self._d2d_done.get((k, j))
            if ev is not None:
                cs.wait_event(ev)
            st[j][:n].copy_(v, non_blocking=True)
            done = torch.cuda.Event()
            done.record(cs)
            from .loader import H2D_EVENTS
            H2D_EVENTS.append(done)  # (a ring slot of loader.PinnedRing is rewritten only after the copy that read it)
        main.wait_event(done)
        cur[:n].copy_(st[j][:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(main)
        self._d2d_done[(k, j)] = ev

    def redraw(self):
        """New negative-query indices and MLM word choices (host RNG, like the reference does on every forward),
        written into the static index tensors the graph reads (one arena upload)."""
        m = self.model
        neg = draws.neg_index(self._groups, self._n_real)
        mw = None
        if hasattr(self.plan, "masked_words"):
            mw = draws.masked_words(self._wm_cpu, self.batch["words_weight"])
        self._draws = (neg, mw)
        self._arr["p.neg_index"] = neg
        if mw is not None:
            self._arr["p.masked_words"] = mw
        if self._pull is None:
            self.arena.upload(self._arr, only=self.DRAWN)  # ~1 KB instead of the whole 57 KB arena
        # (with the ring: the next replay publishes self._arr's draws itself)

    def set_draws(self, neg_index, masked_words=None):
        """replay with given host draws (tests / reproducing a recorded step)"""
        neg = neg_index.cpu().numpy()
        mw = masked_words.cpu().numpy().astype(bool) if masked_words is not None else self._draws[1]
        self._draws = (neg, mw)
        self._arr["p.neg_index"] = neg
        if mw is not None:
            self._arr["p.masked_words"] = mw
        self.arena.upload(self._arr)

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


class SplitGraphedStep(GraphedStep):
    """The same captured step cut where the reference's loop holds the reins (train.py:64-72): THREE graphs over one
    memory pool -- `model(**batch)` | `criterion(outputs, batch)` | `loss.backward()` -- so that the unchanged caller's
    `optimizer.zero_grad()` between the second and the third lands where it always did (mesm_amd/autograph.py replays
    them from inside MESM.forward, Criterion.forward and ONE autograd node's backward: the structure of
    torch.cuda.make_graphed_callables, with this build's arena / draw / capacity machinery).  The activations the
    backward graph reads live in the pool until the next forward replay overwrites them: one forward in flight."""

    def __init__(self, *a, **kw):
        if kw.get("reducer") is not None:
            raise ValueError("SplitGraphedStep: gradient collectives ride in the one-graph step (GraphedStep)")
        super().__init__(*a, **kw)

    def _capture(self, side):
        if self._pull is not None:
            raise RuntimeError("SplitGraphedStep: MESM_STEP_PULL is a one-graph option")
        pool = torch.cuda.graph_pool_handle()
        self.g_fwd, self.g_crit, self.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        b = self.batch
        kw = dict(stream=side, pool=pool, capture_error_mode="thread_local")
        with torch.cuda.graph(self.g_fwd, **kw):
            self.counter.add_(1)
            out = self.model(**b, dataset_name=self.dataset_name, is_training=True, plan=self.plan)
        with torch.cuda.graph(self.g_crit, **kw):
            losses, total = self.crit(out, b, True)
        self.model.zero_grad(set_to_none=True)
        with torch.cuda.stream(side):
            self.seed = torch.ones_like(total)
        with torch.cuda.graph(self.g_bwd, **kw):
            total.backward(self.seed)
        gb = self.model.gradbuf()
        self.grad_params = [p for p in gb.params if p.grad is not None]
        det = lambda t: t.detach() if torch.is_tensor(t) else t
        self.out = {k: ([{kk: det(vv) for kk, vv in d.items()} for d in v] if isinstance(v, list) else det(v))
                    for k, v in out.items()}
        self.total = total.detach()
        vals = list(losses.values())
        # the loss entries are elements of ONE vector (criterion.loss_entries): a replay hands out a copy of it, one launch
        pack = getattr(vals[0], "_pack", None) if vals else None
        same = pack is not None and all(getattr(v, "_pack", None) is pack for v in vals)
        self.loss_vec = pack.vec.detach() if same else None
        self.loss_index = {k: int(v._i) for k, v in losses.items()} if same else None
        self.losses = {k: v.detach() for k, v in losses.items()}

    def run(self, redraw=True):
        raise RuntimeError("SplitGraphedStep is replayed in three parts (forward_replay / criterion_replay / backward_replay)")

    def forward_replay(self):
        # the 273-address comparison of GraphedStep.run costs 0.1 ms of the caller's critical path: here it runs when the
        # model says tensors may have moved (MESM._apply / a re-flattening bump _addr_gen), and the two flat buffers'
        # addresses are compared every time
        m = self.model
        gen = m.__dict__.get("_addr_gen", 0)
        fp = m._flat_params
        quick = (fp.data_ptr() if fp is not None else 0, m.gradbuf().flat.data_ptr())
        if gen != getattr(self, "_addr_gen", None) or quick != getattr(self, "_quick_ptrs", None):
            if self._param_ptrs() != self._ptrs:
                raise RuntimeError("SplitGraphedStep: a parameter or the gradient buffer moved after capture")
            self._addr_gen, self._quick_ptrs = gen, quick
        self.g_fwd.replay()
        return self.out

    def criterion_replay(self):
        self.g_crit.replay()
        if self.loss_vec is not None:
            from .criterion import loss_entries
            losses = loss_entries(self.loss_vec.clone(), self.loss_index)  # (float() of all entries = one transfer)
        else:
            losses = {k: v.clone() for k, v in self.losses.items()}
        return losses, self.total

    def backward_replay(self, grad=None):
        if grad is not None:
            self.seed.copy_(grad)
        self.g_bwd.replay()


class StepCache:
    """One captured graph per batch-shape bucket: `run(batch)` replays a graph whose shape key
    (N, Lv, Lw, feature dims) and capture-time capacities (largest video group, GT-clip count, group video
    length, target windows) fit the batch, capturing a new one when none does.  `pad=(Lv_cap, Lw_cap)` first
    pads every batch to fixed clip / word extents (masked positions, what the reference's own collate does up to
    the batch maximum), so that the key is the number of pairs alone: real loaders emit a different (N, Lv, Lw,
    grouping) almost every batch -- items are videos with all their queries, dataset/base.py:164-207 -- and
    only a handful of distinct N.  `captures` counts the graphs built."""

    # batch entries with a clip / word axis at dim 1
    CLIP_KEYS = ("video_feat", "video_mask", "clip_mask", "saliency_label")
    WORD_KEYS = ("words_id", "words_mask", "words_weight", "unknown_mask", "words_label")

    def __init__(self, model, criterion, dataset_name, reducer=None, max_graphs=16, pad=None, pairs=None,
                 group_caps=None):
        """pairs: also pad the PAIR axis up to the next multiple of `pairs` (batching.pad_pairs: dummy pairs behind
        the real ones, the real count a device scalar of the captured step) -- the reference's loaders emit a different
        number of pairs almost every batch (an item is a video with all its queries, dataset/base.py:116-162), and
        with pairs=8 a QVHighlights epoch of 12-group batches replays from a handful of graphs.
        group_caps: ascending group-size buckets, e.g. (5, 9): a batch is served by a graph captured with room for
        video groups of up to the smallest bucket that holds its largest group (9 = the QVHighlights maximum), so that
        (pair bucket, group bucket) alone decide which graph a batch replays -- groups cost SS-MESM key slots
        (group size x clips per pair), which is why there is more than one bucket."""
        self.model, self.crit, self.dataset_name = model, criterion, dataset_name
        self.reducer = reducer
        self.steps = {}
        self.max_graphs = max_graphs
        self.pad = pad
        self.pairs = pairs
        self.group_caps = tuple(sorted(group_caps)) if group_caps else None
        self.captures = 0
        self.replays = 0

    @staticmethod
    def key(batch):
        # (the pair extent from a mask: big feature tensors of a pair-padded batch arrive with their real rows only)
        n = batch["video_mask"].shape[0]
        return ((n,) + tuple(batch["video_feat"].shape[1:]), (n,) + tuple(batch["words_id"].shape[1:]))

    @staticmethod
    def _pad_dim1(t, L):
        if t is None or not torch.is_tensor(t) or t.dim() < 2 or t.shape[1] >= L:
            return t
        out = t.new_zeros((t.shape[0], L) + tuple(t.shape[2:]))
        out[:, :t.shape[1]] = t
        return out

    def padded(self, batch):
        """the batch with its clip / word axes zero-padded to `pad` (new dict; tensors shared when nothing to do)"""
        if self.pad is None:
            return batch
        Lv, Lw = self.pad
        if batch["video_feat"].shape[1] > Lv or batch["words_id"].shape[1] > Lw:
            raise ValueError("StepCache: batch extents (%d clips, %d words) exceed pad=%r"
                             % (batch["video_feat"].shape[1], batch["words_id"].shape[1], self.pad))
        b = dict(batch)
        for k in self.CLIP_KEYS:
            if k in b:
                b[k] = self._pad_dim1(b[k], Lv)
        for k in self.WORD_KEYS:
            if k in b:
                b[k] = self._pad_dim1(b[k], Lw)
        return b

    def run(self, batch, redraw=True):
        batch = self.padded(batch)
        if self.pairs:
            from .batching import pad_pairs
            n = batch["video_feat"].shape[0]
            batch = pad_pairs(batch, _round_up(n, self.pairs))
        k = self.key(batch)
        gcap = None
        if self.group_caps:
            gmax = int(batch["num_clips"].max())
            gcap = next((c for c in self.group_caps if c >= gmax), gmax)
            k = k + (gcap,)
        for gs in self.steps.get(k, []):
            try:
                gs.load_batch(batch, redraw=redraw)
            except ValueError:
                continue
            self.replays += 1
            return gs.run(redraw=False), gs
        if sum(len(v) for v in self.steps.values()) >= self.max_graphs:
            self.steps.pop(next(iter(self.steps)))
        from .synthetic import to_device
        static = to_device({kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in batch.items() if kk != "_n_real"},
                           next(self.model.parameters()).device)
        if "_n_real" in batch:
            static["_n_real"] = batch["_n_real"]
            P = static["video_mask"].shape[0]
            for kk in ("video_feat", "words_id"):  # big tensors arrive with their real rows: pad on the device
                t = static[kk]
                if t.shape[0] < P:
                    static[kk] = torch.cat([t, t[:1].expand((P - t.shape[0],) + tuple(t.shape[1:]))]).contiguous()
        gs = GraphedStep(self.model, self.crit, static, self.dataset_name, warmup=1, reducer=self.reducer,
                         caps="auto", group_cap=gcap)
        self.steps.setdefault(k, []).append(gs)
        self.captures += 1
        return gs.run(redraw=redraw), gs

    def pipeline(self, keep_raw=True):
        """the host half of `run` as a picklable object for loader workers (loader.HostPipeline).  keep_raw=False once
        every bucket of the stream has its graph: the raw batch (needed only to capture) then does not travel."""
        from .hostplan import HostSpec
        from .loader import HostPipeline
        return HostPipeline(HostSpec.from_model(self.model, self.crit, self.dataset_name), pad=self.pad, pairs=self.pairs,
                            group_caps=self.group_caps, keep_raw=keep_raw)

    def run_prepared(self, prep):
        """`run` for a batch prepared by a loader worker (loader.HostPipeline.prepare): the graph of its bucket key takes
        the arrays as they are (the two host draws are made here); a bucket without a graph yet goes through `run` on
        the raw batch the worker kept alongside (prep["raw"]), which captures it."""
        for gs in self.steps.get(prep["key"], []):
            try:
                gs.load_prepared(prep)
            except ValueError:
                continue
            self.replays += 1
            return gs.run(redraw=False), gs
        if prep.get("raw") is None:
            raise ValueError("StepCache.run_prepared: no graph for bucket %r and the prepared batch carries no raw batch" % (prep["key"],))
        return self.run(prep["raw"], redraw=True)
