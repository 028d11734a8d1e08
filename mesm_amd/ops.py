"""autograd.Function blocks over the gfx950 kernels.

Granularity is chosen for fusion, not to mirror torch.nn: a Function covers a whole
nn.Linear(+epilogue), a whole FFN, a whole nn.MultiheadAttention, ... and its backward is
hand-written on the same kernels.  Parameter gradients are accumulated in place into the
model's flat gradient buffer (gradbuf.py); autograd only routes activation gradients.

All activations are batch-first (N, L, d) contiguous fp32; the reference's (L, N, d)
layout (transformer.py:93-96) is never materialised.
"""
import os

import torch
from torch.autograd import Function

from . import kernels as kn
from ._lib import ACT_NONE, ACT_PRELU, ACT_RELU
from .gradbuf import flush_ready, grad_target

NO_DROP = (0.0, 0)


class DropState:
    """Per-step dropout seeds: every dropout site of a forward draws the next seed; the
    backward replays the (p, seed) it saved.  Off (p = 0) in eval mode."""

    def __init__(self):
        self.training = False
        self.base = 0
        self.counter = 0

    def begin(self, training, base):
        self.training = training
        self.base = int(base) & 0x7FFFFFFF
        self.counter = 0

    def next(self, p):
        if not self.training or p <= 0.0:
            return NO_DROP
        self.counter += 1
        return (float(p), (self.base * 1000003 + self.counter * 104729) & 0xFFFFFFFF)


drop_state = DropState()


def _2d(t):
    return t.reshape(-1, t.shape[-1])


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# reduce indices per workgroup a small weight gradient aims at (0: the standalone rule only) and the workgroups it may become:
# same-device step times 3.748 (off) / 3.70 (512) / 3.684 (600) / 3.73 (800) ms; 512 workgroups 3.72 (profiles/r5d/dw_split_ab.txt)
_DW_SMALL = int(os.environ.get("MESM_DW_SMALL", "2"))  # cap of the split for gradients over fewer than 512 rows (0: none; same-device 3.638 -> 3.624 ms)
_DW_WGS = int(os.environ.get("MESM_DW_WGS", "256"))
_DW_DEPTH = int(os.environ.get("MESM_DW_DEPTH", "600"))


def _dw_split(n_out, k_in, rows):
    """Split-k factor of a weight-gradient GEMM (n_out x k_in output, `rows` reduce indices): aim at one
    workgroup per CU (256) of the tile the dispatcher will pick -- 32 x 32 for small outputs, 64 x 64
    beyond (tools/gemm_sweep.py: 256x256x2400 8.6 us at split 4 against 11.3 at 16; 512x256 best at 2;
    1024x256x4800 best at 4; the unaligned 2818 / 5003-wide ones gain 15 % at 4)."""
    t32 = ((n_out + 31) // 32) * ((k_in + 31) // 32)
    t64 = ((n_out + 63) // 64) * ((k_in + 63) // 64)
    if t32 <= 128:
        s = 256 // t32
    elif t64 <= 128:
        s = 256 // t64
    else:
        s = 4 if t64 < 256 else 2  # 256 x 2818 x 2400: 49 us at 4 (52 at 1); 5003 x 256 x 1024: 37 us at 2 (43 at 1 or 4)
    if _DW_DEPTH > 0 and t64 <= 128:
        # inside the step a weight gradient never runs alone: it shares its launch with the dX product of the same block,
        # whose tiles are 256 deep.  At split 4 a 256 x 256 gradient over 4800 rows is 64 workgroups of 1200 reduce indices
        # -- they end the launch alone, 15 us after the 300 dX tiles beside them.  Slices about as deep as the neighbours'
        # (and still at most one workgroup per CU for this member) end together with them.
        s = max(s, min(rows // _DW_DEPTH, _DW_WGS // t64))
    if _DW_SMALL > 0 and rows < 512:
        s = min(s, _DW_SMALL)
    return max(1, min(s, rows // 64, 32))


def _accum_dw(dz, xin, gw, gb_view, x2=None, b_act=ACT_NONE, b_drop=NO_DROP, slope=None):
    """gw (N_out, K_in) += dz^T @ f(xin (+x2));  gb_view (N_out) += colsum(dz)."""
    rows = dz.shape[0]
    s = _dw_split(gw.shape[0], gw.shape[1], rows)
    kn.gemm(dz, xin, gw, trans_a=True, B2=x2, colsum=gb_view, b_act=b_act, b_drop=b_drop,
            slope=slope, split_k=s, accumulate=2)  # atomic: two streams may add into the same view


def _rows(t, rows):
    return t if rows is None else t[rows[0]:rows[1]]


class DropSink:
    """Hand-over of `mask * dY` for the post-norm pattern y = LayerNorm(res + dropout(block(.))).
    The block's wrapper hangs a sink on its output tensor; the LayerNorm that consumes that tensor
    makes its backward kernel write dropout(dX; p, seed) as a second output (mesm_layernorm_bwd2) and
    parks it here; the block's backward, which autograd runs next and which receives that same dX as
    its dY, takes it instead of launching an element-wise mask kernel.  Any other dataflow (output
    consumed elsewhere, views in between, eval mode) simply never fills the sink and the block falls
    back to kn.dropout."""
    __slots__ = ("p", "seed", "src", "dz")

    def __init__(self, drop):
        self.p, self.seed = drop
        self.src = self.dz = None


class ReluSink:
    """Hand-over of the ReLU mask for y = relu(linear(.)): the block that CONSUMES y (a Linear's dX GEMM epilogue, a
    LayerNorm's backward store) writes its d y already masked by y > 0 and parks the tensor here; the producing
    block's backward, which receives that same tensor as its gradient, then skips its own mask launch
    (act_bias_bwd: 16 per step).  Any other dataflow -- y has several consumers, so autograd hands over a sum --
    fails the identity check and the producer masks as before (masking twice is harmless: the mask is idempotent
    and linear)."""
    __slots__ = ("t",)

    def __init__(self):
        self.t = None


def _relu_masked(rsink, dy2):
    if rsink is not None and rsink.t is not None:
        t, rsink.t = rsink.t, None
        return t.data_ptr() == dy2.data_ptr() and t.numel() == dy2.numel()
    return False


def _sink_for(out_drop):
    return DropSink(out_drop) if out_drop[0] > 0 else None


def _tag(y, sink):
    if sink is not None:
        y._mesm_sink = sink
    return y


def _masked_dy(sink, dy2, out_drop):
    """dropout-mask replay on the incoming gradient of a block whose output went through `out_drop`"""
    if sink is not None and sink.dz is not None and sink.src is not None \
            and sink.src.data_ptr() == dy2.data_ptr() and sink.src.numel() == dy2.numel():
        dz = sink.dz
        sink.src = sink.dz = None
        return _2d(dz)
    return kn.dropout(dy2, *out_drop)


# ----------------------------------------------------------------------------- blocks, phases, lockstep
# A BLOCK is a class with two static methods, fwd(ctx, *args) and bwd(ctx, *grads), written like the forward /
# backward of a torch.autograd.Function -- except that they may be GENERATORS: every `yield` ends a LAUNCH PHASE
# (kernels.phase: the gemm / layernorm / attention launches made since the previous yield are mutually
# independent and are issued together, grouped by kind).  `make_fn(block)` wraps a block as an ordinary
# autograd.Function (its phases run one after the other).  `par([...])` runs SEVERAL blocks as ONE autograd node
# whose phases advance in LOCKSTEP: phase k of every block shares its grouped launches, forward and backward.
# That is how the step's independent chains -- the enhance stack beside the SS-MESM stack, the MLM stack beside
# the rest of the SS-MESM stack (model.py:184-207, 307-332) -- stop being one serial chain of latency-bound
# launches: their small problems ride in the launches of the large ones, on the same hardware queue.
import inspect
import os


def _drive(thunks):
    """Run blocks in lockstep, one launch phase per round; returns their return values.  thunks: zero-argument
    callables returning either the block's result (a single-phase block) or a generator (one `yield` per phase
    boundary).  Parameter-gradient readiness (gradbuf.flush_ready, the hook a DDP reducer listens to) is reported
    once every phase of every block has been ISSUED -- a block acquires its gradient views up front, their kernels
    may sit in a later phase."""
    n = len(thunks)
    results = [None] * n
    gens = [None] * n
    outer = kn._phase is not None  # the caller's own phase (gemm_group) collects: only single-phase blocks fit in it
    live = []
    kn.defer_side(+1)  # slope-gradient reductions ride in the block's later GEMM launches: flushed once, below
    try:
        return _drive_phases(thunks, results, gens, outer, live)
    except BaseException:
        kn.gemm_drop_side()
        raise
    finally:
        kn.defer_side(-1)


def _drive_phases(thunks, results, gens, outer, live):
    with kn.phase():
        for i, t in enumerate(thunks):
            r = t()
            if inspect.isgenerator(r):
                gens[i] = r
                try:
                    next(r)
                    live.append(i)
                except StopIteration as e:
                    results[i] = e.value
            else:
                results[i] = r
    while live:
        assert not outer, "a multi-phase block was called inside an open launch phase"
        nxt = []
        with kn.phase():
            for i in live:
                try:
                    next(gens[i])
                    nxt.append(i)
                except StopIteration as e:
                    results[i] = e.value
        live = nxt
    if not outer:
        kn.gemm_flush_side()  # (slope-gradient reductions no later GEMM launch of the block has carried)
        flush_ready()
    return results


class _Sub:
    """What a block sees as `ctx` when it runs inside a par() node."""

    def __init__(self, needs):
        self.needs_input_grad = tuple(needs)
        self._saved = ()
        self.materialize = True
        self.nondiff = []

    def save_for_backward(self, *ts):
        self._saved = ts

    @property
    def saved_tensors(self):
        return self._saved

    def set_materialize_grads(self, v):
        self.materialize = bool(v)

    def mark_non_differentiable(self, *ts):
        self.nondiff.extend(ts)


# tools/fanin.py: {(data_ptr, shape) of a block INPUT: [blocks that returned a gradient for it]} -- which tensors get
# their gradient from several nodes (the engine then adds them with element-wise launches).  None = off.
FANIN_DEBUG = None


def _dbg_inputs(args):
    return [(a.data_ptr(), tuple(a.shape)) if torch.is_tensor(a) and a.is_floating_point() and a.requires_grad else None
            for a in args]


def _dbg_record(keys, grads, name):
    for k, g in zip(keys, grads):
        if k is not None and g is not None:
            FANIN_DEBUG.setdefault(k, []).append(name)


def make_fn(block):
    """A block as a stand-alone autograd.Function."""

    class Fn(Function):
        @staticmethod
        def forward(ctx, *args):
            if FANIN_DEBUG is not None:
                ctx._dbg = _dbg_inputs(args)
            return _drive([lambda: block.fwd(ctx, *args)])[0]

        @staticmethod
        def backward(ctx, *gs):
            out = _drive([lambda: block.bwd(ctx, *gs)])[0]
            if FANIN_DEBUG is not None:
                _dbg_record(ctx._dbg, out, block.__name__)
            return out

    Fn.__name__ = Fn.__qualname__ = block.__name__.replace("Block", "Fn")
    return Fn


class ParFn(Function):
    """Several blocks as one autograd node, phases in lockstep.  apply(spec, *flat): spec = ((block, nargs), ...),
    flat = the blocks' arguments back to back; returns the blocks' outputs back to back."""

    @staticmethod
    def forward(ctx, spec, *flat):
        ctx.set_materialize_grads(False)
        if FANIN_DEBUG is not None:
            ctx._dbg = _dbg_inputs(flat)
        subs, runs, pos = [], [], 0
        for block, n in spec:
            sub = _Sub(ctx.needs_input_grad[1 + pos:1 + pos + n])
            subs.append(sub)
            runs.append(lambda block=block, sub=sub, a=flat[pos:pos + n]: block.fwd(sub, *a))
            pos += n
        outs = _drive(runs)
        flat_out, counts, saved, spans, meta, nondiff = [], [], [], [], [], []
        for sub, o in zip(subs, outs):
            t = o if isinstance(o, tuple) else (o,)
            counts.append(len(t))
            flat_out.extend(t)
            meta.append([(x.shape, x.dtype, x.device) if torch.is_tensor(x) else None for x in t])
            spans.append((len(saved), len(sub._saved)))
            saved.extend(sub._saved)
            sub._saved = ()
            nondiff.extend(sub.nondiff)
        ctx.save_for_backward(*saved)
        if nondiff:
            ctx.mark_non_differentiable(*nondiff)
        ctx.subs, ctx.spec, ctx.counts, ctx.spans, ctx.meta = subs, spec, counts, spans, meta
        return tuple(flat_out)

    @staticmethod
    def backward(ctx, *gs):
        saved = ctx.saved_tensors
        runs, slots, pos = [], [], 0
        res = [None] * len(ctx.spec)
        for i, ((block, n), sub, cnt, (a, m)) in enumerate(zip(ctx.spec, ctx.subs, ctx.counts, ctx.spans)):
            g = list(gs[pos:pos + cnt])
            pos += cnt
            if all(x is None for x in g):  # nothing reached this block's outputs: autograd would not call it
                res[i] = (None,) * n
                continue
            if sub.materialize:
                g = [torch.zeros(mt[0], dtype=mt[1], device=mt[2]) if (x is None and mt is not None) else x
                     for x, mt in zip(g, ctx.meta[i])]
            sub._saved = saved[a:a + m]
            runs.append(lambda block=block, sub=sub, g=g: block.bwd(sub, *g))
            slots.append(i)
        for i, r in zip(slots, _drive(runs)):
            res[i] = tuple(r) if isinstance(r, (tuple, list)) else (r,)
        out = [None]
        for r, (block, n) in zip(res, ctx.spec):
            assert len(r) == n, (block.__name__, len(r), n)
            if FANIN_DEBUG is not None:
                _dbg_record(ctx._dbg[len(out) - 1:len(out) - 1 + n], r, "par:" + block.__name__)
            out.extend(r)
        return tuple(out)


class Call:
    """A deferred block invocation: run(call) executes it alone, par([calls]) in lockstep with others.
    post(out) -> what the caller gets (sink tagging)."""
    __slots__ = ("block", "args", "post")

    def __init__(self, block, args, post=None):
        self.block, self.args, self.post = block, tuple(args), post


_SERIAL_PAR = os.environ.get("MESM_SERIAL_PAR") == "1"  # A/B switch: every call of a round as a node of its own


def par(calls):
    """Run the calls as ONE autograd node with their launch phases in lockstep; returns one result per call."""
    calls = [c for c in calls]
    if len(calls) == 1 or _SERIAL_PAR:
        return [run(c) for c in calls]
    spec = tuple((c.block, len(c.args)) for c in calls)
    flat = [a for c in calls for a in c.args]
    outs = ParFn.apply(spec, *flat)
    res, pos = [], 0
    counts = ParFn_counts(outs, calls)
    for c, n in zip(calls, counts):
        o = outs[pos:pos + n]
        pos += n
        o = o[0] if n == 1 else tuple(o)
        res.append(c.post(o) if c.post is not None else o)
    return res


def ParFn_counts(outs, calls):
    """outputs per call: every block declares N_OUT (a number or a function of its arguments)"""
    counts = []
    for c in calls:
        n = c.block.N_OUT
        counts.append(n(*c.args) if callable(n) else n)
    assert sum(counts) == len(outs), (counts, len(outs))
    return counts


def lockstep(chains):
    """Run several CHAINS side by side.  A chain is a generator that yields ops.Call objects one at a time and
    receives each call's result back (`x = yield ops.linear_call(...)`); a LIST of calls (entries may be None) forks
    the chain for one round and comes back as the list of results; `yield None` sits a round out.  Every round
    the calls the chains have just yielded run as ONE autograd node with their launch phases in lockstep (par):
    independent stacks -- enhance beside SS-MESM, MLM beside the rest of SS-MESM, the input projections of every
    modality -- share their launches instead of queueing behind each other.  Returns the chains' return values."""
    chains = list(chains)
    n = len(chains)
    results, send = [None] * n, [None] * n
    live = list(range(n))
    while live:
        calls, who, nxt = [], [], []
        for i in live:
            try:
                c = chains[i].send(send[i])
            except StopIteration as e:
                results[i] = e.value
                continue
            send[i] = None
            nxt.append(i)
            if isinstance(c, (list, tuple)):  # a chain forks: several independent calls in this round
                send[i] = [None] * len(c)
                for j, cj in enumerate(c):
                    if cj is not None:
                        calls.append(cj)
                        who.append((i, j))
            elif c is not None:
                calls.append(c)
                who.append((i, None))
        live = nxt
        if calls:
            for (i, j), o in zip(who, par(calls)):
                if j is None:
                    send[i] = o
                else:
                    send[i][j] = o
    return results


def seq(chain):
    """a chain on its own"""
    return lockstep([chain])[0]


def delayed(chain, rounds):
    """the chain, starting `rounds` rounds later"""
    for _ in range(rounds):
        yield None
    return (yield from chain)


_FN_CACHE = {}


def run(call):
    fn = _FN_CACHE.get(call.block)
    if fn is None:
        fn = _FN_CACHE[call.block] = make_fn(call.block)
    o = fn.apply(*call.args)
    return call.post(o) if call.post is not None else o


# ----------------------------------------------------------------------------- fork (n-ary gradient fan-in)
class ForkFn(Function):
    """x -> n aliases of x for n consumers; the backward sums the gradients that arrive in ONE launch (kn.add_n) where
    the autograd engine would add them pairwise, n - 1 element-wise launches (19 fan-in adds per step, four in
    a row on the decoder's query tensor)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        live = [_c(g) for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        while len(live) > 8:
            live = [kn.add_n(live[:8])] + live[8:]
        return (kn.add_n(live) if live[0].numel() % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in live)
                else torch.stack(live).sum(0)), None


def fork(x, n):
    """n aliases of x whose gradients are summed by one launch; x itself when nothing is to be gained"""
    if n < 3 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return ForkFn.apply(x, n)


# ----------------------------------------------------------------------------- stacked outputs written in place
class Slot:
    """One slot buf[k] of a stacked output (`torch.stack` without the copy): the kernel that produces the k-th tensor
    writes it there (layer_norm_call(slot=), ref_init / ref_step), and `stacked` hands the buffer out as the stack.  A plain
    object, not a tensor, so that an autograd node does not see the buffer as one of its inputs."""
    __slots__ = ("buf", "k", "t")

    def __init__(self, buf, k):
        self.buf, self.k, self.t = buf, k, buf[k]


class StackedFn(Function):
    """torch.stack(xs) of tensors that already live in the slots of `holder[0]`, in order: no copy forward (the decoder's
    per-layer outputs and reference points, transformer.py:411-415, were two concatenation launches per step); backward
    hands every producer its slice of the gradient, like torch.stack's own."""

    @staticmethod
    def forward(ctx, holder, *xs):
        buf = holder[0]
        assert buf.shape[0] == len(xs)
        for k, x in enumerate(xs):
            assert x.data_ptr() == buf[k].data_ptr() and x.shape == buf.shape[1:] and x.is_contiguous(), \
                "stacked: tensor %d does not live in its slot" % k
        ctx.n = len(xs)
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g[k] for k in range(ctx.n))


def stacked(buf, xs):
    return StackedFn.apply([buf], *xs)


# ----------------------------------------------------------------------------- Linear
class LinearBlock:
    """y = dropout_out( relu?( dropout_in(x [+ x2]) @ W[rows]^T + b[rows] ) ) [+ residual].

    Covers nn.Linear sites with their neighbours fused: with_pos_embed add (x2), the input
    dropout of LinearLayer (model.py:421-431), ReLU (model.py:408,432), and
    `residual + dropout(linear(.))` (transformer.py:534,538,645,648,753,792,795).
    """
    N_OUT = 1

    @staticmethod
    def fwd(ctx, x, x2, residual, w, b, rows, relu, in_drop, out_drop, sink=None, rsink=None, in_relu=None):
        ctx.sink, ctx.rsink, ctx.in_relu = sink, rsink, in_relu
        wv, bv = _rows(w, rows), (_rows(b, rows) if b is not None else None)
        x = _c(x)
        x2c = _c(x2) if x2 is not None else None
        K = x.shape[-1]
        N = wv.shape[0]
        y = (torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=torch.float32) if relu
             else kn.deep_out(x.shape[:-1] + (N,), K, x.device, fwd=True))
        res2 = _2d(_c(residual)) if residual is not None else None
        kn.gemm(_2d(x), wv, _2d(y), trans_b=True, A2=_2d(x2c) if x2c is not None else None,
                bias=bv, e_act=ACT_RELU if relu else ACT_NONE, a_drop=in_drop, e_drop=out_drop,
                residual=res2)
        ctx.save_for_backward(x, x2c, y if relu else None)
        ctx.w, ctx.b, ctx.rows = w, b, rows
        ctx.relu, ctx.in_drop, ctx.out_drop = relu, in_drop, out_drop
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def bwd(ctx, dy):
        x, x2, y = ctx.saved_tensors
        w, b, rows = ctx.w, ctx.b, ctx.rows
        dy = _c(dy)
        dy2 = _2d(dy)
        if ctx.out_drop[0] > 0:
            dz = _masked_dy(ctx.sink, dy2, ctx.out_drop)
        elif ctx.relu:
            dz = dy2 if _relu_masked(ctx.rsink, dy2) else kn.act_bias_bwd(dy2, _2d(y), ACT_RELU)
        else:
            dz = dy2
        gw, wdirect = grad_target(w)
        gb, bdirect = grad_target(b) if b is not None else (None, True)
        dx = None
        need_dx = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        wv = _rows(w, rows)
        if SKINNY_BWD and wv.shape[0] <= kn.SKINNY_OUT and x2 is None and ctx.in_drop[0] == 0.0 and dz.is_contiguous():
            # a 1-4 feature head: dX, dW and db from one launch instead of two degenerate GEMMs
            ir = ctx.in_relu
            dx = kn.skinny_linear_bwd(dz, _2d(x), wv, _rows(gw, rows), _rows(gb, rows) if gb is not None else None,
                                      need_dx=need_dx, relu_mask=ir is not None and need_dx)
            if dx is not None:
                dx = dx.view(x.shape)
                if ir is not None:
                    ir.t = dx
            return (dx if ctx.needs_input_grad[0] else None, None,
                    dy if ctx.has_res and ctx.needs_input_grad[2] else None,
                    None if wdirect else gw, None if (b is None or bdirect) else gb,
                    None, None, None, None, None, None, None)
        # dW and dX are independent: one launch when both are small
        _accum_dw(dz, _2d(x), _rows(gw, rows), _rows(gb, rows) if gb is not None else None,
                  x2=_2d(x2) if x2 is not None else None, b_drop=ctx.in_drop)
        if need_dx:
            ir = ctx.in_relu if (x2 is None and ctx.in_drop[0] == 0.0) else None
            dx = kn.deep_out(x.shape, dz.shape[1], x.device)  # (the ReLU mask of the ir route is linear in the k-slices too)
            if ir is not None:  # x = relu(z) of the previous Linear: write d z (see ReluSink)
                kn.gemm(dz, _rows(w, rows), _2d(dx), aux=_2d(x), e_actgrad=ACT_RELU)
                ir.t = dx
            else:
                kn.gemm(dz, _rows(w, rows), _2d(dx), e_drop=ctx.in_drop)
        return (dx if ctx.needs_input_grad[0] else None,
                dx if (x2 is not None and ctx.needs_input_grad[1]) else None,
                dy if ctx.has_res and ctx.needs_input_grad[2] else None,
                None if wdirect else gw, None if (b is None or bdirect) else gb,
                None, None, None, None, None, None, None)


# A/B switch: MESM_SKINNY_BWD=0 sends the 1-4 feature heads' backward through the GEMM entry again
SKINNY_BWD = os.environ.get("MESM_SKINNY_BWD", "1") != "0"

# set only by mesm_amd.testing.no_relu() (a context manager for the kink control run of tests/test_model_gpu.py)
_NO_RELU = False


def linear_call(x, w, b, *, x2=None, residual=None, rows=None, relu=False, in_drop=NO_DROP,
                out_drop=NO_DROP):
    if _NO_RELU:
        relu = False
    sink = _sink_for(out_drop)
    rsink = ReluSink() if (relu and out_drop[0] == 0.0 and residual is None) else None

    def post(y):
        if rsink is not None:
            y._mesm_relu = rsink
        return _tag(y, sink)

    return Call(LinearBlock, (x, x2, residual, w, b, rows, relu, in_drop, out_drop, sink, rsink,
                              getattr(x, "_mesm_relu", None)), post)


def linear(x, w, b, **kw):
    return run(linear_call(x, w, b, **kw))


# ----------------------------------------------------------------------------- FFN
class FFNBlock:
    """y = residual + dropout_out( dropout_mid(prelu(x W1^T + b1)) W2^T + b2 ).

    linear2(dropout(activation(linear1(.)))) with activation = nn.PReLU (one learnable
    slope) and the residual add: transformer.py:537-538, 603-604, 608-609, 647-648, 794-795.
    The hidden activation h = dropout_mid(prelu(z)) is written once by the first GEMM's epilogue and
    saved next to z: as an operand transform of the second GEMM and of the dW2 GEMM the PReLU +
    mask hash was recomputed by every output tile (82 us vs 47 us for the 4800 x 256 x 1024 GEMM).
    """
    N_OUT = 1

    @staticmethod
    def fwd(ctx, x, residual, w1, b1, slope, w2, b2, mid_drop, out_drop, sink=None):
        ctx.sink = sink
        x = _c(x)
        F_ = w1.shape[0]
        z = torch.empty(x.shape[:-1] + (F_,), device=x.device, dtype=torch.float32)
        h = torch.empty_like(z)
        # one pass writes z (pre-activation, kept for the PReLU gradient) and h = dropout(prelu(z))
        kn.gemm(_2d(x), w1, _2d(h), trans_b=True, bias=b1, e_act=ACT_PRELU, slope=slope, e_drop=mid_drop,
                pre_out=_2d(z))
        yield
        y = kn.rows_out(x, K=F_, fwd=True)
        kn.gemm(_2d(h), w2, _2d(y), trans_b=True, bias=b2, e_drop=out_drop,
                residual=_2d(_c(residual)) if residual is not None else None)
        ctx.save_for_backward(x, z, h)
        ctx.res_is_x = residual is not None and residual.data_ptr() == x.data_ptr() and residual.shape == x.shape
        ctx.params = (w1, b1, slope, w2, b2)
        ctx.mid_drop, ctx.out_drop = mid_drop, out_drop
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def bwd(ctx, dy):
        x, z, h = ctx.saved_tensors
        w1, b1, slope, w2, b2 = ctx.params
        dy = _c(dy)
        dy2 = _2d(dy)
        dz2 = _masked_dy(ctx.sink, dy2, ctx.out_drop) if ctx.out_drop[0] > 0 else dy2
        gw2, d_w2 = grad_target(w2)
        gb2, d_b2 = grad_target(b2)
        gw1, d_w1 = grad_target(w1)
        gb1, d_b1 = grad_target(b1)
        gs, d_s = grad_target(slope)
        dz1 = torch.empty_like(z)
        _accum_dw(dz2, _2d(h), gw2, gb2)
        kn.gemm(dz2, w2, _2d(dz1), e_drop=ctx.mid_drop, aux=_2d(z), e_actgrad=ACT_PRELU,
                slope=slope, dslope=gs)
        yield
        dx = None
        fold_res = ctx.res_is_x and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
        _accum_dw(_2d(dz1), _2d(x), gw1, gb1)
        if ctx.needs_input_grad[0]:
            dx = kn.rows_out(x, K=z.shape[-1])
            # residual input IS x: its gradient (dy) rides the epilogue of the dX GEMM
            kn.gemm(_2d(dz1), w1, _2d(dx), residual=dy2 if fold_res else None)
        return (dx, dy if ctx.has_res and ctx.needs_input_grad[1] and not fold_res else None,
                None if d_w1 else gw1, None if d_b1 else gb1, None if d_s else gs,
                None if d_w2 else gw2, None if d_b2 else gb2, None, None, None)


def ffn_call(x, residual, w1, b1, slope, w2, b2, mid_drop=NO_DROP, out_drop=NO_DROP):
    sink = _sink_for(out_drop)
    return Call(FFNBlock, (x, residual, w1, b1, slope, w2, b2, mid_drop, out_drop, sink), lambda y: _tag(y, sink))


def ffn(x, residual, w1, b1, slope, w2, b2, mid_drop=NO_DROP, out_drop=NO_DROP):
    return run(ffn_call(x, residual, w1, b1, slope, w2, b2, mid_drop, out_drop))


# ----------------------------------------------------------------------------- LayerNorm
class LayerNormBlock:
    """nn.LayerNorm over the last dim (transformer.py:536,539,646,649,754,793,796,400;
    model.py:430)."""
    N_OUT = 1

    @staticmethod
    def fwd(ctx, x, gamma, beta, eps, drop=NO_DROP, sink=None, in_relu=None, slot=None):
        x = _c(x)
        y, mean, rstd = kn.layernorm_fwd(x, gamma, beta, eps, drop, out=slot.t if slot is not None else None)
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.beta, ctx.drop = gamma, beta, drop
        ctx.sink = sink  # DropSink of the block that produced x (post-norm pattern), or None
        ctx.in_relu = in_relu if x.shape[-1] <= 256 else None  # ReluSink of the Linear + ReLU that produced x
        return y

    @staticmethod
    def bwd(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        gg, dg = grad_target(ctx.gamma)
        gb, db = grad_target(ctx.beta)
        sink = ctx.sink if ctx.needs_input_grad[0] else None
        ir = ctx.in_relu if (ctx.needs_input_grad[0] and sink is None) else None
        dx = kn.layernorm_bwd(_c(dy), x, ctx.gamma, mean, rstd, gg, gb, need_dx=ctx.needs_input_grad[0],
                              drop=ctx.drop, drop2=(sink.p, sink.seed) if sink is not None else None,
                              relu_in=ir is not None)
        if sink is not None:
            dx, sink.dz = dx
            sink.src = dx
        if ir is not None:
            ir.t = dx
        return (dx, None if dg else gg, None if db else gb, None, None, None, None, None)


class LayerNormForkBlock:
    """(y, y) = LayerNorm(x) for an output with TWO consumers (the decoder's norm1: the cross-attention block and the
    residual of its output projection, transformer.py:754-793): the second output is an alias of the first, and the
    backward kernel adds the two incoming gradients while it loads them (dyb) -- where the autograd engine would launch an
    element-wise add in front of this block's backward."""
    N_OUT = 2

    @staticmethod
    def fwd(ctx, x, gamma, beta, eps, sink):
        ctx.set_materialize_grads(False)
        x = _c(x)
        y, mean, rstd = kn.layernorm_fwd(x, gamma, beta, eps, NO_DROP)
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.beta, ctx.sink = gamma, beta, sink
        return y, y.view_as(y)

    @staticmethod
    def bwd(ctx, dy, dy2):
        x, mean, rstd = ctx.saved_tensors
        if dy is None:
            dy, dy2 = dy2, None
        gg, dg = grad_target(ctx.gamma)
        gb, db = grad_target(ctx.beta)
        need_dx = ctx.needs_input_grad[0]
        sink = ctx.sink if need_dx else None
        if dy2 is not None and not need_dx:  # (parameter gradients only: no kernel form with two gradients; tiny and rare)
            dy, dy2 = dy + dy2, None
        dx = kn.layernorm_bwd(_c(dy), x, ctx.gamma, mean, rstd, gg, gb, need_dx=need_dx,
                              drop2=(sink.p, sink.seed) if sink is not None else None,
                              dyb=_c(dy2) if dy2 is not None else None)
        if sink is not None:
            dx, sink.dz = dx
            sink.src = dx
        return (dx, None if dg else gg, None if db else gb, None, None)


class LayerNormPosBlock:
    """(y, y + add) = LayerNorm(x): the second output is the `with_pos_embed` query of the attention block
    that consumes y (transformer.py:512, 577, 640), written by the same kernel; the backward adds the two
    incoming gradients while loading them."""
    N_OUT = 2

    @staticmethod
    def fwd(ctx, x, gamma, beta, eps, add, sink):
        ctx.set_materialize_grads(False)
        x, add = _c(x), _c(add)
        y, mean, rstd, y2 = kn.layernorm_fwd(x, gamma, beta, eps, NO_DROP, add=add)
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.beta, ctx.sink = gamma, beta, sink
        return y, y2

    @staticmethod
    def bwd(ctx, dy, dy2):
        x, mean, rstd = ctx.saved_tensors
        if dy is None:
            dy, dyb = dy2, None
        else:
            dyb = dy2
        gg, dg = grad_target(ctx.gamma)
        gb, db = grad_target(ctx.beta)
        sink = ctx.sink if ctx.needs_input_grad[0] else None
        dx = kn.layernorm_bwd(_c(dy), x, ctx.gamma, mean, rstd, gg, gb, need_dx=ctx.needs_input_grad[0],
                              drop2=(sink.p, sink.seed) if sink is not None else None,
                              dyb=_c(dyb) if (dyb is not None and ctx.needs_input_grad[0]) else None)
        if not ctx.needs_input_grad[0] and dyb is not None:
            # parameter gradients of the second consumer's share (no dx requested: rare, tiny tensors)
            kn.layernorm_bwd(_c(dyb), x, ctx.gamma, mean, rstd, gg, gb, need_dx=False)
        if sink is not None:
            dx, sink.dz = dx
            sink.src = dx
        return (dx, None if dg else gg, None if db else gb, None, dy2 if ctx.needs_input_grad[4] else None, None)


def layer_norm_call(x, gamma, beta, eps=1e-5, drop=NO_DROP, add=None, fork=False, slot=None):
    """fork: -> (y, alias of y) for an output with two consumers (their gradients meet inside the backward kernel).
    slot (ops.Slot): y is written into that slot of a stacked output (ops.stacked) instead of a tensor of its own."""
    if slot is not None:
        assert not fork and add is None
    if fork and torch.is_grad_enabled() and add is None and drop[0] == 0.0:
        return Call(LayerNormForkBlock, (x, gamma, beta, eps, getattr(x, "_mesm_sink", None)))
    if fork:
        c = layer_norm_call(x, gamma, beta, eps, drop, add)
        post = c.post
        c.post = (lambda y: (lambda z: (z, z))(post(y) if post is not None else y))
        return c
    if add is not None:
        assert drop[0] == 0.0
        return Call(LayerNormPosBlock, (x, gamma, beta, eps, add, getattr(x, "_mesm_sink", None)))
    return Call(LayerNormBlock, (x, gamma, beta, eps, drop, getattr(x, "_mesm_sink", None),
                                 getattr(x, "_mesm_relu", None), slot))


def layer_norm(x, gamma, beta, eps=1e-5, drop=NO_DROP, add=None):
    """drop = (p, seed): the Dropout that follows the LayerNorm (LinearLayer) rides the same kernels.
    add: also return y + add (-> tuple)."""
    return run(layer_norm_call(x, gamma, beta, eps, drop, add))


# ----------------------------------------------------------------------------- LN -> FFN -> + x
class NormFFNBlock:
    """y = x + dropout_out( dropout_mid(prelu(LN(x) W1^T + b1)) W2^T + b2 ): the pre-norm feed-forward block of
    the T2V layers (transformer.py:536-538, 601-609) as ONE autograd block.  x reaches y on two routes; the
    LayerNorm backward kernel adds the residual route's gradient (dy) while it stores dx, and also emits dx
    under the dropout mask of the block that produced x (sink_in), so neither an element-wise add nor a mask
    kernel is launched."""
    N_OUT = 1

    @staticmethod
    def fwd(ctx, x, gamma, beta, eps, w1, b1, slope, w2, b2, mid_drop, out_drop, sink, sink_in):
        x = _c(x)
        h, mean, rstd = kn.layernorm_fwd(x, gamma, beta, eps)
        yield
        F_ = w1.shape[0]
        z = torch.empty(x.shape[:-1] + (F_,), device=x.device, dtype=torch.float32)
        a = torch.empty_like(z)
        # one pass writes z (pre-activation, kept for the PReLU gradient) and a = dropout(prelu(z))
        kn.gemm(_2d(h), w1, _2d(a), trans_b=True, bias=b1, e_act=ACT_PRELU, slope=slope, e_drop=mid_drop,
                pre_out=_2d(z))
        yield
        y = kn.rows_out(x, K=F_, fwd=True)
        kn.gemm(_2d(a), w2, _2d(y), trans_b=True, bias=b2, e_drop=out_drop, residual=_2d(x))
        ctx.save_for_backward(x, mean, rstd, h, z, a)
        ctx.params = (gamma, beta, w1, b1, slope, w2, b2)
        ctx.mid_drop, ctx.out_drop = mid_drop, out_drop
        ctx.sink, ctx.sink_in = sink, sink_in
        return y

    @staticmethod
    def bwd(ctx, dy):
        x, mean, rstd, h, z, a = ctx.saved_tensors
        gamma, beta, w1, b1, slope, w2, b2 = ctx.params
        dy = _c(dy)
        dy2 = _2d(dy)
        dz2 = _masked_dy(ctx.sink, dy2, ctx.out_drop) if ctx.out_drop[0] > 0 else dy2
        gw2, d_w2 = grad_target(w2)
        gb2, d_b2 = grad_target(b2)
        gw1, d_w1 = grad_target(w1)
        gb1, d_b1 = grad_target(b1)
        gs, d_s = grad_target(slope)
        gg, d_g = grad_target(gamma)
        gbt, d_bt = grad_target(beta)
        dz1 = torch.empty_like(z)
        _accum_dw(dz2, _2d(a), gw2, gb2)
        kn.gemm(dz2, w2, _2d(dz1), e_drop=ctx.mid_drop, aux=_2d(z), e_actgrad=ACT_PRELU, slope=slope, dslope=gs)
        yield
        dh = kn.rows_out(h, K=z.shape[-1])
        _accum_dw(_2d(dz1), _2d(h), gw1, gb1)
        kn.gemm(_2d(dz1), w1, _2d(dh))
        yield
        dx = None
        if ctx.needs_input_grad[0]:
            sink = ctx.sink_in
            dx = kn.layernorm_bwd(dh, x, gamma, mean, rstd, gg, gbt, addend=dy,
                                  drop2=(sink.p, sink.seed) if sink is not None else None)
            if sink is not None:
                dx, sink.dz = dx
                sink.src = dx
        else:
            kn.layernorm_bwd(dh, x, gamma, mean, rstd, gg, gbt, need_dx=False)
        return (dx, None if d_g else gg, None if d_bt else gbt, None, None if d_w1 else gw1, None if d_b1 else gb1,
                None if d_s else gs, None if d_w2 else gw2, None if d_b2 else gb2, None, None, None, None)


def norm_ffn_call(x, gamma, beta, w1, b1, slope, w2, b2, eps=1e-5, mid_drop=NO_DROP, out_drop=NO_DROP):
    sink = _sink_for(out_drop)
    return Call(NormFFNBlock, (x, gamma, beta, eps, w1, b1, slope, w2, b2, mid_drop, out_drop, sink,
                               getattr(x, "_mesm_sink", None)), lambda y: _tag(y, sink))


def norm_ffn(x, gamma, beta, w1, b1, slope, w2, b2, eps=1e-5, mid_drop=NO_DROP, out_drop=NO_DROP):
    return run(norm_ffn_call(x, gamma, beta, w1, b1, slope, w2, b2, eps, mid_drop, out_drop))


# ----------------------------------------------------------------------------- attention core
class AttentionFn(Function):
    """softmax(scale * q k^T, masks) -> dropout -> @ v for packed heads, no projections:
    the core of attention.py:329-386 (decoder self / cross attention, dk may differ from dv).
    q2 / k2 (optional): split heads, head h sees [q_h || q2_h] and [k_h || k2_h] -- the cross attention's
    per-head [content || position] concatenation (transformer.py:778-784) read in place."""

    @staticmethod
    def forward(ctx, q, k, v, H, kpad, qpad, scale, drop, q2, k2):
        q, k, v = _c(q), _c(k), _c(v)
        if q2 is not None:
            q2, k2 = _c(q2), _c(k2)
        o, lse = kn.attn_fwd(q, k, v, H, kpad=kpad, qpad=qpad, scale=scale, drop=drop, q2=q2, k2=k2)
        ctx.save_for_backward(q, k, v, o, lse, q2, k2)
        ctx.cfg = (H, kpad, qpad, scale, drop)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, q2, k2 = ctx.saved_tensors
        H, kpad, qpad, scale, drop = ctx.cfg
        res = kn.attn_bwd(_c(do), q, k, v, o, lse, H, kpad=kpad, qpad=qpad, scale=scale, drop=drop, q2=q2, k2=k2)
        dq, dk, dv = res[:3]
        dq2, dk2 = (res[3], res[4]) if q2 is not None else (None, None)
        return dq, dk, dv, None, None, None, None, None, dq2, dk2


def attention(q, k, v, H, kpad=None, qpad=None, scale=None, drop=NO_DROP, q2=None, k2=None):
    if scale is None:
        scale = ((q.shape[-1] * (2 if q2 is not None else 1)) // H) ** -0.5
    return AttentionFn.apply(q, k, v, H, kpad, qpad, scale, drop, q2, k2)


# ----------------------------------------------------------------------------- decoder attention blocks
class DecSelfAttnBlock:
    """Decoder self-attention up to the attention output (transformer.py:737-750):

        q = sa_qcontent(tgt) + sa_qpos(query_pos);  k = sa_kcontent(tgt) + sa_kpos(query_pos);  v = sa_v(tgt)
        a = softmax(q k^T / sqrt(dh)) v

    wt / bt = the (3d, d) pack [sa_qcontent ; sa_kcontent ; sa_v], wp / bp = the (2d, d) pack [sa_qpos ; sa_kpos]
    (gradbuf.Pack: the members sit back to back in the flat buffers).  Forward: the position term and the value
    projection first (disjoint columns of one (rows, 3d) buffer), then the content projection accumulated onto the
    position term, then the attention core.  Backward: the attention gradients land in one (rows, 3d) buffer; d tgt is
    ONE GEMM with K = 3d, d query_pos one with K = 2d, two weight-gradient GEMMs -- a single grouped launch, and no
    gradient fan-in left for autograd to add."""
    N_OUT = 1

    @staticmethod
    def fwd(ctx, tgt, qpos, wt, bt, wp, bp, H, drop):
        tgt, qpos = _c(tgt), _c(qpos)
        n, nq, d = tgt.shape
        qkv = torch.empty(n, nq, 3 * d, device=tgt.device, dtype=torch.float32)
        q2 = _2d(qkv)
        kn.gemm(_2d(qpos), wp, q2[:, :2 * d], trans_b=True, bias=bp)
        kn.gemm(_2d(tgt), wt[2 * d:], q2[:, 2 * d:], trans_b=True, bias=bt[2 * d:])
        yield
        kn.gemm(_2d(tgt), wt[:2 * d], q2[:, :2 * d], trans_b=True, bias=bt[:2 * d], accumulate=1)
        yield
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        o, lse = kn.attn_fwd(q, k, v, H, drop=drop)
        ctx.save_for_backward(tgt, qpos, qkv, o, lse)
        ctx.params = (wt, bt, wp, bp)
        ctx.cfg = (H, drop)
        return o

    @staticmethod
    def bwd(ctx, do):
        tgt, qpos, qkv, o, lse = ctx.saved_tensors
        wt, bt, wp, bp = ctx.params
        H, drop = ctx.cfg
        n, nq, d = tgt.shape
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        dqkv = (kn.zeros((n, nq, 3 * d), tgt.device) if kn.attn_bwd_adds_dq(n, H, nq, nq, d // H, d // H)
                else torch.empty(n, nq, 3 * d, device=tgt.device, dtype=torch.float32))
        kn.attn_bwd_into(_c(do), q, k, v, o, lse, H, dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:], drop=drop)
        yield
        g2 = _2d(dqkv)
        gwt, _ = grad_target(wt)
        gbt, _ = grad_target(bt)
        gwp, _ = grad_target(wp)
        gbp, _ = grad_target(bp)
        dtgt = dqp = None
        _accum_dw(g2, _2d(tgt), gwt, gbt)
        _accum_dw(g2[:, :2 * d], _2d(qpos), gwp, gbp)
        if ctx.needs_input_grad[0]:
            dtgt = torch.empty_like(tgt)
            kn.gemm(g2, wt, _2d(dtgt))
        if ctx.needs_input_grad[1]:
            dqp = torch.empty_like(qpos)
            kn.gemm(g2[:, :2 * d], wp, _2d(dqp))
        return dtgt, dqp, None, None, None, None, None, None


def dec_self_attn_call(tgt, qpos, wt, bt, wp, bp, H, drop=NO_DROP):
    return Call(DecSelfAttnBlock, (tgt, qpos, wt, bt, wp, bp, H, drop))


def dec_self_attn(tgt, qpos, wt, bt, wp, bp, H, drop=NO_DROP):
    return run(dec_self_attn_call(tgt, qpos, wt, bt, wp, bp, H, drop))


class DecCrossAttnBlock:
    """Decoder cross-attention up to the attention output (transformer.py:757-789), conditional-DETR heads:

        q_h = [ ca_qcontent(tgt) (+ ca_qpos(query_pos) on layer 0) ]_h || [ ca_qpos_sine(qsine) ]_h
        k_h = [ ca_kcontent(memory) (+ ca_kpos(pos) on layer 0) ]_h   || [ ca_kpos(pos) ]_h
        a   = softmax(q k^T / sqrt(2 dh), key padding) ca_v(memory)

    wkv / bkv = the (2d, d) pack [ca_kcontent ; ca_v].  The memory-side projections land in one (rows, 3d) buffer
    [kcontent | v | kpos] that the attention kernel reads as split heads (k, k2, + k_add on layer 0) in place;
    backward: d memory is ONE GEMM with K = 2d, every weight gradient a split-K GEMM of the same grouped launch;
    layer 0's kpos weights collect both routes by accumulation."""
    N_OUT = 1

    @staticmethod
    def fwd(ctx, tgt, qs, qpp, memory, pos, mem_pad, wqc, bqc, wkv, bkv, wkp, bkp, first, H, drop, mem_share=None):
        ctx.set_materialize_grads(False)
        ctx.mem_share = mem_share  # GradShare of the decoder's layers for d memory (every layer reads the same memory)
        tgt, qs, memory, pos = _c(tgt), _c(qs), _c(memory), _c(pos)
        n, nq, d = tgt.shape
        lm = memory.shape[1]
        kvp = torch.empty(n, lm, 3 * d, device=tgt.device, dtype=torch.float32)
        k3 = _2d(kvp)
        qc = torch.empty_like(tgt)
        kn.gemm(_2d(memory), wkv, k3[:, :2 * d], trans_b=True, bias=bkv)
        kn.gemm(_2d(pos), wkp, k3[:, 2 * d:], trans_b=True, bias=bkp)
        kn.gemm(_2d(tgt), wqc, _2d(qc), trans_b=True, bias=bqc,
                residual=_2d(_c(qpp)) if (first and qpp is not None) else None)
        yield
        kc, cv, kp = kvp[..., :d], kvp[..., d:2 * d], kvp[..., 2 * d:]
        o, lse = kn.attn_fwd(qc, kc, cv, H, kpad=mem_pad, drop=drop, q2=qs, k2=kp, k_add=kp if first else None)
        ctx.save_for_backward(tgt, qs, memory, pos, qc, kvp, o, lse)
        ctx.params = (wqc, bqc, wkv, bkv, wkp, bkp)
        ctx.cfg = (first, H, drop, mem_pad, qpp is not None)
        return o

    @staticmethod
    def bwd(ctx, do):
        tgt, qs, memory, pos, qc, kvp, o, lse = ctx.saved_tensors
        wqc, bqc, wkv, bkv, wkp, bkp = ctx.params
        first, H, drop, mem_pad, has_qpp = ctx.cfg
        n, nq, d = tgt.shape
        lm = memory.shape[1]
        dev = tgt.device
        kc, cv, kp = kvp[..., :d], kvp[..., d:2 * d], kvp[..., 2 * d:]
        # several 64-key tiles add into dq atomically: both query halves start from zero (one fill)
        dq2x = (kn.zeros((2, n, nq, d), dev) if kn.attn_bwd_adds_dq(n, H, nq, lm, 2 * d // H, d // H, split=True)
                else torch.empty(2, n, nq, d, device=dev, dtype=torch.float32))
        dqc, dqs = dq2x[0], dq2x[1]
        dkvp = torch.empty(n, lm, 3 * d, device=dev, dtype=torch.float32)
        kn.attn_bwd_into(_c(do), qc, kc, cv, o, lse, H, dqc, dkvp[..., :d], dkvp[..., d:2 * d], kpad=mem_pad,
                         drop=drop, q2=qs, k2=kp, dq2=dqs, dk2=dkvp[..., 2 * d:], k_add=kp if first else None)
        yield
        g3 = _2d(dkvp)
        gwqc, _ = grad_target(wqc)
        gbqc, _ = grad_target(bqc)
        gwkv, _ = grad_target(wkv)
        gbkv, _ = grad_target(bkv)
        gwkp, _ = grad_target(wkp)
        gbkp, _ = grad_target(bkp)
        dtgt = dmem = None
        _accum_dw(_2d(dqc), _2d(tgt), gwqc, gbqc)
        _accum_dw(g3[:, :2 * d], _2d(memory), gwkv, gbkv)
        _accum_dw(g3[:, 2 * d:], _2d(pos), gwkp, gbkp)
        if first:  # kpos also fed the content half of the keys
            _accum_dw(g3[:, :d], _2d(pos), gwkp, gbkp)
        if ctx.needs_input_grad[0]:
            dtgt = torch.empty_like(tgt)
            kn.gemm(_2d(dqc), wqc, _2d(dtgt))
        if ctx.needs_input_grad[3]:
            share = ctx.mem_share
            if share is not None:
                if share.seen == 0:
                    share.buf = torch.empty_like(memory)
                kn.gemm(g3[:, :2 * d], wkv, _2d(share.buf), accumulate=0 if share.seen == 0 else 1)
                share.seen += 1
                if share.seen == share.n:  # every layer has added its share
                    dmem, share.buf, share.seen = share.buf, None, 0
            else:
                dmem = torch.empty_like(memory)
                kn.gemm(g3[:, :2 * d], wkv, _2d(dmem))
        return (dtgt, dqs if ctx.needs_input_grad[1] else None,
                dqc if (first and has_qpp and ctx.needs_input_grad[2]) else None, dmem,
                None, None, None, None, None, None, None, None, None, None, None, None)


def dec_cross_attn_call(tgt, qs, qpp, memory, pos, mem_pad, wqc, bqc, wkv, bkv, wkp, bkp, first, H, drop=NO_DROP,
                        mem_share=None):
    return Call(DecCrossAttnBlock, (tgt, qs, qpp, memory, pos, mem_pad, wqc, bqc, wkv, bkv, wkp, bkp, first, H, drop,
                                    mem_share))


def dec_cross_attn(tgt, qs, qpp, memory, pos, mem_pad, wqc, bqc, wkv, bkv, wkp, bkp, first, H, drop=NO_DROP):
    return run(dec_cross_attn_call(tgt, qs, qpp, memory, pos, mem_pad, wqc, bqc, wkv, bkv, wkp, bkp, first, H, drop))


# ----------------------------------------------------------------------------- packed MHA
class GradShare:
    """One gradient buffer for a tensor that n blocks of a chain read (the key / value source of every layer of a
    T2V stack: transformer.py:216-242 hands the same `src_txt` to every layer): each block's dX GEMM accumulates into
    the buffer through its read-modify-write epilogue, the block whose backward runs LAST (the first layer) returns the
    total and the others return None -- autograd has no fan-in left to add with element-wise launches.  Only for
    blocks that all take part in the backward (a chain: layer l + 1 consumes layer l)."""
    __slots__ = ("n", "buf", "seen")

    def __init__(self, n):
        self.n, self.buf, self.seen = n, None, 0


class MHABlock:
    """A whole nn.MultiheadAttention call plus its residual:

        out = residual + dropout_out( out_proj( attn( xqp Wq, (xk+pk) Wk, xk Wv ) ) )

    (transformer.py:523-534 cross-attention of the T2V layers, :642-645 encoder self attention; in_proj_weight
    rows are q,k,v -- torch/nn/functional.py multi_head_attention_forward).  xqp = xq + pos is handed in
    ready-made (the LayerNorm / assembly kernel that produced xq wrote it as a second output; None = no
    positional term, xqp is xq) and is an autograd input of its own, so the two routes of the query gradient
    are returned separately and summed by the consumer's kernel, not by an element-wise launch.
    `self_attn=True`: Q and K come from xqp, V from xq: Q and K projections run as ONE GEMM over
    in_proj_weight[0:2d].  Otherwise, when the key has no positional term (use_txt_pos=False in every shipped
    config; SegSenRecon passes None), K and V run as ONE GEMM over in_proj_weight[d:3d].
    Three launch phases forward (projections | attention core | output projection) and three backward.
    """
    N_OUT = 1

    @staticmethod
    def fwd(ctx, xq, xqp, xk, pk, residual, w_in, b_in, w_out, b_out, H, kpad, qpad, attn_drop,
            out_drop, self_attn, group=0, sink=None, kv_share=None, join_qp=False, xkp=None):
        # xkp: xk + its position embedding formed WITHOUT autograd (ops.gather_add): the key projection reads it as a plain
        # operand (a second-operand product runs outside the grouped launch) and its gradient belongs to xk -- the key side's
        # backward is then the position-free one: one dX product, shared across the layers of a stack
        ctx.set_materialize_grads(False)
        ctx.sink = sink
        ctx.kv_share = kv_share
        # join_qp: xqp = xq + a constant formed WITHOUT autograd (the very first query of a stack): its gradient
        # belongs to xq and joins the residual route in the dX GEMM's epilogue instead of an element-wise add
        ctx.join_qp = join_qp
        xq = _c(xq)
        has_p = xqp is not None
        xqp = _c(xqp) if has_p else xq
        d = xq.shape[-1]
        N, Lq = xq.shape[0], xq.shape[1]
        dev = xq.device
        if self_attn:
            Lk = Lq
            qkv = torch.empty(N, Lq, 3 * d, device=dev, dtype=torch.float32)
            q2 = _2d(qkv)
            if has_p:
                kn.gemm(_2d(xqp), w_in[:2 * d], q2[:, :2 * d], trans_b=True, bias=b_in[:2 * d])
                kn.gemm(_2d(xq), w_in[2 * d:], q2[:, 2 * d:], trans_b=True, bias=b_in[2 * d:])
            else:
                kn.gemm(_2d(xq), w_in, q2, trans_b=True, bias=b_in)
            q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
            xk = pk = None
        else:
            q = torch.empty(N, Lq, d, device=dev, dtype=torch.float32)
            kn.gemm(_2d(xqp), w_in[:d], _2d(q), trans_b=True, bias=b_in[:d])
            xk = _c(xk)
            if xkp is not None:
                pk = _c(xkp)  # (saved in pk's slot; ctx.has_xkp tells the backward which it is)
            else:
                pk = _c(pk) if pk is not None else None
            Lk = xk.shape[1]
            kv = torch.empty(N, Lk, 2 * d, device=dev, dtype=torch.float32)
            kv2 = _2d(kv)
            if xkp is not None:
                kn.gemm(_2d(pk), w_in[d:2 * d], kv2[:, :d], trans_b=True, bias=b_in[d:2 * d])
                kn.gemm(_2d(xk), w_in[2 * d:], kv2[:, d:], trans_b=True, bias=b_in[2 * d:])
            elif pk is None:
                kn.gemm(_2d(xk), w_in[d:], kv2, trans_b=True, bias=b_in[d:])
            else:
                kn.gemm(_2d(xk), w_in[d:2 * d], kv2[:, :d], trans_b=True, A2=_2d(pk), bias=b_in[d:2 * d])
                kn.gemm(_2d(xk), w_in[2 * d:], kv2[:, d:], trans_b=True, bias=b_in[2 * d:])
            k, v = kv[..., :d], kv[..., d:]
        yield
        o, lse = kn.attn_fwd(q, k, v, H, kpad=kpad, qpad=qpad, drop=attn_drop, group=group)
        yield
        out = torch.empty(N, Lq, d, device=dev, dtype=torch.float32)
        kn.gemm(_2d(o), w_out, _2d(out), trans_b=True, bias=b_out, e_drop=out_drop,
                residual=_2d(_c(residual)) if residual is not None else None)
        ctx.save_for_backward(xq, xqp if has_p else None, xk, pk, q, k, v, o, lse)
        ctx.has_xkp = xkp is not None and not self_attn
        ctx.params = (w_in, b_in, w_out, b_out)
        ctx.res_is_xq = residual is not None and residual.data_ptr() == xq.data_ptr() and residual.shape == xq.shape
        ctx.cfg = (H, kpad, qpad, attn_drop, out_drop, self_attn, residual is not None, group, has_p)
        return out

    @staticmethod
    def bwd(ctx, dy):
        xq, xqp, xk, pk, q, k, v, o, lse = ctx.saved_tensors
        xkp = None
        if ctx.has_xkp:
            xkp, pk = pk, None
        w_in, b_in, w_out, b_out = ctx.params
        H, kpad, qpad, attn_drop, out_drop, self_attn, has_res, group, has_p = ctx.cfg
        if xqp is None:
            xqp = xq
        d = xq.shape[-1]
        N, Lq = xq.shape[0], xq.shape[1]
        dev = xq.device
        dy = _c(dy)
        dy2 = _2d(dy)
        dz = _masked_dy(ctx.sink, dy2, out_drop) if out_drop[0] > 0 else dy2
        gwo, d_wo = grad_target(w_out)
        gbo, d_bo = grad_target(b_out)
        gwi, d_wi = grad_target(w_in)
        gbi, d_bi = grad_target(b_in)
        do = torch.empty_like(o)
        _accum_dw(dz, _2d(o), gwo, gbo)
        kn.gemm(dz, w_out, _2d(do))
        yield
        need_q = ctx.needs_input_grad[0]
        need_qp = has_p and ctx.needs_input_grad[1]
        need_k, need_pk = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        fold = ctx.res_is_xq and need_q and ctx.needs_input_grad[4]  # the residual input IS xq: dy joins d xq
        dxq = dxqp = dxk = dpk = None
        if self_attn:
            # dq is added atomically only when several 64-key tiles contribute
            dqkv = (kn.zeros((N, Lq, 3 * d), dev) if kn.attn_bwd_adds_dq(N, H, Lq, Lq, d // H, d // H)
                    else torch.empty(N, Lq, 3 * d, device=dev, dtype=torch.float32))
            kn.attn_bwd_into(do, q, k, v, o, lse, H, dqkv[..., :d], dqkv[..., d:2 * d],
                             dqkv[..., 2 * d:], kpad=kpad, qpad=qpad, drop=attn_drop, group=group)
            yield
            g2 = _2d(dqkv)
            if has_p:
                _accum_dw(g2[:, :2 * d], _2d(xqp), gwi[:2 * d], gbi[:2 * d])
                _accum_dw(g2[:, 2 * d:], _2d(xq), gwi[2 * d:], gbi[2 * d:])
                if need_qp:
                    dxqp = torch.empty_like(xq)
                    kn.gemm(g2[:, :2 * d], w_in[:2 * d], _2d(dxqp))
                if need_q:
                    dxq = torch.empty_like(xq)
                    kn.gemm(g2[:, 2 * d:], w_in[2 * d:], _2d(dxq), residual=dy2 if fold else None)
            else:
                _accum_dw(g2, _2d(xq), gwi, gbi)
                if need_q:
                    dxq = torch.empty_like(xq)
                    kn.gemm(g2, w_in, _2d(dxq), residual=dy2 if fold else None)
        else:
            Lk = k.shape[1]
            dq = (kn.zeros((N, Lq, d), dev) if kn.attn_bwd_adds_dq(N, H, Lq, Lk, d // H, d // H)
                  else torch.empty(N, Lq, d, device=dev, dtype=torch.float32))
            dkv = torch.empty(N, Lk, 2 * d, device=dev, dtype=torch.float32)
            kn.attn_bwd_into(do, q, k, v, o, lse, H, dq, dkv[..., :d], dkv[..., d:], kpad=kpad,
                             qpad=qpad, drop=attn_drop, group=group)
            yield
            g2 = _2d(dkv)
            # dWq, dWkv, dX(query side), dX(key side): all independent
            _accum_dw(_2d(dq), _2d(xqp), gwi[:d], gbi[:d])
            if xkp is not None:
                _accum_dw(g2[:, :d], _2d(xkp), gwi[d:2 * d], gbi[d:2 * d])
                _accum_dw(g2[:, d:], _2d(xk), gwi[2 * d:], gbi[2 * d:])
            elif pk is None:
                _accum_dw(g2, _2d(xk), gwi[d:], gbi[d:])
            else:
                _accum_dw(g2[:, :d], _2d(xk), gwi[d:2 * d], gbi[d:2 * d], x2=_2d(pk))
                _accum_dw(g2[:, d:], _2d(xk), gwi[2 * d:], gbi[2 * d:])
            if has_p and ctx.join_qp and fold and not need_qp:
                dxq = torch.empty_like(xq)
                kn.gemm(_2d(dq), w_in[:d], _2d(dxq), residual=dy2)
            elif has_p:
                if need_qp:
                    dxqp = torch.empty_like(xq)
                    kn.gemm(_2d(dq), w_in[:d], _2d(dxqp))
                if fold:
                    dxq = dy  # the only route from xq itself is the residual
            elif need_q:
                dxq = torch.empty_like(xq)
                kn.gemm(_2d(dq), w_in[:d], _2d(dxq), residual=dy2 if fold else None)
            if need_k or need_pk:
                share = ctx.kv_share
                if pk is None and share is not None:
                    if share.seen == 0:
                        share.buf = torch.empty_like(xk)
                    kn.gemm(g2, w_in[d:], _2d(share.buf), accumulate=0 if share.seen == 0 else 1)
                    share.seen += 1
                    if share.seen == share.n:  # every reader has added its share
                        dxk, share.buf, share.seen = share.buf, None, 0
                elif pk is None:
                    dxk = torch.empty_like(xk)
                    kn.gemm(g2, w_in[d:], _2d(dxk))
                else:
                    dk_in = torch.empty_like(xk)
                    kn.gemm(g2[:, :d], w_in[d:2 * d], _2d(dk_in))
            if (need_k or need_pk) and pk is not None:
                yield
                dpk = dk_in if need_pk else None
                if need_k:
                    dxk = torch.empty_like(xk)
                    kn.gemm(g2[:, d:], w_in[2 * d:], _2d(dxk), residual=_2d(dk_in))
        return (dxq, dxqp, dxk, dpk, dy if has_res and ctx.needs_input_grad[4] and not fold else None,
                None if d_wi else gwi, None if d_bi else gbi, None if d_wo else gwo,
                None if d_bo else gbo, None, None, None, None, None, None, None, None, None, None, None)


def mha_call(xq, xqp, xk, pk, residual, w_in, b_in, w_out, b_out, H, kpad=None, qpad=None,
             attn_drop=NO_DROP, out_drop=NO_DROP, self_attn=False, group=0, kv_share=None, join_qp=False, xkp=None):
    sink = _sink_for(out_drop)
    return Call(MHABlock, (xq, xqp, xk, pk, residual, w_in, b_in, w_out, b_out, H, kpad, qpad, attn_drop,
                           out_drop, self_attn, group, sink, kv_share, join_qp, xkp), lambda y: _tag(y, sink))


def mha(xq, xqp, xk, pk, residual, w_in, b_in, w_out, b_out, H, kpad=None, qpad=None,
        attn_drop=NO_DROP, out_drop=NO_DROP, self_attn=False, group=0):
    """xqp: xq + its position embedding, ready-made (None: no positional term).
    group: rows per independent batch when several batches are stacked (mask quirk Q1)."""
    return run(mha_call(xq, xqp, xk, pk, residual, w_in, b_in, w_out, b_out, H, kpad, qpad, attn_drop, out_drop,
                        self_attn, group))


# ----------------------------------------------------------------------------- sine embeddings
class QuerySineFn(Function):
    """gen_sineembed_for_position, transformer.py:43-59 (gradient flows to the reference
    points of decoder layer 0)."""

    @staticmethod
    def forward(ctx, ref, D):
        ref = _c(ref)
        ctx.save_for_backward(ref)
        return kn.query_sine_fwd(ref, D)

    @staticmethod
    def backward(ctx, dout):
        (ref,) = ctx.saved_tensors
        return kn.query_sine_bwd(ref, _c(dout)), None


def query_sine(ref, D):
    return QuerySineFn.apply(ref, D)


# ----------------------------------------------------------------------------- losses
class NLLSmoothFn(Function):
    """Criterion.cal_nll_loss (criterion.py:291-306): per-row label-smoothed NLL, 0 on masked rows."""

    @staticmethod
    def forward(ctx, logit, label, mask, eps):
        ctx.set_materialize_grads(False)  # (no zero fill for the gradient of `correct`)
        logit = _c(logit)
        C = logit.shape[-1]
        l2 = logit.view(-1, C)
        row_loss, row_lse, correct = kn.nll_smooth_fwd(l2, _c(label).view(-1), _c(mask).view(-1), eps)
        ctx.save_for_backward(l2, _c(label).view(-1), row_lse, _c(mask).view(-1))
        ctx.eps = eps
        ctx.shape = logit.shape
        correct = correct.view(logit.shape[:-1])
        ctx.mark_non_differentiable(correct)
        return row_loss.view(logit.shape[:-1]), correct

    @staticmethod
    def backward(ctx, g_loss, _g_correct):
        l2, label, row_lse, mask = ctx.saved_tensors
        rg = (_c(g_loss).view(-1) * mask.to(torch.float32)).contiguous()
        dl = kn.nll_smooth_bwd(l2, label, row_lse, rg, ctx.eps)
        return dl.view(ctx.shape), None, None, None


def nll_smooth(logit, label, mask, eps=0.1):
    return NLLSmoothFn.apply(logit, label, mask, eps)


class SaliencyLossFn(Function):
    """Criterion.loss_saliency (criterion.py:139-221) as one fused reduction."""

    @staticmethod
    def forward(ctx, s_pos, s_neg, label64, vmask, pos_idx, neg_idx, rank_coef, margin):
        s_pos, s_neg = _c(s_pos), _c(s_neg)
        out = kn.saliency_loss_fwd(s_pos, s_neg, label64, vmask, pos_idx, neg_idx, rank_coef, margin)
        ctx.save_for_backward(s_pos, s_neg, label64, vmask, pos_idx, neg_idx)
        ctx.cfg = (rank_coef, margin)
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        s_pos, s_neg, label64, vmask, pos_idx, neg_idx = ctx.saved_tensors
        gs = g.reshape(1).to(torch.float32).contiguous()
        ds_pos, ds_neg = kn.saliency_loss_bwd(s_pos, s_neg, label64, vmask, pos_idx, neg_idx,
                                              ctx.cfg[0], ctx.cfg[1], gs)
        return ds_pos, ds_neg, None, None, None, None, None, None


def saliency_loss(s_pos, s_neg, label64, vmask, pos_idx, neg_idx, rank_coef, margin):
    return SaliencyLossFn.apply(s_pos, s_neg, label64, vmask, pos_idx, neg_idx, rank_coef, margin)


class RowDotFn(Function):
    """saliency score (model.py:301-302): s[n,l] = <a[n,l,:], b[n,:]> * scale."""

    @staticmethod
    def forward(ctx, a, b, scale):
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        ctx.scale = scale
        return kn.rowdot_fwd(a, b, scale)

    @staticmethod
    def backward(ctx, ds):
        a, b = ctx.saved_tensors
        da, db = kn.rowdot_bwd(a, b, _c(ds), ctx.scale)
        return da, db, None


def rowdot(a, b, scale):
    return RowDotFn.apply(a, b, scale)


class GatherRowsFn(Function):
    """y = x2d[idx] for a 2-D x2d and an integer index tensor of any shape (rows may repeat).
    Backward is a scatter-add with float atomics (index_add_), not ATen's sort-based
    index_put(accumulate=True) path (60 us per call at these sizes)."""

    @staticmethod
    def forward(ctx, x2d, idx):
        flat = idx.reshape(-1)
        ctx.save_for_backward(flat)
        ctx.rows = x2d.shape[0]
        return x2d.index_select(0, flat).view(*idx.shape, x2d.shape[1])

    @staticmethod
    def backward(ctx, dy):
        (flat,) = ctx.saved_tensors
        dx = torch.zeros(ctx.rows, dy.shape[-1], device=dy.device, dtype=dy.dtype)
        dx.index_add_(0, flat, dy.reshape(-1, dy.shape[-1]))
        return dx, None


def gather_rows(x2d, idx):
    return GatherRowsFn.apply(x2d, idx)


# ----------------------------------------------------------------------------- assembly blocks (csrc/glue.hip)
_STACK_GRAD_ALL = os.environ.get("MESM_STACK_GRAD_ALL") == "1"  # A/B: the round-2 behaviour (every float output differentiable)


class StackRowsFn(Function):
    """outs[t] = [x_t ; x_t[idx]] or [x_t ; x_t]: the negative pass stacked behind the positive one
    (model.py:260-299), every tensor of the stage in one launch; backward = one gather-sum per float tensor."""

    @staticmethod
    def forward(ctx, idx, gather, *xs):
        ctx.set_materialize_grads(False)
        outs = kn.stack_rows(list(xs), gather, idx)
        ctx.idx, ctx.gather, ctx.N = idx, gather, xs[0].shape[0]
        # a stacked copy of a tensor that carries no gradient (position embeddings, masks) carries none either: left
        # differentiable, its consumers return gradients for it that the engine sums with element-wise launches
        # (three x 15 MB for the stacked video positions) before this function's backward drops them.  ONE call:
        # mark_non_differentiable replaces what an earlier call marked.
        ctx.mark_non_differentiable(*[o for i, (o, x) in enumerate(zip(outs, xs))
                                      if not x.is_floating_point() or not (ctx.needs_input_grad[2 + i] or _STACK_GRAD_ALL)])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        res = []
        for i, g in enumerate(gs):
            if g is None or not ctx.needs_input_grad[2 + i]:
                res.append(None)
            else:
                res.append(kn.unstack_rows(_c(g), ctx.idx if ctx.gather[i] else None, ctx.N))
        return (None, None) + tuple(res)


def stack_rows(xs, gather, idx):
    return StackRowsFn.apply(idx, tuple(gather), *xs)


class PrependFn(Function):
    """xo = [tok ; x] along the sequence (+ po = [ptok ; pos], xp = xo + po, pado = [first ; pad]).  Gradients of
    parameter tokens (global token / its position embedding) are added by the kernel into the flat gradient
    buffer; a per-row token (reconstructed sentence) gets its gradient as a tensor."""

    @staticmethod
    def forward(ctx, tok, x, ptok, pos, pad, first_pad):
        ctx.set_materialize_grads(False)
        x = _c(x)
        tokc = _c(tok)
        xo, po, xp, pado = kn.prepend_fwd(tokc, x, _c(ptok) if ptok is not None else None,
                                          _c(pos) if pos is not None else None, pad, first_pad)
        ctx.per_row = tokc.dim() == 2 and tokc.shape[0] == x.shape[0] and tokc.numel() == x.shape[0] * x.shape[2]
        ctx.tok, ctx.ptok, ctx.shape = tok, ptok, x.shape
        outs = [xo]
        if po is not None:
            outs += [po, xp]
        if pado is not None:
            ctx.mark_non_differentiable(pado)
            outs.append(pado)
        ctx.n_out = len(outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        B, L, D = ctx.shape
        dxo = gs[0]
        dpo = dxp = None
        if ctx.ptok is not None:
            dpo, dxp = gs[1], gs[2]
        dxo = _c(dxo) if dxo is not None else None
        dxp = _c(dxp) if dxp is not None else None
        dpo = _c(dpo) if dpo is not None else None
        dev = (dxo if dxo is not None else dxp).device
        dx = torch.empty(B, L, D, device=dev, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        dtok_ret = dptok_ret = None
        if ctx.per_row:
            dtok = torch.empty(B, D, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
            dtok_ret = dtok
        else:
            dtok, direct = grad_target(ctx.tok) if ctx.needs_input_grad[0] else (None, True)
            dtok_ret = None if direct else dtok
        dptok = None
        if ctx.ptok is not None and ctx.needs_input_grad[2]:
            dptok, direct = grad_target(ctx.ptok)
            dptok_ret = None if direct else dptok
        kn.prepend_bwd(dxo, dxp, dpo, dx, dtok, dptok, B, L, D, ctx.per_row)
        flush_ready()
        return dtok_ret, dx, dptok_ret, None, None, None


def prepend(tok, x, ptok=None, pos=None, pad=None, first_pad=True):
    """-> xo [, po, xp] [, pado]"""
    return PrependFn.apply(tok, x, ptok, pos, pad, first_pad)


class SplitTokenFn(Function):
    """(mem[:, 0], mem[:, 1:], mem[:Bd, 1:]) as contiguous tensors, one launch; the backward assembles d mem from
    whichever of the three gradients exist (transformer.py:196-198)."""

    @staticmethod
    def forward(ctx, mem, Bd):
        ctx.set_materialize_grads(False)
        mem = _c(mem)
        g, loc, dec = kn.split_token_fwd(mem, Bd)
        ctx.shape, ctx.Bd = mem.shape, Bd
        return (g, loc, dec) if Bd else (g, loc)

    @staticmethod
    def backward(ctx, dg, dloc, ddec=None):
        B, S, D = ctx.shape
        t = next(x for x in (dg, dloc, ddec) if x is not None)
        dmem = kn.split_token_bwd(_c(dg) if dg is not None else None, _c(dloc) if dloc is not None else None,
                                  _c(ddec) if ddec is not None else None, B, S - 1, D, ctx.Bd, t.device)
        return dmem, None


def split_token(mem, Bd=0):
    return SplitTokenFn.apply(mem, Bd)


class TokenMixFn(Function):
    """y[r] = m2[r] ? tok2 : (m1[r] ? tok1 : x[r])  (model.py:361-394, 493-501)."""

    @staticmethod
    def forward(ctx, x, m1, tok1, m2, tok2):
        ctx.set_materialize_grads(False)
        x = _c(x)
        m1 = _c(m1)
        m2 = _c(m2) if m2 is not None else None
        y = kn.token_mix_fwd(x, m1, _c(tok1).reshape(-1), m2, _c(tok2).reshape(-1) if tok2 is not None else None)
        ctx.m1, ctx.m2, ctx.tok1, ctx.tok2 = m1, m2, tok1, tok2
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        D = dy.shape[-1]
        dx = torch.empty_like(dy) if ctx.needs_input_grad[0] else None
        rets = []
        bufs = []
        for slot, tok in ((2, ctx.tok1), (4, ctx.tok2)):
            if tok is None or not ctx.needs_input_grad[slot]:
                bufs.append(None)
                rets.append(None)
                continue
            if getattr(tok, "_mesm_gb", None) is not None:  # a parameter: straight into the flat gradient buffer
                g, direct = grad_target(tok)
                bufs.append(g)
                rets.append(None if direct else g)
            else:
                g = kn.zeros((D,), dy.device)
                bufs.append(g)
                rets.append(g.view(tok.shape))
        kn.token_mix_bwd(dy, ctx.m1, ctx.m2, dx, bufs[0], bufs[1])
        if not kn.glue_deferring():  # (queued in a phase: ops._drive reports readiness once the phase is issued)
            flush_ready()
        return dx, None, rets[0], None, rets[1]


def token_mix(x, m1, tok1, m2=None, tok2=None):
    return TokenMixFn.apply(x, m1, tok1, m2, tok2)


class GatherRows2Fn(Function):
    """y[j] = valid[j] ? x2d[idx[j]] : 0, optionally L2-normalised; backward through the inverse map `inv`
    (source row -> gathered position or -1) built on the host: no zero fill, no atomics."""

    @staticmethod
    def forward(ctx, x2d, idx, inv, valid, normalize):
        x2d = _c(x2d)
        y, rn = kn.gather_rows_fwd(x2d, idx, valid, normalize)
        ctx.save_for_backward(y if normalize else None, rn)
        ctx.idx_shape, ctx.inv, ctx.valid, ctx.rows, ctx.normalize = idx.shape, inv, valid, x2d.shape[0], normalize
        if not ctx.needs_input_grad[0]:  # rows of a constant (position embeddings): consumers' gradients for them are dropped
            ctx.mark_non_differentiable(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        y, rn = ctx.saved_tensors
        dx = kn.gather_rows_bwd(_c(dy), y, rn, ctx.inv, ctx.valid, ctx.rows, ctx.normalize)
        return dx, None, None, None, None


def gather_rows2(x2d, idx, inv, valid=None, normalize=False):
    if not (torch.is_grad_enabled() and x2d.requires_grad):
        return kn.gather_rows_fwd(_c(x2d), idx, valid, normalize)[0]
    return GatherRows2Fn.apply(x2d, idx, inv, valid, normalize)


class AddTileFn(Function):
    """[a + b] repeated `reps` times along dim 0, formed WITHOUT autograd (a constant for the engine: the consumer block
    folds the gradient of its use into the gradient of the stacked a, mha's join_qp)."""

    @staticmethod
    def forward(ctx, a, b, reps):
        out = kn.add_tile(_c(a), _c(b), reps)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, g):
        return None, None, None


class GatherAddFn(Function):
    """a2d[idx] + b2d[idx] (rows masked by valid), formed WITHOUT autograd: the key-side input of an attention block whose
    gradient the block folds into the gradient of its plain key input (MHABlock xkp)."""

    @staticmethod
    def forward(ctx, a2d, b2d, idx, valid):
        y = kn.gather_add(_c(a2d), _c(b2d), idx, valid)
        ctx.mark_non_differentiable(y)
        return y

    @staticmethod
    def backward(ctx, g):
        return None, None, None, None


def glue_block(fn_cls, n_out, name):
    """An assembly Function as a block for par() / lockstep: inside the node's launch phases its kernels are QUEUED
    (kn.glue_deferred) and leave with the phase's one grouped assembly launch, forward and backward -- for assembly calls
    that are independent of everything else in their round."""

    class B:
        N_OUT = n_out

        @staticmethod
        def fwd(ctx, *a):
            with kn.glue_deferred():
                return fn_cls.forward(ctx, *a)

        @staticmethod
        def bwd(ctx, *g):
            with kn.glue_deferred():
                return fn_cls.backward(ctx, *g)

    B.__name__ = B.__qualname__ = name
    return B


StackRowsBlock = glue_block(StackRowsFn, lambda idx, gather, *xs: len(xs), "StackRowsBlock")
TokenMixBlock = glue_block(TokenMixFn, 1, "TokenMixBlock")
GatherRows2Block = glue_block(GatherRows2Fn, 1, "GatherRows2Block")
AddTileBlock = glue_block(AddTileFn, 1, "AddTileBlock")
GatherAddBlock = glue_block(GatherAddFn, 1, "GatherAddBlock")


def stack_rows_call(xs, gather, idx):
    return Call(StackRowsBlock, (idx, tuple(gather)) + tuple(xs))


def token_mix_call(x, m1, tok1, m2=None, tok2=None):
    return Call(TokenMixBlock, (x, m1, tok1, m2, tok2))


def gather_rows2_call(x2d, idx, inv, valid=None, normalize=False):
    return Call(GatherRows2Block, (x2d, idx, inv, valid, normalize))


def add_tile_call(a, b, reps):
    return Call(AddTileBlock, (a, b, reps))


def gather_add_call(a2d, b2d, idx, valid=None):
    return Call(GatherAddBlock, (a2d, b2d, idx, valid))


# ----------------------------------------------------------------------------- decoder reference points
class RefInitFn(Function):
    """ref (n, nq, 2) = sigmoid(refpoints_unsigmoid (nq, 2)) for every pair (transformer.py:197, 361), one launch each
    way; the parameter's gradient goes straight into its view."""

    @staticmethod
    def forward(ctx, p, n):
        pc = _c(p)
        out = torch.empty((n,) + tuple(p.shape), device=p.device, dtype=torch.float32)
        kn.check(kn.lib().mesm_ref_init_fwd(kn.ptr(pc), kn.ptr(out), n, pc.numel(), kn.stream_ptr()), "mesm_ref_init_fwd")
        ctx.save_for_backward(out)
        ctx.p = p
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        g, direct = grad_target(ctx.p)
        kn.check(kn.lib().mesm_ref_init_bwd(kn.ptr(out), kn.ptr(_c(dout)), kn.ptr(g), out.shape[0], ctx.p.numel(),
                                            kn.stream_ptr()), "mesm_ref_init_bwd")
        flush_ready()
        return (None if direct else g), None


def ref_init(p, n):
    return RefInitFn.apply(p, n)


class RefUpdateFn(Function):
    """sigmoid(delta + inverse_sigmoid(ref)) (transformer.py:36-40, 392-394; model.py:250)."""

    @staticmethod
    def forward(ctx, delta, ref, eps):
        delta, ref = _c(delta), _c(ref)
        out = torch.empty_like(delta)
        kn.check(kn.lib().mesm_ref_update_fwd(kn.ptr(delta), kn.ptr(ref), kn.ptr(out), delta.numel(), float(eps),
                                              kn.stream_ptr()), "mesm_ref_update_fwd")
        ctx.save_for_backward(out, ref)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dout):
        out, ref = ctx.saved_tensors
        dout = _c(dout)
        dd = torch.empty_like(out)
        dr = torch.empty_like(out) if ctx.needs_input_grad[1] else None
        kn.check(kn.lib().mesm_ref_update_bwd(kn.ptr(out), kn.ptr(ref), kn.ptr(dout), kn.ptr(dd), kn.ptr(dr),
                                              out.numel(), float(ctx.eps), kn.stream_ptr()), "mesm_ref_update_bwd")
        return dd, dr, None


def ref_update(delta, ref, eps=1e-3):
    return RefUpdateFn.apply(delta, ref, eps)


class RefInitSineFn(Function):
    """(ref, ref, ref, qsine, qsine): ref (n, nq, 2) = sigmoid(refpoints_unsigmoid) for every pair, written into its slot of
    the stacked reference points, and its sine embedding (transformer.py:343-351), ONE launch (were ref_init | query_sine).
    The three aliases of ref are for its three consumers -- the stacked output, the layer-0 width modulation and the first
    refinement --, the two of qsine for ref_point_head and the modulation: the backward kernel sums their gradients itself
    (were two launches and four element-wise adds of the autograd engine) into the parameter's gradient view."""

    @staticmethod
    def forward(ctx, p, n, D, slot):
        ctx.set_materialize_grads(False)
        pc = _c(p)
        ref = slot.t if slot is not None else torch.empty((n,) + tuple(p.shape), device=p.device, dtype=torch.float32)
        assert ref.shape == (n,) + tuple(p.shape) and p.shape[-1] == 2
        qs = torch.empty(tuple(ref.shape[:-1]) + (D,), device=p.device, dtype=torch.float32)
        kn.check(kn.lib().mesm_ref_step_fwd(kn.ptr(pc), pc.numel(), None, None, 0.0, None, None, kn.ptr(ref), kn.ptr(qs),
                                            None, ref.numel() // 2, D, kn.stream_ptr()), "mesm_ref_step_fwd")
        ctx.save_for_backward(ref)
        ctx.p, ctx.D = p, D
        return ref, ref.view_as(ref), ref.view_as(ref), qs, qs.view_as(qs)

    @staticmethod
    def backward(ctx, da, db, dc, dqs, dqs2):
        (ref,) = ctx.saved_tensors
        g, direct = grad_target(ctx.p)
        ds = [_c(t) if t is not None else None for t in (da, db, dc, dqs, dqs2)]
        kn.check(kn.lib().mesm_ref_init_sine_bwd(kn.ptr(ref), kn.ptr(ds[0]), kn.ptr(ds[1]), kn.ptr(ds[2]), kn.ptr(ds[3]),
                                                 kn.ptr(ds[4]), kn.ptr(g), ref.shape[0], ctx.p.numel(), ctx.D,
                                                 kn.stream_ptr()), "mesm_ref_init_sine_bwd")
        flush_ready()
        return (None if direct else g), None, None, None


def ref_init_sine(p, n, D, slot=None):
    """-> (ref for the stack, ref for qsine_scale, ref for the first refinement, qsine for ref_point_head, qsine for
    qsine_scale)"""
    return RefInitSineFn.apply(p, n, D, slot)


class RefStepFn(Function):
    """(new_ref, qsine, qscaled) at the boundary between two decoder layers (transformer.py:389-397 of layer l, :349-376 of
    layer l + 1), ONE launch each way (were ref_update | query_sine | qsine_scale and ref_update_bwd | qsine_scale_bwd):
    new_ref = sigmoid(delta + inverse_sigmoid(prev)) into its slot of the stacked reference points; from its DETACHED
    value (transformer.py:397) qsine = gen_sineembed_for_position(new_ref) -- no gradient -- and
    qscaled = qsine * scale * sigmoid(anchor) / new_ref[..., 1] -- gradients to scale and anchor only."""

    @staticmethod
    def forward(ctx, delta, prev, scale, anchor, D, slot, eps):
        ctx.set_materialize_grads(False)
        delta, prev, scale, anchor = _c(delta), _c(prev), _c(scale), _c(anchor)
        ref = slot.t if slot is not None else torch.empty_like(delta)
        assert ref.shape == delta.shape == prev.shape and delta.shape[-1] == 2
        R = delta.numel() // 2
        assert scale.numel() == R * D and anchor.numel() == R
        qs = torch.empty(tuple(delta.shape[:-1]) + (D,), device=delta.device, dtype=torch.float32)
        qsc = torch.empty_like(qs)
        kn.check(kn.lib().mesm_ref_step_fwd(None, 0, kn.ptr(delta), kn.ptr(prev), float(eps), kn.ptr(scale), kn.ptr(anchor),
                                            kn.ptr(ref), kn.ptr(qs), kn.ptr(qsc), R, D, kn.stream_ptr()),
                 "mesm_ref_step_fwd")
        ctx.save_for_backward(ref, prev, qs, scale, anchor)
        ctx.eps, ctx.D = eps, D
        ctx.mark_non_differentiable(qs)
        return ref, qs, qsc

    @staticmethod
    def backward(ctx, dref, _dqs, dqsc):
        ref, prev, qs, scale, anchor = ctx.saved_tensors
        nd, npv, nsc, nan = ctx.needs_input_grad[:4]
        R = ref.numel() // 2
        want_ref = (nd or npv) and dref is not None
        want_q = (nsc or nan) and dqsc is not None
        dd = torch.empty_like(ref) if want_ref else None
        dp = torch.empty_like(ref) if (want_ref and npv) else None
        dsc = torch.empty_like(scale) if (want_q and nsc) else None
        dan = torch.empty_like(anchor) if want_q else None
        if want_ref or want_q:
            kn.check(kn.lib().mesm_ref_step_bwd(kn.ptr(ref), kn.ptr(prev), kn.ptr(_c(dref)) if want_ref else None,
                                                float(ctx.eps), kn.ptr(qs), kn.ptr(scale), kn.ptr(anchor),
                                                kn.ptr(_c(dqsc)) if want_q else None, kn.ptr(dd), kn.ptr(dp), kn.ptr(dsc),
                                                kn.ptr(dan), R, ctx.D, kn.stream_ptr()), "mesm_ref_step_bwd")
        return (dd if nd else None, dp, dsc, dan if nan else None, None, None, None)


def ref_step(delta, prev, scale, anchor, D, slot=None, eps=1e-3):
    return RefStepFn.apply(delta, prev, scale, anchor, D, slot, eps)


class QSineScaleFn(Function):
    """qsine * scale * (sigmoid(anchor) / ref[..., 1]) (transformer.py:370-376)."""

    @staticmethod
    def forward(ctx, qsine, scale, anchor, ref):
        qsine, anchor, ref = _c(qsine), _c(anchor), _c(ref)
        scale = _c(scale) if scale is not None else None
        D = qsine.shape[-1]
        R = qsine.numel() // D
        out = torch.empty_like(qsine)
        kn.check(kn.lib().mesm_qsine_scale_fwd(kn.ptr(qsine), kn.ptr(scale), kn.ptr(anchor), kn.ptr(ref), kn.ptr(out),
                                               R, D, kn.stream_ptr()), "mesm_qsine_scale_fwd")
        ctx.save_for_backward(qsine, scale, anchor, ref)
        return out

    @staticmethod
    def backward(ctx, dout):
        qsine, scale, anchor, ref = ctx.saved_tensors
        dout = _c(dout)
        D = qsine.shape[-1]
        R = qsine.numel() // D
        dq = torch.empty_like(qsine)
        ds = torch.empty_like(qsine) if scale is not None else None
        da = torch.empty_like(anchor)
        dr = torch.empty_like(ref)
        kn.check(kn.lib().mesm_qsine_scale_bwd(kn.ptr(qsine), kn.ptr(scale), kn.ptr(anchor), kn.ptr(ref), kn.ptr(dout),
                                               kn.ptr(dq), kn.ptr(ds), kn.ptr(da), kn.ptr(dr), R, D,
                                               kn.stream_ptr()), "mesm_qsine_scale_bwd")
        return dq, ds, da, (dr if ctx.needs_input_grad[3] else None)


def qsine_scale(qsine, scale, anchor, ref):
    return QSineScaleFn.apply(qsine, scale, anchor, ref)
