"""Explicit fork/join of an independent sub-graph onto a second HIP stream, forward AND backward.

Independent branches of a captured HIP graph do run concurrently on MI355X (1.6x on chains of
small GEMMs, tools/probe_graph_par.py), and the kernels of this model are far too small to fill
256 CUs.  Letting autograd manage a second stream does not work, though: the engine inserts its
cross-stream waits where a gradient is PRODUCED in host order, and a wait in the middle of a stream
blocks everything enqueued behind it, so the two streams ended up running one after the other.

Here the fork and join points are placed by hand, as early / late as the data allows:

  forward   fork  = an event recorded on the main stream when the branch inputs exist
            join  = caller waits `branch.done_fwd` right before the outputs are consumed
  backward  fork  = the moment the output gradients exist (SideCall.backward, which the engine
                    calls early because the SideCall node is created late in the forward)
            join  = JoinGrad.backward of every input, which the engine calls LATE because the
                    JoinGrad nodes are created early in the forward (right where the inputs are made)

SideCall runs `fn` under torch.enable_grad() on detached inputs (a private autograd sub-graph whose
nodes are therefore bound to the side stream) and replays it with torch.autograd.backward on the
side stream; the input gradients are handed to the JoinGrad blocks, never returned through the
engine.  Parameter gradients are accumulated by the kernels into the flat buffer with atomics.
"""
import torch
from torch.autograd import Function


class Branch:
    def __init__(self, device, on_fork=None):
        self.stream = torch.cuda.Stream(device=device)
        self.on_fork = on_fork      # called on the main stream right before the backward fork
        self.fork_event = None      # set by the caller (forward fork point)
        self.done_fwd = None
        self.done_bwd = None
        self.in_grads = {}

    def mark_fork(self):
        self.fork_event = torch.cuda.Event()
        self.fork_event.record(torch.cuda.current_stream())


class JoinGrad(Function):
    """Identity.  Backward: wait for the side stream's backward, then add the gradient the branch
    produced for this input.  Create it EARLY in the forward (its backward then runs late)."""

    @staticmethod
    def forward(ctx, x, branch, slot):
        ctx.branch, ctx.slot = branch, slot
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        br = ctx.branch
        extra = br.in_grads.pop(ctx.slot, None)
        if extra is None:
            return g, None, None
        main = torch.cuda.current_stream()
        main.wait_event(br.done_bwd)
        extra.record_stream(main)
        return (extra if g is None else g + extra), None, None


class SideCall(Function):
    @staticmethod
    def forward(ctx, branch, fn, slots, *inputs):
        main = torch.cuda.current_stream()
        side = branch.stream
        side.wait_event(branch.fork_event)
        with torch.cuda.stream(side), torch.enable_grad():
            ins = []
            for t in inputs:
                t.record_stream(side)
                ins.append(t.detach().requires_grad_(t.requires_grad))
            outs = fn(*ins)
            outs = outs if isinstance(outs, tuple) else (outs,)
            branch.done_fwd = torch.cuda.Event()
            branch.done_fwd.record(side)
        for o in outs:
            o.record_stream(main)
        ctx.branch, ctx.ins, ctx.outs, ctx.slots, ctx.n_in = branch, ins, outs, slots, len(inputs)
        ctx.set_materialize_grads(False)
        return tuple(o.detach() for o in outs)

    @staticmethod
    def backward(ctx, *gouts):
        br, side = ctx.branch, ctx.branch.stream
        main = torch.cuda.current_stream()
        if br.on_fork is not None:
            br.on_fork()
        pairs = [(o, g) for o, g in zip(ctx.outs, gouts) if g is not None and o.requires_grad]
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            for _, g in pairs:
                g.record_stream(side)
            if pairs:
                torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
            for slot, t in zip(ctx.slots, ctx.ins):
                if t.grad is not None:
                    br.in_grads[slot] = t.grad
                    t.grad = None
            br.done_bwd = torch.cuda.Event()
            br.done_bwd.record(side)
        ctx.ins = ctx.outs = None
        return (None, None, None) + (None,) * ctx.n_in


def side_call(branch, fn, joined_inputs):
    """joined_inputs: list of (slot, tensor) where tensor = JoinGrad.apply(x, branch, slot) made
    earlier.  Returns the outputs of fn (produced on the side stream: wait branch.done_fwd on the
    consuming stream before reading them)."""
    slots = [s for s, _ in joined_inputs]
    tensors = [t for _, t in joined_inputs]
    return SideCall.apply(branch, fn, slots, *tensors)
