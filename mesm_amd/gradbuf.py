"""Flat fp32 gradient buffer shared by every trainable parameter of a model.

The kernels accumulate weight / bias / LayerNorm gradients straight into views of ONE
contiguous buffer (dW GEMMs with split-K atomics, LayerNorm dgamma/dbeta atomics), and
``param.grad`` is made to alias that view.  This keeps ``loss.backward()`` /
``optimizer.step()`` / ``clip_grad_norm_`` of the reference's train loop (train.py:68-72)
working unchanged while giving data-parallel training a single buffer to all-reduce
(ddp.py) instead of 262 tensors.

Zeroing protocol (matches ``optimizer.zero_grad()`` with set_to_none=True or False being
called anywhere between two backward passes, train.py:68): ``begin_step()`` is called by
the model's training forward; the first gradient written in the following backward checks
the parameters: all ``.grad is None`` -> one memset of the flat buffer; otherwise the
existing contents are kept (gradient accumulation) and only still-None views are zeroed.
Parameters that no kernel touches keep ``.grad = None`` exactly like under autograd.
"""
import torch


def _zero(t):
    """memset of a gradient buffer: on the GPU through mesm_fill_ranges (write-through stores: the 55 MB do not sit dirty
    in L2 until the end of the launch), elsewhere -- or for an odd size -- torch's fill"""
    if t.is_cuda and t.is_contiguous() and t.data_ptr() % 16 == 0 and (t.numel() * t.element_size()) % 16 == 0 and t.numel() > 0:
        from . import kernels as kn  # (late: kernels imports this module)
        kn.fill_zero(t)
    else:
        t.zero_()


class Pack:
    """Several parameters laid out back to back in the flat buffers, seen as ONE matrix / vector: e.g. the
    decoder's sa_qcontent / sa_kcontent / sa_v weights as a (3d, d) matrix, so that the three projections of the
    same input are one GEMM forward, one dX GEMM and one dW GEMM backward (and autograd has no gradient fan-in to
    add up).  `weight()` / `grad()` are views of the flat parameter / gradient buffers; a Pack quacks like a
    parameter for ops.grad_target (it acquires every member)."""

    def __init__(self, gb, members, shape):
        self.gb, self.members, self.shape = gb, members, shape
        self.offset = gb.offsets[next(i for i, q in enumerate(gb.params) if q is members[0])]
        self.numel = sum(m.numel() for m in members)
        self._w = None

    @property
    def flat(self):
        return self.gb.flat

    def weight(self, flat_params):
        """view of the flat PARAMETER buffer (model.flat_params()); carries the hooks grad_target looks for"""
        if self._w is None or self._w.data_ptr() != flat_params.data_ptr() + 4 * self.offset:
            w = flat_params[self.offset:self.offset + self.numel].view(self.shape)
            w._mesm_gb = self
            self._w = w
        return self._w

    def acquire(self, _w):
        for m in self.members:
            self.gb.acquire(m)
        return self.gb.flat[self.offset:self.offset + self.numel].view(self.shape)


class GradBuffer:
    def __init__(self, named_params, packs=None):
        """packs: {key: [parameter names]} -- members are placed back to back (in the given order) in the
        flat layout; every member's numel must be a multiple of 4.  The ORDER OF `params` (optimizer state
        indices, checkpoints) is unaffected: only `offsets` are permuted."""
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        by_name = dict(named_params)
        first_of, member_of = {}, set()
        for key, names in (packs or {}).items():
            assert all(by_name[n].numel() % 4 == 0 for n in names), key
            first_of[names[0]] = names
            member_of.update(names)
        off_of = {}
        off = 0
        for n, p in named_params:
            if n in off_of:
                continue
            for m in first_of.get(n, [n] if n not in member_of else []):
                off_of[m] = off
                off += (by_name[m].numel() + 3) // 4 * 4  # keep every view 16-byte aligned
        assert len(off_of) == len(self.params), "a pack's first member must come first in named_parameters order"
        self.offsets = [off_of[n] for n in self.names]
        self.numel = off
        self.packs = {}
        for key, names in (packs or {}).items():
            ms = [by_name[n] for n in names]
            rows = sum(m.shape[0] for m in ms)
            self.packs[key] = Pack(self, ms, (rows,) + tuple(ms[0].shape[1:]))
        self.flat = None
        self.pending = False
        self.on_ready = None  # optional callback(param) used by ddp.GradReducer
        for p in self.params:
            p._mesm_gb = self
            # gradients that reach a parameter through plain autograd (tokens, embeddings used by
            # torch glue ops) are moved into the flat buffer as soon as they are accumulated
            p.register_post_accumulate_grad_hook(self._adopt)

    def _adopt(self, p):
        if self.flat is None:
            return
        g, v = p.grad, p._mesm_gview
        if g is None or g.data_ptr() == v.data_ptr():
            return
        p.grad = None  # let _open() see the pre-backward state of this parameter
        if self.pending:
            self._open()
        v.copy_(g)  # (overwrites the whole view: no zero fill in front of it)
        p.grad = v
        if self.on_ready is not None:
            self.on_ready(p)

    def ensure(self, device):
        if self.flat is not None and self.flat.device == device:
            return
        self.flat = torch.zeros(self.numel, device=device, dtype=torch.float32)
        for p, off in zip(self.params, self.offsets):
            p._mesm_gview = self.flat[off:off + p.numel()].view(p.shape)
            if p.grad is not None:
                p.grad = None

    def begin_step(self):
        self.pending = True

    def _open(self):
        self.pending = False
        if all(p.grad is None for p in self.params):
            _zero(self.flat)
        else:
            for p in self.params:
                if p.grad is None:
                    p._mesm_gview.zero_()

    def acquire(self, p):
        """View to ACCUMULATE p's gradient into; makes p.grad alias it."""
        if self.pending:
            self._open()
        v = p._mesm_gview
        g = p.grad
        if g is None:
            p.grad = v
        elif g.data_ptr() != v.data_ptr():
            v.copy_(g)
            p.grad = v
        if self.on_ready is not None:
            _touched.append(p)  # reported by flush_ready() once the writing kernels are enqueued
        return v

    def zero(self):
        if self.flat is not None:
            _zero(self.flat)

    def attach_all(self):
        """Point every parameter's .grad at its view (used by graph-captured steps)."""
        for p in self.params:
            p.grad = p._mesm_gview


_touched = []


def flush_ready():
    """Called at the end of every backward block: the kernels that accumulate into the views
    handed out since the last flush are now enqueued, so a reducer may order a collective after them."""
    while _touched:
        p = _touched.pop()
        gb = p._mesm_gb
        if gb.on_ready is not None:
            gb.on_ready(p)


def grad_target(p):
    """(tensor to accumulate into, direct) for a parameter-like tensor.  `direct` means the
    tensor aliases p.grad and autograd must receive None for it."""
    gb = getattr(p, "_mesm_gb", None)
    if gb is not None and gb.flat is not None:
        return gb.acquire(p), True
    return torch.zeros_like(p), False
