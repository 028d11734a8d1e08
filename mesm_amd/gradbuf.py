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


class GradBuffer:
    def __init__(self, named_params):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4  # keep every view 16-byte aligned
        self.numel = off
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
        else:
            v.zero_()
        v.copy_(g)
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
            self.flat.zero_()
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
            self.flat.zero_()

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
