"""Optimizer tail of the reference train loop on the flat buffers.

    train.py:68-72     optimizer.zero_grad(); loss.backward();
                       nn.utils.clip_grad_norm_(model.parameters(), opt.grad_clip); optimizer.step()
    runner.py:348-352  AdamW(lr, weight_decay) + StepLR(lr_drop, gamma)

The model's gradients already live in ONE flat fp32 buffer (gradbuf.py).  FlatAdamW moves the
parameters into a second flat buffer with the same layout (every `param.data` becomes a view, so the
modules, state_dict() and checkpoints are unaffected), keeps exp_avg / exp_avg_sq flat as well, and
runs clipping + the AdamW update for all 273 tensors in two kernel launches (csrc/optim.hip)
instead of ~8 foreach launches over 262 views.  Step count and learning rate are device scalars,
so `step()` can be captured in the same HIP graph as the training step.

Drop-in use in the reference loop:

    optimizer, lr_scheduler = build_optimizer(opt, model)        # this module's build_optimizer
    ...
    optimizer.zero_grad(); loss.backward()
    optimizer.step(grad_clip=opt.grad_clip)     # replaces clip_grad_norm_ + optimizer.step()

`clip_grad_norm_(model, max_norm)` is also provided for loops that keep the two calls apart.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


def _sumsq(gb, step_t=None):
    flat = gb.flat
    partials = torch.empty(1024, device=flat.device, dtype=torch.float32)
    np_out = ctypes.c_int32(0)
    check(lib().mesm_grad_sumsq(ptr(flat), flat.numel(), ptr(partials), ctypes.byref(np_out), ptr(step_t),
                                stream_ptr()), "mesm_grad_sumsq")
    return partials, np_out.value


def clip_grad_norm_(model, max_norm):
    """nn.utils.clip_grad_norm_(model.parameters(), max_norm) on the flat gradient buffer (two
    launches).  Returns the total norm as a device scalar (no host sync)."""
    gb = model.gradbuf()
    partials, n_p = _sumsq(gb)
    norm = torch.empty(1, device=gb.flat.device, dtype=torch.float32)
    check(lib().mesm_clip_grad(ptr(gb.flat), gb.flat.numel(), ptr(partials), n_p, float(max_norm), ptr(norm),
                               stream_ptr()), "mesm_clip_grad")
    return norm[0]


class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt of
    the bias-corrected second moment, amsgrad off) over flat buffers."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model = model
        gb = model.gradbuf()
        params = list(gb.params)
        super().__init__([{"params": params}], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.gb = gb
        self.flat_p = self.exp_avg = self.exp_avg_sq = None
        self.step_t = self.lr_t = None
        self._active = None
        self._active_key = None
        self.last_norm = None
        if params[0].is_cuda:
            # the flat state is created NOW, not at the first step(): a HIP graph captured after this
            # constructor sees the final parameter addresses (model.flat_params is idempotent and
            # build_model has normally done it already)
            self._ensure()

    # ------------------------------------------------------------------ flat state
    def _ensure(self):
        gb = self.gb
        dev = next(iter(gb.params)).device
        gb.ensure(dev)
        flat = self.model.flat_params()  # every param.data is a view of it (no-op when already flat)
        if self.flat_p is flat:
            return
        self.flat_p = flat
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.step_t = torch.zeros(1, device=dev, dtype=torch.int32)
        self.lr_t = torch.full((1,), float(self.param_groups[0]["lr"]), device=dev, dtype=torch.float32)

    def _active_mask(self):
        """groups of 4 elements that belong to a parameter whose .grad is not None this step"""
        gb = self.gb
        key = tuple(p.grad is not None for p in gb.params)
        if key != self._active_key:
            mask = torch.zeros(gb.numel // 4, dtype=torch.uint8)
            for p, off, on in zip(gb.params, gb.offsets, key):
                if on:
                    mask[off // 4:(off + p.numel() + 3) // 4] = 1
            self._active = mask.to(self.flat_p.device)
            self._active_key = key
        return self._active

    # ------------------------------------------------------------------ optimizer protocol
    @torch.no_grad()
    def step(self, closure=None, grad_clip=0.0):
        """grad_clip > 0 fuses nn.utils.clip_grad_norm_(params, grad_clip) into the update (the
        gradients themselves are left unscaled)."""
        assert closure is None
        self._ensure()
        g = self.param_groups[0]
        if float(g["lr"]) != getattr(self, "_lr_host", None):  # StepLR edits param_groups[0]["lr"]
            self._lr_host = float(g["lr"])
            self.lr_t.fill_(self._lr_host)
        active = self._active_mask()
        if grad_clip and grad_clip > 0:
            partials, n_p = _sumsq(self.gb, self.step_t)
            self.last_norm = torch.empty(1, device=self.flat_p.device, dtype=torch.float32)
        else:
            partials, n_p = None, 0
            self.step_t += 1
            self.last_norm = None
        b1, b2 = g["betas"]
        check(lib().mesm_adamw_step(ptr(self.flat_p), ptr(self.gb.flat), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                    ptr(active), self.flat_p.numel(), ptr(partials), n_p,
                                    float(grad_clip or 0.0), ptr(self.lr_t), float(b1), float(b2),
                                    float(g["eps"]), float(g["weight_decay"]), ptr(self.step_t),
                                    ptr(self.last_norm), stream_ptr()), "mesm_adamw_step")

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad(set_to_none=set_to_none)

    # ------------------------------------------------------------------ checkpoints (torch.optim.AdamW format)
    def state_dict(self):
        self._ensure()
        step = int(self.step_t.item())
        state = {}
        for i, (p, off) in enumerate(zip(self.gb.params, self.gb.offsets)):
            sl = slice(off, off + p.numel())
            state[i] = {"step": torch.tensor(float(step)),
                        "exp_avg": self.exp_avg[sl].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[sl].view(p.shape).clone()}
        groups = [{k: v for k, v in self.param_groups[0].items() if k != "params"}]
        groups[0]["params"] = list(range(len(self.gb.params)))
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        self._ensure()
        step = 0
        for i, (p, off) in enumerate(zip(self.gb.params, self.gb.offsets)):
            st = sd["state"].get(i)
            if st is None:
                continue
            sl = slice(off, off + p.numel())
            self.exp_avg[sl].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[sl].copy_(st["exp_avg_sq"].reshape(-1))
            step = max(step, int(float(st["step"])))
        self.step_t.fill_(step)
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v


def build_optimizer(opt, model):
    """runner.py:348-352 with the flat AdamW: (optimizer, StepLR scheduler)."""
    optimizer = FlatAdamW(model, lr=opt.lr, weight_decay=opt.weight_decay)
    lr_scheduler = torch.optim.lr_scheduler.StepLR(optimizer, opt.lr_drop, gamma=opt.gamma)
    return optimizer, lr_scheduler
