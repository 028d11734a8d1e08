"""One host -> device transfer for a set of small tensors.

The data-dependent host decisions of a step (model.Plan, criterion.TargetPlan: ~35 index / mask tensors of a few
hundred bytes each) used to reach the device as ~35 separate pageable copies, each of which blocks the host
until the stream has drained.  An Arena lays all of them out in ONE device buffer (16-byte aligned segments,
every tensor a typed view of it) with two pinned host mirrors; `upload(arrays)` packs the numpy arrays into
the mirror that is not in flight and issues a single asynchronous copy.  A HIP graph captured with arena views
as its index tensors is retargeted to a new batch by one `upload` (graphed.GraphedStep.load_batch)."""
import numpy as np
import torch

_TORCH = {np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32,
          np.dtype(np.float64): torch.float64, np.dtype(np.bool_): torch.bool, np.dtype(np.uint8): torch.uint8}


class Arena:
    def __init__(self, arrays, device, first=()):
        """arrays: {name: np.ndarray}; shapes and dtypes are fixed from here on.  first: names laid out at the front,
        next to each other (what `upload(only=...)` refreshes on its own: one short copy)."""
        self.specs, off = [], 0
        order = [n for n in first if n in arrays] + [n for n in arrays if n not in first]
        for name in order:
            a = arrays[name]
            a = np.ascontiguousarray(a)
            self.specs.append((name, a.shape, a.dtype, off, a.nbytes))
            off += (a.nbytes + 15) // 16 * 16
        self.nbytes = max(off, 16)
        self.device = torch.device(device)
        self.dev = torch.empty(self.nbytes, dtype=torch.uint8, device=self.device)
        self.views = {}
        for name, shape, dt, o, nb in self.specs:
            t = self.dev[o:o + nb].view(_TORCH[np.dtype(dt)]) if nb else torch.empty(0, dtype=_TORCH[np.dtype(dt)], device=self.device)
            self.views[name] = t.view(shape)
        self._pins, self._turn = None, 0
        self.upload(arrays)

    def _check(self, arrays):
        if set(arrays) != {s[0] for s in self.specs}:
            raise ValueError("Arena.upload: tensor set changed: %s" % sorted(set(arrays) ^ {s[0] for s in self.specs}))
        for name, shape, dt, _, _ in self.specs:
            a = arrays[name]
            if tuple(a.shape) != tuple(shape) or np.dtype(a.dtype) != np.dtype(dt):
                raise ValueError("Arena.upload: %s does not fit (%s %s -> %s %s)" % (name, shape, dt, a.shape, a.dtype))

    def span(self, names):
        """byte range [lo, hi) of the arena that covers the named tensors (16-byte granules)"""
        hit = [(o, o + (nb + 15) // 16 * 16) for name, _, _, o, nb in self.specs if name in names]
        if not hit:
            return 0, 0
        return min(h[0] for h in hit), min(max(h[1] for h in hit), self.nbytes)

    def pack(self, arrays, lo, hi, out):
        """the bytes [lo, hi) of the arena image of `arrays` into the uint8 numpy array `out` (length hi - lo)"""
        for name, _, _, o, nb in self.specs:
            if nb and lo <= o < hi:
                out[o - lo:o - lo + nb] = np.ascontiguousarray(arrays[name]).reshape(-1).view(np.uint8)

    def check(self, arrays):
        """raise ValueError if `arrays` cannot be uploaded into this arena (nothing is modified)"""
        self._check(arrays)

    def upload(self, arrays, only=None):
        """only: names whose values changed since the last full upload (everything else is already on the device):
        the byte range that covers them is packed and copied instead of the whole arena."""
        self._check(arrays)
        lo, hi = 0, self.nbytes
        if only is not None:
            hit = [(o, o + (nb + 15) // 16 * 16) for name, _, _, o, nb in self.specs if name in only]
            if hit:
                lo, hi = min(h[0] for h in hit), min(max(h[1] for h in hit), self.nbytes)
        if self.device.type != "cuda":
            host = self.dev.numpy()
            for name, _, _, o, nb in self.specs:
                host[o:o + nb] = np.ascontiguousarray(arrays[name]).reshape(-1).view(np.uint8)
            return
        if self._pins is None:
            self._pins = [[torch.empty(self.nbytes, dtype=torch.uint8, pin_memory=True), None] for _ in range(2)]
        buf, ev = self._pins[self._turn]
        if ev is not None:
            ev.synchronize()  # the copy that used this mirror two uploads ago
        host = buf.numpy()
        for name, _, _, o, nb in self.specs:
            if nb and lo <= o < hi:
                host[o:o + nb] = np.ascontiguousarray(arrays[name]).reshape(-1).view(np.uint8)
        if lo == 0 and hi == self.nbytes:
            self.dev.copy_(buf, non_blocking=True)
        else:
            self.dev[lo:hi].copy_(buf[lo:hi], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._pins[self._turn][1] = ev
        self._turn ^= 1
