"""Loader-side host path (/root/reference dataset/base.py:288-384 `collate` + `prepare_batch_input`
behind `DataLoader(num_workers=8)` in train.py): the host work that a batch needs before the captured step can replay on
it -- clip / word padding to the graph's fixed extents, pair padding to the pair bucket, the forward's index plan, the
criterion's flattened targets -- done in loader WORKER processes and shipped as one `prepared` dict, so that the training
process only draws the two host-RNG decisions of the reference's forward, uploads one arena and replays
(graphed.StepCache.run_prepared).  Round-3 measurement this answers: 10.6 ms per step single-threaded in the training
process against a 4.67 ms replay.

    cache = StepCache(model, crit, dataset_name, pad=(Lv, Lw), pairs=16, group_caps=(5, 9))
    loader = prepared_loader(dataset_of_collated_batches, cache.pipeline(), num_workers=4)
    for prep in loader:
        total, step = cache.run_prepared(prep)

Workers are FORKED (no exec: the GPU box refuses an exec from a process that has initialised the GPU) and never touch
the device.  The big feature tensors do NOT travel through the DataLoader's queues (round 4 measured that path at 15.6 ms
per step against 8.7 in-process: two copies and a pickle of 35 MB per batch): `prepared_loader(..., ring=True)` gives the
workers a ring of SHARED, PAGE-LOCKED host slots (allocated before the fork, registered with the HIP runtime once) --
a worker writes the padded features of batch i into slot i % slots and returns only a descriptor; the training process
copies from the slot to the device asynchronously (`GraphedStep._stage_big`), the small arrays come as before."""
import collections

import numpy as np
import torch

from .hostplan import HostSpec, _round_up


class HostPipeline:
    """StepCache's host half, picklable.  prepare(raw collated host batch) -> prepared dict:
       key      bucket key of the graph that serves it (StepCache.key + group bucket)
       arr      {name: tensor} arena content (placeholder draws), pmeta / tmeta, wm (word validity), caps
       big      {video_feat, words_id}: the real rows, padded to the fixed clip / word extents
       groups, n_real, num_clips, words_weight
       raw      the raw batch when keep_raw (a bucket's first batch has to go through StepCache.run to be captured)"""
    CLIP_KEYS = ("video_feat", "video_mask", "clip_mask", "saliency_label")
    WORD_KEYS = ("words_id", "words_mask", "words_weight", "unknown_mask", "words_label")
    BIG = 1 << 20

    def __init__(self, spec, pad=None, pairs=None, group_caps=None, keep_raw=True):
        assert isinstance(spec, HostSpec)
        self.spec, self.pad, self.pairs = spec, pad, pairs
        self.group_caps = tuple(sorted(group_caps)) if group_caps else None
        self.keep_raw = keep_raw

    @staticmethod
    def _pad_dim1(t, L):
        if t is None or not torch.is_tensor(t) or t.dim() < 2 or t.shape[1] >= L:
            return t
        out = t.new_zeros((t.shape[0], L) + tuple(t.shape[2:]))
        out[:, :t.shape[1]] = t
        return out

    def padded(self, batch):
        if self.pad is None:
            return batch
        Lv, Lw = self.pad
        if batch["video_feat"].shape[1] > Lv or batch["words_id"].shape[1] > Lw:
            raise ValueError("HostPipeline: batch extents (%d clips, %d words) exceed pad=%r"
                             % (batch["video_feat"].shape[1], batch["words_id"].shape[1], self.pad))
        b = dict(batch)
        for k in self.CLIP_KEYS:
            if k in b:
                b[k] = self._pad_dim1(b[k], Lv)
        for k in self.WORD_KEYS:
            if k in b:
                b[k] = self._pad_dim1(b[k], Lw)
        return b

    def prepare(self, raw):
        from .batching import pad_pairs
        batch = self.padded(raw)
        if self.pairs:
            batch = pad_pairs(batch, _round_up(batch["video_feat"].shape[0], self.pairs))
        n = batch["video_mask"].shape[0]
        key = ((n,) + tuple(batch["video_feat"].shape[1:]), (n,) + tuple(batch["words_id"].shape[1:]))
        gcap = None
        if self.group_caps:
            gmax = int(batch["num_clips"].max())
            gcap = next((c for c in self.group_caps if c >= gmax), gmax)
            key = key + (gcap,)
        groups = [int(g) for g in batch["num_clips"].tolist()]
        n_real = batch.get("_n_real")
        host = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        for k in ("norm_span", "norm_moment"):
            if isinstance(batch.get(k), list):
                host[k] = batch[k]
        caps = self.spec.resolve_caps("auto", batch, groups, gcap)
        # placeholder draws: the training process draws (the reference's RNG stream lives there)
        N = n
        neg0 = torch.zeros(N, dtype=torch.int64).numpy()
        wm = self.spec.words_mask(host)
        mw0 = torch.zeros((N, wm.shape[1]), dtype=torch.bool).numpy() if self.spec.rec_fw else None
        arr, pmeta, tmeta, wm = self.spec.host_arrays(host, groups, caps, n_real, neg0, mw0, self.BIG)
        big = {k: batch[k].contiguous() for k in ("video_feat", "words_id")
               if batch[k].numel() * batch[k].element_size() > self.BIG}
        # the ~35 small arrays travel as NUMPY arrays: pickled by value into the loader's pipe (57 KB in all), where torch
        # tensors would each go through a shared-memory file descriptor (~3 ms per batch on the receiving side)
        return {"key": key, "spec": self.spec, "groups": groups, "n_real": n_real, "caps": caps,
                "arr": {k: np.array(v, copy=True, order="C") for k, v in arr.items()},
                "pmeta": pmeta, "tmeta": tmeta, "wm": wm, "big": big, "num_clips": batch["num_clips"],
                "words_weight": batch.get("words_weight"), "raw": raw if self.keep_raw else None}

    def big_bytes(self, raw):
        """ring-slot bytes the big tensors of `raw` need once prepared (shapes only, nothing is copied)"""
        n = raw["video_feat"].shape[0]
        if self.pairs:
            n = _round_up(n, self.pairs)
        tot = 0
        for k, L in (("video_feat", self.pad[0] if self.pad else None), ("words_id", self.pad[1] if self.pad else None)):
            t = raw[k]
            ext = max(t.shape[1], L) if (L is not None and t.dim() >= 2) else (t.shape[1] if t.dim() >= 2 else 1)
            nb = n * ext * int(np.prod(t.shape[2:])) * t.element_size()
            if nb > self.BIG:
                tot += (nb + 255) // 256 * 256
        return tot

    __call__ = prepare


class PinnedRing:
    """`slots` host buffers of `slot_bytes` each in ONE shared-memory allocation that forked workers inherit, page-locked
    (cudaHostRegister: a device copy from it is asynchronous) when a GPU is present.  put(slot, {name: tensor}) copies the
    tensors into the slot back to back and returns picklable descriptors; get(descriptors) gives zero-copy views."""
    _DT = {"float32": torch.float32, "int64": torch.int64, "int32": torch.int32, "float16": torch.float16,
           "uint8": torch.uint8, "bool": torch.bool, "float64": torch.float64}

    def __init__(self, slot_bytes, slots, pin=True):
        self.slot_bytes = (int(slot_bytes) + 4095) // 4096 * 4096
        self.slots = int(slots)
        self.buf = torch.empty(self.slot_bytes * self.slots, dtype=torch.uint8).share_memory_()
        self.pinned = False
        if pin and torch.cuda.is_available():
            try:
                rc = torch.cuda.cudart().cudaHostRegister(self.buf.data_ptr(), self.buf.numel(), 0)
                self.pinned = int(rc) == 0
            except Exception:  # a runtime without host registration: the ring still works, copies are synchronous
                self.pinned = False
        if self.pinned:
            # the registration ends before the shared buffer is freed, in the process that made it
            import os
            import weakref
            weakref.finalize(self, PinnedRing._unregister, self.buf.data_ptr(), os.getpid())

    @staticmethod
    def _unregister(addr, pid):
        import os
        if os.getpid() != pid:  # forked workers inherit the object, not the registration
            return
        try:
            torch.cuda.cudart().cudaHostUnregister(addr)
        except Exception:
            pass

    def put(self, slot, tensors):
        off, desc = 0, {}
        base = int(slot) % self.slots * self.slot_bytes
        for k, t in tensors.items():
            t = t.contiguous()
            nb = t.numel() * t.element_size()
            if off + nb > self.slot_bytes:
                raise ValueError("PinnedRing: %s (%d bytes at offset %d) does not fit a slot of %d bytes"
                                 % (k, nb, off, self.slot_bytes))
            dst = self.buf[base + off:base + off + nb].view(t.dtype).view(t.shape)
            dst.copy_(t)
            desc[k] = ("ring", int(slot) % self.slots, off, tuple(t.shape), str(t.dtype).replace("torch.", ""))
            off += (nb + 255) // 256 * 256
        return desc

    def get(self, desc):
        out = {}
        for k, d in desc.items():
            if not (isinstance(d, tuple) and len(d) == 5 and d[0] == "ring"):
                out[k] = d
                continue
            _, slot, off, shape, dt = d
            dtype = self._DT[dt]
            nb = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
            base = slot * self.slot_bytes + off
            out[k] = self.buf[base:base + nb].view(dtype).view(shape)
        return out


class _PreparedDataset(torch.utils.data.Dataset):
    """dataset of COLLATED raw batches (what the reference's collate returns) -> prepared batches"""

    def __init__(self, batches, pipeline, ring=None):
        self.batches, self.pipeline, self.ring = batches, pipeline, ring

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, i):
        prep = self.pipeline.prepare(self.batches[i])
        if self.ring is not None and prep["big"]:
            prep["big"] = self.ring.put(i, prep["big"])   # only a descriptor travels back
            for k in ("wm", "num_clips", "words_weight"):  # small tensors by value as well (re-wrapped on arrival)
                if torch.is_tensor(prep.get(k)):
                    prep[k] = ("np", prep[k].numpy())
        return prep


# events of the host -> device copies GraphedStep._stage_big issued most recently (appended there): a ring slot may be
# rewritten only after the copy that read it has finished
H2D_EVENTS = collections.deque(maxlen=16)


class PreparedLoader:
    """iterable over prepared batches; with a ring, `big` comes back as views of the ring's slots.

    CONTRACT of the ring path: the `big` views of a prepared batch are valid UNTIL THE NEXT BATCH IS FETCHED
    from this iterator -- the fetch hands the slot's index on to a worker once the device copies that read it have
    finished (the events GraphedStep._stage_big records, at most `keep` per batch: video_feat and words_id).  Consume a
    batch (StepCache.run_prepared) before fetching the next one; `list(loader)` or a look-ahead queue of prepared batches
    would hold views of slots that workers overwrite.  Without a ring (ring=False) batches own their memory."""

    def __init__(self, loader, ring, keep=2):
        self.loader, self.ring, self.keep = loader, ring, keep

    def __len__(self):
        return len(self.loader)

    def _fence(self, all_=False):
        # the DataLoader hands index m + prefetch x workers to a worker when batch m is fetched here; the slot it will
        # fill was read by a copy issued >= 2 batches ago (ring size, prepared_loader): wait for everything but the
        # copies of the batch fetched last (normally finished long ago: an event query)
        evs = list(H2D_EVENTS)
        for ev in (evs if all_ else evs[:-self.keep]):
            ev.synchronize()

    def __iter__(self):
        if self.ring is None:
            yield from self.loader
            return
        self._fence(all_=True)  # (a new pass starts at slot 0 again)
        it = iter(self.loader)
        while True:
            self._fence()
            try:
                prep = next(it)
            except StopIteration:
                return
            if isinstance(prep.get("big"), dict):
                prep["big"] = self.ring.get(prep["big"])
            for k in ("wm", "num_clips", "words_weight"):
                v = prep.get(k)
                if isinstance(v, tuple) and len(v) == 2 and v[0] == "np":
                    prep[k] = torch.from_numpy(v[1])
            yield prep


def _identity(x):
    return x


def prepared_loader(batches, pipeline, num_workers=4, pin_memory=False, prefetch_factor=2, persistent=True, ring=False,
                    slot_bytes=None):
    """DataLoader over a sequence (or map-style dataset) of collated raw host batches whose workers run
    `pipeline.prepare` (forked workers).  pin_memory: have the loader's pin thread copy every tensor into pinned memory
    -- measured on the bench box (tools/loader_path_probe.py) this costs more than it saves for 35 MB feature tensors
    (the pageable copy already runs on a copy stream into a double-buffered device staging area, GraphedStep._stage_big),
    hence off by default.  For a real dataset use
    `DataLoader(dataset, batch_size=..., collate_fn=lambda items: pipeline.prepare(collate(items)), ...)` the same way."""
    kw = {}
    if num_workers > 0:
        kw = dict(multiprocessing_context="fork", prefetch_factor=prefetch_factor, persistent_workers=persistent)
    rg = None
    if ring and num_workers > 0:
        # ring=True: the big feature tensors through a shared page-locked ring (module docstring).  slot_bytes: what the
        # largest prepared batch needs (default: probed on the first batch, + 25 %); slots = batches in flight + 3
        if slot_bytes is None:
            if not isinstance(batches, (list, tuple)):
                raise ValueError("prepared_loader(ring=True): pass slot_bytes (the largest batch's big tensors, "
                                 "HostPipeline.big_bytes) for a dataset that is not a list of collated batches")
            slot_bytes = max(pipeline.big_bytes(b) for b in batches) + 4096
        rg = PinnedRing(slot_bytes, prefetch_factor * num_workers + 3)
    # (collate_fn = identity: the default conversion of un-batched samples would turn the key tuples into lists)
    dl = torch.utils.data.DataLoader(_PreparedDataset(batches, pipeline, rg), batch_size=None, shuffle=False,
                                     num_workers=num_workers, pin_memory=pin_memory, collate_fn=_identity, **kw)
    return PreparedLoader(dl, rg) if rg is not None else dl
