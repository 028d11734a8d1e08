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
the device; big feature tensors come back through shared memory and are pinned by the DataLoader's pin thread."""
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
        return {"key": key, "spec": self.spec, "groups": groups, "n_real": n_real, "caps": caps,
                "arr": {k: torch.from_numpy(np.array(v, copy=True, order="C")) for k, v in arr.items()},
                "pmeta": pmeta, "tmeta": tmeta, "wm": wm, "big": big, "num_clips": batch["num_clips"],
                "words_weight": batch.get("words_weight"), "raw": raw if self.keep_raw else None}

    __call__ = prepare


class _PreparedDataset(torch.utils.data.Dataset):
    """dataset of COLLATED raw batches (what the reference's collate returns) -> prepared batches"""

    def __init__(self, batches, pipeline):
        self.batches, self.pipeline = batches, pipeline

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, i):
        return self.pipeline.prepare(self.batches[i])


def _identity(x):
    return x


def prepared_loader(batches, pipeline, num_workers=4, pin_memory=False, prefetch_factor=2, persistent=True):
    """DataLoader over a sequence (or map-style dataset) of collated raw host batches whose workers run
    `pipeline.prepare` (forked workers).  pin_memory: have the loader's pin thread copy every tensor into pinned memory
    -- measured on the bench box (tools/loader_path_probe.py) this costs more than it saves for 35 MB feature tensors
    (the pageable copy already runs on a copy stream into a double-buffered device staging area, GraphedStep._stage_big),
    hence off by default.  For a real dataset use
    `DataLoader(dataset, batch_size=..., collate_fn=lambda items: pipeline.prepare(collate(items)), ...)` the same way."""
    kw = {}
    if num_workers > 0:
        kw = dict(multiprocessing_context="fork", prefetch_factor=prefetch_factor, persistent_workers=persistent)
    # (collate_fn = identity: the default conversion of un-batched samples would turn the key tuples into lists)
    return torch.utils.data.DataLoader(_PreparedDataset(batches, pipeline), batch_size=None, shuffle=False,
                                       num_workers=num_workers, pin_memory=pin_memory, collate_fn=_identity, **kw)
