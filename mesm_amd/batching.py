"""Batch assembly for the hot path: the reference's collate functions and
`prepare_batch_input`, producing the exact batch dict `MESM.forward` / `Criterion.forward` consume.

    collate            dataset/base.py:288-355        Charades-STA / TACoS samples (one video per group)
    collate_qvh        dataset/qvhighlights.py:214-284 QVHighlights samples (one segment per query)
    pad_sequences_1d   utils/data_utils.py:34-82       zero-padded (n, Lmax, ...) tensor + bool mask
    prepare_batch_input dataset/base.py:358-384        host -> device (words_weight stays on the host, Q4),
                                                      norm_moment / norm_span for tensor-target datasets

The reference pads with one Python-level copy per row (0.19 s of a 2 s CPU step); here a
padded tensor is one `torch.cat` of the rows and one scatter through a precomputed row index, and the
device copy moves every tensor of the batch in ONE transfer: all payloads are packed into a single
host staging buffer (16-byte aligned segments), copied once, and the batch tensors are views of the device
buffer.  Results are identical to the reference's (tests/test_batching_cpu.py compares with fixtures
produced by the real functions).
"""
import os

import torch


def pad_sequences_1d(sequences, dtype=torch.long, device=torch.device("cpu"), fixed_length=None):
    """-> (padded (n, Lmax, *extra), mask (n, Lmax) bool).  Only torch dtypes (the path never uses numpy ones)."""
    if isinstance(sequences[0], list):
        sequences = [torch.tensor(s, dtype=dtype, device=device) for s in sequences]
    extra = tuple(sequences[0].shape[1:])
    lengths = [len(s) for s in sequences]
    L = fixed_length if fixed_length is not None else max(lengths)
    n = len(sequences)
    padded = torch.zeros((n, L) + extra, dtype=dtype, device=device)
    lens = torch.tensor(lengths, device=device)
    mask = torch.arange(L, device=device)[None, :] < lens[:, None]
    if sum(lengths):
        flat = torch.cat([s.to(dtype) for s in sequences], dim=0)
        padded[mask] = flat  # row-major order of the True cells == concatenation order
    return padded, mask


def _common_tail(out, words_id, words_weight, unknown_mask, words_label):
    out["words_id"] = torch.cat(words_id, dim=0)
    if out["words_id"].ndim == 2:
        out["words_mask"] = out["words_id"] != 0
    elif out["words_id"].ndim == 3:
        out["words_mask"] = None
    else:
        raise ValueError(f"words_id has shape {out['words_id'].shape}")
    out["words_weight"] = torch.cat(words_weight, dim=0)
    if words_label[0] is not None:
        out["unknown_mask"] = torch.cat(unknown_mask, dim=0)
        out["words_label"] = torch.cat(words_label, dim=0)


def collate(batch):
    """dataset/base.py:288-355: the group's video is repeated once per query."""
    out = {}
    num_clips, video_feat, video_id, duration = [], [], [], []
    moment, sentence, words_id, words_weight, unknown_mask, words_label = [], [], [], [], [], []
    start_idx, end_idx, clip_mask, pos_idx, neg_idx, qid = [], [], [], [], [], []
    for e in batch:
        n = e["num_clips"]
        num_clips.append(n)
        video_feat += [e["video_feat"]] * n
        video_id += [e["video_id"]] * n
        duration += [e["duration"]] * n
        moment += e["moment"]
        sentence += e["sentence"]
        words_id += e["words_id"]
        words_weight += e["words_weight"]
        unknown_mask += e["unknown_mask"]
        words_label += e["words_label"]
        start_idx += e["start_idx"]
        end_idx += e["end_idx"]
        clip_mask += e["clip_mask"]
        pos_idx += e["pos_idx"]
        neg_idx += e["neg_idx"]
        qid += e["qid"]
    out["num_clips"] = torch.LongTensor(num_clips)
    out["video_feat"], out["video_mask"] = pad_sequences_1d(video_feat, dtype=video_feat[0].dtype)
    out["duration"] = torch.Tensor(duration)
    out["moment"] = torch.Tensor(moment)
    _common_tail(out, words_id, words_weight, unknown_mask, words_label)
    out["start_idx"] = torch.LongTensor(start_idx)
    out["end_idx"] = torch.LongTensor(end_idx)
    out["clip_mask"], _ = pad_sequences_1d(clip_mask, dtype=clip_mask[0].dtype)
    if pos_idx[0] is not None:
        out["pos_idx"] = torch.stack(pos_idx, dim=0)
        out["neg_idx"] = torch.stack(neg_idx, dim=0)
    out["qid"], out["video_id"], out["sentence"] = qid, video_id, sentence
    return out


def collate_qvh(batch):
    """dataset/qvhighlights.py:214-284: per-query segments, list-of-dict targets, float64 saliency labels."""
    out = {}
    num_clips, video_feat, video_id, duration = [], [], [], []
    norm_moment, norm_span, sentence, words_id, words_weight, unknown_mask, words_label = [], [], [], [], [], [], []
    saliency_label, clip_mask, pos_idx, neg_idx, qid = [], [], [], [], []
    for e in batch:
        num_clips.append(e["num_clips"])
        video_feat += e["video_feat"]
        video_id += e["video_id"]
        duration += e["duration"]
        sentence += e["sentence"]
        words_id += e["words_id"]
        words_weight += e["words_weight"]
        unknown_mask += e["unknown_mask"]
        words_label += e["words_label"]
        qid += e["qid"]
        if "norm_moment" in e:
            norm_moment += e["norm_moment"]
            norm_span += e["norm_span"]
            saliency_label += e["saliency_label"]
            clip_mask += e["clip_mask"]
            pos_idx += e["pos_idx"]
            neg_idx += e["neg_idx"]
    out["num_clips"] = torch.LongTensor(num_clips)
    out["video_feat"], out["video_mask"] = pad_sequences_1d(video_feat, dtype=video_feat[0].dtype)
    out["duration"] = torch.Tensor(duration)
    _common_tail(out, words_id, words_weight, unknown_mask, words_label)
    if len(norm_moment) > 0:
        out["norm_moment"] = [dict(moments=m) for m in norm_moment]
        out["norm_span"] = [dict(spans=s) for s in norm_span]
        out["saliency_label"], _ = pad_sequences_1d(saliency_label, dtype=saliency_label[0].dtype)
        out["clip_mask"], _ = pad_sequences_1d(clip_mask, dtype=clip_mask[0].dtype)
        if pos_idx[0] is not None:
            out["pos_idx"] = torch.stack(pos_idx, dim=0)
            out["neg_idx"] = torch.stack(neg_idx, dim=0)
    out["qid"], out["video_id"], out["sentence"] = qid, video_id, sentence
    return out


HOST_SIDE_BIG = 1 << 20  # (= graphed.GraphedStep.BIG: tensors above 1 MiB are features, not plan inputs)


def attach_host_side(batch):
    """Keep the HOST copies of everything small in a collated batch under `batch["_host"]` (call it at the end of the collate, in
    the loader worker: `collate_fn=lambda items: attach_host_side(collate_qvh(items))`).  `prepare_batch_input` -- the reference's
    (dataset/base.py:358-384) or this module's -- moves the batch's tensors to the device and drops the host tensors; the step's
    index plans, flattened targets and host-RNG draws are host arithmetic on the masks, labels and target windows, so a model
    driven through the unchanged `model(**batch)` call (mesm_amd/autograph.py) otherwise has to fetch them back with a device
    synchronisation in front of every forward.  `_host` is a plain dict: prepare_batch_input leaves it alone and
    `model(**batch)` hands it through; with it the forward launch needs no synchronisation and the host runs ahead of the device.
    Contents: every tensor of at most 1 MiB, the per-pair target lists, and -- for pre-extracted word features -- the two forms of
    the word-validity mask (hostplan.HostSpec.words_mask with and without the text normalisation)."""
    import numpy as np
    host = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.numel() * v.element_size() <= HOST_SIDE_BIG:
            host[k] = v
        elif isinstance(v, list) and v and isinstance(v[0], dict):
            host[k] = [dict(d) for d in v]
    w = batch.get("words_id")
    if torch.is_tensor(w) and w.dim() == 3:
        a = w.numpy()
        n = np.maximum(np.sqrt((a * a).sum(-1, keepdims=True)), 1e-5)
        host["_words_mask_norm"] = torch.from_numpy((a / n).sum(-1) != 0)
        host["_words_mask_raw"] = torch.from_numpy(a.sum(-1) != 0)
    if "moment" in batch and "norm_span" not in batch:  # what prepare_batch_input derives on the device (base.py:380-384)
        host["norm_moment"] = batch["moment"] / batch["duration"].unsqueeze(1)
        host["norm_span"] = span_xx_to_cxw(host["norm_moment"])
    batch["_host"] = host
    return batch


def span_xx_to_cxw(xx):
    return torch.stack([xx.sum(-1) * 0.5, xx[..., 1] - xx[..., 0]], dim=-1)


class _Packer:
    """All tensors of a batch in ONE host -> device transfer."""

    def __init__(self):
        self.items, self.off = [], 0

    def add(self, t):
        t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        self.items.append((t, self.off, nbytes))
        self.off += (nbytes + 15) // 16 * 16
        return len(self.items) - 1

    def ship(self, device, non_blocking):
        host = torch.empty(max(self.off, 16), dtype=torch.uint8, pin_memory=False)
        for t, off, nbytes in self.items:
            if nbytes:
                host[off:off + nbytes] = t.reshape(-1).view(torch.uint8)
        dev = host.to(device, non_blocking=non_blocking)
        return [dev[off:off + nbytes].view(t.dtype).view(t.shape) for t, off, nbytes in self.items]


# prepare_batch_input keeps the host side of a host batch it sends to a GPU (attach_host_side) unless MESM_KEEP_HOST_SIDE=0
_KEEP_HOST_SIDE = os.environ.get("MESM_KEEP_HOST_SIDE", "1") != "0"


def prepare_batch_input(batched_data, device, non_blocking=False):
    """dataset/base.py:358-384 (mutates and returns `batched_data`).  One addition: a HOST batch on its way to a GPU keeps the
    host copies of its small tensors under `batched_data["_host"]` (attach_host_side: they are in hand here, and the step's
    host arithmetic needs them -- without them the unchanged `model(**batch)` call has to fetch them back behind a device
    synchronisation, ~0.5 ms per step); a batch whose collate attached them already is left alone."""
    device = torch.device(device)
    if (_KEEP_HOST_SIDE and device.type == "cuda" and "_host" not in batched_data
            and all(not v.is_cuda for v in batched_data.values() if isinstance(v, torch.Tensor))):
        attach_host_side(batched_data)
    big = [v for k, v in batched_data.items() if isinstance(v, torch.Tensor) and k != "words_weight" and v.numel() * v.element_size() > (1 << 20)]
    if device.type == "cpu":
        moved = {k: v for k, v in batched_data.items()}
    elif big and all(v.is_pinned() for v in big):
        # a DataLoader(pin_memory=True) batch: every tensor is page-locked already -- asynchronous copies straight from where
        # they are, like the reference does (packing 35 MB of features into a staging buffer first costs more than it saves)
        moved = dict(batched_data)
        for k, v in batched_data.items():
            if k == "words_weight":
                continue
            if isinstance(v, torch.Tensor):
                moved[k] = v.to(device, non_blocking=non_blocking)
            elif k in ("norm_moment", "norm_span"):
                f = "moments" if k == "norm_moment" else "spans"
                moved[k] = [{f: e[f].to(device, non_blocking=non_blocking)} for e in v]
    else:
        pk, slots = _Packer(), {}
        for k, v in batched_data.items():
            if k == "words_weight":
                continue
            if isinstance(v, torch.Tensor):
                slots[k] = pk.add(v)
            elif k in ("norm_moment", "norm_span"):
                f = "moments" if k == "norm_moment" else "spans"
                slots[k] = [pk.add(e[f]) for e in v]
        dev = pk.ship(device, non_blocking)
        moved = dict(batched_data)
        for k, s in slots.items():
            if isinstance(s, list):
                f = "moments" if k == "norm_moment" else "spans"
                moved[k] = [{f: dev[i]} for i in s]
            else:
                moved[k] = dev[s]
    batched_data.update(moved)
    if "moment" in batched_data and "norm_span" not in batched_data:
        moment, duration = batched_data["moment"], batched_data["duration"]
        batched_data["norm_moment"] = moment / duration.unsqueeze(1)
        batched_data["norm_span"] = span_xx_to_cxw(batched_data["norm_moment"])
    return batched_data


def pad_pairs(batch, P, big=1 << 20):
    """The batch with its pair axis padded to P pairs: copies of pair 0 (finite everywhere, valid masks), each a video
    group of its own, appended BEHIND the real pairs; `_n_real` = the number of real pairs.  A step captured for P pairs
    (graphed.StepCache(pairs=...)) replays such a batch with the real count in device memory: padding pairs take no
    part in any loss, mean, negative draw or mask wrap and get zero gradients, so the step equals the unpadded one.
    Works on host or device tensors; per-pair lists (QVHighlights targets) are extended the same way."""
    N = batch["video_feat"].shape[0]
    if P < N:
        raise ValueError("pad_pairs: %d pairs do not fit %d" % (N, P))
    out = dict(batch)
    out["_n_real"] = N
    k = P - N
    if k == 0:
        return out
    for key, v in batch.items():
        if key == "num_clips":
            out[key] = torch.cat([v, torch.ones(k, dtype=v.dtype, device=v.device)])
        elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == N and not v.is_cuda \
                and v.numel() * v.element_size() > big and key in ("video_feat", "words_id"):
            # big host feature tensors keep their N rows: graphed.GraphedStep copies them into the first N rows of its
            # static input, and whatever finite rows an earlier batch left behind serve as the padding pairs
            continue
        elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == N:
            out[key] = torch.cat([v, v[:1].expand(k, *v.shape[1:])])
        elif isinstance(v, (list, tuple)) and len(v) == N:
            out[key] = list(v) + [v[0]] * k
    return out
