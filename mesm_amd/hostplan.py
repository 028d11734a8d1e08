"""The HOST half of a training step's batch handling, without the model: everything `graphed.GraphedStep.load_batch`
computes on the CPU for a new batch -- the forward's index plan (model.py:184-207, 260, 307-325 of the reference), the
criterion's flattened targets, the word-validity mask, the capture-time capacities -- as plain functions of a small,
picklable `HostSpec`, so that the same code runs in the training process (GraphedStep) and in loader WORKER processes
(mesm_amd/loader.py: the reference hides its collate behind `DataLoader(num_workers=8)`, /root/reference
dataset/base.py:288-384 + train.py; here the plan / target arrays and the pair padding are part of what a worker
returns, and the training process only uploads and replays).

The two host-RNG draws of the reference's forward (negative query index, masked-LM word choice: model.py:260,
361-384) are NOT made here: they stay in the training process, in the reference's RNG stream (`GraphedStep.redraw`)."""
import numpy as np
import torch


def _round_up(x, m):
    return (x + m - 1) // m * m


class HostSpec:
    """what the host-side plans need to know about the model / criterion (picklable: plain Python values)"""
    FIELDS = ("rec_fw", "rec_ss", "max_words_l", "normalize_txt", "num_queries", "dataset_name", "multi_clip", "gamma")

    def __init__(self, **kw):
        for k in self.FIELDS:
            setattr(self, k, kw[k])

    @classmethod
    def from_model(cls, model, criterion, dataset_name):
        return cls(rec_fw=bool(model.rec_fw), rec_ss=bool(model.rec_ss), max_words_l=int(model.max_words_l),
                   normalize_txt=bool(model.normalize_txt), num_queries=int(model.num_queries), dataset_name=dataset_name,
                   multi_clip=bool(criterion.multi_clip), gamma=float(criterion.gamma))

    @classmethod
    def from_args(cls, args):
        """from the reference's option namespace (runner.py:255-345): usable before / without building the model"""
        return cls(rec_fw=bool(args.rec_fw), rec_ss=bool(args.rec_ss), max_words_l=int(args.max_words_l),
                   normalize_txt=bool(getattr(args, "normalize_txt", True)), num_queries=int(args.num_queries),
                   dataset_name=args.dataset_name, multi_clip=args.dataset_name in ["qvhighlights"],
                   gamma=float(args.iou_gamma))

    def __eq__(self, other):
        return isinstance(other, HostSpec) and all(getattr(self, k) == getattr(other, k) for k in self.FIELDS)

    # ---- the forward's plan: MESM.plan_arrays reads only these two switches of `self` when the draws are handed in
    def plan_arrays(self, *a, **kw):
        from .model import MESM
        return MESM.plan_arrays(self, *a, **kw)

    # ---- word validity on the host
    def words_mask(self, host):
        """the collate mask for token ids (cut like model.py:114-116), the non-zero rows of pre-extracted features
        (post_process_text, model.py:145-152)"""
        if host.get("_words_mask") is not None:  # formed on the device with the arithmetic below (autograph._Fetch)
            return host["_words_mask"]
        w = host["words_id"]
        if w.dim() != 3:
            return host["words_mask"][:, :self.max_words_l]
        # numpy on purpose: a multi-threaded torch CPU reduction stalls ~17 ms next to a busy HIP queue on this
        # platform (measured, tools/load_batch_probe.py); these are 2 MB
        a = w.numpy()
        if self.normalize_txt:
            n = np.maximum(np.sqrt((a * a).sum(-1, keepdims=True)), 1e-5)
            a = a / n
        return torch.from_numpy(a.sum(-1) != 0)

    # ---- capture-time capacities
    def resolve_caps(self, caps, batch, groups, group_cap=None):
        """caps: None = exact extents of `batch`; "auto" = bucketed capacities so that other batches of the same
        (N, Lv, Lw, groups) replay; or a dict with any of Lc / Lss / M / T / Tmax.  group_cap (with "auto"): room for
        video groups of up to that many queries and GT-clip runs of any length -- the capacities then depend on the
        batch's SHAPE and group_cap only, which is what lets a loader worker compute them for a graph it never saw."""
        if caps is None:
            return {}
        if caps == "auto":
            caps = {}
            N, Lv = batch["video_mask"].shape
            if self.rec_fw:
                caps["Lc"] = Lv if group_cap else min(Lv, _round_up(int(batch["clip_mask"].sum(1).max()), 8))
            gmax = max(max(groups), group_cap or 0)
            if self.rec_ss:
                caps["M"] = gmax  # sentence slots per pair (SS branch): other groupings with <= M fit
            if self.rec_ss and self.dataset_name == "qvhighlights":
                vm = batch["video_mask"].cpu()
                lens = [int(c.sum()) for c in torch.split(vm, groups)]
                full = all(g == 1 for g in groups) and bool(vm.all()) and not group_cap
                # 64 = one key tile of the attention kernels; a group cannot hold more than its pairs' clips
                caps["Lss"] = Lv if full else (gmax * Lv if group_cap else min(_round_up(max(lens), 64), gmax * Lv))
            if self.multi_clip:
                Q = self.num_queries
                tmax = max(len(t["spans"]) for t in batch["norm_span"])
                caps["Tmax"] = max(tmax, min(5, Q))  # QVHighlights keeps <= 5 windows (qvhighlights.py:148-150)
                caps["T"] = N * caps["Tmax"]
        return dict(caps)

    # ---- everything small the captured step reads
    def target_arrays(self, host, caps):
        """the criterion's half: {"t." + name: array}, meta (TargetPlan.arrays)"""
        from .criterion import TargetPlan
        tarr, tmeta = TargetPlan.arrays(host, self.multi_clip, self.gamma, T_cap=caps.get("T"), Tmax_cap=caps.get("Tmax"))
        return {"t." + k: v for k, v in tarr.items()}, tmeta

    def host_arrays(self, host, groups, caps, n_real, neg_index, masked_words, big, targets=True):
        """{name: numpy array}: "p." the forward's plan, "t." the criterion's target plan, "b." batch tensors of at
        most `big` bytes; plus the two metas and the word mask.  neg_index / masked_words None: drawn here (the caller
        is the training process); workers pass placeholders and the training process draws (GraphedStep.redraw)."""
        wm = self.words_mask(host)
        P = host["video_mask"].shape[0]
        if wm.shape[0] < P:  # a big feature tensor that arrived with its real rows only (batching.pad_pairs)
            wm = torch.cat([wm, wm[:1].expand(P - wm.shape[0], wm.shape[1])])
        parr, pmeta = self.plan_arrays(host["video_mask"].numpy(), wm.numpy(), groups, self.dataset_name, True,
                                       clip_mask=host["clip_mask"].numpy() if "clip_mask" in host else None,
                                       neg_index=neg_index, masked_words=masked_words,
                                       words_weight=host.get("words_weight"), Lc_cap=caps.get("Lc"),
                                       Lss_cap=caps.get("Lss"), M_cap=caps.get("M"), n_valid=n_real)
        arr = {"p." + k: v for k, v in parr.items()}
        for k, v in host.items():
            # (num_clips has one entry per GROUP and is only read on the host: the step reads the plans)
            if torch.is_tensor(v) and k not in ("words_weight", "num_clips") and not k.startswith("_") and not v.is_cuda \
                    and v.numel() * v.element_size() <= big:
                arr["b." + k] = v.numpy()
        # the criterion's arrays LAST: they sit together at the end of the arena, so a caller that replays the forward
        # before the criterion (autograph.py) can build and upload them while the forward graph already runs
        # (targets=False: the caller does that itself, target_arrays)
        tmeta = None
        if targets:
            tarr, tmeta = self.target_arrays(host, caps)
            arr.update(tarr)
        return arr, pmeta, tmeta, wm
