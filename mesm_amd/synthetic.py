"""Synthetic batches in the shape of the reference's collate output.

The batch dict is what ``prepare_batch_input`` (dataset/base.py:358-384) hands to
``model(**batch, dataset_name=..., is_training=True)`` and to ``criterion(outputs, batch)``:
features are seeded N(0,1), lengths / ground-truth runs / labels follow closed-form
patterns so that every run of bench.py and every test sees the same data.
"""
import argparse

import torch

# Model / loss hyper-parameters shared by all shipped configs (config/*/*.json) and the
# per-dataset differences (SURVEY.md §5 "Config / flags").
BASE_ARGS = dict(
    hidden_dim=256, dropout=0.1, nheads=8, dim_feedforward=1024, num_recfw_layers=2,
    t2v_layers=2, enc_layers=2, dec_layers=2, pre_norm=False, position_embedding="sine",
    input_dropout=0.5, num_queries=10, use_txt_pos=False, n_input_proj=2, rec_fw=True,
    rec_ss=True, num_recss_layers=4, set_cost_span=10, set_cost_giou=1, span_loss_type="l1",
    aux_loss=True, saliency_margin=0.2, loss_span_coef=10, loss_giou_coef=1,
    loss_saliency_coef=1, eos_coef=0.1, iou_gamma=0.9, recss_tau=0.5, normalize_txt=True,
    tokenizer_type="GloVeNLTK", load_vocab_pkl=True, text_model_path=None, device="cpu",
)

WORKLOADS = {
    # name: (dataset_name, N, groups, Lv, Lw, Dv, Dt, vocab_size, extra args)
    "C1": dict(dataset_name="charades", groups=[1, 1], Lv=75, Lw=16, v_feat_dim=4098, t_feat_dim=300,
               vocab_size=1111, share_MLP=True, set_cost_class=4, loss_label_coef=4, rank_coef=1,
               use_triplet=False, loss_recfw_coef=0.1, loss_recss_coef=0.1, max_video_l=75,
               max_words_l=16),
    "C2": dict(dataset_name="charades", groups=[2] * 16, Lv=75, Lw=16, v_feat_dim=2818, t_feat_dim=512,
               vocab_size=1113, share_MLP=True, set_cost_class=4, loss_label_coef=4, rank_coef=1,
               use_triplet=False, loss_recfw_coef=0.1, loss_recss_coef=0.1, max_video_l=75,
               max_words_l=16),
    "C3a": dict(dataset_name="qvhighlights", groups=[1] * 32, Lv=75, Lw=32, v_feat_dim=2818,
                t_feat_dim=512, vocab_size=5002, share_MLP=True, set_cost_class=4, loss_label_coef=4,
                rank_coef=12, use_triplet=True, loss_recfw_coef=0.5, loss_recss_coef=0.1,
                max_video_l=75, max_words_l=32),
    "C3b": dict(dataset_name="qvhighlights", groups=[4] * 8, Lv=75, Lw=32, v_feat_dim=2818,
                t_feat_dim=512, vocab_size=5002, share_MLP=True, set_cost_class=4, loss_label_coef=4,
                rank_coef=12, use_triplet=True, loss_recfw_coef=0.5, loss_recss_coef=0.1,
                max_video_l=75, max_words_l=32),
    "C5": dict(dataset_name="tacos", groups=[8, 8], Lv=512, Lw=16, v_feat_dim=4098, t_feat_dim=300,
               vocab_size=1111, share_MLP=False, set_cost_class=6, loss_label_coef=6, rank_coef=1,
               use_triplet=True, loss_recfw_coef=0.1, loss_recss_coef=0.1, max_video_l=600,
               max_words_l=16),
}


def make_args(workload=None, **overrides):
    """argparse.Namespace with the fields runner.build_model / build_criterion read."""
    d = dict(BASE_ARGS)
    if workload is not None:
        w = dict(WORKLOADS[workload])
        for k in ("groups", "Lv", "Lw"):
            w.pop(k)
        d.update(w)
    d.update(overrides)
    return argparse.Namespace(**d)


def make_batch(dataset_name, groups, Lv, Lw, Dv, Dt, num_classes, seed=0, ragged=False,
               duration=150.0):
    """Batch dict on the CPU.  `groups` = queries per video group (sum = N pairs)."""
    g = torch.Generator().manual_seed(seed)
    N = sum(groups)
    qvh = dataset_name == "qvhighlights"
    # video lengths
    if ragged:
        lmin = max(8, Lv * 8 // 15)  # 40 of 75 clips
        vlen = [lmin + ((13 * i) % (Lv - lmin + 1)) for i in range(N)]
        vlen[0] = Lv
    else:
        vlen = [Lv] * N
    video_feat = torch.randn(N, Lv, Dv, generator=g)
    if not qvh:
        # the rows of a group share one video (and therefore its length)
        start = 0
        for gs in groups:
            video_feat[start:start + gs] = video_feat[start]
            for i in range(start, start + gs):
                vlen[i] = vlen[start]
            start += gs
    video_mask = torch.arange(Lv)[None, :] < torch.tensor(vlen)[:, None]
    video_feat = video_feat * video_mask[..., None]
    # words
    wlen = [min(Lw, 4 + ((5 * i) % (Lw - 3))) for i in range(N)]
    words_mask = torch.arange(Lw)[None, :] < torch.tensor(wlen)[:, None]
    words_feat = torch.randn(N, Lw, Dt, generator=g) * words_mask[..., None]
    # zero beyond the sentence: _mask_words L1-normalises the whole row (model.py:369)
    words_weight = (1 + (torch.arange(N)[:, None] + torch.arange(Lw)[None, :]) % 2).long() * words_mask
    words_label = torch.randint(0, num_classes, (N, Lw), generator=g)
    unknown_mask = ((torch.arange(N)[:, None] * 3 + torch.arange(Lw)[None, :]) % 11 == 0) & words_mask
    # ground-truth clip run
    clip_mask = torch.zeros(N, Lv, dtype=torch.bool)
    sal = torch.zeros(N, Lv, dtype=torch.float64)
    pos_idx = torch.zeros(N, 2, dtype=torch.int64)
    neg_idx = torch.zeros(N, 2, dtype=torch.int64)
    moments, windows = [], []
    for i in range(N):
        L = vlen[i]
        k = min(4 + ((7 * i) % 17), L - 6)
        s = (11 * i) % (L - k)
        clip_mask[i, s:s + k] = True
        sal[i, s:s + k] = torch.tensor([1.0 + ((3 * j + i) % 12) for j in range(s, s + k)],
                                       dtype=torch.float64)
        pos_idx[i] = torch.tensor([s, s + k - 1])
        outside = [j for j in range(L) if j < s or j >= s + k]
        neg_idx[i] = torch.tensor([outside[0], outside[-1]])
        clip_len = duration / Lv
        main = [s * clip_len, (s + k) * clip_len]
        moments.append(main)
        win = [main]
        for e in range(i % 3):  # extra 2-clip windows outside the main run
            o = outside[(e + 1) * (len(outside) - 2) // 3]  # distinct windows: no tied targets
            win.append([o * clip_len, (o + 2) * clip_len])
        windows.append(win)
    batch = {
        "video_feat": video_feat, "video_mask": video_mask, "words_id": words_feat,
        "words_mask": None, "words_weight": words_weight, "num_clips": torch.tensor(groups),
        "unknown_mask": unknown_mask, "clip_mask": clip_mask, "words_label": words_label,
        "pos_idx": pos_idx, "neg_idx": neg_idx,
        "duration": torch.full((N,), duration),
    }
    if qvh:
        batch["saliency_label"] = sal
        nm = [torch.tensor(w, dtype=torch.float32) / duration for w in windows]
        batch["norm_moment"] = [{"moments": m} for m in nm]
        batch["norm_span"] = [{"spans": torch.stack([m.sum(-1) * 0.5, m[:, 1] - m[:, 0]], -1)}
                              for m in nm]
    else:
        mom = torch.tensor(moments, dtype=torch.float32)
        nm = mom / duration
        batch["moment"] = mom
        batch["norm_moment"] = nm
        batch["norm_span"] = torch.stack([nm.sum(-1) * 0.5, nm[:, 1] - nm[:, 0]], -1)
    return batch


def workload_batch(name, seed=0, ragged=False):
    w = WORKLOADS[name]
    ncls = w["vocab_size"] + 1
    return make_batch(w["dataset_name"], w["groups"], w["Lv"], w["Lw"], w["v_feat_dim"],
                      w["t_feat_dim"], ncls, seed=seed, ragged=ragged)


def host_draws(batch, seed=0):
    """The two host-RNG draws of the reference, as replayable tensors:
    neg_index (sample_outclass_neg, utils/data_utils.py:113-124: uniform over the queries of
    OTHER groups) and masked_words (_mask_words, model.py:361-384: max(l//3,1) positions per
    pair without replacement, p ~ words_weight; pairs with <=1 word skipped)."""
    g = torch.Generator().manual_seed(1000 + seed)
    groups = batch["num_clips"].tolist()
    N = sum(groups)
    if len(groups) < 2:
        raise IndexError("a batch needs >= 2 video groups (negatives come from other groups)")
    neg = torch.empty(N, dtype=torch.int64)
    start = 0
    for gs in groups:
        cand = torch.tensor([j for j in range(N) if j < start or j >= start + gs])
        for i in range(start, start + gs):
            neg[i] = cand[torch.randint(0, len(cand), (1,), generator=g)]
        start += gs
    words = batch["words_id"]
    wmask = words.abs().sum(-1) != 0 if words.dim() == 3 else batch["words_mask"]
    masked = torch.zeros_like(wmask)
    weight = batch["words_weight"].float()
    for i in range(N):
        l = int(wmask[i].sum())
        if l <= 1:
            continue
        k = max(l // 3, 1)
        p = weight[i, :l] / weight[i, :l].sum()
        masked[i, torch.multinomial(p, k, replacement=False, generator=g)] = True
    return neg, masked


def to_device(batch, device):
    """prepare_batch_input semantics: everything to `device` except words_weight (stays on the
    CPU, dataset/base.py:360-361)."""
    out = {}
    for k, v in batch.items():
        if k == "words_weight" or v is None:
            out[k] = v
        elif isinstance(v, torch.Tensor):
            out[k] = v.to(device)
        elif isinstance(v, list):
            out[k] = [{kk: vv.to(device) for kk, vv in d.items()} for d in v]
        else:
            out[k] = v
    return out


def with_clip_tokens(batch, vocab, ctx=77, seed=0):
    """The same batch in the form the CLIP tokenizer path collates (dataset/base.py:326-355 with
    tokenizer_type 'CLIP'): words_id (N, ctx) int64 token ids, words_mask (N, ctx) bool.  Sentence i has
    len_i + (3 if i odd) tokens, so some run past max_words_l and are cut by CLIP_encode_text
    (model.py:114-116); the per-word tensors (weight, unknown mask, labels) keep their (N, Lw) shape."""
    g = torch.Generator().manual_seed(4000 + seed)
    wm = batch["words_id"].abs().sum(-1) != 0
    N, Lw = wm.shape
    tl = wm.sum(1) + torch.tensor([3 if i % 2 else 0 for i in range(N)])
    mask = torch.arange(ctx)[None, :] < tl[:, None]
    ids = torch.randint(1, vocab - 1, (N, ctx), generator=g) * mask
    ids[torch.arange(N), tl - 1] = vocab - 1  # EOT = the highest id
    cut = mask[:, :Lw]
    out = dict(batch)
    out["words_id"], out["words_mask"] = ids, mask
    out["words_weight"] = (1 + (torch.arange(N)[:, None] + torch.arange(Lw)[None, :]) % 2).long() * cut
    out["unknown_mask"] = batch["unknown_mask"] & cut
    return out
