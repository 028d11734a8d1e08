"""Seeded inputs shared by tools/gen_golden_io.py (which feeds them to the REAL reference) and the tests (which feed
them to this build): per-group dataset samples for the collate functions, and a stub model / criterion / loader
for the inference-row functions.  Pure data generators: nothing here imports the reference."""
import torch


def group_samples(kind, seed):
    """Seeded stand-ins for Dataset.__getitem__ (base.py:164-223 / qvhighlights.py:96-139): one dict per video
    group, ragged video lengths, per-query lists."""
    g = torch.Generator().manual_seed(seed)
    groups = [2, 1, 3] if kind == "base" else [1, 2, 1, 1]
    Dv, Lw = 6, 5
    samples = []
    qid = 0
    for gi, n in enumerate(groups):
        L = 7 + 3 * gi
        e = {"num_clips": n, "video_id": "v%d" % gi if kind == "base" else ["v%d" % gi] * n,
             "duration": 10.0 + gi if kind == "base" else [10.0 + gi] * n}
        if kind == "base":
            e["video_feat"] = torch.randn(L, Dv, generator=g)
            e["moment"] = [[1.0 + q, 4.0 + q + gi] for q in range(n)]
            e["start_idx"] = [1 + q for q in range(n)]
            e["end_idx"] = [3 + q for q in range(n)]
        else:
            e["video_feat"] = [torch.randn(L - q, Dv, generator=g) for q in range(n)]  # QVH: one segment per query
            e["norm_moment"] = [torch.rand(1 + (q + gi) % 3, 2, generator=g).sort(-1)[0] for q in range(n)]
            e["norm_span"] = [torch.stack([m.sum(-1) / 2, m[:, 1] - m[:, 0]], -1) for m in e["norm_moment"]]
            e["saliency_label"] = [torch.randint(0, 13, (L - q,), generator=g).double() for q in range(n)]
        e["sentence"] = ["query %d" % (qid + q) for q in range(n)]
        e["words_id"] = [torch.randint(1, 50, (1, Lw), generator=g) * (torch.arange(Lw) < 2 + q)[None] for q in range(n)]
        e["words_weight"] = [torch.randint(1, 3, (1, Lw), generator=g) for _ in range(n)]
        e["unknown_mask"] = [torch.rand(1, Lw, generator=g) < 0.2 for _ in range(n)]
        e["words_label"] = [torch.randint(0, 50, (1, Lw), generator=g) for _ in range(n)]
        Lq = [L if kind == "base" else L - q for q in range(n)]
        e["clip_mask"] = [torch.arange(Lq[q]) < 3 + q for q in range(n)]
        e["pos_idx"] = [torch.tensor([0, 1 + q]) for q in range(n)]
        e["neg_idx"] = [torch.tensor([4, 5]) for q in range(n)]
        e["qid"] = [qid + q for q in range(n)]
        qid += n
        samples.append(e)
    return samples



class StubModel(torch.nn.Module):
    def __init__(self, outs):
        super().__init__()
        self.outs, self.i = outs, 0

    def forward(self, **kw):
        o = self.outs[self.i]
        self.i += 1
        return o


class StubCriterion(torch.nn.Module):
    weight_dict = {"loss_span": 10.0, "loss_giou": 1.0}

    def forward(self, outputs, batch, is_training=False):
        ls = outputs["pred_spans"].mean()
        lg = outputs["pred_logits"].abs().mean()
        return {"loss_span": ls, "loss_giou": lg, "class_error": ls * 0 + 3.0}, 10 * ls + lg



def mr_inputs(c):
    """-> (loader batches, model outputs per batch) for one mr_results case."""
    g = torch.Generator().manual_seed(c["seed"])
    loader, outs = [], []
    qid = 0
    for n in c["N"]:
        logits = torch.randn(n, c["Q"], 2, generator=g) * 2
        spans = torch.rand(n, c["Q"], 2, generator=g)
        spans[..., 1] = spans[..., 1] * 0.5 + 0.01
        sal = torch.randn(n, c["Lv"], generator=g) * 3
        vlen = torch.randint(c["Lv"] // 2, c["Lv"] + 1, (n,), generator=g)
        vmask = torch.arange(c["Lv"])[None] < vlen[:, None]
        duration = torch.rand(n, generator=g) * 100 + 40
        batch = {"video_mask": vmask, "duration": duration, "qid": list(range(qid, qid + n)),
                 "sentence": ["q%d" % i for i in range(qid, qid + n)],
                 "video_id": ["v%d" % i for i in range(qid, qid + n)]}
        qid += n
        loader.append(batch)
        outs.append({"pred_logits": logits, "pred_spans": spans, "saliency_scores": sal})
    return loader, outs


MR_CASES = {"qvh": dict(clip_len=2, N=[5, 3], Q=10, Lv=20, seed=51, nms_thd=0.7),
            "charades": dict(clip_len=1, N=[4], Q=10, Lv=12, seed=52, nms_thd=0.5),
            "tacos": dict(clip_len=-1, N=[3, 2], Q=6, Lv=30, seed=53, nms_thd=0.3)}
