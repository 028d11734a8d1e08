"""CPU oracle for the inference post-processing — TEST INFRASTRUCTURE, NOT A PRODUCT PATH.

Restates, in plain PyTorch-CPU / Python, how the reference turns model outputs into the ranked
`pred_relevant_windows` rows:

    /root/reference/eval.py:63-92                 foreground softmax score, span_cxw_to_xx * duration,
                                                  sort by score (descending, stable), float(f"{e:.4f}")
    /root/reference/utils/post_processing.py:22-47  clamp to [min_ts, max_ts]; round to multiples of
                                                  clip_len (torch.round: half to even, float32);
                                                  score to 4 decimals
    /root/reference/utils/span_utils.py:26-42     span_cxw_to_xx

Pinned by tests/golden/mr_results.json: rows the real eval.compute_mr_results + PostProcessorDETR produced
when driven with a stub model (tools/gen_golden_io.py).
"""
import torch
import torch.nn.functional as F


def windows(pred_logits, pred_spans, duration, clip_len=2, max_ts_val=150, min_ts_val=0, sort_results=True):
    prob = F.softmax(pred_logits.float().cpu(), -1)
    scores = prob[..., 0]
    spans = pred_spans.float().cpu()
    duration = duration.float().cpu()
    out = []
    for idx in range(spans.shape[0]):
        sp = spans[idx]
        xx = torch.stack([sp[..., 0] - 0.5 * sp[..., 1], sp[..., 0] + 0.5 * sp[..., 1]], dim=-1) * duration[idx]
        rows = torch.cat([xx, scores[idx][:, None]], dim=1).tolist()
        if sort_results:
            rows = sorted(rows, key=lambda x: x[2], reverse=True)
        rows = [[float(f"{e:.4f}") for e in row] for row in rows]
        t = torch.tensor(rows)
        w = torch.clamp(t[:, :2], min=min_ts_val, max=max_ts_val)
        if clip_len != -1:
            w = torch.round(w / clip_len) * clip_len
        rows = torch.cat([w, t[:, 2:3]], dim=1).tolist()
        out.append([e[:2] + [float(f"{e[2]:.4f}")] for e in rows])
    return out
