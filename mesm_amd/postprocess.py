"""Inference post-processing: model outputs -> the ranked `pred_relevant_windows` rows of the
reference's submission format (eval.py:63-92, utils/post_processing.py:22-47).

The per-query arithmetic (foreground softmax score, centre/width -> start/end, x duration) runs as
one kernel (`mesm_windows`); ranking and the reference's decimal formatting
(float(f"{e:.4f}"), then clamp / round-to-clip-length in float32 with half-to-even rounding) are
host work on N x Q x 3 numbers and are done in numpy so that the rows are identical to the
reference's, digit for digit.
"""
import numpy as np
import torch

from ._lib import check, lib, ptr, require_gpu, stream_ptr


def raw_windows(pred_logits, pred_spans, duration):
    """(N, Q, 2) logits, (N, Q, 2) spans (centre, width; normalised), (N,) durations -> (N, Q, 3)
    float32 [start, end, foreground score] on the device."""
    require_gpu(pred_logits, pred_spans, duration)
    lg = pred_logits.detach().float().contiguous()
    sp = pred_spans.detach().float().contiguous()
    du = duration.detach().float().contiguous()
    N, Q, _ = lg.shape
    out = torch.empty(N, Q, 3, device=lg.device, dtype=torch.float32)
    check(lib().mesm_windows(ptr(lg), ptr(sp), ptr(du), ptr(out), N, Q, stream_ptr()), "mesm_windows")
    return out


def _round4(a):
    """float(f"{e:.4f}") element-wise: decimal rounding of the float32 value widened to double."""
    flat = a.astype(np.float64).ravel()
    return np.array([float("%.4f" % e) for e in flat], dtype=np.float64).reshape(a.shape)


def predict_windows(pred_logits, pred_spans, duration, clip_len=2, max_ts_val=150, min_ts_val=0,
                    sort_results=True):
    """-> list (per query) of Q rows [start, end, score] exactly as eval.py + PostProcessorDETR emit."""
    raw = raw_windows(pred_logits, pred_spans, duration).cpu().numpy()  # float32
    out = []
    for rows in raw:
        if sort_results:
            order = sorted(range(rows.shape[0]), key=lambda i: float(rows[i, 2]), reverse=True)  # stable
            rows = rows[order]
        r4 = _round4(rows)
        w = np.clip(r4[:, :2].astype(np.float32), np.float32(min_ts_val), np.float32(max_ts_val))
        if clip_len != -1:
            w = np.round(w / np.float32(clip_len)) * np.float32(clip_len)  # half to even, float32 like torch.round
        sc = _round4(r4[:, 2].astype(np.float32))
        out.append([[float(w[i, 0]), float(w[i, 1]), float(sc[i])] for i in range(rows.shape[0])])
    return out
