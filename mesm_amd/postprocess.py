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


# ----------------------------------------------------------------------------- submission rows + NMS
def saliency_rows(saliency_scores, video_mask):
    """eval.py:66-71: the per-clip scores in fp16 (`.half()`), one list per query cut at its video length."""
    s16 = saliency_scores.detach().to(torch.float16)
    lens = video_mask.sum(1).cpu().tolist()
    s16 = s16.cpu()
    return [s16[j, :int(lens[j])].tolist() for j in range(len(lens))]


def compute_temporal_iou(pred, gt):
    """utils/temporal_nms.py:6-22: intersection over the HULL of the two windows (the reference's own
    "not the correct union")."""
    inter = max(0, min(pred[1], gt[1]) - max(pred[0], gt[0]))
    union = max(pred[1], gt[1]) - min(pred[0], gt[0])
    return 0 if union == 0 else 1.0 * inter / union


def temporal_nms(predictions, nms_thd, max_after_nms=100):
    """utils/temporal_nms.py:25-74: greedy NMS over [start, end, score] rows, best score first; a row is
    dropped when its IoU with a kept row exceeds nms_thd; at most max_after_nms rows survive."""
    if len(predictions) == 1:
        return predictions
    rest = sorted(predictions, key=lambda x: x[2], reverse=True)
    kept = []
    while len(rest) > 1 and len(kept) < max_after_nms:
        head = rest[0]
        rest = [head] + [r for r in rest[1:] if not compute_temporal_iou(head[:2], r[:2]) > nms_thd]
        kept.append(rest.pop(0))
    if len(kept) < max_after_nms and len(rest) >= 1:
        kept.append(rest.pop(0))
    return [[st, ed, s] for st, ed, s in kept]


def post_processing_mr_nms(mr_res, nms_thd, max_before_nms, max_after_nms):
    """eval.py:476-485."""
    out = []
    for e in mr_res:
        e["pred_relevant_windows"] = temporal_nms(e["pred_relevant_windows"][:max_before_nms], nms_thd=nms_thd,
                                                  max_after_nms=max_after_nms)
        out.append(e)
    return out


@torch.no_grad()
def compute_mr_results(model, eval_loader, opt, criterion=None, prepare=None):
    """eval.py:52-117: model.eval(), one forward per batch with is_training=False, submission rows
    (`pred_relevant_windows` ranked + clamped + rounded to clip_len multiples, `pred_saliency_scores`), and the
    weighted loss meters when a criterion is given.  `prepare(batch)` moves a host batch to the device
    (mesm_amd.batching.prepare_batch_input); None = the batches are already there."""
    model.eval()
    if criterion is not None:
        criterion.eval()
    sums, counts = {}, {}
    mr_res = []
    for batch in eval_loader:
        if prepare is not None:
            batch = prepare(batch)
        outputs = model(**batch, dataset_name=opt.dataset_name, is_training=False)
        rows = predict_windows(outputs["pred_logits"], outputs["pred_spans"], batch["duration"],
                               clip_len=opt.clip_len, max_ts_val=opt.max_ts_val,
                               sort_results=getattr(opt, "sort_results", True))
        sal = saliency_rows(outputs["saliency_scores"], batch["video_mask"])
        for idx in range(len(rows)):
            mr_res.append(dict(qid=batch["qid"][idx], query=batch["sentence"][idx], vid=batch["video_id"][idx],
                               pred_relevant_windows=rows[idx], pred_saliency_scores=sal[idx]))
        if criterion is not None:
            loss_dict, loss = criterion(outputs, batch, is_training=False)
            loss_dict = dict(loss_dict)
            loss_dict["loss_overall"] = float(loss)
            for k, v in loss_dict.items():
                w = criterion.weight_dict[k] if k in criterion.weight_dict else 1.0
                sums[k] = sums.get(k, 0.0) + float(v) * w
                counts[k] = counts.get(k, 0) + 1
    return mr_res, {k: sums[k] / counts[k] for k in sums}
