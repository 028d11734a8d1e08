"""Golden vectors for the inference post-processing: the REAL reference code
(/root/reference, build container only; imported, never copied) turns seeded (logits, spans, duration)
into the ranked `pred_relevant_windows` rows a user sees:
    eval.py:63-92         softmax score of the foreground class, span_cxw_to_xx * duration, sort by
                          score, 4-decimal rounding through float(f"{e:.4f}")
    post_processing.py:22-47   clamp to [0, max_ts_val], round to multiples of clip_len, score to 4 decimals
    python tools/gen_golden_windows.py   ->  tests/golden/windows.npz
"""
import os, sys, types
import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
for name in ("ftfy", "nltk", "h5py"):
    sys.modules.setdefault(name, types.ModuleType(name))
import tqdm  # noqa: E402
tqdm.tqdm = lambda x, **k: x
import utils.post_processing as pp  # noqa: E402
pp.tqdm = lambda x, **k: x
from utils.span_utils import span_cxw_to_xx  # noqa: E402

out = {}
cases = {"qvh": dict(N=32, Q=10, clip_len=2, max_ts=150, seed=1),
         "charades": dict(N=7, Q=10, clip_len=1, max_ts=150, seed=2),
         "noclip": dict(N=5, Q=4, clip_len=-1, max_ts=150, seed=3)}
for name, c in cases.items():
    g = torch.Generator().manual_seed(c["seed"])
    logits = torch.randn(c["N"], c["Q"], 2, generator=g) * 2
    spans = torch.rand(c["N"], c["Q"], 2, generator=g)
    spans[..., 1] = spans[..., 1] * 0.6 + 0.01  # widths; some windows stick out of [0, duration]
    duration = torch.rand(c["N"], generator=g) * 140 + 10
    duration[0] = 150.0
    # eval.py:63-92
    prob = F.softmax(logits, -1)
    scores = prob[..., 0]
    res = []
    for idx, (sp, sc) in enumerate(zip(spans, scores)):
        sp = span_cxw_to_xx(sp) * duration[idx]
        rows = torch.cat([sp, sc[:, None]], dim=1).cpu().tolist()
        rows = sorted(rows, key=lambda x: x[2], reverse=True)
        rows = [[float(f"{e:.4f}") for e in row] for row in rows]
        res.append(dict(pred_relevant_windows=rows))
    post = pp.PostProcessorDETR(clip_length=c["clip_len"], min_ts_val=0, max_ts_val=c["max_ts"], min_w_l=2,
                                max_w_l=150, move_window_method="left",
                                process_func_names=("clip_ts", "round_multiple") if c["clip_len"] != -1 else ("clip_ts",))
    res = post(res)
    out[name + ".logits"] = logits.numpy()
    out[name + ".spans"] = spans.numpy()
    out[name + ".duration"] = duration.numpy()
    out[name + ".cfg"] = np.array([c["clip_len"], c["max_ts"]], dtype=np.float64)
    out[name + ".windows"] = np.array([r["pred_relevant_windows"] for r in res], dtype=np.float64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "windows.npz"), **out)
print({k: v.shape for k, v in out.items()})
