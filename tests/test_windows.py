"""Inference post-processing: the oracle (oracle/postprocess_oracle.py) against the submission
rows the REAL eval.compute_mr_results + PostProcessorDETR produced (tests/golden/mr_results.json, made by
tools/gen_golden_io.py with a stub model), and the product path against both (GPU; tests/test_mr_results.py holds the
end-to-end product test incl. saliency rows, NMS and loss meters).  Rows are compared exactly: they are what ends up
in a submission file."""
import json
import os

import numpy as np
import pytest
import torch

from golden_io import GOLDEN
from io_cases import MR_CASES, mr_inputs

G = json.load(open(os.path.join(GOLDEN, "mr_results.json")))
CASES = sorted(MR_CASES)


def _batches(name):
    c = MR_CASES[name]
    loader, outs = mr_inputs(c)
    want, k = [], 0
    for b in loader:
        n = len(b["qid"])
        want.append(np.array([r["pred_relevant_windows"] for r in G[name]["mr_res"][k:k + n]]))
        k += n
    return c, loader, outs, want


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_rows(name):
    from oracle.postprocess_oracle import windows
    c, loader, outs, want = _batches(name)
    for b, o, w in zip(loader, outs, want):
        got = np.array(windows(o["pred_logits"], o["pred_spans"], b["duration"], clip_len=c["clip_len"], max_ts_val=150))
        assert got.shape == w.shape and np.array_equal(got, w)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_product_rows_are_identical(name):
    from mesm_amd.postprocess import predict_windows
    c, loader, outs, want = _batches(name)
    dev = torch.device("cuda:0")
    for b, o, w in zip(loader, outs, want):
        got = np.array(predict_windows(o["pred_logits"].to(dev), o["pred_spans"].to(dev), b["duration"].to(dev),
                                       clip_len=c["clip_len"], max_ts_val=150))
        # start / end are multiples of clip_len (or 4-decimal numbers): exact; the score's 4th decimal may
        # move by one unit when expf rounds differently on the two devices
        assert np.array_equal(got[..., :2], w[..., :2])
        assert np.abs(got[..., 2] - w[..., 2]).max() <= 1.0001e-4
