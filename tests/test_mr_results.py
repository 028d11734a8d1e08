"""Submission rows: mesm_amd.postprocess against tests/golden/mr_results.json, which
tools/gen_golden_io.py produced by driving the REAL eval.compute_mr_results (eval.py:52-117) with a stub model /
loader / criterion and the real post_processing_mr_nms (eval.py:476-485, utils/temporal_nms.py:25-74)."""
import argparse
import copy
import json
import os

import pytest
import torch

from golden_io import GOLDEN
from io_cases import MR_CASES, StubCriterion, StubModel, mr_inputs

G = json.load(open(os.path.join(GOLDEN, "mr_results.json")))


@pytest.mark.parametrize("name", sorted(MR_CASES))
def test_temporal_nms_matches_reference(name):
    from mesm_amd.postprocess import post_processing_mr_nms
    c = MR_CASES[name]
    got = post_processing_mr_nms(copy.deepcopy(G[name]["mr_res"]), nms_thd=c["nms_thd"], max_before_nms=10,
                                 max_after_nms=5)
    assert [e["pred_relevant_windows"] for e in got] == [e["pred_relevant_windows"] for e in G[name]["after_nms"]]
    assert any(len(e["pred_relevant_windows"]) < min(10, c["Q"]) for e in got)  # something was suppressed


def test_temporal_nms_edge_cases():
    from mesm_amd.postprocess import temporal_nms
    assert temporal_nms([[0, 1, 0.5]], 0.5) == [[0, 1, 0.5]]
    rows = [[0, 10, 0.9], [1, 10, 0.8], [20, 30, 0.7], [0, 0, 0.1]]
    assert temporal_nms(rows, 0.5) == [[0, 10, 0.9], [20, 30, 0.7], [0, 0, 0.1]]
    assert temporal_nms(rows, 0.5, max_after_nms=1) == [[0, 10, 0.9]]
    assert temporal_nms(rows, 1.0) == sorted(rows, key=lambda x: -x[2])  # nothing overlaps more than 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MR_CASES))
def test_compute_mr_results_rows_match_reference(name):
    from mesm_amd.postprocess import compute_mr_results
    dev = torch.device("cuda:0")
    c = MR_CASES[name]
    loader, outs = mr_inputs(c)
    loader = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in loader]
    outs = [{k: v.to(dev) for k, v in o.items()} for o in outs]
    opt = argparse.Namespace(dataset_name=name, clip_len=c["clip_len"], max_ts_val=150, sort_results=True)
    rows, meters = compute_mr_results(StubModel(outs), loader, opt, criterion=StubCriterion())
    want = G[name]["mr_res"]
    assert len(rows) == len(want)
    for a, b in zip(rows, want):
        assert (a["qid"], a["query"], a["vid"]) == (b["qid"], b["query"], b["vid"])
        assert a["pred_saliency_scores"] == b["pred_saliency_scores"]  # fp16 values, exact
        wa, wb = a["pred_relevant_windows"], b["pred_relevant_windows"]
        assert [r[:2] for r in wa] == [r[:2] for r in wb]           # start / end: exact
        assert max(abs(x[2] - y[2]) for x, y in zip(wa, wb)) <= 1.0001e-4  # score: 4th decimal may move by one unit
    for k, v in G[name]["loss_meters"].items():
        assert abs(meters[k] - v) < 1e-5 * max(1.0, abs(v)), k
