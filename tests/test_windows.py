"""Inference post-processing: the oracle against the golden rows produced by the
real reference (CPU), and the product path against both (GPU).  The rows are compared exactly: they
are what ends up in a submission file."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "windows.npz"))
CASES = ["qvh", "charades", "noclip"]


def _case(name):
    clip_len, max_ts = G[name + ".cfg"]
    return (torch.from_numpy(G[name + ".logits"]), torch.from_numpy(G[name + ".spans"]),
            torch.from_numpy(G[name + ".duration"]), int(clip_len), float(max_ts), G[name + ".windows"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_rows(name):
    from oracle.postprocess_oracle import windows
    lg, sp, du, clip_len, max_ts, want = _case(name)
    got = np.array(windows(lg, sp, du, clip_len=clip_len, max_ts_val=max_ts))
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_product_rows_are_identical(name):
    from mesm_amd.postprocess import predict_windows
    from oracle.postprocess_oracle import windows
    lg, sp, du, clip_len, max_ts, want = _case(name)
    dev = torch.device("cuda:0")
    got = np.array(predict_windows(lg.to(dev), sp.to(dev), du.to(dev), clip_len=clip_len, max_ts_val=max_ts))
    # start / end are multiples of clip_len (or 4-decimal numbers): exact; the score's 4th decimal may
    # move by one unit when expf rounds differently on the two devices
    assert np.array_equal(got[..., :2], want[..., :2])
    assert np.abs(got[..., 2] - want[..., 2]).max() <= 1.0001e-4
    assert np.array_equal(np.array(windows(lg, sp, du, clip_len=clip_len, max_ts_val=max_ts)), want)
