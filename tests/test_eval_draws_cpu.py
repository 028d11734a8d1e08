"""CPU checks against the second batch of reference fixtures (tools/gen_golden_r2.py):

  * the oracle's inference call (is_training=False: MLM branch and rec_fw loss off) against the
    reference's eval.py:63,102 call;
  * the host-RNG draws of the product (mesm_amd.MESM.draw_neg_index / draw_masked_words) replayed
    under the seeds the reference's own sample_outclass_neg / _mask_words were recorded with.
"""
import numpy as np
import pytest
import torch

from golden_io import CASES, EvalFixture, Fixture, draw_cases
from oracle import mesm_oracle as O


def close(a, b, tol=2e-5):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1.0) < tol


@pytest.mark.parametrize("case", CASES)
def test_oracle_eval_mode_matches_reference(case):
    fx, ev = Fixture(case), EvalFixture(case)
    with torch.no_grad():
        out = O.mesm_forward(fx.sd, fx.cfg, fx.batch, ev.neg_index, None, is_training=False)
        losses, total, _ = O.criterion_forward(out, fx.batch, fx.cfg, is_training=False)
    assert sorted(k for k in out if k != "aux_outputs") == [k for k in ev.keys if k != "aux_outputs"]
    for k, v in ev.out.items():
        if k.startswith("aux0."):
            got = out["aux_outputs"][0][k[5:]]
        else:
            got = out[k]
        if v.dtype == torch.bool:
            assert torch.equal(got, v), k
        else:
            assert close(got, v), k
    assert set(losses) | {"total"} == set(ev.losses)
    for k, v in ev.losses.items():
        got = float(total) if k == "total" else float(losses[k])
        assert abs(got - v) < 1e-4 * max(1.0, abs(v)), (k, got, v)


@pytest.mark.parametrize("ci", range(4))
def test_host_draws_replay_the_reference(ci):
    from mesm_amd.model import MESM
    groups, wmask, weight, seed, want = draw_cases()[ci]
    for tag in ("w", "u"):
        torch.manual_seed(seed)
        np.random.seed(seed)
        neg = MESM.draw_neg_index(groups)
        masked = MESM.draw_masked_words(wmask, weight if tag == "w" else None)
        assert torch.equal(neg, want[tag][0]), tag
        assert torch.equal(masked.bool(), want[tag][1].bool()), tag
        # every negative comes from another video group
        gid = torch.repeat_interleave(torch.arange(len(groups)), torch.tensor(groups))
        assert bool((gid[neg] != gid).all())


def test_single_group_batch_raises_like_the_reference():
    from mesm_amd.model import MESM
    with pytest.raises(IndexError):
        MESM.draw_neg_index([3])
