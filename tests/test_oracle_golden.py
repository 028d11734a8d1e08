"""The CPU oracle (oracle/mesm_oracle.py) against the golden vectors produced by the real
reference (tools/gen_golden.py) and against the reference's own span/gIoU doctest answers."""
import os

import numpy as np
import pytest
import torch

from golden_io import CASES, CLIP_CASES, GOLDEN, Fixture
from oracle import mesm_oracle as O

TOL = 2e-5


def close(a, b, tol=TOL):
    scale = max(float(b.abs().max()), 1.0)
    return float((a - b).abs().max()) / scale < tol


@pytest.fixture(scope="module", params=CASES + CLIP_CASES)
def step(request):
    fx = Fixture(request.param)
    out, losses, total, grads, idx = O.train_step(fx.sd, fx.cfg, fx.batch, fx.neg_index, fx.masked_words)
    return fx, out, losses, total, grads, idx


def test_outputs_match_reference(step):
    fx, out, *_ = step
    for k in ("pred_logits", "pred_spans", "saliency_scores", "neg_saliency_scores",
              "recfw_words_logit", "recon_feat", "projed_recon_feat", "projed_video_feat",
              "expanded_words_feat", "enhanced_video_feat", "projed_words_feat"):
        assert close(out[k].detach(), fx.out[k]), k
    assert close(out["aux_outputs"][0]["pred_logits"].detach(), fx.out["aux0.pred_logits"])
    assert close(out["aux_outputs"][0]["pred_spans"].detach(), fx.out["aux0.pred_spans"])
    assert torch.equal(out["expanded_words_mask"], fx.out["expanded_words_mask"].bool())
    assert torch.equal(out["words_mask"], fx.out["words_mask"].bool())


def test_losses_match_reference(step):
    fx, _, losses, total, _, _ = step
    for k, v in fx.losses.items():
        if k == "total":
            assert abs(float(total) - v) < 1e-4 * max(1.0, abs(v)), (k, float(total), v)
        else:
            assert abs(float(losses[k]) - v) < 1e-4 * max(1.0, abs(v)), (k, float(losses[k]), v)


def test_matching_is_bit_exact(step):
    fx, _, _, _, _, idx = step
    for layer, ind in zip(["main", "aux0"], idx):
        got = set()
        for b, (q, t) in enumerate(ind):
            for qq, tt in zip(q.tolist(), t.tolist()):
                got.add((b, qq, tt))
        assert got == fx.matched_pairs(layer), layer


def test_gradients_match_reference(step):
    fx, _, _, _, grads, _ = step
    assert set(grads) == set(fx.grads), set(grads) ^ set(fx.grads)
    for k, g in fx.grads.items():
        assert close(grads[k], g, 1e-4), k


def test_span_doctest_values():
    z = np.load(os.path.join(GOLDEN, "span_doctests.npz"))
    t = lambda k: torch.from_numpy(z[k])
    assert torch.allclose(O.span_xx_to_cxw(t("xx")), t("cxw"))
    assert torch.allclose(O.span_cxw_to_xx(t("cxw")), t("back"))
    iou, union = O.temporal_iou(t("a"), t("b"))
    assert torch.allclose(iou, t("iou")) and torch.allclose(union, t("union"))
    assert torch.allclose(O.generalized_temporal_iou(t("a"), t("b")), t("giou"))
    # the literal answers printed in utils/span_utils.py:54-60 and :105-109
    assert torch.allclose(iou, torch.tensor([[0.6667, 0.2], [0.0, 0.5]]), atol=1e-4)
    assert torch.allclose(O.generalized_temporal_iou(t("a"), t("b")),
                          torch.tensor([[0.6667, 0.2], [-0.2, 0.5]]), atol=1e-4)
