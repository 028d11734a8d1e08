"""The production (vectorized) host draws have the reference's distribution.

Both forms of mesm_amd.draws -- the loop-for-loop reference stream and the vectorized one -- and, where
/root/reference is present (the build container), the reference's own `sample_outclass_neg` / `MESM._mask_words`,
are held to the EXACT probabilities with a chi-square test on 20k draws:

  * negatives (utils/data_utils.py:113-124): uniform over the pairs of the other video groups;
  * masked words (model/model.py:361-384): k = max(l // 3, 1) of the first l positions by successive sampling without
    replacement proportional to the weight row (inclusion probabilities and the law of the unordered SET computed by
    enumeration), nothing for l <= 1.
"""
import itertools
import os
import sys

import numpy as np
import pytest
import torch
from scipy.stats import chi2

from mesm_amd import draws

DRAWS = 20000
ALPHA = 1e-4  # per test; the seeds are fixed, so this is a regression bound, not a flaky gate

GROUPS = [3, 1, 4, 2]
# word counts 0..10 (l <= 1: skipped; l = 2..5: k = 1; 6..8: k = 2; 9, 10: k = 3), ragged weights incl. a zero weight
LENS = [0, 1, 2, 3, 5, 6, 7, 9, 10]
LW = 12


def _weights():
    g = np.random.RandomState(5)
    w = g.uniform(0.2, 3.0, size=(len(LENS), LW)).astype(np.float32)
    w[4, 1] = 0.0                       # a word that can never be drawn
    for i, l in enumerate(LENS):
        w[i, l:] = 0.0                  # like the collate: zeros behind the last word
    return torch.from_numpy(w)


def _mask():
    m = torch.zeros(len(LENS), LW, dtype=torch.bool)
    for i, l in enumerate(LENS):
        m[i, :l] = True
    return m


def _set_law(p, k):
    """P(unordered set) under successive sampling without replacement ~ p, by enumeration"""
    l = len(p)
    law = {}
    for perm in itertools.permutations(range(l), k):
        pr, rest = 1.0, 1.0
        for j in perm:
            pr *= p[j] / rest
            rest -= p[j]
        key = tuple(sorted(perm))
        law[key] = law.get(key, 0.0) + pr
    return law


def _chi2_ok(counts, expect):
    counts, expect = np.asarray(counts, float), np.asarray(expect, float)
    live = expect > 0
    assert counts[~live].sum() == 0, "an impossible outcome was drawn"
    stat = ((counts[live] - expect[live]) ** 2 / expect[live]).sum()
    return stat <= chi2.ppf(1 - ALPHA, int(live.sum()) - 1), stat


def _ref_functions():
    root = "/root/reference"
    if not os.path.isdir(root):
        return None
    sys.path.insert(0, root)
    try:
        from model.model import MESM as RefMESM
        from utils.data_utils import sample_outclass_neg
        return RefMESM, sample_outclass_neg
    except Exception:  # the reference's imports need packages this box may lack
        return None
    finally:
        sys.path.remove(root)


def _neg_samplers():
    s = {"vectorized": lambda: draws.neg_index_vectorized(GROUPS),
         "reference-stream": lambda: draws.neg_index_reference(GROUPS).numpy()}
    ref = _ref_functions()
    if ref is not None:
        s["reference"] = lambda: ref[1](torch.tensor(GROUPS)).numpy()
    return s


@pytest.mark.parametrize("which", ["vectorized", "reference-stream", "reference"])
def test_negative_draw_is_uniform_over_the_other_groups(which):
    samplers = _neg_samplers()
    if which not in samplers:
        pytest.skip("/root/reference is not importable here")
    np.random.seed(11)
    torch.manual_seed(11)
    N = sum(GROUPS)
    gid = np.repeat(np.arange(len(GROUPS)), GROUPS)
    n = DRAWS if which == "vectorized" else DRAWS // 4   # (the loop forms cost ~0.3 ms per call)
    counts = np.zeros((N, N))
    for _ in range(n):
        neg = samplers[which]()
        counts[np.arange(N), neg] += 1
    for i in range(N):
        expect = np.where(gid != gid[i], n / (N - GROUPS[gid[i]]), 0.0)
        ok, stat = _chi2_ok(counts[i], expect)
        assert ok, (which, i, stat)


def _mask_samplers():
    wm, w = _mask(), _weights()
    s = {"vectorized": lambda wt: draws.masked_words_vectorized(wm, w if wt else None),
         "reference-stream": lambda wt: draws.masked_words_reference(wm, w if wt else None).bool().numpy()}
    ref = _ref_functions()
    if ref is not None:
        feat, tok = torch.zeros(len(LENS), LW, 1), torch.zeros(1)
        s["reference"] = lambda wt: ref[0]._mask_words(None, feat, wm, tok, proj=False,
                                                       weight=w if wt else None)[1].bool().numpy()
    return s


@pytest.mark.parametrize("weighted", [True, False])
@pytest.mark.parametrize("which", ["vectorized", "reference-stream", "reference"])
def test_masked_word_draw_has_the_successive_sampling_law(which, weighted):
    samplers = _mask_samplers()
    if which not in samplers:
        pytest.skip("/root/reference is not importable here")
    np.random.seed(13)
    n = DRAWS if which == "vectorized" else DRAWS // 4
    w = torch.nn.functional.normalize(_weights().float(), dim=1, p=1).numpy().astype(np.float64)
    seen = [dict() for _ in LENS]
    for _ in range(n):
        m = samplers[which](weighted)
        for i in range(len(LENS)):
            key = tuple(np.flatnonzero(m[i]))
            seen[i][key] = seen[i].get(key, 0) + 1
    for i, l in enumerate(LENS):
        if l <= 1:
            assert seen[i] == {(): n}, (which, l, seen[i])   # the `l <= 1: continue` of model.py:371
            continue
        k = max(l // 3, 1)
        p = w[i, :l] if weighted else np.full(l, 1.0 / l)
        law = _set_law(p, k)
        assert all(len(key) == k and max(key) < l for key in seen[i]), (which, l)
        keys = sorted(law)
        ok, stat = _chi2_ok([seen[i].get(key, 0) for key in keys], [law[key] * n for key in keys])
        assert ok, (which, weighted, l, stat)
        assert set(seen[i]) <= set(keys)


def test_vectorized_draws_follow_the_global_numpy_seed():
    wm, w = _mask(), _weights()
    np.random.seed(3)
    a = (draws.neg_index_vectorized(GROUPS), draws.masked_words_vectorized(wm, w))
    np.random.seed(3)
    b = (draws.neg_index_vectorized(GROUPS), draws.masked_words_vectorized(wm, w))
    c = (draws.neg_index_vectorized(GROUPS), draws.masked_words_vectorized(wm, w))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert not (np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]))


def test_padding_pairs_and_group_boundaries():
    np.random.seed(0)
    neg = draws.neg_index([2, 3, 1, 2], n_valid=6)
    assert neg.shape == (8,) and neg.dtype == np.int64 and (neg[6:] == 0).all()
    gid = np.repeat(np.arange(3), [2, 3, 1])
    assert (gid[neg[:6]] != gid).all() and neg[:6].max() < 6
    with pytest.raises(ValueError):
        draws.neg_index([2, 3, 1, 2], n_valid=4)


@pytest.mark.parametrize("mode", draws.MODES)
def test_error_cases_raise_like_the_per_pair_calls(mode):
    with pytest.raises(IndexError):              # candidate[...][0] on an empty candidate list
        draws.neg_index([3], mode=mode)
    wm = torch.zeros(2, 8, dtype=torch.bool)
    wm[0, :6] = True
    wm[1, :3] = True
    w = torch.ones(2, 8)
    w[:, 6:] = 0
    w[1, 3:] = 0
    assert draws.masked_words(wm, w, mode=mode).sum(1).tolist() == [2, 1]
    few = w.clone()
    few[0, 1:6] = 0                              # one non-zero weight, two words to mask
    with pytest.raises(ValueError, match="Fewer non-zero"):
        draws.masked_words(wm, few, mode=mode)
    tail = w.clone()
    tail[1, 5] = 1.0                             # weight behind the last word: p[:l] no longer sums to 1
    with pytest.raises(ValueError, match="do not sum to 1"):
        draws.masked_words(wm, tail, mode=mode)
    neg = w.clone()
    neg[0, 2] = -1.0
    with pytest.raises(ValueError):
        draws.masked_words(wm, neg, mode=mode)


def test_vectorized_is_cheap():
    """the point of the vectorized form: the C3a step's draws (32 pairs) in well under the device step"""
    import time
    g = np.random.RandomState(0)
    groups = [2] * 16
    wm = torch.zeros(32, 32, dtype=torch.bool)
    w = torch.zeros(32, 32)
    for i in range(32):
        l = int(g.randint(4, 25))
        wm[i, :l] = True
        w[i, :l] = torch.from_numpy(g.uniform(0.5, 2.0, l).astype(np.float32))
    t0 = time.perf_counter()
    for _ in range(200):
        draws.neg_index(groups, mode="vectorized")
        draws.masked_words(wm, w, mode="vectorized")
    per = (time.perf_counter() - t0) / 200 * 1e3
    assert per < 0.5, "vectorized draws took %.3f ms per step" % per
