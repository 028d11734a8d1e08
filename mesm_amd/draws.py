"""The two host-RNG draws of a training forward: the negative query of every pair (`sample_outclass_neg`,
utils/data_utils.py:113-124 of the reference) and the masked-LM word positions (`_mask_words`, model/model.py:361-384).

Two implementations of each, with the SAME distribution:

  * ``reference``: loop for loop what the reference does -- one `torch.randperm` per pair, one `np.random.choice` per
    pair -- so that a run seeded like a reference run makes the same draws (pinned against the reference's own functions
    under recorded seeds, tests/test_eval_draws_cpu.py).  3.5 ms of Python per 32-pair step.
  * ``vectorized`` (the default, `MESM_DRAWS=vectorized`): one numpy expression over all pairs, fed by the same global
    `np.random` stream the reference's `_mask_words` uses (so `np.random.seed` / the caller's `set_seed` still governs
    it).  `cand[randperm(len(cand))][0]` is a uniform pick among the pairs of the other video groups = one uniform
    integer per pair; `np.random.choice(l, k, replace=False, p=p)` keeps the first k distinct values of an i.i.d.
    stream from p (it redraws with the found entries zeroed, which continues that stream), i.e. successive sampling
    without replacement with probability proportional to p = the k largest of log p + Gumbel noise.  The error cases
    of the per-pair call (fewer non-zero weights than words to mask, weights that do not sum to one over the valid
    prefix, negative / NaN weights, a batch with a single video group) raise the same exception types.
    tests/test_draws_cpu.py holds both forms (and the reference's functions where /root/reference is present) to
    the exact inclusion probabilities with a chi-square test.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

MODES = ("vectorized", "reference")
MODE = os.environ.get("MESM_DRAWS", "vectorized")
if MODE not in MODES:
    raise ValueError("MESM_DRAWS=%r: expected one of %s" % (MODE, MODES))


# ---------------------------------------------------------------------------------------------- reference stream
def neg_index_reference(groups):
    """sample_outclass_neg (utils/data_utils.py:113-124): for every pair one query index drawn uniformly from the
    OTHER video groups, one torch.randperm per pair (torch's global host generator).  -> int64 tensor (N,)"""
    if len(groups) < 2:
        raise IndexError("index 0 is out of bounds: negatives need >= 2 video groups in a batch")
    N = sum(groups)
    neg, start = [], 0
    for g in groups:
        cand = torch.cat([torch.arange(0, start), torch.arange(start + g, N)])
        for _ in range(g):
            neg.append(cand[torch.randperm(cand.shape[0])][0])
        start += g
    return torch.stack(neg)


def masked_words_reference(words_mask_cpu, words_weight):
    """_mask_words (model.py:361-384): max(l // 3, 1) of the first l positions per pair, without replacement,
    p ~ words_weight, numpy's global RNG; pairs with <= 1 word are skipped.  -> tensor like words_mask_cpu"""
    masked = torch.zeros_like(words_mask_cpu)
    weight = F.normalize(words_weight.float().cpu(), dim=1, p=1) if words_weight is not None else None
    for i, l in enumerate(words_mask_cpu.count_nonzero(dim=1)):
        l = int(l)
        if l <= 1:
            continue
        k = max(l // 3, 1)
        p = weight[i, :l].numpy() if weight is not None else None
        choices = np.random.choice(np.arange(0, l), k, replace=False, p=p)
        masked[i, choices] = 1
    return masked


# ---------------------------------------------------------------------------------------------- vectorized
def neg_index_vectorized(groups, rng=None):
    """One uniform integer per pair over the N - g pairs outside its own group of g (same law as the reference's
    randperm()[0]).  rng: anything with random_sample (default: numpy's global RandomState).  -> int64 array (N,)"""
    if len(groups) < 2:
        raise IndexError("index 0 is out of bounds: negatives need >= 2 video groups in a batch")
    rng = np.random if rng is None else rng
    g = np.asarray(groups, dtype=np.int64)
    N = int(g.sum())
    start = np.repeat(np.cumsum(g) - g, g)     # first pair of every pair's own group
    size = np.repeat(g, g)
    cand = N - size                            # >= 1 with two or more (non-empty) groups
    if N and int(cand.min()) < 1:
        raise IndexError("index 0 is out of bounds: a pair has no query outside its video group")
    u = np.minimum((rng.random_sample(N) * cand).astype(np.int64), cand - 1)
    return np.where(u < start, u, u + size)


_ATOL64 = float(np.sqrt(np.finfo(np.float64).eps))
_ATOL32 = float(np.sqrt(np.finfo(np.float32).eps))  # np.random.choice widens its tolerance for float32 weights


def masked_words_vectorized(words_mask, words_weight, rng=None):
    """Gumbel top-k over all pairs at once: k_i = max(l_i // 3, 1) of the first l_i positions (l_i = number of valid
    words; 0 picks where l_i <= 1), successive sampling without replacement proportional to the L1-normalised weight
    row (uniform without weights).  -> bool array like words_mask"""
    rng = np.random if rng is None else rng
    wm = words_mask.numpy() if torch.is_tensor(words_mask) else np.asarray(words_mask)
    wm = wm.astype(bool, copy=False)
    N, L = wm.shape
    l = wm.sum(1)
    k = np.where(l > 1, np.maximum(l // 3, 1), 0)
    cols = np.arange(L)
    valid = cols[None, :] < l[:, None]          # the reference indexes the first l positions, whatever the mask's shape
    key = rng.gumbel(size=(N, L))
    if words_weight is not None:
        w = words_weight.numpy() if torch.is_tensor(words_weight) else np.asarray(words_weight)
        w = w.astype(np.float32, copy=False)
        p = (w / np.maximum(np.abs(w).sum(1, keepdims=True), 1e-12))[:, :L].astype(np.float64)  # F.normalize(p=1)
        if p.shape[1] < L:
            raise IndexError("words_weight has %d columns, the word mask %d" % (p.shape[1], L))
        live = k > 0
        pv = np.where(valid & live[:, None], p, 0.0)
        if np.isnan(pv).any():
            raise ValueError("probabilities contain NaN")
        if (pv < 0).any():
            raise ValueError("probabilities are not non-negative")
        if (np.abs(pv.sum(1) - 1.0)[live] > max(_ATOL64, _ATOL32)).any():
            raise ValueError("probabilities do not sum to 1")
        if ((pv > 0).sum(1) < k).any():
            raise ValueError("Fewer non-zero entries in p than size")
        with np.errstate(divide="ignore"):
            key = key + np.log(pv)
    key = np.where(valid, key, -np.inf)
    # rank of every position in its row by descending key; the k best are the draw
    order = np.argsort(-key, axis=1, kind="stable")
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.broadcast_to(cols, (N, L)), axis=1)
    return (rank < k[:, None]) & valid


# ---------------------------------------------------------------------------------------------- what the step calls
def real_groups(groups, n_valid):
    """the video groups that hold the first n_valid pairs (the rest are padding pairs, batching.pad_pairs)"""
    if n_valid is None:
        return list(groups)
    real, tot = [], 0
    for g in groups:
        if tot >= n_valid:
            break
        real.append(g)
        tot += g
    if tot != n_valid:
        raise ValueError("n_valid = %d does not end at a group boundary of %s" % (n_valid, list(groups)))
    return real


def neg_index(groups, n_valid=None, mode=None):
    """negative query index of every pair over the REAL groups as an int64 array; padding pairs point at pair 0 (their
    rows are never read by a loss)"""
    real = real_groups(groups, n_valid)
    if (mode or MODE) == "reference":
        neg = neg_index_reference(real).numpy()
    else:
        neg = neg_index_vectorized(real)
    pad = sum(groups) - sum(real)
    return np.concatenate([neg, np.zeros(pad, dtype=np.int64)]) if pad else neg


def masked_words(words_mask_cpu, words_weight, mode=None):
    """masked-LM positions as a bool array (N, Lw)"""
    if (mode or MODE) == "reference":
        wm = words_mask_cpu if torch.is_tensor(words_mask_cpu) else torch.from_numpy(np.asarray(words_mask_cpu))
        return masked_words_reference(wm, words_weight).bool().numpy()
    return masked_words_vectorized(words_mask_cpu, words_weight)
