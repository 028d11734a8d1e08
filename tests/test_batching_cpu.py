"""Batch assembly (mesm_amd/batching.py) against fixtures produced by the REAL reference
functions (tools/gen_golden_io.py -> tests/golden/collate.npz): dataset/base.py `collate`,
dataset/qvhighlights.py `collate`, `prepare_batch_input`, `pad_sequences_1d`.  Exact equality, dtypes included."""
import json
import os

import numpy as np
import pytest
import torch

from golden_io import GOLDEN
from io_cases import group_samples

Z = np.load(os.path.join(GOLDEN, "collate.npz"))
META = json.loads(bytes(Z["meta.json"]).decode())


def check(prefix, got):
    keys = {k[len(prefix):].split(".")[0] for k in list(Z.files) + list(META) if k.startswith(prefix)}
    assert set(got) == keys, (sorted(got), sorted(keys))
    for k, v in got.items():
        if torch.is_tensor(v):
            want = torch.from_numpy(Z[prefix + k])
            assert v.dtype == want.dtype and v.shape == want.shape and torch.equal(v, want), k
        elif isinstance(v, list) and v and isinstance(v[0], dict):
            f = META[prefix + k]
            assert [len(x[f]) for x in v] == Z[prefix + k + ".sizes"].tolist(), k
            assert torch.equal(torch.cat([x[f] for x in v]), torch.from_numpy(Z[prefix + k + ".cat"])), k
        else:
            assert v == META[prefix + k], k


@pytest.mark.parametrize("kind", ["base", "qvh"])
def test_collate_and_prepare_match_the_reference(kind):
    from mesm_amd import batching as B
    fn = B.collate if kind == "base" else B.collate_qvh
    out = fn(group_samples(kind, META[kind + ".seed"]))
    check(kind + ".out.", out)
    prep = B.prepare_batch_input(dict(out), torch.device("cpu"))
    check(kind + ".prep.", prep)


def test_pad_sequences_1d():
    from mesm_amd.batching import pad_sequences_1d
    p, m = pad_sequences_1d([[1, 2, 3], [1, 2], [3, 4, 7, 9]], dtype=torch.long)
    assert p.tolist() == [[1, 2, 3, 0], [1, 2, 0, 0], [3, 4, 7, 9]] and m.dtype == torch.bool
    assert m.tolist() == [[True] * 3 + [False], [True] * 2 + [False] * 2, [True] * 4]
    seqs = [torch.randn(2, 3, 4), torch.randn(4, 3, 4), torch.randn(0, 3, 4), torch.randn(1, 3, 4)]
    p, m = pad_sequences_1d(seqs, dtype=torch.float32, fixed_length=5)
    assert p.shape == (4, 5, 3, 4) and m.sum(1).tolist() == [2, 4, 0, 1]
    for i, s in enumerate(seqs):
        assert torch.equal(p[i, :len(s)], s) and float(p[i, len(s):].abs().sum()) == 0.0


@pytest.mark.gpu
def test_prepare_batch_input_packs_one_transfer():
    from mesm_amd import batching as B
    out = B.collate_qvh(group_samples("qvh", META["qvh.seed"]))
    host = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}
    prep = B.prepare_batch_input(out, torch.device("cuda:0"))
    assert not prep["words_weight"].is_cuda  # stays on the host (base.py:360-361)
    for k, v in host.items():
        if torch.is_tensor(v) and k != "words_weight":
            assert prep[k].is_cuda and prep[k].dtype == v.dtype and torch.equal(prep[k].cpu(), v), k
    assert all(e["spans"].is_cuda for e in prep["norm_span"])
    assert torch.equal(torch.cat([e["moments"] for e in prep["norm_moment"]]).cpu(),
                       torch.cat([e["moments"] for e in host["norm_moment"]]))


def test_arena_layout_puts_the_drawn_arrays_first_and_cpu_upload_refreshes_everything():
    """arena.Arena (the one-transfer home of a step's small tensors): `first` names lead the layout side by side;
    on a CPU device an upload (full or `only=...`) rewrites the views in place."""
    import numpy as np
    from mesm_amd.arena import Arena
    arr = {"a": np.arange(5, dtype=np.int64), "neg": np.arange(3, dtype=np.int64), "b": np.ones((2, 3), np.float32),
           "mw": np.zeros((2, 4), np.bool_)}
    ar = Arena(arr, "cpu", first=("neg", "mw"))
    names = [s[0] for s in ar.specs]
    assert names[:2] == ["neg", "mw"] and sorted(names) == sorted(arr)
    offs = {s[0]: s[3] for s in ar.specs}
    assert offs["neg"] == 0 and offs["mw"] == 32  # 24 bytes rounded up to the 16-byte grid
    assert torch.equal(ar.views["a"], torch.arange(5)) and ar.views["b"].shape == (2, 3)
    arr2 = dict(arr, neg=np.array([7, 8, 9], dtype=np.int64), mw=np.ones((2, 4), np.bool_))
    ar.upload(arr2, only=("neg", "mw"))
    assert ar.views["neg"].tolist() == [7, 8, 9] and bool(ar.views["mw"].all())
    assert torch.equal(ar.views["a"], torch.arange(5))
    with pytest.raises(ValueError):
        ar.upload(dict(arr, neg=np.arange(4, dtype=np.int64)))


def test_loader_workers_hand_feature_tensors_over_through_the_shared_ring():
    """loader.PinnedRing (shared host slots; page-locked when a GPU is there): batches prepared by forked DataLoader
    workers come back with their big tensors as views of the ring, equal to what prepare() gives in-process -- over two
    passes of more batches than the ring has slots (every slot is reused)."""
    import torch
    from mesm_amd import synthetic
    from mesm_amd.hostplan import HostSpec
    from mesm_amd.loader import HostPipeline, prepared_loader
    args = synthetic.make_args("C2")
    w = synthetic.WORKLOADS["C2"]
    spec = HostSpec.from_args(args)
    pipe = HostPipeline(spec, pad=(w["Lv"], w["Lw"]), pairs=8, group_caps=(5, 9), keep_raw=False)
    pipe.BIG = 1 << 12   # (small synthetic features still count as "big" here)
    batches = [synthetic.make_batch(w["dataset_name"], [2, 1, 3][: 1 + i % 3], w["Lv"], w["Lw"], 64, 32, w["vocab_size"] + 1,
                                    seed=40 + i, ragged=True) for i in range(9)]
    want = [pipe.prepare(b) for b in batches]
    loader = prepared_loader(batches, pipe, num_workers=2, prefetch_factor=1, ring=True)
    assert loader.ring.slots == 5
    for _ in range(2):
        got = 0
        for prep, ref in zip(loader, want):
            assert set(prep["big"]) == set(ref["big"]) and prep["big"]
            for k, v in ref["big"].items():
                assert prep["big"][k].shape == v.shape and torch.equal(prep["big"][k], v), k
            assert prep["key"] == ref["key"] and prep["groups"] == ref["groups"]
            got += 1
        assert got == len(batches)


@pytest.mark.parametrize("kind", ["base", "qvh"])
def test_host_side_kept_by_the_collate_survives_prepare_batch_input(kind):
    """batching.attach_host_side: the host copies of everything small ride under batch["_host"], a plain dict that
    prepare_batch_input (dataset/base.py:358-384 semantics: tensors and the two target lists move, everything else stays) leaves
    alone; its entries are the batch's own values, and for tensor-target datasets the derived norm_moment / norm_span are there too."""
    from mesm_amd import batching as B
    fn = B.collate if kind == "base" else B.collate_qvh
    out = B.attach_host_side(fn(group_samples(kind, META[kind + ".seed"])))
    host = out["_host"]
    assert isinstance(host, dict) and "video_mask" in host and "num_clips" in host
    for k, v in host.items():
        if torch.is_tensor(v) and k in out and torch.is_tensor(out[k]):
            assert v.numel() * v.element_size() <= B.HOST_SIDE_BIG and torch.equal(v, out[k]), k
    assert "video_feat" not in host or out["video_feat"].numel() * out["video_feat"].element_size() <= B.HOST_SIDE_BIG
    prep = B.prepare_batch_input(dict(out), torch.device("cpu"))
    assert prep["_host"] is host
    check(kind + ".prep.", {k: v for k, v in prep.items() if k != "_host"})
    if kind == "base":
        assert torch.equal(host["norm_moment"], prep["norm_moment"]) and torch.equal(host["norm_span"], prep["norm_span"])
    else:
        assert [torch.equal(a["spans"], b["spans"]) for a, b in zip(host["norm_span"], prep["norm_span"])]
    w = out["words_id"]
    if w.dim() == 3:
        assert host["_words_mask_raw"].shape == w.shape[:2] and host["_words_mask_norm"].dtype == torch.bool


@pytest.mark.gpu
def test_prepare_batch_input_keeps_the_host_side_of_a_host_batch_bound_for_the_gpu():
    """batching.prepare_batch_input: a host batch on its way to a GPU keeps the host copies of its small tensors under `_host`
    (what the collate-side attach_host_side leaves there), a batch that carries them already is left alone, a CPU target and
    MESM_KEEP_HOST_SIDE=0 give the reference's key set."""
    from mesm_amd import batching as B
    out = B.collate_qvh(group_samples("qvh", META["qvh.seed"]))
    want = B.attach_host_side({k: v for k, v in out.items()})["_host"]
    prep = B.prepare_batch_input(dict(out), torch.device("cuda:0"))
    assert set(prep["_host"]) == set(want)
    for k, v in want.items():
        if torch.is_tensor(v):
            assert not prep["_host"][k].is_cuda and torch.equal(prep["_host"][k], v), k
    mine = B.attach_host_side(dict(out))
    token = mine["_host"]
    assert B.prepare_batch_input(mine, torch.device("cuda:0"))["_host"] is token
    assert "_host" not in B.prepare_batch_input(dict(out), torch.device("cpu"))
    B._KEEP_HOST_SIDE = False
    try:
        assert "_host" not in B.prepare_batch_input(dict(out), torch.device("cuda:0"))
    finally:
        B._KEEP_HOST_SIDE = True
