"""Data-parallel reduction on CPU with gloo, world_size 2.

  * parity target of the sharded path: every rank ends with the MEAN over ranks of the single-process
    (oracle) gradient on each rank's shard of video groups -- driven through shard_groups, the flat
    gradient buffer and the hooked GradReducer with the real model's parameter set (qvh_tiny weights);
  * a toy autograd model: learning step, overlapped steps, unused parameters;
  * bucket layout and launch order.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

GROUPS = [2, 1, 2, 1, 1]  # 5 video groups, 7 pairs: rank 0 gets groups 0,2,4, rank 1 gets 1,3


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Parameter(torch.randn(40, 16))
        self.b = nn.Parameter(torch.randn(16))
        self.c = nn.Parameter(torch.randn(16, 8))
        self.unused = nn.Parameter(torch.randn(5))
        self.d = nn.Parameter(torch.randn(8))

    def forward(self, x):
        h = torch.tanh(x @ self.a + self.b)
        # `c` is used twice -> two gradient contributions per step
        return ((h @ self.c) * self.d).sum() + (h @ self.c).pow(2).mean()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mesm_amd.ddp import init_process_group_from_env
    init_process_group_from_env(torch.device("cpu"))


def _worker(rank, world, port, out, inline=False):
    _init(rank, world, port)
    from mesm_amd.ddp import GradReducer
    from mesm_amd.gradbuf import GradBuffer
    m = Toy()
    gb = GradBuffer([(n, p) for n, p in m.named_parameters()])
    gb.ensure(torch.device("cpu"))
    red = GradReducer(gb, n_buckets=3, inline=inline)
    results = []
    for step in range(3):
        torch.manual_seed(100 * step + rank)
        x = torch.randn(12, 40)
        gb.begin_step()
        loss = m(x)
        m.zero_grad(set_to_none=True)
        loss.backward()  # reducer.finish() runs as an autograd-engine callback
        results.append({n: (p.grad.clone() if p.grad is not None else None) for n, p in m.named_parameters()})
    out[rank] = results
    dist.destroy_process_group()


def _local_grads(step, rank):
    m = Toy()
    torch.manual_seed(100 * step + rank)
    x = torch.randn(12, 40)
    m(x).backward()
    return {n: p.grad for n, p in m.named_parameters()}


@pytest.mark.timeout(120)
@pytest.mark.parametrize("inline", [False, True])
def test_gloo_world2_mean_of_rank_gradients(inline):
    """inline: blocking collectives issued from inside backward (the one-linear-chain mode of a captured step)"""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, inline), nprocs=world, join=True)
    for step in range(3):
        want = [_local_grads(step, r) for r in range(world)]
        for r in range(world):
            got = out[r][step]
            for n in ("a", "b", "c", "d"):
                mean = sum(w[n] for w in want) / world
                assert torch.allclose(got[n], mean, atol=1e-6), (step, r, n)
            assert got["unused"] is None  # never touched -> stays None, like under autograd


# ----------------------------------------------------------------------------- the real parameter set
def _tiny_global_batch():
    from golden_io import Fixture
    from mesm_amd import synthetic
    fx = Fixture("qvh_tiny")
    c = fx.cfg
    batch = synthetic.make_batch("qvhighlights", GROUPS, c["Lv"], c["Lw"], c["v_feat_dim"], c["t_feat_dim"],
                                 c["vocab_size"] + 1, seed=21, ragged=True)
    return fx, batch


def _shard_oracle_grads(fx, batch, rank, world):
    from mesm_amd import synthetic
    from mesm_amd.ddp import shard_groups
    from oracle import mesm_oracle as O
    shard = shard_groups(batch, rank, world)
    neg, masked = synthetic.host_draws(shard, seed=rank)  # per-rank RNG streams
    _, _, total, grads, _ = O.train_step(fx.sd, fx.cfg, shard, neg, masked)
    return shard, grads, float(total)


def _worker_real(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    _init(rank, world, port)
    torch.set_num_threads(2)
    from mesm_amd.ddp import GradReducer
    from mesm_amd.gradbuf import GradBuffer, flush_ready, grad_target
    fx, batch = _tiny_global_batch()
    _, grads, _ = _shard_oracle_grads(fx, batch, rank, world)
    params = [(n, nn.Parameter(v.clone())) for n, v in fx.sd.items() if v.is_floating_point()]
    gb = GradBuffer(params)
    gb.ensure(torch.device("cpu"))
    red = GradReducer(gb, n_buckets=5)

    class Emit(torch.autograd.Function):
        """stands in for the model's backward blocks: gradients arrive in reverse parameter order, the
        shared encoder weights in two contributions (positive + negative pass), and on rank 1 one
        parameter gets NO contribution at all (like unknown_token without an OOV word)."""

        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            for n, p in reversed(params):
                if n not in grads:
                    continue
                if n == "unknown_token" and rank == 1:
                    continue
                parts = 2 if n.startswith("enhance_encoder") else 1
                for _ in range(parts):
                    v, direct = grad_target(p)
                    assert direct
                    v.add_(grads[n] / parts)
                    flush_ready()
            return g

    res = []
    for step in range(3):
        gb.begin_step()
        for _, p in params:
            p.grad = None
        x = torch.zeros(1, requires_grad=True)
        Emit.apply(x).sum().backward()
        res.append({n: p.grad.clone() for n, p in params if p.grad is not None})
    out[rank] = (res, list(red.launch_log), len(red.buckets))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gloo_world2_sharded_oracle_gradients_are_averaged():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_real, args=(world, port, out), nprocs=world, join=True)
    fx, batch = _tiny_global_batch()
    per_rank = [_shard_oracle_grads(fx, batch, r, world)[1] for r in range(world)]
    per_rank[1] = {k: v for k, v in per_rank[1].items() if k != "unknown_token"}
    names = set(per_rank[0]) | set(per_rank[1])
    for r in range(world):
        steps, log, nb = out[r]
        # same bucket order on both ranks in every step: last bucket first, then downwards
        assert log == list(range(nb - 1, -1, -1)) * 3, log
        for got in steps:
            for n in names:
                mean = sum(g.get(n, torch.zeros_like(fx.sd[n])) for g in per_rank) / world
                assert n in got or (r == 1 and n == "unknown_token"), n
                if n in got:
                    assert torch.allclose(got[n], mean, rtol=1e-5, atol=1e-7), (r, n)


def test_shard_groups_partitions_rows_by_video_group():
    from mesm_amd.ddp import shard_groups
    fx, batch = _tiny_global_batch()
    N = sum(GROUPS)
    batch["qid"] = list(range(N))
    s0, s1 = shard_groups(batch, 0, 2), shard_groups(batch, 1, 2)
    assert s0["num_clips"].tolist() == [2, 2, 1] and s1["num_clips"].tolist() == [1, 1]
    assert s0["qid"] == [0, 1, 3, 4, 6] and s1["qid"] == [2, 5]
    assert torch.equal(s0["video_feat"], batch["video_feat"][[0, 1, 3, 4, 6]])
    assert len(s1["norm_span"]) == 2 and torch.equal(s1["norm_span"][1]["spans"], batch["norm_span"][5]["spans"])
    assert torch.equal(s1["saliency_label"], batch["saliency_label"][[2, 5]])
    with pytest.raises(ValueError):  # 5 groups over 3 ranks leaves rank 2 with one group: no negatives
        shard_groups(batch, 2, 3)


def test_buckets_cover_the_flat_buffer():
    from mesm_amd.ddp import GradReducer
    from mesm_amd.gradbuf import GradBuffer
    m = Toy()
    gb = GradBuffer([(n, p) for n, p in m.named_parameters()])
    gb.ensure(torch.device("cpu"))
    red = GradReducer(gb, n_buckets=3)
    red._make_buckets()
    assert red.buckets[0][0] == 0 and red.buckets[-1][1] == gb.numel
    for (l0, h0, _), (l1, _, _) in zip(red.buckets[:-1], red.buckets[1:]):
        assert h0 == l1


def _agree_worker(rank, world, port, out):
    _init(rank, world, port)
    from mesm_amd.ddp import flat_checksum, ranks_agree
    torch.manual_seed(3)
    flat = torch.randn(1000)
    same = ranks_agree(flat)                       # identical buffers on both ranks
    flat2 = flat.clone()
    if rank == 1:
        flat2[517] += 1e-3                         # one element differs on one rank
    diff = ranks_agree(flat2)
    perm = flat.clone()
    if rank == 1:
        perm = perm.flip(0)                        # same multiset, other order: the position-weighted term sees it
    out.put((rank, same, diff, ranks_agree(perm), flat_checksum(flat).tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_rank_agreement_check():
    """the check bench.py runs after its first data-parallel step (every rank must hold the same reduced gradient
    buffer): agrees on equal buffers, trips on a single differing element and on a permuted buffer"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_agree_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in ps:
        p.start()
    res = [out.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, diff, perm, cs in res:
        assert same == (True, 0.0)
        assert diff[0] is False and diff[1] > 0
        assert perm[0] is False
