"""Data-parallel reduction on CPU with gloo, world_size 2: the flat gradient buffer +
GradReducer must leave every rank with the MEAN over ranks of the per-rank gradients, both
on the learning step (everything reduced at the end) and on later steps (bucket collectives
launched from inside backward)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


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


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from mesm_amd.ddp import GradReducer, init_process_group_from_env
    from mesm_amd.gradbuf import GradBuffer
    init_process_group_from_env(torch.device("cpu"))
    m = Toy()
    gb = GradBuffer([(n, p) for n, p in m.named_parameters()])
    gb.ensure(torch.device("cpu"))
    red = GradReducer(gb, n_buckets=3)
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
def test_gloo_world2_mean_of_rank_gradients():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for step in range(3):
        want = [_local_grads(step, r) for r in range(world)]
        for r in range(world):
            got = out[r][step]
            for n in ("a", "b", "c", "d"):
                mean = sum(w[n] for w in want) / world
                assert torch.allclose(got[n], mean, atol=1e-6), (step, r, n)
            assert got["unused"] is None  # never touched -> stays None, like under autograd


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
