"""The data-parallel exchange on CPU: two gloo ranks, clips sharded by rank,
one bucketed SUM all-reduce of the flat gradient buffer, 1/world folded into
the optimizer's grad_scale."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class _Opt:          # the two attributes sync_gradients() touches on FlatAdam
    def __init__(self, n, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.flat_g = torch.rand((n,), generator=g)
        self.flat_p = torch.full((n,), float(rank))
        self.grad_scale = 1.0


def _worker(rank, world, port, n, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from master_thesis_amd import dist as mdist
    opt = _Opt(n, rank)
    mdist.broadcast_parameters(opt)
    local = opt.flat_g.clone()
    mdist.BUCKET_FLOATS = 1000          # several buckets, ragged tail
    mdist.allreduce_flat_(opt.flat_g, 1000)
    opt.grad_scale = 1.0 / world
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered)
    ok = torch.allclose(opt.flat_g, want) and opt.grad_scale == 0.5 and float(opt.flat_p.max()) == 0.0
    # weak scaling bookkeeping: every rank owns its own clips (seed = base + rank)
    from master_thesis_amd.synthetic import make_batch
    (x, _), _, _ = make_batch(1, 2, 16, 16, 'cpu', seed=1234 + rank)
    xs = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(xs, x)
    ok = ok and not torch.equal(xs[0], xs[1])
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce():
    world, n = 2, 4321
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
