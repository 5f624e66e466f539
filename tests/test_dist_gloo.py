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


def _worker_flat_adam(rank, world, port, out):
    """The real FlatAdam + the real sync_gradients on CPU tensors (gloo)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from master_thesis_amd import dist as mdist
    from master_thesis_amd.optim import FlatAdam
    torch.manual_seed(7 + rank)                      # ranks start from DIFFERENT parameters
    params = [torch.nn.Parameter(torch.randn(5, 3, 3, 3)), torch.nn.Parameter(torch.randn(5)),
              torch.nn.Parameter(torch.randn(33, 7))]
    opt = FlatAdam(params, lr=1e-3)
    mdist.broadcast_parameters(opt)
    gathered = [torch.empty_like(opt.flat_p) for _ in range(world)]
    dist.all_gather(gathered, opt.flat_p)
    ok = torch.equal(gathered[0], gathered[1])       # ... and agree after the broadcast
    ok = ok and all(p.data_ptr() == opt.flat_p.data_ptr() + 4 * off for p, (off, k) in zip(params, opt._slices))
    g = torch.Generator().manual_seed(100 + rank)
    for p in params:                                 # "backward": a rank-local gradient in the flat windows
        p.grad.copy_(torch.rand(p.shape, generator=g))
    local = opt.flat_g.clone()
    mdist.BUCKET_FLOATS = 100                        # several buckets, ragged tail
    mdist.sync_gradients(opt)
    both = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    ok = ok and torch.allclose(opt.flat_g * opt.grad_scale, (both[0] + both[1]) / 2) and opt.grad_scale == 0.5
    ok = ok and mdist.transport() == 'torch.distributed/gloo'
    # the exchange overlapped with the backward: windows of the flat buffer reported by the engine's hook go on
    # the wire early (here: the last parameter, as a backward would finish it first), sync_gradients sends the rest
    from master_thesis_amd import engine as E
    ok = ok and E.GRADS_READY is mdist.grads_ready
    for early in ([[2]], [[2], [0, 1]], []):
        for p in params:
            p.grad.copy_(torch.rand(p.shape, generator=g))
        local = opt.flat_g.clone()
        mdist.arm_overlap(opt)
        for group in early:
            E.GRADS_READY([params[i] for i in group])
        ok = ok and len(mdist._pend.ranges) == len(early)
        mdist.sync_gradients(opt)
        dist.all_gather(both, local)
        ok = ok and torch.allclose(opt.flat_g, both[0] + both[1]) and mdist._pend.flat is None
    # a window reported twice in one backward is an error, not a silent double reduction
    mdist.arm_overlap(opt)
    E.GRADS_READY([params[2]])
    try:
        E.GRADS_READY([params[2]])
        ok = False
    except RuntimeError:
        pass
    mdist.sync_gradients(opt)
    # ... and without arm_overlap the hook does nothing
    E.GRADS_READY([params[2]])
    ok = ok and mdist._pend.flat is None and not mdist._pend.ranges
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_two_rank_sync_gradients_with_flat_adam():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_flat_adam, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_flat_adam_state_dict_layouts():
    """load_state_dict accepts FlatAdam's own layout and torch.optim.Adam's (reference / Lightning
    checkpoints), restores the device-side {step, lr}, and re-adopts a parameter that was moved."""
    from master_thesis_amd.optim import FlatAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(6))]
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=5e-4)
    for p in ref.param_groups[0]['params']:
        p.grad = torch.randn_like(p)
    ref.step()
    ref.step()
    opt = FlatAdam(ps, lr=1e-3)
    opt.load_state_dict(ref.state_dict())
    assert opt.step_count == 2 and opt.param_groups[0]['lr'] == 5e-4
    assert opt.dev_state.tolist() == [2.0, 5e-4] or abs(opt.dev_state[1].item() - 5e-4) < 1e-10
    st = ref.state_dict()['state']
    assert torch.equal(opt.exp_avg[:12].view(4, 3), st[0]['exp_avg']) and torch.equal(opt.exp_avg_sq[12:], st[1]['exp_avg_sq'])
    opt2 = FlatAdam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-3)
    opt2.load_state_dict(opt.state_dict())
    assert opt2.step_count == 2 and torch.equal(opt2.exp_avg, opt.exp_avg) and float(opt2.dev_state[0]) == 2.0
    # a parameter that no longer aliases the flat buffer is re-adopted
    ps[1].data = ps[1].data.clone() * 2
    assert opt.check_aliasing() == 1
    assert ps[1].data_ptr() == opt.flat_p.data_ptr() + 4 * 12 and torch.equal(opt.flat_p[12:], ps[1].detach().reshape(-1))
    try:
        opt.load_state_dict({'state': {0: {}}, 'param_groups': ref.state_dict()['param_groups']})
        raise AssertionError('a state dict over a different parameter set must be refused')
    except ValueError:
        pass
