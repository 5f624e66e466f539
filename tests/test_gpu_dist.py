"""The data-parallel step on real kernels.  One MI355X is enough for the logic: two ranks share
the card over gloo (CUDA tensors staged through the host); with >= 2 devices the same test runs
over nccl = RCCL with the gradient exchange behind the C ABI (mt_allreduce_sum_f32).  A
world-size-1 RCCL communicator exercises the C-ABI plumbing on any box."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)


def _in_turn(rank, world, fn):
    """Ranks that SHARE one card run their compute one after the other (the collectives stay concurrent).
    Why (round 4, scratch/r4_two_proc_fwd.py / r4_two_proc_repro.py): with two or three processes time-slicing one
    MI355X, about one forward pass in thirty returned a few different pixels -- always first in a launch of the fp32 VALU
    kernels of csrc/conv_tiny.hip (one accumulator register of a 16-lane quarter wave), never with MT355_TINY=0 (117
    passes), never in a single process (bit-identical over 40 passes), which is how the product runs: one process per
    GPU.  It made this file fail twice in six runs before the ranks took turns; the cause was not found in the kernels
    (no uninitialised LDS / register / memory read, barriers in place) and is recorded as an open observation in
    DESIGN.md section 6."""
    out = None
    for r in range(world):
        if r == rank:
            out = fn()
            torch.cuda.synchronize()
        dist.barrier(group=_turn_group())        # (a group of its own: fn may post collectives on the default one)
    return out


_TURN = []


def _turn_group():
    if not _TURN:
        _TURN.append(dist.new_group(backend='gloo'))
    return _TURN[0]


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rank(rank, world, port, backend, out, inplace=True):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(here, 'golden')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import seeded as S
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    ndev = torch.cuda.device_count()
    dev = torch.device('cuda', rank % ndev)
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    _turn_group()
    import master_thesis_amd as mt
    from master_thesis_amd import dist as mdist
    from master_thesis_amd import engine as E
    E.INPLACE_GRADS = inplace       # False = MT355_INPLACE_GRADS=0: gradients are returned to autograd, which accumulates them
    model = mt.DFPN(mt.VGGFeatures(device=dev), **KW)
    sd = S.add_buffers(S.seeded_state_dict(S.dfpn_shapes(), seed=rank))     # ranks start apart ...
    model.load_state_dict(sd)
    model = model.to(dev)
    opt = model.configure_optimizers()['optimizer']
    mdist.init_comm(dev)             # the transport bootstrap is a collective of its own: done here, by all ranks together
    mdist.broadcast_parameters(opt)                                         # ... and agree after the broadcast
    (x, m), y, info = S.synthetic_batch(2, 2, 64, 64, seed=31)             # clip r belongs to rank r
    sl = slice(rank, rank + 1)
    batch = ((x[sl].to(dev), m[sl].to(dev)), y[sl].to(dev),
             tuple(i[sl].to(dev) if torch.is_tensor(i) else i for i in info))
    def own_step():
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        return opt.flat_g.clone()
    local = _in_turn(rank, world, own_step)         # this rank's own gradient (nothing exchanged: not armed)

    # the same step again with the exchange overlapped with the backward: the four conv stacks of the DFPN report
    # their windows of the flat buffer as the backward finishes them (dist.grads_ready; the all-reduces are posted
    # asynchronously, in the same order on every rank), sync_gradients sends the rest
    def armed_step():
        opt.zero_grad()
        mdist.arm_overlap(opt)
        loss = model.training_step(batch, 0)
        loss.backward()
        return len(mdist._pend.ranges)
    early = _in_turn(rank, world, armed_step)
    mdist.sync_gradients(opt)                       # the real exchange (RCCL through the C ABI on nccl)
    flat = (opt.flat_g * opt.grad_scale).clone()
    both = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    want = sum(both) / world
    err = float((flat - want).abs().max() / want.abs().max())
    opt.step()
    ps = [torch.empty_like(opt.flat_p) for _ in range(world)]
    dist.all_gather(ps, opt.flat_p)
    same = all(torch.equal(ps[0], p) for p in ps[1:])
    out[rank] = (err, bool(same), float(local.norm()), mdist.transport(), early)
    mdist.shutdown()
    dist.destroy_process_group()


def _run(world, backend, inplace=True):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank, args=(world, _free_port(), backend, out, inplace), nprocs=world, join=True)
    res = dict(out)
    assert len(res) == world
    for r, (err, same, norm, transport, early) in res.items():
        # (round 4: the step is bit-reproducible run to run -- resize backward in gather form, loss sums in a fixed order,
        # scratch/r4_chain_determinism.py: 0 of 15 passes of either step differ bitwise -- so the exchanged gradient is the
        # mean of the ranks' own to rounding of the one addition; rounds 1-3 allowed 1e-3 here for float atomics.  A window
        # that was dropped or reduced twice would be off by O(1).)
        assert norm > 0 and err < 1e-6 and same and early == (4 if inplace else 0), (r, err, same, norm, early)
    return res


def test_two_ranks_one_card_gloo():
    """DP step on HIP kernels: the exchanged flat gradient is the mean of the ranks' own gradients
    and every rank holds the same parameters after Adam."""
    res = _run(2, 'gloo')
    assert res[0][3] == 'torch.distributed/gloo'


def test_two_ranks_returned_gradients_are_not_sent_early():
    """MT355_INPLACE_GRADS=0: the conv stacks RETURN their weight gradients and autograd adds them into
    the flat buffer after the stack's backward hook has run.  Nothing may go on the wire from the hook then (the
    window is not final: the early all-reduce would send stale data and race with the add); sync_gradients sends
    everything, and the exchanged gradient is still the mean of the ranks' own."""
    _run(2, 'gloo', inplace=False)


def _rank_graphed(rank, world, port, out, steps):
    """bench.py's N > 1 step (run_rank / make_steps.graphed): forward + backward replayed from the captured hipGraph, then the
    gradient exchange and the Adam launch outside the graph."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(here, 'golden')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import seeded as S
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dev = torch.device('cuda', rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _turn_group()
    import master_thesis_amd as mt
    from master_thesis_amd import dist as mdist
    from master_thesis_amd.graph import GraphedTrainStep
    model = mt.DFPN(mt.VGGFeatures(device=dev), **KW)
    model.load_state_dict(S.add_buffers(S.seeded_state_dict(S.dfpn_shapes(), seed=0)))          # (the frozen VGG is the same everywhere)
    model = model.to(dev)
    opt = model.configure_optimizers()['optimizer']
    opt.flat_p.add_(0.01 * rank)                                                                # the trainable ones start apart ...
    mdist.init_comm(dev)
    mdist.broadcast_parameters(opt)                                                             # ... and agree after this
    (x, m), y, info = S.synthetic_batch(world, 2, 64, 64, seed=31)                             # clip r belongs to rank r
    sl = slice(rank, rank + 1)
    batch = ((x[sl].to(dev), m[sl].to(dev)), y[sl].to(dev), tuple(i[sl].to(dev) if torch.is_tensor(i) else i for i in info))
    g = _in_turn(rank, world, lambda: GraphedTrainStep(model, opt, batch, include_optimizer=False))
    losses = []
    for _ in range(steps):
        loss = _in_turn(rank, world, g.step)       # (compute in turn on the shared card: _in_turn; the exchange is concurrent)
        mdist.sync_gradients(opt)
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    ps = [torch.empty_like(opt.flat_p) for _ in range(world)]
    dist.all_gather(ps, opt.flat_p)
    out[rank] = (losses, opt.flat_p.cpu(), all(torch.equal(ps[0], q) for q in ps[1:]), opt.grad_scale)
    mdist.shutdown()
    dist.destroy_process_group()


def test_bench_multi_rank_step_graph_then_exchange_then_adam():
    """The N > 1 branch of bench.py on one card (two ranks over gloo): three steps of graph replay -> sync_gradients -> Adam
    outside the graph.  Every rank holds bit-identical parameters afterwards, and they are the parameters ONE process
    reaches when it does what data parallelism defines (__main__.py:20,69: DDP averages the RANKS'
    gradients): per step the mean of the two clips' own gradients, each from its own eager backward, then Adam.  (A
    single-rank step on the CONCATENATED batch is a different function: masked_l1(reduction='sum') divides the summed
    error by the batch's summed mask, utils.py:166-169, so its loss is not the mean of the clips' losses; the rank mean
    of the losses is checked against the clips' own losses instead.)"""
    import sys
    import seeded as S
    steps, world = 3, 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_graphed, args=(world, _free_port(), out, steps), nprocs=world, join=True)
    res = dict(out)
    assert len(res) == world and all(r[2] for r in res.values()) and all(r[3] == 0.5 for r in res.values())
    assert torch.equal(res[0][1], res[1][1])
    # the single-process reference
    import master_thesis_amd as mt
    dev = torch.device('cuda', 0)
    model = mt.DFPN(mt.VGGFeatures(device=dev), **KW)
    model.load_state_dict(S.add_buffers(S.seeded_state_dict(S.dfpn_shapes(), seed=0)))
    model = model.to(dev)
    opt = model.configure_optimizers()['optimizer']
    theta0 = opt.flat_p.clone()
    (x, m), y, info = S.synthetic_batch(world, 2, 64, 64, seed=31)
    clips = []
    for r in range(world):
        sl = slice(r, r + 1)
        clips.append(((x[sl].to(dev), m[sl].to(dev)), y[sl].to(dev), tuple(i[sl].to(dev) if torch.is_tensor(i) else i for i in info)))
    ref_losses = []
    for _ in range(steps):
        opt.zero_grad()
        per_clip = []
        for c in clips:                      # the weight-gradient reduces ADD into the flat buffer: two backward passes sum
            loss = model.training_step(c, 0)
            loss.backward()
            per_clip.append(float(loss))
        opt.grad_scale = 1.0 / world
        opt.step()
        ref_losses.append(per_clip)
    torch.cuda.synchronize()
    upd = (opt.flat_p - theta0).double().cpu()
    err = float((res[0][1].double() - opt.flat_p.double().cpu()).norm() / upd.norm())
    loss_err = max(abs(res[r][0][s_] - ref_losses[s_][r]) / abs(ref_losses[s_][r]) for r in range(world) for s_ in range(steps))
    print(f'DP (graph + exchange + Adam, 2 ranks) vs one process: parameters {err:.2e} of the update, per-clip losses {loss_err:.2e}')
    # (an entry whose gradient is at the rounding level moves by +-lr in either run: L2 over all entries, not the maximum)
    assert err < 2e-2 and loss_err < 1e-4, (err, loss_err)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs')
def test_two_ranks_two_cards_rccl_c_abi():
    res = _run(2, 'nccl')
    assert res[0][3].startswith('rccl')


def test_rccl_communicator_world_size_one():
    """mt_comm_unique_id / mt_comm_init / mt_allreduce_sum_f32 / mt_comm_destroy on one GPU: a
    one-rank all-reduce leaves the buffer unchanged and is ordered with the streams around it."""
    from master_thesis_amd import _lib as L
    from master_thesis_amd import dist as mdist
    torch.cuda.set_device(0)
    comm = mdist.RcclComm(mdist.RcclComm.unique_id(), 0, 1, 'cuda:0')
    assert (comm.rank, comm.world) == (0, 1)
    x = torch.arange(5_000_000, device='cuda:0', dtype=torch.float32)
    y = x * 2                                    # queued on the current stream before the collective
    comm.allreduce_sum_(y, bucket_floats=1_000_000)
    z = y + 1                                    # queued after it: must see the reduced buffer
    torch.cuda.synchronize()
    assert torch.equal(z, x * 2 + 1)
    with pytest.raises(L.MT355Error):
        L.call('mt_allreduce_sum_f32', None, L.ptr(y), 10, L.stream_ptr())
    comm.destroy()
