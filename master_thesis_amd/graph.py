"""Replay a whole training step (forward + losses + backward [+ Adam]) from ONE
captured hipGraph: the DFPN / CHN steps are chains of ~660 / ~2300 launches.
Measured in round 4 (bench.py: ``ms_per_step`` is the replay, ``eager_ms_per_step``
the same step launched from Python, same protocol): DFPN 99.2 ms replayed / 100.1 ms
eager, CHN 86.1 / 88.4 ms -- the eager path, which is what a plain Lightning loop
runs, is 1-3 % behind on an idle host.  The graph keeps the step independent of
host jitter, which matters when 8 ranks wait for each other every step.  The batch
lives in static tensors; ``step(batch)`` copies a new batch in and replays."""
import gc

import torch


def _live_autograd_tensors():
    """CUDA tensors that still carry a grad_fn: an autograd graph that survives into the capture
    would bring its AccumulateGrad nodes (bound to the warm-up stream) along."""
    out = []
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda and o.grad_fn is not None:
                out.append(o)
        except Exception:      # noqa: BLE001 - objects half way through destruction
            pass
    return out


def _copy_batch_(dst, src):
    (dx, dm), dy, dinfo = dst
    (sx, sm), sy, sinfo = src
    dx.copy_(sx); dm.copy_(sm); dy.copy_(sy)
    for d, s in zip(dinfo, sinfo):
        if torch.is_tensor(d):
            d.copy_(s)


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_batch, include_optimizer=True, warmup=2):
        self.model, self.opt = model, optimizer
        self.batch = example_batch
        self.include_optimizer = include_optimizer
        before = {id(t) for t in _live_autograd_tensors()}     # (the caller's own business)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up on a side stream (allocator, lazy inits)
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # nothing of the warm-up steps may survive into the capture: a live autograd graph would
        # carry AccumulateGrad nodes bound to the side stream (round-1 GPU faults).  Cycles are
        # collected, then it is CHECKED: a tensor with a grad_fn that appeared during the warm-up and
        # is still alive (a logger that keeps attached losses, a hook that stores activations, ...)
        # must be dropped or detached by whoever holds it.
        gc.collect()
        live = [t for t in _live_autograd_tensors() if id(t) not in before]
        if live:
            raise RuntimeError(
                f'GraphedTrainStep: {len(live)} CUDA tensor(s) of the warm-up steps still carry an autograd graph '
                f'(shapes {[tuple(t.shape) for t in live[:4]]}); detach or drop them before capturing')
        # the weight packs must be part of the captured step: with the optimizer in the graph its
        # step re-packs them; without it the graph starts with a re-pack.  The device tables of the
        # batched re-pack are (re)built here, eagerly: a capture cannot copy them from the host.
        from . import engine as E
        E.repack_all()
        E.drop_epoch_cache()       # padded biases / fp32 packs are re-made inside the capture
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad()
        with torch.cuda.graph(self.graph):
            if not include_optimizer:
                E.repack_all()             # the replay starts from the parameters as they are then
            self.opt.zero_grad()
            self.loss = self.model.training_step(self.batch, 0)
            self.loss.backward()
            if include_optimizer:
                self.opt.step()
        # the static loss scalar is handed out detached: the captured step's autograd graph must not
        # outlive the capture (a later capture in the same process would trip over it)
        self.loss = self.loss.detach()
        if include_optimizer:
            self.opt.step_count -= 1           # capturing records the update, it does not run it
        # the captured launches point at these buffers and tables: they live as long as the graph
        self._pack_state = E.pack_state()

    def _eager(self):
        self.opt.zero_grad()
        loss = self.model.training_step(self.batch, 0)
        loss.backward()
        if self.include_optimizer:
            self.opt.step()
        return loss

    def step(self, batch=None):
        if batch is not None and batch is not self.batch:
            _copy_batch_(self.batch, batch)
        if self.include_optimizer:
            self.opt.sync_device_lr()          # an LR scheduler / a resumed checkpoint changed it
        self.graph.replay()
        if self.include_optimizer:
            self.opt.step_count += 1
        return self.loss
