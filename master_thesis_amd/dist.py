"""Data-parallel gradient exchange: one process per GPU, clips
sharded over ranks, ONE sum-all-reduce per step over the flat fp32 gradient
buffer of the parameters that actually receive gradients (DFPN 55.7 MB, CHN
58.8 MB), issued in a few large buckets through torch.distributed (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests).  The 1/world
averaging is folded into the Adam kernel (FlatAdam.grad_scale)."""
import torch
import torch.distributed as dist

BUCKET_FLOATS = 4 * 1024 * 1024      # 16 MB buckets: a handful of large collectives


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def allreduce_flat_(flat, bucket_floats=BUCKET_FLOATS, async_op=False):
    """In-place SUM all-reduce of a 1-D tensor in contiguous buckets."""
    if not is_dist():
        return []
    works = []
    for off in range(0, flat.numel(), bucket_floats):
        works.append(dist.all_reduce(flat[off:off + bucket_floats], op=dist.ReduceOp.SUM, async_op=True))
    if not async_op:
        for w in works:
            w.wait()
    return works


def sync_gradients(optimizer):
    """Average the gradients of a FlatAdam optimizer across ranks."""
    if not is_dist():
        optimizer.grad_scale = 1.0
        return
    # The backward (or the replayed step graph) is drained first: the collective then starts
    # on an idle queue instead of being ordered behind ~1.6 k queued launches through a
    # cross-stream event (with gloo on CUDA tensors that wait was observed to take seconds).
    # Cost: one host wake-up per step (tens of microseconds against a ~200 ms step).
    if optimizer.flat_g.is_cuda:
        torch.cuda.current_stream().synchronize()
    allreduce_flat_(optimizer.flat_g)
    optimizer.grad_scale = 1.0 / dist.get_world_size()


def broadcast_parameters(optimizer, src=0):
    if is_dist():
        dist.broadcast(optimizer.flat_p, src=src)
