"""Data-parallel gradient exchange: one process per GPU, clips
sharded over ranks, ONE sum-all-reduce per step over the flat fp32 gradient
buffer of the parameters that actually receive gradients (DFPN 55.7 MB, CHN
58.8 MB).  The 1/world averaging is folded into the Adam kernel
(FlatAdam.grad_scale).

On GPUs the collective goes through the C ABI: libmt355 owns an RCCL
communicator (``mt_comm_init``: xGMI underneath) and ``mt_allreduce_sum_f32``
enqueues the buckets on a side HIP stream that waits for the backward through
an event; the Adam launch waits for the buckets through another.  Nothing
synchronises with the host.  ``torch.distributed`` is the bootstrap only (it
carries RCCL's 128-byte unique id to the ranks) -- and the whole transport for
CPU tensors (gloo: the tests) or when ``MT355_COMM=torch``."""
import ctypes as C
import os
import sys

import torch
import torch.distributed as dist

from . import _lib as L

BUCKET_FLOATS = 4 * 1024 * 1024      # 16 MB buckets: a handful of large collectives
_comm = None                         # the process's RcclComm (one GPU per process)
_comm_failed = None


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rccl_library_path():
    """The RCCL build this process already maps (torch's), else the loader default."""
    p = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
    return p if os.path.exists(p) else ''


class RcclComm:
    """An RCCL communicator owned by libmt355 + the side stream its collectives run on."""

    def __init__(self, id_bytes, rank, world, device):
        assert len(id_bytes) == 128
        self.device = torch.device(device)
        self.handle = C.c_void_p()
        buf = C.create_string_buffer(bytes(id_bytes), 128)
        with torch.cuda.device(self.device):
            L.call('mt_comm_init', rccl_library_path().encode(), C.cast(buf, C.c_void_p), rank, world, C.byref(self.handle))
            self.stream = torch.cuda.Stream(device=self.device)
        r, w = L.i32(), L.i32()
        L.call('mt_comm_info', self.handle, C.byref(r), C.byref(w))
        self.rank, self.world = r.value, w.value
        assert (self.rank, self.world) == (rank, world), (self.rank, self.world, rank, world)
        self.ready = torch.cuda.Event()
        self.done = torch.cuda.Event()

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        L.call('mt_comm_unique_id', rccl_library_path().encode(), C.cast(buf, C.c_void_p))
        return bytes(buf.raw)

    def enqueue_range_(self, flat, lo, hi, bucket_floats=None):
        """SUM all-reduce of flat[lo:hi] in buckets on the side stream, after everything already queued on the
        current stream.  Does NOT make the current stream wait: pair with join()."""
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        bucket_floats = bucket_floats or BUCKET_FLOATS
        cur = torch.cuda.current_stream(self.device)
        self.ready.record(cur)
        self.stream.wait_event(self.ready)
        base = flat.data_ptr()
        for off in range(lo, hi, bucket_floats):
            k = min(bucket_floats, hi - off)
            L.call('mt_allreduce_sum_f32', self.handle, C.c_void_p(base + 4 * off), k, C.c_void_p(self.stream.cuda_stream))

    def join(self, flat):
        """The current stream continues after the last bucket enqueued so far."""
        cur = torch.cuda.current_stream(self.device)
        self.done.record(self.stream)
        cur.wait_event(self.done)
        flat.record_stream(self.stream)

    def allreduce_sum_(self, flat, bucket_floats=None):
        """In-place SUM all-reduce of a 1-D fp32 CUDA tensor: the buckets run on the side stream
        after everything already queued on the current stream; the current stream continues after
        the last bucket.  Asynchronous with respect to the host."""
        self.enqueue_range_(flat, 0, flat.numel(), bucket_floats)
        self.join(flat)

    def destroy(self):
        if self.handle:
            torch.cuda.synchronize(self.device)
            L.call('mt_comm_destroy', self.handle)
            self.handle = C.c_void_p()


def init_comm(device=None):
    """Create this rank's RcclComm (idempotent).  Needs an initialised torch.distributed group of
    any backend for the bootstrap.  Returns None -- after saying why on stderr -- when RCCL cannot
    be used (MT355_COMM=torch, no CUDA device, RCCL refuses): the exchange then goes through
    torch.distributed."""
    global _comm, _comm_failed
    if _comm is not None or _comm_failed is not None:
        return _comm
    if os.environ.get('MT355_COMM', 'rccl') != 'rccl' or not torch.cuda.is_available() or not dist.is_initialized():
        _comm_failed = 'disabled'
        return None
    device = torch.device(device if device is not None else ('cuda', torch.cuda.current_device()))
    rank, world = dist.get_rank(), dist.get_world_size()
    # Every rank runs the SAME sequence of bootstrap collectives whatever fails locally:
    # (1) all-gather of (host, device uuid / index): RCCL refuses two ranks on one device, so that case is
    #     detected here instead of by an init error; (2) broadcast of (id | None, error); (3) MIN over "my
    #     communicator came up".  A rank that raised in between still takes part in the next collective.
    me = (os.uname().nodename, _device_identity(device))
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    shared = len(set(everyone)) < world
    box = [None, None]
    if rank == 0 and not shared:
        try:
            box[0] = RcclComm.unique_id()
        except Exception as e:   # noqa: BLE001 - reported to every rank through the broadcast below
            box[1] = repr(e)
    dist.broadcast_object_list(box, src=0)
    if shared:
        _comm_failed = 'several ranks share one device'
    elif box[0] is None:
        _comm_failed = f'rank 0 could not create an RCCL id: {box[1]}'
    else:
        try:
            _comm = RcclComm(box[0], rank, world, device)
        except Exception as e:   # noqa: BLE001 - every rank must end up on the SAME transport
            _comm_failed = repr(e)
    if _comm is None and (rank == 0 or (not shared and box[0] is not None)):
        print(f'[mt355] rank {rank}: RCCL communicator not used ({_comm_failed}); using torch.distributed',
              file=sys.stderr, flush=True)
    # one rank failing moves everybody to the fallback (a split transport would deadlock)
    ok = torch.tensor([1 if _comm is not None else 0], device=device if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok) == 0 and _comm is not None:
        _comm.destroy()
        _comm, _comm_failed = None, 'another rank failed'
    return _comm


def _device_identity(device):
    """Something two ranks share exactly when they sit on the same card (uuid when torch exposes it)."""
    try:
        return str(torch.cuda.get_device_properties(device).uuid)
    except Exception:            # noqa: BLE001 - older builds have no uuid field
        vis = os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('CUDA_VISIBLE_DEVICES', ''))
        return f'{vis}#{torch.device(device).index}'


def transport():
    """'rccl (C ABI)' | 'torch.distributed/<backend>' | 'none': what sync_gradients uses."""
    if not is_dist():
        return 'none'
    return 'rccl (C ABI, mt_allreduce_sum_f32)' if _comm is not None else f'torch.distributed/{dist.get_backend()}'


def shutdown():
    global _comm, _comm_failed
    if _comm is not None:
        _comm.destroy()
    _comm, _comm_failed = None, None


def allreduce_flat_(flat, bucket_floats=None, async_op=False):
    """In-place SUM all-reduce of a 1-D tensor in contiguous buckets through torch.distributed
    (CPU tensors / the fallback transport)."""
    if not is_dist():
        return []
    bucket_floats = bucket_floats or BUCKET_FLOATS
    works = []
    for off in range(0, flat.numel(), bucket_floats):
        works.append(dist.all_reduce(flat[off:off + bucket_floats], op=dist.ReduceOp.SUM, async_op=True))
    if not async_op:
        for w in works:
            w.wait()
    return works


# ---- the exchange overlapped with the backward --------------------------------------------------
# The flat gradient buffer follows the parameter order, the backward finishes it from the back: each conv stack
# (engine.ConvStackFn: FlowEstimator 256 / 64, the correlation mixer, SeparableConv4d -- 21 / 21 / 9 / 7 MB of the
# DFPN's 55.7 MB) reports its parameters when its backward has enqueued their last weight-gradient reduce, and that
# window of the buffer goes on the wire while the backward of the stacks in front of it still runs.  Eager steps
# only (what a Lightning loop runs): inside a captured graph there is no hook point, the buckets follow the replay.
# The loop arms ONE backward with arm_overlap(optimizer) and closes it with sync_gradients(optimizer), which sends
# whatever was not reported and makes the stream wait for all of it.  MT355_DP_OVERLAP=0: everything at the end.
OVERLAP = os.environ.get('MT355_DP_OVERLAP', '1') != '0'


class _Pending:
    flat = None        # the armed optimizer's flat gradient buffer
    ranges = []        # [lo, hi) windows already on the wire
    works = []         # torch.distributed work handles (fallback transport)


_pend = _Pending()


def arm_overlap(optimizer):
    """The next backward may send finished windows of ``optimizer.flat_g`` early (no-op when not distributed)."""
    if not (OVERLAP and is_dist()):
        return
    if _pend.flat is not None and _pend.ranges:
        raise RuntimeError('arm_overlap: the previous backward was never closed with sync_gradients()')
    _pend.flat, _pend.ranges, _pend.works = optimizer.flat_g, [], []


def _send(flat, lo, hi):
    for a, b in _pend.ranges:
        if lo < b and a < hi:
            raise RuntimeError(f'gradient window [{lo}, {hi}) was already exchanged in this backward (a module ran twice, or '
                               'gradients are accumulated over several backward passes): set MT355_DP_OVERLAP=0')
    comm = init_comm(flat.device) if flat.is_cuda else None
    if comm is not None:
        comm.enqueue_range_(flat, lo, hi)
    else:
        if flat.is_cuda and dist.get_backend() != 'nccl':
            torch.cuda.current_stream().synchronize()        # gloo stages CUDA tensors through the host
        for off in range(lo, hi, BUCKET_FLOATS):
            _pend.works.append(dist.all_reduce(flat[off:min(hi, off + BUCKET_FLOATS)], op=dist.ReduceOp.SUM, async_op=True))
    _pend.ranges.append((lo, hi))


def grads_ready(params):
    """engine hook: the backward has enqueued the last gradient launch of ``params`` (one conv stack)."""
    flat = _pend.flat
    if flat is None:
        return
    if flat.is_cuda and torch.cuda.is_current_stream_capturing():
        return
    gs = [p.grad for p in params if p is not None and p.requires_grad]
    if not gs or any(not getattr(g, '_mt_flat', False) or g._base is not flat for g in gs):
        return                                   # (gradients that autograd still has to accumulate: not final yet)
    base = flat.data_ptr()
    lo = min((g.data_ptr() - base) // 4 for g in gs)
    hi = max((g.data_ptr() - base) // 4 + g.numel() for g in gs)
    if sum(g.numel() for g in gs) == hi - lo:    # one contiguous window
        _send(flat, lo, hi)


def sync_gradients(optimizer):
    """Average the gradients of a FlatAdam optimizer across ranks (sum here, 1/world in Adam): sends every
    window of the flat buffer the armed backward has not sent already, then orders the stream behind all of it."""
    if not is_dist():
        optimizer.grad_scale = 1.0
        return
    flat = optimizer.flat_g
    if _pend.flat is not flat:
        _pend.flat, _pend.ranges, _pend.works = flat, [], []
    pos = 0
    for a, b in sorted(_pend.ranges) + [(flat.numel(), flat.numel())]:
        if pos < a:
            _send(flat, pos, a)
        pos = max(pos, b)
    comm = init_comm(flat.device) if flat.is_cuda else None
    if comm is not None:
        comm.join(flat)                           # stream-ordered: no host synchronisation
    for w in _pend.works:
        w.wait()
    _pend.flat, _pend.ranges, _pend.works = None, [], []
    optimizer.grad_scale = 1.0 / dist.get_world_size()


def _install_hook():
    from . import engine as E
    E.GRADS_READY = grads_ready


_install_hook()


def broadcast_parameters(optimizer, src=0):
    """Every rank starts from rank ``src``'s parameters."""
    if is_dist():
        dist.broadcast(optimizer.flat_p, src=src)
        from . import engine as E
        E.bump_weight_epoch()           # packed weight images made before the broadcast are stale
