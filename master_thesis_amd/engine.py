"""Host side of the convolution engine: NHWC4 tensors, weight packing and the
autograd nodes that chain the C-ABI kernels (mt_conv_run / mt_wgrad_run / ...).

Activations inside the engine are float32 torch tensors of shape (N, H, W, Cp)
with Cp = channels padded to a multiple of 4 (padding holds zeros).  Every
heavy op is a libmt355 call on the current HIP stream; torch only owns memory
and the autograd graph.
"""
import ctypes as C
import os

import torch

from . import _lib as L

WEIGHT_EPOCH = 0          # bumped by the optimizer: packed-weight cache key
_pack_cache = {}
PROFILE = None            # list -> every conv-engine launch appends (kernel, flops, ev0, ev1)
GRADS_READY = None        # callable(params): a conv stack's backward has enqueued its last gradient launch (dist.py)
# MFMA arithmetic of the conv engine (include/mt355.h, mt_precision):
#   'f16x3'   2-term fp16 split of power-of-two scaled tensors, 3 partial products:
#             ~7e-8 relative per product (fp32-grade).  Halo-tile kernels only;
#             every other layer of the step falls back to bf16x6.  (default)
#   'bf16x6'  3-term bf16 split, 6 partial products: fp32-equivalent
#   'bf16x3'  2-term bf16 split, 3 partial products: ~4e-6 relative per layer
#   'f32'     fp32-input MFMA
PRECISION = os.environ.get('MT355_PRECISION', 'f16x3')
# The VGG feature passes are never differentiated (model_vgg.py:43 detaches, and
# model_dfpn.py:242,510 run them under no_grad) and feed only smooth functions
# (cosine correlation, L1 of features): they run the cheaper 2-term bf16 split
# (~4e-6 relative per layer, far inside the 1e-3 output bar).  The DFPN aligner
# inside the CHN step is also under no_grad, but its flow goes through a
# NEAREST-neighbour warp of the visibility maps: extra rounding noise there
# flips mask pixels, so it keeps the full precision unless asked otherwise.
PRECISION_VGG = os.environ.get('MT355_PRECISION_VGG', 'bf16x3')
PRECISION_NOGRAD = os.environ.get('MT355_PRECISION_NOGRAD', 'same')
USE_HALO = os.environ.get('MT355_HALO', '1') != '0'   # LDS halo-tile kernel for stride-1 3x3/5x5 layers
_PREC = {'f32': L.PREC_F32, 'bf16x3': L.PREC_BF16X3, 'bf16x6': L.PREC_BF16X6, 'f16x3': L.PREC_F16X3}
_KTAG = {'f32': '', 'bf16x3': ',2', 'bf16x6': ',3', 'f16x3': ',2,f16'}
_HTAG = {'bf16x3': '2,bf16', 'bf16x6': '3,bf16', 'f16x3': '2,f16'}


# arithmetic of the layers the halo kernels do not take (strided, transposed, ups):
# 'same' = PRECISION; 'bf16x6' keeps them on the 3-term bf16 split
PRECISION_GENERIC = os.environ.get('MT355_PRECISION_GENERIC', 'same')


def eff_precision(halo):
    """Precision a launch really uses."""
    if PRECISION == 'f16x3' and not halo and PRECISION_GENERIC != 'same':
        return PRECISION_GENERIC
    return PRECISION


def nograd_precision():
    """Precision of conv stacks run under torch.no_grad (the aligner in CHN)."""
    if PRECISION == 'f32' or PRECISION_NOGRAD == 'same':
        return PRECISION
    return PRECISION_NOGRAD


def vgg_precision():
    if PRECISION == 'f32' or PRECISION_VGG == 'same':
        return PRECISION
    return PRECISION_VGG


class precision:
    """``with precision('bf16x3'): ...`` scoped override."""

    def __init__(self, name):
        assert name in _PREC, name
        self.name = name

    def __enter__(self):
        global PRECISION
        self.old, PRECISION = PRECISION, self.name

    def __exit__(self, *a):
        global PRECISION
        PRECISION = self.old


# blocks a producer / consumer weight-gradient launch aims at: ONE round of the 256 CUs (one 512-thread block per CU).  Round 4, same box,
# DFPN step's weight gradients: 256 blocks 31.4 ms, 512 (two rounds: twice the slabs, twice the pipeline fills) 32.3, 384 38.9, 128 52.8
WGRAD_PC_BLOCKS = int(os.environ.get('MT355_WGRAD_PC_BLOCKS', '256'))
WGRAD_HALO_BLOCKS = int(os.environ.get('MT355_WGRAD_HALO_BLOCKS', '512'))    # the single-role kernels (stride-2 gathers, three-term splits)
# 16-bit terms of dY that enter a weight gradient on the producer / consumer kernel (mt_wgrad.d_terms): 2 = both
# (three partial products, fp32-grade per product); 1 = the leading term only (two products, default since round 3).
# With one term dY is rounded to 11 bits PER PIXEL before it is summed over 10^5 .. 10^6 pixels: independent rounding
# errors, 2e-4 relative on a gradient of pure noise (tests/test_gpu_kernels.py::test_weight_gradient_terms) and not
# measurable on the training step -- flat gradient against the fp64 oracle at the benchmarked geometry 5.6e-5 / 8.2e-5
# (DFPN / CHN) with either setting, the fp32 CPU oracle itself 5.2e-5 / 8.7e-5; trajectories: tests/test_gpu_trajectory.py.
# Forward and backward-DATA products always keep both terms of both operands.  The field is a WISH: the library applies it
# where the kernel has the form (fp16 terms, producer / consumer and stride-2 kernels: mt_wgrad_plan.d_terms).
WGRAD_D_TERMS = int(os.environ.get('MT355_WGRAD_D_TERMS', '1'))
# Launches shared by several layers (the four growth convolutions of a dense block): cover their useful 32 x 32 channel
# products with tiles of freely chosen 32-channel groups (mt_wgrad.tiles, tile_cover below) instead of 64-aligned tiles
# with a skip mask (0: A/B).
WGRAD_TILE_LIST = os.environ.get('MT355_WGRAD_TILE_LIST', '1') != '0'

# ---- which kernel runs: asked from the library, never decided here ------------------------------------------------
# Rounds 1-3 mirrored the C dispatch rules in this file.  Now the library's own
# decision (csrc/dispatch.cpp) is queried through the ABI -- mt_conv_plan_query / mt_wgrad_plan_query -- before a weight
# is packed or a workspace is sized, and the profiling labels are the names it reports.  Plans are cached by geometry
# (a ctypes call per launch would cost the eager path ~2 us x 2300 launches per CHN step).
_conv_plans, _wgrad_plans, _tile_covers = {}, {}, {}
KERNEL_INFO = {}          # profiling label -> (MFMA products per multiply-add, executed / algorithmic MFMA work)


def _conv_plan(d, cin_real, cout_real):
    """(precision name, mt_conv_plan) of the launch ``d`` describes (geometry fields set).  The precision policy is
    the host's (eff_precision: layers the halo kernels do not take may run another split), the kernel is the library's."""
    key = (d.x.C, d.x.H, d.x.W, d.x.sh, d.x.sw, d.y.C, d.y.H, d.KH, d.stride, d.pad, d.transposed, d.ups, d.algo, PRECISION,
           PRECISION_GENERIC, cin_real, cout_real)
    hit = _conv_plans.get(key)
    if hit is None:
        d.KW = d.KH
        d.precision = _PREC[PRECISION]
        plan = L.ConvPlan()
        L.call('mt_conv_plan_query', C.byref(d), cin_real, cout_real, C.byref(plan))
        prec = eff_precision(plan.path == L.PATH_HALO)
        if prec != PRECISION:
            d.precision = _PREC[prec]
            L.call('mt_conv_plan_query', C.byref(d), cin_real, cout_real, C.byref(plan))
        name = plan.kernel.decode()
        KERNEL_INFO[name] = (plan.products, plan.wino_num / plan.wino_den)
        hit = _conv_plans[key] = (prec, plan, name)
    return hit


def _tiles_key(d):
    """The tile list of an mt_wgrad as a hashable value."""
    return (d.n_tiles, bytes(d.tiles)) if d.n_tiles else 0


def tile_cover(need, na, nb):
    """Cover the wanted 32 x 32 channel products ``need`` = {(D group, G group)} with 64 x 64 tiles made of two D groups
    x two G groups chosen freely (mt_wgrad_tile): greedy, most still-uncovered products first, compact tiles on ties.
    Returns [(a0, a1, b0, b1, quads)], quads bit (2 i + j) = the product of D group i with G group j of the tile is
    wanted (a product one tile has delivered is masked in the later ones).  The four growth convolutions of a dense
    block (32 outputs from 64 / 96 / 128 / 160 inputs: 14 products) come out as 4 tiles against five 64-aligned ones."""
    remaining, tiles = set(need), []
    while remaining:
        best = None
        for a0 in range(na):
            for a1 in range(a0, na):
                for b0 in range(nb):
                    for b1 in range(b0, nb):
                        cnt = len({(a0, b0), (a0, b1), (a1, b0), (a1, b1)} & remaining)
                        rank = (cnt, -(a1 - a0) - (b1 - b0))
                        if best is None or rank > best[0]:
                            best = (rank, (a0, a1, b0, b1))
        a0, a1, b0, b1 = best[1]
        quads = 0
        for i, ga in enumerate((a0, a1)):
            for j, gb in enumerate((b0, b1)):
                if (ga, gb) in remaining:
                    quads |= 1 << (2 * i + j)
                    remaining.discard((ga, gb))
        tiles.append((a0, a1, b0, b1, quads))
    return tiles


def _wgrad_plan(d, a_real, b_real):
    """(precision name, mt_wgrad_plan, label) of the weight gradient ``d`` describes (geometry, skip_tiles, d_terms set)."""
    key = (d.d.N, d.d.H, d.d.W, d.d.C, d.d.sh, d.d.sw, d.g.H, d.g.W, d.g.C, d.g.sh, d.g.sw, d.KH, d.stride, d.pad, d.ups,
           d.algo, d.skip_tiles, d.d_terms, PRECISION, PRECISION_GENERIC, a_real, b_real, _tiles_key(d))
    hit = _wgrad_plans.get(key)
    if hit is None:
        d.KW = d.KH
        d.precision = _PREC[PRECISION]
        plan = L.WgradPlan()
        L.call('mt_wgrad_plan_query', C.byref(d), a_real, b_real, C.byref(plan))
        prec = eff_precision(bool(plan.halo))
        if prec != PRECISION:
            d.precision = _PREC[prec]
            L.call('mt_wgrad_plan_query', C.byref(d), a_real, b_real, C.byref(plan))
        name = plan.kernel.decode()
        KERNEL_INFO[name] = (plan.products, 1.0)
        hit = _wgrad_plans[key] = (prec, plan, name)
    return hit


class _Prof:
    """HIP events around ONE launch on the current stream (bench.py roofline)."""

    def __init__(self, name, flops):
        self.on = PROFILE is not None
        if self.on:
            self.name, self.flops = name, flops
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            PROFILE.append((self.name, self.flops, self.e0, self.e1))


def bump_weight_epoch(changed=None):
    """The parameters changed (optimizer step, new model): per-epoch caches are
    dropped, packs that were used during the epoch that ends are refreshed in one
    batch (repack_all), the others are forgotten.  ``changed`` = (lo, hi) byte range
    of the memory that was rewritten (FlatAdam: its flat parameter buffer): packs of
    weights outside it -- the frozen VGG, the aligner inside a CHN step: two thirds of
    the packs there -- are left alone."""
    global WEIGHT_EPOCH
    ended = WEIGHT_EPOCH
    WEIGHT_EPOCH += 1
    _pack_cache.clear()
    capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    if not capturing:      # (the set must not change under a capture: its tables are fixed)
        for k in [k for k, e in _packs.items() if e.used < ended]:
            del _packs[k]
    live = {e.wa.data_ptr() for e in _packs.values() if e.wa is not None}
    for k in [k for k, e in _wamax.items() if e.out.data_ptr() not in live]:
        del _wamax[k]                    # scale scalars live as long as a pack reads them
    repack_all(changed)


def cpad(c):
    """Channel padding of engine tensors: a multiple of 8 (the ABI needs 4; 8
    makes every stride-1 3x3/5x5 layer eligible for the halo-tile kernels)."""
    return (c + 7) // 8 * 8


def view_of(t):
    """mt_view of a 4-D (N,H,W,C) tensor whose channel stride is 1 (a channel
    window of a wider NHWC buffer is fine)."""
    assert t.dim() == 4 and t.stride(3) == 1 and t.dtype == torch.float32, (t.shape, t.stride())
    n, h, w, c = t.shape
    v = L.View(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2), n, h, w, c, n, 0)
    v._src = t
    return v


def swap_view(t, groups):
    """SeparableConv4d's permute (model_dfpn.py:621-623) as addressing: ``t`` is
    (G*P1, h, w, C) = [g][p1][p2][c]; the result views it as G*P2 images
    (g, p2) whose h x w pixels run over p1 (needs P1 == P2 == h*w)."""
    n, h, w, c = t.shape
    assert t.is_contiguous()
    p1 = n // groups
    p2 = h * w
    assert p1 * groups == n and p1 == p2
    v = L.View(t.data_ptr(), c, w * p2 * c, p2 * c, groups * p2, h, w, c, p2, p1 * p2 * c)
    v._src = t
    return v


def null_view():
    return L.View(0, 0, 0, 0, 0, 0, 0, 0, 0, 0)


# ------------------------------------------------------- tensor magnitudes --
_amax_chunk = {}          # device -> [chunk tensor, next free slot]
AMAX_CHUNK = 512


def new_amax(device):
    """A zeroed device scalar (a slot of a zero-filled chunk: one fill launch per
    AMAX_CHUNK scalars instead of one each; the slice keeps its chunk alive)."""
    device = torch.device(device)
    ent = _amax_chunk.get(device)
    capturing = torch.cuda.is_current_stream_capturing() if device.type == 'cuda' else False
    if ent is None or ent[1] >= AMAX_CHUNK or ent[2] != capturing:
        # (a chunk allocated outside a graph capture must not be handed out inside one and vice versa)
        ent = [torch.zeros((AMAX_CHUNK,), device=device, dtype=torch.float32), 0, capturing]
        _amax_chunk[device] = ent
    i = ent[1]
    ent[1] += 1
    return ent[0][i:i + 1]


def absmax_into(t, amax):
    """amax[0] = max(amax[0], max|t|) for an engine tensor / channel window."""
    L.call('mt_absmax', C.byref(view_of(t)), L.ptr(amax), L.stream_ptr())
    return amax


def amax_of(t):
    """Device scalar max|t| of an engine tensor, computed once per tensor object
    (the f16x3 kernels derive their power-of-two scale from it)."""
    a = getattr(t, '_mt_amax', None)
    if a is None:
        a = absmax_into(t, new_amax(t.device))
        try:
            t._mt_amax = a
        except AttributeError:
            pass
    return a


# ------------------------------------------------------------ weight packs --
# The packed images are a pure function of the parameters and a model needs the
# same set every step.  A pack is made on first use (single launches); from then
# on it lives in a registry, and bump_weight_epoch() (= the optimizer just
# changed the parameters) refreshes EVERY registered pack with two launches
# driven by device-side tables: mt_absmax_segments + mt_weight_pack_jobs.
# Entries not used during the epoch that just ended are dropped.
REPACK_BATCHED = os.environ.get('MT355_REPACK_BATCHED', '1') != '0'


class _PackEntry:
    __slots__ = ('out', 'wa', 'jobs', 'keep', 'used')

    def __init__(self, out, wa, jobs, keep):
        self.out, self.wa, self.jobs, self.keep, self.used = out, wa, jobs, keep, WEIGHT_EPOCH


class _AmaxEntry:
    __slots__ = ('out', 'weights')

    def __init__(self, out, weights):
        self.out, self.weights = out, weights


_packs = {}            # key -> _PackEntry
_wamax = {}            # key -> _AmaxEntry
_wamax_chunks = {}     # device -> [list of zero-filled chunks, next free slot in the last one]
_tables = {}           # 'sig' -> signature of the device tables, 'segs'/'jobs' -> (device tensor, host array, count)


def _new_weight_amax(device):
    """Scale scalars of WEIGHTS live in their own chunks: a re-pack zeroes the
    chunks and recomputes every scalar (weights may shrink)."""
    device = torch.device(device)
    ent = _wamax_chunks.setdefault(device, [[], 1024])
    if ent[1] >= 1024:
        ent[0].append(torch.zeros((1024,), device=device, dtype=torch.float32))
        ent[1] = 0
    i = ent[1]
    ent[1] += 1
    return ent[0][-1][i:i + 1]


def _amax_for(weights):
    """Device scalar >= max|w| over ``weights`` (one tensor: a layer's scale; several:
    the shared scale of a stacked pack)."""
    key = tuple((w.data_ptr(), w._version) for w in weights)
    ent = _wamax.get(key)
    if ent is None:
        wa = _new_weight_amax(weights[0].device)
        for w in weights:
            L.call('mt_absmax_flat', L.ptr(w), w.numel(), L.ptr(wa), L.stream_ptr())
        ent = _wamax[key] = _AmaxEntry(wa, list(weights))
    return ent.out


def _weight_amax(w):
    return _amax_for([w])


def stack_amax(weights):
    """One device scalar >= max|w| over several weights (the scale of a stacked pack)."""
    return _amax_for(list(weights))


def _job(src_ptr, a, b, t, sa, sb, st, dst, ap, bp, prec, flip, wa, kind, bp_total=None, b_off=0):
    bp_total = bp if bp_total is None else bp_total
    fo = L.load().mt_weight_pack_halo_frag_order(ap, bp_total, t, _PREC[prec]) if kind == 1 else 0
    return L.PackJob(src_ptr, dst.data_ptr(), wa.data_ptr() if wa is not None else None, sa, sb, st, a, b, t, ap, bp,
                     _PREC[prec], 1 if flip else 0, kind, bp_total, b_off, fo)


def _run_job(j):
    """Single-launch execution of one pack job (first use of a pack)."""
    if j.kind == 2:
        L.call('mt_weight_pack_planes', j.src, j.A, j.B, j.T, j.sa, j.sb, j.st, j.dst, j.AP, j.BP, j.precision, j.flip,
               j.w_amax, L.stream_ptr())
    elif j.kind == 1:
        L.call('mt_weight_pack_halo_part', j.src, j.A, j.B, j.T, j.sa, j.sb, j.st, j.dst, j.AP, j.BP, j.precision, j.flip,
               j.w_amax, j.BP_total, j.b_off, L.stream_ptr())
    else:
        L.call('mt_weight_pack_split', j.src, j.A, j.B, j.T, j.sa, j.sb, j.st, j.dst, j.AP, j.BP, j.precision, j.flip,
               j.w_amax, L.stream_ptr())


def _pack(w, a_dim, kind, prec, flip=False, halo=False, planes=None):
    """Packed copy of a conv weight for precision ``prec``: rows
    [AP][T][BP] (fp32 / bf16 planes) or, for the halo kernel, per-tile LDS
    images.  ``a_dim`` selects which of the two leading dims of ``w`` becomes
    the GEMM's output-channel axis.  Returns (packed, w_amax or None)."""
    d0, d1, kh, kw = w.shape
    t = kh * kw
    assert w.is_contiguous()
    if a_dim == 0:
        a, b, sa, sb = d0, d1, d1 * t, t
    else:
        a, b, sa, sb = d1, d0, t, d1 * t
    if prec == 'f32':      # reference arithmetic: plain fp32 rows, re-made every epoch
        key = (w.data_ptr(), w._version, kind, WEIGHT_EPOCH)
        hit = _pack_cache.get(key)
        if hit is None:
            out = torch.empty((cpad(a), t, cpad(b)), device=w.device, dtype=torch.float32)
            L.call('mt_weight_pack', L.ptr(w), a, b, t, sa, sb, 1, L.ptr(out), cpad(a), cpad(b), L.stream_ptr())
            hit = _pack_cache[key] = (out, w)
        return hit[0], None
    key = (w.data_ptr(), w._version, kind, halo, prec, planes)
    ent = _packs.get(key)
    if ent is None:
        wa = _weight_amax(w) if prec == 'f16x3' else None
        if planes is not None:      # stride-2 layer: fragment stream of the planes kernel (planes = mt_conv.transposed)
            n = L.load().mt_weight_pack_planes_elems(cpad(a), cpad(b), t, planes)
            out = torch.zeros((n,), device=w.device, dtype=torch.bfloat16)
            job = _job(w.data_ptr(), a, b, t, sa, sb, 1, out, cpad(a), cpad(b), prec, bool(planes), wa, 2)
            _run_job(job)
            ent = _packs[key] = _PackEntry(out, wa, [job], [w])
            return ent.out, ent.wa
        if halo:
            n = L.load().mt_weight_pack_halo_elems(cpad(a), cpad(b), t, _PREC[prec])
            out = torch.zeros((n,), device=w.device, dtype=torch.bfloat16)   # (the image may be shorter than the bound)
        else:
            kp = (t * cpad(b) + 31) // 32 * 32
            out = torch.empty((3 if prec == 'bf16x6' else 2, cpad(a), kp), device=w.device, dtype=torch.bfloat16)
        job = _job(w.data_ptr(), a, b, t, sa, sb, 1, out, cpad(a), cpad(b), prec, flip, wa, 1 if halo else 0)
        _run_job(job)
        # the entry keeps ``w`` alive so that its address cannot be recycled for a different weight
        ent = _packs[key] = _PackEntry(out, wa, [job], [w])
    ent.used = WEIGHT_EPOCH
    return ent.out, ent.wa


def pack_stacked_bwd(sources, a0, a_cnt, prec, wa=None):
    """Halo-kernel weight image of the GATHER form of a backward-data through
    several Conv2d layers at once: the launch's K axis is the concatenation of
    the output channels of ``sources`` (Conv2d weights (O_l, I_l, k, k), in the
    order their output gradients sit in the input tensor), its output channels
    are the input channels [a0, a0 + a_cnt) of every source, taps flipped:
        dX[:, a0:a0+a_cnt] = sum_l conv_T(dZ_l, W_l[:, a0:a0+a_cnt]).
    ``wa``: the shared F16X3 scale scalar (stack_amax)."""
    key = (tuple((w.data_ptr(), w._version) for w in sources), a0, a_cnt, prec, 'stack')
    ent = _packs.get(key)
    if ent is None:
        assert prec != 'f32'
        k = sources[0].shape[2]
        t = k * k
        ap, bp_total = cpad(a_cnt), sum(cpad(w.shape[0]) for w in sources)
        if prec == 'f16x3' and wa is None:
            wa = stack_amax(sources)
        n = L.load().mt_weight_pack_halo_elems(ap, bp_total, t, _PREC[prec])
        out = torch.zeros((n,), device=sources[0].device, dtype=torch.bfloat16)
        jobs, off = [], 0
        for w in sources:
            o, i = w.shape[0], w.shape[1]
            assert w.is_contiguous() and a0 + a_cnt <= i and w.shape[2] == k
            jobs.append(_job(w.data_ptr() + 4 * a0 * t, a_cnt, o, t, t, i * t, 1, out, ap, cpad(o), prec, True, wa, 1,
                             bp_total, off))
            off += cpad(o)
        for j in jobs:
            _run_job(j)
        ent = _packs[key] = _PackEntry(out, wa, jobs, list(sources))
    ent.used = WEIGHT_EPOCH
    return ent.out, ent.wa


def drop_epoch_cache():
    """Forget the per-epoch copies (fp32 packs, padded biases) so that the next use re-makes
    them: a graph capture must contain those launches."""
    _pack_cache.clear()


def pack_state():
    """Everything a captured graph's pack launches point at (entries, tables, scale
    chunks): a graph keeps this alive for as long as it may be replayed."""
    tables = {k: (dict(v) if isinstance(v, dict) else v) for k, v in _tables.items()}      # (snapshots: the live dicts get cleared)
    return (list(_packs.values()), list(_wamax.values()), tables,
            [c for chunks in _wamax_chunks.values() for c in chunks[0]])


def _touched(weights, changed):
    return changed is None or any(changed[0] <= w.data_ptr() < changed[1] for w in weights)


def repack_all(changed=None):
    """Refresh the registered packs from the current parameters: zero their weight scale scalars,
    one mt_absmax_segments launch, one mt_weight_pack_jobs launch.  ``changed`` (see
    bump_weight_epoch): only packs with a source weight inside that byte range."""
    ents = [e for e in _packs.values() if _touched(e.keep, changed)]
    if not ents:
        return
    amaxes = [e for e in _wamax.values() if _touched(e.weights, changed)]
    if not REPACK_BATCHED:
        for e in amaxes:
            e.out.zero_()
            for w in e.weights:
                L.call('mt_absmax_flat', L.ptr(w), w.numel(), L.ptr(e.out), L.stream_ptr())
        for e in ents:
            for j in e.jobs:
                _run_job(j)
        return
    # (buffer addresses are part of the signature: an entry that was dropped and re-made has the same key)
    sig = (tuple((k, e.out.data_ptr()) for k, e in _packs.items()), tuple((k, e.out.data_ptr()) for k, e in _wamax.items()), changed)
    tab = _tables.setdefault('by_range', {})
    if _tables.get('sig') != sig[:2]:
        tab.clear()                       # the set of packs changed: every cached table is stale
        _tables['sig'] = sig[:2]
    cur = tab.get(changed)
    if cur is None:
        assert not torch.cuda.is_current_stream_capturing(), \
            'the set of weight packs changed inside a graph capture (run the step eagerly first)'
        dev = ents[0].out.device
        jobs = [j for e in ents for j in e.jobs]
        segs = [L.AmaxSeg(w.data_ptr(), w.numel(), e.out.data_ptr()) for e in amaxes for w in e.weights]

        def table(items, ctype):
            host = (ctype * len(items))(*items)
            t = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(dev)
            return t, host, len(items)
        # the scale scalars to reset: positions inside the zero-filled chunks they were handed out from
        zero = []
        for chunks in _wamax_chunks.values():
            for c in chunks[0]:
                idx = [(e.out.data_ptr() - c.data_ptr()) // 4 for e in amaxes
                       if c.data_ptr() <= e.out.data_ptr() < c.data_ptr() + 4 * c.numel()]
                if len(idx) == c.numel() or (changed is None and idx):
                    zero.append((c, None))
                elif idx:
                    zero.append((c, torch.tensor(idx, device=c.device, dtype=torch.long)))
        cur = tab[changed] = dict(jobs=table(jobs, L.PackJob), segs=table(segs, L.AmaxSeg) if segs else None, zero=zero)
        _tables['jobs'], _tables['segs'] = cur['jobs'], cur['segs']      # (kept alive with the graph: pack_state)
    if cur['segs'] is not None:
        for c, idx in cur['zero']:
            if idx is None:
                c.zero_()
            else:
                c.index_fill_(0, idx, 0.0)
        t, host, n = cur['segs']
        L.call('mt_absmax_segments', L.ptr(t), n, L.stream_ptr())
    t, host, n = cur['jobs']
    L.call('mt_weight_pack_jobs', L.ptr(t), C.cast(host, C.c_void_p), n, L.stream_ptr())


def _pad_bias(b, n):
    if b is None:
        return None
    if n == cpad(n) and b.is_contiguous():     # (kernels read the bias one float at a time)
        return b                      # already a whole number of 8-channel groups: no padded copy
    key = (b.data_ptr(), b._version, 'bias', WEIGHT_EPOCH)
    hit = _pack_cache.get(key)
    if hit is not None:
        return hit[0]
    out = torch.empty((cpad(n),), device=b.device, dtype=torch.float32)
    L.call('mt_pad_vector', L.ptr(b), n, L.ptr(out), cpad(n), L.stream_ptr())
    _pack_cache[key] = (out, b)
    return out


class LayerSpec:
    """One conv-shaped layer.  kind 'c' = Conv2d (weight O,I,k,k), 't' =
    ConvTranspose2d (weight I,O,k,k, output_padding 1).  ``ups``: the input is
    nearest-upsampled x2 first (RRDBNet).  ``swap_groups``: the input is the
    previous output seen through ``swap_view``."""

    def __init__(self, kind, cin, cout, k, stride=1, act=L.ACT_NONE, ups=0, swap_groups=0):
        self.kind, self.cin, self.cout, self.k = kind, cin, cout, k
        self.stride, self.pad, self.act, self.ups = stride, k // 2, act, ups
        self.swap_groups = swap_groups
        # (c0, n): of the gradient wrt this layer's INPUT only channels c0 .. c0+n-1 are ever read (the first layer
        # of FlowEstimator: ops.PackFlowEstimatorFn differentiates the two flow channels only); None = all of it
        self.dx_window = None

    def out_hw(self, h, w):
        h, w = h << self.ups, w << self.ups
        if self.kind == 'c':
            return ((h + 2 * self.pad - self.k) // self.stride + 1,
                    (w + 2 * self.pad - self.k) // self.stride + 1)
        return ((h - 1) * self.stride - 2 * self.pad + self.k + 1,
                (w - 1) * self.stride - 2 * self.pad + self.k + 1)


def _amax_ptr(prec, explicit, view):
    if prec != 'f16x3':
        return None, None
    a = explicit if explicit is not None else amax_of(view._src)
    return a.data_ptr(), a


def _out_amax(prec, explicit, tensor):
    """Device scalar the launch raises to max|output| (f16x3 only): the caller's
    running scalar for windows of a shared buffer, else a fresh one remembered on
    the output tensor so that amax_of() of the next layer finds it."""
    if prec != 'f16x3':
        return None, None
    if explicit is not None:
        return explicit.data_ptr(), explicit
    a = new_amax(tensor.device)
    try:
        tensor._mt_amax = a
    except AttributeError:
        pass
    return a.data_ptr(), a


def _fill_forward_desc(d, spec, xv, w, b, out, res, alpha, beta, x_amax, y_amax, plan, prec):
    """The mt_conv fields of one forward layer on the MFMA kernels (d.x / d.y are set by the caller).  Returns what
    must stay alive until the launch has been enqueued."""
    transposed = 0 if spec.kind == 'c' else 1
    halo, planes = plan.wt_layout == L.WT_HALO, plan.wt_layout == L.WT_PLANES
    packed, wa = _pack(w, 0 if spec.kind == 'c' else 1, 'f', prec, halo=halo, planes=transposed if planes else None)
    d.wt = packed.data_ptr()
    d.wt_layout = plan.wt_layout
    pb = _pad_bias(b, spec.cout)
    d.bias = pb.data_ptr() if pb is not None else None
    d.KH = d.KW = spec.k
    d.stride, d.pad, d.ups = spec.stride, spec.pad, spec.ups
    d.transposed = transposed
    d.act, d.alpha = spec.act, alpha
    d.res = view_of(res) if res is not None else null_view()
    d.beta = beta
    d.dsrc = null_view()
    d.precision = _PREC[prec]
    d.algo = 0 if USE_HALO else 1
    d.x_amax, keep = _amax_ptr(prec, x_amax, xv)
    d.w_amax = wa.data_ptr() if wa is not None else None
    d.y_amax, keep_y = _out_amax(prec, y_amax, out)
    return (packed, wa, pb, keep, keep_y)


def conv_forward(spec, xv, x_shape, w, b, out=None, res=None, alpha=1.0, beta=0.0, x_amax=None, y_amax=None):
    """Run one layer forward.  xv: mt_view of the input; x_shape (N,H,W).
    ``x_amax``: device scalar >= max|x| (defaults to amax_of the view's tensor;
    pass it for channel windows of a shared buffer).  ``y_amax``: scalar to
    raise to max|out| (windows of a shared buffer: the buffer's running max)."""
    n, h, w_ = x_shape
    ho, wo = spec.out_hw(h, w_)
    if out is None:
        out = torch.empty((n, ho, wo, cpad(spec.cout)), device=w.device, dtype=torch.float32)
    d = L.Conv()
    d.x, d.y = xv, view_of(out)
    d.KH = d.KW = spec.k
    d.stride, d.pad, d.ups = spec.stride, spec.pad, spec.ups
    d.transposed = 0 if spec.kind == 'c' else 1
    d.algo = 0 if USE_HALO else 1
    prec, plan, name = _conv_plan(d, spec.cin, spec.cout)
    if res is None and w.is_contiguous() and plan.tiny_in_ok:
        pb = _pad_bias(b, spec.cout)
        d.bias = pb.data_ptr() if pb is not None else None
        d.act, d.alpha, d.beta = spec.act, alpha, 0.0
        d.res, d.dsrc = null_view(), null_view()
        d.y_amax, keep_y = _out_amax(prec, y_amax, out)
        with _Prof('conv_tiny_in_kernel', 2.0 * n * ho * wo * spec.cin * spec.cout * 9):
            L.call('mt_conv_tiny_in_run', C.byref(d), L.ptr(w), spec.cin, spec.cin * 9, 9, 1, 0, L.stream_ptr())
        return out
    if res is None and w.is_contiguous() and plan.tiny_out_ok:
        pb = _pad_bias(b, spec.cout)
        d.bias = pb.data_ptr() if pb is not None else None
        d.act, d.alpha, d.beta = spec.act, alpha, 0.0
        d.res, d.dsrc = null_view(), null_view()
        d.y_amax, keep_y = _out_amax(prec, y_amax, out)
        kk = spec.k * spec.k
        with _Prof('conv_tiny_out_kernel', 2.0 * n * ho * wo * spec.cin * spec.cout * kk):
            L.call('mt_conv_tiny_out_run', C.byref(d), L.ptr(w), spec.cout, spec.cin * kk, kk, 1, 0, 0, 1, L.stream_ptr())
        return out
    keep = _fill_forward_desc(d, spec, xv, w, b, out, res, alpha, beta, x_amax, y_amax, plan, prec)
    if spec.kind == 'c':
        flops = 2.0 * n * ho * wo * spec.cin * spec.cout * spec.k * spec.k
    else:
        flops = 2.0 * n * h * w_ * spec.cin * spec.cout * spec.k * spec.k
    with _Prof(name, flops):
        L.call('mt_conv_run', C.byref(d), L.stream_ptr())
    return out


def conv_backward_data(spec, dz, w, yv, dsrc=None, dact=L.ACT_NONE, dact_c0=0, accumulate=False, alpha=1.0,
                       dz_amax=None, y_amax=None, packed=None, res=None, beta=0.0, dx_window=None):
    """dIn = bwd-data(dZ) written through the view ``yv`` (the layer's input
    geometry), optionally gated by act'(dsrc) (the previous layer's saved
    output)."""
    d = L.Conv()
    dzv = view_of(dz)
    d.x, d.y = dzv, yv
    d.algo = 0 if USE_HALO else 1
    # the backward-data of a stride-1 "same" Conv2d is itself a stride-1 "same" convolution with tap-flipped,
    # channel-swapped weights: described as that FORWARD convolution it takes the halo-tile kernels (if the library says so)
    fwd_like = spec.kind == 'c' and spec.stride == 1 and spec.pad == spec.k // 2
    d.KH = d.KW = spec.k
    d.stride, d.pad, d.ups = spec.stride, spec.pad, 0
    d.transposed = 0 if (spec.kind != 'c' or fwd_like) else 1
    # (gathered real channels = the conv's outputs; produced real channels = its inputs, or the wanted window of them)
    prec, plan, name = _conv_plan(d, spec.cout, dx_window[1] if dx_window is not None else spec.cin)
    as_fwd = fwd_like and plan.path == L.PATH_HALO
    if fwd_like and not as_fwd:              # (no halo kernel for it: the transposed gather of the generic kernel)
        d.transposed = 1
        prec, plan, name = _conv_plan(d, spec.cout, spec.cin)
    if (dx_window is not None and packed is None and res is None and not accumulate and dsrc is None and w is not None
            and w.is_contiguous() and fwd_like and plan.tiny_out_ok):
        # only channels c0 .. c0+n-1 of the input gradient are wanted: a 128 -> n forward conv of dZ with the
        # tap-reversed kernel, written into that channel window (the rest of yv stays unwritten)
        c0, nch = dx_window
        kk = spec.k * spec.k
        d.bias = None
        d.act, d.alpha, d.beta = L.ACT_NONE, alpha, 0.0
        d.res, d.dsrc = null_view(), null_view()
        d.y_amax = None
        with _Prof('conv_tiny_out_kernel', 2.0 * dzv.N * dzv.H * dzv.W * nch * spec.cout * kk):
            L.call('mt_conv_tiny_out_run', C.byref(d), C.c_void_p(w.data_ptr() + 4 * c0 * kk), nch, kk, spec.cin * kk, 1, 1,
                   c0, 0, L.stream_ptr())
        return
    if dx_window is not None and fwd_like:   # (the window was only an option of the strip kernel: plan the whole gradient)
        prec, plan, name = _conv_plan(d, spec.cout, spec.cin)
        as_fwd = plan.path == L.PATH_HALO
    if (packed is None and res is None and not accumulate and dact_c0 == 0 and w is not None and w.is_contiguous()
            and fwd_like and plan.tiny_in_ok):
        # backward-data of a Conv2d with <= 4 output channels = a forward conv of dZ with the tap-reversed kernel
        d.bias = None
        d.act, d.alpha, d.beta = L.ACT_NONE, alpha, 0.0
        d.res = null_view()
        d.dsrc = view_of(dsrc) if dsrc is not None else null_view()
        d.dact, d.dact_c0, d.accumulate = dact, 0, 0
        d.y_amax, keep_y = _out_amax(prec, y_amax, yv._src)
        with _Prof('conv_tiny_in_kernel', 2.0 * dzv.N * dzv.H * dzv.W * spec.cin * spec.cout * 9):
            L.call('mt_conv_tiny_in_run', C.byref(d), L.ptr(w), spec.cout, 9, spec.cin * 9, 1, 1, L.stream_ptr())
        return
    if packed is not None:      # a stacked image from pack_stacked_bwd (halo-tile kernels only)
        assert as_fwd
        packed, wa = packed
    else:
        planes = plan.wt_layout == L.WT_PLANES
        packed, wa = _pack(w, 1 if spec.kind == 'c' else 0, 'bflip' if as_fwd else 'b', prec, flip=as_fwd, halo=as_fwd,
                           planes=d.transposed if planes else None)
    d.wt = packed.data_ptr()
    d.wt_layout = plan.wt_layout
    d.bias = None
    d.act, d.alpha = L.ACT_NONE, alpha
    d.res, d.beta = (view_of(res), beta) if res is not None else (null_view(), 0.0)
    d.dsrc = view_of(dsrc) if dsrc is not None else null_view()
    d.dact, d.dact_c0 = dact, dact_c0
    d.accumulate = 1 if accumulate else 0
    d.precision = _PREC[prec]
    d.x_amax, keep = _amax_ptr(prec, dz_amax, dzv)
    d.w_amax = wa.data_ptr() if wa is not None else None
    d.y_amax, keep_y = _out_amax(prec, y_amax, yv._src)
    side = d.x if spec.kind == 'c' else yv          # the conv's output side
    flops = 2.0 * side.N * side.H * side.W * spec.cin * spec.cout * spec.k * spec.k
    with _Prof(name, flops):
        L.call('mt_conv_run', C.byref(d), L.stream_ptr())


def _splits(blocks_base, pixels):
    s = max(1, min((1536 + blocks_base - 1) // blocks_base, pixels // 512))
    return max(1, min(s, 512))


INPLACE_GRADS = os.environ.get('MT355_INPLACE_GRADS', '1') != '0'


def inplace_grad(p):
    """The flat-buffer gradient window of parameter ``p`` (FlatAdam tags them) or None.
    With a window the weight-gradient reduce ADDS into it and the autograd Function returns
    None for that parameter: no temporary, no AccumulateGrad add launch per parameter.
    (torch.autograd.grad() on such a model sees no gradient for conv parameters:
    MT355_INPLACE_GRADS=0 restores returned gradients.)"""
    g = getattr(p, 'grad', None) if p is not None else None
    if INPLACE_GRADS and g is not None and getattr(g, '_mt_flat', False) and g.is_contiguous():
        return g
    return None


def _batch_splits(blocks, n, ntiles, cus=256):
    """Pixel ranges per (tile, problem) of a batched weight-gradient grid.  The kernel runs ONE block per CU (132 KB of
    LDS), so the grid executes in rounds of ``cus`` blocks and a round costs one block's pixel tiles (+ ~24 tiles' worth of
    pipeline fill and the 147 KB slab it writes): pick the count that minimises rounds x tiles per block instead of the one
    that merely reaches a block target -- 180 (tile, problem) pairs x 3 ranges = 540 blocks ran as three rounds of which the
    last was a tenth full (measured: 14.1 -> see DESIGN.md section 3, "Batched weight gradients")."""
    forced = os.environ.get('MT355_WGRAD_BATCH_SPLITS')
    if forced == 'old':
        return max(1, min((WGRAD_PC_BLOCKS + blocks * n - 1) // (blocks * n), ntiles // 8 if ntiles >= 8 else 1, 256))
    if forced:
        return max(1, int(forced))
    best = (None, 1)
    for s_ in range(1, 65):
        if s_ > 1 and ntiles // s_ < 8:
            break
        rounds = -(-(blocks * n * s_) // cus)
        cost = rounds * (-(-ntiles // s_) + 24)
        if best[0] is None or cost < best[0]:
            best = (cost, s_)
    return best[1]


class WgradBatch:
    """Weight gradients that do not depend on one another (the dense blocks of the RRDB trunk),
    recorded during the backward chain and launched afterwards: one mt_wgrad_run_batch grid per
    geometry (blockIdx.z = layer) instead of one launch per layer.  A single dense block has only
    131 k pixels, so on its own it is cut into ~100 pixel ranges to fill the chip -- ~25 pixel
    tiles per block, a 147 KB slab written per block and read back by the reduce; batched, every
    block walks hundreds of tiles and the slabs shrink ~30x."""

    def __init__(self):
        self.items = []

    def add(self, d, keep, finish, blocks, ntiles, slab_floats, kname, flops):
        key = (d.d.N, d.d.H, d.d.W, d.d.C, d.d.sn, d.d.sh, d.d.sw, d.d.n_in, d.d.sg, d.g.C, d.g.sn, d.g.sh, d.g.sw,
               d.g.n_in, d.g.sg, d.precision, d.with_bias, d.skip_tiles, d.d_terms, _tiles_key(d))
        self.items.append((key, d, keep, finish, blocks, ntiles, slab_floats, kname, flops))

    def flush(self):
        groups = {}
        for it in self.items:
            groups.setdefault(it[0], []).append(it)
        self.items = []
        for group in groups.values():
            for c0 in range(0, len(group), L.WGRAD_BATCH_MAX):
                chunk = group[c0: c0 + L.WGRAD_BATCH_MAX]
                n = len(chunk)
                _, d0, _, _, blocks, ntiles, per, kname, _ = chunk[0]
                splits = _batch_splits(blocks - bin(int(d0.skip_tiles)).count('1'), n, ntiles)
                slabs = torch.empty((n * splits * per,), device=torch.device('cuda', torch.cuda.current_device()),
                                    dtype=torch.float32)
                arr = (L.Wgrad * n)()
                for i, it in enumerate(chunk):
                    it[1].splits = splits
                    it[1].slabs = slabs.data_ptr() + 4 * i * splits * per
                    C.memmove(C.byref(arr, i * C.sizeof(L.Wgrad)), C.byref(it[1]), C.sizeof(L.Wgrad))
                KERNEL_INFO[kname + f' x{n}'] = KERNEL_INFO.get(kname, (3, 1.0))
                with _Prof(kname + f' x{n}', sum(it[8] for it in chunk)):
                    L.call('mt_wgrad_run_batch', arr, n, L.stream_ptr())
                for i, it in enumerate(chunk):
                    it[3](slabs[i * splits * per: (i + 1) * splits * per], splits)


def conv_backward_weight(spec, dz, xin_view, w, need_bias=True, scale=1.0, dz_amax=None, x_amax=None, sub_blocks=None,
                         grad_w=None, grad_b=None, defer=None):
    """(dW, db) of one layer.  dz: (N,Ho,Wo,CoP) gradient wrt the layer's
    pre-activation output; xin_view: mt_view of the layer's input.

    ``sub_blocks`` = [(a0, cout_i, cin_i, w_i), ...]: ONE launch for several
    Conv2d layers that read channel prefixes of the same input and whose output
    gradients sit side by side in ``dz`` (the four growth convs of a dense
    block): the launch forms the whole (sum cout_i) x (max cin_i) product and
    every layer takes its sub-block of the slabs.  Returns [(dW_i, db_i), ...].

    ``grad_w`` / ``grad_b`` (sub_blocks: a 5th / 6th tuple entry): gradient buffers to ADD into
    (see inplace_grad); the corresponding return value is then None.

    ``defer``: a WgradBatch.  A layer the batched kernel takes (stride-1 3x3, two-term split) is
    only recorded; WgradBatch.flush() launches it together with the others of its geometry and
    fills the returned tensors then (dz, the input and the max|.| scalars must stay untouched
    until the flush)."""
    dzv = view_of(dz)
    if spec.kind == 'c':
        dv, gv, a, b = dzv, xin_view, spec.cout, spec.cin
        da, ga = dz_amax, x_amax
    else:
        dv, gv, a, b = xin_view, dzv, spec.cin, spec.cout
        da, ga = x_amax, dz_amax
    ap, bp, t = cpad(a), cpad(b), spec.k * spec.k
    pixels = dv.N * dv.H * dv.W
    d = L.Wgrad()
    d.d, d.g = dv, gv
    d.KH = d.KW = spec.k
    d.stride, d.pad = spec.stride, spec.pad
    d.ups = spec.ups if spec.kind == 'c' else 0
    d.algo = 0 if USE_HALO else 1
    d.d_terms = WGRAD_D_TERMS        # (a wish: the library applies it where the kernel has the one-term form)
    if sub_blocks is not None:
        # block-triangular product: 64 x 64 tiles (a-tile, b-tile) that no sub-block reads are skipped
        nbt = (bp + 63) // 64
        need = set()
        for sb in sub_blocks:
            for at_ in range(sb[0] // 64, (sb[0] + sb[1] - 1) // 64 + 1):
                for bt_ in range(0, (sb[2] - 1) // 64 + 1):
                    need.add(at_ * nbt + bt_)
        d.skip_tiles = sum(1 << i for i in range(min(32, ((ap + 63) // 64) * nbt)) if i not in need)
    # which kernel, how many blocks per pixel range, how many pixel tiles: the library's answer (csrc/dispatch.cpp)
    prec, plan, kname = _wgrad_plan(d, a, b)
    if sub_blocks is not None and WGRAD_TILE_LIST and plan.tile_list_ok and ap % 32 == 0 and bp % 32 == 0:
        # the useful 32 x 32 products, covered by tiles of freely chosen 32-channel groups (fewer than the 64-aligned ones?)
        ckey = (tuple((sb[0], sb[1], sb[2]) for sb in sub_blocks), ap, bp)
        cover = _tile_covers.get(ckey)
        if cover is None:
            need = {(ga, gb) for sb in sub_blocks for ga in range(sb[0] // 32, (sb[0] + sb[1] + 31) // 32)
                    for gb in range((sb[2] + 31) // 32)}
            cover = _tile_covers[ckey] = tile_cover(need, ap // 32, bp // 32)
        if len(cover) <= L.WGRAD_TILES_MAX and len(cover) < plan.blocks - bin(int(d.skip_tiles)).count('1'):
            d.skip_tiles, d.n_tiles, bias_done = 0, len(cover), set()
            for i, (a0_, a1_, b0_, b1_, quads) in enumerate(cover):
                t_ = d.tiles[i]
                t_.a[0], t_.a[1], t_.b[0], t_.b[1], t_.quads = a0_, a1_, b0_, b1_, quads
                t_.bias = 0 if {a0_, a1_} <= bias_done else 1        # every D group's bias sums are written by (at least) one tile
                bias_done |= {a0_, a1_}
            prec, plan, kname = _wgrad_plan(d, a, b)
    halo, pc = bool(plan.halo), bool(plan.pc)
    blocks, ntiles = plan.blocks, plan.ntiles
    if halo:
        want = WGRAD_PC_BLOCKS if pc else WGRAD_HALO_BLOCKS        # blocks of the launch: one 512-thread block per CU for pc
        d.splits = max(1, min((want + blocks - 1) // blocks, ntiles // 8 if ntiles >= 8 else 1, 256))
    else:
        d.skip_tiles = 0
        d.splits = _splits(blocks, pixels)
    # on the halo path the bias gradient rides along as one more slab "tap"
    fused_bias = bool(plan.bias_tap_ok) and need_bias and spec.kind == 'c'
    d.with_bias = 1 if fused_bias else 0
    ts = t + (1 if fused_bias else 0)
    d.precision = _PREC[prec]
    d.d_amax, keep1 = _amax_ptr(prec, da, dv)
    d.g_amax, keep2 = _amax_ptr(prec, ga, gv)
    useful = spec.cin * spec.cout if sub_blocks is None else sum(sb[1] * sb[2] for sb in sub_blocks)
    flops = 2.0 * pixels * useful * t                 # (algorithmic FLOP: only the sub-blocks that are kept)
    batchable = defer is not None and bool(plan.batchable)
    # <= 4 real channels on one side, 64 / 128 on the other, stride-1 3x3: the fp32 VALU kernel (conv_tiny.hip)
    tiny_w = bool(plan.tiny_ok) and spec.kind == 'c' and sub_blocks is None
    if tiny_w:
        rows = 8 if max(ap, bp) == 128 else 16
        nt_ = dv.N * ((dv.H + rows - 1) // rows) * ((dv.W + 15) // 16)
        d.splits = max(1, min(512, nt_))
        d.skip_tiles = 0
        slabs = torch.empty((d.splits * ap * ts * bp,), device=w.device, dtype=torch.float32)
        d.slabs = slabs.data_ptr()
        with _Prof('wgrad_tiny_kernel', flops):
            L.call('mt_wgrad_tiny_run', C.byref(d), min(spec.cin, spec.cout), L.stream_ptr())
        batchable = False
    elif not batchable:
        slabs = torch.empty((d.splits * ap * ts * bp,), device=w.device, dtype=torch.float32)
        d.slabs = slabs.data_ptr()
        with _Prof(kname, flops):
            L.call('mt_wgrad_run', C.byref(d), L.stream_ptr())
    if sub_blocks is not None:
        assert spec.kind == 'c' and (fused_bias or not need_bias)
        outs, subs, keep = [], (L.ReduceSub * len(sub_blocks))(), []
        all_inplace = None
        for n_, sb in enumerate(sub_blocks):
            a0, ca, cb, wi = sb[:4]
            gwi, gbi = (sb[4], sb[5]) if len(sb) > 4 else (None, None)
            inplace = gwi is not None and (gbi is not None or not need_bias)
            assert all_inplace in (None, inplace), 'sub-blocks must agree on in-place accumulation'
            all_inplace = inplace
            dwi = gwi if inplace else torch.empty_like(wi)
            dbi = (gbi if inplace else torch.empty((ca,), device=wi.device, dtype=torch.float32)) if need_bias else None
            keep += [dwi, dbi]
            subs[n_] = L.ReduceSub(a0, ca, cb, cb * t, t, 1, dwi.data_ptr(),
                                   dbi.data_ptr() if (fused_bias and dbi is not None) else None)
            outs.append((None, None) if inplace else (dwi, dbi))
        # every layer's sub-block (rows a0.. of every slab) in one reduce launch
        def finish(slabs_, splits_):
            L.call('mt_wgrad_reduce_multi', L.ptr(slabs_), splits_, ap, t, ts, bp, subs, len(sub_blocks),
                   1 if all_inplace else 0, scale, L.stream_ptr())
        if batchable:
            defer.add(d, (keep, keep1, keep2, dz, w), finish, blocks, ntiles, ap * ts * bp, kname, flops)
        else:
            finish(slabs, d.splits)
        return outs
    inplace = grad_w is not None and (grad_b is not None or not need_bias) and (fused_bias or not need_bias or scale == 1.0)
    dw = grad_w if inplace else torch.empty_like(w)
    db = (grad_b if inplace else torch.empty((spec.cout,), device=w.device, dtype=torch.float32)) if need_bias else None
    # both Conv2d (O,I,k,k) and ConvTranspose2d (I,O,k,k): a = dim 0, b = dim 1
    def finish(slabs_, splits_):
        L.call('mt_wgrad_reduce', L.ptr(slabs_), splits_, ap, t, ts, bp, L.ptr(dw), a, b, b * t, t, 1, 1 if inplace else 0, scale,
               L.ptr(db) if fused_bias else None, L.stream_ptr())
    if batchable and (fused_bias or not need_bias):
        defer.add(d, (dw, db, keep1, keep2, dz, w), finish, blocks, ntiles, ap * ts * bp, kname, flops)
        return (None, None) if inplace else (dw, db)
    if batchable:       # (the separate bias pass below is not deferred: launch this one now)
        slabs = torch.empty((d.splits * ap * ts * bp,), device=w.device, dtype=torch.float32)
        d.slabs = slabs.data_ptr()
        with _Prof(kname, flops):
            L.call('mt_wgrad_run', C.byref(d), L.stream_ptr())
    finish(slabs, d.splits)
    if need_bias and not fused_bias:
        ws = torch.empty((256 * dz.shape[3],), device=w.device, dtype=torch.float32)
        L.call('mt_bias_grad', C.byref(dzv), L.ptr(db), spec.cout, L.ptr(ws), 1 if inplace else 0, L.stream_ptr())
        if scale != 1.0:
            db.mul_(scale)
    return (None, None) if inplace else (dw, db)


def dact_(g, saved, act):
    """g *= act'(saved), in place."""
    if act != L.ACT_NONE:
        gv = view_of(g)
        L.call('mt_dact', C.byref(gv), C.byref(view_of(saved)), act, C.byref(gv), L.stream_ptr())
    return g


class ConvStackFn(torch.autograd.Function):
    """A chain of conv-shaped layers (nn.Sequential of Conv2d / ConvTranspose2d
    + ReLU in the reference: model_dfpn.py:596-605, 638-660, 686-711) as ONE
    autograd node on NHWC4 tensors."""

    @staticmethod
    def forward(ctx, x, specs, keep, *params):
        assert x.is_contiguous() and x.dim() == 4
        acts = [x]
        cur = x
        with precision(PRECISION if keep else nograd_precision()):
            cur = ConvStackFn._run(specs, params, cur, acts)
        ctx.specs = specs
        # Function.forward always runs with grad disabled: keep the saved
        # activations only when something upstream needs a gradient
        ctx.acts = acts if keep else None
        ctx.save_for_backward(*params)
        return cur

    @staticmethod
    def _run(specs, params, cur, acts):
        for i, sp in enumerate(specs):
            assert sp.ups == 0
            w, b = params[2 * i], params[2 * i + 1]
            if sp.swap_groups:
                xv = swap_view(cur, sp.swap_groups)
                shape = (xv.N, xv.H, xv.W)
            else:
                xv, shape = view_of(cur), cur.shape[:3]
            cur = conv_forward(sp, xv, shape, w, b)
            acts.append(cur)
        return cur

    @staticmethod
    def backward(ctx, dy):
        specs, acts, params = ctx.specs, ctx.acts, ctx.saved_tensors
        grads = [None] * len(params)
        dz = dy.contiguous()
        if specs[-1].act != L.ACT_NONE:
            dz = dact_(dz.clone(), acts[-1], specs[-1].act)
        need_x = ctx.needs_input_grad[0]
        for i in range(len(specs) - 1, -1, -1):
            sp, w = specs[i], params[2 * i]
            xin = acts[i]
            xin_view = swap_view(xin, sp.swap_groups) if sp.swap_groups else view_of(xin)
            dw, db = conv_backward_weight(sp, dz, xin_view, w, grad_w=inplace_grad(w),
                                          grad_b=inplace_grad(params[2 * i + 1]))
            grads[2 * i], grads[2 * i + 1] = dw, db
            if i == 0 and not need_x:
                dz = None
                break
            dx = torch.empty_like(xin)
            yv = swap_view(dx, sp.swap_groups) if sp.swap_groups else view_of(dx)
            prev_act = specs[i - 1].act if i > 0 else L.ACT_NONE
            if sp.swap_groups:
                assert prev_act == L.ACT_NONE
            conv_backward_data(sp, dz, w, yv, dsrc=xin if prev_act != L.ACT_NONE else None, dact=prev_act,
                               dx_window=sp.dx_window if i == 0 else None)
            dz = dx
            acts[i + 1] = None   # release the saved activation
        # (only when EVERY gradient of the stack was added into its flat-buffer window by the reduce launches above: a
        # gradient that is returned instead -- MT355_INPLACE_GRADS=0, a parameter outside a FlatAdam -- is accumulated
        # by autograd AFTER this hook, so its window is not final yet and must not go on the wire early)
        if GRADS_READY is not None and all(g is None for g in grads):
            GRADS_READY(params)
        return (dz, None, None) + tuple(grads)


def will_differentiate(*tensors):
    """True when autograd will record this call (Function.forward itself always
    runs with grad mode off, and ctx.needs_input_grad ignores torch.no_grad)."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def conv_stack(x, specs, params):
    return ConvStackFn.apply(x, specs, will_differentiate(x, *params), *params)


# ------------------------------------------------------------ layout ops ---
def to_nhwc4(x, scale=None, shift=None, cp=None):
    """(N,C,H,W) (any strides) -> (N,H,W,Cp) with optional per-channel affine."""
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, cp or cpad(c)), device=x.device, dtype=torch.float32)
    if out.shape[3] > (c + 3) // 4 * 4:     # the kernel writes whole groups of 4 channels
        out.zero_()
    ov = view_of(out)
    L.call('mt_nchw_to_nhwc4', L.ptr(x), x.stride(0), x.stride(1), x.stride(2), x.stride(3), n, c, h, w,
           L.ptr(scale), L.ptr(shift), C.byref(ov), 0, L.stream_ptr())
    return out


def from_nhwc4(x4, c, scale=None, shift=None):
    n, h, w, _ = x4.shape
    out = torch.empty((n, c, h, w), device=x4.device, dtype=torch.float32)
    L.call('mt_nhwc4_to_nchw', C.byref(view_of(x4)), 0, L.ptr(out), out.stride(0), out.stride(1), out.stride(2),
           out.stride(3), c, L.ptr(scale), L.ptr(shift), 0, L.stream_ptr())
    return out


def maxpool2x2(x4):
    n, h, w, c = x4.shape
    out = torch.empty((n, h // 2, w // 2, c), device=x4.device, dtype=torch.float32)
    L.call('mt_maxpool2x2', C.byref(view_of(x4)), C.byref(view_of(out)), L.stream_ptr())
    return out


class ToNHWC4Fn(torch.autograd.Function):
    """Differentiable (N,C,H,W) -> (N,H,W,Cp) layout change."""

    @staticmethod
    def forward(ctx, x):
        ctx.c = x.shape[1]
        return to_nhwc4(x)

    @staticmethod
    def backward(ctx, d4):
        return from_nhwc4(d4.contiguous(), ctx.c)


class FromNHWC4Fn(torch.autograd.Function):
    """Differentiable (N,H,W,Cp) -> (N,C,H,W) layout change."""

    @staticmethod
    def forward(ctx, x4, c):
        return from_nhwc4(x4.contiguous(), c)

    @staticmethod
    def backward(ctx, d):
        return to_nhwc4(d), None


def pack_images(out4, b, r, sources):
    """Fill the NHWC4 tensor ``out4`` (B*R images) from (B,C,R,H,W)/(B,C,H,W)
    tensors through mt_pack_pairs.  sources: (tensor, c_off, scale, shift)."""
    n, h, w, _ = out4.shape
    assert n == b * r
    arr = (L.PackSrc * len(sources))()
    keep = []
    for i, (t, c_off, scale, shift) in enumerate(sources):
        t = t.contiguous()
        keep.append(t)
        hw = h * w
        if t.dim() == 5:
            c = t.shape[1]
            assert t.shape == (b, c, r, h, w)
            arr[i] = L.PackSrc(t.data_ptr(), c * r * hw, hw, r * hw, w, 1, c, c_off,
                               scale.data_ptr() if scale is not None else None,
                               shift.data_ptr() if shift is not None else None)
        else:
            c = t.shape[1]
            assert t.shape == (b, c, h, w)
            arr[i] = L.PackSrc(t.data_ptr(), c * hw, 0, hw, w, 1, c, c_off,
                               scale.data_ptr() if scale is not None else None,
                               shift.data_ptr() if shift is not None else None)
    ov = view_of(out4)
    L.call('mt_pack_pairs', arr, len(sources), b, r, C.byref(ov), L.stream_ptr())
    return out4
