"""Copy-and-Hallucinate Network (CHN) on the mt355 HIP kernels.

Drop-in for the training path of ``master_thesis/model_chn.py``: ``CHN``,
``RRDBNet``, ``RRDB``, ``ResidualDenseBlock5C`` with the reference's constructor
arguments, state-dict keys (``nn.conv_first.N``, ``nn.rrdb_trunk.K.rdbJ.convI``,
``nn.{trunk_conv,upconv1,upconv2,hr_conv,conv_last}``) and loss names/order.
The whole RRDBNet is ONE autograd node: dense-block concatenations are channel
windows of a single 192-channel NHWC buffer (no torch.cat copies), residual
scalings are fused into conv epilogues, and the backward accumulates the dense
gradients in place.
"""
import ctypes as C
import functools

import os

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from . import ops
from .lightning_shim import LightningModule
from .optim import FlatAdam, reference_parameter_list
from .utils import LossesUtils, take

LRELU, NONE = L.ACT_LRELU, L.ACT_NONE


def _axpby(x, a, z, b, out, amax=None):
    """out = a*x + b*z; ``amax``: optional device scalar the launch raises to max|out| (two-level
    reduction through the stream's scratch, csrc/common.h: raise_amax_block -- with one atomic per
    wave on the scalar itself this cost +10 ms per CHN step and a separate mt_absmax pass was cheaper)."""
    L.call('mt_axpby', C.byref(E.view_of(x)), a, C.byref(E.view_of(z)) if z is not None else None, b,
           C.byref(E.view_of(out)), L.ptr(amax), L.stream_ptr())
    return out


class RRDBNetFn(torch.autograd.Function):
    """RRDBNet.forward (model_chn.py:538-555) and its backward on NHWC4."""

    @staticmethod
    def forward(ctx, x4, net, names, keep, *params):
        with E.precision(E.PRECISION if keep else E.nograd_precision()):
            out, saved = RRDBNetFn._run(x4, net, names, params, keep)
        ctx.saved, ctx.net, ctx.names = saved, net, names
        ctx.save_for_backward(*params)
        return out

    @staticmethod
    def _run(x4, net, names, params, keep):
        P = dict(zip(names, params))
        nf, gc, nb = net.nf, net.gc, net.nb
        cw = nf + 4 * gc
        saved = {}

        f16 = E.PRECISION == 'f16x3'     # the f16x3 kernels need max|x| of every conv input

        def conv(key, xin, cin, cout, stride=1, act=NONE, ups=0, out=None, res=None, alpha=1.0, beta=0.0,
                 x_amax=None, y_amax=None):
            sp = E.LayerSpec('c', cin, cout, 3, stride, act, ups)
            return E.conv_forward(sp, E.view_of(xin), xin.shape[:3], P[key + '.weight'], P[key + '.bias'], out=out,
                                  res=res, alpha=alpha, beta=beta, x_amax=x_amax, y_amax=y_amax)

        # conv_first: five plain convs (two stride-2), no activation (:522-528)
        cur, first = x4, [x4]
        for i, s in enumerate((1, 1, 2, 1, 2)):
            cur = conv(f'conv_first.{i}', cur, net.in_nc if i == 0 else nf, nf, stride=s)
            first.append(cur)
        f0 = cur
        n, h, w, _ = f0.shape

        def new_buf():
            return torch.empty((n, h, w, cw), device=f0.device, dtype=torch.float32)

        def track(window, amax):     # running max|.| of everything written into a dense-block buffer
            if f16:
                E.absmax_into(window, amax)

        bufs = []
        if RRDBNetFn._two_chains(n, f0):
            buf, buf_amax, bufs = RRDBNetFn._trunk_forward_chains(f0, P, nf, gc, nb, conv, f16, keep)
        else:
            buf, buf_amax = new_buf(), E.new_amax(f0.device)
            buf[..., :nf].copy_(f0)
            track(buf[..., :nf], buf_amax)
        for k in range(nb if bufs else 0, nb):      # (skipped when the two chains above ran the trunk)
            rrdb_x = buf[..., :nf]
            for j in (1, 2, 3):
                key = f'rrdb_trunk.{k}.rdb{j}.'
                for i in range(1, 5):
                    dst = buf[..., nf + (i - 1) * gc: nf + i * gc]
                    # (the launch itself raises the buffer's running max|.| by what it writes)
                    conv(f'{key}conv{i}', buf[..., :nf + (i - 1) * gc], nf + (i - 1) * gc, gc, act=LRELU, out=dst,
                         x_amax=buf_amax, y_amax=buf_amax if f16 else None)
                nxt, nxt_amax = new_buf(), E.new_amax(f0.device)
                # 0.2 * conv5(cat) + x  (:623) written as the next block's x
                conv(f'{key}conv5', buf, cw, nf, out=nxt[..., :nf], res=buf[..., :nf], alpha=0.2, beta=1.0,
                     x_amax=buf_amax, y_amax=nxt_amax if f16 else None)
                bufs.append((buf, buf_amax))
                buf, buf_amax = nxt, nxt_amax
            # 0.2 * y + x  (:585), in place in the next block's x window
            _axpby(buf[..., :nf], 0.2, rrdb_x, 1.0, buf[..., :nf], amax=buf_amax if f16 else None)
        z = buf[..., :nf]
        y = conv('trunk_conv', z, nf, nf, res=f0, alpha=1.0, beta=1.0, x_amax=buf_amax)   # (:548)
        u1 = conv('upconv1', y, nf, nf, act=LRELU, ups=1)                        # (:549-551)
        u2 = conv('upconv2', u1, nf, nf, act=LRELU, ups=1)                       # (:552-554)
        hr = conv('hr_conv', u2, nf, nf, act=LRELU)
        out = conv('conv_last', hr, nf, net.out_nc)                              # (:555)
        if keep:
            saved = dict(first=first, bufs=bufs, z=buf, z_amax=buf_amax, y=y, u1=u1, u2=u2, hr=hr)
        return out, saved

    # ---- the trunk as independent launch chains on several streams ------------------------------------------
    # A dense-block launch at the trunk's size (32 images of 64 x 64 at cfg 3) is ONE wave of 1024 blocks that lives
    # ~45 us: its ramp-up (every block first fetches its weight tile and halo) and its tail leave the matrix pipe idle
    # for a third of the launch, and the next launch depends on it.  The images of a batch do not depend on one
    # another, so the two halves of the batch run as two independent launch chains -- the second on a side stream,
    # forked / joined with events (inside a captured step: two parallel branches of the graph) -- and one chain's
    # ramp-up and tail overlap the other's main loop.  Every buffer the chains share is allocated before the fork
    # and lives until after the join; every weight pack is touched (made, on first use) before the fork; each chain
    # keeps its own running max|.| scalars, merged after the join for the launches that see the whole batch.
    CHAINS = int(os.environ.get('MT355_RRDB_CHAINS', '2'))
    CHAIN_MIN_TILES = 512
    _side = {}

    @staticmethod
    def _two_chains(n, t):
        """Chains pay where a launch over 1 / CHAINS of the batch still fills the chip (>= CHAIN_MIN_TILES 8 x 16 tiles:
        cfg 3 has 512 per half, cfg 5 1536); on small batches they only double the launch count -- RRDBNet forward, eager,
        16 images at 256^2: 15.7 ms in two chains, 7.3 ms for ONE image in one chain (scratch/r3_rdb_test.py)."""
        c = RRDBNetFn.CHAINS
        h, w = t.shape[1], t.shape[2]
        return (c >= 2 and n % c == 0 and n >= c and t.is_cuda and E.USE_HALO and E.PRECISION != 'f32'
                and E.PROFILE is None and (n // c) * ((h + 7) // 8) * ((w + 15) // 16) >= RRDBNetFn.CHAIN_MIN_TILES)

    @staticmethod
    def _chain_streams(device, c=None):
        """(streams, image slices factory): chain 0 runs on the current stream, the others on side streams."""
        c = RRDBNetFn.CHAINS if c is None else c
        st = RRDBNetFn._side.setdefault(device, [])
        while len(st) < c - 1:
            st.append(torch.cuda.Stream(device=device))
        return [torch.cuda.current_stream(device)] + st[:c - 1]

    @staticmethod
    def _fork(streams):
        ev = torch.cuda.Event()
        ev.record(streams[0])
        for s_ in streams[1:]:
            s_.wait_event(ev)

    @staticmethod
    def _join(streams):
        for s_ in streams[1:]:
            ev = torch.cuda.Event()
            ev.record(s_)
            streams[0].wait_event(ev)

    @staticmethod
    def _trunk_forward_chains(f0, P, nf, gc, nb, conv, f16, keep, nchains=None):
        """The nb x 3 dense blocks (model_chn.py:558-623) on ``nchains`` slices of the batch at once."""
        n, h, w, _ = f0.shape
        cw = nf + 4 * gc
        dev = f0.device
        streams = RRDBNetFn._chain_streams(dev, nchains)
        nc = len(streams)
        nbuf = 3 * nb + 1
        # every buffer up front (main stream); without a backward to feed, four rotate (an RRDB's input is read
        # again by the residual after its third block)
        pool = [torch.empty((n, h, w, cw), device=dev, dtype=torch.float32) for _ in range(nbuf if keep else 4)]
        halves = [slice(c * (n // nc), (c + 1) * (n // nc)) for c in range(nc)]
        amax = [torch.zeros((nbuf,), device=dev, dtype=torch.float32) for _ in range(nc)] if f16 else [None] * nc
        # touch every weight pack on the main stream (first use makes it: the chains must only ever read them)
        prec = E.eff_precision(True)
        for k in range(nb):
            for j in (1, 2, 3):
                for i in range(1, 6):
                    E._pack(P[f'rrdb_trunk.{k}.rdb{j}.conv{i}.weight'], 0, 'f', prec, halo=True)
        pool[0][..., :nf].copy_(f0)
        RRDBNetFn._fork(streams)

        def a_(c, i):
            return amax[c][i:i + 1] if f16 else None

        def rdb(c, k, j, idx):
            sl = halves[c]
            buf, nxt = pool[idx % len(pool)][sl], pool[(idx + 1) % len(pool)][sl]
            key = f'rrdb_trunk.{k}.rdb{j}.'
            for i in range(1, 5):
                conv(f'{key}conv{i}', buf[..., :nf + (i - 1) * gc], nf + (i - 1) * gc, gc, act=LRELU,
                     out=buf[..., nf + (i - 1) * gc: nf + i * gc], x_amax=a_(c, idx), y_amax=a_(c, idx))
            conv(f'{key}conv5', buf, cw, nf, out=nxt[..., :nf], res=buf[..., :nf], alpha=0.2, beta=1.0,
                 x_amax=a_(c, idx), y_amax=a_(c, idx + 1))

        for c in range(nc):
            with torch.cuda.stream(streams[c]):
                if f16:
                    E.absmax_into(pool[0][halves[c]][..., :nf], a_(c, 0))
        idx = 0
        for k in range(nb):
            for j in (1, 2, 3):
                for c in range(nc):        # (alternating, so that eager launches of all chains arrive together)
                    with torch.cuda.stream(streams[c]):
                        rdb(c, k, j, idx)
                idx += 1
            for c in range(nc):
                with torch.cuda.stream(streams[c]):
                    cur = pool[idx % len(pool)][halves[c]][..., :nf]
                    rrdb_x = pool[(idx - 3) % len(pool)][halves[c]][..., :nf]
                    _axpby(cur, 0.2, rrdb_x, 1.0, cur, amax=a_(c, idx) if f16 else None)     # 0.2 * y + x  (:585)
        RRDBNetFn._join(streams)
        merged = torch.stack(amax).amax(0) if f16 else None
        bufs = [(pool[i], merged[i:i + 1] if f16 else None) for i in range(nbuf - 1)] if keep else [None]
        last = pool[(nbuf - 1) % len(pool)]
        return last, (merged[nbuf - 1: nbuf] if f16 else None), bufs

    @staticmethod
    def _trunk_backward_chains(P, G, bufs, dcur, nf, gc, nb, f16, wbatch):
        """Backward of the nb x 3 dense blocks in gather form (see _rdb_backward_gather) as two chains over the two
        halves of the batch; the weight gradients see the whole batch and are recorded into ``wbatch`` (launched by
        the caller after the chains have joined).  Returns the gradient wrt the trunk input."""
        n, h, w, _ = dcur.shape
        cw = nf + 4 * gc
        dev = dcur.device
        streams = RRDBNetFn._chain_streams(dev)
        nc = len(streams)
        halves = [slice(c * (n // nc), (c + 1) * (n // nc)) for c in range(nc)]
        nblk = 3 * nb
        prec = E.eff_precision(True)
        # -- before the fork, on the main stream: buffers, scale scalars, weight packs
        dzbs = [torch.empty((n, h, w, cw), device=dev, dtype=torch.float32) for _ in range(nblk)]
        douts = [torch.empty((n, h, w, nf), device=dev, dtype=torch.float32) for _ in range(nb)]   # grad wrt each RRDB's input
        dins = [torch.empty((n, h, w, nf), device=dev, dtype=torch.float32) for _ in range(nb)]    # ... wrt its rdb1's input
        dba = [torch.zeros((nblk,), device=dev, dtype=torch.float32) for _ in range(nc)] if f16 else [None] * nc
        dba_all = torch.empty((nblk,), device=dev, dtype=torch.float32) if f16 else None           # max over all chains
        packs = {}
        for k in range(nb):
            for j in (1, 2, 3):
                key = f'rrdb_trunk.{k}.rdb{j}.'
                W = [None] + [P[f'{key}conv{i}.weight'] for i in (1, 2, 3, 4, 5)]
                wa = E.stack_amax(W[1:]) if prec == 'f16x3' else None
                packs[key] = ([None] + [E.pack_stacked_bwd([W[l] for l in range(5, i, -1)], nf + (i - 1) * gc, gc, prec, wa)
                                        for i in (1, 2, 3, 4)],
                              E.pack_stacked_bwd([W[l] for l in (5, 4, 3, 2, 1)], 0, nf, prec, wa))

        def spec(cin, cout):
            return E.LayerSpec('c', cin, cout, 3, 1, NONE, 0)

        def a_(c, i):
            return dba[c][i:i + 1] if f16 else None

        def chain_rdb(c, key, buf, bidx, last):
            """The five gather launches of one dense block on half ``c``.  ``last``: None = write 0.2 * d_in into the
            dz5 window of the block that runs next (bidx + 1); else the tensor that takes d_in = dense + 5 dz5."""
            sl = halves[c]
            dzb, bufc = dzbs[bidx][sl], buf[sl]
            pk, pk_in = packs[key]
            for i in (4, 3, 2, 1):
                kk = nf + (4 - i) * gc
                c0 = nf + (i - 1) * gc
                E.conv_backward_data(spec(gc, kk), dzb[..., :kk], None, E.view_of(dzb[..., kk: kk + gc]),
                                     dsrc=bufc[..., c0: c0 + gc], dact=LRELU, dz_amax=a_(c, bidx), y_amax=a_(c, bidx),
                                     packed=pk[i])
            if last is None:
                E.conv_backward_data(spec(nf, cw), dzb, None, E.view_of(dzbs[bidx + 1][sl][..., :nf]), dz_amax=a_(c, bidx),
                                     packed=pk_in, alpha=0.2, res=dzb[..., :nf], beta=1.0, y_amax=a_(c, bidx + 1))
            else:
                E.conv_backward_data(spec(nf, cw), dzb, None, E.view_of(last[sl]), dz_amax=a_(c, bidx), packed=pk_in,
                                     res=dzb[..., :nf], beta=5.0, y_amax=E.new_amax(dev) if f16 else None)

        RRDBNetFn._fork(streams)
        bidx = 0                                     # blocks in backward order: (nb-1, rdb3), (nb-1, rdb2), ...
        order = []
        for k in range(nb - 1, -1, -1):
            drr = dcur if k == nb - 1 else douts[k + 1]
            for c in range(nc):
                with torch.cuda.stream(streams[c]):
                    sl = halves[c]
                    # rdb3: dz5 = 0.2 * (0.2 * drr); rdb2 / rdb1 get theirs from the launch that ends the block before
                    _axpby(drr[sl], 0.04, drr[sl], 0.0, dzbs[bidx][sl][..., :nf], amax=a_(c, bidx) if f16 else None)
            for j in (3, 2, 1):
                key = f'rrdb_trunk.{k}.rdb{j}.'
                buf, buf_amax = bufs[3 * k + (j - 1)]
                for c in range(nc):
                    with torch.cuda.stream(streams[c]):
                        chain_rdb(c, key, buf, bidx, None if j > 1 else dins[k])
                order.append((key, buf, buf_amax, bidx))
                bidx += 1
            for c in range(nc):
                with torch.cuda.stream(streams[c]):
                    sl = halves[c]
                    _axpby(dins[k][sl], 1.0, drr[sl], 1.0, douts[k][sl])       # RRDB: out = 0.2 * y3 + x
        RRDBNetFn._join(streams)
        # -- after the join: the weight gradients of every block over the whole batch (recorded; the caller flushes)
        if f16:
            torch.amax(torch.stack(dba), 0, out=dba_all)
        for key, buf, buf_amax, b in order:
            dzb = dzbs[b]
            da = dba_all[b:b + 1] if f16 else None
            W = [None] + [P[f'{key}conv{i}.weight'] for i in (1, 2, 3, 4, 5)]
            Bi = [None] + [P[f'{key}conv{i}.bias'] for i in (1, 2, 3, 4, 5)]
            dw, db = E.conv_backward_weight(spec(cw, nf), dzb[..., :nf], E.view_of(buf), W[5], x_amax=buf_amax, dz_amax=da,
                                            grad_w=E.inplace_grad(W[5]), grad_b=E.inplace_grad(Bi[5]), defer=wbatch)
            G[f'{key}conv5.weight'], G[f'{key}conv5.bias'] = dw, db
            subs = [((4 - i) * gc, gc, nf + (i - 1) * gc, W[i], E.inplace_grad(W[i]), E.inplace_grad(Bi[i]))
                    for i in (1, 2, 3, 4)]
            res = E.conv_backward_weight(spec(nf + 3 * gc, 4 * gc), dzb[..., nf: cw], E.view_of(buf[..., :nf + 3 * gc]), W[1],
                                         x_amax=buf_amax, dz_amax=da, sub_blocks=subs, defer=wbatch)
            for i, (dwi, dbi) in zip((1, 2, 3, 4), res):
                G[f'{key}conv{i}.weight'], G[f'{key}conv{i}.bias'] = dwi, dbi
        return douts[0]

    @staticmethod
    def _rdb_backward_gather(key, P, G, buf, buf_amax, dzb, dba, nf, gc, f16, nxt=None, defer=None):
        """Backward of one ResidualDenseBlock5C (model_chn.py:602-623) in GATHER form.

        The layer-by-layer form scatters: the backward-data of conv_i adds into ALL the channels
        conv_i read (read-modify-write of up to 160 channels per launch, K = 32 only).  Here the
        gradient of each channel window is produced by ONE launch that gathers from every later
        layer: the output gradients are kept side by side as dzb = [dz5 | dz4 | dz3 | dz2 | dz1],
        so window i reads the channel PREFIX [dz5 .. dz(i+1)] against the stacked, tap-flipped
        weights [W5[:, win_i]; W4[:, win_i]; ...] and writes dz_i (gated by LeakyReLU'(buf[win_i]))
        right behind it; the last launch gives the block-input gradient with the residual path
        fused.  Same MFMA work, every gradient channel is written once, no accumulate passes.

        The caller has put dz5 = 0.2 * e into dzb[..., :nf] (e = the gradient that reaches this
        block's output, i.e. the residual path's share of d_in) with ``dba`` >= max|dz5|.  With
        ``nxt = (window, amax)`` -- the dz5 window of the block that runs next in this backward --
        the last launch writes 0.2 * d_in = 0.2 * dense + dz5 straight into it (and raises ``amax``):
        no separate scale and max passes between the dense blocks of one RRDB.  Without ``nxt`` it
        returns d_in = dense + 5 * dz5 (5 * fl(0.2 e) is e to one rounding)."""
        cw = nf + 4 * gc
        W = [None] + [P[f'{key}conv{i}.weight'] for i in (1, 2, 3, 4, 5)]
        prec = E.eff_precision(True)
        wa = E.stack_amax(W[1:]) if prec == 'f16x3' else None

        def spec(cin, cout):
            return E.LayerSpec('c', cin, cout, 3, 1, NONE, 0)

        Bi = [None] + [P[f'{key}conv{i}.bias'] for i in (1, 2, 3, 4, 5)]
        dw, db = E.conv_backward_weight(spec(cw, nf), dzb[..., :nf], E.view_of(buf), W[5], x_amax=buf_amax, dz_amax=dba,
                                        grad_w=E.inplace_grad(W[5]), grad_b=E.inplace_grad(Bi[5]), defer=defer)
        G[f'{key}conv5.weight'], G[f'{key}conv5.bias'] = dw, db
        for i in (4, 3, 2, 1):
            kk = nf + (4 - i) * gc                  # channels of [dz5 .. dz(i+1)]
            c0 = nf + (i - 1) * gc                  # window of conv_i's output inside buf
            srcs = [W[l] for l in range(5, i, -1)]
            E.conv_backward_data(spec(gc, kk), dzb[..., :kk], None, E.view_of(dzb[..., kk: kk + gc]),
                                 dsrc=buf[..., c0: c0 + gc], dact=LRELU, dz_amax=dba, y_amax=dba,
                                 packed=E.pack_stacked_bwd(srcs, c0, gc, prec, wa))
        packed_in = E.pack_stacked_bwd([W[l] for l in (5, 4, 3, 2, 1)], 0, nf, prec, wa)
        if nxt is not None:
            d_in = None
            E.conv_backward_data(spec(nf, cw), dzb, None, E.view_of(nxt[0]), dz_amax=dba, packed=packed_in,
                                 alpha=0.2, res=dzb[..., :nf], beta=1.0, y_amax=nxt[1])
        else:
            d_in = torch.empty(dzb.shape[:3] + (nf,), device=dzb.device, dtype=torch.float32)
            E.conv_backward_data(spec(nf, cw), dzb, None, E.view_of(d_in), dz_amax=dba, packed=packed_in,
                                 res=dzb[..., :nf], beta=5.0)
        # the four growth convs: one weight-gradient launch (dz4..dz1 are adjacent, conv_i reads buf[:c0_i])
        subs = [((4 - i) * gc, gc, nf + (i - 1) * gc, W[i], E.inplace_grad(W[i]), E.inplace_grad(Bi[i]))
                for i in (1, 2, 3, 4)]
        res = E.conv_backward_weight(spec(nf + 3 * gc, 4 * gc), dzb[..., nf: cw], E.view_of(buf[..., :nf + 3 * gc]), W[1],
                                     x_amax=buf_amax, dz_amax=dba, sub_blocks=subs, defer=defer)
        for i, (dwi, dbi) in zip((1, 2, 3, 4), res):
            G[f'{key}conv{i}.weight'], G[f'{key}conv{i}.bias'] = dwi, dbi
        return d_in

    @staticmethod
    def backward(ctx, dout):
        net, names, S = ctx.net, ctx.names, ctx.saved
        P = dict(zip(names, ctx.saved_tensors))
        nf, gc, nb = net.nf, net.gc, net.nb
        cw = nf + 4 * gc
        G = {}

        def spec(cin, cout, stride=1, ups=0):
            return E.LayerSpec('c', cin, cout, 3, stride, NONE, ups)

        f16 = E.PRECISION == 'f16x3'
        # one weight-gradient launch per dense block instead of four (halo-tile path only: it
        # carries the bias sums along); MT355_RDB_BATCHED_WGRAD=0 restores the per-layer launches
        batched_wgrad = E.USE_HALO and E.PRECISION != 'f32' and os.environ.get('MT355_RDB_BATCHED_WGRAD', '1') != '0'
        # gather form of the dense-block backward (see _rdb_backward_gather); MT355_RDB_GATHER=0: layer by layer
        gather = batched_wgrad and os.environ.get('MT355_RDB_GATHER', '1') != '0'
        # the launch that ends a dense block's backward writes the next block's dz5 window (MT355_RDB_FUSED_DZ5=0: separate passes)
        fused_dz5 = os.environ.get('MT355_RDB_FUSED_DZ5', '1') != '0'

        def wgrad(key, sp, dz, xin, scale=1.0, x_amax=None, dz_amax=None):
            dw, db = E.conv_backward_weight(sp, dz, E.view_of(xin), P[key + '.weight'], scale=scale, x_amax=x_amax,
                                            dz_amax=dz_amax, grad_w=E.inplace_grad(P[key + '.weight']),
                                            grad_b=E.inplace_grad(P[key + '.bias']))
            G[key + '.weight'], G[key + '.bias'] = dw, db

        def bdata(key, sp, dz, dx, **kw):
            E.conv_backward_data(sp, dz, P[key + '.weight'], E.view_of(dx), **kw)
            return dx

        hr, u2, u1, y, zbuf = S['hr'], S['u2'], S['u1'], S['y'], S['z']
        dz = dout.contiguous()
        # conv_last <- hr_conv <- upconv2 <- upconv1
        sp = spec(nf, net.out_nc)
        wgrad('conv_last', sp, dz, hr)
        dz = bdata('conv_last', sp, dz, torch.empty_like(hr), dsrc=hr, dact=LRELU)
        sp = spec(nf, nf)
        wgrad('hr_conv', sp, dz, u2)
        dz = bdata('hr_conv', sp, dz, torch.empty_like(u2), dsrc=u2, dact=LRELU)
        for key, src in (('upconv2', u1), ('upconv1', y)):
            sp = spec(nf, nf, ups=1)
            wgrad(key, sp, dz, src)
            dup = torch.empty_like(dz)                       # grad wrt the up-sampled input
            bdata(key, spec(nf, nf), dz, dup)
            dsrc = torch.empty_like(src)
            L.call('mt_sumpool', C.byref(E.view_of(dup)), C.byref(E.view_of(dsrc)), 1, L.stream_ptr())
            dz = E.dact_(dsrc, src, LRELU) if key == 'upconv2' else dsrc
        dy = dz                                              # grad wrt y = f0 + trunk_conv(z)
        n, h, w, _ = dy.shape
        sp = spec(nf, nf)
        wgrad('trunk_conv', sp, dy, zbuf[..., :nf], x_amax=S['z_amax'])
        dcur = bdata('trunk_conv', sp, dy, torch.empty_like(dy))      # grad wrt the trunk output
        dbuf = torch.empty((n, h, w, cw), device=dy.device, dtype=torch.float32)
        dbuf2 = torch.empty_like(dbuf) if gather else None    # the gather form alternates two gradient buffers
        # The weight gradients of the dense blocks wait for the end of the chain and run as two batched
        # launches (E.WgradBatch); every block then keeps its own gradient buffer until the flush
        # (3 nb x 100 MB at B = 8, 256^2: the card has 288 GB).  MT355_RDB_WGRAD_BATCH=0: one launch per block.
        wbatch = E.WgradBatch() if gather and os.environ.get('MT355_RDB_WGRAD_BATCH', '1') != '0' else None
        bufs = S['bufs']
        if gather and wbatch is not None and fused_dz5 and RRDBNetFn._two_chains(n, dy):
            dcur = RRDBNetFn._trunk_backward_chains(P, G, bufs, dcur, nf, gc, nb, f16, wbatch)
            nb = 0            # (the loop below is skipped)
        for k in range(nb - 1, -1, -1):
            drr = dcur                                        # grad wrt this RRDB's output
            d_rdb, s_rdb = drr, 0.2                           # grad wrt rdb3's output = 0.2 * drr
            if gather:
                # rdb3: dz5 = 0.2 * (0.2 * drr); rdb2 / rdb1 get theirs from the launch that ends the block before
                if wbatch is not None:
                    dbuf, dbuf2 = torch.empty_like(dbuf), torch.empty_like(dbuf)
                dzb, dzn = dbuf, dbuf2
                dba = E.new_amax(dy.device) if f16 else None
                _axpby(drr, 0.04, drr, 0.0, dzb[..., :nf], amax=dba)
                for j in (3, 2, 1):
                    key = f'rrdb_trunk.{k}.rdb{j}.'
                    buf, buf_amax = bufs[3 * k + (j - 1)]
                    nxt = (dzn[..., :nf], E.new_amax(dy.device) if f16 else None) if j > 1 and fused_dz5 else None
                    d_rdb = RRDBNetFn._rdb_backward_gather(key, P, G, buf, buf_amax, dzb, dba, nf, gc, f16, nxt=nxt,
                                                           defer=wbatch)
                    if nxt is not None:
                        dzb, dzn, dba = dzn, (torch.empty_like(dzb) if wbatch is not None else dzb), nxt[1]
                    elif j > 1:          # (A/B path: separate scale and max passes)
                        if wbatch is not None:
                            dzb = torch.empty_like(dzb)
                        dba = E.new_amax(dy.device) if f16 else None
                        _axpby(d_rdb, 0.2, d_rdb, 0.0, dzb[..., :nf], amax=dba)
                dcur = _axpby(d_rdb, 1.0, drr, 1.0, torch.empty_like(drr))
                continue
            for j in (3, 2, 1):
                key = f'rrdb_trunk.{k}.rdb{j}.'
                buf, buf_amax = bufs[3 * k + (j - 1)]
                # running max|.| of what this block's launches write into dbuf (it bounds every window)
                dba = E.new_amax(dy.device) if f16 else None
                # out = 0.2 * conv5(cat) + x  with d(out) = s_rdb * d_rdb
                sp5 = spec(cw, nf)
                wgrad(f'{key}conv5', sp5, d_rdb, buf, scale=0.2 * s_rdb, x_amax=buf_amax)
                E.conv_backward_data(sp5, d_rdb, P[f'{key}conv5.weight'], E.view_of(dbuf), dsrc=buf, dact=LRELU,
                                     dact_c0=nf + 3 * gc, alpha=0.2 * s_rdb, y_amax=dba)
                for i in (4, 3, 2, 1):
                    cin = nf + (i - 1) * gc
                    spi = spec(cin, gc)
                    dzi = dbuf[..., cin: cin + gc]
                    if not batched_wgrad:
                        wgrad(f'{key}conv{i}', spi, dzi, buf[..., :cin], x_amax=buf_amax, dz_amax=dba)
                    E.conv_backward_data(spi, dzi, P[f'{key}conv{i}.weight'], E.view_of(dbuf[..., :cin]), dsrc=buf,
                                         dact=LRELU if i > 1 else NONE, dact_c0=cin - gc if i > 1 else 0,
                                         accumulate=True, dz_amax=dba, y_amax=dba)
                if batched_wgrad:
                    # the four growth convs read channel prefixes of ``buf`` and their (now final) output
                    # gradients are the adjacent windows dbuf[nf : nf + 4 gc]: one weight-gradient launch
                    # over the whole 4 gc x (nf + 3 gc) product, each layer takes its sub-block
                    subs = [((i - 1) * gc, gc, nf + (i - 1) * gc, P[f'{key}conv{i}.weight']) for i in (1, 2, 3, 4)]
                    res = E.conv_backward_weight(spec(nf + 3 * gc, 4 * gc), dbuf[..., nf: nf + 4 * gc],
                                                 E.view_of(buf[..., :nf + 3 * gc]), P[f'{key}conv1.weight'],
                                                 x_amax=buf_amax, dz_amax=dba, sub_blocks=subs)
                    for i, (dw, db) in zip((1, 2, 3, 4), res):
                        G[f'{key}conv{i}.weight'], G[f'{key}conv{i}.bias'] = dw, db
                # grad wrt the block input: dense part + residual path
                d_in = torch.empty((n, h, w, nf), device=dy.device, dtype=torch.float32)
                _axpby(dbuf[..., :nf], 1.0, d_rdb, s_rdb, d_in)
                d_rdb, s_rdb = d_in, 1.0
            # RRDB: out = 0.2 * y3 + x  ->  d(x) = d(rdb1 input) + drr
            dcur = _axpby(d_rdb, 1.0, drr, 1.0, torch.empty_like(drr))
        if wbatch is not None:
            wbatch.flush()
        df0 = _axpby(dcur, 1.0, dy, 1.0, torch.empty_like(dy))
        first = S['first']
        dz = df0
        strides = (1, 1, 2, 1, 2)
        for i in range(4, -1, -1):
            sp = spec(net.in_nc if i == 0 else nf, nf, stride=strides[i])
            wgrad(f'conv_first.{i}', sp, dz, first[i])
            if i > 0:
                dz = bdata(f'conv_first.{i}', sp, dz, torch.empty_like(first[i]))
        ctx.saved = None
        return (None, None, None, None) + tuple(G[nm] for nm in names)


class RRDBNet(nn.Module):
    """model_chn.py:506-555."""

    def __init__(self, in_nc, out_nc, nb=10, nf=64, gc=32):
        super().__init__()
        self.in_nc, self.out_nc, self.nb, self.nf, self.gc = in_nc, out_nc, nb, nf, gc
        self.conv_first = nn.Sequential(
            nn.Conv2d(in_nc, nf, (3, 3), padding=1),
            nn.Conv2d(nf, nf, (3, 3), padding=1),
            nn.Conv2d(nf, nf, (3, 3), stride=(2, 2), padding=1),
            nn.Conv2d(nf, nf, (3, 3), padding=1),
            nn.Conv2d(nf, nf, (3, 3), stride=(2, 2), padding=1))
        rrdb_block_f = functools.partial(RRDB, nf=nf, gc=gc)
        self.rrdb_trunk = nn.Sequential(*[rrdb_block_f() for _ in range(nb)])
        self.trunk_conv = nn.Conv2d(nf, nf, (3, 3), (1, 1), 1, bias=True)
        self.upconv1 = nn.Conv2d(nf, nf, (3, 3), (1, 1), 1, bias=True)
        self.upconv2 = nn.Conv2d(nf, nf, (3, 3), (1, 1), 1, bias=True)
        self.hr_conv = nn.Conv2d(nf, nf, (3, 3), (1, 1), 1, bias=True)
        self.conv_last = nn.Conv2d(nf, out_nc, (3, 3), (1, 1), 1, bias=True)

    def forward4(self, x4):
        """NHWC4 in -> NHWC4 out."""
        names, params = zip(*self.named_parameters())
        return RRDBNetFn.apply(x4, self, names, E.will_differentiate(x4, *params), *params)

    def forward(self, x):
        """(N,C,H,W) -> (N,out_nc,H,W), model_chn.py:538-555."""
        return E.FromNHWC4Fn.apply(self.forward4(E.ToNHWC4Fn.apply(x)), self.out_nc)


class RRDB(nn.Module):
    """model_chn.py:558-585 (parameter container; arithmetic in RRDBNetFn)."""

    def __init__(self, nf, gc=32):
        super().__init__()
        self.rdb1 = ResidualDenseBlock5C(nf, gc)
        self.rdb2 = ResidualDenseBlock5C(nf, gc)
        self.rdb3 = ResidualDenseBlock5C(nf, gc)


class ResidualDenseBlock5C(nn.Module):
    """model_chn.py:588-623 (parameter container)."""

    def __init__(self, nf=64, gc=32, bias=True):
        super().__init__()
        self.conv1 = nn.Conv2d(nf, gc, (3, 3), (1, 1), 1, bias=bias)
        self.conv2 = nn.Conv2d(nf + gc, gc, (3, 3), (1, 1), 1, bias=bias)
        self.conv3 = nn.Conv2d(nf + 2 * gc, gc, (3, 3), (1, 1), 1, bias=bias)
        self.conv4 = nn.Conv2d(nf + 3 * gc, gc, (3, 3), (1, 1), 1, bias=bias)
        self.conv5 = nn.Conv2d(nf + 4 * gc, nf, (3, 3), (1, 1), 1, bias=bias)


class CHN(LightningModule):
    """model_chn.py:18-503 (training / validation path)."""
    LOSSES_NAMES = ['loss_nh', 'loss_vh', 'loss_nvh', 'loss_perceptual', 'loss_grad']

    def __init__(self, model_vgg, model_lpips, model_aligner, **kwargs):
        super().__init__()
        self.nn = RRDBNet(in_nc=9, out_nc=3, nb=20)
        self.register_buffer('mean', torch.as_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1))
        self.register_buffer('std', torch.as_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1, 1))
        self.model_vgg = model_vgg
        self.model_lpips = model_lpips
        self.model_aligner = model_aligner
        self.kwargs = kwargs

    def forward(self, x_target, v_target, x_refs_aligned, v_refs_aligned, v_maps):
        """model_chn.py:44-85."""
        b, c, f, h, w = x_refs_aligned.size()
        mean, std = self.mean.reshape(3), self.std.reshape(3)
        scale, shift = (1.0 / std).contiguous(), (-mean / std).contiguous()
        nn_input = ops.pack_chn_input(x_target, v_target, x_refs_aligned, v_refs_aligned, v_maps, scale, shift)
        out4 = self.nn.forward4(nn_input)                                  # (b*f,h,w,8)
        # de-normalise, clamp, compose with the visible target pixels: one fused kernel each way
        y_hat, y_hat_comp = ops.ComposeClampFn.apply(out4, x_target, v_target, mean.contiguous(), std.contiguous(), b, f)
        return y_hat, y_hat_comp

    def _step(self, batch):
        (x, m), y, info = batch
        t, r_list = CHN.get_indexes(x.size(2))
        x_ref_aligned, v_ref_aligned, v_map = self.model_aligner.align(
            x[:, :, t], m[:, :, t], take(x, 2, r_list), take(m, 2, r_list))
        y_hat, y_hat_comp = self(x[:, :, t], 1 - m[:, :, t], x_ref_aligned, v_ref_aligned, v_map)
        return self.compute_loss(y[:, :, t], (1 - m)[:, :, t], y_hat, y_hat_comp, v_map)

    def training_step(self, batch, batch_idx):
        """model_chn.py:256-280."""
        loss, loss_items = self._step(batch)
        self._log_losses(loss, loss_items, 'training')
        return loss

    def validation_step(self, batch, batch_idx):
        """model_chn.py:282-307: the losses and the PSNR / SSIM (/ LPIPS when a model is given)
        of the first reference's composite, all of it on the device."""
        (x, m), y, info = batch
        t, r_list = CHN.get_indexes(x.size(2))
        x_ref_aligned, v_ref_aligned, v_map = self.model_aligner.align(
            x[:, :, t], m[:, :, t], take(x, 2, r_list), take(m, 2, r_list))
        y_hat, y_hat_comp = self(x[:, :, t], 1 - m[:, :, t], x_ref_aligned, v_ref_aligned, v_map)
        loss, loss_items = self.compute_loss(y[:, :, t], (1 - m)[:, :, t], y_hat, y_hat_comp, v_map)
        self._log_losses(loss, loss_items, 'validation')
        self._log_metrics(y_hat_comp[:, :, 0], y[:, :, t], 'validation')
        return loss

    def _log_metrics(self, y_hat_comp, y_target, split):
        """model_chn.py:408-422."""
        from .utils import MeasuresUtils
        with torch.no_grad():
            self.log('measures_psnr_{}'.format(split), MeasuresUtils.psnr(y_hat_comp, y_target))
            self.log('measures_ssim_{}'.format(split), MeasuresUtils.ssim(y_hat_comp, y_target))
            if self.model_lpips is not None:
                self.log('measures_lpips_{}'.format(split), MeasuresUtils.lpips(y_hat_comp, y_target, self.model_lpips))

    # ---- test-time inpainting drivers (model_chn.py:87-254, 309-323) --------------------------------
    # The three algorithms are sequences of ONE primitive -- "fill the hole of a target frame from one
    # reference frame": align, hallucinate, shrink the hole by what the reference could see, re-fill
    # what is left with the fill colour -- steered by the percentage of the frame that is still a
    # hole.  ``_HoleFiller`` owns that primitive and its state (frame, hole mask, last composite) on
    # the device.  The percentage never goes to the host: it becomes a device-side predicate; a pass
    # can be *committed conditionally* on the device (``torch.where`` on the state), and the only
    # thing the host ever reads is the one-byte "go on?" answer, through a pinned non-blocking copy.
    # Drivers work on COPIES of x and m (the reference's inpaint_ip / inpaint_cp update their
    # arguments in place through unsqueeze views); the returned sequences are the same.
    class _HoleFiller:
        def __init__(self, model, x_t, m_t, e):
            self.model, self.e = model, float(e)
            self.x, self.m = x_t.contiguous(), m_t.contiguous()          # (1,3,H,W), (1,1,H,W)
            self.fill = torch.as_tensor([0.485, 0.456, 0.406], dtype=torch.float32, device=x_t.device).view(1, 3, 1, 1)
            self.y = None                                                 # last composite (1,3,H,W)
            self._flag = torch.empty((1,), dtype=torch.uint8).pin_memory() if x_t.is_cuda else torch.empty((1,), dtype=torch.uint8)
            self._ev = torch.cuda.Event() if x_t.is_cuda else None

        @torch.no_grad()
        def fill_from(self, x_r, m_r, commit=None):
            """One pass against reference (x_r (1,3,H,W), m_r (1,1,H,W)).  ``commit``: optional device
            bool -- the pass only takes effect where it is true (a speculative pass costs a launch
            sequence, never a wrong pixel).  Returns the hole percentage after the pass as a device
            scalar (model_chn.py:131,186,246)."""
            xa, va, v_map = self.model.model_aligner.align(self.x, self.m, x_r.unsqueeze(2).contiguous(),
                                                           m_r.unsqueeze(2).contiguous())
            _, y_comp = self.model(self.x, (1 - self.m).contiguous(), xa, va, v_map)
            m_new = self.m - v_map[:, :, 0]
            x_new = (1 - m_new) * y_comp[:, :, 0] + m_new * self.fill
            y_new = y_comp[:, :, 0]
            if commit is not None:
                m_new, x_new = torch.where(commit, m_new, self.m), torch.where(commit, x_new, self.x)
                y_new = torch.where(commit, y_new, self.y) if self.y is not None else y_new
            self.m, self.x, self.y = m_new, x_new, y_new
            return m_new.sum() * 100 / m_new.numel()

        def host(self, flag):
            """The one-byte answer the control flow needs (pinned copy + event: the stream is not drained
            by a .item() of a reduction, and nothing else of the pass is waited for)."""
            if self._ev is None:
                return bool(flag)
            self._flag.copy_(flag.to(torch.uint8).reshape(1), non_blocking=True)
            self._ev.record()
            self._ev.synchronize()
            return bool(self._flag[0])

    @torch.no_grad()
    def _align_forward(self, x_t, m_t, x_r, m_r):
        xa, va, v_map = self.model_aligner.align(x_t.contiguous(), m_t.contiguous(), x_r.contiguous(), m_r.contiguous())
        y_hat, y_hat_comp = self(x_t.contiguous(), (1 - m_t).contiguous(), xa, va, v_map)
        return y_hat_comp, v_map

    def _fill_until_small(self, filler, candidates, ref_of):
        """Consume reference candidates until the hole is below the threshold or none are left (at
        least one pass always runs): the loop of model_chn.py:108-131 / 160-186."""
        more = True
        for k, r in enumerate(candidates):
            if not more:
                break
            inp_per = filler.fill_from(*ref_of(r))
            more = filler.host(inp_per > filler.e) if k + 1 < len(candidates) else False
        return filler

    @torch.no_grad()
    def inpaint_ff(self, x, m, s=1, D=20, e=1):
        """Frame-by-Frame algorithm, model_chn.py:87-133.  x (3,F,H,W), m (1,F,H,W) -> (3,F,H,W): every
        frame is filled independently from its neighbours by distance."""
        y_inpainted = torch.zeros_like(x)
        for t in range(x.size(1)):
            filler = CHN._HoleFiller(self, x[:, t].unsqueeze(0), m[:, t].unsqueeze(0), e)
            self._fill_until_small(filler, CHN.get_indexes_ff(t, x.size(1), s=s, D=D),
                                   lambda r: (x[:, r].unsqueeze(0), m[:, r].unsqueeze(0)))
            y_inpainted[:, t] = filler.y[0]
        return y_inpainted

    @torch.no_grad()
    def inpaint_ip(self, x, m, s=1, D=20, e=1):
        """Inpaint-and-Propagate algorithm, model_chn.py:135-190: frames from the centre outwards, each
        one first against the frames already inpainted (now hole-free), then against raw neighbours."""
        y_seq, m_seq = x.clone(), m.clone()
        order = sorted(range(x.size(1)), key=lambda t: abs(t - x.size(1) // 2))
        for t in order:
            filler = CHN._HoleFiller(self, y_seq[:, t].unsqueeze(0), m_seq[:, t].unsqueeze(0), e)
            self._fill_until_small(filler, CHN.get_indexes_ip(t, order, s, D),
                                   lambda r: (y_seq[:, r].unsqueeze(0), m_seq[:, r].unsqueeze(0)))
            y_seq[:, t], m_seq[:, t] = filler.y[0], 0
        return y_seq

    @torch.no_grad()
    def inpaint_cp(self, x, m, N=20, s=1, e=1):
        """Copy-and-Propagate algorithm, model_chn.py:192-254: N sweeps over alternating frame sets, each
        frame copies from its two neighbours at distance s; a frame is closed (hallucinated content
        kept, mask cleared) once its hole is below e percent or in the last two sweeps.  The only host
        read is one vector of "is this frame still open" flags per sweep (a sweep's targets and
        references are disjoint sets, so the flags cannot change inside it); closing a frame is a
        device-side conditional commit."""
        y_seq, m_seq = x.clone(), m.clone()
        f = x.size(1)
        for i in range(N):
            t_list = [t for t in range(f) if (t // s) % (s if s > 1 else 2) == i % 2]
            if not t_list:
                continue
            open_flags = (m_seq[0, t_list].flatten(1).sum(1) != 0).tolist()          # ONE host read per sweep
            for t, is_open in zip(t_list, open_flags):
                if not is_open:
                    continue
                for r in (t - s, t + s):
                    if not 0 <= r < f:
                        continue
                    filler = CHN._HoleFiller(self, y_seq[:, t].unsqueeze(0), m_seq[:, t].unsqueeze(0), e)
                    inp_per = filler.fill_from(y_seq[:, r].unsqueeze(0), m_seq[:, r].unsqueeze(0))
                    close = (inp_per < e) if i < N - 2 else torch.ones_like(inp_per, dtype=torch.bool)
                    # closed: the composite stays and the mask is cleared; open: copied pixels + fill colour
                    y_seq[:, t] = torch.where(close, filler.y, filler.x)[0]
                    m_seq[:, t] = torch.where(close, torch.zeros_like(filler.m), filler.m)[0]
        return y_seq

    def test_step(self, batch, batch_idx):
        """model_chn.py:309-323: the three drivers on the first clip of the batch.  The reference
        writes .avi files through cv2 (out of scope); the sequences are returned instead."""
        (x, m), y, info = batch
        return {'ff': self.inpaint_ff(x[0], m[0]), 'ip': self.inpaint_ip(x[0], m[0]),
                'cp': self.inpaint_cp(x[0], m[0])}

    @staticmethod
    def get_indexes_ff(t, max_t, s, D):
        """model_chn.py:459-481."""
        ref_candidates = list(range(max_t))
        ref_candidates.pop(t)
        ref_candidates_dist = list(map(lambda x: abs(x - t), ref_candidates))
        ref_candidates_sorted = [r[1] for r in sorted(zip(ref_candidates_dist, ref_candidates))]
        return list(filter(lambda x: abs(x - t) <= D and abs(x - t) % s == 0, ref_candidates_sorted))

    @staticmethod
    def get_indexes_ip(t, t_list, s, D):
        """model_chn.py:483-503."""
        t_list_inpainted = list(reversed(t_list[:t_list.index(t)]))
        t_list_ff = CHN.get_indexes_ff(t, len(t_list), s, D)
        t_list_ff = [t_item for t_item in t_list_ff if t_item not in t_list_inpainted]
        return t_list_inpainted + t_list_ff

    def compute_loss(self, y_target, v_target, y_hat, y_hat_comp, v_map):
        """model_chn.py:324-376.  The reference passes 5-D tensors to the Sobel
        conv (:371-373) and raises for more than one reference frame; here the
        frame axis is folded into the batch (identical for one frame), as
        :365-366 already does for the perceptual term."""
        b, c, h, w = y_target.size()
        f = y_hat.size(2)
        target_img = y_target.unsqueeze(2).repeat(1, 1, f, 1, 1)
        nh_mask = v_target.unsqueeze(2).repeat(1, 1, f, 1, 1)
        loss_nh = LossesUtils.masked_l1(y_hat, target_img, nh_mask, reduction='sum', weight=0.50)
        vh_mask = v_map
        loss_vh = LossesUtils.masked_l1(y_hat, target_img, vh_mask, reduction='sum', weight=2)
        nvh_mask = (1 - nh_mask) - vh_mask
        loss_nvh = LossesUtils.masked_l1(y_hat_comp, target_img, nvh_mask, reduction='sum', weight=1)
        y_hat_f = y_hat.transpose(1, 2).reshape(-1, c, h, w)
        target_f = target_img.transpose(1, 2).reshape(-1, c, h, w)
        # perceptual term (utils.py:171-191 on the folded tensors, :364-369).  The folded target holds every
        # clip's target frame f times: its VGG taps are computed once per clip and repeated (bit-identical
        # images give bit-identical features), which removes 3/8 of the perceptual VGG pass at f = 4.
        input_vgg = self.model_vgg(y_hat_f.contiguous())
        loss_perceptual = 0
        for a, t in zip(input_vgg, self.model_vgg(y_target.contiguous())):
            t_rep = t.unsqueeze(1).expand(-1, f, -1, -1, -1).reshape(b * f, *t.shape[1:])
            loss_perceptual = loss_perceptual + LossesUtils.l1(a, t_rep)
        loss_perceptual = loss_perceptual * 0.50 / len(input_vgg)
        loss_grad = LossesUtils.grad(y_hat_f, target_f, reduction='mean', weight=1)
        loss = loss_nh + loss_vh + loss_nvh + loss_perceptual + loss_grad
        return loss, [loss_nh, loss_vh, loss_nvh, loss_perceptual, loss_grad]

    def trainable_parameters(self):
        """Only the RRDBNet receives gradients (aligner, VGG, LPIPS are frozen
        in effect: their .grad stays None in the reference)."""
        return list(self.nn.parameters())

    def configure_optimizers(self):
        """model_chn.py:378-391."""
        optimizer = FlatAdam(self.trainable_parameters(), lr=self.kwargs['lr'], all_params=reference_parameter_list(self))
        lr_scheduler = torch.optim.lr_scheduler.StepLR(
            optimizer, step_size=self.kwargs['lr_scheduler_step_size'], gamma=self.kwargs['lr_scheduler_gamma'])
        return {'optimizer': optimizer, 'lr_scheduler': lr_scheduler}

    def _log_losses(self, loss, loss_items, split):
        """model_chn.py:393-406."""
        self.log('loss_{}'.format(split), loss)
        for i, loss_item_id in enumerate(CHN.LOSSES_NAMES):
            self.log('loss_{}_{}'.format(loss_item_id, split), loss_items[i])

    @staticmethod
    def get_indexes(size):
        """model_chn.py:442-457."""
        t, r_list = size // 2, list(range(size))
        r_list.pop(t)
        return t, r_list
