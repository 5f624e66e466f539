"""Autograd wrappers of the streaming kernels (warp, resize, packs, L1-type
losses, correlation).  Every function here launches libmt355 kernels on the
current HIP stream and raises if the library is missing."""
import ctypes as C

import torch

from . import _lib as L
from .engine import cpad, view_of


def _chk(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32, (t.device, t.dtype)


# ------------------------------------------------------------------ warp ---
def _warp_layout(x, v, xo, vo):
    b, c, f, h, w = x.shape
    assert x.stride(4) == 1 and x.stride(3) == w
    lay = L.WarpLayout(b, f, c, h, w, x.stride(0), x.stride(1), x.stride(2), 0, 0,
                       xo.stride(0), xo.stride(1), xo.stride(2), 0, 0)
    if v is not None:
        assert v.stride(4) == 1 and v.stride(3) == w
        lay.vb, lay.vf = v.stride(0), v.stride(2)
        lay.wb, lay.wf = vo.stride(0), vo.stride(2)
    return lay


class WarpFn(torch.autograd.Function):
    """FlowsUtils.align_set (utils.py:77-104): differentiable wrt the grid."""

    @staticmethod
    def forward(ctx, x, v, flow, align_corners, want_idx):
        _chk(x, v, flow)
        b, c, f, h, w = x.shape
        flow = flow.contiguous()
        assert flow.shape == (b, f, h, w, 2)
        if x.stride(4) != 1 or x.stride(3) != w:
            x = x.contiguous()
        if v is not None and (v.stride(4) != 1 or v.stride(3) != w):
            v = v.contiguous()
        xo = torch.empty((b, c, f, h, w), device=x.device, dtype=torch.float32)
        vo = torch.empty((b, 1, f, h, w), device=x.device, dtype=torch.float32) if v is not None else None
        idx = torch.empty((b, f, h, w, 4), device=x.device, dtype=torch.int32) if want_idx else None
        lay = _warp_layout(x, v, xo, vo)
        L.call('mt_warp_fwd', L.ptr(x), L.ptr(v), L.ptr(flow), C.byref(lay), int(align_corners), L.ptr(xo),
               L.ptr(vo), L.ptr(idx), L.stream_ptr())
        ctx.save_for_backward(x, flow)
        ctx.ac = int(align_corners)
        ctx.mark_non_differentiable(*[t for t in (vo, idx) if t is not None])
        return xo, vo, idx

    @staticmethod
    def backward(ctx, dxo, _dvo, _didx):
        x, flow = ctx.saved_tensors
        if not ctx.needs_input_grad[2]:
            return None, None, None, None, None
        dxo = dxo.contiguous()
        dgrid = torch.empty_like(flow)
        lay = _warp_layout(x, None, dxo, None)
        L.call('mt_warp_bwd_grid', L.ptr(x), L.ptr(flow), L.ptr(dxo), C.byref(lay), ctx.ac, L.ptr(dgrid),
               L.stream_ptr())
        return None, None, dgrid, None, None


def align_set(x, v, flow, align_corners=True, want_idx=False):
    xo, vo, idx = WarpFn.apply(x, v, flow, align_corners, want_idx)
    return (xo, vo, idx) if want_idx else (xo, vo)


# ---------------------------------------------------------------- resize ---
class ResizeFn(torch.autograd.Function):
    """F.interpolate (nearest, or bilinear align_corners=False) on a tensor
    seen as [planes][H][W][CI]: ``x`` is (..., H, W) for ci == 1 or
    (..., H, W, CI) channel-last (flows)."""

    @staticmethod
    def forward(ctx, x, size, mode, ci):
        _chk(x)
        x = x.contiguous()
        if ci == 1:
            lead, (h, w) = x.shape[:-2], x.shape[-2:]
            out = torch.empty(tuple(lead) + tuple(size), device=x.device, dtype=torch.float32)
        else:
            lead, (h, w) = x.shape[:-3], x.shape[-3:-1]
            assert x.shape[-1] == ci
            out = torch.empty(tuple(lead) + tuple(size) + (ci,), device=x.device, dtype=torch.float32)
        planes = 1
        for d in lead:
            planes *= d
        ho, wo = size
        L.call('mt_resize_fwd', L.ptr(x), h * w * ci, w * ci, ci, planes, h, w, L.ptr(out), ho * wo * ci, wo * ci,
               ci, ho, wo, ci, 1 if mode == 'bilinear' else 0, L.stream_ptr())
        ctx.geom = (planes, h, w, ho, wo, ci, mode, x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        planes, h, w, ho, wo, ci, mode, shape = ctx.geom
        assert mode == 'bilinear', 'nearest resize is only used on no-grad tensors'
        dout = dout.contiguous()
        dx = torch.empty(shape, device=dout.device, dtype=torch.float32)       # (gather form: every element is written)
        L.call('mt_resize_bilinear_bwd', L.ptr(dout), ho * wo * ci, wo * ci, ci, planes, ho, wo, L.ptr(dx),
               h * w * ci, w * ci, ci, h, w, ci, L.stream_ptr())
        return dx, None, None, None


def resize_planar(x, size, mode):
    """x (..., H, W) -> (..., size[0], size[1])."""
    return ResizeFn.apply(x, tuple(size), mode, 1)


def resize_flow(flow, size, mode='nearest'):
    """FlowsUtils.resize_flow (utils.py:106-126) without the two permutes."""
    return ResizeFn.apply(flow, tuple(size), mode, 2)


# ----------------------------------------------------------------- packs ---
def _src(t, sb, sr, sc, sy, sx, c, c_off, scale=None, shift=None):
    return L.PackSrc(t.data_ptr(), sb, sr, sc, sy, sx, c, c_off,
                     scale.data_ptr() if scale is not None else None,
                     shift.data_ptr() if shift is not None else None)


class PackFlowEstimatorFn(torch.autograd.Function):
    """The 10-channel FlowEstimator input (model_dfpn.py:733-741) written
    straight into a (B*R, H, W, cpad(10)) NHWC4 tensor; the images are normalised on
    the fly ((x-mean)/std, model_dfpn.py:71-72).  Differentiable wrt flow_pre."""

    @staticmethod
    def forward(ctx, x_t, m_t, x_r, m_r, flow_pre, scale, shift):
        _chk(x_t, m_t, x_r, m_r, flow_pre)
        x_t, m_t, x_r, m_r = x_t.contiguous(), m_t.contiguous(), x_r.contiguous(), m_r.contiguous()
        flow_pre = flow_pre.contiguous()
        b, c, r, h, w = x_r.shape
        out = torch.empty((b * r, h, w, cpad(10)), device=x_r.device, dtype=torch.float32)
        hw = h * w
        srcs = (L.PackSrc * 5)(
            _src(x_r, c * r * hw, hw, r * hw, w, 1, 3, 0, scale, shift),
            _src(x_t, c * hw, 0, hw, w, 1, 3, 3, scale, shift),
            _src(m_r, r * hw, hw, 0, w, 1, 1, 6),
            _src(m_t, hw, 0, 0, w, 1, 1, 7),
            _src(flow_pre, r * hw * 2, hw * 2, 1, w * 2, 2, 2, 8))
        ov = view_of(out)
        L.call('mt_pack_pairs', srcs, 5, b, r, C.byref(ov), L.stream_ptr())
        ctx.shape = (b, r, h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        b, r, h, w = ctx.shape
        dflow = dout[..., 8:10].reshape(b, r, h, w, 2).contiguous() if ctx.needs_input_grad[4] else None
        return None, None, None, None, dflow, None, None


def pack_chn_input(x_t, v_t, x_ra, v_ra, v_maps, scale, shift):
    """The 9-channel CHN input (model_chn.py:70-80) as (B*R, H, W, cpad(9)) NHWC4."""
    _chk(x_t, v_t, x_ra, v_ra, v_maps)
    x_t, v_t, x_ra, v_ra, v_maps = (t.contiguous() for t in (x_t, v_t, x_ra, v_ra, v_maps))
    b, c, r, h, w = x_ra.shape
    hw = h * w
    out = torch.empty((b * r, h, w, cpad(9)), device=x_ra.device, dtype=torch.float32)
    srcs = (L.PackSrc * 5)(
        _src(x_t, c * hw, 0, hw, w, 1, 3, 0, scale, shift),
        _src(x_ra, c * r * hw, hw, r * hw, w, 1, 3, 3, scale, shift),
        _src(v_t, hw, 0, 0, w, 1, 1, 6),
        _src(v_ra, r * hw, hw, 0, w, 1, 1, 7),
        _src(v_maps, r * hw, hw, 0, w, 1, 1, 8))
    ov = view_of(out)
    L.call('mt_pack_pairs', srcs, 5, b, r, C.byref(ov), L.stream_ptr())
    return out


# ----------------------------------------------------------- correlation ---
class ComposeClampFn(torch.autograd.Function):
    """Output stage of CHN.forward (model_chn.py:81-85) from the network's NHWC output:
    (y_hat, y_hat_comp) = (clamp(o*std+mean, 0, 1), v*x_t + (1-v)*y_hat), both (B,3,F,H,W).
    Differentiable wrt the network output only (x_t and v are data)."""

    @staticmethod
    def forward(ctx, out4, x_t, v_t, mean3, std3, b, f):
        _chk(out4, x_t, v_t, mean3, std3)
        x_t, v_t = x_t.contiguous(), v_t.contiguous()
        h, w = out4.shape[1:3]
        y_hat = torch.empty((b, 3, f, h, w), device=out4.device, dtype=torch.float32)
        y_comp = torch.empty_like(y_hat)
        L.call('mt_compose_clamp_fwd', C.byref(view_of(out4)), L.ptr(x_t), L.ptr(v_t), L.ptr(mean3), L.ptr(std3), b, f,
               L.ptr(y_hat), L.ptr(y_comp), L.stream_ptr())
        ctx.save_for_backward(out4, v_t, mean3, std3)
        ctx.bf = (b, f)
        return y_hat, y_comp

    @staticmethod
    def backward(ctx, d_hat, d_comp):
        out4, v_t, mean3, std3 = ctx.saved_tensors
        b, f = ctx.bf
        d_hat = d_hat.contiguous() if d_hat is not None else None
        d_comp = d_comp.contiguous() if d_comp is not None else None
        d_o = torch.empty_like(out4)
        L.call('mt_compose_clamp_bwd', C.byref(view_of(out4)), L.ptr(v_t), L.ptr(mean3), L.ptr(std3), b, f, L.ptr(d_hat),
               L.ptr(d_comp), C.byref(view_of(d_o)), L.stream_ptr())
        return d_o, None, None, None, None, None, None


def correlation(ft4, fr4, v_t, v_r, b, r):
    """CorrelationVGG.correlation_masked_4d (model_dfpn.py:534-565) on NHWC4
    features.  ft4 (B,h,w,C), fr4 (B*R,h,w,C); v_t (B,P)/None, v_r (B*R,P)/None.
    Returns (B*R, P_t, P_r).  No gradient (features are no-grad upstream)."""
    _chk(ft4, fr4, v_t, v_r)
    p = ft4.shape[1] * ft4.shape[2]
    cch = ft4.shape[3]
    out = torch.empty((b * r, p, p), device=ft4.device, dtype=torch.float32)
    # the fused P = 256 kernel needs no workspace; the general path normalises into one
    fused = p == 256 and cch % 64 == 0 and cch >= 128
    ws = None if fused else torch.empty(((b + b * r) * p * cch,), device=ft4.device, dtype=torch.float32)
    if v_t is not None:
        v_t, v_r = v_t.contiguous(), v_r.contiguous()
    L.call('mt_corr_masked', C.byref(view_of(ft4)), C.byref(view_of(fr4)), L.ptr(v_t), L.ptr(v_r), b, r,
           L.ptr(out), L.ptr(ws), L.stream_ptr())
    return out


def softmax_3d(x):
    """CorrelationVGG.softmax_3d (model_dfpn.py:567-583).  x (B,R,h,w,h,w) contiguous."""
    _chk(x)
    x = x.contiguous()
    b, r, h, w = x.shape[:4]
    out = torch.empty_like(x)
    L.call('mt_softmax3d', L.ptr(x), b, r, h * w, L.ptr(out), L.stream_ptr())
    return out


# ---------------------------------------------------------------- losses ---
class L1SumFn(torch.autograd.Function):
    """sum_i w[b] |a_i m_i - b_i m_i| and (optionally) sum of w*m, as a
    2-vector [l1_sum, mask_sum].  a, b: (B, C, *S); mask (B, MC, *S) or None;
    w (B,) or None.  Differentiable wrt a and b."""

    @staticmethod
    def forward(ctx, a, b, mask, w, count_mask):
        _chk(a, b, mask, w)
        a, b = a.contiguous(), b.contiguous()
        if mask is not None:
            mask = mask.contiguous()
        bsz, ch = a.shape[0], a.shape[1]
        s = a.numel() // (bsz * ch)
        mc = mask.shape[1] if mask is not None else 1
        acc = torch.zeros((2,), device=a.device, dtype=torch.float32)
        L.call('mt_l1_fwd', L.ptr(a), L.ptr(b), L.ptr(mask), L.ptr(w), bsz, ch, mc, s, int(count_mask), L.ptr(acc),
               L.stream_ptr())
        ctx.save_for_backward(a, b, mask, w)
        ctx.geom = (bsz, ch, mc, s)
        return acc

    @staticmethod
    def backward(ctx, dacc):
        a, b, mask, w = ctx.saved_tensors
        bsz, ch, mc, s = ctx.geom
        scale = dacc[0:1].contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            L.call('mt_l1_bwd', L.ptr(a), L.ptr(b), L.ptr(mask), L.ptr(w), bsz, ch, mc, s, L.ptr(scale), L.ptr(da),
                   L.ptr(db), 0, L.stream_ptr())
        return da, db, None, None, None


class SobelL1Fn(torch.autograd.Function):
    """sum |sobel(a) - sobel(b)| over H and V responses (utils.py:193-224);
    a, b: (N, C, H, W); differentiable wrt a."""

    @staticmethod
    def forward(ctx, a, b):
        _chk(a, b)
        a, b = a.contiguous(), b.contiguous()
        n, c, h, w = a.shape
        acc = torch.zeros((1,), device=a.device, dtype=torch.float32)
        L.call('mt_sobel_l1_fwd', L.ptr(a), L.ptr(b), n * c, h, w, L.ptr(acc), L.stream_ptr())
        ctx.save_for_backward(a, b)
        return acc

    @staticmethod
    def backward(ctx, dacc):
        a, b = ctx.saved_tensors
        n, c, h, w = a.shape
        da = torch.empty_like(a)
        L.call('mt_sobel_l1_bwd', L.ptr(a), L.ptr(b), n * c, h, w, L.ptr(dacc.contiguous()), L.ptr(da),
               L.stream_ptr())
        return da, None
