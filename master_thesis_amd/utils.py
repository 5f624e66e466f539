"""``FlowsUtils`` / ``LossesUtils`` / ``TransformsUtils`` of the hot path on the
mt355 kernels.  Same names, argument meaning and return layouts as
``master_thesis/utils.py`` (cited per method); every tensor lives on the GPU.
"""
import torch

from . import ops


_index_cache = {}


def take(x, dim, idx):
    """x.index_select(dim, idx) for a Python list ``idx`` without building the
    index tensor from host memory on every call (that H2D copy is not allowed
    while a hipGraph is being captured): runs of consecutive indices become
    slices, anything else uses a cached device index."""
    idx = list(idx)
    runs, start = [], 0
    for i in range(1, len(idx) + 1):
        if i == len(idx) or idx[i] != idx[i - 1] + 1:
            runs.append((idx[start], idx[i - 1] + 1))
            start = i
    if len(runs) <= 2:
        parts = [x.narrow(dim, a, b - a) for a, b in runs]
        return parts[0].contiguous() if len(parts) == 1 else torch.cat(parts, dim=dim)
    key = (x.device, tuple(idx))
    if key not in _index_cache:
        _index_cache[key] = torch.tensor(idx, device=x.device, dtype=torch.long)
    return x.index_select(dim, _index_cache[key])


class FlowsUtils:
    @staticmethod
    def align_set(x, v, flow):
        """utils.py:77-104.  x (B,C,F,H,W), v (B,1,F,H,W), flow (B,F,H,W,2) ->
        bilinear-warped x and nearest-warped v (one fused kernel)."""
        return ops.align_set(x, v, flow, align_corners=True)

    @staticmethod
    def resize_flow(flow, size, mode='nearest'):
        """utils.py:106-126."""
        return ops.resize_flow(flow, size, mode)


class TransformsUtils:
    @staticmethod
    def resize_set(x, v, y, size):
        """utils.py:511-549: bilinear for x and y, nearest for v."""
        s = (size, size)
        return (ops.resize_planar(x, s, 'bilinear'), ops.resize_planar(v, s, 'nearest'),
                ops.resize_planar(y, s, 'bilinear'))

    @staticmethod
    def resize_set_bis(x_target, m_target, x_ref, m_ref, size):
        """utils.py:551-595 (identity when the size already matches)."""
        h, w = x_ref.shape[-2:]
        if (h, w) == tuple(size):
            return x_target, m_target, x_ref, m_ref
        return (ops.resize_planar(x_target, size, 'bilinear'), ops.resize_planar(m_target, size, 'nearest'),
                ops.resize_planar(x_ref, size, 'bilinear'), ops.resize_planar(m_ref, size, 'nearest'))


class LossesUtils:
    @staticmethod
    def masked_l1(y_hat, y, mask, batch_mask=None, reduction='mean', weight=1):
        """utils.py:138-169.  The batch mask is applied on the device as a
        per-sample weight (no host sync, no boolean gather): 'mean' divides by
        the number of SELECTED elements, 'sum' by the selected mask sum, and an
        all-false mask yields 0 (the reference returns ``zeros(1)``; here the
        zero is a 0-d tensor)."""
        w = None
        if batch_mask is not None:
            w = torch.as_tensor(batch_mask, device=y_hat.device).to(torch.float32)
        if mask.shape != y_hat.shape:
            assert mask.shape[0] == y_hat.shape[0] and mask.shape[1] == 1 \
                and mask.shape[2:] == y_hat.shape[2:], (mask.shape, y_hat.shape)
        acc = ops.L1SumFn.apply(y_hat, y, mask, w, reduction == 'sum')
        if reduction == 'sum':
            return weight * acc[0] / (acc[1] + 1e-9)
        per_sample = y_hat.numel() // y_hat.shape[0]
        if w is None:
            return weight * acc[0] / float(y_hat.numel())
        return weight * acc[0] / (w.sum().clamp(min=1.0) * float(per_sample))

    @staticmethod
    def l1(a, b):
        """F.l1_loss(a, b) (model_dfpn.py:257, utils.py:189)."""
        n = a.numel()
        acc = ops.L1SumFn.apply(a.reshape(1, 1, n), b.reshape(1, 1, n), None, None, False)
        return acc[0] / float(n)

    @staticmethod
    def perceptual(y_hat, y, model_vgg, weight=1):
        """utils.py:171-191 (the VGG taps are detached: no gradient)."""
        input_vgg = model_vgg(y_hat.contiguous())
        target_vgg = model_vgg(y.contiguous())
        loss = 0
        for a, b in zip(input_vgg, target_vgg):
            loss = loss + LossesUtils.l1(a, b)
        return loss * weight / len(input_vgg), input_vgg, target_vgg

    @staticmethod
    def grad(y_hat, y, reduction, weight=1):
        """utils.py:193-224 (Sobel H and V responses, L1)."""
        acc = ops.SobelL1Fn.apply(y_hat, y)[0]
        n, c, h, w = y_hat.shape
        if reduction == 'sum':   # mask of ones: sum / (numel + 1e-9)
            return weight * acc / (float(n * 2 * c * h * w) + 1e-9)
        return weight * acc / float(n * 2 * c * h * w)
