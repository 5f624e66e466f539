"""``FlowsUtils`` / ``LossesUtils`` / ``TransformsUtils`` of the hot path on the
mt355 kernels.  Same names, argument meaning and return layouts as
``master_thesis/utils.py`` (cited per method); every tensor lives on the GPU.
"""
import ctypes as C

import torch

from . import _lib as L
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


    @staticmethod
    def dilate(images, filter_size, iterations):
        """utils.py:618-640 (cv2.dilate with a MORPH_CROSS element, ``iterations`` times) on the
        device: images (1,F,H,W) -- or any (B,1,F,H,W) batch of mask planes -- one fused launch."""
        return TransformsUtils.dilate_fill(images, None, filter_size, iterations)[0]

    @staticmethod
    def dilate_fill(m, y, filter_size, iterations, fill=(0.485, 0.456, 0.406)):
        """The mask half of MasterThesisDataset.__getitem__ (dataset.py:158-169) in ONE launch:
        m' = dilate(m), x = (1 - m') * y + m' * fill.  m (1,F,H,W) with y (3,F,H,W), or batched
        (B,1,F,H,W) with (B,3,F,H,W); y = None: dilation only.  Returns (m', x or None)."""
        assert m.is_cuda and m.dtype == torch.float32
        batched = m.dim() == 5
        m5 = (m if batched else m.unsqueeze(0)).contiguous()
        b, one, f, h, w = m5.shape
        assert one == 1
        kw, kh = (filter_size, filter_size) if isinstance(filter_size, int) else filter_size   # cv2 sizes are (w, h)
        m_out = torch.empty_like(m5)
        x = fill_t = y5 = None
        if y is not None:
            y5 = (y if batched else y.unsqueeze(0)).contiguous()
            assert y5.shape == (b, 3, f, h, w) and y5.dtype == torch.float32
            x = torch.empty_like(y5)
            fill_t = torch.as_tensor(fill, dtype=torch.float32, device=m.device).contiguous()
        L.call('mt_mask_dilate_fill', L.ptr(m5), L.ptr(y5), L.ptr(fill_t), b, f, h, w, int(kh), int(kw), int(iterations),
               L.ptr(m_out), L.ptr(x), L.stream_ptr())
        if not batched:
            m_out, x = m_out[0], (x[0] if x is not None else None)
        return m_out, x


class MeasuresUtils:
    """utils.py:404-463: validation metrics, kept on the device (the reference moves every image to
    the host and calls scikit-image 0.18.3).  Both return 0-d float64 device tensors: nothing
    synchronises until the logger reads them.  LPIPS stays external (an AlexNet the caller owns)."""

    @staticmethod
    def psnr(y_hat, y):
        """utils.py:407-429 -> skimage peak_signal_noise_ratio(y[f], y_hat[f]) per image: the
        ground truth is image_true; data_range = 1 when min(y) >= 0, else 2 (float images);
        NaN items count as 100, infinite ones (identical images) are dropped from the mean."""
        assert y_hat.shape == y.shape and y.dim() == 4
        y_hat, y = y_hat.contiguous().float(), y.contiguous().float()
        n = y.shape[0]
        out = torch.empty((n, 3), device=y.device, dtype=torch.float64)
        L.call('mt_psnr_sums', L.ptr(y_hat), L.ptr(y), n, y[0].numel(), L.ptr(out), L.stream_ptr())
        mse = out[:, 0] / float(y[0].numel())
        data_range = torch.where(out[:, 1] >= 0, 1.0, 2.0).to(torch.float64)
        items = 10 * torch.log10(data_range ** 2 / mse)
        items = torch.where(torch.isnan(items), torch.full_like(items, 100.0), items)
        keep = torch.isfinite(items)
        return torch.where(keep, items, torch.zeros_like(items)).sum() / keep.sum()

    @staticmethod
    def ssim(y_hat, y, win_size=7, K1=0.01, K2=0.03):
        """utils.py:431-448 -> skimage structural_similarity(..., multichannel=True) per image:
        uniform 7x7 window, sample covariance, float64, data_range = 2 for float images (skimage
        0.18.3 takes dmax - dmin of the dtype), mean of the map over the pixels whose window lies
        inside the image, mean over channels, mean over images."""
        assert y_hat.shape == y.shape and y.dim() == 4
        y_hat, y = y_hat.contiguous().float(), y.contiguous().float()
        n, c, h, w = y.shape
        out = torch.empty((n * c,), device=y.device, dtype=torch.float64)
        L.call('mt_ssim_sums', L.ptr(y), L.ptr(y_hat), n * c, h, w, int(win_size), 2.0, float(K1), float(K2), L.ptr(out),
               L.stream_ptr())
        return (out / float((h - win_size + 1) * (w - win_size + 1))).mean()

    @staticmethod
    def lpips(y_hat, y, model=None):
        """utils.py:450-463 (the network is the caller's; its forward runs wherever it lives)."""
        with torch.no_grad():
            return model.forward(2 * y_hat - 1, 2 * y - 1).flatten().mean()


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


class MovementsUtils:
    """utils.py:227-401: synthetic camera movement for a still image -- the data either side of the
    training step (dataset.py:596-601), kept on the device: the 3x3 matrix algebra is host-side as
    upstream (a few dozen flops), the sampling grids and the bilinear warp of all frames run on the GPU
    (one fused warp launch for the whole sequence)."""

    def __init__(self, max_displacement, max_scaling, max_rotation):
        self.max_displacement, self.max_scaling, self.max_rotation = max_displacement, max_scaling, max_rotation

    def random_affine(self, rng=None):
        """utils.py:243-260 (same draws, same order; the matrix is skimage's
        AffineTransform(translation, scale, rotation).params written out)."""
        import numpy as np
        rng = rng or np.random
        tx, ty = rng.randint(low=-self.max_displacement, high=self.max_displacement, size=2) \
            if self.max_displacement > 0 else (0, 0)
        sx, sy = rng.uniform(low=1 - self.max_scaling, high=1 + self.max_scaling, size=2)
        rot = rng.uniform(low=-self.max_rotation, high=self.max_rotation)
        return torch.tensor([[sx * np.cos(rot), -sy * np.sin(rot), tx], [sx * np.sin(rot), sy * np.cos(rot), ty],
                             [0.0, 0.0, 1.0]], dtype=torch.float32)

    @staticmethod
    def identity_affine():
        return torch.eye(3, dtype=torch.float32)

    @staticmethod
    def affine_inverse(affine):
        return torch.inverse(affine)

    @staticmethod
    def stack_transformations(affine_matrices, t):
        """utils.py:347-377."""
        out = torch.zeros(affine_matrices.size(), dtype=torch.float32)
        out[t] = affine_matrices[t]
        for i in reversed(range(t)):
            out[i] = torch.matmul(torch.inverse(affine_matrices[i]), out[i + 1])
        for i in range(t + 1, len(affine_matrices)):
            out[i] = torch.matmul(affine_matrices[i], out[i - 1])
        return out

    @staticmethod
    def affine2theta(param, h, w):
        """utils.py:379-401."""
        return torch.tensor([[param[0, 0], param[0, 1] * h / w, param[0, 2] * 2 / w + param[0, 0] + param[0, 1] - 1],
                             [param[1, 0] * w / h, param[1, 1], param[1, 2] * 2 / h + param[1, 0] + param[1, 1] - 1]],
                            dtype=torch.float32)

    @staticmethod
    def _affine_grid(theta, h, w):
        """F.affine_grid(theta, [n, c, h, w], align_corners=True) on theta's device: base
        coordinates linspace(-1, 1)."""
        n = theta.shape[0]
        xs = torch.linspace(-1, 1, w, device=theta.device)
        ys = torch.linspace(-1, 1, h, device=theta.device)
        base = torch.stack([xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w),
                            torch.ones((h, w), device=theta.device)], dim=2)
        return torch.matmul(base.view(1, h * w, 3), theta.transpose(1, 2)).view(n, h, w, 2)

    def simulate_movement(self, x, n, affine_matrices=None):
        """utils.py:262-318.  x (C,H,W) on the GPU -> (y (C,n,H,W), flow_inv (n,H,W,2), affine_matrices)."""
        c, h, w = x.size()
        if affine_matrices is None:
            affine_matrices = [self.random_affine() for _ in range(n - 1)]
            affine_matrices = affine_matrices[: n // 2] + [MovementsUtils.identity_affine()] + affine_matrices[n // 2:]
        inv = [MovementsUtils.affine_inverse(a) for a in affine_matrices]
        st = MovementsUtils.stack_transformations(torch.stack(list(affine_matrices)), t=n // 2)
        st_inv = MovementsUtils.stack_transformations(torch.stack(inv), t=n // 2)
        theta = torch.stack([MovementsUtils.affine2theta(a, h, w) for a in st]).to(x.device)
        theta_inv = torch.stack([MovementsUtils.affine2theta(a, h, w) for a in st_inv]).to(x.device)
        flow = MovementsUtils._affine_grid(theta, h, w)
        flow_inv = MovementsUtils._affine_grid(theta_inv, h, w)
        # every frame samples the same still image: (1,C,n,H,W) view, one warp launch
        xs = x.unsqueeze(0).unsqueeze(2).expand(1, c, n, h, w).contiguous()
        y, _ = ops.align_set(xs, None, flow.unsqueeze(0), align_corners=True)
        return y[0], flow_inv, affine_matrices
