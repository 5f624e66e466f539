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
