"""The warp tail of the Copy-and-Paste aligner (``master_thesis/model_cpn.py:75-91``).

Only the warp convention of ``CPN.align`` is on the hot path:
an affine sampling grid (``affine_grid``, align_corners=False) followed by a
bilinear ``grid_sample`` (align_corners=False) of the frames and of the
visibility maps (thresholded at 0.5).  The A_Encoder / A_Regressor networks
that produce ``theta`` need external weights and are not built.
"""
import torch

from . import ops


class CPN(torch.nn.Module):
    @staticmethod
    def affine_grid(theta, size):
        """F.affine_grid(theta, size, align_corners=False): base coordinates
        (2i + 1)/W - 1, as (N,H,W,2).  Tiny host-side tensor algebra."""
        n, _, h, w = size
        xs = (torch.arange(w, device=theta.device, dtype=torch.float32) * 2 + 1) / w - 1
        ys = (torch.arange(h, device=theta.device, dtype=torch.float32) * 2 + 1) / h - 1
        base = torch.stack([xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w),
                            torch.ones((h, w), device=theta.device)], dim=2)       # (H,W,3)
        return torch.matmul(base.view(1, h * w, 3), theta.transpose(1, 2)).view(n, h, w, 2)

    @staticmethod
    def warp(x_refs, m_refs, m_target, theta_rt):
        """x_refs (B,C,F,H,W), m_refs (B,1,F,H,W), m_target (B,1,H,W),
        theta_rt (B*F,2,3) -> (x_aligned, v_aligned, v_maps), model_cpn.py:75-91."""
        b, c, ref_n, h, w = x_refs.size()
        grid = CPN.affine_grid(theta_rt, [theta_rt.size(0), c, h, w]).reshape(b, ref_n, h, w, 2)
        x_aligned, _ = ops.align_set(x_refs, None, grid, align_corners=False)
        v_lin, _ = ops.align_set(1 - m_refs, None, grid, align_corners=False)
        v_aligned = (v_lin > 0.5).float()
        v_maps = (v_aligned - (1 - m_target.unsqueeze(2))).clamp(0, 1)
        return x_aligned, v_aligned, v_maps
