"""Hand-built test inputs shared by make_golden.py and the tests."""
import torch

import seeded as S


def warp_grid(b, f, h, w, name):
    """Affine grids plus hand-placed hard cases: exact .5 ties (in pixel
    space), the +-1 borders, and out-of-range samples."""
    g = S.affine_flow(b, f, h, w, 0, name).clone()
    # pixel coordinate k + 0.5 in normalised units (align_corners=True)
    for k in range(min(h, w) - 1):
        g[0, 0, 0, k, 0] = (k + 0.5) / (w - 1) * 2 - 1
        g[0, 0, 0, k, 1] = (k + 0.5) / (h - 1) * 2 - 1
    g[0, 0, 1, :, 0] = torch.linspace(-1, 1, w)
    g[0, 0, 1, :, 1] = -1.0
    g[0, 0, 2, :, 0] = 1.0
    g[0, 0, 2, :, 1] = torch.linspace(-1, 1, w)
    g[0, 0, 3, :, 0] = torch.linspace(-1.5, 1.5, w)
    g[0, 0, 3, :, 1] = torch.linspace(1.5, -1.5, w)
    g[0, 0, 4, :, :] = 1.0 + 1.0 / (w - 1)      # exactly half a pixel outside
    return g
