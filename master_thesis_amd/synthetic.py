"""Synthetic clips in the reference's batch layout (dataset.py:180-181 after
collation), generated on the device: y ~ U[0,1); one rectangle
mask per frame covering 5-15 % of the pixels, position jittered per frame, grown
by four 3x3 cross dilations (data.py:337-338,349-351); x = (1-m)*y + m*fill
(dataset.py:168-169); ground-truth movement = small random affine grids
(align_corners=True) on the even samples (flows_use), zeros elsewhere."""
import math

import torch
import torch.nn.functional as F


def make_batch(b, f, h, w, device, seed=1234):
    gen = torch.Generator(device='cpu').manual_seed(seed)
    y = torch.rand((b, 3, f, h, w), generator=gen)
    m = torch.zeros((b, 1, f, h, w))
    for i in range(b):
        area = 0.05 + 0.10 * float(torch.rand((), generator=gen))
        aspect = 0.5 + float(torch.rand((), generator=gen))
        mh = max(2, min(h - 10, int(round(math.sqrt(area * h * w * aspect)))))
        mw = max(2, min(w - 10, int(round(area * h * w / mh))))
        y0 = int(torch.randint(4, max(5, h - mh - 4), (), generator=gen))
        x0 = int(torch.randint(4, max(5, w - mw - 4), (), generator=gen))
        for j in range(f):
            dy, dx = (int(v) for v in torch.randint(-3, 4, (2,), generator=gen))
            ya, xa = min(max(0, y0 + dy), h - mh), min(max(0, x0 + dx), w - mw)
            m[i, 0, j, ya:ya + mh, xa:xa + mw] = 1
    if torch.device(device).type == 'cuda':
        # the dataset's mask dilation + masked fill (dataset.py:158-169) as ONE HIP launch on the device
        from .utils import TransformsUtils
        y, m = y.to(device), m.to(device)
        m, x = TransformsUtils.dilate_fill(m, y, (3, 3), 4)
    else:
        # (CPU tensors for the host-side tests: four cross-shaped 3x3 dilations with torch ops)
        mm = m.reshape(b * f, 1, h, w)
        for _ in range(4):
            p = F.pad(mm, (1, 1, 1, 1))
            mm = torch.stack([p[:, :, 1:-1, 1:-1], p[:, :, :-2, 1:-1], p[:, :, 2:, 1:-1], p[:, :, 1:-1, :-2],
                              p[:, :, 1:-1, 2:]]).amax(0)
        m = mm.reshape(b, 1, f, h, w)
        fill = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1)
        x = (1 - m) * y + m * fill
    flows_use = torch.tensor([i % 2 == 0 for i in range(b)])
    p = torch.rand((b * f, 4), generator=gen) * 2 - 1
    ang, sc = p[:, 0] * 0.2, 1 + p[:, 1] * 0.1
    theta = torch.zeros((b * f, 2, 3))
    theta[:, 0, 0], theta[:, 0, 1] = sc * torch.cos(ang), -sc * torch.sin(ang)
    theta[:, 1, 0], theta[:, 1, 1] = sc * torch.sin(ang), sc * torch.cos(ang)
    theta[:, :, 2] = p[:, 2:4] * (50.0 / (w / 2))
    gt = F.affine_grid(theta, [b * f, 1, h, w], align_corners=True).reshape(b, f, h, w, 2)
    gt = gt * flows_use.view(b, 1, 1, 1, 1).float()
    info = (['synthetic'] * b, ['0'] * b, flows_use.to(device), flows_use.clone().to(device), gt.to(device),
            gt.clone().to(device))
    return (x.to(device), m.to(device)), y.to(device), info
