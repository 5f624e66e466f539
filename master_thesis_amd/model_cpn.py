"""The warp tail of the Copy-and-Paste aligner (``master_thesis/model_cpn.py:75-91``).

The warp convention of ``CPN.align`` is on the hot path: an
affine sampling grid (``affine_grid``, align_corners=False) followed by a
bilinear ``grid_sample`` (align_corners=False) of the frames and of the
visibility maps (thresholded at 0.5).  The two ALIGNER networks that produce
``theta`` (A_Encoder, A_Regressor, model_cpn.py:121-174) run on
the conv engine, forward only (the aligner is frozen); module and parameter
names are the reference's, so its checkpoints load (the keys of CPN's own
inpainting network -- Encoder, CM_Module, Decoder -- are ignored).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import engine as E
from . import ops


class _Conv2d(nn.Module):
    """model_cpn.py:299-314: nn.Sequential(Conv2d[, activation]) under ``.conv``."""

    def __init__(self, in_ch, out_ch, k, stride, pad, relu=True):
        super().__init__()
        layers = [nn.Conv2d(in_ch, out_ch, k, stride, pad)]
        if relu:
            layers.append(nn.ReLU())
        self.conv = nn.Sequential(*layers)
        self.spec = E.LayerSpec('c', in_ch, out_ch, k, stride, L.ACT_RELU if relu else L.ACT_NONE)

    def params(self):
        return [self.conv[0].weight, self.conv[0].bias]


class A_Encoder(nn.Module):
    """model_cpn.py:121-151."""

    def __init__(self):
        super().__init__()
        self.conv12 = _Conv2d(4, 64, 5, 2, 2)
        self.conv2 = _Conv2d(64, 64, 3, 1, 1)
        self.conv23 = _Conv2d(64, 128, 3, 2, 1)
        self.conv3 = _Conv2d(128, 128, 3, 1, 1)
        self.conv34 = _Conv2d(128, 256, 3, 2, 1)
        self.conv4a = _Conv2d(256, 256, 3, 1, 1)
        self.conv4b = _Conv2d(256, 256, 3, 1, 1)
        self.register_buffer('mean', torch.FloatTensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.FloatTensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def forward4(self, in_f, in_v):
        """in_f (N,3,h,w), in_v (N,1,h,w) -> NHWC features (N,28,28,256).  The normalisation is
        applied after the bilinear resize to 224 x 224 (it commutes with it) while packing."""
        n = in_f.shape[0]
        fr = ops.resize_planar(in_f.contiguous(), (224, 224), 'bilinear')
        vr = ops.resize_planar(in_v.contiguous(), (224, 224), 'bilinear')
        std, mean = self.std.reshape(3), self.mean.reshape(3)
        x4 = torch.zeros((n, 224, 224, E.cpad(4)), device=in_f.device, dtype=torch.float32)
        E.pack_images(x4, n, 1, [(fr, 0, (1.0 / std).contiguous(), (-mean / std).contiguous()), (vr, 3, None, None)])
        mods = (self.conv12, self.conv2, self.conv23, self.conv3, self.conv34, self.conv4a, self.conv4b)
        return E.conv_stack(x4, [m.spec for m in mods], [p for m in mods for p in m.params()])

    def forward(self, in_f, in_v):
        return E.from_nhwc4(self.forward4(in_f, in_v), 256)


class A_Regressor(nn.Module):
    """model_cpn.py:154-174 (conv5a / conv5b are applied twice, conv6a / conv6b never: as upstream)."""

    def __init__(self):
        super().__init__()
        self.conv45 = _Conv2d(512, 512, 3, 2, 1)
        self.conv5a = _Conv2d(512, 512, 3, 1, 1)
        self.conv5b = _Conv2d(512, 512, 3, 1, 1)
        self.conv56 = _Conv2d(512, 512, 3, 2, 1)
        self.conv6a = _Conv2d(512, 512, 3, 1, 1)
        self.conv6b = _Conv2d(512, 512, 3, 1, 1)
        self.fc = nn.Linear(512, 6)

    def forward4(self, x4):
        """x4: NHWC (N,28,28,512) = [target features | reference features] -> theta (N,2,3)."""
        mods = (self.conv45, self.conv5a, self.conv5b, self.conv56, self.conv5a, self.conv5b)
        y = E.conv_stack(x4, [m.spec for m in mods], [p for m in mods for p in m.params()])
        # global average pool + 512 -> 6 linear head: 6 K multiply-adds per pair, plain tensor ops
        return F.linear(y.mean(dim=(1, 2)), self.fc.weight, self.fc.bias).view(-1, 2, 3)

    def forward(self, feat1, feat2):
        x4 = torch.cat([E.to_nhwc4(feat1), E.to_nhwc4(feat2)], dim=3)
        return self.forward4(x4)


class CPN(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.A_Encoder = A_Encoder()
        self.A_Regressor = A_Regressor()

    @torch.no_grad()
    def align(self, x_target, m_target, x_refs, m_refs):
        """model_cpn.py:31-91: the aligner protocol of CHN (x_aligned, v_aligned, v_maps)."""
        b, c, ref_n, h, w = x_refs.size()
        # one encoder pass over the target and all reference frames
        frames = torch.cat([x_target, x_refs.transpose(1, 2).reshape(-1, c, h, w)], dim=0)
        masks = torch.cat([m_target, m_refs.transpose(1, 2).reshape(-1, 1, h, w)], dim=0)
        feats = self.A_Encoder.forward4(frames, masks)                     # (b + b*r, 28, 28, 256)
        ft, fr = feats[:b], feats[b:]
        pair = torch.empty((b * ref_n,) + tuple(ft.shape[1:3]) + (512,), device=ft.device, dtype=torch.float32)
        pair.view(b, ref_n, *pair.shape[1:])[..., :256] = ft.unsqueeze(1)
        pair[..., 256:] = fr
        theta_rt = self.A_Regressor.forward4(pair)
        return CPN.warp(x_refs, m_refs, m_target, theta_rt)

    def load_state_dict(self, state_dict, strict=False):
        """Accepts the reference's checkpoints: keys of CPN's own inpainting network (Encoder,
        CM_Module, Decoder) and a leading 'module.' are ignored (model_cpn.py:93-118)."""
        own = self.state_dict()
        sd = {k.replace('module.', ''): v for k, v in state_dict.items()}
        return super().load_state_dict({k: v for k, v in sd.items() if k in own}, strict=strict)

    @staticmethod
    def init_model_with_state(checkpoint_path, device='cpu'):
        """model_cpn.py:93-118."""
        model = CPN()
        model.load_state_dict(dict(torch.load(checkpoint_path, map_location=device)))
        for param in model.parameters():
            param.requires_grad = False
        return model.to(device)

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
