"""VGG-16 'D' feature extractor on the mt355 conv engine (forward only).

Mirrors ``master_thesis/model_vgg.py:11-65``: same constructor
``VGGFeatures(features, device)``, same ``forward(x, normalize_input=True)``
returning the five post-max-pool tensors (detached), same ``features.N.*``
state-dict keys.  (The reference derives from torchvision's VGG and therefore
also carries an unused ``classifier``; those keys are dropped on load.)
"""
import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E

VGG16_CFG_D = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M',
               512, 512, 512, 'M', 512, 512, 512, 'M')


def make_layers(cfg=VGG16_CFG_D):
    """torchvision.models.vgg.make_layers(cfg, batch_norm=False) layout."""
    layers, cin = [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


class VGGFeatures(nn.Module):
    # tensors of torchvision's VGG.classifier (3 Linear layers) that the reference's VGGFeatures carries unused
    # (model_vgg.py:11) and this one drops: optim.reference_parameter_list() keeps their places in the parameter order
    REFERENCE_CLASSIFIER_TENSORS = 6

    def __init__(self, features=None, device='cpu'):
        super().__init__()
        self.features = features if features is not None else make_layers()
        self.vgg_mean = torch.as_tensor([0.485, 0.456, 0.406]).view(3, 1, 1).to(device)
        self.vgg_std = torch.as_tensor([0.229, 0.224, 0.225]).view(3, 1, 1).to(device)

    def _affine(self, device):
        mean, std = self.vgg_mean.to(device).reshape(3), self.vgg_std.to(device).reshape(3)
        return (1.0 / std).contiguous(), (-mean / std).contiguous()

    @torch.no_grad()
    def forward4(self, x4, n_taps=5):
        """NHWC4 in (N,H,W,4: already normalised) -> list of NHWC4 taps.  Stops
        after tap ``n_taps`` (DFPN only needs pool4, model_dfpn.py:516-518)."""
        taps, cur = [], x4
        with E.precision(E.vgg_precision()):
            return self._forward4(cur, n_taps)

    def _forward4(self, cur, n_taps):
        taps = []
        for layer in self.features:
            if isinstance(layer, nn.Conv2d):
                spec = E.LayerSpec('c', layer.in_channels, layer.out_channels, 3, 1, L.ACT_RELU)
                cur = E.conv_forward(spec, E.view_of(cur), cur.shape[:3], layer.weight, layer.bias)
            elif isinstance(layer, nn.MaxPool2d):
                cur = E.maxpool2x2(cur)
                taps.append(cur)
                if len(taps) == n_taps:
                    break
        return taps

    @torch.no_grad()
    def forward(self, x, normalize_input=True):
        """model_vgg.py:26-44."""
        scale, shift = self._affine(x.device) if normalize_input else (None, None)
        x4 = E.to_nhwc4(x, scale, shift)
        return [E.from_nhwc4(t, t.shape[3]) for t in self.forward4(x4)]

    def load_state_dict(self, state_dict, strict=True):
        state_dict = {k: v for k, v in state_dict.items() if k.startswith('features.')}
        return super().load_state_dict(state_dict, strict)

    @staticmethod
    def get_pretrained_model(device):
        """model_vgg.py:46-65 downloads ImageNet weights; there is no network
        here, so a user-supplied torchvision ``vgg16`` state dict must be loaded
        with ``load_state_dict`` instead."""
        raise RuntimeError('no network: build VGGFeatures() and load a vgg16 state_dict')
