"""Import the reference package (read-only, /root/reference) in THIS container.

The reference's hot path (model_dfpn.py / model_chn.py / utils.py / model_vgg.py)
only needs torch for arithmetic.  The packages below are imported at module
scope by the reference but carry no hot-path arithmetic and are not installed
here; they are registered as inert modules so that
``import master_thesis`` succeeds.  Nothing from the reference is copied: this
file only arranges ``sys.modules`` and ``sys.path``.

Used ONLY by tests/golden/make_golden.py (fixture generation, this container).
Never imported on the GPU box: /root/reference does not exist there.
"""
import sys
import types

import torch.nn as nn

REFERENCE_ROOT = '/root/reference'
VGG16_CFG_D = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M',
               512, 512, 512, 'M', 512, 512, 512, 'M']


def _vgg_make_layers(cfg, batch_norm=False):
    # torchvision==0.11.1 vgg.make_layers semantics for batch_norm=False:
    # Conv2d(3x3, pad 1) + ReLU(inplace) per int entry, MaxPool2d(2, 2) per 'M'.
    layers, in_c = [], 3
    for v in cfg:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(in_c, v, kernel_size=3, padding=1),
                       nn.ReLU(inplace=True)]
            in_c = v
    return nn.Sequential(*layers)


class _VGG(nn.Module):
    def __init__(self, features, *a, **k):
        super().__init__()
        self.features = features


class _LightningModule(nn.Module):
    def log(self, *a, **k):
        pass


def import_reference():
    sys.dont_write_bytecode = True
    if 'master_thesis' in sys.modules:
        return sys.modules['master_thesis']
    pl = types.ModuleType('pytorch_lightning')
    pl.LightningModule = _LightningModule
    pl.LightningDataModule = object
    sys.modules['pytorch_lightning'] = pl
    for name in ('cv2', 'jpeg4py', 'lpips', 'skimage', 'skimage.metrics',
                 'skimage.transform'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['skimage'].metrics = sys.modules['skimage.metrics']
    sys.modules['skimage'].transform = sys.modules['skimage.transform']
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvv = types.ModuleType('torchvision.models.vgg')
    tvm.VGG = _VGG
    tvv.cfgs = {'D': VGG16_CFG_D}
    tvv.make_layers = _vgg_make_layers
    tvv.model_urls = {}
    tvm.vgg = tvv
    tv.models = tvm
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tvm
    sys.modules['torchvision.models.vgg'] = tvv
    sys.path.insert(0, REFERENCE_ROOT)
    import master_thesis as mt
    return mt


def make_vgg(mt):
    return mt.VGGFeatures(_vgg_make_layers(VGG16_CFG_D), 'cpu')
