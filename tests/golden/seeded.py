"""Seeded weights and inputs shared by the fixture generator and the tests.

Everything is drawn from ``numpy.random.Generator(PCG64)`` whose stream is
bit-stable across machines, keyed by the parameter NAME, so the reference model
(make_golden.py, this container) and the oracle / HIP product path (tests, GPU
box) get identical weights without storing them.
"""
import zlib

import numpy as np
import torch

VGG16_CFG_D = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M',
               512, 512, 512, 'M', 512, 512, 512, 'M')


def _rng(name, seed):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def uniform(name, shape, lo=0.0, hi=1.0, seed=0):
    a = _rng(name, seed).random(size=tuple(shape), dtype=np.float32)
    return torch.from_numpy(a * np.float32(hi - lo) + np.float32(lo))


def vgg_shapes(prefix):
    shapes, idx, cin = {}, 0, 3
    for v in VGG16_CFG_D:
        if v == 'M':
            idx += 1
        else:
            shapes[f'{prefix}features.{idx}.weight'] = (v, cin, 3, 3)
            shapes[f'{prefix}features.{idx}.bias'] = (v,)
            cin = v
            idx += 2
    return shapes


def _conv(shapes, key, cout, cin, k, transposed=False):
    shapes[key + '.weight'] = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    shapes[key + '.bias'] = (cout,)


def dfpn_shapes(prefix='', with_vgg=True):
    s = {}
    c4 = prefix + 'corr.conv.'
    for key, co, ci in (('conv_1.0', 128, 1), ('conv_1.2', 256, 128),
                        ('conv_1.4', 256, 256), ('conv_2.0', 256, 256),
                        ('conv_2.2', 128, 256), ('conv_2.4', 1, 128)):
        _conv(s, c4 + key, co, ci, 3)
    mix = ((0, 256, 256, 5), (2, 256, 256, 3), (4, 16, 256, 3), (6, 16, 16, 5),
           (8, 16, 16, 3), (10, 8, 16, 3), (12, 8, 8, 5), (14, 8, 8, 3),
           (16, 4, 8, 3), (18, 4, 4, 5), (20, 4, 4, 3), (22, 2, 4, 3),
           (23, 2, 2, 5), (24, 2, 2, 3))
    for idx, co, ci, k in mix:
        _conv(s, f'{prefix}corr_mixer.mixer.{idx}', co, ci, k)
    flow = ((0, 'c', 5), (2, 'c', 3), (4, 'c', 5), (6, 'c', 5), (8, 'c', 3),
            (10, 'c', 3), (12, 'c', 5), (14, 'c', 3), (16, 'c', 3),
            (18, 'c', 5), (20, 'c', 3), (22, 't', 3), (24, 'c', 5),
            (26, 'c', 3), (28, 't', 3), (30, 'c', 5), (32, 'c', 3),
            (34, 't', 5), (36, 'c', 5), (38, 'c', 3), (40, 'c', 3))
    for name in ('flow_64', 'flow_256'):
        for idx, kind, k in flow:
            cin = 10 if idx == 0 else 128
            cout = 2 if idx == 40 else 128
            _conv(s, f'{prefix}{name}.nn.{idx}', cout, cin, k, kind == 't')
    if with_vgg:
        s.update(vgg_shapes(prefix + 'model_vgg.'))
        s.update(vgg_shapes(prefix + 'corr.model_vgg.'))
    return s


def rrdbnet_shapes(prefix, nb=20, in_nc=9, out_nc=3, nf=64, gc=32):
    s = {}
    for i in range(5):
        _conv(s, f'{prefix}conv_first.{i}', nf, in_nc if i == 0 else nf, 3)
    for k in range(nb):
        for j in (1, 2, 3):
            p = f'{prefix}rrdb_trunk.{k}.rdb{j}.'
            for i in range(1, 5):
                _conv(s, f'{p}conv{i}', gc, nf + (i - 1) * gc, 3)
            _conv(s, p + 'conv5', nf, nf + 4 * gc, 3)
    for name in ('trunk_conv', 'upconv1', 'upconv2', 'hr_conv'):
        _conv(s, prefix + name, nf, nf, 3)
    _conv(s, prefix + 'conv_last', out_nc, nf, 3)
    return s


def chn_shapes(prefix='', nb=20):
    s = rrdbnet_shapes(prefix + 'nn.', nb)
    s.update(vgg_shapes(prefix + 'model_vgg.'))
    s.update(dfpn_shapes(prefix + 'model_aligner.'))
    return s


CPN_ENC = (('conv12', 4, 64, 5), ('conv2', 64, 64, 3), ('conv23', 64, 128, 3), ('conv3', 128, 128, 3),
           ('conv34', 128, 256, 3), ('conv4a', 256, 256, 3), ('conv4b', 256, 256, 3))
CPN_REG = (('conv45', 512, 512, 3), ('conv5a', 512, 512, 3), ('conv5b', 512, 512, 3), ('conv56', 512, 512, 3),
           ('conv6a', 512, 512, 3), ('conv6b', 512, 512, 3))


def cpn_state_dict(seed=0):
    """Seeded weights of the two ALIGNER networks of CPN (A_Encoder, A_Regressor, model_cpn.py:121-174).
    The regressor head is shrunk towards the identity transform so that the warp samples inside the frames."""
    sd = {}
    for mod, layers in (('A_Encoder', CPN_ENC), ('A_Regressor', CPN_REG)):
        for name, cin, cout, k in layers:
            bound = float(np.sqrt(6.0 / (cin * k * k)))
            sd[f'{mod}.{name}.conv.0.weight'] = uniform(f'cpn.{mod}.{name}.w', (cout, cin, k, k), -bound, bound, seed)
            sd[f'{mod}.{name}.conv.0.bias'] = uniform(f'cpn.{mod}.{name}.b', (cout,), -0.1, 0.1, seed)
    sd['A_Regressor.fc.weight'] = uniform('cpn.fc.w', (6, 512), -1, 1, seed) * 2e-3
    sd['A_Regressor.fc.bias'] = torch.tensor([1.0, 0.0, 0.0, 0.0, 1.0, 0.0]) + uniform('cpn.fc.b', (6,), -0.05, 0.05, seed)
    sd['A_Encoder.mean'] = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    sd['A_Encoder.std'] = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return sd


def _canonical(name):
    # The reference registers the SAME VGG module under several names
    # (model_vgg, corr.model_vgg, model_aligner.model_vgg, ...): key the draw on
    # the VGG-local name so every alias gets identical values.
    i = name.find('features.')
    return 'vgg.' + name[i:] if i >= 0 else name


def seeded_state_dict(shapes, seed=0, gain=1.0):
    """He-uniform weights (bound sqrt(6/fan_in)*gain), biases U(-0.1, 0.1)."""
    sd = {}
    for name, shape in shapes.items():
        key = _canonical(name)
        if name.endswith('.bias'):
            sd[name] = uniform(key, shape, -0.1, 0.1, seed)
        else:
            wname = name[:-len('.weight')]
            transposed = False
            # ConvTranspose2d weights are (cin, cout, k, k): fan_in = cin*k*k
            parts = wname.split('.')
            if parts[-2] == 'nn' and parts[-1] in ('22', '28', '34') \
                    and 'flow_' in wname:
                transposed = True
            fan_in = (shape[0] if transposed else shape[1]) * shape[2] * shape[3]
            bound = float(np.sqrt(6.0 / fan_in)) * gain
            sd[name] = uniform(key, shape, -bound, bound, seed)
    return sd


def add_buffers(sd, prefixes=('',)):
    for p in prefixes:
        sd[p + 'mean'] = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1)
        sd[p + 'std'] = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1, 1)
    return sd


def affine_flow(b, f, h, w, seed=0, name='flow'):
    """Absolute normalised sampling grids (B,F,H,W,2) from small random
    affines (align_corners=True convention, utils.py:27-31)."""
    import torch.nn.functional as F
    p = uniform(name, (b * f, 6), -1, 1, seed)
    theta = torch.zeros(b * f, 2, 3)
    ang = p[:, 0] * 0.2
    sc = 1 + p[:, 1] * 0.1
    theta[:, 0, 0] = sc * torch.cos(ang)
    theta[:, 0, 1] = -sc * torch.sin(ang)
    theta[:, 1, 0] = sc * torch.sin(ang)
    theta[:, 1, 1] = sc * torch.cos(ang)
    theta[:, :, 2] = p[:, 2:4] * 0.3
    g = F.affine_grid(theta, [b * f, 1, h, w], align_corners=True)
    return g.reshape(b, f, h, w, 2)


def synthetic_batch(b, f, h, w, seed=0, flows_use=None):
    """A batch tuple in the layout of dataset.py:180-181 after collation."""
    y = uniform('y', (b, 3, f, h, w), 0, 1, seed)
    m = torch.zeros(b, 1, f, h, w)
    pos = uniform('mpos', (b, f, 4), 0, 1, seed)
    for i in range(b):
        for j in range(f):
            mh = int(h * (0.2 + 0.2 * float(pos[i, j, 0])))
            mw = int(w * (0.2 + 0.2 * float(pos[i, j, 1])))
            y0 = int((h - mh) * float(pos[i, j, 2]))
            x0 = int((w - mw) * float(pos[i, j, 3]))
            m[i, 0, j, y0:y0 + mh, x0:x0 + mw] = 1
    fill = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1)
    x = (1 - m) * y + m * fill
    if flows_use is None:
        flows_use = [i % 2 == 0 for i in range(b)]
    flows_use = torch.tensor(flows_use, dtype=torch.bool)
    gt = affine_flow(b, f, h, w, seed, 'gt_movement')
    gt = gt * flows_use.view(b, 1, 1, 1, 1).float()
    m_mov = affine_flow(b, f, h, w, seed, 'm_movement')
    info = (['bg'] * b, ['0'] * b, flows_use, flows_use.clone(), gt, m_mov)
    return (x, m), y, info
