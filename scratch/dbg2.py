import sys, ctypes as C, collections
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden'); sys.path.insert(0, 'tests')
import torch, torch.nn.functional as F
import seeded as S
from master_thesis_amd import engine as E, _lib as L
from test_gpu_kernels import CONV_CASES, _torch_layer
dev = 'cuda:0'
for case in CONV_CASES[:4]:
    n, h, w, cin, cout, k, stride, kind, act = case
    tag = 'conv.' + '.'.join(map(str, case))
    x = S.uniform(tag + '.x', (n, cin, h, w), -1, 1)
    wshape = (cout, cin, k, k) if kind == 'c' else (cin, cout, k, k)
    wt = S.uniform(tag + '.w', wshape, -0.2, 0.2)
    bs = S.uniform(tag + '.b', (cout,), -0.5, 0.5)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    yr = _torch_layer(xr, wr, br, k, stride, kind, act)
    gy = S.uniform(tag + '.gy', tuple(yr.shape), -1, 1)
    yr.backward(gy)
    spec = E.LayerSpec(kind, cin, cout, k, stride, act)
    xg = x.to(dev).requires_grad_(True)
    wg, bg = wt.to(dev).requires_grad_(True), bs.to(dev).requires_grad_(True)
    x4 = E.ToNHWC4Fn.apply(xg)
    y4 = E.conv_stack(x4, [spec], [wg, bg])
    y = E.FromNHWC4Fn.apply(y4, cout)
    y.backward(gy.to(dev))
    err = (xg.grad.cpu() - xr.grad).abs()
    print(case, 'fwd err', float((y.detach().cpu()-yr.detach()).abs().max()), 'dx err', float(err.max()), 'ref', float(xr.grad.abs().max()))
    bad = (err > 1e-3).nonzero()
    print(' nbad', len(bad), bad[:5].tolist())
    if len(bad):
        print(' by n', collections.Counter(bad[:,0].tolist()), 'by y', sorted(collections.Counter(bad[:,2].tolist()).items()))
        # mask disagreement
        m_ref = (yr.detach() > 0); m_gpu = (y.detach().cpu() > 0)
        dis = (m_ref != m_gpu).nonzero()
        print(' mask disagreements', len(dis), dis[:5].tolist(), [float(yr.detach()[tuple(d)]) for d in dis[:5]])
