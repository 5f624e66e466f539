import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden'); sys.path.insert(0, 'tests')
import torch, torch.nn.functional as F
import seeded as S
from master_thesis_amd import engine as E, _lib as L
dev = 'cuda:0'
def l2(a, b): return float((a.detach().double().cpu() - b.double()).norm() / b.double().norm())
# single-layer as_fwd bwd-data vs fp64 for several shapes, halo on/off
for (n,h,w,cin,cout,k) in [(2,24,24,12,64,5), (2,24,24,64,64,3), (2,16,16,160,32,3), (2,16,16,192,64,3), (1,40,24,64,64,3), (2,24,24,16,64,5)]:
    x = S.uniform('x', (n,cin,h,w), -1, 1); wt = S.uniform('w', (cout,cin,k,k), -0.1, 0.1); bs = S.uniform('b', (cout,), -0.1, 0.1)
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, bs.double(), padding=k//2)
    gy = S.uniform('gy', tuple(yr.shape), -1, 1)
    yr.backward(gy.double())
    for prec in ('bf16x3', 'bf16x6'):
        for halo in (False, True):
            E.PRECISION = prec; E.USE_HALO = halo
            spec = E.LayerSpec('c', cin, cout, k, 1, 0)
            xg = x.to(dev).requires_grad_(True); wg = wt.to(dev).requires_grad_(True); bg = bs.to(dev).requires_grad_(True)
            y = E.FromNHWC4Fn.apply(E.conv_stack(E.ToNHWC4Fn.apply(xg), [spec], [wg, bg]), cout)
            y.backward(gy.to(dev))
            print((n,h,w,cin,cout,k), prec, 'halo' if halo else 'igemm', 'fwd %.2e dx %.2e dw %.2e' % (l2(y, yr.detach()), l2(xg.grad, xr.grad), l2(wg.grad, wr.grad)))
