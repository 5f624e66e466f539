import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch, torch.nn.functional as F
import seeded as S
from master_thesis_amd import engine as E, _lib as L
dev = 'cuda:0'
def rel(a, b): return float((a.double().cpu()-b.double()).norm()/b.double().norm())
for (n,h,w,cin,cout,k) in [(2,32,32,128,128,3), (2,32,32,64,32,3), (2,32,32,12,64,3), (1,64,64,128,128,5)]:
    x = S.uniform('x', (n,cin,h,w), -1, 1); wt = S.uniform('w', (cout,cin,k,k), -0.1, 0.1); bs = S.uniform('b', (cout,), -0.1, 0.1)
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, bs.double(), padding=k//2)
    gy = S.uniform('gy', tuple(yr.shape), -1, 1)
    yr.backward(gy.double())
    for prec in ('f32', 'bf16x3', 'bf16x6', 'f16x3'):
        E.PRECISION = prec
        spec = E.LayerSpec('c', cin, cout, k, 1, 0)
        xg = x.to(dev).requires_grad_(True); wg = wt.to(dev).requires_grad_(True); bg = bs.to(dev).requires_grad_(True)
        y = E.FromNHWC4Fn.apply(E.conv_stack(E.ToNHWC4Fn.apply(xg), [spec], [wg, bg]), cout)
        y.backward(gy.to(dev))
        print((n,h,w,cin,cout,k), prec, 'fwd %.2e dx %.2e dw %.2e' % (rel(y.detach(), yr.detach()), rel(xg.grad, xr.grad), rel(wg.grad, wr.grad)))
# CPU fp32 reference error for comparison
    x32 = x.clone().requires_grad_(True); w32 = wt.clone().requires_grad_(True)
    y32 = F.conv2d(x32, w32, bs, padding=k//2); y32.backward(gy)
    print('   cpu32 fwd %.2e dx %.2e dw %.2e' % (rel(y32.detach(), yr.detach()), rel(x32.grad, xr.grad), rel(w32.grad, wr.grad)))
