import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch, torch.nn.functional as F
import seeded as S
from master_thesis_amd import engine as E, _lib as L
dev = 'cuda:0'
n,h,w,cin,cout,k = 3,16,16,128,256,3
wt = S.uniform('w', (cout,cin,k,k), -0.2, 0.2)
gy = S.uniform('gy', (n,cout,h,w), -1, 1)
ref = torch.nn.grad.conv2d_input((n,cin,h,w), wt, gy, padding=1)
spec = E.LayerSpec('c', cin, cout, k, 1, 0)
dz = E.to_nhwc4(gy.to(dev))
wg = wt.to(dev)
for name in ('mt_conv_run', 'mt_conv_run_naive'):
    dx = torch.zeros((n,h,w,cin), device=dev)
    d = L.Conv()
    d.x, d.y = E.view_of(dz), E.view_of(dx)
    d.wt = E._pack(wg, 1, 'b').data_ptr()
    d.KH = d.KW = k; d.stride, d.pad, d.transposed = 1, 1, 1
    d.alpha = 1.0; d.res, d.dsrc = E.null_view(), E.null_view()
    L.call(name, C.byref(d), L.stream_ptr())
    got = dx.permute(0,3,1,2).cpu()
    err = (got - ref).abs()
    print(name, 'max err', float(err.max()), 'ref max', float(ref.abs().max()))
    bad = (err > 1e-3).nonzero()
    print(' n bad', len(bad), bad[:10].tolist())
    if len(bad):
        import collections
        print(' by n', collections.Counter(bad[:,0].tolist()), 'by y', sorted(collections.Counter(bad[:,2].tolist()).items()), 'by x', sorted(collections.Counter(bad[:,3].tolist()).items()))
        print(' by c', sorted(collections.Counter((bad[:,1]//32).tolist()).items()))
