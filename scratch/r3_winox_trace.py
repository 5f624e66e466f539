"""Lab: stage timeline of igemm_winox_kernel blocks (library built with -DMT_WINOX_TRACE=1)."""
import sys, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
import numpy as np, torch
from master_thesis_amd import engine as E, _lib as L
lib = ctypes.CDLL(L.LIB_PATH if hasattr(L, 'LIB_PATH') else 'master_thesis_amd/libmt355.so')
dev = 'cuda:0'
E.PRECISION = 'f16x3'; E.USE_HALO = True
def run(n, h, w, cin, cout):
    spec = E.LayerSpec('c', cin, cout, 3, 1, 1)
    x = torch.randn((n, h, w, E.cpad(cin)), device=dev); wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev)
    for _ in range(3): E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b)
    torch.cuda.synchronize()
    nb = n * (h // 8) * (w // 16)
    buf = (ctypes.c_ulonglong * (4096 * 16))()
    assert lib.mt_lab_winox_trace(buf, 4096 * 16) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16)[:min(nb, 4096), :11].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) * 0.01          # us (100 MHz)
    names = ['entry', 'loads issued', 'store_v done', 'barrier', 'taps(ch0)', 'barrier', 'store_v', 'barrier', 'loop end', 'epilogue', 'amax']
    print(f'== N{n} {h}x{w} {cin}->{cout}: {nb} blocks; kernel span {t.max():.1f} us')
    st = t[:, 0]
    print('  block start: p10 %.1f  p50 %.1f  p75 %.1f p90 %.1f  max %.1f' % tuple(np.percentile(st, [10, 50, 75, 90, 100])))
    print('  block life : p10 %.1f  p50 %.1f  p90 %.1f' % tuple(np.percentile(t[:, 10] - t[:, 0], [10, 50, 90])))
    first = st < np.percentile(st, 40)
    for nm, sel in (('first-round blocks', first), ('late blocks', ~first)):
        d = np.diff(t[sel], axis=1)
        print('  ' + nm + ': ' + ', '.join(f'{names[i+1]} {np.median(d[:, i]):.2f}' for i in range(10)))
for sh in [(32, 64, 64, 64, 32), (32, 64, 64, 160, 32), (16, 64, 64, 160, 32), (32, 64, 64, 192, 64)]:
    run(*sh)
