"""Lab: which blocks of igemm_winox_kernel are slow, and in which stage (library built with -DMT_WINOX_TRACE=1)."""
import sys, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
import numpy as np, torch
from master_thesis_amd import engine as E, _lib as L
lib = ctypes.CDLL(L.LIB_PATH)
dev = 'cuda:0'
E.PRECISION = 'f16x3'; E.USE_HALO = True
names = ['entry', 'loads issued', 'store_v', 'barrier', 'taps0', 'barrier', 'store_v', 'barrier', 'loop end', 'epilogue', 'amax']
def run(n, h, w, cin, cout):
    spec = E.LayerSpec('c', cin, cout, 3, 1, 1)
    x = torch.randn((n, h, w, E.cpad(cin)), device=dev); wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev)
    for _ in range(3): E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b)
    torch.cuda.synchronize()
    nb = n * (h // 8) * (w // 16)
    buf = (ctypes.c_ulonglong * (4096 * 16))()
    assert lib.mt_lab_winox_trace(buf, 4096 * 16) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16)[:min(nb, 4096), :15].astype(np.int64)
    t = (t - t[:, 0].min()) * 0.01
    life = t[:, 10] - t[:, 0]
    print(f'== N{n} {h}x{w} {cin}->{cout}: {nb} blocks; span {t.max():.1f} us; life percentiles 10/50/75/90/99: ' + ' '.join('%.1f' % v for v in np.percentile(life, [10, 50, 75, 90, 99])))
    xcd = np.arange(len(t)) % 8
    print('  life by XCD (median): ' + ' '.join('%.1f' % np.median(life[xcd == k]) for k in range(8)))
    print('  end by XCD (max):     ' + ' '.join('%.1f' % t[xcd == k, 10].max() for k in range(8)))
    ep = t[:, [8, 11, 12, 13, 14, 9]]
    t = t[:, :11]
    d = np.diff(t, axis=1)
    de = np.diff(ep, axis=1)
    enames = ['barrier', 'LDS post+barrier', 'read+stores(a=0)', 'post+barriers(a=1)', 'read+stores(a=1)']
    order = np.argsort(life)
    for nm, sel in (('fastest 25%', order[:len(order) // 4]), ('slowest 25%', order[-len(order) // 4:])):
        print('  ' + nm + f' (start {np.median(t[sel, 0]):.1f}): ' + ', '.join(f'{names[i+1]} {np.median(d[sel, i]):.2f}' for i in range(10)))
        print('      epilogue: ' + ', '.join(f'{enames[i]} {np.median(de[sel, i]):.2f}' for i in range(5)))
    slow = order[-len(order) // 4:]
    print('  slowest blocks: local index within XCD (bid >> 3) min/median/max %d %d %d' % (np.min(slow >> 3), np.median(slow >> 3), np.max(slow >> 3)))
    hist, _ = np.histogram(slow >> 3, bins=8, range=(0, len(t) // 8))
    print('  slow blocks by eighth of the XCD queue: ' + ' '.join(str(v) for v in hist))
for sh in [(16, 64, 64, 160, 32), (32, 64, 64, 160, 32), (32, 64, 64, 64, 32)]:
    run(*sh)
