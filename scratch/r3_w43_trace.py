"""Lab: stage timeline of igemm_wino43_kernel blocks (conv_halo_wino43.hip built with -DMT_W43_TRACE=1)."""
import sys, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
import numpy as np, torch
from master_thesis_amd import engine as E, _lib as L
lib = ctypes.CDLL(L.LIB_PATH)
dev = 'cuda:0'
E.PRECISION = 'f16x3'; E.USE_HALO = True
def run(n, h, w, cin, cout):
    spec = E.LayerSpec('c', cin, cout, 3, 1, 1)
    x = torch.randn((n, h, w, E.cpad(cin)), device=dev); wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev)
    for _ in range(3): E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b)
    e1.record(); torch.cuda.synchronize()
    nb = min(8192, n * ((h + 15) // 16) * ((w + 15) // 16) * ((cout + 127) // 128))
    buf = (ctypes.c_ulonglong * (8192 * 8))()
    assert lib.mt_lab_w43_trace(buf, 8192 * 8) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8)[:nb, :6].astype(np.int64)
    t = (t - t[:, 0].min()) * 0.01
    d = np.diff(t, axis=1)
    life = t[:, 5] - t[:, 0]
    nm = ['prologue', 'main loop', 'exchange', 'finish (stores)', 'amax']
    print(f'== N{n} {h}x{w} {cin}->{cout}: {nb} blocks traced, launch {e0.elapsed_time(e1) / 5 * 1e3:.0f} us; span {t.max():.0f} us; life p10/50/90 ' + ' '.join('%.1f' % v for v in np.percentile(life, [10, 50, 90])))
    print('   medians: ' + ', '.join(f'{nm[i]} {np.median(d[:, i]):.2f}' for i in range(5)) + ';  p90: ' + ', '.join(f'{nm[i]} {np.percentile(d[:, i], 90):.2f}' for i in range(5)))
for sh in [(32, 256, 256, 128, 128), (8192, 16, 16, 256, 256), (32, 64, 64, 128, 128), (8, 256, 256, 64, 128)]:
    run(*sh)
