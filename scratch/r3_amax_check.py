"""Lab / debugging: the max|y| scalar every f16x3 launch raises against torch's max of the output."""
import sys
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E, _lib as L
dev = 'cuda:0'
E.PRECISION = 'f16x3'; E.USE_HALO = True
torch.manual_seed(0)
bad = 0
def check(tag, out, c):
    global bad
    a = float(out._mt_amax)
    t = float(out[..., :c].abs().max())
    ok = abs(a - t) <= 1e-6 * t
    if not ok: bad += 1
    print(('ok  ' if ok else 'BAD ') + f'{tag}: scalar {a:.6g} torch {t:.6g}', flush=True)
for rep in range(2):
  for (n, h, w, cin, cout, k, st) in [(4, 64, 64, 64, 32, 3, 1), (4, 64, 64, 160, 32, 3, 1), (4, 64, 64, 192, 64, 3, 1), (2, 128, 128, 128, 128, 3, 1),
                                    (2, 128, 128, 128, 128, 5, 1), (2, 64, 64, 128, 256, 3, 1), (64, 16, 16, 256, 256, 3, 1), (2, 128, 128, 10, 128, 5, 1),
                                    (2, 128, 128, 128, 2, 5, 1), (2, 128, 128, 128, 1, 3, 1), (2, 128, 128, 3, 64, 3, 1), (2, 128, 128, 128, 128, 5, 2),
                                    (2, 128, 128, 64, 64, 3, 2), (2, 64, 64, 64, 64, 3, 1), (1, 40, 56, 128, 128, 3, 1), (3, 24, 40, 96, 32, 3, 1), (2, 64, 64, 32, 32, 3, 1)]:
    spec = E.LayerSpec('c', cin, cout, k, st, k // 2)
    x = torch.randn((n, h, w, E.cpad(cin)), device=dev) * (1 + rep)
    if E.cpad(cin) != cin: x[..., cin:] = 0
    wt = torch.randn((cout, cin, k, k), device=dev) * 0.05; b = torch.randn((cout,), device=dev)
    y = E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b)
    y = y[0] if isinstance(y, tuple) else y
    check(f'fwd N{n} {h}x{w} {cin}->{cout} k{k} s{st}', y, cout)
    dz = torch.randn_like(y) * 3
    if E.cpad(cout) != cout: dz[..., cout:] = 0
    dx = torch.empty_like(x)
    E.conv_backward_data(spec, dz, wt, E.view_of(dx))
    if hasattr(dx, '_mt_amax'): check(f'bwd N{n} {h}x{w} {cout}->{cin} k{k} s{st}', dx, cin)
    else: print('     (bwd: no scalar on dx)')
print('BAD:', bad)
