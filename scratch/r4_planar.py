# round 4 lab: does the dense-block kernel care whether a 32-channel chunk of the block buffer is a 128-byte piece every
# 768 bytes (NHWC, 192 channels) or a plane of its own?  Build with -DMT_WINOX_PLANAR=1 for mode "planar".
import sys
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E, _lib as L
dev = 'cuda:0'
mode = sys.argv[1]
E.PRECISION = 'f16x3'; E.USE_HALO = True
for (n, h, w, cin, cout) in [(16, 64, 64, 160, 32), (32, 64, 64, 160, 32), (32, 64, 64, 96, 32), (32, 128, 128, 160, 32)]:
    torch.manual_seed(0)
    spec = E.LayerSpec('c', cin, cout, 3, 1, 1)
    xl = torch.randn((n, h, w, 192), device=dev)
    wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.05; b = torch.zeros((cout,), device=dev)
    amax = xl.abs().max().reshape(1).clone()
    if mode == 'planar':
        x = xl.view(n, h, w, 6, 32).permute(3, 0, 1, 2, 4).contiguous()
        v = E.view_of(x[0]); v.C = cin
    else:
        x = xl
        v = E.view_of(x[..., :cin])
    out = torch.empty((n, h, w, 32), device=dev)
    run = lambda: E.conv_forward(spec, v, (n, h, w), wt, b, out=out, x_amax=amax)
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(f'{mode}: N{n} {h}x{w} {cin}->{cout}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us   checksum {out.double().sum().item():.6f} {out.abs().max().item():.6f}', flush=True)
