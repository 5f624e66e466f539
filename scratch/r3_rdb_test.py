"""Fused dense-block forward (mt_rdb_fwd) against the layer-by-layer launches: RRDBNet forward, both paths, a size scan."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import torch, time
import master_thesis_amd as mt
from master_thesis_amd import engine as E
from master_thesis_amd.model_chn import RRDBNet, RRDBNetFn
DEV = 'cuda:0'
torch.manual_seed(0)
net = RRDBNet(9, 3, nb=20).to(DEV)
for n, hw in ((1, 64), (1, 256), (2, 256), (4, 256), (8, 256), (16, 256), (1, 512), (4, 512), (32, 128)):
    x = torch.randn((n, 9, hw, hw), device=DEV)
    outs, ts = {}, {}
    for fused in (0, 1, 0, 1):
        RRDBNetFn.FUSED_MAX_TILES = 10 ** 9 if fused else 0
        with torch.no_grad():
            y = net(x); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                y = net(x)
            torch.cuda.synchronize()
            ts[fused] = (time.perf_counter() - t0) / 3
        outs[fused] = y
    d = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
    tiles = n * (hw // 4 // 8) * max(1, hw // 4 // 16)
    print(f'n={n} {hw}x{hw} (trunk tiles {tiles}): layer-by-layer {ts[0] * 1e3:.2f} ms, fused {ts[1] * 1e3:.2f} ms, diff {d:.1e}', flush=True)
