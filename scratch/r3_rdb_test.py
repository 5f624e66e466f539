"""Fused dense-block forward (mt_rdb_fwd) against the layer-by-layer launches: RRDBNet forward, both paths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import torch, time
import seeded as S
import master_thesis_amd as mt
from master_thesis_amd import engine as E
from master_thesis_amd.model_chn import RRDBNet
DEV = 'cuda:0'
torch.manual_seed(0)
net = RRDBNet(9, 3, nb=20).to(DEV)
for n, hw in ((4, 64), (32, 256)):
    x = torch.randn((n, 9, hw, hw), device=DEV)
    outs = {}
    for fused in (0, 1, 0, 1):
        E.RDB_FUSED = bool(fused)
        with torch.no_grad():
            y = net(x); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                y = net(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
        outs[fused] = y
        print(f'n={n} {hw}x{hw} fused={fused}: {dt * 1e3:.2f} ms', flush=True)
    d = (outs[0] - outs[1]).abs().max() / outs[0].abs().max()
    print('   max-normalised difference fused vs layer-by-layer: %.2e' % float(d), flush=True)
