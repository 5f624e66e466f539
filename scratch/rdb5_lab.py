"""Lab: the 192 -> 64 3x3 layer of a dense block (conv5 / block-input gradient shape), plain and with the fused residual."""
import sys, os
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E
dev = 'cuda:0'
def bench(n, h, w, cin, cout, res, reps=20):
    spec = E.LayerSpec('c', cin, cout, 3, 1, 0)
    x = torch.randn((n, h, w, cin), device=dev); wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev)
    r = torch.randn((n, h, w, cout), device=dev) if res else None
    out = torch.empty((n, h, w, cout), device=dev)
    run = lambda: E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b, out=out, res=r, alpha=0.2, beta=1.0 if res else 0.0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{os.environ.get("TAG","")} N{n} {h}x{w} {cin}->{cout} res={res}: {ms*1e3:.1f} us {2.0*n*h*w*cin*cout*9/ms/1e9:.0f} TF', flush=True)
for sh in [(32, 64, 64, 192, 64, False), (32, 64, 64, 192, 64, True), (32, 64, 64, 128, 128, False), (32, 128, 128, 192, 64, False), (128, 64, 64, 192, 64, False), (32, 64, 64, 192, 128, False)]:
    bench(*sh)
