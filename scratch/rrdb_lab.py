"""Lab: RRDB-shaped layers timed from a captured hipGraph (no host launch overhead); inputs are channel
windows of a 192-channel buffer as in RRDBNetFn."""
import sys, os
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E
dev = 'cuda:0'
def bench(n, h, w, cin, cout, reps=20):
    spec = E.LayerSpec('c', cin, cout, 3, 1, 2)
    buf = torch.randn((n, h, w, 192), device=dev)
    x = buf[..., :cin]
    out = torch.empty((n, h, w, 192), device=dev)[..., 64:64 + cout] if cout == 32 else torch.empty((n, h, w, cout), device=dev)
    wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev)
    xa = E.absmax_into(x, E.new_amax(dev)); ya = E.new_amax(dev)
    run = lambda: E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b, out=out, x_amax=xa, y_amax=ya)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{os.environ.get("TAG","")} N{n} {h}x{w} {cin}->{cout}: {ms*1e3:.1f} us {2.0*n*h*w*cin*cout*9/ms/1e9:.0f} TF', flush=True)
for sh in [(32, 64, 64, 64, 32), (32, 64, 64, 96, 32), (32, 64, 64, 128, 32), (32, 64, 64, 160, 32), (32, 64, 64, 192, 64),
           (24, 128, 128, 64, 32), (24, 128, 128, 160, 32), (24, 128, 128, 192, 64)]:
    bench(*sh)
