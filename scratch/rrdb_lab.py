"""Lab: RRDB-shaped layers (env switches set by the caller)."""
import sys, os
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E
dev = 'cuda:0'
def bench(n, h, w, cin, cout, reps=20):
    spec = E.LayerSpec('c', cin, cout, 3, 1, 2)
    x = torch.randn((n, h, w, E.cpad(cin)), device=dev); wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev)
    run = lambda: E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{os.environ.get("TAG","")} N{n} {h}x{w} {cin}->{cout}: {ms*1e3:.1f} us {2.0*n*h*w*cin*cout*9/ms/1e9:.0f} TF', flush=True)
for sh in [(32, 64, 64, 64, 32), (32, 64, 64, 96, 32), (32, 64, 64, 128, 32), (32, 64, 64, 160, 32), (32, 64, 64, 192, 64),
           (24, 128, 128, 64, 32), (24, 128, 128, 160, 32), (24, 128, 128, 192, 64)]:
    bench(*sh)
