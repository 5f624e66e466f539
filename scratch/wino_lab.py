"""Lab: one 3x3 layer forward, direct register-B vs Winograd variants (MT355_WINO / MT355_WINO_ABL set by the caller)."""
import sys, os
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E
dev = 'cuda:0'
def bench(n, h, w, cin, cout, reps=10):
    spec = E.LayerSpec('c', cin, cout, 3, 1, 1)
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
    print(f'WINO={os.environ.get("MT355_WINO","1")} ABL={os.environ.get("MT355_WINO_ABL","0")} N{n} {h}x{w} {cin}->{cout}: {ms:.3f} ms {2.0*n*h*w*cin*cout*9/ms/1e9:.0f} TF', flush=True)
for sh in [(32, 256, 256, 128, 128), (8192, 16, 16, 256, 256), (32, 128, 128, 128, 128)]:
    bench(*sh)
