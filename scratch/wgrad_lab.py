"""Lab: weight-gradient time of representative layers (env switches set by the caller)."""
import sys, os
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E
dev = 'cuda:0'
def bench(n, h, w, cin, cout, k, reps=10):
    spec = E.LayerSpec('c', cin, cout, k, 1, 1)
    x = torch.randn((n, h, w, E.cpad(cin)), device=dev); wt = torch.randn((cout, cin, k, k), device=dev) * 0.05
    dz = torch.randn((n, h, w, E.cpad(cout)), device=dev)
    run = lambda: E.conv_backward_weight(spec, dz, E.view_of(x), wt)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{os.environ.get("TAG","")} wgrad N{n} {h}x{w} {cin}->{cout} k{k}: {ms:.3f} ms {2.0*n*h*w*cin*cout*k*k/ms/1e9:.0f} TF', flush=True)
for sh in [(32, 256, 256, 128, 128, 3), (8192, 16, 16, 256, 256, 3), (32, 128, 128, 128, 128, 3), (32, 256, 256, 128, 128, 5),
           (32, 64, 64, 160, 128, 3), (32, 64, 64, 192, 64, 3)]:
    bench(*sh)
