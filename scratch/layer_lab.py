"""Lab: forward time of representative layers (env MT355_WINO etc. set by the caller)."""
import sys, os
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E
dev = 'cuda:0'
def bench(n, h, w, cin, cout, k, reps=10):
    spec = E.LayerSpec('c', cin, cout, k, 1, 1)
    x = torch.randn((n, h, w, E.cpad(cin)), device=dev); wt = torch.randn((cout, cin, k, k), device=dev) * 0.05
    b = torch.zeros((cout,), device=dev)
    run = lambda: E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{os.environ.get("TAG","")} N{n} {h}x{w} {cin}->{cout} k{k}: {ms:.3f} ms {2.0*n*h*w*cin*cout*k*k/ms/1e9:.0f} TF', flush=True)
for sh in [(32, 256, 256, 128, 128, 3), (8192, 16, 16, 256, 256, 3), (32, 128, 128, 128, 128, 3), (32, 256, 256, 128, 128, 5),
           (32, 64, 64, 160, 32, 3), (32, 64, 64, 192, 64, 3), (32, 256, 256, 10, 128, 5), (32, 256, 256, 128, 2, 3), (40, 256, 256, 64, 64, 3)]:
    bench(*sh)
def bench_s2(n, h, w, c, k, kind, reps=10):
    spec = E.LayerSpec(kind, c, c, k, 2, 1)
    x = torch.randn((n, h, w, c), device=dev); wt = torch.randn((c, c, k, k), device=dev) * 0.05; b = torch.zeros((c,), device=dev)
    run = lambda: E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ho = h // 2 if kind == 'c' else h
    print(f'{os.environ.get("TAG","")} s2 {kind} N{n} {h}x{w} {c} k{k}: {ms:.3f} ms {2.0*n*ho*ho*c*c*k*k/ms/1e9:.0f} TF', flush=True)
for a in [(32, 256, 256, 128, 5, 'c'), (32, 128, 128, 128, 3, 'c'), (32, 64, 64, 128, 3, 't'), (32, 128, 128, 128, 5, 't'), (32, 256, 256, 64, 3, 'c')]:
    bench_s2(*a)
