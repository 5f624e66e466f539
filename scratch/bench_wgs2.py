import sys, torch
sys.path.insert(0, '.')
from master_thesis_amd import engine as E, _lib as L
dev = 'cuda:0'
E.PRECISION = 'f16x3'
def bench(kind, n, h, cin, cout, k, reps=10):
    spec = E.LayerSpec(kind, cin, cout, k, 2, 1)
    if kind == 'c':
        x = torch.randn((n, h, h, E.cpad(cin)), device=dev); ho, _ = spec.out_hw(h, h)
        dz = torch.randn((n, ho, ho, E.cpad(cout)), device=dev); wt = torch.randn((cout, cin, k, k), device=dev) * 0.05
    else:
        x = torch.randn((n, h, h, E.cpad(cin)), device=dev); ho, _ = spec.out_hw(h, h)
        dz = torch.randn((n, ho, ho, E.cpad(cout)), device=dev); wt = torch.randn((cin, cout, k, k), device=dev) * 0.05
    run = lambda: E.conv_backward_weight(spec, dz, E.view_of(x), wt)
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    px = n * (ho * ho if kind == 'c' else h * h)
    print(f'wgrad s2 {kind} N{n} in{h} {cin}->{cout} k{k}: {ms:.3f} ms {2.0 * px * cin * cout * k * k / ms / 1e9:.1f} TF', flush=True)
for a in [('c', 32, 256, 128, 128, 5), ('c', 32, 128, 128, 128, 3), ('t', 32, 128, 128, 128, 5), ('t', 32, 64, 128, 128, 3)]:
    bench(*a)
