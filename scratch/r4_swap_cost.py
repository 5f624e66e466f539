"""Round 4: what does SeparableConv4d's permute-as-addressing (engine.swap_view) cost the 256 -> 256 layer that reads through it,
the backward-data that writes through it and the weight gradient that gathers through it, against the same launches on plain views?"""
import sys
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E
dev = 'cuda:0'
G, P = 32, 256                      # 32 (clip, reference) pairs x 256 positions = 8192 images of 16 x 16
spec = E.LayerSpec('c', 256, 256, 3, 1, 1)
x = torch.randn((G * P, 16, 16, 256), device=dev)
wt = torch.randn((256, 256, 3, 3), device=dev) * 0.03
b = torch.zeros((256,), device=dev)
dz = torch.randn((G * P, 16, 16, 256), device=dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, view in (('plain', lambda t: E.view_of(t)), ('swapped', lambda t: E.swap_view(t, G))):
    out = torch.empty_like(x)
    dx = torch.empty_like(x)
    f = timeit(lambda: E.conv_forward(spec, view(x), (G * P, 16, 16), wt, b, out=out))
    bd = timeit(lambda: E.conv_backward_data(spec, dz, wt, view(dx)))
    wg = timeit(lambda: E.conv_backward_weight(spec, dz, view(x), wt))
    print(f'{name:8s} forward {f:.3f} ms  backward-data {bd:.3f} ms  weight gradient {wg:.3f} ms', flush=True)
# the alternative: a physical transpose of the 2.1 GB tensor
t = timeit(lambda: x.view(G, P, P, 256).transpose(1, 2).contiguous())
print(f'physical transpose of the (G, P1, P2, C) tensor (torch): {t:.3f} ms')
