"""Lab (round 4): a ResidualDenseBlock5C forward as it runs today (GATHER form: five launches, each re-stages the channel
prefix: 64, 96, 128, 160 -> 32 and 192 -> 64) against the SCATTER form (every channel window is staged ONCE and contributes
to all the later layers' pre-activations: 64 -> 192, then 32 -> 160, 128, 96, 64, partial sums added in place), on the
existing kernels, to size the idea before any kernel work."""
import os, sys
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E
from master_thesis_amd import _lib as L
dev = 'cuda:0'


def timeit(run, reps=10):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def lab(n, h, w):
    buf = torch.randn((n, h, w, 256), device=dev) * 0.5
    out5 = torch.empty((n, h, w, 64), device=dev)
    amax = E.new_amax(dev); E.absmax_into(buf, amax)
    gather = []
    for i in range(5):
        cin, cout = 64 + 32 * i, (32 if i < 4 else 64)
        spec = E.LayerSpec('c', cin, cout, 3, 1, L.ACT_LRELU if i < 4 else L.ACT_NONE)
        wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.03
        b = torch.zeros((cout,), device=dev)
        xin = buf[..., :cin]
        out = buf[..., cin:cin + 32] if i < 4 else out5
        gather.append((spec, xin, wt, b, out))
    scatter = []
    for i in range(5):
        c0, cin = (0, 64) if i == 0 else (32 + 32 * i, 32)
        cout = 192 - 32 * i
        spec = E.LayerSpec('c', cin, cout, 3, 1, L.ACT_NONE)
        wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.03
        b = torch.zeros((cout,), device=dev)
        xin = buf[..., c0:c0 + cin]
        out = buf[..., 256 - cout:256]
        scatter.append((spec, xin, wt, b, out, i > 0))
    tot_g = tot_s = 0.0
    for spec, xin, wt, b, out in gather:
        run = lambda: E.conv_forward(spec, E.view_of(xin), xin.shape[:3], wt, b, out=out, x_amax=amax, y_amax=amax)
        us = timeit(run); tot_g += us
        print(f'  gather  {spec.cin:3d} -> {spec.cout:3d}: {us:6.1f} us  {2.0*n*h*w*spec.cin*spec.cout*9/us/1e6:5.0f} TFLOP/s', flush=True)
    for spec, xin, wt, b, out, acc in scatter:
        run = lambda: E.conv_forward(spec, E.view_of(xin), xin.shape[:3], wt, b, out=out, res=out if acc else None,
                                     beta=1.0 if acc else 0.0, x_amax=amax, y_amax=amax)
        us = timeit(run); tot_s += us
        print(f'  scatter {spec.cin:3d} -> {spec.cout:3d}{" +=" if acc else "   "}: {us:6.1f} us  {2.0*n*h*w*spec.cin*spec.cout*9/us/1e6:5.0f} TFLOP/s', flush=True)
    print(f'N{n} {h}x{w}: gather {tot_g:.1f} us, scatter {tot_s:.1f} us per dense block', flush=True)


for shape in [(32, 64, 64), (16, 64, 64), (24, 128, 128), (12, 128, 128)]:
    lab(*shape)
