import sys, os, time
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E, _lib as L
dev='cuda:0'
def bench(n,h,w,cin,cout,k,prec,halo=True,reps=5, kind='fwd'):
    E.PRECISION=prec; E.USE_HALO=halo
    spec = E.LayerSpec('c', cin, cout, k, 1, 1)
    x = torch.randn((n,h,w,E.cpad(cin)), device=dev); wt = torch.randn((cout,cin,k,k), device=dev)*0.05; b = torch.zeros((cout,), device=dev)
    dz = torch.randn((n,h,w,E.cpad(cout)), device=dev)
    def run():
        if kind == 'fwd': E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b)
        elif kind == 'wgrad': E.conv_backward_weight(spec, dz, E.view_of(x), wt)
        else:
            dx = torch.empty_like(x); E.conv_backward_data(spec, dz, wt, E.view_of(dx))
    run(); run(); torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    fl = 2.0*n*h*w*cin*cout*k*k
    print(f'{kind} {prec} halo={halo} N{n} {h}x{w} {cin}->{cout} k{k}: {ms:.3f} ms  {fl/ms/1e9:.1f} TF', flush=True)
if __name__ == '__main__':
    shapes = [(32,256,256,128,128,3), (8192,16,16,256,256,3), (32,64,64,160,32,3), (32,64,64,192,64,3), (32,256,256,128,128,5)]
    for sh in shapes:
        for prec in ('bf16x3','bf16x6','f16x3'):
            bench(*sh, prec, True, kind=sys.argv[1] if len(sys.argv)>1 else 'fwd')
