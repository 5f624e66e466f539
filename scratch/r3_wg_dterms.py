"""Round 3, review item 4(ii): weight gradients with ONE 16-bit term of dY (two partial products) against two terms
(three products): time per layer on the producer / consumer kernel and error against an fp64 reference."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
import torch
import torch.nn.functional as F
from master_thesis_amd import engine as E
from bench_layer import bench
dev = 'cuda:0'


def err(n, h, w, cin, cout, k):
    spec = E.LayerSpec('c', cin, cout, k, 1, 1)
    g = torch.Generator(device='cpu').manual_seed(1)
    x = torch.randn((n, h, w, cin), generator=g).to(dev)
    # a gradient-like dY: heavy-tailed magnitudes, random signs
    dz = (torch.randn((n, h, w, cout), generator=g) * torch.exp(2 * torch.randn((n, h, w, cout), generator=g))).to(dev)
    wt = torch.zeros((cout, cin, k, k), device=dev)
    dw, db = E.conv_backward_weight(spec, dz, E.view_of(x), wt)
    xr = x.permute(0, 3, 1, 2).double().requires_grad_(False)
    wr = wt.double().requires_grad_(True)
    y = F.conv2d(xr, wr, None, padding=k // 2)
    ref, = torch.autograd.grad(y, wr, dz.permute(0, 3, 1, 2).double())
    e = float((dw.double() - ref).norm() / ref.norm())
    eb = float((db.double() - dz.double().sum((0, 1, 2))).norm() / dz.double().sum((0, 1, 2)).norm())
    return e, eb


for terms in (2, 1, 2, 1):
    E.WGRAD_D_TERMS = terms
    print(f'--- d_terms = {terms}', flush=True)
    for sh in [(32, 256, 256, 128, 128, 3), (8192, 16, 16, 256, 256, 3), (32, 64, 64, 160, 128, 3), (32, 64, 64, 192, 64, 3),
               (32, 128, 128, 128, 128, 3), (32, 256, 256, 128, 128, 5), (32, 128, 128, 128, 128, 5)]:
        bench(*sh, 'f16x3', True, reps=10, kind='wgrad')
for terms in (2, 1):
    E.WGRAD_D_TERMS = terms
    for sh in [(4, 64, 64, 128, 128, 3), (4, 64, 64, 128, 128, 5), (64, 16, 16, 256, 256, 3)]:
        print(f'd_terms {terms} {sh}: rel L2 error of dW, db vs fp64:', err(*sh), flush=True)
