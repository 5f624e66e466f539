"""One weight-gradient shape, a few launches (PMC target)."""
import sys
sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E
dev = 'cuda:0'
n, h, w, cin, cout, k = 32, 256, 256, 128, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 3
spec = E.LayerSpec('c', cin, cout, k, 1, 1)
x = torch.randn((n, h, w, E.cpad(cin)), device=dev); wt = torch.randn((cout, cin, k, k), device=dev) * 0.05
dz = torch.randn((n, h, w, E.cpad(cout)), device=dev)
for _ in range(4):
    E.conv_backward_weight(spec, dz, E.view_of(x), wt)
torch.cuda.synchronize()
