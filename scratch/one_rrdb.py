import sys; sys.path.insert(0, '.')
import torch
from master_thesis_amd import engine as E
dev = 'cuda:0'
n, h, w, cin, cout = 32, 64, 64, 160, 32
spec = E.LayerSpec('c', cin, cout, 3, 1, 2)
buf = torch.randn((n, h, w, 192), device=dev); x = buf[..., :cin]
out = torch.empty((n, h, w, 192), device=dev)[..., 64:96]
wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.05; b = torch.zeros((cout,), device=dev)
xa = E.absmax_into(x, E.new_amax(dev)); ya = E.new_amax(dev)
for _ in range(6): E.conv_forward(spec, E.view_of(x), x.shape[:3], wt, b, out=out, x_amax=xa, y_amax=ya)
torch.cuda.synchronize()
