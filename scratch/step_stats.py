"""N eager training steps of one model (argv[1] = dfpn|chn, argv[2] = steps) for rocprofv3 --stats:
every kernel's calls / N = launches per step."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import master_thesis_amd as mt
from master_thesis_amd.synthetic import make_batch
which, n = sys.argv[1], int(sys.argv[2])
dev = torch.device('cuda', 0)
kw = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
torch.manual_seed(0)
vgg = mt.VGGFeatures(device=dev)
model = mt.DFPN(vgg, **kw).to(dev) if which == 'dfpn' else mt.CHN(vgg, None, mt.DFPN(vgg, **kw), **kw).to(dev)
opt = model.configure_optimizers()['optimizer']
batch = make_batch(8, 5, 256, 256, dev, seed=1234)
for _ in range(n):
    opt.zero_grad(); loss = model.training_step(batch, 0); loss.backward(); opt.step()
torch.cuda.synchronize()
print(which, n, float(loss))
