import sys, torch
sys.path.insert(0, '.')
import master_thesis_amd as mt
from master_thesis_amd.synthetic import make_batch
dev = torch.device('cuda', 0)
kw = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
torch.manual_seed(0)
vgg = mt.VGGFeatures(device=dev); dfpn = mt.DFPN(vgg, **kw).to(dev); chn = mt.CHN(vgg, None, dfpn, **kw).to(dev)
opt = chn.configure_optimizers()['optimizer']
batch = make_batch(8, 5, 256, 256, dev)
for i in range(3):
    opt.zero_grad(); loss = chn.training_step(batch, 0); loss.backward(); opt.step()
torch.cuda.synchronize(); print(float(loss))
