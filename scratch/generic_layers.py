"""Per-launch listing of the generic (non-halo) conv-engine launches of one DFPN step."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import master_thesis_amd as mt
from master_thesis_amd import engine as E
from master_thesis_amd.synthetic import make_batch
dev = torch.device('cuda', 0)
kw = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
torch.manual_seed(0)
dfpn = mt.DFPN(mt.VGGFeatures(device=dev), **kw).to(dev)
opt = dfpn.configure_optimizers()['optimizer']
batch = make_batch(8, 5, 256, 256, dev, seed=1234)
def step():
    opt.zero_grad(); loss = dfpn.training_step(batch, 0); loss.backward(); opt.step()
step(); step()
E.PROFILE = []
step()
torch.cuda.synchronize()
rows = [(n, f, e0.elapsed_time(e1)) for n, f, e0, e1 in E.PROFILE]
E.PROFILE = None
tot = sum(r[2] for r in rows)
print('total engine ms', round(tot, 1))
for n, f, ms in sorted(rows, key=lambda r: -r[2]):
    if 'halo' in n and ms < 1.5: continue
    print(f'{n:46s} {ms:7.3f} ms {f/ms/1e9:7.1f} TF  {f/1e9:9.1f} GF')
