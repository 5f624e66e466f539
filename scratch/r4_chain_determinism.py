"""Round 4: is the CHN step bit-reproducible in ONE process when the RRDB trunk runs as two concurrent launch chains (two streams)?
(The fp32 VALU kernels showed sporadic differences when several PROCESSES shared the card.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import master_thesis_amd as mt
from master_thesis_amd.synthetic import make_batch
DEV = 'cuda:0'
KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
vgg = mt.VGGFeatures(device=DEV)
dfpn = mt.DFPN(vgg, **KW).to(DEV)
chn = mt.CHN(vgg, None, dfpn, **KW).to(DEV)
batch = make_batch(8, 5, 256, 256, DEV, seed=1234)
for kind, model in (('chn', chn), ('dfpn', dfpn)):
    opt = model.configure_optimizers()['optimizer']
    first, bad = None, 0
    for it in range(passes):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        torch.cuda.synchronize()
        cur = (float(loss), opt.flat_g.clone())
        if first is None:
            first = cur
            continue
        if cur[0] != first[0] or not torch.equal(cur[1], first[1]):
            bad += 1
            d = (cur[1] - first[1]).abs()
            print(f'{kind} pass {it}: loss {cur[0]!r} vs {first[0]!r}; gradient: {int((d > 0).sum())} entries differ, max {float(d.max()):.3e} '
                  f'of max|g| {float(first[1].abs().max()):.3e}', flush=True)
    print(f'{kind}: {bad} of {passes - 1} passes differ bitwise from the first', flush=True)
    del opt
