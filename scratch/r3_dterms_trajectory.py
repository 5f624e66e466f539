"""Round 3, review item 4(ii) gate: the loss trajectory of four optimizer steps with two-product weight gradients
(d_terms = 1) against the three-product run, same batch and weights (2 clips of the benchmarked batch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import master_thesis_amd as mt
from master_thesis_amd import engine as E
from master_thesis_amd.synthetic import make_batch
DEV = 'cuda:0'
KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)


def cut(batch, n):
    (x, m), y, info = batch
    c = lambda t: t[:n].contiguous() if torch.is_tensor(t) else t[:n]
    return (c(x), c(m)), c(y), tuple(c(i) for i in info)


batch = cut(make_batch(8, 5, 256, 256, DEV, seed=1234), 2)
res = {}
for kind in ('dfpn', 'chn'):
    for terms in (2, 1):
        E.WGRAD_D_TERMS = terms
        torch.manual_seed(0)
        vgg = mt.VGGFeatures(device=DEV)
        dfpn = mt.DFPN(vgg, **KW).to(DEV)
        model = dfpn if kind == 'dfpn' else mt.CHN(vgg, None, dfpn, **KW).to(DEV)
        opt = model.configure_optimizers()['optimizer']
        losses = []
        for _ in range(4):
            opt.zero_grad()
            loss = model.training_step(batch, 0)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        res[(kind, terms)] = (losses, opt.flat_g.clone())
        del model, opt, dfpn, vgg
        torch.cuda.empty_cache()
    a, b = res[(kind, 2)], res[(kind, 1)]
    rel = [abs(x - y) / abs(x) for x, y in zip(a[0], b[0])]
    print(kind, 'losses 3-product', a[0])
    print(kind, 'losses 2-product', b[0])
    print(kind, 'relative difference per step', ['%.2e' % r for r in rel], flush=True)
