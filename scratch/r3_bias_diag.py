import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np, torch
import seeded as S
from oracle import ref_torch as O
from _gradcheck import oracle_grads
import master_thesis_amd as mt
from master_thesis_amd import engine as E
import bench
torch.set_num_threads(bench.host_cores())
DEV = 'cuda:0'
KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
sd = S.add_buffers(S.seeded_state_dict(S.dfpn_shapes()))
batch = S.synthetic_batch(2, 2, 64, 64, seed=9)
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g9_grads.npz'))
names = [k[len('dfpn.norm/'):] for k in g.files if k.startswith('dfpn.norm/')]
g64 = dict(zip(names, oracle_grads(O.dfpn_training_step, sd, batch, torch.float64, names)))
(x, m), y, info = batch
gb = ((x.to(DEV), m.to(DEV)), y.to(DEV), tuple(i.to(DEV) if torch.is_tensor(i) else i for i in info))
for terms in (2, 1):
    E.WGRAD_D_TERMS = terms
    model = mt.DFPN(mt.VGGFeatures(device=DEV), **KW); model.load_state_dict(sd); model = model.to(DEV)
    loss = model.training_step(gb, 0); loss.backward()
    P = dict(model.named_parameters())
    rows = []
    for n in names:
        gg = P[n].grad.double().cpu()
        rows.append((float((gg - g64[n]).norm() / (g64[n].norm() + 1e-30)), abs(float(gg.norm()) - float(g[f'dfpn.norm/{n}'])) / float(g[f'dfpn.norm/{n}']), n))
    rows.sort(reverse=True)
    print('d_terms', terms, 'worst vs fp64 / vs golden norm:')
    for r in rows[:8]:
        print('   %.2e  %.2e  %s' % r)
