import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import numpy as np, torch
import seeded as S
from oracle import ref_torch as O
import master_thesis_amd as mt
DEV='cuda:0'; KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
# nearest resize check
g = np.load('tests/golden/g3_resize.npz')
fl64 = S.affine_flow(1, 2, 64, 64, 0, 'g3.f64')
got = mt.FlowsUtils.resize_flow(fl64.to(DEV), (16,16)).cpu().numpy()
d = np.abs(got - g['flow.64to16.nearest']); print('nearest maxdiff', d.max(), (d>0).sum(), np.argwhere(d>0)[:5])
ref = O.resize_flow(fl64, (16,16)).numpy(); print('oracle vs golden', np.abs(ref-g['flow.64to16.nearest']).max(), 'gpu vs oracle', np.abs(ref-got).max())

def grads_oracle(sd, batch, dtype, which):
    sd2 = {k: v.to(dtype) for k, v in sd.items()}
    names = [k for k in sd2 if which(k)]
    for n in names: sd2[n] = sd2[n].clone().requires_grad_(True)
    (x, m), y, info = batch
    cast = lambda t: t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t
    b2 = ((cast(x), cast(m)), cast(y), tuple(cast(i) for i in info))
    t0=time.time()
    loss, _ = O.dfpn_training_step(sd2, b2)
    gs = torch.autograd.grad(loss, [sd2[n] for n in names])
    print('oracle', dtype, float(loss), time.time()-t0, 's')
    return names, [g_.double() for g_ in gs]

sd = S.add_buffers(S.seeded_state_dict(S.dfpn_shapes()))
batch = S.synthetic_batch(2, 2, 64, 64, seed=9)
which = lambda k: ('model_vgg' not in k) and (k.endswith('weight') or k.endswith('bias'))
names, g32 = grads_oracle(sd, batch, torch.float32, which)
_, g64 = grads_oracle(sd, batch, torch.float64, which)
model = mt.DFPN(mt.VGGFeatures(device=DEV), **KW); model.load_state_dict(sd); model = model.to(DEV)
(x, m), y, info = batch
info = tuple(i.to(DEV) if torch.is_tensor(i) else i for i in info)
loss = model.training_step(((x.to(DEV), m.to(DEV)), y.to(DEV), info), 0); loss.backward()
print('gpu loss', float(loss))
P = dict(model.named_parameters())
rows=[]
for n, a32, a64 in zip(names, g32, g64):
    gg = P[n].grad.double().cpu()
    e_gpu = float((gg-a64).norm()/(a64.norm()+1e-30)); e_cpu = float((a32-a64).norm()/(a64.norm()+1e-30))
    rows.append((e_gpu, e_cpu, n))
rows.sort(reverse=True)
for r in rows[:12]: print('%.3e gpu  %.3e cpu32  %s' % r)
tot_gpu = np.sqrt(sum(float((P[n].grad.double().cpu()-a).norm()**2) for n,a in zip(names,g64)))/np.sqrt(sum(float(a.norm()**2) for a in g64))
tot_cpu = np.sqrt(sum(float((b-a).norm()**2) for b,a in zip(g32,g64)))/np.sqrt(sum(float(a.norm()**2) for a in g64))
print('global rel err gpu', tot_gpu, 'cpu32', tot_cpu)
