"""Round 4 diagnostic: after ONE optimizer step the CHN loss of the HIP run differs from the CPU oracle's by 6e-4 (the HIP
run and its one-rounding twin agree to 1e-5).  Which parameter tensors' FIRST Adam update differs?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')]
import torch
import seeded as S
from oracle import ref_torch as O
import master_thesis_amd as mt
from master_thesis_amd import engine as E
import bench
torch.set_num_threads(bench.host_cores())
DEV = 'cuda:0'
KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
kind = sys.argv[1] if len(sys.argv) > 1 else 'chn'
E.WGRAD_D_TERMS = int(os.environ.get('TERMS', '2'))
sd = S.add_buffers(S.seeded_state_dict(S.chn_shapes()), ('', 'model_aligner.'))
batch = S.synthetic_batch(1, 2, 64, 64, seed=3, flows_use=[True])
(x, m), y, info = batch
gb = ((x.to(DEV), m.to(DEV)), y.to(DEV), tuple(i.to(DEV) if torch.is_tensor(i) else i for i in info))
vgg = mt.VGGFeatures(device=DEV)
dfpn = mt.DFPN(vgg, **KW)
chn = mt.CHN(vgg, None, dfpn, **KW)
chn.load_state_dict(sd)
chn = chn.to(DEV)
model = dfpn if kind == 'dfpn' else chn
opt = model.configure_optimizers()['optimizer']
opt.zero_grad()
loss = model.training_step(gb, 0)
loss.backward()
torch.cuda.synchronize()
names = [n for n, p in model.named_parameters() if (('model_vgg' not in n) if kind == 'dfpn' else n.startswith('nn.'))]
params = dict(model.named_parameters())
g_hip = {n: params[n].grad.detach().cpu().clone() for n in names}
if kind == 'dfpn':
    sdc = {k[len('model_aligner.'):]: v.clone() for k, v in sd.items() if k.startswith('model_aligner.')}
    step = O.dfpn_training_step
else:
    sdc = {k: v.clone() for k, v in sd.items()}
    step = O.chn_training_step
res = {}
for dt in (torch.float32, torch.float64):
    s2 = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sdc.items()}
    for n in names:
        s2[n] = s2[n].clone().requires_grad_(True)
    cast = lambda t: t.to(dt) if torch.is_tensor(t) and t.is_floating_point() else t
    l, _ = step(s2, ((cast(x), cast(m)), cast(y), tuple(cast(i) for i in info)))
    res[dt] = (float(l), [g.double() for g in torch.autograd.grad(l, [s2[n] for n in names])])
print('loss hip', float(loss), 'cpu32', res[torch.float32][0], 'cpu64', res[torch.float64][0])
rows = []
upd = lambda g: g / (g.abs() + 1e-8)          # Adam's first update in units of lr
tot = {'hip': 0.0, 'c32': 0.0, 'n': 0}
for n, g32, g64 in zip(names, res[torch.float32][1], res[torch.float64][1]):
    gh = g_hip[n].double()
    ref = float(g64.norm()) + 1e-300
    e_h, e_c = float((gh - g64).norm()) / ref, float((g32 - g64).norm()) / ref
    u_h, u_c = float((upd(gh) - upd(g64)).abs().mean()), float((upd(g32) - upd(g64)).abs().mean())
    tot['hip'] += float((upd(gh) - upd(g64)).abs().sum()); tot['c32'] += float((upd(g32) - upd(g64)).abs().sum()); tot['n'] += gh.numel()
    rows.append((u_h, n, e_h, e_c, u_c, float(g64.abs().max()), float(g64.abs().median()), gh.numel()))
rows.sort(reverse=True)
print(f"mean |first Adam update - fp64's| in units of lr: HIP {tot['hip']/tot['n']:.4f}, CPU fp32 {tot['c32']/tot['n']:.4f}")
print('worst tensors by update difference (HIP):  upd_hip  upd_c32  relL2_hip  relL2_c32  max|g|  median|g|  numel  name')
for u_h, n, e_h, e_c, u_c, gmax, gmed, k in rows[:25]:
    print(f'  {u_h:.4f}  {u_c:.4f}  {e_h:.2e}  {e_c:.2e}  {gmax:.2e}  {gmed:.2e}  {k:8d}  {n}')
