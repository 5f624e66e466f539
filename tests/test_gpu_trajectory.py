"""Training trajectories: 20 optimizer steps of the DFPN and the CHN step on one small clip (64 x 64 frames; the DFPN's
networks still run at their fixed 256 x 256 / 64 x 64 / 16 x 16 resolutions, model_dfpn.py:74-77), five ways:

* H1: HIP, weight gradients with ONE fp16 term of dY (two partial products: the default, engine.WGRAD_D_TERMS = 1),
* H3: HIP, both terms (three partial products: fp32 grade per product),
* H3': H3 again from initial parameters that differ by ONE fp32 rounding (every weight times 1 +- 2^-24),
* C32: the CPU oracle in fp32 with ``torch.optim.Adam`` -- what the reference trains with (model_dfpn.py:295-308,
  model_chn.py:378-391, training_step :135-158 / :256-280),
* C64: the same in fp64 -- the trajectory exact arithmetic would take.

Gate of the one-term form.  What was asked for -- the one-term loss trajectory within
1e-5 of the three-product one over 20 steps -- cannot hold for ANY two fp32 implementations of this step, the reference's
own included, and the test shows that instead of hiding it.  Adam divides by sqrt(v): in the first steps EVERY parameter
moves by about lr whatever the size of its gradient, an entry whose gradient is at the rounding level moves in a direction
that is noise, and ReLU / clamp / nearest kinks do the rest.  Measured (round 4, gpurun_out/r4b_traj.log): after 20 DFPN steps
H3 and its one-rounding twin H3' differ by 1.8e-2 in the loss and by 0.80 of the whole update |theta_20 - theta_0| in the
parameters; H3 vs C32 2.5e-2 / 0.84; H1 vs H3 2.0e-2 / 0.79.  On the CHN step the fp32 CPU oracle is the outlier: its
gradient at this clip is off by 1e-2 per tensor against fp64 (kink flips at 16 x 16 trunk resolution; HIP: 5e-4), its first
Adam update deviates from the fp64 one by 0.0065 lr on average (HIP, one or two terms alike: 0.0002;
scratch/r4_traj_diag.py), and from step 2 on it descends along another path (loss 5.897 against 5.685 at step 3).
The criterion is therefore RELATIVE TO THAT SENSITIVITY, against a yardstick trajectory Y (CHN: C64; DFPN: C32 -- twenty
fp64 DFPN steps are three minutes of CPU and were run once, gpurun_out/r4e_traj.log: H3 / H1 / H3' / C32 sit 7.7e-2 / 7.3e-2 /
8.3e-2 / 6.4e-2 in the loss and 0.85 / 0.84 / 0.86 / 0.80 in the parameters from C64, one cloud):
(i) before the dynamics have amplified anything (steps 1-2) H1 = H3 to 2e-5;
(ii) over the whole trajectory H1 stays as close to Y as the other members of the cloud do (H3, H3', and C32 where Y is
     C64): within a factor 2 of the worst of them in the loss (a maximum over 20 chaotic steps moves by that much from run
     to run) and 1.25 in the parameters -- i.e. the one-term form trains no worse than the reference's own fp32 arithmetic;
(iii) every run trains (the loss falls).  A BIAS of the one-term gradients would show as H1 leaving the cloud.
Measured on the CHN step: loss H3 4.2e-2, H1 2.4e-2, C32 4.7e-2; parameters H3 0.55, H1 0.67, C32 1.01 from C64.
"""
import sys

import pytest
import torch

import seeded as S
from oracle import ref_torch as O

pytestmark = pytest.mark.gpu
KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
DEV = 'cuda:0'
STEPS = 20


def _to_dev(batch):
    (x, m), y, info = batch
    return (x.to(DEV), m.to(DEV)), y.to(DEV), tuple(i.to(DEV) if torch.is_tensor(i) else i for i in info)


def _hip_run(kind, sd, batch, terms, jitter=False):
    import master_thesis_amd as mt
    from master_thesis_amd import engine as E
    old, E.WGRAD_D_TERMS = E.WGRAD_D_TERMS, terms
    try:
        vgg = mt.VGGFeatures(device=DEV)
        dfpn = mt.DFPN(vgg, **KW)
        chn = mt.CHN(vgg, None, dfpn, **KW)
        chn.load_state_dict(sd)
        chn = chn.to(DEV)
        model = dfpn if kind == 'dfpn' else chn
        opt = model.configure_optimizers()['optimizer']
        if jitter:          # one fp32 rounding on every trainable weight
            g = torch.Generator(device='cpu').manual_seed(99)
            sign = (torch.randint(0, 2, (opt.flat_p.numel(),), generator=g).float() * 2 - 1).to(DEV)
            opt.flat_p.mul_(1.0 + sign * 2.0 ** -24)
            from master_thesis_amd import engine as E2
            E2.bump_weight_epoch()
        gbatch = _to_dev(batch)
        losses = []
        for _ in range(STEPS):
            opt.zero_grad()
            loss = model.training_step(gbatch, 0)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        torch.cuda.synchronize()
        names = [n for n, _ in model.named_parameters()]
        return losses, {n: p.detach().cpu().clone() for n, p in model.named_parameters()}, names
    finally:
        E.WGRAD_D_TERMS = old


def _cpu_run(kind, sd, batch, trainable, dtype=torch.float32):
    """The oracle's step + torch.optim.Adam on the host, in ``dtype``."""
    cast = lambda t: t.to(dtype) if torch.is_tensor(t) and t.is_floating_point() else t     # noqa: E731
    sd = {k: cast(v.clone()) for k, v in sd.items()}
    (x, m), y, info = batch
    batch = ((cast(x), cast(m)), cast(y), tuple(cast(i) for i in info))
    if kind == 'dfpn':
        sd = {k[len('model_aligner.'):]: v for k, v in sd.items() if k.startswith('model_aligner.')}
        step = O.dfpn_training_step
    else:
        step = O.chn_training_step
    leaves = []
    for n in trainable:
        sd[n] = sd[n].clone().requires_grad_(True)
        leaves.append(sd[n])
    opt = torch.optim.Adam(leaves, lr=KW['lr'])
    losses = []
    for _ in range(STEPS):
        opt.zero_grad()
        loss, _ = step(sd, batch)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses, {n: sd[n].detach() for n in trainable}


def _dist(a, b, ref0, names):
    """|a - b| over |b - theta_0|, L2 over all trainable parameters."""
    num = sum(float((a[n].double() - b[n].double()).pow(2).sum()) for n in names)
    den = sum(float((b[n].double() - ref0[n].double()).pow(2).sum()) for n in names)
    return (num / den) ** 0.5


def _maxrel(a, b):
    return max(abs(x - y) / abs(y) for x, y in zip(a, b))


@pytest.mark.parametrize('kind', ['dfpn', 'chn'])
def test_one_term_weight_gradients_track_training(kind):
    sys.path.insert(0, '.')
    sd = S.add_buffers(S.seeded_state_dict(S.chn_shapes()), ('', 'model_aligner.'))
    batch = S.synthetic_batch(1, 2, 64, 64, seed=3, flows_use=[True])
    l3, p3, names = _hip_run(kind, sd, batch, 2)
    l1, p1, _ = _hip_run(kind, sd, batch, 1)
    l3j, p3j, _ = _hip_run(kind, sd, batch, 2, jitter=True)
    if kind == 'dfpn':
        trainable = [n for n in names if 'model_vgg' not in n]
        key = lambda n: 'model_aligner.' + n                                     # noqa: E731
    else:
        trainable = [n for n in names if n.startswith('nn.')]
        key = lambda n: n                                                        # noqa: E731
    lc, pc = _cpu_run(kind, sd, batch, trainable)
    theta0 = {n: sd[key(n)] for n in trainable}
    d = lambda a, b: _dist(a, b, theta0, trainable)                              # noqa: E731
    runs = {'H3': (l3, p3), 'H1': (l1, p1), "H3'": (l3j, p3j)}
    if kind == 'chn':
        yname, (ly, py) = 'C64', _cpu_run(kind, sd, batch, trainable, torch.float64)
        runs['C32'] = (lc, pc)
    else:
        yname, (ly, py) = 'C32', (lc, pc)
    ry = {k: _maxrel(v[0], ly) for k, v in runs.items()}
    dy = {k: d(v[1], py) for k, v in runs.items()}
    early = _maxrel(l1[:2], l3[:2])
    fmt = lambda ls: ' '.join(f'{v:.5f}' for v in ls)                            # noqa: E731
    print(f'{kind}: losses {yname} {fmt(ly)}')
    for k, v in runs.items():
        print(f'      losses {k:3s} {fmt(v[0])}')
    print(f'{kind}: max relative loss difference to {yname} over {STEPS} steps: ' + ', '.join(f'{k} {v:.2e}' for k, v in ry.items()) +
          f'; H1-H3 {_maxrel(l1, l3):.2e} (steps 1-2: {early:.2e}), H3\'-H3 {_maxrel(l3j, l3):.2e}')
    print(f'{kind}: parameter distance to {yname} in units of |theta_{STEPS} - theta_0|: ' + ', '.join(f'{k} {v:.3e}' for k, v in dy.items()) +
          f'; H1-H3 {d(p1, p3):.3e}, H3\'-H3 {d(p3j, p3):.3e}')
    for name, ls in [(yname, ly)] + [(k, v[0]) for k, v in runs.items()]:
        assert min(ls[-3:]) < 0.95 * ls[0], f'{name} must be training: {ls}'
    assert early <= 2e-5, (early, l1[:2], l3[:2])
    cloud = [k for k in runs if k != 'H1']
    assert ry['H1'] <= 2.0 * max(ry[k] for k in cloud) + 1e-4, ry
    assert dy['H1'] <= 1.25 * max(dy[k] for k in cloud) + 1e-3, dy
