"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py [name ...]

Inputs and weights come from tests/golden/seeded.py (regenerated from seeds by
the tests, never stored); only the reference's OUTPUTS are stored.  Nothing
from the reference's source travels: the .npz files are data.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import seeded as S  # noqa: E402
from _ref_import import import_reference, make_vgg  # noqa: E402
from make_inputs import warp_grid  # noqa: E402

mt = import_reference()
KW = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB', flush=True)


def ref_dfpn():
    vgg = make_vgg(mt)
    model = mt.DFPN(vgg, **KW)
    sd = S.add_buffers(S.seeded_state_dict(S.dfpn_shapes()))
    model.load_state_dict(sd)
    return model


def ref_chn(nb=20):
    vgg = make_vgg(mt)
    aligner = mt.DFPN(vgg, **KW)
    model = mt.CHN(vgg, None, aligner, **KW)
    assert nb == 20
    sd = S.add_buffers(S.seeded_state_dict(S.chn_shapes()),
                       ('', 'model_aligner.'))
    model.load_state_dict(sd)
    return model


def grad_summary(model, names):
    out = {}
    for n, p in model.named_parameters():
        if n not in names:
            continue
        g = p.grad
        out['norm/' + n] = g.norm().double()
        flat = g.reshape(-1)
        out['head/' + n] = flat[:: max(1, flat.numel() // 64)][:64]
    return out


# ---------------------------------------------------------------- fixtures --
def g1_corr():
    M = mt.model_dfpn.CorrelationVGG
    res = {}
    for tag, (b, r, c) in {'small': (2, 2, 64), 'c512': (1, 1, 512)}.items():
        ft = S.uniform(f'g1.ft.{tag}', (b, c, 16, 16), -1, 2)
        fr = S.uniform(f'g1.fr.{tag}', (b, c, r, 16, 16), -1, 2)
        vt = (S.uniform(f'g1.vt.{tag}', (b, 1, 16, 16)) > 0.3).float()
        vr = (S.uniform(f'g1.vr.{tag}', (b, 1, r, 16, 16)) > 0.3).float()
        res[f'{tag}.masked'] = M.correlation_masked_4d(ft, vt, fr, vr)
        res[f'{tag}.unmasked'] = M.correlation_masked_4d(ft, None, fr, None)
        # the optional 3D softmax over (reference frame, reference position), model_dfpn.py:567-583;
        # scaled so that the distribution is far from uniform
        res[f'{tag}.softmax'] = M.softmax_3d(res[f'{tag}.masked'] * 20.0)
    save('g1_corr', **res)


def g2_warp():
    res = {}
    for s in (16, 64):
        b, f = 2, 2
        x = S.uniform(f'g2.x.{s}', (b, 3, f, s, s))
        v = (S.uniform(f'g2.v.{s}', (b, 1, f, s, s)) > 0.4).float()
        g = warp_grid(b, f, s, s, f'g2.g.{s}')
        xa, va = mt.FlowsUtils.align_set(x, v, g)
        res[f'x.{s}'], res[f'v.{s}'] = xa, va
        # gradient wrt the grid of sum(xa * wgt)
        g2 = g.clone().requires_grad_(True)
        wgt = S.uniform(f'g2.w.{s}', (b, 3, f, s, s), -1, 1)
        xa2, _ = mt.FlowsUtils.align_set(x, v, g2)
        (xa2 * wgt).sum().backward()
        res[f'dgrid.{s}'] = g2.grad
    save('g2_warp', **res)


def g3_resize():
    res = {}
    fl = S.affine_flow(2, 2, 16, 16, 0, 'g3.f16')
    res['flow.16to64.bilinear'] = mt.FlowsUtils.resize_flow(fl, (64, 64), mode='bilinear')
    fl64 = S.affine_flow(1, 2, 64, 64, 0, 'g3.f64')
    res['flow.64to256.bilinear'] = mt.FlowsUtils.resize_flow(fl64, (256, 256), mode='bilinear')[:, :, ::5, ::7]
    res['flow.64to16.nearest'] = mt.FlowsUtils.resize_flow(fl64, (16, 16))
    fl48 = S.affine_flow(1, 1, 48, 80, 0, 'g3.f48')
    res['flow.48x80to40x56.bilinear'] = mt.FlowsUtils.resize_flow(fl48, (40, 56), mode='bilinear')
    res['flow.48x80to16.nearest'] = mt.FlowsUtils.resize_flow(fl48, (16, 16))
    x = S.uniform('g3.x', (2, 3, 2, 64, 64))
    v = (S.uniform('g3.v', (2, 1, 2, 64, 64)) > 0.4).float()
    y = S.uniform('g3.y', (2, 3, 2, 64, 64))
    xn, vn, yn = mt.TransformsUtils.resize_set(x, v, y, 16)
    res['set16.x'], res['set16.v'], res['set16.y'] = xn, vn, yn
    a, b_, c, d = mt.TransformsUtils.resize_set_bis(x[:, :, 0], v[:, :, 0], x, v, (32, 32))
    res['bis32.xt'], res['bis32.mt'], res['bis32.xr'], res['bis32.mr'] = a, b_, c, d
    x96 = S.uniform('g3.x96', (1, 3, 1, 96, 72))
    v96 = (S.uniform('g3.v96', (1, 1, 1, 96, 72)) > 0.4).float()
    a, b_, c, d = mt.TransformsUtils.resize_set_bis(x96[:, :, 0], v96[:, :, 0], x96, v96, (64, 64))
    res['bis64.xt'], res['bis64.mt'] = a, b_
    save('g3_resize', **res)


def g4_modules():
    model = ref_dfpn()
    res = {}
    with torch.no_grad():
        corr = S.uniform('g4.corr', (1, 1, 16, 16, 16, 16), -1, 1)
        res['sepconv4d'] = model.corr.conv(corr)
        vol = S.uniform('g4.vol', (1, 2, 16, 16, 16, 16), -1, 1)
        res['mixer'] = model.corr_mixer(vol)
        xt = S.uniform('g4.xt', (1, 3, 64, 64), -2, 2)
        mt_ = (S.uniform('g4.mt', (1, 1, 64, 64)) > 0.7).float()
        xr = S.uniform('g4.xr', (1, 3, 2, 64, 64), -2, 2)
        mr = (S.uniform('g4.mr', (1, 1, 2, 64, 64)) > 0.7).float()
        fp = S.affine_flow(1, 2, 64, 64, 0, 'g4.fp')
        res['flow_est'] = model.flow_64(xt, mt_, xr, mr, fp)
    save('g4_modules', **res)


def g5_dfpn_fwd():
    """BASELINE config 1: DFPN forward, frames_n=2, 128x128, batch 1."""
    model = ref_dfpn()
    (x, m), y, info = S.synthetic_batch(1, 2, 128, 128, seed=5)
    t, r = mt.DFPN.get_indexes(2)
    t0 = time.time()
    with torch.no_grad():
        corr, f16, f64, f = model(x[:, :, t], m[:, :, t], x[:, :, r], m[:, :, r])
        xa, va, vmap = model.align(x[:, :, t], m[:, :, t], x[:, :, r], m[:, :, r])
    print('cfg1 reference forward+align', time.time() - t0, 's')
    save('g5_dfpn_fwd', corr=corr, flow_16=f16, flow_64=f64, flow=f,
         x_aligned=xa, v_aligned=va, v_map=vmap)


def g6_dfpn_loss():
    model = ref_dfpn()
    res = {}
    for tag, use in {'mixed': [True, False], 'all': [True, True],
                     'none': [False, False]}.items():
        batch = S.synthetic_batch(2, 3, 64, 64, seed=6, flows_use=use)
        with torch.no_grad():
            (x, m), y, info = batch
            t, r = mt.DFPN.get_indexes(3)
            out = mt.DFPN._train_val_wrapper(model, x, m, y, info[4], info[2], t, r)
            loss, items = model.compute_loss(*out, t, r)
        res[f'{tag}.loss'] = loss
        res[f'{tag}.items'] = torch.stack([i.reshape(()) if i.numel() == 1 else i for i in items])
    save('g6_dfpn_loss', **res)


def g7_rrdbnet():
    model = ref_chn()
    x = S.uniform('g7.x', (1, 9, 32, 32), -1, 1)
    with torch.no_grad():
        y = model.nn(x)
        rdb = model.nn.rrdb_trunk[0].rdb1(S.uniform('g7.rdb', (1, 64, 16, 16), -1, 1))
    save('g7_rrdbnet', y=y, rdb=rdb)


def g8_chn_loss():
    model = ref_chn()
    res = {}
    for r in (1, 2):
        b, h = 2, 64
        yt = S.uniform(f'g8.yt.{r}', (b, 3, h, h))
        vt = (S.uniform(f'g8.vt.{r}', (b, 1, h, h)) > 0.3).float()
        yh = S.uniform(f'g8.yh.{r}', (b, 3, r, h, h))
        vm = ((S.uniform(f'g8.vm.{r}', (b, 1, r, h, h)) > 0.5).float()
              * (1 - vt).unsqueeze(2))
        yc = vt.unsqueeze(2) * yt.unsqueeze(2) + (1 - vt).unsqueeze(2) * yh
        with torch.no_grad():
            if r == 1:
                loss, items = model.compute_loss(yt, vt, yh, yc, vm)
            else:
                # the reference raises here for R>1 (model_chn.py:371-373):
                # the golden is the reference's own LossesUtils.* on F-folded
                # tensors, as model_chn.py:365-366 already does for the
                # perceptual term.
                fold = lambda a: a.transpose(1, 2).reshape(-1, a.shape[1], h, h)  # noqa: E731
                tgt = yt.unsqueeze(2).repeat(1, 1, r, 1, 1)
                nh = vt.unsqueeze(2).repeat(1, 1, r, 1, 1)
                items = [
                    mt.LossesUtils.masked_l1(yh, tgt, nh, reduction='sum', weight=0.5),
                    mt.LossesUtils.masked_l1(yh, tgt, vm, reduction='sum', weight=2),
                    mt.LossesUtils.masked_l1(yc, tgt, (1 - nh) - vm, reduction='sum', weight=1),
                    mt.LossesUtils.perceptual(fold(yh), fold(tgt), model_vgg=model.model_vgg, weight=0.5)[0],
                    mt.LossesUtils.grad(fold(yh), fold(tgt), reduction='mean', weight=1)]
                loss = sum(items)
        res[f'r{r}.loss'] = loss
        res[f'r{r}.items'] = torch.stack(list(items))
    save('g8_chn_loss', **res)


def g9_grads():
    # DFPN: one training_step + backward, B=2 (flows_use mixed), F=2, 64x64
    model = ref_dfpn()
    batch = S.synthetic_batch(2, 2, 64, 64, seed=9)
    loss = model.training_step(batch, 0)
    loss.backward()
    names = [n for n, p in model.named_parameters() if p.grad is not None]
    res = {'dfpn.loss': loss}
    res.update({'dfpn.' + k: v for k, v in grad_summary(model, names).items()})
    none = [n for n, p in model.named_parameters() if p.grad is None]
    assert all('model_vgg' in n for n in none), none
    # CHN: B=1, F=2, 64x64
    model = ref_chn()
    batch = S.synthetic_batch(1, 2, 64, 64, seed=10)
    loss = model.training_step(batch, 0)
    loss.backward()
    names = [n for n, p in model.named_parameters() if p.grad is not None]
    assert all(n.startswith('nn.') for n in names)
    res['chn.loss'] = loss
    res.update({'chn.' + k: v for k, v in grad_summary(model, names).items()})
    save('g9_grads', **res)


def g10_vgg():
    vgg = ref_dfpn().model_vgg
    x = S.uniform('g10.x', (2, 3, 64, 64))
    with torch.no_grad():
        taps = vgg(x)
        taps_raw = vgg(x, normalize_input=False)
    save('g10_vgg', **{f'tap{i}': t for i, t in enumerate(taps)},
         **{f'raw{i}': t for i, t in enumerate(taps_raw)})


def g11_cpn_warp():
    """The warp tail of CPN.align (model_cpn.py:75-91), run through the
    reference's own torch calls with fixed thetas."""
    import torch.nn.functional as F
    b, r, h, w = 1, 3, 32, 48
    x = S.uniform('g11.x', (b, 3, r, h, w))
    m = (S.uniform('g11.m', (b, 1, r, h, w)) > 0.6).float()
    m_t = (S.uniform('g11.mt', (b, 1, h, w)) > 0.6).float()
    theta = torch.tensor([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]],
                          [[0.9, 0.1, 0.2], [-0.1, 1.1, -0.3]],
                          [[1.2, -0.3, -0.5], [0.2, 0.8, 0.4]]])
    c = 3
    grid = F.affine_grid(theta, [theta.size(0), c, h, w], align_corners=False)
    xa = F.grid_sample(x.transpose(1, 2).reshape(-1, c, h, w), grid,
                       align_corners=False).reshape(b, r, c, h, w).transpose(1, 2)
    va = (F.grid_sample(1 - m.transpose(1, 2).reshape(-1, 1, h, w), grid,
                        align_corners=False)
          .reshape(b, r, 1, h, w).transpose(1, 2) > 0.5).float()
    vmaps = (va - (1 - m_t.unsqueeze(2))).clamp(0, 1)
    save('g11_cpn_warp', theta=theta, x_aligned=xa, v_aligned=va, v_maps=vmaps)


def g12_inpaint():
    """The three test-time drivers of CHN (model_chn.py:87-254) on a 4-frame 64x64 clip
    (inpaint_cp with N = 2 iterations), and the index helpers."""
    model = ref_chn()
    (x, m), y, info = S.synthetic_batch(1, 4, 64, 64, seed=12)
    res = {}
    with torch.no_grad():
        res['ff'] = model.inpaint_ff(x[0].clone(), m[0].clone())
        res['ip'] = model.inpaint_ip(x[0].clone(), m[0].clone())
        res['cp'] = model.inpaint_cp(x[0].clone(), m[0].clone(), N=2)
    C = mt.CHN
    res['idx_ff'] = np.array([C.get_indexes_ff(t, 7, s, D) + [-1] * 7 for t in (0, 3, 6) for s, D in ((1, 20), (2, 4))],
                             dtype=object).tolist() and \
        np.array([(C.get_indexes_ff(t, 7, s, D) + [-1] * 7)[:7] for t in (0, 3, 6) for s, D in ((1, 20), (2, 4))])
    tl = sorted(range(7), key=lambda i: abs(i - 3))
    res['idx_ip'] = np.array([(C.get_indexes_ip(t, tl, 1, 20) + [-1] * 7)[:7] for t in tl])
    save('g12_inpaint', **res)


def g13_cpn_align():
    """CPN.align with its aligner networks (A_Encoder + A_Regressor, model_cpn.py:31-91, 121-174)."""
    model = mt.CPN()
    missing, unexpected = model.load_state_dict(S.cpn_state_dict(), strict=False)
    assert not unexpected and all(not k.startswith('A_') for k in missing), (missing, unexpected)
    model.eval()
    b, r, h, w = 2, 2, 64, 96
    x_t = S.uniform('g13.xt', (b, 3, h, w))
    m_t = (S.uniform('g13.mt', (b, 1, h, w)) > 0.8).float()
    x_r = S.uniform('g13.xr', (b, 3, r, h, w))
    m_r = (S.uniform('g13.mr', (b, 1, r, h, w)) > 0.8).float()
    with torch.no_grad():
        ft = model.A_Encoder(x_t, m_t)
        fr = model.A_Encoder(x_r.transpose(1, 2).reshape(-1, 3, h, w), m_r.transpose(1, 2).reshape(-1, 1, h, w))
        theta = model.A_Regressor(ft.unsqueeze(1).repeat(1, r, 1, 1, 1).reshape(-1, *ft.shape[1:]), fr)
        xa, va, vm = model.align(x_t, m_t, x_r, m_r)
    save('g13_cpn_align', feat_t=ft[:, ::16, ::4, ::4], theta=theta, x_aligned=xa, v_aligned=va, v_maps=vm)


def movement_matrices(n, seed_name):
    """n explicit 3x3 affine matrices (pixels; translation, anisotropic scale, rotation), identity in the
    middle: the form MovementsUtils.random_affine produces (utils.py:243-260)."""
    mats = []
    u = S.uniform(seed_name, (n, 5), -1, 1)
    for i in range(n):
        tx, ty, rot = float(u[i, 0]) * 12, float(u[i, 1]) * 12, float(u[i, 4]) * 0.1
        sx, sy = 1 + float(u[i, 2]) * 0.05, 1 + float(u[i, 3]) * 0.05
        mats.append(torch.tensor([[sx * np.cos(rot), -sy * np.sin(rot), tx],
                                  [sx * np.sin(rot), sy * np.cos(rot), ty], [0.0, 0.0, 1.0]], dtype=torch.float32))
    mats[n // 2] = torch.eye(3)
    return mats


def g14_movement():
    """MovementsUtils.simulate_movement with explicit matrices (utils.py:262-318).  The method itself
    cannot run in this image (its affine_inverse, utils.py:345, hands a Tensor to np.linalg.inv and then to
    torch.from_numpy: a numpy-2 / torch-2.10 skew, an ordinary TypeError), so the golden chains the
    reference's own helpers stack_transformations / affine2theta (utils.py:347-401) with the two torch calls
    the method makes (utils.py:308-318); the per-matrix inverses come from torch.inverse."""
    import torch.nn.functional as F
    M = mt.MovementsUtils
    x = S.uniform('g14.x', (3, 48, 64))
    n, (c, h, w) = 5, x.shape
    mats = torch.stack(movement_matrices(n, 'g14.m'))
    stacked = M.stack_transformations(mats, t=n // 2)
    stacked_inv = M.stack_transformations(torch.stack([torch.inverse(a) for a in mats]), t=n // 2)
    theta = torch.stack([M.affine2theta(a, h, w) for a in stacked])
    theta_inv = torch.stack([M.affine2theta(a, h, w) for a in stacked_inv])
    flow = F.affine_grid(theta, [n, c, h, w], align_corners=True)
    flow_inv = F.affine_grid(theta_inv, [n, c, h, w], align_corners=True)
    y = F.grid_sample(x.unsqueeze(0).repeat(n, 1, 1, 1), flow, align_corners=True).permute(1, 0, 2, 3)
    save('g14_movement', y=y, flow_inv=flow_inv, stacked=stacked, theta=theta)


ALL = [g1_corr, g2_warp, g3_resize, g4_modules, g5_dfpn_fwd, g6_dfpn_loss,
       g7_rrdbnet, g8_chn_loss, g9_grads, g10_vgg, g11_cpn_warp, g12_inpaint, g13_cpn_align, g14_movement]

if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count())
    want = set(sys.argv[1:])
    for fn in ALL:
        if want and fn.__name__ not in want and fn.__name__.split('_')[0] not in want:
            continue
        t0 = time.time()
        fn()
        print(f'  {fn.__name__} {time.time() - t0:.1f}s', flush=True)
