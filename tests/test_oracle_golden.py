"""Pin the CPU oracle (oracle/ref_torch.py) against the golden vectors that
tests/golden/make_golden.py captured from the reference itself."""
import numpy as np
import pytest
import torch

import seeded as S
from oracle import ref_torch as O

RTOL = 1e-4  # oracle vs reference: same torch ops, only association differs


def close(a, b, rtol=RTOL, atol=1e-5):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.fixture(scope='module')
def dfpn_sd():
    return S.add_buffers(S.seeded_state_dict(S.dfpn_shapes()))


@pytest.fixture(scope='module')
def chn_sd():
    return S.add_buffers(S.seeded_state_dict(S.chn_shapes()), ('', 'model_aligner.'))


def test_g1_correlation(golden):
    g = golden('g1_corr')
    for tag, (b, r, c) in {'small': (2, 2, 64), 'c512': (1, 1, 512)}.items():
        ft = S.uniform(f'g1.ft.{tag}', (b, c, 16, 16), -1, 2)
        fr = S.uniform(f'g1.fr.{tag}', (b, c, r, 16, 16), -1, 2)
        vt = (S.uniform(f'g1.vt.{tag}', (b, 1, 16, 16)) > 0.3).float()
        vr = (S.uniform(f'g1.vr.{tag}', (b, 1, r, 16, 16)) > 0.3).float()
        close(O.correlation_masked_4d(ft, vt, fr, vr), g[f'{tag}.masked'])
        close(O.correlation_masked_4d(ft, None, fr, None), g[f'{tag}.unmasked'])
        close(O.softmax_3d(O.correlation_masked_4d(ft, vt, fr, vr) * 20.0), g[f'{tag}.softmax'])


def test_g2_warp(golden):
    from make_inputs import warp_grid
    g = golden('g2_warp')
    for s in (16, 64):
        x = S.uniform(f'g2.x.{s}', (2, 3, 2, s, s))
        v = (S.uniform(f'g2.v.{s}', (2, 1, 2, s, s)) > 0.4).float()
        grid = warp_grid(2, 2, s, s, f'g2.g.{s}')
        xa, va = O.align_set(x, v, grid)
        close(xa, g[f'x.{s}'])
        assert np.array_equal(va.numpy(), g[f'v.{s}'])
        g2 = grid.clone().requires_grad_(True)
        wgt = S.uniform(f'g2.w.{s}', (2, 3, 2, s, s), -1, 1)
        (O.align_set(x, v, g2)[0] * wgt).sum().backward()
        close(g2.grad, g[f'dgrid.{s}'], atol=1e-4)


def test_g3_resize(golden):
    g = golden('g3_resize')
    fl = S.affine_flow(2, 2, 16, 16, 0, 'g3.f16')
    close(O.resize_flow(fl, (64, 64), 'bilinear'), g['flow.16to64.bilinear'])
    fl64 = S.affine_flow(1, 2, 64, 64, 0, 'g3.f64')
    close(O.resize_flow(fl64, (256, 256), 'bilinear')[:, :, ::5, ::7], g['flow.64to256.bilinear'])
    assert np.array_equal(O.resize_flow(fl64, (16, 16)).numpy(), g['flow.64to16.nearest'])
    fl48 = S.affine_flow(1, 1, 48, 80, 0, 'g3.f48')
    close(O.resize_flow(fl48, (40, 56), 'bilinear'), g['flow.48x80to40x56.bilinear'])
    assert np.array_equal(O.resize_flow(fl48, (16, 16)).numpy(), g['flow.48x80to16.nearest'])
    x = S.uniform('g3.x', (2, 3, 2, 64, 64))
    v = (S.uniform('g3.v', (2, 1, 2, 64, 64)) > 0.4).float()
    y = S.uniform('g3.y', (2, 3, 2, 64, 64))
    xn, vn, yn = O.resize_set(x, v, y, 16)
    close(xn, g['set16.x']); close(yn, g['set16.y'])
    assert np.array_equal(vn.numpy(), g['set16.v'])
    a, b, c, d = O.resize_set_bis(x[:, :, 0], v[:, :, 0], x, v, (32, 32))
    close(a, g['bis32.xt']); close(c, g['bis32.xr'])
    assert np.array_equal(b.numpy(), g['bis32.mt']) and np.array_equal(d.numpy(), g['bis32.mr'])
    x96 = S.uniform('g3.x96', (1, 3, 1, 96, 72))
    v96 = (S.uniform('g3.v96', (1, 1, 1, 96, 72)) > 0.4).float()
    a, b, _, _ = O.resize_set_bis(x96[:, :, 0], v96[:, :, 0], x96, v96, (64, 64))
    close(a, g['bis64.xt'])
    assert np.array_equal(b.numpy(), g['bis64.mt'])


def test_g4_modules(golden, dfpn_sd):
    g = golden('g4_modules')
    with torch.no_grad():
        corr = S.uniform('g4.corr', (1, 1, 16, 16, 16, 16), -1, 1)
        close(O.separable_conv4d(dfpn_sd, 'corr.conv.', corr), g['sepconv4d'], atol=1e-4)
        vol = S.uniform('g4.vol', (1, 2, 16, 16, 16, 16), -1, 1)
        close(O.corr_mixer(dfpn_sd, 'corr_mixer.', vol), g['mixer'], atol=1e-4)
        xt = S.uniform('g4.xt', (1, 3, 64, 64), -2, 2)
        mt_ = (S.uniform('g4.mt', (1, 1, 64, 64)) > 0.7).float()
        xr = S.uniform('g4.xr', (1, 3, 2, 64, 64), -2, 2)
        mr = (S.uniform('g4.mr', (1, 1, 2, 64, 64)) > 0.7).float()
        fp = S.affine_flow(1, 2, 64, 64, 0, 'g4.fp')
        close(O.flow_estimator(dfpn_sd, 'flow_64.', xt, mt_, xr, mr, fp), g['flow_est'], atol=1e-4)


def test_g5_dfpn_forward_cfg1(golden, dfpn_sd):
    g = golden('g5_dfpn_fwd')
    (x, m), y, info = S.synthetic_batch(1, 2, 128, 128, seed=5)
    t, r = O.get_indexes(2)
    with torch.no_grad():
        corr, f16, f64, f = O.dfpn_forward(dfpn_sd, x[:, :, t], m[:, :, t], x[:, :, r], m[:, :, r])
        xa, va, vmap = O.dfpn_align(dfpn_sd, x[:, :, t], m[:, :, t], x[:, :, r], m[:, :, r])
    close(corr, g['corr'], atol=1e-4); close(f16, g['flow_16'], atol=1e-4)
    close(f64, g['flow_64'], atol=1e-4); close(f, g['flow'], atol=1e-4)
    close(xa, g['x_aligned'], atol=1e-3)
    assert (va.numpy() != g['v_aligned']).mean() < 1e-3
    assert (vmap.numpy() != g['v_map']).mean() < 1e-3


def test_g6_dfpn_loss(golden, dfpn_sd):
    g = golden('g6_dfpn_loss')
    for tag, use in {'mixed': [True, False], 'all': [True, True], 'none': [False, False]}.items():
        batch = S.synthetic_batch(2, 3, 64, 64, seed=6, flows_use=use)
        with torch.no_grad():
            loss, items = O.dfpn_training_step(dfpn_sd, batch)
        close(loss.reshape(-1), g[f'{tag}.loss'].reshape(-1), rtol=1e-3)
        close(torch.stack([i.reshape(()) for i in items]), g[f'{tag}.items'], rtol=1e-3)


def test_g7_rrdbnet(golden, chn_sd):
    g = golden('g7_rrdbnet')
    with torch.no_grad():
        y = O.rrdbnet(chn_sd, 'nn.', S.uniform('g7.x', (1, 9, 32, 32), -1, 1))
        rdb = O.residual_dense_block(chn_sd, 'nn.rrdb_trunk.0.rdb1.', S.uniform('g7.rdb', (1, 64, 16, 16), -1, 1))
    close(y, g['y'], rtol=1e-3, atol=1e-4); close(rdb, g['rdb'])


def test_g8_chn_loss(golden, chn_sd):
    g = golden('g8_chn_loss')
    for r in (1, 2):
        b, h = 2, 64
        yt = S.uniform(f'g8.yt.{r}', (b, 3, h, h))
        vt = (S.uniform(f'g8.vt.{r}', (b, 1, h, h)) > 0.3).float()
        yh = S.uniform(f'g8.yh.{r}', (b, 3, r, h, h))
        vm = (S.uniform(f'g8.vm.{r}', (b, 1, r, h, h)) > 0.5).float() * (1 - vt).unsqueeze(2)
        yc = vt.unsqueeze(2) * yt.unsqueeze(2) + (1 - vt).unsqueeze(2) * yh
        with torch.no_grad():
            loss, items = O.chn_compute_loss(chn_sd, yt, vt, yh, yc, vm)
        close(loss, g[f'r{r}.loss'], rtol=1e-4)
        close(torch.stack(items), g[f'r{r}.items'], rtol=1e-4)


def _grad_check(sd, names, loss, g, tag):
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    for n, gr in zip(names, grads):
        ref = float(g[f'{tag}.norm/{n}'])
        assert abs(float(gr.norm()) - ref) <= 2e-3 * ref + 1e-7, n
        flat = gr.reshape(-1)
        head = flat[:: max(1, flat.numel() // 64)][:64]
        close(head, g[f'{tag}.head/{n}'], rtol=2e-3, atol=2e-3 * ref / max(1.0, flat.numel() ** 0.5))


def test_g9_grads_dfpn(golden, dfpn_sd):
    g = golden('g9_grads')
    sd = dict(dfpn_sd)
    names = [k[len('dfpn.norm/'):] for k in g.files if k.startswith('dfpn.norm/')]
    for n in names:
        sd[n] = sd[n].clone().requires_grad_(True)
    loss, _ = O.dfpn_training_step(sd, S.synthetic_batch(2, 2, 64, 64, seed=9))
    close(loss.reshape(-1), g['dfpn.loss'].reshape(-1), rtol=1e-4)
    _grad_check(sd, names, loss, g, 'dfpn')


def test_g9_grads_chn(golden, chn_sd):
    g = golden('g9_grads')
    sd = dict(chn_sd)
    names = [k[len('chn.norm/'):] for k in g.files if k.startswith('chn.norm/')]
    assert len(names) == 2 * (5 + 300 + 5)
    for n in names:
        sd[n] = sd[n].clone().requires_grad_(True)
    loss, _ = O.chn_training_step(sd, S.synthetic_batch(1, 2, 64, 64, seed=10))
    close(loss.reshape(-1), g['chn.loss'].reshape(-1), rtol=1e-4)
    _grad_check(sd, names, loss, g, 'chn')


def test_g10_vgg(golden, dfpn_sd):
    g = golden('g10_vgg')
    x = S.uniform('g10.x', (2, 3, 64, 64))
    with torch.no_grad():
        taps = O.vgg_features(dfpn_sd, 'model_vgg.', x)
        raw = O.vgg_features(dfpn_sd, 'model_vgg.', x, False)
    for i in range(5):
        close(taps[i], g[f'tap{i}'], atol=1e-4); close(raw[i], g[f'raw{i}'], atol=1e-4)


def test_g11_cpn_warp(golden):
    g = golden('g11_cpn_warp')
    b, r, h, w = 1, 3, 32, 48
    x = S.uniform('g11.x', (b, 3, r, h, w))
    m = (S.uniform('g11.m', (b, 1, r, h, w)) > 0.6).float()
    m_t = (S.uniform('g11.mt', (b, 1, h, w)) > 0.6).float()
    xa, va, vm = O.cpn_warp(x, m, m_t, torch.from_numpy(g['theta']))
    close(xa, g['x_aligned'])
    assert np.array_equal(va.numpy(), g['v_aligned']) and np.array_equal(vm.numpy(), g['v_maps'])


def test_g12_inpainting_drivers(golden, chn_sd):
    """The frame-by-frame / inpaint-and-propagate / copy-and-propagate drivers and their index
    helpers against the reference's own runs (model_chn.py:87-254, 459-503)."""
    g = golden('g12_inpaint')
    for i, (t, (s, d)) in enumerate((t, sd) for t in (0, 3, 6) for sd in ((1, 20), (2, 4))):
        assert (O.get_indexes_ff(t, 7, s, d) + [-1] * 7)[:7] == list(g['idx_ff'][i])
    tl = sorted(range(7), key=lambda i: abs(i - 3))
    for i, t in enumerate(tl):
        assert (O.get_indexes_ip(t, tl, 1, 20) + [-1] * 7)[:7] == list(g['idx_ip'][i])
    (x, m), y, info = S.synthetic_batch(1, 4, 64, 64, seed=12)
    with torch.no_grad():
        close(O.inpaint_ff(chn_sd, x[0], m[0]), g['ff'], rtol=2e-3)
        close(O.inpaint_ip(chn_sd, x[0], m[0]), g['ip'], rtol=2e-3)
        close(O.inpaint_cp(chn_sd, x[0], m[0], N=2), g['cp'], rtol=2e-3)


def test_g13_cpn_aligner(golden):
    """A_Encoder / A_Regressor / CPN.align against the reference (model_cpn.py:31-91, 121-174)."""
    g = golden('g13_cpn_align')
    sd = S.cpn_state_dict()
    b, r, h, w = 2, 2, 64, 96
    x_t = S.uniform('g13.xt', (b, 3, h, w))
    m_t = (S.uniform('g13.mt', (b, 1, h, w)) > 0.8).float()
    x_r = S.uniform('g13.xr', (b, 3, r, h, w))
    m_r = (S.uniform('g13.mr', (b, 1, r, h, w)) > 0.8).float()
    with torch.no_grad():
        close(O.cpn_a_encoder(sd, x_t, m_t)[:, ::16, ::4, ::4], g['feat_t'], rtol=1e-3, atol=1e-4)
        xa, va, vm, theta = O.cpn_align(sd, x_t, m_t, x_r, m_r)
    close(theta, g['theta'], rtol=1e-3, atol=1e-5)
    close(xa, g['x_aligned'], rtol=1e-3, atol=1e-4)
    assert (va.numpy() != g['v_aligned']).mean() < 1e-3 and (vm.numpy() != g['v_maps']).mean() < 1e-3


def _movement_matrices(n, seed_name):
    mats = []
    u = S.uniform(seed_name, (n, 5), -1, 1)
    for i in range(n):
        tx, ty, rot = float(u[i, 0]) * 12, float(u[i, 1]) * 12, float(u[i, 4]) * 0.1
        sx, sy = 1 + float(u[i, 2]) * 0.05, 1 + float(u[i, 3]) * 0.05
        mats.append(torch.tensor([[sx * np.cos(rot), -sy * np.sin(rot), tx],
                                  [sx * np.sin(rot), sy * np.cos(rot), ty], [0.0, 0.0, 1.0]], dtype=torch.float32))
    mats[n // 2] = torch.eye(3)
    return torch.stack(mats)


def test_g14_simulate_movement(golden):
    """Synthetic camera movement (utils.py:262-401) against the reference's helpers."""
    g = golden('g14_movement')
    x = S.uniform('g14.x', (3, 48, 64))
    mats = _movement_matrices(5, 'g14.m')
    close(O.stack_transformations(mats, 2), g['stacked'], rtol=1e-5)
    y, flow_inv = O.simulate_movement(x, mats)
    close(y, g['y'], rtol=1e-4, atol=1e-5)
    close(flow_inv, g['flow_inv'], rtol=1e-4, atol=1e-5)
