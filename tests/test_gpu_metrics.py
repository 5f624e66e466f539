"""HIP kernels either side of the training step against the CPU oracle
(oracle/ref_metrics.py: restated skimage 0.18.3 / cv2 algorithms, "parity unpinned")."""
import numpy as np
import pytest
import torch

import seeded as S
from oracle import ref_metrics as R

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def mt():
    import master_thesis_amd as mt
    mt._lib.load()
    return mt


@pytest.mark.parametrize('shape,size,it', [((1, 5, 64, 64), (3, 3), 4), ((1, 3, 50, 37), (3, 3), 4), ((1, 2, 33, 70), (5, 3), 2),
                                            ((1, 2, 40, 40), (3, 7), 1), ((1, 1, 16, 16), (3, 3), 0), ((1, 2, 256, 256), (3, 3), 4)])
def test_dilate_bit_exact(mt, shape, size, it):
    m = (S.uniform(f'dil.{shape}', shape) > 0.97).float()
    got = mt.TransformsUtils.dilate(m.to(DEV), size, it).cpu().numpy()
    assert np.array_equal(got, R.dilate(m.numpy(), size, it))


def test_dilate_fill_and_batched(mt):
    b, f, h, w = 3, 4, 45, 52
    m = (S.uniform('df.m', (b, 1, f, h, w)) > 0.98).float()
    y = S.uniform('df.y', (b, 3, f, h, w))
    md, x = mt.TransformsUtils.dilate_fill(m.to(DEV), y.to(DEV), (3, 3), 4)
    for i in range(b):
        ref_m = R.dilate(m[i].numpy(), (3, 3), 4)
        assert np.array_equal(md[i].cpu().numpy(), ref_m)
        np.testing.assert_allclose(x[i].cpu().numpy(), R.masked_fill(ref_m, y[i].numpy()), rtol=0, atol=1e-7)
    # the (1,F,H,W) / (3,F,H,W) form of the reference's call site
    md1, x1 = mt.TransformsUtils.dilate_fill(m[0].to(DEV), y[0].to(DEV), (3, 3), 4)
    assert torch.equal(md1, md[0]) and torch.equal(x1, x[0])


def test_make_batch_on_device_matches_host(mt):
    from master_thesis_amd.synthetic import make_batch
    (xg, mg), yg, ig = make_batch(2, 3, 64, 64, DEV, seed=7)
    (xc, mc), yc, ic = make_batch(2, 3, 64, 64, 'cpu', seed=7)
    assert torch.equal(mg.cpu(), mc) and torch.equal(yg.cpu(), yc)
    np.testing.assert_allclose(xg.cpu().numpy(), xc.numpy(), rtol=0, atol=1e-7)


@pytest.mark.parametrize('shape', [(2, 3, 64, 64), (3, 3, 50, 37), (1, 3, 256, 256), (2, 1, 7, 9)])
def test_psnr_ssim_vs_oracle(mt, shape):
    y = S.uniform(f'ms.y{shape}', shape)
    yh = (y + 0.08 * (S.uniform(f'ms.n{shape}', shape) - 0.5)).clamp(0, 1)
    p = mt.MeasuresUtils.psnr(yh.to(DEV), y.to(DEV))
    s = mt.MeasuresUtils.ssim(yh.to(DEV), y.to(DEV))
    assert p.dtype == torch.float64 and p.dim() == 0 and p.is_cuda
    np.testing.assert_allclose(float(p), R.psnr(yh.numpy(), y.numpy()), rtol=1e-10)
    np.testing.assert_allclose(float(s), R.ssim(yh.numpy(), y.numpy()), rtol=1e-9)


def test_psnr_edge_cases(mt):
    y = S.uniform('pe.y', (3, 3, 32, 32))
    yh = y.clone()
    yh[1] = (yh[1] + 0.1).clamp(0, 1)                     # images 0 and 2 identical -> inf -> dropped
    np.testing.assert_allclose(float(mt.MeasuresUtils.psnr(yh.to(DEV), y.to(DEV))), R.psnr(yh.numpy(), y.numpy()), rtol=1e-10)
    yn = y * 2 - 1                                        # negative ground truth -> data_range 2
    yhn = (yn + 0.05).clamp(-1, 1)
    np.testing.assert_allclose(float(mt.MeasuresUtils.psnr(yhn.to(DEV), yn.to(DEV))), R.psnr(yhn.numpy(), yn.numpy()), rtol=1e-10)
    assert float(mt.MeasuresUtils.ssim(y.to(DEV), y.to(DEV))) == pytest.approx(1.0, abs=1e-12)
    assert torch.isnan(mt.MeasuresUtils.psnr(y.to(DEV), y.to(DEV)))      # every item infinite: mean of nothing (as np.mean([]))


def test_chn_validation_step_logs_metrics(mt):
    kw = dict(lr=1e-4, lr_scheduler_step_size=50, lr_scheduler_gamma=0.5)
    vgg = mt.VGGFeatures(device=DEV)
    chn = mt.CHN(vgg, None, mt.DFPN(vgg, **kw), **kw)
    chn.load_state_dict(S.add_buffers(S.seeded_state_dict(S.chn_shapes()), ('', 'model_aligner.')))
    chn = chn.to(DEV)
    (x, m), y, info = S.synthetic_batch(1, 3, 64, 64, seed=4)
    batch = ((x.to(DEV), m.to(DEV)), y.to(DEV), tuple(i.to(DEV) if torch.is_tensor(i) else i for i in info))
    loss = chn.validation_step(batch, 0)
    assert torch.isfinite(loss)
    for k in ('measures_psnr_validation', 'measures_ssim_validation', 'loss_validation'):
        assert k in chn.logged and np.isfinite(float(chn.logged[k])), k
    assert 0 < float(chn.logged['measures_ssim_validation']) <= 1
