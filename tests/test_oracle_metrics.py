"""CPU checks of oracle/ref_metrics.py: no reference fixture exists for these
rows (skimage / cv2 are absent: "parity unpinned"), so the restatement is held against independent
brute-force formulations and known answers."""
import numpy as np

from oracle import ref_metrics as R


def _brute_ssim(a, b, win=7, K1=0.01, K2=0.03, data_range=2.0):
    a, b = a.astype(np.float64), b.astype(np.float64)
    h, w = a.shape
    pad = win // 2
    vals = []
    for i in range(pad, h - pad):
        for j in range(pad, w - pad):
            x, y = a[i - pad:i + pad + 1, j - pad:j + pad + 1].ravel(), b[i - pad:i + pad + 1, j - pad:j + pad + 1].ravel()
            ux, uy = x.mean(), y.mean()
            vx, vy = x.var(ddof=1), y.var(ddof=1)                   # sample covariance: NP / (NP - 1)
            vxy = ((x - ux) * (y - uy)).sum() / (x.size - 1)
            C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
            vals.append(((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2)))
    return float(np.mean(vals))


def test_ssim_restatement_against_brute_force():
    rng = np.random.default_rng(3)
    a = rng.random((19, 23)).astype(np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape), 0, 1).astype(np.float32)
    np.testing.assert_allclose(R.skimage_ssim_plane(a, b), _brute_ssim(a, b), rtol=1e-10)
    assert abs(R.skimage_ssim_plane(a, a) - 1.0) < 1e-12
    y = rng.random((2, 3, 16, 16)).astype(np.float32)
    yh = np.clip(y + 0.05, 0, 1).astype(np.float32)
    want = np.mean([[_brute_ssim(y[f, c], yh[f, c]) for c in range(3)] for f in range(2)])
    np.testing.assert_allclose(R.ssim(yh, y), want, rtol=1e-10)


def test_psnr_known_answers():
    y = np.full((1, 3, 8, 8), 0.5, np.float32)
    np.testing.assert_allclose(R.psnr(y + 0.1, y), 10 * np.log10(1 / 0.01), rtol=1e-6)      # data_range 1
    yn = y.copy()
    yn[0, 0, 0, 0] = -0.25                                                               # min(y) < 0 -> data_range 2
    np.testing.assert_allclose(R.skimage_psnr(yn[0], yn[0] + 0.1), 10 * np.log10(4 / 0.01), rtol=1e-6)
    two = np.concatenate([y, y])
    np.testing.assert_allclose(R.psnr(np.concatenate([y, y + 0.1]), two), 20.0, rtol=1e-6)  # identical image: inf, dropped
    try:
        R.skimage_psnr(y[0] + 1.0, y[0])
        raise AssertionError('values outside [-1, 1] must raise')
    except ValueError:
        pass


def test_dilate_is_the_l1_ball_and_fill():
    m = np.zeros((1, 2, 21, 17), np.float32)
    m[0, 0, 10, 8] = 1
    m[0, 1, 0, 16] = 1                                                   # a corner: the border never contributes
    d = R.dilate(m, (3, 3), 4)
    yy, xx = np.mgrid[:21, :17]
    assert np.array_equal(d[0, 0], (np.abs(yy - 10) + np.abs(xx - 8) <= 4).astype(np.float32))
    assert np.array_equal(d[0, 1], (np.abs(yy - 0) + np.abs(xx - 16) <= 4).astype(np.float32))
    assert np.array_equal(R.dilate(m, (3, 3), 0), m)
    wide = R.dilate(m, (5, 3), 1)[0, 0]                                  # cv2 sizes are (width, height)
    assert wide[10, 6] == 1 and wide[10, 10] == 1 and wide[8, 8] == 0 and wide[9, 8] == 1 and wide.sum() == 7
    y = np.random.default_rng(0).random((3, 2, 21, 17)).astype(np.float32)
    x = R.masked_fill(d, y)
    assert np.array_equal(x[:, d[0] == 0], y[:, d[0] == 0])
    np.testing.assert_allclose(x[:, 0, 10, 8], [0.485, 0.456, 0.406], rtol=1e-6)
