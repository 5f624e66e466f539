"""CPU oracle of the rows either side of the training step.  TEST
INFRASTRUCTURE ONLY: imported by tests/ (never by master_thesis_amd/).

PARITY UNPINNED.  The arithmetic of these reference functions lives in third-party packages that
are absent from this image and from /root/reference, so the reference itself cannot be run on them
and it holds no fixture for them:

* ``MeasuresUtils.psnr`` / ``ssim`` (master_thesis/utils.py:407-448) call ``scikit-image==0.18.3``
  (requirements.txt:45) ``skimage.metrics.peak_signal_noise_ratio`` / ``structural_similarity``;
* ``TransformsUtils.dilate`` (utils.py:618-640) calls ``opencv-python==4.5.4.58``
  (requirements.txt:29) ``cv2.dilate`` with ``cv2.getStructuringElement(cv2.MORPH_CROSS, size)``.

What follows restates the published algorithms of those pinned versions (float64 numpy; the
window filter is ``scipy.ndimage.uniform_filter``, the very function skimage calls), anchored on
the reference's call sites (argument order, ``multichannel=True``, float32 images in [0, 1], the
NaN -> 100 / drop-inf rule of ``psnr``, the ``(1,F,H,W)`` mask layout of ``dilate``)."""
import math

import numpy as np
from scipy.ndimage import uniform_filter

_DTYPE_RANGE_F32 = (-1.0, 1.0)        # skimage.util.dtype.dtype_range[np.float32]


def skimage_psnr(image_true, image_test):
    """skimage 0.18.3 metrics/simple_metrics.py peak_signal_noise_ratio(image_true, image_test),
    data_range=None, float32 inputs."""
    image_true, image_test = np.asarray(image_true, np.float32), np.asarray(image_test, np.float32)
    dmin, dmax = _DTYPE_RANGE_F32
    true_min, true_max = float(image_true.min()), float(image_true.max())
    if true_max > dmax or true_min < dmin:
        raise ValueError('im_true has intensity values outside the range expected for its data type.')
    data_range = dmax if true_min >= 0 else dmax - dmin
    err = np.mean((image_true - image_test) ** 2, dtype=np.float64)       # float32 differences, float64 mean
    with np.errstate(divide='ignore'):
        return 10 * np.log10((data_range ** 2) / err)


def psnr(y_hat, y):
    """MeasuresUtils.psnr (utils.py:407-429): note the argument swap at :418-420 -- the ground truth
    ``y`` is skimage's image_true.  y_hat, y: (B,C,H,W) arrays."""
    items = [skimage_psnr(y[f], y_hat[f]) for f in range(y.shape[0])]
    items = [100 if math.isnan(i) else i for i in items]
    return np.mean([i for i in items if not np.isinf(i) and not np.isnan(i)])


def skimage_ssim_plane(im1, im2, win_size=7, K1=0.01, K2=0.03):
    """skimage 0.18.3 metrics/_structural_similarity.py structural_similarity on ONE channel of a
    float32 image: uniform window, use_sample_covariance=True, data_range = dmax - dmin = 2."""
    data_range = _DTYPE_RANGE_F32[1] - _DTYPE_RANGE_F32[0]
    im1, im2 = im1.astype(np.float64), im2.astype(np.float64)
    NP = win_size ** im1.ndim
    cov_norm = NP / (NP - 1)
    ux, uy = uniform_filter(im1, size=win_size), uniform_filter(im2, size=win_size)
    uxx, uyy, uxy = (uniform_filter(im1 * im1, size=win_size), uniform_filter(im2 * im2, size=win_size),
                     uniform_filter(im1 * im2, size=win_size))
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean()


def ssim(y_hat, y):
    """MeasuresUtils.ssim (utils.py:431-448): per image, multichannel=True = the mean over channels
    of the per-channel SSIM; mean over images."""
    items = []
    for f in range(y.shape[0]):
        items.append(np.mean([skimage_ssim_plane(np.asarray(y[f, c], np.float32), np.asarray(y_hat[f, c], np.float32))
                              for c in range(y.shape[1])]))
    return np.mean(items)


def cross_element(filter_size):
    """cv2.getStructuringElement(cv2.MORPH_CROSS, (w, h)): the anchor row and the anchor column."""
    kw, kh = filter_size
    el = np.zeros((kh, kw), bool)
    el[kh // 2, :] = True
    el[:, kw // 2] = True
    return el


def dilate(images, filter_size, iterations):
    """TransformsUtils.dilate (utils.py:618-640): images (1,F,H,W); cv2.dilate, default border
    (constant, never the maximum), the element applied ``iterations`` times."""
    images = np.asarray(images, np.float32)
    el = cross_element(filter_size)
    kh, kw = el.shape
    out = np.empty_like(images)
    for f in range(images.shape[1]):
        cur = images[0, f]
        h, w = cur.shape
        for _ in range(iterations):
            p = np.full((h + kh - 1, w + kw - 1), -np.inf, np.float32)
            p[kh // 2: kh // 2 + h, kw // 2: kw // 2 + w] = cur
            nxt = np.full_like(cur, -np.inf)
            for dy in range(kh):
                for dx in range(kw):
                    if el[dy, dx]:
                        nxt = np.maximum(nxt, p[dy: dy + h, dx: dx + w])
            cur = nxt
        out[0, f] = cur
    return out


def masked_fill(m, y, fill=(0.485, 0.456, 0.406)):
    """dataset.py:168-169: x = (1 - m) * y + m * FILL_COLOR;  m (1,F,H,W), y (3,F,H,W)."""
    fill = np.asarray(fill, np.float32).reshape(3, 1, 1, 1)
    return (1 - m) * y + m * fill
