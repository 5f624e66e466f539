"""Conditioning of 1-D Winograd F(4,3) (6 points) in the engine's arithmetic vs F(2,3) and the direct form, against fp64.
Layer-like statistics: many outputs, error normalised by max|y| (as tests/test_gpu_kernels.py does)."""
import numpy as np
from fractions import Fraction as Fr
import sys
sys.path.insert(0, 'scratch')
from r3_wino25_cond import cook_toom, split22


def run(m, r, pts, C=128, npos=256, seed=0, scale_bt=1.0):
    AT, G, BT = cook_toom(m, r, pts)
    n = m + r - 1
    rng = np.random.default_rng(seed)
    d = rng.uniform(-1, 1, (npos, C, r, n))          # (position, channel, kernel row, n inputs)
    g = rng.uniform(-0.1, 0.1, (C, r, r))
    ref = np.zeros((npos, m))
    for j in range(m):
        ref[:, j] = np.einsum('pckt,ckt->p', d[:, :, :, j:j + r], g)
    V = np.einsum('ij,pckj->pcki', BT.astype(np.float32), d.astype(np.float32)).astype(np.float32)
    U = np.einsum('ij,ckj->cki', G.astype(np.float32), g.astype(np.float32)).astype(np.float32)
    sv, su = 2.0 ** (14 - np.ceil(np.log2(np.abs(V).max()))), 2.0 ** (14 - np.ceil(np.log2(np.abs(U).max())))
    Vs, Us = split22(V * sv) / sv, split22(U * su) / su
    # fp32 accumulation over (channel, kernel row) in chunks of 16 (an MFMA k-step), sequential over the chunks
    M = np.zeros((npos, n), dtype=np.float32)
    prod = (Us[None] * Vs)                             # (p, c, k, n) float64 exact products
    pr = prod.transpose(0, 2, 1, 3).reshape(npos, r * C, n)
    for k0 in range(0, r * C, 16):
        M = (M + pr[:, k0:k0 + 16].sum(1).astype(np.float32)).astype(np.float32)
    y = (M.astype(np.float32) @ AT.astype(np.float32).T).astype(np.float32)
    return float(np.abs(y - ref).max() / np.abs(ref).max())


def direct(r, C=128, npos=256, seed=0, m=2):
    rng = np.random.default_rng(seed)
    n = m + r - 1
    d = rng.uniform(-1, 1, (npos, C, r, n))
    g = rng.uniform(-0.1, 0.1, (C, r, r))
    ref = np.zeros((npos, m)); y = np.zeros((npos, m), dtype=np.float32)
    sd, sg = 2.0 ** 14, 2.0 ** (14 - np.ceil(np.log2(0.1)))
    ds, gs = split22(d * sd) / sd, split22(g * sg) / sg
    for j in range(m):
        ref[:, j] = np.einsum('pckt,ckt->p', d[:, :, :, j:j + r], g)
        pr = (ds[:, :, :, j:j + r] * gs[None]).transpose(0, 2, 3, 1).reshape(npos, -1)
        acc = np.zeros(npos, dtype=np.float32)
        for k0 in range(0, pr.shape[1], 16):
            acc = (acc + pr[:, k0:k0 + 16].sum(1).astype(np.float32)).astype(np.float32)
        y[:, j] = acc
    return float(np.abs(y - ref).max() / np.abs(ref).max())


if __name__ == '__main__':
    print('direct 3-tap   %.2e' % direct(3))
    print('F(2,3) 0,1,-1  %.2e' % run(2, 3, [0, 1, -1]))
    for pts in ([0, 1, -1, 2, -2], [0, 1, -1, Fr(1, 2), -Fr(1, 2)], [0, 1, -1, 2, -Fr(1, 2)], [0, Fr(1, 2), -Fr(1, 2), 1, -2],
                [0, 1, -1, Fr(1, 2), -2], [0, Fr(1, 2), -Fr(1, 2), 2, -2], [0, 1, -1, Fr(3, 2), -Fr(3, 2)]):
        print('F(4,3)', [str(p) for p in pts], '%.2e' % run(4, 3, pts), '  (256 ch: %.2e)' % run(4, 3, pts, C=256))
    print('F(2,5) 0,+-1,+-2 %.2e' % run(2, 5, [0, 1, -1, 2, -2]))
