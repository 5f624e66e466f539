"""Conditioning of 1-D Winograd F(2,5) under the engine's arithmetic (2-term fp16 split of the TRANSFORMED operands,
exact products, fp32 accumulation and fp32 transforms) against the direct form in the same arithmetic, both
measured against fp64.  Candidate point sets for the 6-point interpolation."""
import itertools
import numpy as np
from fractions import Fraction as Fr


def cook_toom(m, r, pts):
    """A^T (m x n), G (n x r), B^T (n x n) for F(m, r) with n = m + r - 1 points (last = infinity)."""
    n = m + r - 1
    assert len(pts) == n - 1
    P = [Fr(p) for p in pts]
    # A^T[i][j] = p_j^i (j < n-1); infinity column = [0..0,1]
    AT = [[(P[j] ** i) for j in range(n - 1)] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    # G[j][k] = p_j^k / prod_{l != j}(p_j - p_l); infinity row = [0..0,1]
    G = []
    for j in range(n - 1):
        den = Fr(1)
        for l in range(n - 1):
            if l != j:
                den *= (P[j] - P[l])
        G.append([P[j] ** k / den for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    # B^T from the polynomial identities: row j (finite) = coefficients of prod_{l != j}(x - p_l) ; row inf = coefficients of prod_l (x - p_l)
    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for k, y in enumerate(b):
                out[i + k] += x * y
        return out
    BT = []
    for j in range(n - 1):
        poly = [Fr(1)]
        for l in range(n - 1):
            if l != j:
                poly = polymul(poly, [-P[l], Fr(1)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))
    poly = [Fr(1)]
    for l in range(n - 1):
        poly = polymul(poly, [-P[l], Fr(1)])
    BT.append(poly)
    f = lambda M: np.array([[float(v) for v in row] for row in M])
    return f(AT), f(G), f(BT)


def split22(v):
    """hi + lo of the 2-term fp16 split (values pre-scaled into fp16's range)."""
    v = v.astype(np.float32)
    hi = v.astype(np.float16).astype(np.float32)
    lo = (v - hi).astype(np.float16).astype(np.float32)
    return (hi.astype(np.float64) + lo.astype(np.float64))


def run(pts, C=128, trials=200, seed=0, scale_rows=None):
    AT, G, BT = cook_toom(2, 5, pts)
    if scale_rows is not None:           # move a diagonal scaling from B^T into G (any D: (D^-1 G) . (D B^T))
        D = np.array(scale_rows, dtype=np.float64)
        BT = BT * D[:, None]
        G = G / D[:, None]
    rng = np.random.default_rng(seed)
    e_w, e_d = [], []
    for _ in range(trials):
        d = rng.standard_normal((C, 5, 6))            # (channel, kernel row, 6 inputs)
        g = rng.standard_normal((C, 5, 5)) * 0.05
        ref = np.zeros(2)
        for j in range(2):
            ref[j] = sum((d[:, :, j:j + 5] * g).sum() for _ in [0])
        # Winograd in engine arithmetic: transforms in fp32, operands split to 22 bits, products exact, fp32 accumulate
        V = np.einsum('ij,ckj->cki', BT.astype(np.float32), d.astype(np.float32)).astype(np.float32)
        U = np.einsum('ij,ckj->cki', G.astype(np.float32), g.astype(np.float32)).astype(np.float32)
        sv, su = 2.0 ** (14 - np.ceil(np.log2(np.abs(V).max()))), 2.0 ** (14 - np.ceil(np.log2(np.abs(U).max())))
        Vs, Us = split22(V * sv) / sv, split22(U * su) / su
        m = np.zeros(6, dtype=np.float32)
        for c in range(C):                               # fp32 accumulation chain over (channel, kernel row)
            for k in range(5):
                m = (m + (Us[c, k] * Vs[c, k]).astype(np.float32)).astype(np.float32)
        y = (AT.astype(np.float32) @ m).astype(np.float32)
        e_w.append(np.abs(y - ref).max() / np.abs(ref).max())
        # direct form, same arithmetic
        sd, sg = 2.0 ** (14 - np.ceil(np.log2(np.abs(d).max()))), 2.0 ** (14 - np.ceil(np.log2(np.abs(g).max())))
        ds, gs = split22(d * sd) / sd, split22(g * sg) / sg
        yd = np.zeros(2, dtype=np.float32)
        for c in range(C):
            for k in range(5):
                for t in range(5):
                    yd = (yd + np.array([ds[c, k, t] * gs[c, k, t], ds[c, k, t + 1] * gs[c, k, t]]).astype(np.float32)).astype(np.float32)
        e_d.append(np.abs(yd - ref).max() / np.abs(ref).max())
    return (np.median(e_w), np.max(e_w), np.median(e_d), np.max(e_d), np.abs(BT).sum(1).max(), np.abs(G).sum(1).max(), np.abs(AT).sum(1).max())


if __name__ == '__main__':
    for pts in ([0, 1, -1, 2, -2], [0, 1, -1, Fr(1, 2), -Fr(1, 2)], [0, 1, -1, 2, -Fr(1, 2)], [0, 1, -1, Fr(1, 2), -2],
                [0, Fr(1, 2), -Fr(1, 2), 1, -2], [0, 1, -1, Fr(3, 2), -Fr(3, 2)]):
        r = run(pts, trials=60)
        print('points', [str(p) for p in pts], 'wino median %.2e max %.2e | direct median %.2e max %.2e | |B^T|row %.1f |G|row %.2f |A^T|row %.1f' % r, flush=True)
