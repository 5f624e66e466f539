// Implicit-GEMM convolution engine for gfx950 (MI355X): forward / backward-data
// (one kernel, two gather modes) and weight-gradient, on fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate, 64 FLOP/clk/SIMD).
//
// Replaces the F.conv2d / F.conv_transpose2d call-sites of the reference and
// their autograd backward (model_dfpn.py:596-605,638-660,686-711,
// model_chn.py:522-535,602-606, model_vgg.py:40-41).
//
// GEMM view (forward / backward-data):  C[pixel][o] = sum_k A[pixel][k] * Wt[o][k],
// k = (kh, kw, c) flattened, A gathered on the fly from an NHWC4 view.
// Block = 256 threads = 4 waves; block tile BM x BN = (WGM*TM*32) x (WGN*TN*32),
// BK = 32.  A and B tiles are staged global -> registers -> LDS (double
// buffered, one barrier per K step); every lane reads its MFMA operands with
// ds_read_b128 (lane half h owns k = 8q+4h .. 8q+4h+3, so MFMA j of a group of
// four consumes the k-pair {8q+j, 8q+4+j}); LDS rows are padded to 36 floats,
// which keeps the b128 reads conflict-free.
#include "common.h"
#include "conv_common.h"
#include "dispatch.h"

namespace {

template <int WGM, int WGN, int TM, int TN>
__global__ __launch_bounds__(256) void igemm_kernel(ConvK p) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int RA = BM / 32, RB = BN / 32;
  __shared__ __attribute__((aligned(16))) float sA[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) float sB[2][BN * LDK];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM, n0 = blockIdx.y * BN;
  if (p.zb > 1) {
    const int z = blockIdx.z;
    p.x += (long)(z / p.zdiv) * p.xz;
    p.wt += (long)z * p.wz;
    p.y += (long)z * p.yz;
  }

  // ---- per-thread staging coordinates -----------------------------------
  const int col4 = tid & 7;        // which float4 of the 32-wide k slice
  const int row0 = tid >> 3;       // 0..31
  long a_off[RA];
  int a_oy[RA], a_ox[RA];
  bool a_ok[RA];
  const int HWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = m0 + row0 + 32 * i;
    a_ok[i] = m < p.M;
    int mm = a_ok[i] ? m : 0;
    int n = mm / HWo;
    int r = mm - n * HWo;
    a_oy[i] = r / p.Wo;
    a_ox[i] = r - a_oy[i] * p.Wo;
    a_off[i] = (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;
  }
  // (kh, kw, c) of this thread's float4 inside the current k tile
  int kc = col4 * 4, kkh = 0, kkw = 0;
  while (kc >= p.C) { kc -= p.C; if (++kkw == p.KW) { kkw = 0; ++kkh; } }
  const int HL = p.H << p.ups, WL = p.W << p.ups;
  const int smask = p.stride - 1, sshift = p.stride >> 1;  // stride in {1,2}

  f32x4 ra[RA], rb[RB];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      bool ok = a_ok[i] && kkh < p.KH;
      int iy, ix;
      if (!p.transposed) {
        iy = a_oy[i] * p.stride - p.pad + kkh;
        ix = a_ox[i] * p.stride - p.pad + kkw;
        ok = ok && (unsigned)iy < (unsigned)HL && (unsigned)ix < (unsigned)WL;
      } else {
        int ty = a_oy[i] + p.pad - kkh, tx = a_ox[i] + p.pad - kkw;
        ok = ok && ty >= 0 && tx >= 0 && ((ty | tx) & smask) == 0;
        iy = ty >> sshift; ix = tx >> sshift;
        ok = ok && iy < HL && ix < WL;
      }
      iy >>= p.ups; ix >>= p.ups;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(p.x + a_off[i] + (long)iy * p.xsh + (long)ix * p.xsw + kc);
      ra[i] = v;
    }
    const int k = kt * BK + col4 * 4;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      int n = n0 + row0 + 32 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < p.Co && k < p.K) v = *reinterpret_cast<const f32x4*>(p.wt + (long)n * p.K + k);
      rb[i] = v;
    }
    // advance (kh, kw, c) by BK for the next tile
    kc += BK;
    while (kc >= p.C) { kc -= p.C; if (++kkw == p.KW) { kkw = 0; ++kkh; } }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; ++i)
      *reinterpret_cast<f32x4*>(&sA[buf][(row0 + 32 * i) * LDK + col4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < RB; ++i)
      *reinterpret_cast<f32x4*>(&sB[buf][(row0 + 32 * i) * LDK + col4 * 4]) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const float* A = &sA[buf][(wm * TM * 32 + li) * LDK + 4 * lh];
    const float* B = &sB[buf][(wn * TN * 32 + li) * LDK + 4 * lh];
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      f32x4 fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const f32x4*>(A + a * 32 * LDK + q * 8);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = *reinterpret_cast<const f32x4*>(B + b * 32 * LDK + q * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  igemm_epilogue<TM, TN>(p, acc, m0, n0, wm, wn, li, lh);
}

// ================================================================== split ==
// Split-bf16 variants.  Every fp32 operand is written as a sum of NS bf16
// terms, a = a_0 + a_1 (+ a_2), a_0 = bf16(a), a_1 = bf16(a - a_0), ... and the
// fp32 product is accumulated from the bf16 x bf16 partial products (each exact
// in fp32) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//   NS = 2 ("bf16x3"): a0b0 + a0b1 + a1b0          17-18 significant bits,
//                      ~4e-6 relative per layer, 16/3 = 5.3x the fp32-MFMA rate
//   NS = 3 ("bf16x6"): all i + j <= 2 (6 products)  >= 24 significant bits =
//                      fp32-equivalent, 16/6 = 2.7x the fp32-MFMA rate
// Weights are split once at pack time; activations are split while they are
// staged into LDS.  K per LDS stage: BKT = 32 (NS = 2) or 16 (NS = 3) so that
// both fit two blocks per CU.
// FMT = 1 ("f16x3", NS = 2): fp16 terms of operands scaled by per-tensor powers
// of two (see conv_common.h); the accumulators are unscaled before the epilogue.
template <int WGM, int WGN, int TM, int TN, int NS, bool PAR, int FMT = 0>
__global__ __launch_bounds__(256) void igemm_split_kernel(ConvK p) {
  constexpr int BKT = NS == 3 ? 16 : 32;            // k per LDS stage
  constexpr int LD = BKT + 8;                       // bf16 per LDS row (+16 B pad: conflict-free b128 reads)
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int FPR = BKT / 4, RPP = 256 / FPR;     // float4 per A row, A rows per pass
  constexpr int RA = BM / RPP;
  constexpr int CPR = BKT / 8;                      // 16-B chunks per B row
  constexpr int NCH = NS * BN * CPR, RB = (NCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) __bf16 sA[2][NS][BM * LD];
  __shared__ __attribute__((aligned(16))) __bf16 sB[2][NS][BN * LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM, n0 = blockIdx.y * BN;
  const float scA = FMT ? pow2_scale(*p.xamax) : 1.f;
  const float inv = FMT ? 1.f / (scA * pow2_scale(*p.wamax)) : 1.f;

  // PAR: a stride-2 transposed gather (conv backward-data / conv_transpose
  // forward) only touches taps with kh == (oy + pad) mod 2: instead of
  // multiplying 3/4 zeros, blockIdx.z enumerates the four output-parity
  // classes, each a dense stride-1 gather over its own tap subset
  //   iy = yy + (py + pad - kh0)/2 - jh,   kh = kh0 + 2*jh.
  PixMap pm = {p.Ho, p.Wo, p.M, 1, 0, 0};
  int kh0 = 0, kw0 = 0, nkh = p.KH, nkw = p.KW, dyc = 0, dxc = 0;
  if (PAR) {
    pm.py = blockIdx.z >> 1; pm.px = blockIdx.z & 1; pm.ostep = 2;
    pm.Hc = (p.Ho - pm.py + 1) >> 1; pm.Wc = (p.Wo - pm.px + 1) >> 1;
    pm.M = p.N * pm.Hc * pm.Wc;
    kh0 = (pm.py + p.pad) & 1; kw0 = (pm.px + p.pad) & 1;
    nkh = (p.KH - kh0 + 1) >> 1; nkw = (p.KW - kw0 + 1) >> 1;
    dyc = (pm.py + p.pad - kh0) >> 1; dxc = (pm.px + p.pad - kw0) >> 1;
    if (m0 >= pm.M) return;
  }
  const int Kc = nkh * nkw * p.C;

  const int col4 = tid % FPR, row0 = tid / FPR;
  long a_off[RA];
  int a_oy[RA], a_ox[RA];
  bool a_ok[RA];
  const int HWo = pm.Hc * pm.Wc;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = m0 + row0 + RPP * i;
    a_ok[i] = m < pm.M;
    int mm = a_ok[i] ? m : 0;
    int n = mm / HWo;
    int r = mm - n * HWo;
    a_oy[i] = r / pm.Wc;
    a_ox[i] = r - a_oy[i] * pm.Wc;
    a_off[i] = (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;
  }
  int kc = col4 * 4, kkh = 0, kkw = 0;
  while (kc >= p.C) { kc -= p.C; if (++kkw == nkw) { kkw = 0; ++kkh; } }
  const int HL = p.H << p.ups, WL = p.W << p.ups;
  const int smask = p.stride - 1, sshift = p.stride >> 1;
  const __bf16* wt16 = reinterpret_cast<const __bf16*>(p.wt);
  const long plane = (long)p.Co * p.Kp;
  // PAR: per-chunk (tap, channel) trackers for the weight tile, whose k' is a
  // compacted enumeration of the class's taps
  int bkc[RB], bjh[RB], bjw[RB];
  if (PAR) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int w = (tid + 256 * i) % (BN * CPR);
      bkc[i] = (w % CPR) * 8; bjh[i] = 0; bjw[i] = 0;
      while (bkc[i] >= p.C) { bkc[i] -= p.C; if (++bjw[i] == nkw) { bjw[i] = 0; ++bjh[i]; } }
    }
  }

  f32x4 ra[RA];
  bf16x8 rb[RB];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      bool ok = a_ok[i] && kkh < nkh;
      int iy, ix;
      if (PAR) {
        iy = a_oy[i] + dyc - kkh; ix = a_ox[i] + dxc - kkw;
        ok = ok && (unsigned)iy < (unsigned)HL && (unsigned)ix < (unsigned)WL;
      } else if (!p.transposed) {
        iy = a_oy[i] * p.stride - p.pad + kkh;
        ix = a_ox[i] * p.stride - p.pad + kkw;
        ok = ok && (unsigned)iy < (unsigned)HL && (unsigned)ix < (unsigned)WL;
      } else {
        int ty = a_oy[i] + p.pad - kkh, tx = a_ox[i] + p.pad - kkw;
        ok = ok && ty >= 0 && tx >= 0 && ((ty | tx) & smask) == 0;
        iy = ty >> sshift; ix = tx >> sshift;
        ok = ok && iy < HL && ix < WL;
      }
      iy >>= p.ups; ix >>= p.ups;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(p.x + a_off[i] + (long)iy * p.xsh + (long)ix * p.xsw + kc);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int c = tid + 256 * i;               // chunk id over [plane][BN rows][CPR x 16 B]
      const int pl = c / (BN * CPR), w = c % (BN * CPR), row = w / CPR, ch = w % CPR;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
      if (PAR) {
        if (c < NCH && n0 + row < p.Co && bjh[i] < nkh)
          v = *reinterpret_cast<const bf16x8*>(wt16 + pl * plane + (long)(n0 + row) * p.Kp +
                                               (long)((kh0 + 2 * bjh[i]) * p.KW + kw0 + 2 * bjw[i]) * p.C + bkc[i]);
        bkc[i] += BKT;
        while (bkc[i] >= p.C) { bkc[i] -= p.C; if (++bjw[i] == nkw) { bjw[i] = 0; ++bjh[i]; } }
      } else if (c < NCH && n0 + row < p.Co) {
        v = *reinterpret_cast<const bf16x8*>(wt16 + pl * plane + (long)(n0 + row) * p.Kp + kt * BKT + ch * 8);
      }
      rb[i] = v;
    }
    kc += BKT;
    while (kc >= p.C) { kc -= p.C; if (++kkw == nkw) { kkw = 0; ++kkh; } }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      bf16x4 parts[NS];
      split4f<NS, FMT>(ra[i], scA, parts);
#pragma unroll
      for (int s = 0; s < NS; ++s)
        *reinterpret_cast<bf16x4*>(&sA[buf][s][(row0 + RPP * i) * LD + col4 * 4]) = parts[s];
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int c = tid + 256 * i;
      const int pl = c / (BN * CPR), w = c % (BN * CPR), row = w / CPR, ch = w % CPR;
      if (c < NCH) *reinterpret_cast<bf16x8*>(&sB[buf][pl][row * LD + ch * 8]) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (Kc + BKT - 1) / BKT;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const int ao = (wm * TM * 32 + li) * LD + 8 * lh;
    const int bo = (wn * TN * 32 + li) * LD + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks) {
      bf16x8 fa[TM][NS], fb[TN][NS];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          fa[a][s] = *reinterpret_cast<const bf16x8*>(&sA[buf][s][ao + a * 32 * LD + ks * 16]);
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          fb[b][s] = *reinterpret_cast<const bf16x8*>(&sB[buf][s][bo + b * 32 * LD + ks * 16]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) mfma_splitf<NS, FMT>(fa[a], fb[b], acc[a][b]);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }
  if (FMT) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] *= inv;
  }
  igemm_epilogue_map<TM, TN>(p, pm, acc, m0, n0, wm, wn, li, lh);
}

// one thread per output element; the self-check restatement of the above
__global__ void conv_naive_kernel(ConvK p) {
  long total = (long)p.M * p.Co;
  const int HWo = p.Ho * p.Wo;
  const int HL = p.H << p.ups, WL = p.W << p.ups;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int o = (int)(e % p.Co);
    int m = (int)(e / p.Co);
    int n = m / HWo, rem = m - n * HWo, oy = rem / p.Wo, ox = rem - oy * p.Wo;
    float acc = 0.f;
    for (int kh = 0; kh < p.KH; ++kh)
      for (int kw = 0; kw < p.KW; ++kw) {
        int iy, ix;
        if (!p.transposed) {
          iy = oy * p.stride - p.pad + kh; ix = ox * p.stride - p.pad + kw;
          if (iy < 0 || ix < 0 || iy >= HL || ix >= WL) continue;
        } else {
          int ty = oy + p.pad - kh, tx = ox + p.pad - kw;
          if (ty < 0 || tx < 0 || ty % p.stride || tx % p.stride) continue;
          iy = ty / p.stride; ix = tx / p.stride;
          if (iy >= HL || ix >= WL) continue;
        }
        iy >>= p.ups; ix >>= p.ups;
        const float* xp = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn + (long)iy * p.xsh + (long)ix * p.xsw;
        const float* wp = p.wt + (long)o * p.K + (long)(kh * p.KW + kw) * p.C;
        for (int c = 0; c < p.C; ++c) acc = fmaf(xp[c], wp[c], acc);
      }
    float v = acc;
    if (p.bias) v += p.bias[o];
    v = mt_act_fn(v, p.act) * p.alpha;
    if (p.res) v += p.beta * p.res[(long)n * p.rsn + (long)oy * p.rsh + (long)ox * p.rsw + o];
    float* yp = p.y + (long)(n / p.yni) * p.ysg + (long)(n % p.yni) * p.ysn + (long)oy * p.ysh + (long)ox * p.ysw + o;
    if (p.accumulate) v += *yp;
    if (p.ds && o >= p.dact_c0) v *= mt_dact_fn(p.ds[(long)n * p.dsn + (long)oy * p.dsh + (long)ox * p.dsw + o], p.dact);
    *yp = v;
  }
}

template <int WGM, int WGN, int TM, int TN>
void launch_igemm(const ConvK& k, hipStream_t s, int prec = MT_PREC_F32) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  dim3 grid((k.M + BM - 1) / BM, (k.Co + BN - 1) / BN, k.zb);
  if (k.par) {   // four output-parity classes of a stride-2 transposed gather
    const int mc = k.N * ((k.Ho + 1) / 2) * ((k.Wo + 1) / 2);
    dim3 g4((mc + BM - 1) / BM, (k.Co + BN - 1) / BN, 4);
    if (prec == MT_PREC_BF16X3) hipLaunchKernelGGL((igemm_split_kernel<WGM, WGN, TM, TN, 2, true>), g4, dim3(256), 0, s, k);
    else if (prec == MT_PREC_F16X3) hipLaunchKernelGGL((igemm_split_kernel<WGM, WGN, TM, TN, 2, true, 1>), g4, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((igemm_split_kernel<WGM, WGN, TM, TN, 3, true>), g4, dim3(256), 0, s, k);
    return;
  }
  if (prec == MT_PREC_BF16X3) hipLaunchKernelGGL((igemm_split_kernel<WGM, WGN, TM, TN, 2, false>), grid, dim3(256), 0, s, k);
  else if (prec == MT_PREC_F16X3) hipLaunchKernelGGL((igemm_split_kernel<WGM, WGN, TM, TN, 2, false, 1>), grid, dim3(256), 0, s, k);
  else if (prec == MT_PREC_BF16X6) hipLaunchKernelGGL((igemm_split_kernel<WGM, WGN, TM, TN, 3, false>), grid, dim3(256), 0, s, k);
  else hipLaunchKernelGGL((igemm_kernel<WGM, WGN, TM, TN>), grid, dim3(256), 0, s, k);
}

// ------------------------------------------------------------------ wgrad --

// dW[a][tap][b] partial over one pixel range.  GEMM with K = pixels: both
// operands are pixel-major, so LDS tiles are [32 pixels][channels] and the
// MFMA operands (one f32 per lane) are read with conflict-free ds_read_b32.
template <int WGM, int WGN, int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_kernel(WgK p) {
  constexpr int BMA = WGM * TM * 32, BNB = WGN * TN * 32;
  constexpr int LDA = BMA + 4, LDB = BNB + 4;
  constexpr int FA = BMA / 4, FB = BNB / 4;          // float4 per pixel row
  constexpr int PA = 256 / FA, PB = 256 / FB;        // pixel rows per pass
  constexpr int RA = 32 / PA, RB = 32 / PB;
  __shared__ __attribute__((aligned(16))) float sD[2][32 * LDA];
  __shared__ __attribute__((aligned(16))) float sG[2][32 * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN, li = lane & 31, lh = lane >> 5;
  int bid = blockIdx.x;
  const int bt = bid % p.btiles; bid /= p.btiles;
  const int at = bid % p.atiles; bid /= p.atiles;
  const int tap = bid;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int a0 = at * BMA, b0 = bt * BNB;
  const int split = blockIdx.y;
  const int pbeg = split * p.per_split;
  const int pend = min(p.M, pbeg + p.per_split);
  const int HWd = p.Hd * p.Wd;
  const int HL = p.Hg << p.ups, WL = p.Wg << p.ups;

  const int acol = (tid % FA) * 4, arow = tid / FA;
  const int bcol = (tid % FB) * 4, brow = tid / FB;
  f32x4 rd[RA], rg[RB];
  auto load_chunk = [&](int pix0) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int pidx = pix0 + arow + PA * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pidx < pend && a0 + acol < p.Cd) {
        int n = pidx / HWd, rem = pidx - n * HWd, y = rem / p.Wd, x = rem - y * p.Wd;
        v = *reinterpret_cast<const f32x4*>(p.d + (long)(n / p.dni) * p.dsg + (long)(n % p.dni) * p.dsn + (long)y * p.dsh + (long)x * p.dsw + a0 + acol);
      }
      rd[i] = v;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      int pidx = pix0 + brow + PB * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pidx < pend && b0 + bcol < p.Cg) {
        int n = pidx / HWd, rem = pidx - n * HWd, y = rem / p.Wd, x = rem - y * p.Wd;
        int iy = y * p.stride - p.pad + kh, ix = x * p.stride - p.pad + kw;
        if ((unsigned)iy < (unsigned)HL && (unsigned)ix < (unsigned)WL) {
          iy >>= p.ups; ix >>= p.ups;
          v = *reinterpret_cast<const f32x4*>(p.g + (long)(n / p.gni) * p.gsg + (long)(n % p.gni) * p.gsn + (long)iy * p.gsh + (long)ix * p.gsw + b0 + bcol);
        }
      }
      rg[i] = v;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(&sD[buf][(arow + PA * i) * LDA + acol]) = rd[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(&sG[buf][(brow + PB * i) * LDB + bcol]) = rg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nchunks = pend > pbeg ? (pend - pbeg + 31) / 32 : 0;
  if (nchunks > 0) {
    load_chunk(pbeg);
    store_chunk(0);
  }
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) load_chunk(pbeg + (ch + 1) * 32);
    const float* A = &sD[buf][lh * LDA + wm * TM * 32 + li];
    const float* B = &sG[buf][lh * LDB + wn * TN * 32 + li];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a] = A[2 * j * LDA + a * 32];
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = B[2 * j * LDB + b * 32];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (ch + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }
  const int T = p.KH * p.KW;
  float* slab = p.slabs + (long)split * p.Cd * T * p.Cg;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ai = a0 + wm * TM * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ai >= p.Cd) continue;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int bi = b0 + wn * TN * 32 + b * 32 + li;
        if (bi < p.Cg) slab[((long)ai * T + tap) * p.Cg + bi] = acc[a][b][r];
      }
    }
}

// Split-bf16 weight gradient: K = pixels.  Both operands are pixel-major in
// HBM, so the staging pass transposes while it splits: one task = (pixel pair,
// 4-channel group) -> two 16-B loads, and per channel and plane ONE 32-bit LDS
// store of the two neighbouring pixels, into [channel][pixel] tiles that the
// MFMA operands then read along k with ds_read_b128.
template <int WGM, int WGN, int TM, int TN, int NS, int FMT = 0>
__global__ __launch_bounds__(256) void wgrad_split_kernel(WgK p) {
  constexpr int BKT = NS == 3 ? 16 : 32;                        // pixels per LDS stage
  constexpr int LD = BKT + 8, PPN = BKT / 2;                    // pixel pairs per stage
  constexpr int BMA = WGM * TM * 32, BNB = WGN * TN * 32;
  constexpr int TA = PPN * (BMA / 4), TB = PPN * (BNB / 4);     // staging tasks per stage
  constexpr int RA = (TA + 255) / 256, RB = (TB + 255) / 256;
  __shared__ __attribute__((aligned(16))) __bf16 sD[2][NS][BMA * LD];
  __shared__ __attribute__((aligned(16))) __bf16 sG[2][NS][BNB * LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN, li = lane & 31, lh = lane >> 5;
  int bid = blockIdx.x;
  const int bt = bid % p.btiles; bid /= p.btiles;
  const int at = bid % p.atiles; bid /= p.atiles;
  const int tap = bid;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int a0 = at * BMA, b0 = bt * BNB;
  const int split = blockIdx.y;
  const int pbeg = split * p.per_split;
  const int pend = min(p.M, pbeg + p.per_split);
  const int HWd = p.Hd * p.Wd;
  const int HL = p.Hg << p.ups, WL = p.Wg << p.ups;
  const float scD = FMT ? pow2_scale(*p.damax) : 1.f, scG = FMT ? pow2_scale(*p.gamax) : 1.f;
  const float inv = FMT ? 1.f / (scD * scG) : 1.f;

  f32x4 rd[RA][2], rg[RB][2];
  auto load_chunk = [&](int pix0) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int t = tid + 256 * i, pp = t % PPN, cg = t / PPN;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int pidx = pix0 + 2 * pp + e;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < TA && pidx < pend && a0 + 4 * cg < p.Cd) {
          int n = pidx / HWd, rem = pidx - n * HWd, y = rem / p.Wd, x = rem - y * p.Wd;
          v = *reinterpret_cast<const f32x4*>(p.d + (long)(n / p.dni) * p.dsg + (long)(n % p.dni) * p.dsn +
                                              (long)y * p.dsh + (long)x * p.dsw + a0 + 4 * cg);
        }
        rd[i][e] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int t = tid + 256 * i, pp = t % PPN, cg = t / PPN;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int pidx = pix0 + 2 * pp + e;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < TB && pidx < pend && b0 + 4 * cg < p.Cg) {
          int n = pidx / HWd, rem = pidx - n * HWd, y = rem / p.Wd, x = rem - y * p.Wd;
          int iy = y * p.stride - p.pad + kh, ix = x * p.stride - p.pad + kw;
          if ((unsigned)iy < (unsigned)HL && (unsigned)ix < (unsigned)WL) {
            iy >>= p.ups; ix >>= p.ups;
            v = *reinterpret_cast<const f32x4*>(p.g + (long)(n / p.gni) * p.gsg + (long)(n % p.gni) * p.gsn +
                                                (long)iy * p.gsh + (long)ix * p.gsw + b0 + 4 * cg);
          }
        }
        rg[i][e] = v;
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int t = tid + 256 * i, pp = t % PPN, cg = t / PPN;
      if (t < TA) {
        bf16x4 p0[NS], p1[NS];
        split4f<NS, FMT>(rd[i][0], scD, p0);
        split4f<NS, FMT>(rd[i][1], scD, p1);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            bf16x2 v = {p0[s][c], p1[s][c]};
            *reinterpret_cast<bf16x2*>(&sD[buf][s][(4 * cg + c) * LD + 2 * pp]) = v;
          }
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int t = tid + 256 * i, pp = t % PPN, cg = t / PPN;
      if (t < TB) {
        bf16x4 p0[NS], p1[NS];
        split4f<NS, FMT>(rg[i][0], scG, p0);
        split4f<NS, FMT>(rg[i][1], scG, p1);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            bf16x2 v = {p0[s][c], p1[s][c]};
            *reinterpret_cast<bf16x2*>(&sG[buf][s][(4 * cg + c) * LD + 2 * pp]) = v;
          }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nchunks = pend > pbeg ? (pend - pbeg + BKT - 1) / BKT : 0;
  if (nchunks > 0) {
    load_chunk(pbeg);
    store_chunk(0);
  }
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) load_chunk(pbeg + (ch + 1) * BKT);
    const int ao = (wm * TM * 32 + li) * LD + 8 * lh;
    const int bo = (wn * TN * 32 + li) * LD + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks) {
      bf16x8 fa[TM][NS], fb[TN][NS];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          fa[a][s] = *reinterpret_cast<const bf16x8*>(&sD[buf][s][ao + a * 32 * LD + ks * 16]);
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          fb[b][s] = *reinterpret_cast<const bf16x8*>(&sG[buf][s][bo + b * 32 * LD + ks * 16]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) mfma_splitf<NS, FMT>(fa[a], fb[b], acc[a][b]);
    }
    if (ch + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }
  const int T = p.KH * p.KW;
  float* slab = p.slabs + (long)split * p.Cd * T * p.Cg;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ai = a0 + wm * TM * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ai >= p.Cd) continue;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int bi = b0 + wn * TN * 32 + b * 32 + li;
        if (bi < p.Cg) slab[((long)ai * T + tap) * p.Cg + bi] = acc[a][b][r] * inv;
      }
    }
}

// packed weights for the split paths: dst[plane][a][Kp], plane s = s-th bf16
// term, Kp = row length padded to a multiple of 32 with zeros
__global__ void weight_pack_split_kernel(const float* src, int A, int B, int T, long sa, long sb, long st, __bf16* dst,
                                         int AP, int BP, int Kp, int NS, int flip, const float* wamax) {
  pack_split_body(src, A, B, T, sa, sb, st, dst, AP, BP, Kp, NS, flip, wamax);
}

// dst[a][t][b] (+)= sum_s slabs[s][a][t][b]; 256 threads = 64 consecutive
// elements x 4 interleaved split lanes, combined through LDS.  TS >= T is the
// slab's tap count: tap T (column 0), when present, holds the bias sums.
struct ReduceSubs { mt_reduce_sub s[8]; };

__device__ __forceinline__ void wgrad_reduce_body(const float* slabs, int splits, int AP, int T, int TS, int BP,
                                                  float* dst, int A, int B, long sa, long sb, long st, int accumulate,
                                                  float scale, float* db) {
  // 256 threads = 64 groups of FOUR consecutive b x 4 interleaved split lanes: every load is 16 bytes (a wave reads
  // 1 KB of a slab per instruction; the one-float form moved 3.1 TB/s, and a DFPN step reads 84 MB of slabs 124 times),
  // four slabs in flight per lane.  BP is a multiple of 8, so the groups are aligned; b >= B is padding and not written.
  __shared__ f32x4 part[4][64];
  const long per = (long)AP * TS * BP;
  const int B4 = (B + 3) >> 2;
  const long total = (long)A * TS * B4;
  const int e_l = threadIdx.x & 63, sl = threadIdx.x >> 6;
  for (long e0 = (long)blockIdx.x * 64; e0 < total; e0 += (long)gridDim.x * 64) {
    const long e = e0 + e_l;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int b = 0, t = 0, a = 0;
    if (e < total) {
      b = 4 * (int)(e % B4);
      long r = e / B4;
      t = (int)(r % TS);
      a = (int)(r / TS);
      const float* s = slabs + ((long)a * TS + t) * BP + b;
      int i = sl;
      for (; i + 12 < splits; i += 16) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(s + i * per), v1 = *reinterpret_cast<const f32x4*>(s + (i + 4) * per);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(s + (i + 8) * per), v3 = *reinterpret_cast<const f32x4*>(s + (i + 12) * per);
        acc += (v0 + v1) + (v2 + v3);
      }
      for (; i < splits; i += 4) acc += *reinterpret_cast<const f32x4*>(s + i * per);
    }
    part[sl][e_l] = acc;
    __syncthreads();
    if (sl == 0 && e < total) {
      const f32x4 v = ((part[0][e_l] + part[1][e_l]) + (part[2][e_l] + part[3][e_l])) * scale;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (b + j >= B) break;
        if (t < T) {
          float* d = dst + a * sa + (b + j) * sb + t * st;
          *d = accumulate ? *d + v[j] : v[j];
        } else if (b + j == 0 && db) {
          db[a] = accumulate ? db[a] + v[j] : v[j];
        }
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* slabs, int splits, int AP, int T, int TS, int BP,
                                                           float* dst, int A, int B, long sa, long sb, long st,
                                                           int accumulate, float scale, float* db) {
  wgrad_reduce_body(slabs, splits, AP, T, TS, BP, dst, A, B, sa, sb, st, accumulate, scale, db);
}

// several sub-blocks (row ranges a0.. of the same slabs) in one launch: blockIdx.y = sub-block
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const float* slabs, int splits, int AP, int T, int TS,
                                                                 int BP, ReduceSubs subs, int accumulate, float scale) {
  const mt_reduce_sub u = subs.s[blockIdx.y];
  wgrad_reduce_body(slabs + (long)u.a0 * TS * BP, splits, AP, T, TS, BP, u.dst, u.A, u.B, u.sa, u.sb, u.st, accumulate,
                    scale, u.db);
}

__global__ void bias_grad_partial_kernel(const float* d, long dsn, long dsh, long dsw, int Hd, int Wd, int CP,
                                         int M, float* ws) {
  // block b sums pixels [b*chunk, ...) for every channel; thread t owns channel t % CP rows t / CP
  const int C4 = CP / 4;
  const int rows = blockDim.x / C4;
  const int c4 = threadIdx.x % C4, r = threadIdx.x / C4;
  const int HW = Hd * Wd;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (r < rows) {
    // four independent loads in flight per thread (the block is alone on its CU)
    const long step = (long)gridDim.x * rows;
    long pix = (long)blockIdx.x * rows + r;
    auto at = [&](long q) {
      int n = (int)(q / HW), rem = (int)(q - (long)n * HW), y = rem / Wd, x = rem - y * Wd;
      return *reinterpret_cast<const f32x4*>(d + (long)n * dsn + (long)y * dsh + (long)x * dsw + c4 * 4);
    };
    for (; pix + 3 * step < M; pix += 4 * step) {
      const f32x4 v0 = at(pix), v1 = at(pix + step), v2 = at(pix + 2 * step), v3 = at(pix + 3 * step);
      acc += (v0 + v1) + (v2 + v3);
    }
    for (; pix < M; pix += step) acc += at(pix);
  }
  extern __shared__ float sm[];   // [rows][CP]
  if (r < rows) *reinterpret_cast<f32x4*>(&sm[r * CP + c4 * 4]) = acc;
  __syncthreads();
  for (int c = threadIdx.x; c < CP; c += blockDim.x) {
    float s = 0.f;
    for (int i = 0; i < rows; ++i) s += sm[i * CP + c];
    ws[(long)blockIdx.x * CP + c] = s;
  }
}

__global__ void bias_grad_final_kernel(const float* ws, int nblk, int CP, float* db, int C, int accumulate) {
  // 64 channels x 4 partial-sum lanes per block
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (int i = sl; i < nblk; i += 4) s += ws[(long)i * CP + c];
  part[sl][threadIdx.x & 63] = s;
  __syncthreads();
  if (sl == 0 && c < C) {
    const float v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    db[c] = accumulate ? db[c] + v : v;
  }
}

__global__ void weight_pack_kernel(const float* src, int A, int B, int T, long sa, long sb, long st, float* dst,
                                   int AP, int BP) {
  const long total = (long)AP * T * BP;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int b = (int)(e % BP);
    long r = e / BP;
    int t = (int)(r % T);
    int a = (int)(r / T);
    dst[e] = (a < A && b < B) ? src[a * sa + b * sb + t * st] : 0.f;
  }
}

__global__ void pad_vector_kernel(const float* src, int n, float* dst, int np) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < np) dst[i] = i < n ? src[i] : 0.f;
}

int wgrad_cfg(const mt_wgrad* d, int* atiles, int* btiles, int* cfg) {
  // pick the block tile from the channel counts
  const int cd = d->d.C, cg = d->g.C;
  if (cd > 64 && cg > 64) { *cfg = 0; *atiles = (cd + 127) / 128; *btiles = (cg + 127) / 128; }        // 128 x 128
  else if (cd <= 32) { *cfg = 2; *atiles = 1; *btiles = (cg + 127) / 128; }                             // 32 x 128
  else if (cg <= 32) { *cfg = 3; *atiles = (cd + 127) / 128; *btiles = 1; }                             // 128 x 32
  else { *cfg = 1; *atiles = (cd + 63) / 64; *btiles = (cg + 63) / 64; }                                // 64 x 64
  return 0;
}

}  // namespace

int mt_halo_try_launch(const mt_conv* d, void* stream);   // conv_halo.hip
int mt_halo_planes_try_launch(const mt_conv* d, void* stream);   // conv_halo_planes.hip

extern "C" int mt_conv_run(const mt_conv* d, void* stream) {
  if (d && d->algo == MT_ALGO_AUTO && mt_halo_try_launch(d, stream)) return mt_check_launch("mt_conv_run(halo)");
  if (d && d->algo == MT_ALGO_AUTO && mt_halo_planes_try_launch(d, stream)) return mt_check_launch("mt_conv_run(planes)");
  MT_REQUIRE(!d || d->wt_layout == MT_WT_ROWS, "mt_conv: halo weight tiles given to a layer the halo kernel does not take");
  ConvK k;
  int rc = fill_convk(d, &k);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int prec = d->precision;
  MT_REQUIRE(prec >= MT_PREC_F32 && prec <= MT_PREC_F16X3, "mt_conv: unknown precision %d", prec);
  MT_REQUIRE(prec != MT_PREC_F16X3 || (d->x_amax && d->w_amax), "mt_conv: F16X3 needs x_amax and w_amax");
  if (prec != MT_PREC_F32) k.Kp = ((k.K + 31) / 32) * 32;
  k.par = (prec != MT_PREC_F32 && k.transposed && k.stride == 2 && (k.C & 7) == 0 && k.ups == 0) ? 1 : 0;
  if (k.Co > 64) launch_igemm<2, 2, 2, 2>(k, s, prec);        // 128 x 128
  else if (k.Co > 32) launch_igemm<2, 2, 2, 1>(k, s, prec);   // 128 x 64
  else launch_igemm<4, 1, 1, 1>(k, s, prec);                  // 128 x 32
  return mt_check_launch("mt_conv_run");
}

extern "C" int mt_conv_run_naive(const mt_conv* d, void* stream) {
  ConvK k;
  int rc = fill_convk(d, &k);
  if (rc) return rc;
  hipLaunchKernelGGL(conv_naive_kernel, dim3(mt_grid_1d((int64_t)k.M * k.Co, 256, 65535)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), k);
  return mt_check_launch("mt_conv_run_naive");
}

extern "C" int64_t mt_wgrad_slab_floats(const mt_wgrad* d) {
  if (!d) return 0;
  return (int64_t)d->splits * d->d.C * (d->KH * d->KW + (d->with_bias ? 1 : 0)) * d->g.C;
}

int mt_wgrad_halo_try_launch(const mt_wgrad* d, void* stream);   // conv_halo.hip

extern "C" int mt_wgrad_run(const mt_wgrad* d, void* stream) {
  MT_REQUIRE(d && mt_view_ok(&d->d) && mt_view_ok(&d->g), "mt_wgrad: invalid views");
  MT_REQUIRE(d->slabs && d->splits >= 1, "mt_wgrad: slabs/splits");
  MT_REQUIRE(d->stride == 1 || d->stride == 2, "mt_wgrad: stride must be 1 or 2");
  MT_REQUIRE(d->d_terms >= 0 && d->d_terms <= 2, "mt_wgrad: d_terms must be 0, 1 or 2 (got %d)", d->d_terms);
  MT_REQUIRE(d->d.N == d->g.N, "mt_wgrad: batch mismatch");
  const int HL = d->g.H << d->ups, WL = d->g.W << d->ups;
  MT_REQUIRE((HL + 2 * d->pad - d->KH) / d->stride + 1 == d->d.H && (WL + 2 * d->pad - d->KW) / d->stride + 1 == d->d.W,
             "mt_wgrad: dense tensor %dx%d is not conv(gathered %dx%d)", d->d.H, d->d.W, HL, WL);
  MT_REQUIRE((int64_t)d->d.N * d->d.H * d->d.W < (1ll << 31), "mt_wgrad: too many pixels");
  { const int rc = mt_wgrad_tiles_check(d, mt_wgrad_select(d)); if (rc) return rc; }
  if (d->algo == MT_ALGO_AUTO && mt_wgrad_halo_try_launch(d, stream)) return mt_check_launch("mt_wgrad_run(halo)");
  MT_REQUIRE(!d->with_bias, "mt_wgrad: with_bias is only available on the halo-tile path (stride 1, 3x3/5x5, split precision)");
  WgK k;
  k.d = d->d.p; k.dsn = d->d.sn; k.dsh = d->d.sh; k.dsw = d->d.sw; k.Hd = d->d.H; k.Wd = d->d.W; k.Cd = d->d.C;
  k.g = d->g.p; k.gsn = d->g.sn; k.gsh = d->g.sh; k.gsw = d->g.sw; k.Hg = d->g.H; k.Wg = d->g.W; k.Cg = d->g.C;
  k.dni = d->d.n_in; k.dsg = d->d.sg; k.gni = d->g.n_in; k.gsg = d->g.sg;
  k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.ups = d->ups;
  k.M = d->d.N * d->d.H * d->d.W;
  int per = (k.M + d->splits - 1) / d->splits;
  k.per_split = ((per + 31) / 32) * 32;
  MT_REQUIRE((int64_t)k.per_split * d->splits >= k.M, "mt_wgrad: split arithmetic");
  k.slabs = d->slabs;
  k.damax = d->d_amax; k.gamax = d->g_amax;
  k.skip = 0; k.ntl = 0;
  int cfg;
  wgrad_cfg(d, &k.atiles, &k.btiles, &cfg);
  dim3 grid(k.atiles * k.btiles * d->KH * d->KW, d->splits);
  hipStream_t s = static_cast<hipStream_t>(stream);
  MT_REQUIRE(d->precision >= MT_PREC_F32 && d->precision <= MT_PREC_F16X3, "mt_wgrad: unknown precision");
  MT_REQUIRE(d->precision != MT_PREC_F16X3 || (d->d_amax && d->g_amax), "mt_wgrad: F16X3 needs d_amax and g_amax");
#define MT_WG_LAUNCH(KERN, ...)                                                                         \
  switch (cfg) {                                                                                        \
    case 0: hipLaunchKernelGGL((KERN<2, 2, 2, 2 __VA_ARGS__>), grid, dim3(256), 0, s, k); break;        \
    case 1: hipLaunchKernelGGL((KERN<2, 2, 1, 1 __VA_ARGS__>), grid, dim3(256), 0, s, k); break;        \
    case 2: hipLaunchKernelGGL((KERN<1, 4, 1, 1 __VA_ARGS__>), grid, dim3(256), 0, s, k); break;        \
    default: hipLaunchKernelGGL((KERN<4, 1, 1, 1 __VA_ARGS__>), grid, dim3(256), 0, s, k); break;       \
  }
#define MT_COMMA ,
  if (d->precision == MT_PREC_BF16X3) { MT_WG_LAUNCH(wgrad_split_kernel, MT_COMMA 2) }
  else if (d->precision == MT_PREC_BF16X6) { MT_WG_LAUNCH(wgrad_split_kernel, MT_COMMA 3) }
  else if (d->precision == MT_PREC_F16X3) { MT_WG_LAUNCH(wgrad_split_kernel, MT_COMMA 2 MT_COMMA 1) }
  else { MT_WG_LAUNCH(wgrad_kernel) }
#undef MT_WG_LAUNCH
#undef MT_COMMA
  return mt_check_launch("mt_wgrad_run");
}

extern "C" int mt_wgrad_reduce(const float* slabs, int32_t splits, int32_t AP, int32_t T, int32_t TS, int32_t BP,
                               float* dst, int32_t A, int32_t B, int64_t sa, int64_t sb, int64_t st, int32_t accumulate,
                               float scale, float* db, void* stream) {
  MT_REQUIRE(slabs && dst && splits >= 1 && A <= AP && B <= BP && (TS == T || TS == T + 1),
             "mt_wgrad_reduce: bad arguments");
  // (the slabs are read with 16-byte loads at columns b .. b + 3: ADVICE round 3)
  MT_REQUIRE((BP & 3) == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0, "mt_wgrad_reduce: BP must be a multiple of 4 and slabs 16-byte aligned");
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(mt_grid_1d((int64_t)A * TS * ((B + 3) / 4), 64, 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), slabs, splits, AP, T, TS, BP, dst, A, B, (long)sa, (long)sb,
                     (long)st, accumulate, scale, db);
  return mt_check_launch("mt_wgrad_reduce");
}

extern "C" int mt_wgrad_reduce_multi(const float* slabs, int32_t splits, int32_t AP, int32_t T, int32_t TS, int32_t BP,
                                     const mt_reduce_sub* subs, int32_t nsub, int32_t accumulate, float scale,
                                     void* stream) {
  MT_REQUIRE((BP & 3) == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0, "mt_wgrad_reduce_multi: BP must be a multiple of 4 and slabs 16-byte aligned");
  MT_REQUIRE(slabs && subs && splits >= 1 && nsub >= 1 && nsub <= 8 && (TS == T || TS == T + 1),
             "mt_wgrad_reduce_multi: bad arguments");
  ReduceSubs rs;
  int64_t most = 0;
  for (int i = 0; i < nsub; ++i) {
    const mt_reduce_sub& u = subs[i];
    MT_REQUIRE(u.dst && u.a0 >= 0 && u.A >= 1 && u.a0 + u.A <= AP && u.B <= BP, "mt_wgrad_reduce_multi: sub-block %d", i);
    rs.s[i] = u;
    most = most > (int64_t)u.A * TS * ((u.B + 3) / 4) ? most : (int64_t)u.A * TS * ((u.B + 3) / 4);
  }
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(mt_grid_1d(most, 64, 2048), nsub), dim3(256), 0,
                     static_cast<hipStream_t>(stream), slabs, splits, AP, T, TS, BP, rs, accumulate, scale);
  return mt_check_launch("mt_wgrad_reduce_multi");
}

extern "C" int mt_bias_grad(const mt_view* d, float* db, int32_t C, float* ws, int32_t accumulate, void* stream) {
  MT_REQUIRE(mt_view_flat(d) && db && ws && C <= d->C, "mt_bias_grad: bad arguments");
  MT_REQUIRE(d->C <= 1024, "mt_bias_grad: too many channels");
  const int M = d->N * d->H * d->W;
  const int C4 = d->C / 4;
  const int rows = 256 / C4 > 0 ? 256 / C4 : 1;
  const int threads = rows * C4;
  int nblk = (M + rows * 8 - 1) / (rows * 8);
  if (nblk > 256) nblk = 256;      // ws holds 256 * D.C partial sums
  if (nblk < 1) nblk = 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bias_grad_partial_kernel, dim3(nblk), dim3(threads), rows * d->C * sizeof(float), s, d->p,
                     (long)d->sn, (long)d->sh, (long)d->sw, d->H, d->W, d->C, M, ws);
  hipLaunchKernelGGL(bias_grad_final_kernel, dim3((C + 63) / 64), dim3(256), 0, s, ws, nblk, d->C, db, C, accumulate);
  return mt_check_launch("mt_bias_grad");
}

extern "C" int mt_weight_pack(const float* src, int32_t A, int32_t B, int32_t T, int64_t sa, int64_t sb, int64_t st,
                              float* dst, int32_t AP, int32_t BP, void* stream) {
  MT_REQUIRE(src && dst && A <= AP && B <= BP && (BP & 3) == 0, "mt_weight_pack: bad arguments");
  hipLaunchKernelGGL(weight_pack_kernel, dim3(mt_grid_1d((int64_t)AP * T * BP, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, A, B, T, (long)sa, (long)sb, (long)st, dst, AP, BP);
  return mt_check_launch("mt_weight_pack");
}

extern "C" int mt_pad_vector(const float* src, int32_t n, float* dst, int32_t np, void* stream) {
  MT_REQUIRE(src && dst && n <= np, "mt_pad_vector: bad arguments");
  hipLaunchKernelGGL(pad_vector_kernel, dim3((np + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                     n, dst, np);
  return mt_check_launch("mt_pad_vector");
}

namespace {
// Fused masked cosine correlation of one (clip, reference) pair, P = 256
// positions (CorrelationVGG.correlation_masked_4d, model_dfpn.py:534-565):
//   out[pt][pr] = <t[pt], r[pr]> * ft[pt] * fr[pr],  f = v / (|v| * ||x|| + 1e-9)
// ONE pass over HBM: the block streams the two [256][C] fp32 feature slabs in
// 32-channel chunks, splits them into bf16 hi/lo planes in LDS (bf16x3
// products), accumulates the whole 256 x 256 output in registers and the
// squared row norms on the side, and applies masks and norms in the epilogue.
// Algorithmic traffic only: 2 x 512 KiB in, 256 KiB out per pair (the target
// slab is shared by the R pairs of a clip and stays in L2).
//
// 512 threads = two groups of 4 waves, one wave of each group per SIMD.  The
// LDS holds two chunk images (all 160 KiB).  While one group runs the MFMAs of
// chunk k, the other splits + stores its share of chunk k+1 into the other
// image, then the roles swap: the matrix pipe of every SIMD always has a wave
// feeding it, and the VALU split runs in its shadow.  Each thread keeps two
// register sets of prefetched chunks, so global loads are in flight all the
// time (two chunks ahead).
__global__ __launch_bounds__(512, 1) void corr_fused_kernel(const float* ft, long tsn, long tsh, long tsw,
                                                         const float* fr, long rsn, long rsh, long rsw, int W,
                                                         int C, const float* vt, const float* vr, int R,
                                                         float* out) {
  constexpr int P = 256, CKC = 32, LD = CKC + 8, NS = 2, NI = 4, PL = P * LD;
  // [buf][T|R][plane][P*LD]  = 2*2*2*20480 B = 163840 B (all of the CU's LDS)
  __shared__ __attribute__((aligned(16))) __bf16 sm[2 * 2 * NS * PL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wm = wave & 3, wn = g, li = lane & 31, lh = lane >> 5;
  const int pair = xcd_remap(blockIdx.x, gridDim.x);
  const int clip = pair / R;
  // buffer descriptors (wave-uniform) of the two feature slabs: loads use a 32-bit lane byte
  // offset + a scalar chunk offset, so no 64-bit address registers are held per load
  const int H = P / W;
  const unsigned tbytes = (unsigned)(((long)(H - 1) * tsh + (long)(W - 1) * tsw + C) * 4);
  const unsigned rbytes = (unsigned)(((long)(H - 1) * rsh + (long)(W - 1) * rsw + C) * 4);
  const auto srT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ft + (long)clip * tsn), 0, tbytes, 0x00020000);
  const auto srR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fr + (long)pair * rsn), 0, rbytes, 0x00020000);
  const int col4 = tid & 7, row0 = tid >> 3;
  unsigned toff[NI], roff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int pix = row0 + 64 * i, y = pix / W, x = pix - y * W;
    toff[i] = ((unsigned)((long)y * tsh + (long)x * tsw) + col4 * 4) * 4u;   // bytes
    roff[i] = ((unsigned)((long)y * rsh + (long)x * rsw) + col4 * 4) * 4u;
  }
  f32x4 rt[2][NI], rr[2][NI];
  float qt[NI], qr[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) qt[i] = qr[i] = 0.f;
  auto load = [&](f32x4 (&t)[NI], f32x4 (&r)[NI], int c0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) t[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srT, toff[i], c0 * 4, 0));
#pragma unroll
    for (int i = 0; i < NI; ++i) r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srR, roff[i], c0 * 4, 2));   // nt: read once
  };
  auto store = [&](const f32x4 (&t)[NI], const f32x4 (&r)[NI], int buf) {
    __bf16* bT = sm + buf * (2 * NS * PL);
    __bf16* bR = bT + NS * PL;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      qt[i] += t[i][0] * t[i][0] + t[i][1] * t[i][1] + t[i][2] * t[i][2] + t[i][3] * t[i][3];
      bf16x4 p_[NS];
      split4<NS>(t[i], p_);
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&bT[s * PL + (row0 + 64 * i) * LD + col4 * 4]) = p_[s];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      qr[i] += r[i][0] * r[i][0] + r[i][1] * r[i][1] + r[i][2] * r[i][2] + r[i][3] * r[i][3];
      bf16x4 p_[NS];
      split4<NS>(r[i], p_);
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&bR[s * PL + (row0 + 64 * i) * LD + col4 * 4]) = p_[s];
    }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int ao = (wm * 64 + li) * LD + 8 * lh, bo = (wn * 128 + li) * LD + 8 * lh;
  auto compute = [&](int buf) {
    const __bf16* bT = sm + buf * (2 * NS * PL);
    const __bf16* bR = bT + NS * PL;
#pragma unroll
    for (int ks = 0; ks < CKC / 16; ++ks) {
      bf16x8 fa[2][NS];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int s = 0; s < NS; ++s) fa[a][s] = *reinterpret_cast<const bf16x8*>(&bT[s * PL + ao + a * 32 * LD + ks * 16]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        bf16x8 fb[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) fb[s] = *reinterpret_cast<const bf16x8*>(&bR[s * PL + bo + b * 32 * LD + ks * 16]);
#pragma unroll
        for (int a = 0; a < 2; ++a) mfma_split<NS>(fa[a], fb, acc[a][b]);
      }
    }
  };
  const int nchunk = C / CKC;   // even, >= 4
  const int c_start = (2 * clip) % nchunk;
  // channel offset of chunk c < nchunk (rotated start); the pipeline tail issues no loads at all
  // (nothing relies on how an out-of-range soffset would be bounds-checked)
  auto choff = [&](int c) { int q = c + c_start; q = q >= nchunk ? q - nchunk : q; return q * CKC; };
  // Group g keeps chunk c in register set (c + g) & 1 and runs one phase behind/ahead of the
  // other group:   g0: C0 | S1 | C1 | S2 | ...      g1: S1 | C0 | S2 | C1 | ...
  // (Ck = MFMAs of chunk k from image k&1, Sk = split+store of the group's share of chunk k)
  load(rt[0], rr[0], choff(g));
  load(rt[1], rr[1], choff(1 - g));
  if (g == 0) { store(rt[0], rr[0], 0); load(rt[0], rr[0], choff(2)); }
  else        { store(rt[1], rr[1], 0); load(rt[1], rr[1], choff(2)); }
  __syncthreads();
  if (g == 1) {
    store(rt[0], rr[0], 1);
    load(rt[0], rr[0], choff(3));
    __syncthreads();
  }
  for (int k = 0; k < nchunk; k += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cs = k + h + 1 + g;        // chunk this group stores in this step (register set h^1)
      compute(h);
      __syncthreads();
      if (cs < nchunk) {
        store(rt[h ^ 1], rr[h ^ 1], (h ^ 1) ^ g);
      }
      if (cs + 2 < nchunk) load(rt[h ^ 1], rr[h ^ 1], choff(cs + 2));      // (wave-uniform: the tail issues no loads)
      __syncthreads();
    }
  }
  if (g == 0) __syncthreads();
  float* nT = reinterpret_cast<float*>(sm);
  float* nR = nT + P;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float a = qt[i], b = qr[i];
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if (col4 == 0) {
      const int pix = row0 + 64 * i;
      const float v1 = vt ? vt[(long)clip * P + pix] : 1.f, v2 = vr ? vr[(long)pair * P + pix] : 1.f;
      nT[pix] = v1 / (fabsf(v1) * sqrtf(a) + 1e-9f);
      nR[pix] = v2 / (fabsf(v2) * sqrtf(b) + 1e-9f);
    }
  }
  __syncthreads();
  float* ob = out + (long)pair * P * P;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pt = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float f1 = nT[pt];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int pr = wn * 128 + b * 32 + li;
        __builtin_nontemporal_store(acc[a][b][r] * f1 * nR[pr], &ob[(long)pt * P + pr]);   // written once, read by a later kernel
      }
    }
}
}  // namespace

extern "C" int mt_rownorm(const mt_view* x, const float* v, float* out, void* stream);

// CorrelationVGG.correlation_masked_4d (model_dfpn.py:534-565): masked L2 row
// normalisation of both feature sets, then one 256x256x512 NT GEMM per
// (clip, reference) pair, batched over blockIdx.z.
extern "C" int mt_corr_masked(const mt_view* ft, const mt_view* fr, const float* vt, const float* vr, int32_t B,
                              int32_t R, float* out, float* ws, void* stream) {
  MT_REQUIRE(mt_view_ok(ft) && mt_view_ok(fr) && out, "mt_corr_masked: bad arguments");
  MT_REQUIRE(ft->N == B && fr->N == B * R && ft->H == fr->H && ft->W == fr->W && ft->C == fr->C,
             "mt_corr_masked: feature geometry mismatch");
  const int P = ft->H * ft->W, C = ft->C;
  MT_REQUIRE((P & 3) == 0 && B * R < 65536, "mt_corr_masked: P %% 4 != 0 or too many pairs");
  const long text = ((long)(ft->H - 1) * ft->sh + (long)(ft->W - 1) * ft->sw + C) * 4;
  const long rext = ((long)(fr->H - 1) * fr->sh + (long)(fr->W - 1) * fr->sw + C) * 4;
  if (P == 256 && (C % 64) == 0 && C >= 128 && ft->n_in >= ft->N && fr->n_in >= fr->N && ft->W == fr->W &&
      text < (1L << 28) && rext < (1L << 28)) {
    // fused single-pass kernel (the DFPN case: 16 x 16 positions)
    hipLaunchKernelGGL(corr_fused_kernel, dim3(B * R), dim3(512), 0, static_cast<hipStream_t>(stream), ft->p,
                       (long)ft->sn, (long)ft->sh, (long)ft->sw, fr->p, (long)fr->sn, (long)fr->sh, (long)fr->sw, ft->W,
                       C, vt, vr, R, out);
    return mt_check_launch("mt_corr_masked(fused)");
  }
  MT_REQUIRE(ws, "mt_corr_masked: the general path (not 16 x 16 positions) needs the workspace");
  float* tn = ws;
  float* rn = ws + (size_t)B * P * C;
  int rc = mt_rownorm(ft, vt, tn, stream);
  if (rc) return rc;
  rc = mt_rownorm(fr, vr, rn, stream);
  if (rc) return rc;
  ConvK k = {};
  k.xni = 1; k.xsg = 0; k.yni = 1; k.ysg = 0;
  k.x = tn; k.xsn = 0; k.xsh = 0; k.xsw = C; k.H = 1; k.W = P; k.C = C;
  k.wt = rn; k.bias = nullptr;
  k.y = out; k.ysn = 0; k.ysh = 0; k.ysw = P; k.Ho = 1; k.Wo = P; k.Co = P;
  k.N = 1; k.KH = 1; k.KW = 1; k.stride = 1; k.pad = 0; k.transposed = 0; k.ups = 0;
  k.M = P; k.K = C;
  k.act = MT_ACT_NONE; k.alpha = 1.f;
  k.zb = B * R; k.zdiv = R; k.xz = (long)P * C; k.wz = (long)P * C; k.yz = (long)P * P;
  launch_igemm<2, 2, 2, 1>(k, static_cast<hipStream_t>(stream));
  return mt_check_launch("mt_corr_masked");
}

extern "C" int mt_weight_pack_split(const float* src, int32_t A, int32_t B, int32_t T, int64_t sa, int64_t sb,
                                    int64_t st, void* dst, int32_t AP, int32_t BP, int32_t precision, int32_t flip,
                                    const float* w_amax, void* stream) {
  MT_REQUIRE(src && dst && A <= AP && B <= BP && (BP & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0,
             "mt_weight_pack_split: bad arguments");
  MT_REQUIRE(precision == MT_PREC_BF16X3 || precision == MT_PREC_BF16X6 || precision == MT_PREC_F16X3,
             "mt_weight_pack_split: precision");
  MT_REQUIRE((precision == MT_PREC_F16X3) == (w_amax != nullptr), "mt_weight_pack_split: w_amax goes with F16X3");
  const int Kp = ((T * BP + 31) / 32) * 32;
  hipLaunchKernelGGL(weight_pack_split_kernel, dim3(mt_grid_1d((int64_t)AP * Kp, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, A, B, T, (long)sa, (long)sb, (long)st,
                     reinterpret_cast<__bf16*>(dst), AP, BP, Kp, precision == MT_PREC_BF16X6 ? 3 : 2, flip, w_amax);
  return mt_check_launch("mt_weight_pack_split");
}
