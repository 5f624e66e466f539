// Stride-1 "same" convolutions (3x3 / 5x5) as an LDS halo-tile direct conv on
// split-bf16 MFMA (the bulk of the DFPN / CHN FLOPs: VGG, SeparableConv4d,
// FlowEstimator, RRDBNet, model_dfpn.py:596-605,686-711, model_chn.py:522-535,
// 602-606, model_vgg.py:40-41, and -- with tap-flipped, channel-swapped packed
// weights -- their backward-data).
//
// The implicit-GEMM kernel of conv.hip gathers (and splits into bf16 terms)
// every input pixel once PER TAP.  Here a block owns an 8 x 16 output-pixel
// tile of one image: per K chunk of CK input channels it loads the
// (8+k-1) x (16+k-1) halo patch ONCE, splits it once, and runs all k*k taps out
// of LDS -- a tap is just a row offset into the halo image.  The per-tap
// weight tile [BN][CK] (pre-split bf16 planes) is double-buffered through
// registers; the next chunk's halo loads are in flight during the whole tap
// loop.  One barrier per tap, two at a chunk boundary.
#include "conv_common.h"

namespace {

template <int KS, int NS, int WGM, int WGN, int TM, int TN>
__global__ __launch_bounds__(256, 2) void igemm_halo_kernel(ConvK p, int tiles_y, int tiles_x) {
  constexpr int CK = NS == 3 ? 16 : 32;             // input channels per LDS stage
  constexpr int LD = CK + 8;                        // bf16 per LDS row (+16 B pad)
  constexpr int TH = 8, TW = 16;
  static_assert(WGM * TM * 32 == TH * TW, "pixel tile is 8 x 16");
  constexpr int BN = WGN * TN * 32;
  constexpr int HW_ = TW + KS - 1, HH = TH + KS - 1, HP = HH * HW_;   // halo patch
  constexpr int F4 = CK / 4, NH4 = HP * F4, RH = (NH4 + 255) / 256;
  constexpr int CPR = CK / 8, NCH = NS * BN * CPR, RB = (NCH + 255) / 256;
  constexpr int T = KS * KS;
  __shared__ __attribute__((aligned(16))) __bf16 sH[NS][HP * LD];
  __shared__ __attribute__((aligned(16))) __bf16 sB[2][NS][BN * LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int ty0 = ty * TH, tx0 = tx * TW, n0 = blockIdx.y * BN;
  const int HL = p.H << p.ups, WL = p.W << p.ups;
  const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;

  // ---- halo staging coordinates (chunk-invariant) --------------------------
  long h_off[RH];
  bool h_ok[RH];
#pragma unroll
  for (int i = 0; i < RH; ++i) {
    const int idx = tid + 256 * i;
    const int hp = idx / F4, c4 = idx % F4;
    const int hy = hp / HW_, hx = hp - hy * HW_;
    const int ly = ty0 - p.pad + hy, lx = tx0 - p.pad + hx;
    h_ok[i] = idx < NH4 && (unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL;
    h_off[i] = (long)(ly >> p.ups) * p.xsh + (long)(lx >> p.ups) * p.xsw + 4 * c4;
  }
  const __bf16* wt16 = reinterpret_cast<const __bf16*>(p.wt);
  const long plane = (long)p.Co * p.Kp;

  f32x4 rh[RH];
  bf16x8 rb[RB];
  auto load_halo = [&](int c0) {
#pragma unroll
    for (int i = 0; i < RH; ++i) {
      const int c4 = (tid + 256 * i) % F4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (h_ok[i] && c0 + 4 * c4 < p.C) v = *reinterpret_cast<const f32x4*>(xin + h_off[i] + c0);
      rh[i] = v;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < RH; ++i) {
      const int idx = tid + 256 * i;
      if (idx < NH4) {
        const int hp = idx / F4, c4 = idx % F4;
        bf16x4 parts[NS];
        split4<NS>(rh[i], parts);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&sH[s][hp * LD + 4 * c4]) = parts[s];
      }
    }
  };
  auto load_B = [&](int c0, int tap) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int c = tid + 256 * i;
      const int pl = c / (BN * CPR), w = c % (BN * CPR), row = w / CPR, ch = w % CPR;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
      if (c < NCH && n0 + row < p.Co && c0 + ch * 8 < p.C)
        v = *reinterpret_cast<const bf16x8*>(wt16 + pl * plane + (long)(n0 + row) * p.Kp + (long)tap * p.C + c0 + ch * 8);
      rb[i] = v;
    }
  };
  auto store_B = [&](int buf) {
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

  // MFMA row i of 32-pixel group g = tile pixel (2g + (i >> 4), i & 15)
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = ((2 * (wm * TM + a) + (li >> 4)) * HW_ + (li & 15)) * LD + 8 * lh;
  const int b_base = (wn * TN * 32 + li) * LD + 8 * lh;

  const int nchunk = (p.C + CK - 1) / CK;
  load_halo(0);
  store_halo();
  load_B(0, 0);
  store_B(0);
  __syncthreads();
  int it = 0;
  for (int ch = 0; ch < nchunk; ++ch) {
    const int c0 = ch * CK;
    const bool more = ch + 1 < nchunk;
    if (more) load_halo(c0 + CK);             // in flight during the whole tap loop
    int kh = 0, kw = 0;
    for (int tap = 0; tap < T; ++tap, ++it) {
      const int buf = it & 1;
      const bool last = tap == T - 1;
      if (!last) load_B(c0, tap + 1);
      else if (more) load_B(c0 + CK, 0);
      const int toff = (kh * HW_ + kw) * LD;
#pragma unroll
      for (int ks = 0; ks < CK / 16; ++ks) {
        bf16x8 fa[TM][NS], fb[TN][NS];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int s = 0; s < NS; ++s)
            fa[a][s] = *reinterpret_cast<const bf16x8*>(&sH[s][a_base[a] + toff + ks * 16]);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int s = 0; s < NS; ++s)
            fb[b][s] = *reinterpret_cast<const bf16x8*>(&sB[buf][s][b_base + b * 32 * LD + ks * 16]);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) mfma_split<NS>(fa[a], fb[b], acc[a][b]);
      }
      if (!last || more) store_B(buf ^ 1);
      if (last && more) {
        __syncthreads();                      // every wave is done with this chunk's halo
        store_halo();
      }
      __syncthreads();
      if (++kw == KS) { kw = 0; ++kh; }
    }
  }

  // ---- epilogue (same fused ops as igemm_epilogue, tile-local pixel map) ---
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int oy = ty0 + (row >> 4), ox = tx0 + (row & 15);
      if (oy >= p.Ho || ox >= p.Wo) continue;
      const long yo = (long)(n / p.yni) * p.ysg + (long)(n % p.yni) * p.ysn + (long)oy * p.ysh + (long)ox * p.ysw;
      const long ro = p.res ? (long)n * p.rsn + (long)oy * p.rsh + (long)ox * p.rsw : 0;
      const long so = p.ds ? (long)n * p.dsn + (long)oy * p.dsh + (long)ox * p.dsw : 0;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int o = n0 + wn * TN * 32 + b * 32 + li;
        if (o >= p.Co) continue;
        float v = acc[a][b][r];
        if (p.bias) v += p.bias[o];
        v = mt_act_fn(v, p.act) * p.alpha;
        if (p.res) v += p.beta * p.res[ro + o];
        if (p.accumulate) v += p.y[yo + o];
        if (p.ds && o >= p.dact_c0) v *= mt_dact_fn(p.ds[so + o], p.dact);
        p.y[yo + o] = v;
      }
    }
  }
}

template <int KS, int NS>
void launch_halo(const ConvK& k, hipStream_t s) {
  const int tiles_y = (k.Ho + 7) / 8, tiles_x = (k.Wo + 15) / 16;
  const unsigned nt = (unsigned)k.N * tiles_y * tiles_x;
  if (k.Co > 64) {
    hipLaunchKernelGGL((igemm_halo_kernel<KS, NS, 2, 2, 2, 2>), dim3(nt, (k.Co + 127) / 128), dim3(256), 0, s, k,
                       tiles_y, tiles_x);
  } else if (k.Co > 32) {
    hipLaunchKernelGGL((igemm_halo_kernel<KS, NS, 2, 2, 2, 1>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
  } else {
    hipLaunchKernelGGL((igemm_halo_kernel<KS, NS, 4, 1, 1, 1>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
  }
}

}  // namespace

// Called by mt_conv_run: returns 1 if the launch was taken by the halo kernel.
int mt_halo_try_launch(const mt_conv* d, void* stream) {
  if (d->precision != MT_PREC_BF16X3 && d->precision != MT_PREC_BF16X6) return 0;
  if (d->transposed || d->stride != 1 || d->KH != d->KW || (d->KH != 3 && d->KH != 5) || d->pad != d->KH / 2) return 0;
  if ((d->x.C & 7) != 0) return 0;
  ConvK k;
  if (fill_convk(d, &k) != MT_OK) return 0;
  k.Kp = ((k.K + 31) / 32) * 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool x6 = d->precision == MT_PREC_BF16X6;
  if (d->KH == 3) { if (x6) launch_halo<3, 3>(k, s); else launch_halo<3, 2>(k, s); }
  else { if (x6) launch_halo<5, 3>(k, s); else launch_halo<5, 2>(k, s); }
  return 1;
}
