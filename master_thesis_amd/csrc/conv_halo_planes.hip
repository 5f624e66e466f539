// Stride-2 convolutions and transposed convolutions (FlowEstimator's U-shape, model_dfpn.py:686-711;
// RRDBNet's conv_first, model_chn.py:522-528) and their backward-data on the LDS halo-tile kernel.
//
// Both gathers become stride-1 window convolutions once the stride is moved into the addressing:
//   * strided gather  acc[i][j] = sum X[2i - pad + kh][2j - pad + kw] W[kh][kw]
//     (Conv2d stride 2 forward, ConvTranspose2d backward-data): split X into its four parity planes
//     X_ab[y][x] = X[2y + a][2x + b]; tap kh = 2 dy + a + pad reads plane a at row i + dy, dy in {-1,0,1}:
//     a 3 x 3 WINDOW over each plane, with only the window positions whose (kh, kw) exist (k = 3: 4+2+2+1
//     = 9 of 36; k = 5: 9+6+6+4 = 25).  One launch walks the four planes as extra K chunks.
//   * transposed gather  acc[y][x] = sum X[(y + pad - kh) / 2][...] W[kh][kw] over the divisible taps
//     (ConvTranspose2d forward, Conv2d stride 2 backward-data): the outputs of parity class (A, B),
//     y = 2i + A, see the taps kh = A + pad - 2 dy only: one stride-1 window convolution per class over
//     the whole (small) X, written to every second output pixel.  Four launches, one per class.
// A "unit" is one plane / one class: an input view, a 9-bit window mask, and its weight fragments.
// The kernel is the register-B kernel of conv_halo.hip with the K loop running over (unit, 32-channel
// chunk) pairs and the tap loop over the set bits of the unit's mask; the weights are packed in exactly
// the order the loop consumes them (mt_weight_pack_planes), so the fragment pointer only increments.
#include <cstdlib>
#include "conv_common.h"
#include "dispatch.h"

struct mt_plane_unit {
  long xoff;            // element offset of the unit's input view inside an image
  int H, W;             // extent of the view (positions outside read as zero)
  unsigned mask;        // window positions (dy + 1) * 3 + (dx + 1) in use
  int frag0;            // index of the unit's first weight fragment block (per chunk-and-tap)
};
struct PlanesK {
  mt_plane_unit u[4];
  int nunits;
  long xsh, xsw;        // row / pixel stride of the input views
  int Hc, Wc;           // output positions (i, j) of this launch
  int omul, oya, oxa;   // output pixel = (i * omul + oya, j * omul + oxa)
  int zclasses;         // 1: blockIdx.z = output parity class (A, B): unit z alone, (oya, oxa) = (A, B), Hc / Wc per class
};

namespace {

template <int FMT, int WM>
__global__ __launch_bounds__(256, 2) void igemm_halo_planes_kernel(ConvK p, PlanesK q, int tiles_y, int tiles_x) {
  constexpr int NS = 2, CK = 32, LD = CK + 8, TH = 8, TW = 16, WN = 4 / WM, BN = WN * 32, TM = 4 / WM;
  constexpr int HW_ = TW + 2, HH = TH + 2, HP = HH * HW_;               // 3 x 3 window halo
  constexpr int F4 = CK / 4, NH4 = HP * F4, RH = (NH4 + 255) / 256;
  constexpr int KSTEPS = CK / 16;
  constexpr int HALO_E = NS * HP * LD;
  __shared__ __attribute__((aligned(16))) __bf16 sH[2 * HALO_E];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int ty0 = ty * TH, tx0 = tx * TW, n0 = blockIdx.y * BN;
  const float scA = FMT ? pow2_scale(*p.xamax) : 1.f;
  const float inv = FMT ? 1.f / (scA * pow2_scale(*p.wamax)) : 1.f;
  const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;
  // transposed gather: the four output classes share one launch (their tap counts differ: 1, 2, 2, 4 of a
  // 3 x 3 -- run one after the other they would each pay their own ramp-up and tail)
  // (the kernel argument itself is never written: a modified copy would live in scratch memory)
  const int ubase = q.zclasses ? blockIdx.z : 0;
  const int oya = q.zclasses ? (int)(blockIdx.z >> 1) : q.oya, oxa = q.zclasses ? (int)(blockIdx.z & 1) : q.oxa;
  const int Hc = q.zclasses ? (p.Ho - oya + 1) >> 1 : q.Hc, Wc = q.zclasses ? (p.Wo - oxa + 1) >> 1 : q.Wc;
  if (ty0 >= Hc || tx0 >= Wc) {     // (an empty parity-class tile: it still takes part in the launch's max reduction)
    finish_amax(p, 0.f, sH);
    return;
  }

  // halo staging coordinates (window origin -1): the same for every unit, only the bounds differ
  long h_off[RH];
  int h_y[RH], h_x[RH];
#pragma unroll
  for (int i = 0; i < RH; ++i) {
    const int idx = tid + 256 * i;
    const int hp = idx / F4, c4 = idx % F4;
    const int hy = hp / HW_, hx = hp - hy * HW_;
    h_y[i] = idx < NH4 ? ty0 - 1 + hy : -1;
    h_x[i] = tx0 - 1 + hx;
    h_off[i] = (long)h_y[i] * q.xsh + (long)h_x[i] * q.xsw + 4 * c4;
  }
  f32x4 rh[RH];
  auto load_halo = [&](int unit, int c0) {
    const float* xu = xin + q.u[ubase + unit].xoff;
    const int uh = q.u[ubase + unit].H, uw = q.u[ubase + unit].W;
#pragma unroll
    for (int i = 0; i < RH; ++i) {
      const int c4 = (tid + 256 * i) % F4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)h_y[i] < (unsigned)uh && (unsigned)h_x[i] < (unsigned)uw && c0 + 4 * c4 < p.C)
        v = *reinterpret_cast<const f32x4*>(xu + h_off[i] + c0);
      rh[i] = v;
    }
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RH; ++i) {
      const int idx = tid + 256 * i;
      if (idx < NH4) {
        const int hp = idx / F4, c4 = idx % F4;
        bf16x4 parts[NS];
        split4f<NS, FMT>(rh[i], scA, parts);
#pragma unroll
        for (int s = 0; s < NS; ++s)
          *reinterpret_cast<bf16x4*>(&sH[buf * HALO_E + s * HP * LD + hp * LD + 4 * c4]) = parts[s];
      }
    }
  };
  // weight fragments in consumption order: [unit][chunk][tap of the mask][ntile][wn][kstep][plane][lane] x 16 B
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK, ntile_w = (p.Co + BN - 1) / BN;
  bf16x8 fb[KSTEPS][NS];
  auto load_B = [&](int widx, int ks) {
    const bf16x8* f = wfr + (((long)widx * ntile_w + blockIdx.y) * WN + wn) * (KSTEPS * NS * 64) + ks * NS * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) fb[ks][s] = f[s * 64];
  };

  f32x16 acc[TM][1];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;

  // MFMA row -> tile pixel map of igemm_halo_kernel (conflict-free ds_read_b128)
  const int apx = li < 16 ? li : ((li - HW_) & 15);
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = ((2 * (wm * TM + a) + (li >> 4)) * HW_ + apx) * LD + 8 * lh;

  const int nvc = q.nunits * nchunk;        // virtual chunks: (unit, channel chunk)
  int widx = q.u[ubase].frag0;
  load_halo(0, 0);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) load_B(widx, ks);
  store_halo(0);
  __syncthreads();
  for (int vc = 0; vc < nvc; ++vc) {
    const int cur = vc & 1;
    const bool more = vc + 1 < nvc;
    const int unit = vc / nchunk;
    const int nunit = (vc + 1) / nchunk, nc0 = ((vc + 1) - nunit * nchunk) * CK;
    const __bf16* hb = sH + cur * HALO_E;
    unsigned m = q.u[ubase + unit].mask;
    bool first = true;
    while (m) {
      const int wpos = __builtin_ctz(m);
      m &= m - 1;
      const bool last = m == 0;
      if (first && more) load_halo(nunit, nc0);          // in flight during the taps of this chunk
      first = false;
      if (last && more) store_halo(cur ^ 1);             // converted into the other image under the last tap
      const int toff = ((wpos / 3) * HW_ + (wpos % 3)) * LD;
      bf16x8 fa[KSTEPS][TM][NS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int s = 0; s < NS; ++s)
            fa[ks][a][s] = *reinterpret_cast<const bf16x8*>(&hb[s * HP * LD + a_base[a] + toff + ks * 16]);
      __builtin_amdgcn_sched_barrier(0);
      ++widx;                                            // the next fragment block, whatever unit it belongs to
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
        for (int a = 0; a < TM; ++a) mfma_splitf<NS, FMT>(fa[ks][a], fb[ks], acc[a][0]);
        if (!last || more) load_B(widx, ks);
      }
    }
    if (more) __syncthreads();
  }

  float vmax = 0.f;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    float vals[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vals[r] = acc[a][0][r] * inv;
    vmax = fmaxf(vmax, halo_epilogue<16>(p, n, n0 + wn * 32 + li, vals, [&](int r, int& oy, int& ox) {
      const int row = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int i32 = row & 31;
      const int i = ty0 + (row >> 4), j = tx0 + (i32 < 16 ? i32 : ((i32 - HW_) & 15));
      // (positions past the class's extent map past Ho / Wo or are cut here)
      oy = i < Hc ? i * q.omul + oya : p.Ho;
      ox = j < Wc ? j * q.omul + oxa : p.Wo;
    }));
  }
  finish_amax(p, vmax, sH);
}

__global__ void weight_pack_planes_kernel(const float* src, int A, int B, long sa, long sb, long st, __bf16* dst, int nchunk,
                                          int ntile, int WN, const float* wamax, int transposed, int K) {
  const UnitTaps t = unit_taps(transposed, K, K / 2);
  pack_planes_body(src, A, B, sa, sb, st, dst, nchunk, ntile, WN, wamax, t);
}

}  // namespace

extern "C" int32_t mt_weight_pack_planes_wanted(int32_t AP, int32_t BP, int32_t T, int32_t precision) {
  const int K = T == 9 ? 3 : (T == 25 ? 5 : 0);
  return (mt_switches().planes && K && planes_shape_ok(K, K / 2, BP, AP, precision)) ? 1 : 0;
}

extern "C" int64_t mt_weight_pack_planes_elems(int32_t AP, int32_t BP, int32_t T, int32_t transposed) {
  const int K = T == 9 ? 3 : 5;
  const UnitTaps t = unit_taps(transposed, K, K / 2);
  const int BN = AP > 64 ? 128 : 64;
  const long nf = (long)(t.ntaps[0] + t.ntaps[1] + t.ntaps[2] + t.ntaps[3]) * (BP / 32);
  return nf * ((AP + BN - 1) / BN) * (BN / 32) * 2 * 2 * 64 * 8;           // [frag][ntile][wn][kstep][plane][lane][8]
}

extern "C" int mt_weight_pack_planes(const float* src, int32_t A, int32_t B, int32_t T, int64_t sa, int64_t sb, int64_t st,
                                     void* dst, int32_t AP, int32_t BP, int32_t precision, int32_t transposed,
                                     const float* w_amax, void* stream) {
  MT_REQUIRE(src && dst && A <= AP && B <= BP && (T == 9 || T == 25) && (reinterpret_cast<uintptr_t>(dst) & 15) == 0,
             "mt_weight_pack_planes: bad arguments");
  const int K = T == 9 ? 3 : 5;
  MT_REQUIRE(planes_shape_ok(K, K / 2, BP, AP, precision), "mt_weight_pack_planes: shape / precision not taken by the planes kernel");
  MT_REQUIRE((precision == MT_PREC_F16X3) == (w_amax != nullptr), "mt_weight_pack_planes: w_amax goes with F16X3");
  const int BN = AP > 64 ? 128 : 64, nchunk = BP / 32, ntile = (AP + BN - 1) / BN;
  const long tot = mt_weight_pack_planes_elems(AP, BP, T, transposed) / 2;
  hipLaunchKernelGGL(weight_pack_planes_kernel, dim3(mt_grid_1d(tot, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                     A, B, (long)sa, (long)sb, (long)st, reinterpret_cast<__bf16*>(dst), nchunk, ntile, BN / 32, w_amax,
                     transposed, K);
  return mt_check_launch("mt_weight_pack_planes");
}

// Called by mt_conv_run: returns 1 if the launch was taken.
int mt_halo_planes_try_launch(const mt_conv* d, void* stream) {
  if (d->wt_layout != MT_WT_PLANES || !mt_switches().planes) return 0;
  if (d->stride != 2 || d->KH != d->KW || d->ups != 0 || !planes_shape_ok(d->KH, d->pad, d->x.C, d->y.C, d->precision)) return 0;
  if (d->precision == MT_PREC_F16X3 && (!d->x_amax || !d->w_amax)) return 0;
  ConvK k;
  if (fill_convk(d, &k) != MT_OK) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k.yamax) k.ascr = mt_amax_scratch(s);
  const int fmt = d->precision == MT_PREC_F16X3 ? 1 : 0;
  const UnitTaps t = unit_taps(d->transposed, d->KH, d->pad);
  const int nchunk = k.C / 32;
  PlanesK q = {};
  auto launch = [&](int Hc, int Wc, int nz) {
    q.Hc = Hc; q.Wc = Wc;
    const int tiles_y = (Hc + 7) / 8, tiles_x = (Wc + 15) / 16;
    const unsigned nt = (unsigned)k.N * tiles_y * tiles_x;
    if (nt == 0) return;
    const unsigned ntn = k.Co > 64 ? (k.Co + 127) / 128 : 1;
    ConvK kk = k;
    kk.amax_deferred = kk.yamax && kk.ascr && nt * ntn * nz >= 2 * MT_AMAX_DEFER_BLOCKS;   // (two blocks per CU: from four rounds on)
    if (k.Co > 64) {
      if (fmt) hipLaunchKernelGGL((igemm_halo_planes_kernel<1, 1>), dim3(nt, ntn, nz), dim3(256), 0, s, kk, q, tiles_y, tiles_x);
      else hipLaunchKernelGGL((igemm_halo_planes_kernel<0, 1>), dim3(nt, ntn, nz), dim3(256), 0, s, kk, q, tiles_y, tiles_x);
    } else {
      if (fmt) hipLaunchKernelGGL((igemm_halo_planes_kernel<1, 2>), dim3(nt, 1, nz), dim3(256), 0, s, kk, q, tiles_y, tiles_x);
      else hipLaunchKernelGGL((igemm_halo_planes_kernel<0, 2>), dim3(nt, 1, nz), dim3(256), 0, s, kk, q, tiles_y, tiles_x);
    }
    if (kk.amax_deferred) mt_amax_fold_launch(kk.ascr, kk.yamax, s);
  };
  if (!d->transposed) {
    // one launch over the output; the four parity planes of X are the units
    q.nunits = 4; q.xsh = 2 * k.xsh; q.xsw = 2 * k.xsw; q.omul = 1; q.oya = 0; q.oxa = 0;
    int f0 = 0;
    for (int u = 0; u < 4; ++u) {
      const int a = u >> 1, b = u & 1;
      q.u[u].xoff = (long)a * k.xsh + (long)b * k.xsw;
      q.u[u].H = (k.H - a + 1) / 2; q.u[u].W = (k.W - b + 1) / 2;
      q.u[u].mask = t.mask[u];
      q.u[u].frag0 = f0;
      f0 += t.ntaps[u] * nchunk;
    }
    launch(k.Ho, k.Wo, 1);
  } else {
    // ONE launch, blockIdx.z = output parity class (A, B): one unit each over the whole X
    q.nunits = 1; q.xsh = k.xsh; q.xsw = k.xsw; q.omul = 2; q.zclasses = 1;
    int f0 = 0;
    for (int c = 0; c < 4; ++c) {
      q.u[c].xoff = 0; q.u[c].H = k.H; q.u[c].W = k.W;
      q.u[c].mask = t.mask[c];
      q.u[c].frag0 = f0;
      f0 += t.ntaps[c] * nchunk;
    }
    launch((k.Ho + 1) / 2, (k.Wo + 1) / 2, 4);
  }
  return 1;
}
