// Stride-1 5x5 "same" convolutions with a 1-D Winograd F(2, 5) transform along W (round 3): the 128 -> 128
// layers of FlowEstimator (model_dfpn.py:686-711, and with tap-flipped, channel-swapped weights their
// backward-data) and the 256 -> 256 entry of the correlation mixer (model_dfpn.py:638-660).
//
//   y[oy][2t + j] = sum_ky  A^T ( (G g[ky][:]) .* (B^T x[oy + ky - 2][2t - 2 .. 2t + 3]) ),   j = 0, 1
//
// with the interpolation points 0, 1, -1, 2, -2, inf:
//   B^T = [ 4  0 -5  0  1  0 ]      G = 1/24 [  6   0   0   0   0 ]      A^T = [ 1  1  1  1  1  0 ]
//         [ 0 -4 -4  1  1  0 ]               [ -4  -4  -4  -4  -4 ]            [ 0  1 -1  2 -2  1 ]
//         [ 0  4 -4 -1  1  0 ]               [ -4   4  -4   4  -4 ]
//         [ 0 -2 -1  2  1  0 ]               [  1   2   4   8  16 ]
//         [ 0  2 -1 -2  1  0 ]               [  1  -2   4  -8  16 ]
//         [ 0  4  0 -5  0  1 ]               [  0   0   0   0  24 ]
// Two output pixels of a row cost 5 kernel rows x 6 products instead of 5 x 5 x 2: 1.67x fewer MFMAs.  In the
// engine's arithmetic (operands split into two fp16 terms AFTER the transform, exact products, fp32 accumulate)
// the form is within 1.2 - 1.5x of the direct form's error against fp64 (scratch/r3_wino25_cond.py); |B^T d| <=
// 10 max|x|, so the transformed image is stored scaled by a further 2^-4 to stay inside fp16's range.
//
// Structure = igemm_wino16_kernel (conv_halo_wino.hip): one 512-thread block per CU, the transformed image
//   V[plane][row 12][xi 6][pair 8][32 ch]      (64-byte entries, rows padded by 16 B)
// of an (8 + 4) x 20 halo double-buffered in LDS (148 KB), weight fragments straight from global memory in MFMA
// operand order two taps ahead, the next chunk's image built in rounds under the current chunk's taps, one
// barrier per 32-channel chunk.  What differs: six xi mean six accumulator sets per (pixel pair, channel) --
// 192 registers for the 64 x 32 wave tile of the 3 x 3 kernel.  So the eight waves of a block split the XI axis
// instead of the pixel rows: waves = 2 xi halves x 4 channel quarters, a wave owns all 64 (row, pair) GEMM rows
// of the 8 x 16 tile x 32 channels x three xi (96 accumulator registers) and walks only its 15 of the 30
// (ky, xi) taps; the two xi halves of a channel quarter meet in the epilogue, where each reduces its three xi to
// the two partial sums of the output transform and they exchange them through LDS (the image is dead by then).
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"

namespace {

#ifndef MT_W25_SCALAR_XFORM
#define MT_W25_SCALAR_XFORM 0         // (1: the input transform without packed-fp32 instructions; needs -fno-slp-vectorize)
#endif
#ifndef MT_W25_SPLIT
#define MT_W25_SPLIT MT_SHADOW_DEFAULT   // split form of the transformed values (conv_common.h: split4f SHADOW)
#endif

template <int FMT>
__global__ __launch_bounds__(512, 2) void igemm_wino25_kernel(ConvK p, int tiles_y, int tiles_x) {
  constexpr int NS = 2, CK = 32, TH = 8, TW = 16, BN = 128, TM = 2, NXI = 6, KY = 5;
  constexpr int NTX = TW / 2, ROWS = TH + 4;
  constexpr int ENT = CK, ROW_E = NXI * NTX * ENT + 8, PLANE_E = ROWS * ROW_E, BUF_E = NS * PLANE_E;
  constexpr int NIT = ROWS * NTX * (CK / 4), RI = (NIT + 511) / 512;      // 768 items, 2 rounds (the second half full)
  constexpr int TW_ = KY * 3, KSTEPS = CK / 16;                            // taps of one wave
  static_assert(RI == 2, "two transform rounds per chunk");
  static_assert(2 * BUF_E * 2 <= 160 * 1024, "two images must fit the LDS");
  __shared__ __attribute__((aligned(16))) __bf16 sV[2 * BUF_E];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xh = wave >> 2, wn = wave & 3;                 // xi half, channel quarter
  const int li = lane & 31, lh = lane >> 5;
  // 1-D grid over (pixel tile, n-tile), as igemm_wino16_kernel: the n-tiles of one pixel tile are consecutive
  // blocks of ONE XCD, each XCD owns a contiguous band of pixel tiles
  const int ntn = (p.Co + BN - 1) / BN, npt = p.N * tiles_y * tiles_x;
  const int lid = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  int bid, nty;
  {
    const int q8 = npt >> 3, r8 = npt & 7;
    const int mine = q8 + (xcd < r8 ? 1 : 0), start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int tl = lid / ntn;
    nty = lid - tl * ntn;
    if (tl >= mine) {            // (a padding block: it still takes part in the launch's max reduction)
      finish_amax(p, 0.f, sV);
      return;
    }
    bid = start + tl;
  }
  const int tx_ = bid % tiles_x; bid /= tiles_x;
  const int ty_ = bid % tiles_y;
  const int n = bid / tiles_y;
  const int ty0 = ty_ * TH, tx0 = tx_ * TW, n0 = nty * BN;
  const int HL = p.H << p.ups, WL = p.W << p.ups;      // (ups: the input is read through a x2 nearest up-sampling)
  // the transformed image carries a further 2^-4 (|B^T d| <= 10 max|x|)
  // (bf16 planes keep fp32's exponent range and are not scaled at all)
  const float scA = FMT ? pow2_scale(*p.xamax) * 0.0625f : 1.f;
  const float inv = FMT ? 1.f / (scA * pow2_scale(*p.wamax)) : 1.f;
  const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;
  const int xsw = (int)p.xsw;

  // transform items (halo row r, column pair t, channel quad c4), c4 fastest; round i owns q = tid + 512 i
  int it_row[RI], it_lx[RI], it_dst[RI];
  unsigned it_ok[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int q = tid + 512 * i;
    const int c4 = q & 7, t = (q >> 3) & 7, r = q >> 6;
    const int ly = ty0 - 2 + r;
    it_ok[i] = 0;
    it_lx[i] = tx0 - 2 + 2 * t;
    it_row[i] = (int)((long)(ly >> p.ups) * p.xsh) + 4 * c4;
    it_dst[i] = r * ROW_E + t * ENT + 4 * c4;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int lx = it_lx[i] + b;
      const bool ok = q < NIT && (unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL;
      it_ok[i] |= ok ? (1u << b) : 0u;
    }
    if (q < NIT) it_ok[i] |= 64u;                       // bit 6: the item exists
  }
  auto load_item = [&](f32x4 (&d)[6], int i, int c0) {
    const int c4 = (tid + 512 * i) & 7;
    const bool cin = c0 + 4 * c4 < p.C;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (((it_ok[i] >> b) & 1u) && cin)
        v = *reinterpret_cast<const f32x4*>(xin + (it_row[i] + ((it_lx[i] + b) >> p.ups) * xsw + c0));
      d[b] = v;
    }
  };
  auto store_item = [&](const f32x4 (&d)[6], int i, int buf) {
    if (it_ok[i] & 64u) {
      // B^T d with shared terms: 12 multiply-adds for the six xi
#if MT_W25_SCALAR_XFORM
      // (round 4) element by element in plain fp32 instructions: no v_pk_*_f32 on the pipe the other wave's MFMAs need
      f32x4 v[6];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d0 = d[0][j], d1 = d[1][j], d2 = d[2][j], d3 = d[3][j], d4 = d[4][j], d5 = d[5][j];
        const float t0 = __builtin_fmaf(-4.f, d2, d4), t1 = __builtin_fmaf(-4.f, d1, d3), t2 = d4 - d2, t3 = d3 - d1;
        v[0][j] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
        v[1][j] = t0 + t1;
        v[2][j] = t0 - t1;
        v[3][j] = __builtin_fmaf(2.f, t3, t2);
        v[4][j] = __builtin_fmaf(-2.f, t3, t2);
        v[5][j] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
      }
#else
      const f32x4 t0 = d[4] - 4.f * d[2], t1 = d[3] - 4.f * d[1], t2 = d[4] - d[2], t3 = d[3] - d[1];
      const f32x4 v[6] = {4.f * d[0] - 5.f * d[2] + d[4], t0 + t1, t0 - t1, t2 + 2.f * t3, t2 - 2.f * t3,
                          4.f * d[1] - 5.f * d[3] + d[5]};
#endif
      __bf16* dst = sV + buf * BUF_E + it_dst[i];
#pragma unroll
      for (int xi = 0; xi < NXI; ++xi) {
        bf16x4 parts[NS];
        split4f<NS, FMT, MT_W25_SPLIT>(v[xi], scA, parts);
#pragma unroll
        for (int s = 0; s < NS; ++s)
          *reinterpret_cast<bf16x4*>(dst + s * PLANE_E + xi * NTX * ENT) = parts[s];
      }
    }
  };
  // weight fragments: [tap 30 = ky * 6 + xi][chunk][ntile][wn 4][kstep][plane][lane] x 16 B
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK, ntile_w = (p.Co + BN - 1) / BN;
  bf16x8 fb[2][KSTEPS][NS];
  // wave tap w (0 .. 14) = kernel row w / 3, xi 3 xh + w % 3
  auto load_B = [&](bf16x8 (&f)[NS], int ch, int w, int ks) {
    const int tap = (w / 3) * NXI + 3 * xh + (w % 3);
    const bf16x8* g = wfr + ((((long)tap * nchunk + ch) * ntile_w + nty) * 4 + wn) * (KSTEPS * NS * 64) +
                      ks * NS * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) f[s] = g[s * 64];
  };

  // ---- prologue: the whole first image (the accumulators are not live yet) ---------------------
  {
    f32x4 r0[6], r1[6];
    load_item(r0, 0, 0);
    load_item(r1, 1, 0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) { load_B(fb[0][ks], 0, 0, ks); load_B(fb[1][ks], 0, 1, ks); }
    store_item(r0, 0, 0);
    store_item(r1, 1, 0);
  }
  f32x16 acc[TM][3];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int xl = 0; xl < 3; ++xl)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][xl][r] = 0.f;
  // MFMA row i of M tile a = (output row 4 a + (i >> 3), column pair i & 7)
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = (4 * a + (li >> 3)) * ROW_E + (li & 7) * ENT + 8 * lh + 3 * xh * NTX * ENT;
  __syncthreads();

  f32x4 rh[6];
  // A wave walks 15 taps per chunk -- an odd number -- so the fragment slot of a tap alternates from chunk to chunk:
  // the chunk body is instantiated for both parities (PAR = chunk & 1) and the slot index stays a compile-time constant.
  auto chunk = [&](int ch, auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    const bool more = ch + 1 < nchunk;
    const __bf16* vb = sV + PAR * BUF_E;
    const int nb = PAR ^ 1;
#pragma unroll
    for (int w = 0; w < TW_; ++w) {
      const int ky = w / 3, xl = w % 3;
      const int slot = (w + PAR) & 1;                  // global tap counter 15 ch + w, modulo 2
      const int toff = ky * ROW_E + xl * NTX * ENT;
      // the next chunk's image: round 0 is loaded at tap 0 and written at tap 6, round 1 at taps 7 / 13
      if ((w == 0 || w == 7) && more) load_item(rh, w == 0 ? 0 : 1, (ch + 1) * CK);
      bf16x8 fa[KSTEPS][TM][NS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int s = 0; s < NS; ++s)
            fa[ks][a][s] = *reinterpret_cast<const bf16x8*>(&vb[s * PLANE_E + a_base[a] + toff + ks * 16]);
      __builtin_amdgcn_sched_barrier(0);     // (all fragment reads of the tap are requested ahead of its MFMAs)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
        for (int a = 0; a < TM; ++a) mfma_splitf<NS, FMT>(fa[ks][a], fb[slot][ks], acc[a][xl]);
        // this slot's next contents: the same k-step two taps on
        if (w + 2 < TW_) load_B(fb[slot][ks], ch, w + 2, ks);
        else if (more) load_B(fb[slot][ks], ch + 1, w + 2 - TW_, ks);
      }
      if ((w == 6 || w == 13) && more) store_item(rh, w == 6 ? 0 : 1, nb);
    }
    if (more) __syncthreads();      // the other image is complete, nobody reads this one any more
  };
  for (int ch = 0; ch < nchunk; ch += 2) {
    chunk(ch, std::integral_constant<int, 0>{});
    if (ch + 1 < nchunk) chunk(ch + 1, std::integral_constant<int, 1>{});
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------
  // Output transform  y0 = m0 + m1 + m2 + m3 + m4,  y1 = m1 - m2 + 2 m3 - 2 m4 + m5.  xi half 0 holds m0 m1 m2, half 1
  // m3 m4 m5: each reduces them to its share (P, Q) of (y0, y1); half 0 finishes M tile 0, half 1 finishes M tile 1, and
  // each hands the other the shares of the tile it does not finish: 32 floats per lane through LDS, lane-contiguous.
  // (wave-uniform branches on the xi half: no register copies of the accumulators, no selects)
  auto share = [&](const f32x16 (&m)[3], int r, float& P, float& Q) {
    if (xh == 0) { P = (m[0][r] + m[1][r]) + m[2][r]; Q = m[1][r] - m[2][r]; }
    else { P = m[0][r] + m[1][r]; Q = 2.f * (m[0][r] - m[1][r]) + m[2][r]; }
  };
  __syncthreads();                    // every wave is done with the last image: the LDS is free
  float* xch = reinterpret_cast<float*>(sV);
  {
    float* mine = xch + (wave * 32) * 64 + lane;            // [wave][32 values][64 lanes]
    auto put = [&](const f32x16 (&m)[3]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float P, Q;
        share(m, r, P, Q);
        mine[r * 64] = P;
        mine[(16 + r) * 64] = Q;
      }
    };
    if (xh == 0) put(acc[1]); else put(acc[0]);             // the M tile the OTHER half finishes
  }
  __syncthreads();
  const float* theirs = xch + ((wave ^ 4) * 32) * 64 + lane;
  float vals[32];
  {
    auto take = [&](const f32x16 (&m)[3]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float P, Q;
        share(m, r, P, Q);
        vals[r * 2] = (P + theirs[r * 64]) * inv;
        vals[r * 2 + 1] = (Q + theirs[(16 + r) * 64]) * inv;
      }
    };
    if (xh == 0) take(acc[0]); else take(acc[1]);
  }
  const int o = n0 + wn * 32 + li;
  const int oyb = ty0 + 4 * xh, oxb = tx0 + 8 * lh;          // this wave finishes M tile a = xh: output rows 4 xh ..
  // element (r, j): C/D row (r & 3) + 8 (r >> 2) + 4 lh -> output row r >> 2 of the M tile, column pair (r & 3) + 4 lh
  const float vmax = halo_epilogue<32>(p, n, o, vals, [&](int e, int& oy, int& ox) {
    const int r = e >> 1, j = e & 1;
    oy = oyb + (r >> 2);
    ox = oxb + 2 * (r & 3) + j;
  });
  finish_amax(p, vmax, sV);
}

__global__ void weight_pack_halo_wino25_kernel(const float* src, int A, int B, long sa, long sb, long st, __bf16* dst,
                                               int nchunk, int ntile, int flip, const float* wamax, int nchunk_total,
                                               int ch_off) {
  pack_halo_wino25_body(src, A, B, sa, sb, st, dst, nchunk, ntile, flip, wamax, nchunk_total, ch_off);
}

}  // namespace

void mt_halo_wino25_launch(const void* kp, int fmt, void* stream) {
  const ConvK& k = *static_cast<const ConvK*>(kp);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_y = (k.Ho + 7) / 8, tiles_x = (k.Wo + 15) / 16;
  const unsigned nt = (unsigned)k.N * tiles_y * tiles_x, ntn = (k.Co + 127) / 128;
  const unsigned grid = 8 * ((nt + 7) / 8) * ntn;       // (padded so that every XCD gets the same number of block slots)
  (void)fmt;                                            // (halo_wino25_wanted admits the fp16 split only)
  ConvK kk = k;
  kk.amax_deferred = kk.yamax && kk.ascr && grid >= MT_AMAX_DEFER_BLOCKS;   // (several rounds of blocks: fold the slots afterwards)
  hipLaunchKernelGGL((igemm_wino25_kernel<1>), dim3(grid), dim3(512), 0, s, kk, tiles_y, tiles_x);
  if (kk.amax_deferred) mt_amax_fold_launch(kk.ascr, kk.yamax, s);
}

void mt_pack_halo_wino25_launch(const float* src, int A, int B, long sa, long sb, long st, void* dst, int nchunk, int ntile,
                                int flip, const float* wamax, int nchunk_total, int ch_off, void* stream) {
  const long tot = (long)30 * nchunk * ntile * 4 * 1024;
  hipLaunchKernelGGL(weight_pack_halo_wino25_kernel, dim3(mt_grid_1d(tot, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     src, A, B, sa, sb, st, reinterpret_cast<__bf16*>(dst), nchunk, ntile, flip, wamax, nchunk_total, ch_off);
}
