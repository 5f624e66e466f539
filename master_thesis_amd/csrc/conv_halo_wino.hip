// Stride-1 3x3 "same" convolutions with a 1-D Winograd F(2, 3) transform along W on top of the
// register-B halo-tile kernel (conv_halo.hip): the layers that carry most of the DFPN / CHN FLOPs
// (SeparableConv4d, FlowEstimator, VGG: model_dfpn.py:596-605,686-711, model_vgg.py:40-41, and with
// tap-flipped, channel-swapped weights their backward-data).
//
//   y[oy][2t + j] = sum_ky  A^T ( (G g[ky][:]) .* (B^T x[oy + ky - 1][2t - 1 .. 2t + 2]) )
//   B^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)     G g = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)
//   A^T m = (m0 + m1 + m2, m1 - m2 - m3)
//
// Two output pixels cost 3 rows x 4 products instead of 3 x 3 x 2: 1.5x fewer MFMAs, with
// transforms that only add (no amplification beyond 2x, so the split-fp16 operands keep their
// per-tensor power-of-two scales).  The block owns 8 x 16 output pixels = 8 rows x 8 column pairs:
// the GEMM's M index is (row, pair), its "taps" are the 12 (ky, xi) combinations, and the four
// xi of a pair keep separate accumulators until the epilogue's output transform (in registers: all
// four sit in the same lane).  Per 32-channel K chunk the (8 + 2) x 18 input halo is loaded once,
// transformed and split into two 16-bit planes once, and written as the image
//   V[plane][row 10][xi 4][pair 8][32 ch]      (64-byte entries, rows padded by 16 B)
// whose ds_read_b128 fragment reads are bank-conflict free; weight fragments come straight from
// global memory in MFMA operand order (mt_weight_pack_halo writes G g in that order).
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"
#include "dispatch.h"

namespace {

template <int FMT, int WM>
__global__ __launch_bounds__(256, 2) void igemm_halo_wino_kernel(ConvK p, int tiles_y, int tiles_x) {
  constexpr int NS = 2, CK = 32, TH = 8, TW = 16, WN = 4 / WM, BN = WN * 32, TM = 2 / WM;
  constexpr int NTX = TW / 2, ROWS = TH + 2;
  constexpr int ENT = CK;                              // bf16 per entry (64 B)
  constexpr int ROW_E = 4 * NTX * ENT + 8;             // bf16 per V row: 32 entries + 16 B pad
  constexpr int PLANE_E = ROWS * ROW_E;
  constexpr int NIT = ROWS * NTX * (CK / 4), RI = (NIT + 255) / 256;
  constexpr int T = 12, KSTEPS = CK / 16;
  __shared__ __attribute__((aligned(16))) __bf16 sV[NS * PLANE_E];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx_ = bid % tiles_x; bid /= tiles_x;
  const int ty_ = bid % tiles_y;
  const int n = bid / tiles_y;
  const int ty0 = ty_ * TH, tx0 = tx_ * TW, n0 = blockIdx.y * BN;
  const int HL = p.H << p.ups, WL = p.W << p.ups;
  const float scA = FMT ? pow2_scale(*p.xamax) : 1.f;
  const float inv = FMT ? 1.f / (scA * pow2_scale(*p.wamax)) : 1.f;
  const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;

  // ---- transform items: (halo row r, column pair t, channel quad c4), c4 fastest --------------
  // (32-bit element offsets inside one image; the four pixels of an item are re-derived per load)
  int it_row[RI], it_lx[RI];
  unsigned it_ok[RI];                                   // bit b: pixel b of the item lies inside the image
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int q = tid + 256 * i;
    const int c4 = q & 7, t = (q >> 3) & 7, r = q >> 6;
    const int ly = ty0 - 1 + r;
    it_ok[i] = 0;
    it_lx[i] = tx0 - 1 + 2 * t;
    it_row[i] = (int)((long)(ly >> p.ups) * p.xsh) + 4 * c4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int lx = it_lx[i] + b;
      const bool ok = q < NIT && (unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL;
      it_ok[i] |= ok ? (1u << b) : 0u;
    }
  }
  const int xsw = (int)p.xsw;
  f32x4 rh[RI][4];
  auto load_raw = [&](int c0) {
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int c4 = (tid + 256 * i) & 7;
      const bool cin = c0 + 4 * c4 < p.C;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (((it_ok[i] >> b) & 1u) && cin)
          v = *reinterpret_cast<const f32x4*>(xin + (it_row[i] + ((it_lx[i] + b) >> p.ups) * xsw + c0));
        rh[i][b] = v;
      }
    }
  };
  auto store_v = [&]() {
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int q = tid + 256 * i;
      if (q < NIT) {
        const int c4 = q & 7, t = (q >> 3) & 7, r = q >> 6;
        const f32x4 d0 = rh[i][0], d1 = rh[i][1], d2 = rh[i][2], d3 = rh[i][3];
        const f32x4 v[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
        __bf16* dst = sV + r * ROW_E + t * ENT + 4 * c4;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
          bf16x4 parts[NS];
          split4f<NS, FMT>(v[xi], scA, parts);
#pragma unroll
          for (int s = 0; s < NS; ++s)
            *reinterpret_cast<bf16x4*>(dst + s * PLANE_E + xi * NTX * ENT) = parts[s];
        }
      }
    }
  };
  // weight fragments: [tap 12][chunk][ntile][wn][kstep][plane][lane] x 16 B  (tap = ky * 4 + xi)
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK, ntile_w = (p.Co + BN - 1) / BN;
  bf16x8 fb[KSTEPS][NS];
  auto load_B = [&](int ch, int tap, int ks) {
    const bf16x8* f = wfr + ((((long)tap * nchunk + ch) * ntile_w + blockIdx.y) * WN + wn) * (KSTEPS * NS * 64) +
                      ks * NS * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) fb[ks][s] = f[s * 64];
  };

  f32x16 acc[TM][4];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][xi][r] = 0.f;

  // MFMA row i of M tile a = (output row 4a + (i >> 3), column pair i & 7)
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = (4 * (wm * TM + a) + (li >> 3)) * ROW_E + (li & 7) * ENT + 8 * lh;

  load_raw(0);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) load_B(0, 0, ks);
  store_v();
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const bool more = ch + 1 < nchunk;
    // fully unrolled: the accumulator of a tap (xi = tap & 3) must be a compile-time choice
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
      const int ky = tap >> 2, xi = tap & 3;
      const bool last = tap == T - 1;
      const int toff = ky * ROW_E + xi * NTX * ENT;
      bf16x8 fa[KSTEPS][TM][NS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int s = 0; s < NS; ++s)
            fa[ks][a][s] = *reinterpret_cast<const bf16x8*>(&sV[s * PLANE_E + a_base[a] + toff + ks * 16]);
      __builtin_amdgcn_sched_barrier(0);     // (all fragment reads of the tap are requested ahead of its MFMAs)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
        for (int a = 0; a < TM; ++a) mfma_splitf<NS, FMT>(fa[ks][a], fb[ks], acc[a][xi]);
        if (!last) load_B(ch, tap + 1, ks);
        else if (more) load_B(ch + 1, 0, ks);
      }
      if (tap == T - 4 && more) load_raw((ch + 1) * CK);     // in flight under the last taps
    }
    if (more) {
      __syncthreads();            // every wave is done reading this chunk's image
      store_v();
      __syncthreads();
    }
  }

  // ---- epilogue: output transform (in registers) + the fused ops ------------------------------
  const int o = n0 + wn * 32 + li;
  const int oyb = ty0 + 4 * wm * TM, oxb = tx0 + 8 * lh;
  float vmax = 0.f;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {            // (16 values at a time: this kernel has no registers to spare)
      float vals[16];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = 8 * h + q;
        const float m0 = acc[a][0][r], m1 = acc[a][1][r], m2 = acc[a][2][r], m3 = acc[a][3][r];
        vals[q * 2] = (m0 + m1 + m2) * inv;
        vals[q * 2 + 1] = (m1 - m2 - m3) * inv;
      }
      vmax = fmaxf(vmax, halo_epilogue<16>(p, n, o, vals, [&](int e, int& oy, int& ox) {
        const int r = 8 * h + (e >> 1), j = e & 1;
        oy = oyb + 4 * a + (r >> 2);
        ox = oxb + 2 * (r & 3) + j;
      }));
    }
  }
  finish_amax(p, vmax, sV);
}

// ---- <= 64 output channels: the XI axis split over the waves (round 3) ------------------------
// The dense blocks of the RRDB trunk (model_chn.py:588-623) are 3 x 3 layers with 64 .. 192 gathered and only 32 or
// 64 output channels, at 64 x 64 pixels: ~45 us launches of one wave of blocks.  With a 32-wide n-tile the four waves
// of a block cannot split the channels, and splitting the pixels (igemm_halo_kernel<.., 4, 1, 1, 1>, igemm_halo_wino_
// kernel<., 2>) leaves a wave ONE 32 x 32 tile per fragment pair: every A and every B fragment is fetched for a single
// MFMA triple, with a barrier per tap.  Here the waves split the four xi of the Winograd transform instead: wave w
// owns xi = w for ALL 64 (row, pair) GEMM rows of the 8 x 16 tile and all TN 32-channel tiles -- each A fragment pair
// feeds TN triples, each B pair two, 12 TN MFMAs per (kernel row, xi) tap, 16 TN accumulator registers -- and walks
// only its 3 of the 12 taps; weight fragments come straight from global memory (every wave reads different ones:
// nothing is fetched twice by a block), the image is single-buffered (41 KB: three blocks per CU), two barriers per
// 32-channel chunk.  The four xi meet in the epilogue through LDS: per M tile every wave posts its products, then
// finishes one output row of the tile (y0 = m0 + m1 + m2, y1 = m1 - m2 - m3).
#if MT_WINOX_TRACE            // (lab: wall-clock stamps of the stages of every block, read back by mt_lab_winox_trace)
__device__ unsigned long long winox_trace[4096 * 16];
#define WX_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) winox_trace[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define WX_STAMP(i) do {} while (0)
#endif
#ifndef MT_WINOX_SHADOW       // split form of the transform (conv_common.h: split4f SHADOW): 1 = no instruction that shares the matrix pipe
#define MT_WINOX_SHADOW 2
#endif
#ifndef MT_WINOX_ROTATE        // 1 (lab, no gain): the half transform round alternates between the wave pairs from block to block
#define MT_WINOX_ROTATE 0
#endif
#ifndef MT_WINOX_OCC           // blocks per CU the 32-wide variant is compiled for (3: 168 registers)
#define MT_WINOX_OCC 3
#endif
#ifndef MT_WINOX_PIPE          // 1 (lab, a tie): 32-wide tiles prefetch the next step's image fragments (two register sets)
#define MT_WINOX_PIPE 0
#endif
#ifndef MT_WINOX_TOUCH         // 1 (lab, a tie): 32-wide tiles touch the halo of chunk c + 2 during chunk c (one dword per 128-byte line: L2 prefetch)
#define MT_WINOX_TOUCH 0
#endif
#ifndef MT_WINOX_AHEAD         // 1: 32-wide tiles request the halo two chunks ahead (two register sets)
#define MT_WINOX_AHEAD 0       // (measured: 168 VGPRs at three blocks per CU -> 77 spilled)
#endif
#ifndef MT_WINOX_ABL          // (lab ablations: 1 no halo loads, 2 no transform / split / image stores, 3 no MFMAs, 4 no weight loads, 5 no epilogue)
#define MT_WINOX_ABL 0
#endif
#ifndef MT_WINOX_TH4          // 1: launches of the 32-wide variant that would not fill two blocks per CU twice run 4 x 16-pixel tiles
#define MT_WINOX_TH4 0        // (twice the blocks, half the chain per block; four blocks per CU)
#endif
#ifndef MT_WINOX_TH4_BELOW    // ... below this many 8-row tiles
#define MT_WINOX_TH4_BELOW 1025
#endif
template <int FMT, int TN, int TH_ = 8>
__global__ __launch_bounds__(256, TH_ == 4 ? 4 : (TN == 1 ? MT_WINOX_OCC : 2)) void igemm_winox_kernel(ConvK p, int tiles_y, int tiles_x) {
  constexpr int NS = 2, CK = 32, TH = TH_, TW = 16, BN = 32 * TN, TM = TH / 4;
  constexpr int NTX = TW / 2, ROWS = TH + 2;
  constexpr int ENT = CK, ROW_E = 4 * NTX * ENT + 8, PLANE_E = ROWS * ROW_E;
  constexpr int NIT = ROWS * NTX * (CK / 4), RI = (NIT + 255) / 256;
  constexpr int KSTEPS = CK / 16;
  static_assert(4 * TN * 16 * 64 * 4 <= NS * PLANE_E * 2, "the epilogue exchange of one M tile must fit the image");
#ifndef MT_WINOX_PLANAR
#define MT_WINOX_PLANAR 0
#endif
#ifndef MT_WINOX_LDS_PAD      // (lab: extra LDS elements, to measure what one block per CU fewer costs)
#define MT_WINOX_LDS_PAD 0
#endif
  __shared__ __attribute__((aligned(16))) __bf16 sV[NS * PLANE_E + MT_WINOX_LDS_PAD];

  WX_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, xi = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx_ = bid % tiles_x; bid /= tiles_x;
  const int ty_ = bid % tiles_y;
  const int n = bid / tiles_y;
  const int ty0 = ty_ * TH, tx0 = tx_ * TW, n0 = blockIdx.y * BN;
  const int HL = p.H << p.ups, WL = p.W << p.ups;
  const float scA = FMT ? pow2_scale(*p.xamax) : 1.f;
  const float inv = FMT ? 1.f / (scA * pow2_scale(*p.wamax)) : 1.f;
  const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;

  // ---- transform items: (halo row r, column pair t, channel quad c4), c4 fastest (as igemm_halo_wino_kernel).
  // The halo comes through BUFFER loads: a wave-uniform descriptor of image n (SGPRs) + one 32-bit byte offset per
  // (item, pixel), computed once, + the chunk's channel offset as the scalar offset.  A pixel outside the image gets an
  // offset beyond the descriptor's range and reads as zero: no predicate, no branch, no zero-fill and no 64-bit address
  // arithmetic per load (the loop was VALU-bound: 508 VALU instructions per chunk against 36 MFMAs, a third of them these).
  const unsigned xbytes = (unsigned)(((long)(p.H - 1) * p.xsh + (long)(p.W - 1) * p.xsw + ((p.C + 3) & ~3)) * 4);
  const auto xsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, MT_WINOX_PLANAR ? 0x7fffffffu : xbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  // The last transform round is a half round (640 items on 256 threads): two of the four waves do three rounds, two do two.
  // Which two alternates with the block, so that the blocks that share a CU (three of them, one wave each per SIMD) do
  // not all put their extra round on the same two SIMDs (round 4: MT_WINOX_ROTATE).  (tid ^ 128 keeps the channel quad.)
  const int rot = MT_WINOX_ROTATE ? (int)(((blockIdx.x >> 3) & 1u) << 7) : 0;
  auto item_of = [&](int i) { return (i == RI - 1 ? (tid ^ rot) : tid) + 256 * i; };
  unsigned it_off[RI][4];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int q = item_of(i);
    const int c4 = q & 7, t = (q >> 3) & 7, r = q >> 6;
    const int ly = ty0 - 1 + r;
    const long row = (long)(ly >> p.ups) * p.xsh + 4 * c4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int lx = tx0 - 1 + 2 * t + b;
      const bool ok = q < NIT && (unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL;
      it_off[i][b] = ok ? (unsigned)((row + (long)(lx >> p.ups) * p.xsw) * 4) : OOB;
    }
  }
  // TN == 1 with MT_WINOX_AHEAD: TWO register sets -- the halo of chunk c + 2 is requested during chunk c, a whole chunk
  // (~2.5 us) before it is transformed, instead of ~0.5 us before: the transform between the two barriers of a chunk no
  // longer waits for an HBM round trip (round 4).  Set = chunk parity; every other path uses set 0 only.
  // (lab, MT_WINOX_PLANAR: the input as 32-channel planes [chunk][image][row][pixel][32] -- the view describes plane 0)
  const int plane_b = MT_WINOX_PLANAR ? (int)((long)p.N * p.xsn * 4) : 0;
  constexpr int NSET = (TN == 1 && MT_WINOX_AHEAD) ? 2 : 1;
  f32x4 rh2[NSET][RI][4];
  auto load_raw = [&](int c0, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    f32x4 (&rh)[RI][4] = rh2[SET];
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int c4 = (tid + 256 * i) & 7;
      const unsigned cmask = c0 + 4 * c4 < p.C ? 0u : OOB;       // (the channel tail of the last chunk)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
#if MT_WINOX_ABL != 1
        rh[i][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsr, it_off[i][b] | cmask, MT_WINOX_PLANAR ? (c0 >> 5) * plane_b : c0 * 4, 0));
#else
        rh[i][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
      }
    }
  };
  auto store_v = [&](auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    const f32x4 (&rh)[RI][4] = rh2[SET];
#if MT_WINOX_ABL == 2
    return;
#endif
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int q = item_of(i);
      if (q < NIT) {
        const int c4 = q & 7, t = (q >> 3) & 7, r = q >> 6;
        const f32x4 d0 = rh[i][0], d1 = rh[i][1], d2 = rh[i][2], d3 = rh[i][3];
        const f32x4 v[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
        __bf16* dst = sV + r * ROW_E + t * ENT + 4 * c4;
#pragma unroll
        for (int x4 = 0; x4 < 4; ++x4) {
          bf16x4 parts[NS];
          split4f<NS, FMT, MT_WINOX_SHADOW>(v[x4], scA, parts);
#pragma unroll
          for (int s = 0; s < NS; ++s)
            *reinterpret_cast<bf16x4*>(dst + s * PLANE_E + x4 * NTX * ENT) = parts[s];
        }
      }
    }
  };
  // weight fragments: [tap 12 = ky * 4 + xi][chunk][ntile][wn TN][kstep][plane][lane] x 16 B (mt_weight_pack_halo, order 2)
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK, ntile_w = (p.Co + BN - 1) / BN;
  bf16x8 fb[TN][KSTEPS][NS];
  auto load_B = [&](int ch, int ky, int tn, int ks) {
    const bf16x8* f = wfr + ((((long)(ky * 4 + xi) * nchunk + ch) * ntile_w + blockIdx.y) * TN + tn) * (KSTEPS * NS * 64) +
                      ks * NS * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) fb[tn][ks][s] = f[s * 64];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][tn][r] = 0.f;

  // MFMA row i of M tile a = (output row 4a + (i >> 3), column pair i & 7); this wave's xi plane of the image
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = (4 * a + (li >> 3)) * ROW_E + (li & 7) * ENT + 8 * lh + xi * NTX * ENT;

  // L2 touch of a later chunk's halo: thread t < ROWS * 18 reads one dword of halo pixel (t / 18, t % 18) -- one per 128-byte line of
  // a 32-channel chunk -- through the same range-checked descriptor; the value is only kept alive (round 4)
  float pf_touch = 0.f;
  unsigned touch_off = OOB;
  if (TN == 1 && MT_WINOX_TOUCH && tid < ROWS * (TW + 2)) {
    const int r = tid / (TW + 2), c = tid - r * (TW + 2);
    const int ly = ty0 - 1 + r, lx = tx0 - 1 + c;
    if ((unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL)
      touch_off = (unsigned)(((long)(ly >> p.ups) * p.xsh + (long)(lx >> p.ups) * p.xsw) * 4);
  }
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, NSET - 1>;
  load_raw(0, S0{});
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) load_B(0, 0, tn, ks);
  // what the epilogue reads that does not depend on the result: requested here, so that no load follows a store
  float bv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) bv[tn] = (p.bias && n0 + tn * 32 + li < p.Co) ? p.bias[n0 + tn * 32 + li] : 0.f;
  WX_STAMP(1);
  if (NSET == 2 && CK < p.C) load_raw(CK, S1{});           // (behind the first chunk's halo and fragments in the return order)
  store_v(S0{});
  WX_STAMP(2);
  __syncthreads();
  WX_STAMP(3);
  if constexpr (TN == 1) {
    // 32-wide tiles: TWO sets of weight fragments, each tap's requested two taps ahead (slot = tap parity; three taps per
    // chunk, so the parity of a chunk's first tap alternates: the chunk body exists once per parity), A fragments one k
    // step at a time (LDS latency is a tenth of the L2's).  Issue order matters, because loads return in order (vmcnt):
    // the next chunk's halo is requested AFTER the fragments of this chunk's last tap, so no tap waits for the halo --
    // with one set, fetched one tap ahead and behind the halo loads, a tap waited ~0.7 us for its fragments.
    bf16x8 fb2[KSTEPS][NS];                     // the second set (fb[0] is the first)
    auto load_B2 = [&](bf16x8 (&f)[KSTEPS][NS], int ch, int ky) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const bf16x8* g = wfr + (((long)(ky * 4 + xi) * nchunk + ch) * ntile_w + blockIdx.y) * (KSTEPS * NS * 64) + ks * NS * 64 + lane;
#pragma unroll
        for (int s = 0; s < NS; ++s) f[ks][s] = g[s * 64];
      }
    };
    load_B2(fb2, 0, 1);                         // tap 1 of chunk 0 (tap 0 is in fb[0] since the prologue)
#if MT_WINOX_PIPE
    // (round 4) the six (kernel row, k step) steps of a chunk as a software pipeline: the image fragments of step s + 1 are
    // requested BEFORE the MFMAs of step s are issued (two register sets), so a step's six MFMAs no longer start with an
    // exposed LDS round trip -- the block timeline shows a chunk's three taps taking 1.5 - 2.8 us for 0.4 us of matrix-pipe time
    bf16x8 faA[TM][NS], faB[TM][NS];
    auto read_fa = [&](bf16x8 (&f)[TM][NS], int step) {
      const int toff = (step >> 1) * ROW_E + (step & 1) * 16;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int s = 0; s < NS; ++s) f[a][s] = *reinterpret_cast<const bf16x8*>(&sV[s * PLANE_E + a_base[a] + toff]);
    };
    auto chunk = [&](auto parity, int ch) {
      constexpr int P = decltype(parity)::value;
      const bool more = ch + 1 < nchunk;
      read_fa(faA, 0);
      auto step = [&](auto sc, bf16x8 (&cur)[TM][NS], bf16x8 (&nxt)[TM][NS], bf16x8 (&bw)[KSTEPS][NS]) {
        constexpr int S = decltype(sc)::value;
        constexpr int ky = S >> 1, ks = S & 1;
        if constexpr (S + 1 < 6) read_fa(nxt, S + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < TM; ++a) mfma_splitf<NS, FMT>(cur[a], bw[ks], acc[a][0]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ks == 1) {
          // this slot's next contents: the tap two on; then the halo (behind the fragments of taps 1 and 2 in the return order)
          if (ky == 0) load_B2(bw, ch, 2);
          else if (more) load_B2(bw, ch + 1, ky - 1);
          if (ky == 0 && more) load_raw((ch + 1) * CK, S0{});
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
      if constexpr (P == 0) {
        step(I0{}, faA, faB, fb[0]); step(I1{}, faB, faA, fb[0]); step(I2{}, faA, faB, fb2); step(I3{}, faB, faA, fb2);
        step(I4{}, faA, faB, fb[0]); step(I5{}, faB, faA, fb[0]);
      } else {
        step(I0{}, faA, faB, fb2); step(I1{}, faB, faA, fb2); step(I2{}, faA, faB, fb[0]); step(I3{}, faB, faA, fb[0]);
        step(I4{}, faA, faB, fb2); step(I5{}, faB, faA, fb2);
      }
      if (more) {
        __syncthreads();            // every wave is done reading this chunk's image
        store_v(S0{});
        __syncthreads();
      }
    };
#else
    auto chunk = [&](auto parity, int ch) {
      constexpr int P = decltype(parity)::value;
      const bool more = ch + 1 < nchunk;
      auto tap = [&](auto kyc, bf16x8 (&cur)[KSTEPS][NS]) {
        constexpr int ky = decltype(kyc)::value;
        constexpr int toff = ky * ROW_E;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          bf16x8 fa[TM][NS];
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int s = 0; s < NS; ++s)
              fa[a][s] = *reinterpret_cast<const bf16x8*>(&sV[s * PLANE_E + a_base[a] + toff + ks * 16]);
#pragma unroll
#if MT_WINOX_ABL != 3
          for (int a = 0; a < TM; ++a) mfma_splitf<NS, FMT>(fa[a], cur[ks], acc[a][0]);
#else
          for (int a = 0; a < TM; ++a) acc[a][0][0] += (float)fa[a][0][0] + (float)cur[ks][0][0];
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
#if MT_WINOX_ABL != 4
        // this slot's next contents: the tap two on
        if (ky == 0) load_B2(cur, ch, 2);
        else if (more) load_B2(cur, ch + 1, ky - 1);
#endif
        // (the halo: behind the fragments of taps 1 and 2 in the return order)
        if constexpr (NSET == 2) { if (ky == 0 && ch + 2 < nchunk) load_raw((ch + 2) * CK, std::integral_constant<int, P>{}); }
        else { if (ky == 0 && more) load_raw((ch + 1) * CK, S0{}); }
#if MT_WINOX_TOUCH
        if (ky == 0 && (ch + 2) * CK < p.C)
          pf_touch = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsr, touch_off, (ch + 2) * CK * 4, 0));
#endif
        __builtin_amdgcn_sched_barrier(0);
      };
      using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
      if constexpr (P == 0) { tap(K0{}, fb[0]); tap(K1{}, fb2); tap(K2{}, fb[0]); }
      else { tap(K0{}, fb2); tap(K1{}, fb[0]); tap(K2{}, fb2); }
      if (ch == 0) WX_STAMP(4);
      if (more) {
        __syncthreads();            // every wave is done reading this chunk's image
        if (ch == 0) WX_STAMP(5);
        store_v(std::integral_constant<int, NSET == 2 ? (P ^ 1) : 0>{});
        if (ch == 0) WX_STAMP(6);
        __syncthreads();
        if (ch == 0) WX_STAMP(7);
      }
    };
#endif
    for (int ch = 0; ch < nchunk; ch += 2) {
      chunk(std::integral_constant<int, 0>{}, ch);
      if (ch + 1 < nchunk) chunk(std::integral_constant<int, 1>{}, ch + 1);
    }
  } else {
  for (int ch = 0; ch < nchunk; ++ch) {
      const bool more = ch + 1 < nchunk;
      if (more) load_raw((ch + 1) * CK, S0{});          // in flight under the three taps
  #pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int toff = ky * ROW_E;
        bf16x8 fa[KSTEPS][TM][NS];
  #pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
  #pragma unroll
          for (int a = 0; a < TM; ++a)
  #pragma unroll
            for (int s = 0; s < NS; ++s)
              fa[ks][a][s] = *reinterpret_cast<const bf16x8*>(&sV[s * PLANE_E + a_base[a] + toff + ks * 16]);
        __builtin_amdgcn_sched_barrier(0);     // (all fragment reads of the tap are requested ahead of its MFMAs)
  #pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
  #pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
  #pragma unroll
  #if MT_WINOX_ABL != 3
            for (int a = 0; a < TM; ++a) mfma_splitf<NS, FMT>(fa[ks][a], fb[tn][ks], acc[a][tn]);
  #else
            for (int a = 0; a < TM; ++a) acc[a][tn][0] += (float)fa[ks][a][0][0] + (float)fb[tn][ks][0][0];
  #endif
  #if MT_WINOX_ABL != 4
            if (ky < 2) load_B(ch, ky + 1, tn, ks);
            else if (more) load_B(ch + 1, 0, tn, ks);
  #endif
          }
        }
      }
      if (ch == 0) WX_STAMP(4);
      if (more) {
        __syncthreads();            // every wave is done reading this chunk's image
        if (ch == 0) WX_STAMP(5);
        store_v(S0{});
        if (ch == 0) WX_STAMP(6);
        __syncthreads();
        if (ch == 0) WX_STAMP(7);
      }
    }
}
  WX_STAMP(8);

#if MT_WINOX_ABL == 5          // (lab: no exchange, no epilogue)
  {
    float sacc = 0.f;
    for (int a = 0; a < TM; ++a) for (int tn = 0; tn < TN; ++tn) for (int r = 0; r < 16; ++r) sacc += acc[a][tn][r];
    if (sacc == 1.2345f) p.y[0] = sacc;
    return;
  }
#endif
  // ---- epilogue: the four xi meet through LDS, one M tile at a time; wave w finishes output row 4a + w of both
  // M tiles.  Two phases (conv_common.h: halo_epi_load / halo_epi_store): no load is issued after the first store.
  float* xch = reinterpret_cast<float*>(sV);       // [xi 4][tn][16][64 lanes]
  float vals[TN][8 * TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    __syncthreads();                               // the image (a == 0) / the previous tile's products are dead
    if (a == 0) WX_STAMP(11);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[((xi * TN + tn) * 16 + r) * 64 + lane] = acc[a][tn][r];
    __syncthreads();
    if (a == 0) WX_STAMP(12); else WX_STAMP(14);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * xi + q;                  // C/D rows (r & 3) + 8 (r >> 2) + 4 lh: output row r >> 2 = xi of the M tile
        const float m0 = xch[((0 * TN + tn) * 16 + r) * 64 + lane], m1 = xch[((1 * TN + tn) * 16 + r) * 64 + lane];
        const float m2 = xch[((2 * TN + tn) * 16 + r) * 64 + lane], m3 = xch[((3 * TN + tn) * 16 + r) * 64 + lane];
        vals[tn][8 * a + 2 * q] = (m0 + m1 + m2) * inv;
        vals[tn][8 * a + 2 * q + 1] = (m1 - m2 - m3) * inv;
      }
    if (a == 0) WX_STAMP(13);
  }
  auto pix = [&](int e, int& oy, int& ox) {
    oy = ty0 + 4 * (e >> 3) + xi;
    ox = tx0 + 8 * lh + 2 * ((e & 7) >> 1) + (e & 1);       // column pair (r & 3) + 4 lh
  };
  EpiOps<8 * TM> ops[TN];
  unsigned okm[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) okm[tn] = halo_epi_load<8 * TM>(p, n, n0 + tn * 32 + li, pix, ops[tn]);
  float vmax = 0.f;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
    vmax = fmaxf(vmax, halo_epi_store<8 * TM>(p, n, n0 + tn * 32 + li, vals[tn], pix, bv[tn], ops[tn], okm[tn]));
  WX_STAMP(9);
  finish_amax(p, vmax, sV);
  WX_STAMP(10);
  if (pf_touch == 1.2345e-30f) p.y[0] = pf_touch;          // (never true for real data: keeps the touch loads alive)
}

// ---- 16 x 16-pixel tiles, 8 waves, double-buffered image ------------------------------------
// The 8 x 16 kernel above leaves a wave 12 MFMAs per tap (four accumulator sets cost it half of the
// direct kernel's pixel tile), which is too little to cover the latency of its weight fragments
// (an L2 hit) and of the next chunk's halo -- measured: without those two loads it runs 30 % faster.
// This variant keeps the per-wave tile and spends the CU differently: ONE 512-thread block per CU
// owns 16 x 16 output pixels x 128 channels (waves = 2 row halves x 4 channel quarters, 2 waves per
// SIMD), the transformed image of the (16 + 2) x 18 halo is double-buffered (all 148 KB of it), so
//   * the next chunk's halo is loaded, transformed and written in three rounds spread over the 12
//     taps of the current chunk (one 16-register round in flight), with ONE barrier per chunk;
//   * weight fragments are fetched two taps ahead (two register sets);
//   * the two row-half waves of a channel quarter read the same fragments (the second hits L1).
template <int FMT>
__global__ __launch_bounds__(512, 2) void igemm_wino16_kernel(ConvK p, int tiles_y, int tiles_x) {
  constexpr int NS = 2, CK = 32, TH = 16, TW = 16, BN = 128, TM = 2;
  constexpr int NTX = TW / 2, ROWS = TH + 2;
  constexpr int ENT = CK, ROW_E = 4 * NTX * ENT + 8, PLANE_E = ROWS * ROW_E, BUF_E = NS * PLANE_E;
  constexpr int NIT = ROWS * NTX * (CK / 4), RI = (NIT + 511) / 512;      // 1152 items, 3 rounds
  constexpr int T = 12, KSTEPS = CK / 16;
  static_assert(RI == 3, "three transform rounds per chunk");
  __shared__ __attribute__((aligned(16))) __bf16 sV[2 * BUF_E];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm2 = wave >> 2, wn = wave & 3;
  const int li = lane & 31, lh = lane >> 5;
  // 1-D grid over (pixel tile, n-tile).  Blocks are dealt round-robin over the 8 XCDs: the n-tiles of one
  // pixel tile are consecutive blocks OF ONE XCD (they read the same halo: the second one is an L2 hit),
  // and each XCD owns a contiguous band of pixel tiles (neighbouring halos overlap).
  const int ntn = (p.Co + BN - 1) / BN, npt = p.N * tiles_y * tiles_x;      // (the grid is padded to 8 equal XCD shares)
  const int lid = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  int bid, nty;
  {
    const int q8 = npt >> 3, r8 = npt & 7;                   // tiles per XCD: q8 (+1 for the first r8 XCDs)
    const int mine = q8 + (xcd < r8 ? 1 : 0), start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    // this XCD's blocks: lid = tile_local * ntn + n-tile; the grid is padded to 8 * ceil(...) blocks
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
  const int HL = p.H << p.ups, WL = p.W << p.ups;
  const float scA = FMT ? pow2_scale(*p.xamax) : 1.f;
  const float inv = FMT ? 1.f / (scA * pow2_scale(*p.wamax)) : 1.f;
  const float bv = (p.bias && n0 + wn * 32 + li < p.Co) ? p.bias[n0 + wn * 32 + li] : 0.f;    // (read here: no load after a store)
  const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;
  const int xsw = (int)p.xsw;

  // transform items (halo row r, column pair t, channel quad c4), c4 fastest; round i owns q = tid + 512 i
  int it_row[RI], it_lx[RI], it_dst[RI];
  unsigned it_ok[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int q = tid + 512 * i;
    const int c4 = q & 7, t = (q >> 3) & 7, r = q >> 6;
    const int ly = ty0 - 1 + r;
    it_ok[i] = 0;
    it_lx[i] = tx0 - 1 + 2 * t;
    it_row[i] = (int)((long)(ly >> p.ups) * p.xsh) + 4 * c4;
    it_dst[i] = r * ROW_E + t * ENT + 4 * c4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int lx = it_lx[i] + b;
      const bool ok = q < NIT && (unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL;
      it_ok[i] |= ok ? (1u << b) : 0u;
    }
    if (q < NIT) it_ok[i] |= 16u;                       // bit 4: the item exists
  }
  auto load_item = [&](f32x4 (&d)[4], int i, int c0) {
    const int c4 = (tid + 512 * i) & 7;
    const bool cin = c0 + 4 * c4 < p.C;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (((it_ok[i] >> b) & 1u) && cin)
        v = *reinterpret_cast<const f32x4*>(xin + (it_row[i] + ((it_lx[i] + b) >> p.ups) * xsw + c0));
      d[b] = v;
    }
  };
  auto store_item = [&](const f32x4 (&d)[4], int i, int buf) {
    if (it_ok[i] & 16u) {
      const f32x4 v[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
      __bf16* dst = sV + buf * BUF_E + it_dst[i];
#pragma unroll
      for (int xi = 0; xi < 4; ++xi) {
        bf16x4 parts[NS];
        split4f<NS, FMT>(v[xi], scA, parts);
#pragma unroll
        for (int s = 0; s < NS; ++s)
          *reinterpret_cast<bf16x4*>(dst + s * PLANE_E + xi * NTX * ENT) = parts[s];
      }
    }
  };
  // weight fragments: [tap 12][chunk][ntile][wn 4][kstep][plane][lane] x 16 B
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK, ntile_w = (p.Co + BN - 1) / BN;
  bf16x8 fb[2][KSTEPS][NS];
  auto load_B = [&](bf16x8 (&f)[NS], int ch, int tap, int ks) {
    const bf16x8* g = wfr + ((((long)tap * nchunk + ch) * ntile_w + nty) * 4 + wn) * (KSTEPS * NS * 64) +
                      ks * NS * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) f[s] = g[s * 64];
  };

  // ---- prologue: the whole first image (the accumulators are not live yet) ---------------------
  {
    f32x4 r0[4], r1[4], r2[4];
    load_item(r0, 0, 0);
    load_item(r1, 1, 0);
    load_item(r2, 2, 0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) { load_B(fb[0][ks], 0, 0, ks); load_B(fb[1][ks], 0, 1, ks); }
    store_item(r0, 0, 0);
    store_item(r1, 1, 0);
    store_item(r2, 2, 0);
  }
  f32x16 acc[TM][4];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][xi][r] = 0.f;
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = (8 * wm2 + 4 * a + (li >> 3)) * ROW_E + (li & 7) * ENT + 8 * lh;
  __syncthreads();

  f32x4 rh[4];
  for (int ch = 0; ch < nchunk; ++ch) {
    const bool more = ch + 1 < nchunk;
    const __bf16* vb = sV + (ch & 1) * BUF_E;
    const int nb = (ch & 1) ^ 1;
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
      const int ky = tap >> 2, xi = tap & 3;
      const int toff = ky * ROW_E + xi * NTX * ENT;
      // the next chunk's image: round i is loaded at tap 4 i and transformed + written at tap 4 i + 3
      if ((tap & 3) == 0 && more) load_item(rh, tap >> 2, (ch + 1) * CK);
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
        for (int a = 0; a < TM; ++a) mfma_splitf<NS, FMT>(fa[ks][a], fb[tap & 1][ks], acc[a][xi]);
        // this slot's next contents: the same k-step two taps on
        if (tap + 2 < T) load_B(fb[tap & 1][ks], ch, tap + 2, ks);
        else if (more) load_B(fb[tap & 1][ks], ch + 1, tap + 2 - T, ks);
      }
      if ((tap & 3) == 3 && more) store_item(rh, tap >> 2, nb);
    }
    if (more) __syncthreads();      // the other image is complete, nobody reads this one any more
  }

  // ---- epilogue: output transform (in registers) + the fused ops, one M tile at a time --------
  const int o = n0 + wn * 32 + li;
  const int oyb = ty0 + 8 * wm2, oxb = tx0 + 8 * lh;
  float vmax = 0.f;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    float vals[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[a][0][r], m1 = acc[a][1][r], m2 = acc[a][2][r], m3 = acc[a][3][r];
      vals[r * 2] = (m0 + m1 + m2) * inv;
      vals[r * 2 + 1] = (m1 - m2 - m3) * inv;
    }
    // element (r, j): C/D row (r & 3) + 8 (r >> 2) + 4 lh -> output row r >> 2 of M tile a, column pair (r & 3) + 4 lh
    vmax = fmaxf(vmax, halo_epilogue<32>(p, n, o, vals, [&](int e, int& oy, int& ox) {
      const int r = e >> 1, j = e & 1;
      oy = oyb + 4 * a + (r >> 2);
      ox = oxb + 2 * (r & 3) + j;
    }, bv));
  }
  finish_amax(p, vmax, sV);
}

}  // namespace

// Launch of the F(2,3) family; ``kernel`` (MtHaloKernel: MT_HK_WINOX / MT_HK_WINO16 / MT_HK_HALO_WINO) was chosen by
// mt_halo_select (dispatch.cpp).
void mt_halo_wino_launch(const void* kp, int fmt, int kernel, void* stream) {
  const ConvK& k = *static_cast<const ConvK*>(kp);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_y = (k.Ho + 7) / 8, tiles_x = (k.Wo + 15) / 16;
  const unsigned nt = (unsigned)k.N * tiles_y * tiles_x;
  if (kernel == MT_HK_WINOX) {
    if (k.Co > 32) {
      if (fmt) hipLaunchKernelGGL((igemm_winox_kernel<1, 2>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
      else hipLaunchKernelGGL((igemm_winox_kernel<0, 2>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
    } else {
      if (MT_WINOX_TH4 && fmt && nt < MT_WINOX_TH4_BELOW) {
        const int ty4 = (k.Ho + 3) / 4;
        hipLaunchKernelGGL((igemm_winox_kernel<1, 1, 4>), dim3((unsigned)k.N * ty4 * tiles_x, 1), dim3(256), 0, s, k, ty4, tiles_x);
      } else if (fmt) hipLaunchKernelGGL((igemm_winox_kernel<1, 1>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
      else hipLaunchKernelGGL((igemm_winox_kernel<0, 1>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
    }
    return;
  }
  if (kernel == MT_HK_WINO16) {          // 16 x 16 tiles, 128-wide
    const int ty16 = (k.Ho + 15) / 16;
    const unsigned nt16 = (unsigned)k.N * ty16 * tiles_x, ntn = (k.Co + 127) / 128;
    // (padded so that every XCD gets the same number of block slots: gridDim.x / ntn must be the tile count)
    const unsigned grid = 8 * ((nt16 + 7) / 8) * ntn;
    ConvK kk = k;
    kk.amax_deferred = kk.yamax && kk.ascr && grid >= MT_AMAX_DEFER_BLOCKS;   // (several rounds of blocks: fold the slots afterwards)
    if (fmt) hipLaunchKernelGGL((igemm_wino16_kernel<1>), dim3(grid), dim3(512), 0, s, kk, ty16, tiles_x);
    else hipLaunchKernelGGL((igemm_wino16_kernel<0>), dim3(grid), dim3(512), 0, s, kk, ty16, tiles_x);
    if (kk.amax_deferred) mt_amax_fold_launch(kk.ascr, kk.yamax, s);
    return;
  }
  if (k.Co > 64) {
    if (fmt) hipLaunchKernelGGL((igemm_halo_wino_kernel<1, 1>), dim3(nt, (k.Co + 127) / 128), dim3(256), 0, s, k, tiles_y, tiles_x);
    else hipLaunchKernelGGL((igemm_halo_wino_kernel<0, 1>), dim3(nt, (k.Co + 127) / 128), dim3(256), 0, s, k, tiles_y, tiles_x);
  } else {
    if (fmt) hipLaunchKernelGGL((igemm_halo_wino_kernel<1, 2>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
    else hipLaunchKernelGGL((igemm_halo_wino_kernel<0, 2>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
  }
}

#if MT_WINOX_TRACE
extern "C" __attribute__((visibility("default"))) int mt_lab_winox_trace(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(winox_trace), sizeof(unsigned long long) * n);
}
#endif
