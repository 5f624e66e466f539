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
// of LDS -- a tap is just a row offset into the halo image.  igemm_halo_kernel
// copies the per-tap weight tile (a ready LDS image, 16-bit planes) with LDS-DMA,
// double-buffered, one barrier per tap; igemm_halo_rb_kernel (128 / 64-wide
// channel tiles, >= 2 K chunks) fetches weight fragments straight into registers
// and synchronises once per chunk.  The next chunk's halo loads are in flight
// during the tap loop.
#include <cstdlib>
#include "conv_common.h"
#include "dispatch.h"

namespace {

// CKP = 16 with two-term splits: layers with <= 16 input channels (FlowEstimator's 10 -> 128 entry, RRDBNet's
// 9 -> 64): a 32-channel stage would be half zeros -- half the MFMAs, half the LDS image and weight tiles.
template <int KS, int NS, int FMT, int WGM, int WGN, int TM, int TN, int CKP = (NS == 3 ? 16 : 32)>
__global__ __launch_bounds__(256, 2) void igemm_halo_kernel(ConvK p, int tiles_y, int tiles_x) {
  constexpr int CK = CKP;                           // input channels per LDS stage
  constexpr int LD = CK + 8;                        // bf16 per LDS row (+16 B pad)
  constexpr int TH = 8, TW = 16;
  static_assert(WGM * TM * 32 == TH * TW, "pixel tile is 8 x 16");
  constexpr int BN = WGN * TN * 32;
  constexpr int HW_ = TW + KS - 1, HH = TH + KS - 1, HP = HH * HW_;   // halo patch
  constexpr int F4 = CK / 4, NH4 = HP * F4, RH = (NH4 + 255) / 256;
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
  // F16X3: per-tensor power-of-two scales of the activations and the weights
  const float sA = FMT ? pow2_scale(*p.xamax) : 1.f;
  const float inv = FMT ? 1.f / (sA * pow2_scale(*p.wamax)) : 1.f;
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
  // weights come as the exact LDS image of every (tap, chunk, n-tile) B tile:
  // [tap][chunk][ntile][plane][BN rows][LD] bf16 (mt_weight_pack_halo), so a
  // tile is one contiguous blob and every load instruction moves a full 1 KiB
  const __bf16* wt16 = reinterpret_cast<const __bf16*>(p.wt);
  const int nchunk_w = (p.C + CK - 1) / CK, ntile_w = (p.Co + BN - 1) / BN;
  constexpr int TILE_E = NS * BN * LD;              // bf16 elements per tile image
  constexpr int NCH = TILE_E / 8, RB = (NCH + 255) / 256;

  f32x4 rh[RH];
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
        split4f<NS, FMT>(rh[i], sA, parts);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&sH[s][hp * LD + 4 * c4]) = parts[s];
      }
    }
  };
  // LDS-DMA: the weight tile in HBM is the LDS image itself, so it is copied
  // global -> LDS directly (wave-uniform LDS base + lane * 16 B), no VGPR round
  // trip and no ds_write; __syncthreads() drains it before the tile is read
  auto load_B = [&](int c0, int tap, int buf) {
    const __bf16* tile = wt16 + ((long)(tap * nchunk_w + c0 / CK) * ntile_w + blockIdx.y) * TILE_E;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int c = tid + 256 * i;
      if (c < NCH)
        __builtin_amdgcn_global_load_lds((const unsigned*)(tile + c * 8),
                                         (__attribute__((address_space(3))) unsigned*)(&sB[buf][0][(c - lane) * 8]),
                                         16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // MFMA row i of 32-pixel group g = tile pixel (2g + (i >> 4), px(i)) with
  // px(i) = i for the first tile row and (i - HW_) & 15 for the second: the
  // 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, ...) then touch 16
  // consecutive halo pixels (mod 16) -> 16 distinct 16-B bank slots.  The
  // epilogue uses the same map.
  const int apx = li < 16 ? li : ((li - HW_) & 15);
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = ((2 * (wm * TM + a) + (li >> 4)) * HW_ + apx) * LD + 8 * lh;
  const int b_base = (wn * TN * 32 + li) * LD + 8 * lh;

  const int nchunk = (p.C + CK - 1) / CK;
  load_halo(0);
  store_halo();
  load_B(0, 0, 0);
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
      if (!last) load_B(c0, tap + 1, buf ^ 1);
      else if (more) load_B(c0 + CK, 0, buf ^ 1);
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
          for (int b = 0; b < TN; ++b) mfma_splitf<NS, FMT>(fa[a], fb[b], acc[a][b]);
      }
      if (last && more) {
        __syncthreads();                      // every wave is done with this chunk's halo
        store_halo();
      }
      __syncthreads();
      if (++kw == KS) { kw = 0; ++kh; }
    }
  }

  // ---- epilogue (tile-local pixel map; conv_common.h: halo_epilogue), one 32 x 32 tile at a time ----
  float vmax = 0.f;
#pragma unroll
  for (int b = 0; b < TN; ++b) {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      float vals[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) vals[r] = acc[a][b][r] * inv;
      auto pix = [&](int r, int& oy, int& ox) {
        const int row = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;     // MFMA row -> tile pixel
        const int i32 = row & 31;
        oy = ty0 + (row >> 4);
        ox = tx0 + (i32 < 16 ? i32 : ((i32 - HW_) & 15));
      };
      // one / two accumulator tiles per wave: latency-bound variants that live on occupancy
      if constexpr (TM * TN <= 2) vmax = fmaxf(vmax, halo_epilogue_lean<16>(p, n, n0 + wn * TN * 32 + b * 32 + li, vals, pix));
      else vmax = fmaxf(vmax, halo_epilogue<16>(p, n, n0 + wn * TN * 32 + b * 32 + li, vals, pix));
    }
  }
  finish_amax(p, vmax, sH);
}


// ---- "register-B" variant for 128-wide output-channel tiles ----------------
// Same pixel tile and halo image as igemm_halo_kernel, but the four waves of a
// block split the 128 output channels (wave = 128 pixels x 32 channels) and
// fetch their weight fragments STRAIGHT from global memory in MFMA operand
// order (mt_weight_pack_halo writes them that way: one load instruction = one
// contiguous 1 KiB, no two waves of a block read the same bytes).  The weights
// therefore never pass through LDS: no LDS writes for B (LDS write bandwidth
// is a third of its read bandwidth), no per-tap barrier.  The halo image is
// double-buffered, so a block synchronises ONCE per K chunk (every k*k taps).
// Fragment slot B[ks] is reloaded for the next tap right after the MFMAs that
// read it have been issued.
template <int KS, int FMT, int WM>
__global__ __launch_bounds__(256, 2) void igemm_halo_rb_kernel(ConvK p, int tiles_y, int tiles_x) {
  // WM waves along the 128 pixels x WN = 4 / WM waves along the channels: wave = (128 / WM) pixels x 32 channels
  constexpr int NS = 2, CK = 32, LD = CK + 8, TH = 8, TW = 16, WN = 4 / WM, BN = WN * 32, TM = 4 / WM;
  constexpr int HW_ = TW + KS - 1, HH = TH + KS - 1, HP = HH * HW_;
  constexpr int F4 = CK / 4, NH4 = HP * F4, RH = (NH4 + 255) / 256;
  constexpr int T = KS * KS, KSTEPS = CK / 16;
  constexpr int HALO_E = NS * HP * LD;                 // bf16 elements of one halo image
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
  const int HL = p.H << p.ups, WL = p.W << p.ups;
  const float scA = FMT ? pow2_scale(*p.xamax) : 1.f;
  const float inv = FMT ? 1.f / (scA * pow2_scale(*p.wamax)) : 1.f;
  const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;

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
  f32x4 rh[RH];
  auto load_halo = [&](int c0) {
#pragma unroll
    for (int i = 0; i < RH; ++i) {
      const int c4 = (tid + 256 * i) % F4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (h_ok[i] && c0 + 4 * c4 < p.C) v = *reinterpret_cast<const f32x4*>(xin + h_off[i] + c0);
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
  // weight fragments: [tap][chunk][ntile][wn][kstep][plane][lane] x 16 B
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK, ntile_w = (p.Co + BN - 1) / BN;
  bf16x8 fb[KSTEPS][NS];
  auto load_B = [&](int ch, int tap, int ks) {
    const bf16x8* f = wfr + ((((long)tap * nchunk + ch) * ntile_w + blockIdx.y) * WN + wn) * (KSTEPS * NS * 64) +
                      ks * NS * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) fb[ks][s] = f[s * 64];
  };

  f32x16 acc[TM][1];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;

  // MFMA row -> tile pixel map as in igemm_halo_kernel (conflict-free ds_read_b128)
  const int apx = li < 16 ? li : ((li - HW_) & 15);
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = ((2 * (wm * TM + a) + (li >> 4)) * HW_ + apx) * LD + 8 * lh;

  load_halo(0);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) load_B(0, 0, ks);
  store_halo(0);
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const int cur = ch & 1;
    const bool more = ch + 1 < nchunk;
    const __bf16* hb = sH + cur * HALO_E;
    int kh = 0, kw = 0;
    for (int tap = 0; tap < T; ++tap) {
      const bool last = tap == T - 1;
      // the next chunk's halo: converted into the other image under the last tap's MFMAs
      if (last && more) store_halo(cur ^ 1);
      const int toff = (kh * HW_ + kw) * LD;
      // all A fragments of the tap (both k-steps) are requested before the first MFMA
      bf16x8 fa[KSTEPS][TM][NS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int s = 0; s < NS; ++s)
            fa[ks][a][s] = *reinterpret_cast<const bf16x8*>(&hb[s * HP * LD + a_base[a] + toff + ks * 16]);
      __builtin_amdgcn_sched_barrier(0);     // (the scheduler would sink every read next to its MFMA)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
        for (int a = 0; a < TM; ++a) mfma_splitf<NS, FMT>(fa[ks][a], fb[ks], acc[a][0]);
        // this slot's next contents: the same k-step of the next tap
        if (!last) load_B(ch, tap + 1, ks);
        else if (more) load_B(ch + 1, 0, ks);
      }
      // issued late in the chunk and after the weight loads of the next tap, so only the
      // last tap's weight wait is ordered behind these loads
      if (tap == T - 3 && more) load_halo((ch + 1) * CK);
      if (++kw == KS) { kw = 0; ++kh; }
    }
    if (more) __syncthreads();      // the other image is complete, nobody reads this one any more
  }

  // ---- epilogue (as igemm_halo_kernel: wave = pixels 0..127 x channels wn*32..+31) ---
  float vmax = 0.f;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    float vals[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vals[r] = acc[a][0][r] * inv;
    vmax = fmaxf(vmax, halo_epilogue<16>(p, n, n0 + wn * 32 + li, vals, [&](int r, int& oy, int& ox) {
      const int row = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int i32 = row & 31;
      oy = ty0 + (row >> 4);
      ox = tx0 + (i32 < 16 ? i32 : ((i32 - HW_) & 15));
    }));
  }
  finish_amax(p, vmax, sH);
}


}  // namespace
void mt_halo_wino_launch(const void* convk, int fmt, int kernel, void* stream);      // conv_halo_wino.hip (kernel: MtHaloKernel)
void mt_halo_wino43_launch(const void* convk, void* stream);                          // conv_halo_wino43.hip
void mt_pack_halo_wino43_launch(const float* src, int A, int B, long sa, long sb, long st, void* dst, int nchunk, int ntile,
                                int flip, const float* wamax, int nchunk_total, int ch_off, void* stream);
void mt_halo_wino25_launch(const void* convk, int fmt, void* stream);                 // conv_halo_wino25.hip
void mt_pack_halo_wino25_launch(const float* src, int A, int B, long sa, long sb, long st, void* dst, int nchunk, int ntile,
                                int flip, const float* wamax, int nchunk_total, int ch_off, void* stream);
namespace {
// The kernel is chosen by mt_halo_select (dispatch.cpp); this function only launches it.
template <int KS, int NS, int FMT>
void launch_halo(const ConvK& k, hipStream_t s) {
  const int tiles_y = (k.Ho + 7) / 8, tiles_x = (k.Wo + 15) / 16;
  const unsigned nt = (unsigned)k.N * tiles_y * tiles_x;
  // (the xi-split kernel addresses an input image through a buffer descriptor with 32-bit offsets: images under 2 GB)
  const bool x32 = ((long)(k.H - 1) * k.xsh + (long)(k.W - 1) * k.xsw + k.C + 4) * 4 < (1L << 31);
  const MtHaloSel sel = mt_halo_select(KS, NS, FMT, k.C, k.Co, k.Ho, x32);
  if constexpr (NS == 2) {
    switch (sel.kernel) {
      case MT_HK_WINO43: mt_halo_wino43_launch(&k, s); return;
      case MT_HK_WINOX: case MT_HK_WINO16: case MT_HK_HALO_WINO: mt_halo_wino_launch(&k, FMT, sel.kernel, s); return;
      case MT_HK_WINO25: mt_halo_wino25_launch(&k, FMT, s); return;
      case MT_HK_RB:
        if (k.Co > 64)
          hipLaunchKernelGGL((igemm_halo_rb_kernel<KS, FMT, 1>), dim3(nt, (k.Co + 127) / 128), dim3(256), 0, s, k, tiles_y, tiles_x);
        else
          hipLaunchKernelGGL((igemm_halo_rb_kernel<KS, FMT, 2>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
        return;
      default: break;
    }
    if (sel.ck == 16) {          // <= 16 input channels: 16-channel stages (the packs follow: halo_ck)
      if (k.Co > 64)
        hipLaunchKernelGGL((igemm_halo_kernel<KS, NS, FMT, 2, 2, 2, 2, 16>), dim3(nt, (k.Co + 127) / 128), dim3(256), 0, s, k, tiles_y, tiles_x);
      else if (k.Co > 32)
        hipLaunchKernelGGL((igemm_halo_kernel<KS, NS, FMT, 2, 2, 2, 1, 16>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
      else
        hipLaunchKernelGGL((igemm_halo_kernel<KS, NS, FMT, 4, 1, 1, 1, 16>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
      return;
    }
  }
  if (k.Co > 64) {
    hipLaunchKernelGGL((igemm_halo_kernel<KS, NS, FMT, 2, 2, 2, 2>), dim3(nt, (k.Co + 127) / 128), dim3(256), 0, s, k,
                       tiles_y, tiles_x);
  } else if (k.Co > 32) {
    hipLaunchKernelGGL((igemm_halo_kernel<KS, NS, FMT, 2, 2, 2, 1>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
  } else {
    hipLaunchKernelGGL((igemm_halo_kernel<KS, NS, FMT, 4, 1, 1, 1>), dim3(nt, 1), dim3(256), 0, s, k, tiles_y, tiles_x);
  }
}

}  // namespace

// Called by mt_conv_run: returns 1 if the launch was taken by the halo kernel.
int mt_halo_try_launch(const mt_conv* d, void* stream) {
  const int prec = d->precision;
  if (prec != MT_PREC_BF16X3 && prec != MT_PREC_BF16X6 && prec != MT_PREC_F16X3) return 0;
  if (d->transposed || d->stride != 1 || d->KH != d->KW || (d->KH != 3 && d->KH != 5) || d->pad != d->KH / 2) return 0;
  if ((d->x.C & 7) != 0 || d->wt_layout != MT_WT_HALO) return 0;
  if (prec == MT_PREC_F16X3 && (!d->x_amax || !d->w_amax)) return 0;
  ConvK k;
  if (fill_convk(d, &k) != MT_OK) return 0;
  k.Kp = ((k.K + 31) / 32) * 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k.yamax) k.ascr = mt_amax_scratch(s);
  if (d->KH == 3) {
    if (prec == MT_PREC_BF16X6) launch_halo<3, 3, 0>(k, s);
    else if (prec == MT_PREC_BF16X3) launch_halo<3, 2, 0>(k, s);
    else launch_halo<3, 2, 1>(k, s);
  } else {
    if (prec == MT_PREC_BF16X6) launch_halo<5, 3, 0>(k, s);
    else if (prec == MT_PREC_BF16X3) launch_halo<5, 2, 0>(k, s);
    else launch_halo<5, 2, 1>(k, s);
  }
  return 1;
}

namespace {
// dst[tap][chunk][ntile][plane][BN][LD]: the LDS image of every weight tile of
// igemm_halo_kernel (pad columns and out-of-range rows/channels are zero)
__device__ __forceinline__ void pack_halo_body(const float* src, int A, int B, int T, long sa, long sb, long st, __bf16* dst,
                                               int BN, int CK, int NS, int nchunk, int ntile, int flip,
                                               const float* wamax, int nchunk_total, int ch_off) {
  const float sc = wamax ? pow2_scale(*wamax) : 1.f;    // F16X3: fp16 planes of W * 2^k
  const int LD = CK + 8;
  const long tile_e = (long)NS * BN * LD;
  const long total = (long)T * nchunk * ntile * BN * LD;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int col = (int)(e % LD);
    long r = e / LD;
    const int row = (int)(r % BN); r /= BN;
    const int nt = (int)(r % ntile); r /= ntile;
    const int ch = (int)(r % nchunk);
    const int t = (int)(r / nchunk);
    const int a = nt * BN + row, b = ch * CK + col;
    float v = 0.f;
    if (col < CK && a < A && b < B) v = src[a * sa + b * sb + (flip ? T - 1 - t : t) * st];
    __bf16* tile = dst + ((long)(t * nchunk_total + ch_off + ch) * ntile + nt) * tile_e;
    if (wamax) {
      v *= sc;
      const _Float16 h = (_Float16)v;
      tile[((long)0 * BN + row) * LD + col] = __builtin_bit_cast(__bf16, h);
      tile[((long)1 * BN + row) * LD + col] = __builtin_bit_cast(__bf16, (_Float16)(v - (float)h));
    } else {
      for (int s = 0; s < NS; ++s) {
        const __bf16 h = (__bf16)v;
        tile[((long)s * BN + row) * LD + col] = h;
        v -= (float)h;
      }
    }
  }
}

// Fragment-order weights of igemm_halo_rb_kernel: dst[tap][chunk][ntile][wn WN][kstep 2][plane 2][lane 64][8],
// WN = BN / 32 waves along the channels: lane (li = lane & 31, lh = lane >> 5) holds
// W[a = ntile*BN + wn*32 + li][b = chunk*32 + kstep*16 + 8*lh + j]
__global__ void weight_pack_halo_kernel(const float* src, int A, int B, int T, long sa, long sb, long st, __bf16* dst,
                                        int BN, int CK, int NS, int nchunk, int ntile, int flip, const float* wamax,
                                        int nchunk_total, int ch_off) {
  pack_halo_body(src, A, B, T, sa, sb, st, dst, BN, CK, NS, nchunk, ntile, flip, wamax, nchunk_total, ch_off);
}

__device__ __forceinline__ void pack_halo_rb_body(const float* src, int A, int B, int T, long sa, long sb, long st,
                                                  __bf16* dst, int nchunk, int ntile, int flip, const float* wamax,
                                                  int nchunk_total, int ch_off, int WN) {
  const float sc = wamax ? pow2_scale(*wamax) : 1.f;
  const long total = (long)T * nchunk * ntile * WN * 2 * 64 * 8;      // elements per plane
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), ks = (int)((e >> 9) & 1);
    long r = e >> 10;
    const int wv = (int)(r % WN); r /= WN;
    const int nt = (int)(r % ntile); r /= ntile;
    const int ch = (int)(r % nchunk);
    const int t = (int)(r / nchunk);
    const int a = (nt * WN + wv) * 32 + (lane & 31), b = ch * 32 + ks * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (a < A && b < B) v = src[a * sa + b * sb + (flip ? T - 1 - t : t) * st];
    __bf16* o = dst + ((((((long)t * nchunk_total + ch_off + ch) * ntile + nt) * WN + wv) * 2 + ks) * 2) * 512 + lane * 8 + j;
    if (wamax) {
      v *= sc;
      const _Float16 h = (_Float16)v;
      o[0] = __builtin_bit_cast(__bf16, h);
      o[512] = __builtin_bit_cast(__bf16, (_Float16)(v - (float)h));
    } else {
      const __bf16 h = (__bf16)v;
      o[0] = h;
      o[512] = (__bf16)(v - (float)h);
    }
  }
}

__global__ void weight_pack_halo_wino_kernel(const float* src, int A, int B, long sa, long sb, long st, __bf16* dst, int nchunk,
                                             int ntile, int flip, const float* wamax, int nchunk_total, int ch_off, int WN) {
  pack_halo_wino_body(src, A, B, sa, sb, st, dst, nchunk, ntile, flip, wamax, nchunk_total, ch_off, WN);
}

__global__ void weight_pack_halo_rb_kernel(const float* src, int A, int B, int T, long sa, long sb, long st, __bf16* dst,
                                           int nchunk, int ntile, int flip, const float* wamax, int nchunk_total,
                                           int ch_off, int WN) {
  pack_halo_rb_body(src, A, B, T, sa, sb, st, dst, nchunk, ntile, flip, wamax, nchunk_total, ch_off, WN);
}

// ---- batched re-pack (mt_weight_pack_jobs / mt_absmax_segments) ----------------------------
__global__ void weight_pack_jobs_kernel(const mt_pack_job* jobs) {
  const mt_pack_job j = jobs[blockIdx.y];
  const int NS = j.precision == MT_PREC_BF16X6 ? 3 : 2;
  __bf16* dst = reinterpret_cast<__bf16*>(j.dst);
  if (j.kind == MT_PACK_PLANES) {          // stride-2 layer: fragments in the planes kernel's consumption order
    const int K = j.T == 9 ? 3 : 5, BNp = j.AP > 64 ? 128 : 64;
    const UnitTaps t = unit_taps(j.flip, K, K / 2);
    pack_planes_body(j.src, j.A, j.B, j.sa, j.sb, j.st, dst, j.BP / 32, (j.AP + BNp - 1) / BNp, BNp / 32, j.w_amax, t);
    return;
  }
  if (j.kind == MT_PACK_SPLIT) {
    pack_split_body(j.src, j.A, j.B, j.T, j.sa, j.sb, j.st, dst, j.AP, j.BP, ((j.T * j.BP + 31) / 32) * 32, NS, j.flip,
                    j.w_amax);
    return;
  }
  const int CK = halo_ck(j.BP_total, NS);
  const int BN = j.AP > 64 ? 128 : (j.AP > 32 ? 64 : 32);
  const int nchunk = (j.BP + CK - 1) / CK, ntile = (j.AP + BN - 1) / BN;
  const int nchunk_total = (j.BP_total + CK - 1) / CK, ch_off = j.b_off / CK;
  if (j.frag_order == 6)
    pack_halo_wino43_body(j.src, j.A, j.B, j.sa, j.sb, j.st, dst, nchunk, ntile, j.flip, j.w_amax, nchunk_total, ch_off);
  else if (j.frag_order == 5)
    pack_halo_wino25_body(j.src, j.A, j.B, j.sa, j.sb, j.st, dst, nchunk, ntile, j.flip, j.w_amax, nchunk_total, ch_off);
  else if (j.frag_order == 2)
    pack_halo_wino_body(j.src, j.A, j.B, j.sa, j.sb, j.st, dst, nchunk, ntile, j.flip, j.w_amax, nchunk_total, ch_off, BN / 32);
  else if (j.frag_order)
    pack_halo_rb_body(j.src, j.A, j.B, j.T, j.sa, j.sb, j.st, dst, nchunk, ntile, j.flip, j.w_amax, nchunk_total, ch_off,
                      BN / 32);
  else
    pack_halo_body(j.src, j.A, j.B, j.T, j.sa, j.sb, j.st, dst, BN, CK, NS, nchunk, ntile, j.flip, j.w_amax,
                   nchunk_total, ch_off);
}

__global__ void absmax_segments_kernel(const mt_amax_seg* segs) {
  const mt_amax_seg g = segs[blockIdx.y];
  float m = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < g.n; i += (long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(g.p[i]));
  raise_amax(g.out, m);
}
}  // namespace

extern "C" int64_t mt_weight_pack_halo_elems(int32_t AP, int32_t BP, int32_t T, int32_t precision) {
  const int NS = precision == MT_PREC_BF16X6 ? 3 : 2, CK = halo_ck(BP, NS), LD = CK + 8;
  const int BN = AP > 64 ? 128 : (AP > 32 ? 64 : 32);
  if (NS == 2 && halo_wino43_wanted(BP, AP, T, precision == MT_PREC_F16X3)) T = 18;   // F(4,3): 3 kernel rows x 6 xi
  else if (NS == 2 && halo_wino_wanted(BP, AP, T)) T = 12;      // the (ky, xi) "taps" of the Winograd image (fragment order)
  else if (NS == 2 && halo_wino25_wanted(BP, AP, T, precision == MT_PREC_F16X3)) T = 30;   // F(2,5): 5 kernel rows x 6 xi
  return (int64_t)T * ((BP + CK - 1) / CK) * ((AP + BN - 1) / BN) * NS * BN * LD;
}

static int pack_halo_impl(const float* src, int32_t A, int32_t B, int32_t T, int64_t sa, int64_t sb, int64_t st,
                          void* dst, int32_t AP, int32_t BP, int32_t precision, int32_t flip, const float* w_amax,
                          int32_t BP_total, int32_t b_off, void* stream) {
  MT_REQUIRE(src && dst && A <= AP && B <= BP && (BP & 7) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0,
             "mt_weight_pack_halo: bad arguments");
  MT_REQUIRE(precision == MT_PREC_BF16X3 || precision == MT_PREC_BF16X6 || precision == MT_PREC_F16X3,
             "mt_weight_pack_halo: precision");
  MT_REQUIRE((precision == MT_PREC_F16X3) == (w_amax != nullptr), "mt_weight_pack_halo: w_amax goes with F16X3");
  const int NS = precision == MT_PREC_BF16X6 ? 3 : 2, CK = halo_ck(BP_total, NS);
  MT_REQUIRE(b_off >= 0 && b_off % CK == 0 && b_off + BP <= BP_total && (b_off + BP == BP_total || BP % CK == 0),
             "mt_weight_pack_halo_part: a part must cover whole %d-channel K chunks", CK);
  const int BN = AP > 64 ? 128 : (AP > 32 ? 64 : 32);        // the n-tile igemm_halo uses for Co = AP
  const int nchunk = (BP + CK - 1) / CK, ntile = (AP + BN - 1) / BN;
  const int nchunk_total = (BP_total + CK - 1) / CK, ch_off = b_off / CK;
  // (the same precedence as mt_weight_pack_halo_frag_order: F(4,3) before F(2,3))
  if (NS == 2 && halo_wino43_wanted(BP_total, AP, T, precision == MT_PREC_F16X3)) {   // F(4,3)-transformed fragments for igemm_wino43_kernel
    mt_pack_halo_wino43_launch(src, A, B, (long)sa, (long)sb, (long)st, dst, nchunk, ntile, flip, w_amax, nchunk_total, ch_off, stream);
    return mt_check_launch("mt_weight_pack_halo(wino43)");
  }
  if (NS == 2 && halo_wino_wanted(BP_total, AP, T)) {     // Winograd-transformed fragments for igemm_halo_wino_kernel
    const long tot = (long)12 * nchunk * ntile * (BN / 32) * 1024;
    hipLaunchKernelGGL(weight_pack_halo_wino_kernel, dim3(mt_grid_1d(tot, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, A, B, (long)sa, (long)sb, (long)st,
                       reinterpret_cast<__bf16*>(dst), nchunk, ntile, flip, w_amax, nchunk_total, ch_off, BN / 32);
    return mt_check_launch("mt_weight_pack_halo(wino)");
  }
  if (NS == 2 && halo_wino25_wanted(BP_total, AP, T, precision == MT_PREC_F16X3)) {   // F(2,5)-transformed fragments for igemm_wino25_kernel
    mt_pack_halo_wino25_launch(src, A, B, (long)sa, (long)sb, (long)st, dst, nchunk, ntile, flip, w_amax, nchunk_total, ch_off, stream);
    return mt_check_launch("mt_weight_pack_halo(wino25)");
  }
  if (NS == 2 && halo_rb_wanted(BP_total, AP)) {     // fragment order for igemm_halo_rb_kernel (fits the same buffer)
    const long tot = (long)T * nchunk * ntile * (BN / 32) * 1024;
    hipLaunchKernelGGL(weight_pack_halo_rb_kernel, dim3(mt_grid_1d(tot, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, A, B, T, (long)sa, (long)sb, (long)st,
                       reinterpret_cast<__bf16*>(dst), nchunk, ntile, flip, w_amax, nchunk_total, ch_off, BN / 32);
    return mt_check_launch("mt_weight_pack_halo(rb)");
  }
  const long total = (long)T * nchunk * ntile * BN * (CK + 8);
  hipLaunchKernelGGL(weight_pack_halo_kernel, dim3(mt_grid_1d(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, A, B, T, (long)sa, (long)sb, (long)st,
                     reinterpret_cast<__bf16*>(dst), BN, CK, NS, nchunk, ntile, flip, w_amax, nchunk_total, ch_off);
  return mt_check_launch("mt_weight_pack_halo");
}

extern "C" int mt_weight_pack_halo(const float* src, int32_t A, int32_t B, int32_t T, int64_t sa, int64_t sb,
                                   int64_t st, void* dst, int32_t AP, int32_t BP, int32_t precision, int32_t flip,
                                   const float* w_amax, void* stream) {
  return pack_halo_impl(src, A, B, T, sa, sb, st, dst, AP, BP, precision, flip, w_amax, BP, 0, stream);
}

extern "C" int mt_weight_pack_halo_part(const float* src, int32_t A, int32_t B, int32_t T, int64_t sa, int64_t sb,
                                        int64_t st, void* dst, int32_t AP, int32_t BP, int32_t precision,
                                        int32_t flip, const float* w_amax, int32_t BP_total, int32_t b_off,
                                        void* stream) {
  return pack_halo_impl(src, A, B, T, sa, sb, st, dst, AP, BP, precision, flip, w_amax, BP_total, b_off, stream);
}

extern "C" int32_t mt_weight_pack_halo_frag_order(int32_t AP, int32_t BP_total, int32_t T, int32_t precision) {
  if (precision == MT_PREC_BF16X6) return 0;
  if (halo_wino43_wanted(BP_total, AP, T, precision == MT_PREC_F16X3)) return 6;        // 6: F(4,3) fragments (igemm_wino43_kernel)
  if (halo_wino25_wanted(BP_total, AP, T, precision == MT_PREC_F16X3)) return 5;        // 5: F(2,5) fragments (igemm_wino25_kernel)
  if (halo_wino_wanted(BP_total, AP, T)) return 2;          // 2: Winograd F(2,3) fragments
  if (!halo_rb_wanted(BP_total, AP)) return 0;
  return 1;
}

extern "C" int mt_absmax_segments(const mt_amax_seg* segs_dev, int32_t nseg, void* stream) {
  MT_REQUIRE(segs_dev && nseg > 0 && nseg < 65536, "mt_absmax_segments: bad arguments");
  hipLaunchKernelGGL(absmax_segments_kernel, dim3(16, nseg), dim3(256), 0, static_cast<hipStream_t>(stream), segs_dev);
  return mt_check_launch("mt_absmax_segments");
}

extern "C" int mt_weight_pack_jobs(const mt_pack_job* jobs_dev, const mt_pack_job* jobs_host, int32_t njobs,
                                   void* stream) {
  MT_REQUIRE(jobs_dev && jobs_host && njobs > 0 && njobs < 65536, "mt_weight_pack_jobs: bad arguments");
  for (int i = 0; i < njobs; ++i) {
    const mt_pack_job& j = jobs_host[i];
    MT_REQUIRE(j.src && j.dst && j.A <= j.AP && j.B <= j.BP && (j.BP & 3) == 0 &&
                   (reinterpret_cast<uintptr_t>(j.dst) & 15) == 0,
               "mt_weight_pack_jobs: job %d: bad arguments", i);
    MT_REQUIRE(j.precision == MT_PREC_BF16X3 || j.precision == MT_PREC_BF16X6 || j.precision == MT_PREC_F16X3,
               "mt_weight_pack_jobs: job %d: precision", i);
    MT_REQUIRE((j.precision == MT_PREC_F16X3) == (j.w_amax != nullptr), "mt_weight_pack_jobs: job %d: w_amax goes with F16X3", i);
    if (j.kind == MT_PACK_HALO) {
      const int CK = halo_ck(j.BP_total, j.precision == MT_PREC_BF16X6 ? 3 : 2);
      MT_REQUIRE(j.frag_order == mt_weight_pack_halo_frag_order(j.AP, j.BP_total, j.T, j.precision),
                 "mt_weight_pack_jobs: job %d: frag_order must be mt_weight_pack_halo_frag_order()", i);
      MT_REQUIRE((j.BP & 7) == 0 && j.b_off >= 0 && j.b_off % CK == 0 && j.b_off + j.BP <= j.BP_total &&
                     (j.b_off + j.BP == j.BP_total || j.BP % CK == 0),
                 "mt_weight_pack_jobs: job %d: K part", i);
    } else if (j.kind == MT_PACK_PLANES) {
      MT_REQUIRE((j.T == 9 || j.T == 25) && mt_weight_pack_planes_wanted(j.AP, j.BP, j.T, j.precision) && (j.flip == 0 || j.flip == 1),
                 "mt_weight_pack_jobs: job %d: not a shape of the planes kernel", i);
    } else {
      MT_REQUIRE(j.kind == MT_PACK_SPLIT, "mt_weight_pack_jobs: job %d: kind", i);
    }
  }
  hipLaunchKernelGGL(weight_pack_jobs_kernel, dim3(32, njobs), dim3(256), 0, static_cast<hipStream_t>(stream), jobs_dev);
  return mt_check_launch("mt_weight_pack_jobs");
}
