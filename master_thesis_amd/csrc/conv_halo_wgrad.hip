// Weight gradient of the stride-1 / stride-2 3x3 and 5x5 layers on LDS halo tiles (its own
// translation unit: the template instantiations compile beside those of conv_halo.hip).
#include <cstdlib>
#include <type_traits>
// LDS row padding (16-bit elements) of the producer / consumer kernel: rows of 96 put the four pixel rows of a
// transposed fragment read on banks 0 / 48 / 32 / 16 (72: two of them share 8 banks); +1 % (3x3), +4 % (5x5)
#ifndef MT_WGPC_PAD
#define MT_WGPC_PAD 32
#endif
// split form of the staging waves (conv_common.h: split4f SHADOW): 1 = instructions that do not share the matrix pipe
#ifndef MT_WGPC_SHADOW
#define MT_WGPC_SHADOW 1
#endif
#include <cstring>
#include "conv_common.h"
#include "dispatch.h"

// ------------------------------------------------------------ wgrad (halo) --
// Weight gradient of the same stride-1 layers:
//   dW[a][kh][kw][b] = sum_pixels D[pix][a] * G[pix + (kh,kw) - pad][b]
// K = pixels.  A block owns a 64 x 64 (a, b) channel tile and KHB kernel rows
// (all taps of a 3x3, one row of a 5x5) and walks over 4 x 16 pixel tiles: per
// tile the D pixels and the G halo patch are loaded ONCE, split once, and
// stored pixel-major exactly as they sit in HBM; the MFMA operands (which need
// the pixel axis along k) come out of LDS through ds_read_b64_tr_b16, so a tap
// is again only a row offset.  One k-step = one 16-pixel tile row.
namespace {

typedef __bf16 __attribute__((address_space(3))) lds_bf16;
typedef bf16x4 __attribute__((address_space(3))) lds_bf16x4;

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* base, int off, int ld) {
  // two transposed 4(k) x 16(col) block reads -> this lane's 8 consecutive k
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(base + off + 4 * ld));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ST = 2: the gathered tensor is sampled with stride 2 (a stride-2 Conv2d, or a stride-2
// ConvTranspose2d with the roles of input and output gradient swapped): the patch is
// (TR-1)*2+KHB x (TW-1)*2+KS pixels and the k rows of a G fragment are every second pixel.
// DT = 1: D enters with its leading 16-bit term only (mt_wgrad.d_terms, see wgrad_pc_kernel).
template <int KS, int KHB, int NS, int FMT, int OCC, int ST = 1, int TRP = 4, int DT = NS>
__global__ __launch_bounds__(256, OCC) void wgrad_halo_kernel(WgK p, int tiles_y, int tiles_x, int tiles_per_split,
                                                            int ntiles, int with_bias, int kh_base) {
  constexpr int TR = TRP, TW = 16, PX = TR * TW;         // pixel tile: TR (4 or 8) rows x 16
  constexpr int CH = 64, LD = CH + 8;                    // channel tile and LDS row (bf16)
  constexpr int GH = (TR - 1) * ST + KHB, GW = (TW - 1) * ST + KS, GP = GH * GW;
  constexpr int ND4 = PX * (CH / 4), RD = ND4 / 256;     // float4 loads per thread (D)
  constexpr int NG4 = GP * (CH / 4), RG = (NG4 + 255) / 256;
  constexpr int NT = KHB * KS;
  __shared__ __attribute__((aligned(16))) __bf16 sD[NS][PX * LD];
  __shared__ __attribute__((aligned(16))) __bf16 sG[NS][GP * LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int bid, split, bz_;
  if (!wg_block(p, bid, split, bz_)) return;     // (XCD-aware block order: conv_common.h)
  const int bt = bid % p.btiles; bid /= p.btiles;
  const int at = bid % p.atiles; bid /= p.atiles;
  const int kh0 = kh_base + bid * KHB;     // first kernel row of this block
  if (at * p.btiles + bt < 32 && ((p.skip >> (at * p.btiles + bt)) & 1u)) return;   // tile nobody reads
  const int a0 = at * CH, b0 = bt * CH;
  const int t_beg = split * tiles_per_split, t_end = min(ntiles, t_beg + tiles_per_split);
  const int HL = p.Hg << p.ups, WL = p.Wg << p.ups;
  const float scD = FMT ? pow2_scale(*p.damax) : 1.f, scG = FMT ? pow2_scale(*p.gamax) : 1.f;
  const float inv = 1.f / (scD * scG), inv_d = 1.f / scD;

  // staging coordinates relative to the tile origin; the tile origin itself is wave-uniform, so a
  // load address is (SGPR base) + (32-bit element offset): no per-load 64-bit multiplies
  // D: float4 idx = tid + 256 i  ->  pixel row i, column tid >> 4, channels 4 (tid & 15)
  const int rxD = tid >> 4, c4 = (tid & 15) * 4;
  const bool chD = a0 + c4 < p.Cd, chG = b0 + c4 < p.Cg;
  const unsigned dsh = (unsigned)p.dsh, dsw = (unsigned)p.dsw, gsh = (unsigned)p.gsh, gsw = (unsigned)p.gsw;
  const unsigned offD0 = rxD * dsw + a0 + c4;
  int gyx[RG];                // patch pixel coordinates, gy << 8 | gx
#pragma unroll
  for (int i = 0; i < RG; ++i) {
    const int gpx = (tid >> 4) + 16 * i, gy = gpx / GW, gx = gpx - gy * GW;
    gyx[i] = gy << 8 | gx;
  }

  f32x4 rd[RD], rg[RG];
  auto load_tile = [&](int t) {
    const int tx = t % tiles_x, r = t / tiles_x, ty = r % tiles_y, n = r / tiles_y;
    const int ty0 = ty * TR, tx0 = tx * TW, gy0 = ty0 * ST + kh0 - p.pad, gx0 = tx0 * ST - p.pad;
    const float* dp = p.d + (long)(n / p.dni) * p.dsg + (long)(n % p.dni) * p.dsn + (long)ty0 * p.dsh + (long)tx0 * p.dsw;
    const float* gimg = p.g + (long)(n / p.gni) * p.gsg + (long)(n % p.gni) * p.gsn;
#pragma unroll
    for (int i = 0; i < RD; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ty0 + i < p.Hd && tx0 + rxD < p.Wd && chD) v = *reinterpret_cast<const f32x4*>(dp + i * dsh + offD0);
      rd[i] = v;
    }
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      const int ly = gy0 + (gyx[i] >> 8), lx = gx0 + (gyx[i] & 255);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (tid + 256 * i < NG4 && (unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL && chG) {
        const unsigned off = __umul24(ly >> p.ups, gsh) + __umul24(lx >> p.ups, gsw) + b0 + c4;
        v = *reinterpret_cast<const f32x4*>(gimg + off);
      }
      rg[i] = v;
    }
  };
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};      // DT < NS: the bias sums come from the fp32 values (see wgrad_pc_kernel)
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < RD; ++i) {
      bf16x4 parts[NS];
      split4f<NS, FMT>(rd[i], scD, parts);
      if constexpr (DT != NS) bsum += rd[i];
      const int px = (tid + 256 * i) / (CH / 4);
#pragma unroll
      for (int s = 0; s < DT; ++s) *reinterpret_cast<bf16x4*>(&sD[s][px * LD + c4]) = parts[s];
    }
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      if (tid + 256 * i < NG4) {
        bf16x4 parts[NS];
        split4f<NS, FMT>(rg[i], scG, parts);
        const int gp = (tid + 256 * i) / (CH / 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&sG[s][gp * LD + c4]) = parts[s];
      }
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // bias gradient = sum_pixels D[pix][a]: one more MFMA chain against an
  // all-ones operand, only in the blocks that own b-tile 0 / kernel row 0
  const bool do_bias = with_bias && bt == 0 && kh0 == 0;
  f32x16 acc_bias;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_bias[r] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = FMT ? __builtin_bit_cast(__bf16, (_Float16)1.0f) : (__bf16)1.0f;

  // transposed-read lane coordinates: 16-lane group g = lane >> 4 covers columns
  // 16*(g&1).., k rows 8*(g>>1) + q (+4); lane 4q+p supplies row q, columns 4p..
  const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3, g1 = (lane >> 4) & 1, lh = lane >> 5;
  const int a_off = (8 * lh + q) * LD + wm * 32 + 16 * g1 + 4 * pp;
  const int b_off = (8 * lh + q) * ST * LD + wn * 32 + 16 * g1 + 4 * pp;

  if (t_beg < t_end) load_tile(t_beg);
  for (int t = t_beg; t < t_end; ++t) {
    __syncthreads();                          // previous tile's MFMA reads are done
    store_tile();
    __syncthreads();
    if (t + 1 < t_end) load_tile(t + 1);      // in flight during the MFMA phase
#pragma unroll
    for (int ks = 0; ks < TR; ++ks) {
      bf16x8 fa[NS];
#pragma unroll
      for (int s = 0; s < DT; ++s) fa[s] = tr_frag(sD[s], a_off + ks * 16 * LD, LD);
      if (do_bias && DT == NS) {
#pragma unroll
        for (int s = DT - 1; s >= 0; --s) {
          if constexpr (FMT == 0) acc_bias = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s], ones, acc_bias, 0, 0, 0);
          else acc_bias = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa[s]), __builtin_bit_cast(half8, ones), acc_bias, 0, 0, 0);
        }
      }
#pragma unroll
      for (int kh = 0; kh < KHB; ++kh)
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          bf16x8 fb[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) fb[s] = tr_frag(sG[s], b_off + ((ks * ST + kh) * GW + kw) * LD, ST * LD);
          if constexpr (DT == NS) {
            mfma_splitf<NS, FMT>(fa, fb, acc[kh * KS + kw]);
          } else if constexpr (FMT == 0) {
            acc[kh * KS + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc[kh * KS + kw], 0, 0, 0);
            acc[kh * KS + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[kh * KS + kw], 0, 0, 0);
          } else {
            const half8 a0 = __builtin_bit_cast(half8, fa[0]);
            acc[kh * KS + kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(half8, fb[1]), acc[kh * KS + kw], 0, 0, 0);
            acc[kh * KS + kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(half8, fb[0]), acc[kh * KS + kw], 0, 0, 0);
          }
        }
    }
  }

  const int T = p.KH * p.KW, li = lane & 31;
  const int TS = T + (with_bias ? 1 : 0);        // the bias sums travel as one more "tap" (column 0)
  float* slab = p.slabs + (long)split * p.Cd * TS * p.Cg;
  if constexpr (DT != NS) {
    // thread (pixel column tid >> 4, channels 4 (tid & 15) ..): shuffles over the lane groups, LDS over the four waves
#pragma unroll
    for (int j = 0; j < 4; ++j) { bsum[j] += __shfl_xor(bsum[j], 16, 64); bsum[j] += __shfl_xor(bsum[j], 32, 64); }
    __syncthreads();                             // every wave is done with the last tile's images
    float* bl = reinterpret_cast<float*>(&sD[0][0]);
    if (lane < 16) *reinterpret_cast<f32x4*>(&bl[wave * 64 + lane * 4]) = bsum;
    __syncthreads();
    if (do_bias && wave == 0 && lane < 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ai = a0 + lane * 4 + j;
        const float v = (bl[lane * 4 + j] + bl[64 + lane * 4 + j]) + (bl[128 + lane * 4 + j] + bl[192 + lane * 4 + j]);
        if (ai < p.Cd) slab[((long)ai * TS + T) * p.Cg] = v;
      }
    }
  }
  if (do_bias && DT == NS && wn == 0 && li == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ai = a0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ai < p.Cd) slab[((long)ai * TS + T) * p.Cg] = acc_bias[r] * inv_d;
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tap = (kh0 + t / KS) * KS + (t % KS);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ai = a0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int bi = b0 + wn * 32 + li;
      if (ai < p.Cd && bi < p.Cg) slab[((long)ai * TS + tap) * p.Cg + bi] = acc[t][r] * inv;
    }
  }
}


// ---- producer / consumer variant (stride 1, two-term splits) ---------------------------------
// The kernel above runs ONE wave per SIMD (its nine accumulator tiles need the registers), so the
// split + LDS store of every pixel tile sits between two barriers with the matrix pipe idle, and
// the tile's first fragment reads wait on the second barrier (measured: pipe 42 % busy).  Here a
// 512-thread block has two roles: waves 0-3 (one per SIMD) only issue MFMAs on the image in LDS
// buffer i & 1, waves 4-7 (their SIMD partners) load, split and store the NEXT pixel tile into the
// other buffer -- VALU / LDS-write work that runs in the shadow of the partner's MFMAs.  One
// barrier per pixel tile; 4 x 16-pixel tiles so that two images fit in LDS (2 x 49.5 KB, 3 x 3).
// BATCH: blockIdx.z selects one of up to MT_WGRAD_BATCH_MAX problems of identical geometry (the dense
// blocks of an RRDB trunk: 60 weight gradients of 130 k pixels each run as one grid instead of 60
// launches that each cut their pixels into ~100 slabs); their pointers travel in the kernel arguments.
struct WgBatch {
  const float* d[MT_WGRAD_BATCH_MAX]; const float* g[MT_WGRAD_BATCH_MAX]; float* slabs[MT_WGRAD_BATCH_MAX];
  const float* damax[MT_WGRAD_BATCH_MAX]; const float* gamax[MT_WGRAD_BATCH_MAX];
};
struct WgOne {};
// Rows of a pixel tile of wgrad_pc_kernel.  8-row tiles (possible with one term of D: no D_lo plane, and 72-element LDS
// rows) pay on SMALL images only: on the 16 x 16 images of SeparableConv4d the G patch of a tile shrinks from 1.5x to
// 1.25x its pixels (5.40 -> 5.03 ms for 256 -> 256 on 8192 images), on 128^2 / 256^2 images the doubled staging registers
// spill and the 72-element rows cost bank conflicts (1.37 -> 1.43 ms).  The rule itself is wgpc_tr() in dispatch.h (mt_wgrad_select; the host side asks mt_wgrad_plan_query).

// DT = 1 (mt_wgrad.d_terms): D enters with its LEADING 16-bit term only -- two partial products (D_hi G_hi, D_hi G_lo)
// instead of three, no D_lo plane staged or read.  D then carries 11 (fp16) significant bits: rounding errors that are
// independent from pixel to pixel, so a gradient summed over ~10^5 .. 10^6 pixels is off by ~1e-4 relative.
template <int KS, int KHB, int FMT, bool BATCH = false, int DT = 2, int TRP = 4>
__global__ __launch_bounds__(512, 2) void wgrad_pc_kernel(WgK p, int tiles_y, int tiles_x, int tiles_per_split, int ntiles,
                                                         int with_bias, int kh_base,
                                                         std::conditional_t<BATCH, WgBatch, WgOne> tab) {
  // pixel tile: 4 x 16; with one term of D (no D_lo plane) and 72-element rows two 8 x 16 images fit as well (wgpc_tr)
  constexpr int NS = 2, TR = TRP, TW = 16, PX = TR * TW, CH = 64, LD = CH + (TR == 8 ? 8 : MT_WGPC_PAD);
  static_assert(TR == 4 || (TR == 8 && DT == 1), "8-row tiles need the one-term form");
  constexpr int GH = TR - 1 + KHB, GW = TW - 1 + KS, GP = GH * GW;
  constexpr int ND4 = PX * (CH / 4), RD = ND4 / 256;
  constexpr int NG4 = GP * (CH / 4), RG = (NG4 + 255) / 256;
  constexpr int NT = KHB * KS;
  constexpr int D_E = DT * PX * LD, G_E = NS * GP * LD, BUF_E = D_E + G_E;
  static_assert(2 * BUF_E * 2 <= 160 * 1024, "two images must fit the LDS");
  __shared__ __attribute__((aligned(16))) __bf16 sm[2 * BUF_E];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  // The block's 64 x 64 channel tile = four 32-channel groups: D channels aA.., aB.. (tile columns 0-31 / 32-63 of the
  // staged D image), G channels bA.., bB...  Regular grid: 64-aligned tiles (aB = aA + 32).  Tile list (mt_wgrad.tiles,
  // 3 x 3 only): the groups are free, `quads` says which of the four 32 x 32 products (one per MFMA wave) are wanted.
  int bid, split, bz_;
  if (!wg_block(p, bid, split, bz_)) return;     // (XCD-aware block order: the tiles of one pixel range share an L2)
  int aA, aB, bA, bB, kh0;
  unsigned quads = 15u;
  bool bias_tile;
  if (KS == 3 && p.ntl > 0) {
    const int ti = bid % p.ntl; bid /= p.ntl;
    const unsigned long long e = p.tl[ti];
    aA = (int)(e & 255u) * 32; aB = (int)((e >> 8) & 255u) * 32; bA = (int)((e >> 16) & 255u) * 32; bB = (int)((e >> 24) & 255u) * 32;
    quads = (unsigned)(e >> 32) & 15u;
    bias_tile = ((e >> 40) & 1u) != 0;
    kh0 = kh_base + bid * KHB;
  } else {
    const int bt = bid % p.btiles; bid /= p.btiles;
    const int at = bid % p.atiles; bid /= p.atiles;
    kh0 = kh_base + bid * KHB;
    if (at * p.btiles + bt < 32 && ((p.skip >> (at * p.btiles + bt)) & 1u)) return;   // tile nobody reads
    aA = at * CH; aB = aA + 32; bA = bt * CH; bB = bA + 32;
    bias_tile = bt == 0;
  }
  const float *pd, *pg, *pdamax, *pgamax;
  float* pslabs;
  if constexpr (BATCH) {
    const int z = bz_;
    pd = tab.d[z]; pg = tab.g[z]; pslabs = tab.slabs[z]; pdamax = tab.damax[z]; pgamax = tab.gamax[z];
  } else {
    pd = p.d; pg = p.g; pslabs = p.slabs; pdamax = p.damax; pgamax = p.gamax;
  }
  const int t_beg = split * tiles_per_split, t_end = min(ntiles, t_beg + tiles_per_split);
  const int HL = p.Hg << p.ups, WL = p.Wg << p.ups;
  const float scD = FMT ? pow2_scale(*pdamax) : 1.f, scG = FMT ? pow2_scale(*pgamax) : 1.f;
  const float inv = 1.f / (scD * scG);

  // The bias gradient (sum of D over the pixels, the slabs' extra "tap") is accumulated by the STAGING waves from
  // the fp32 values they load anyway -- every D pixel passes through exactly one of their registers -- in plain fp32:
  // no MFMAs for it on the consumer side, and independent of how many 16-bit terms of D the products use (with
  // d_terms = 1 a bias gradient that is a small difference of large sums would otherwise inherit D's 11-bit rounding).
  const bool do_bias = with_bias && bias_tile && kh0 == 0;
  if (producer) {
    // ---- waves 4-7: global -> registers -> split -> LDS, one tile ahead of the consumers --------
    const int ptid = tid - 256;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    // two register sets: the loads of tiles i+2 and i+3 are in flight while tile i+1 is split and
    // stored (one tile of look-ahead left the loads ~1 tile time to land: less than the loaded-memory
    // latency, so the consumers waited at the barrier)
    f32x4 rd[2][RD], rg[2][RG];
    // Addressing: everything that depends on the tile (image, tile origin) is wave-uniform and
    // lives in SGPRs; a thread adds 32-bit element offsets it computes once.  (64-bit per-load
    // multiplies made the staging waves VALU-bound and kept the MFMA waves waiting at the barrier.)
    // D: float4 idx = ptid + 256 i  ->  pixel row i, column ptid >> 4, channels 4 (ptid & 15)
    const int rxD = ptid >> 4, c4 = (ptid & 15) * 4;
    const int acD = (c4 < 32 ? aA : aB - 32) + c4, bcG = (c4 < 32 ? bA : bB - 32) + c4;      // this thread's first D / G channel
    const bool chD = acD < p.Cd, chG = bcG < p.Cg, ch_full = aA + 32 <= p.Cd && aB + 32 <= p.Cd && bA + 32 <= p.Cg && bB + 32 <= p.Cg;
    const unsigned dsh = (unsigned)p.dsh, dsw = (unsigned)p.dsw, gsh = (unsigned)p.gsh, gsw = (unsigned)p.gsw;
    const unsigned offD0 = rxD * dsw + acD;
    unsigned offG[RG];          // patch pixel -> element offset from the patch origin (ups == 0)
    int gyx[RG];                // patch pixel coordinates, gy << 8 | gx
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      const int gpx = (ptid >> 4) + 16 * i, gy = gpx / GW, gx = gpx - gy * GW;
      gyx[i] = gy << 8 | gx;
      offG[i] = gy * gsh + gx * gsw + bcG;
    }
    auto load_tile = [&](int t, auto set_c) {
      constexpr int S = decltype(set_c)::value;
      const int tx = t % tiles_x, r = t / tiles_x, ty = r % tiles_y, n = r / tiles_y;
      const int ty0 = ty * TR, tx0 = tx * TW, gy0 = ty0 + kh0 - p.pad, gx0 = tx0 - p.pad;
      const float* dimg = pd + (long)(n / p.dni) * p.dsg + (long)(n % p.dni) * p.dsn;
      const float* gimg = pg + (long)(n / p.gni) * p.gsg + (long)(n % p.gni) * p.gsn;
      const float* dp = dimg + (long)ty0 * p.dsh + (long)tx0 * p.dsw;
      const bool inner = p.ups == 0 && ch_full && ty0 + TR <= p.Hd && tx0 + TW <= p.Wd && gy0 >= 0 && gy0 + GH <= HL &&
                         gx0 >= 0 && gx0 + GW <= WL;
      if (inner) {
        const float* gp = gimg + (long)gy0 * p.gsh + (long)gx0 * p.gsw;
#pragma unroll
        for (int i = 0; i < RD; ++i) rd[S][i] = *reinterpret_cast<const f32x4*>(dp + i * dsh + offD0);
#pragma unroll
        for (int i = 0; i < RG; ++i) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (ptid + 256 * i < NG4) v = *reinterpret_cast<const f32x4*>(gp + offG[i]);
          rg[S][i] = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < RD; ++i) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (ty0 + i < p.Hd && tx0 + rxD < p.Wd && chD) v = *reinterpret_cast<const f32x4*>(dp + i * dsh + offD0);
          rd[S][i] = v;
        }
#pragma unroll
        for (int i = 0; i < RG; ++i) {
          const int ly = gy0 + (gyx[i] >> 8), lx = gx0 + (gyx[i] & 255);
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (ptid + 256 * i < NG4 && (unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL && chG) {
            const unsigned off = __umul24(ly >> p.ups, gsh) + __umul24(lx >> p.ups, gsw) + bcG;
            v = *reinterpret_cast<const f32x4*>(gimg + off);
          }
          rg[S][i] = v;
        }
      }
    };
    // tile i lives in LDS buffer i & 1 and travels through register set i & 1
    auto store_tile = [&](auto set_c) {
      constexpr int S = decltype(set_c)::value;
      __bf16* sD = sm + S * BUF_E;
      __bf16* sG = sD + D_E;
#pragma unroll
      for (int i = 0; i < RD; ++i) {
        const int idx = ptid + 256 * i, px = idx / (CH / 4), c = (idx % (CH / 4)) * 4;
        bf16x4 parts[NS];
        split4f<NS, FMT, MT_WGPC_SHADOW>(rd[S][i], scD, parts);
        bsum += rd[S][i];                    // (pixels outside the image were loaded as zeros)
#pragma unroll
        for (int s = 0; s < DT; ++s) *reinterpret_cast<bf16x4*>(&sD[s * PX * LD + px * LD + c]) = parts[s];
      }
#pragma unroll
      for (int i = 0; i < RG; ++i) {
        const int idx = ptid + 256 * i, gpx = idx / (CH / 4), c = (idx % (CH / 4)) * 4;
        if (idx < NG4) {
          bf16x4 parts[NS];
          split4f<NS, FMT, MT_WGPC_SHADOW>(rg[S][i], scG, parts);
#pragma unroll
          for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&sG[s * GP * LD + gpx * LD + c]) = parts[s];
        }
      }
    };
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    const int nt = t_end - t_beg;
    if (nt > 0) load_tile(t_beg, S0);
    if (nt > 1) load_tile(t_beg + 1, S1);
    if (nt > 0) store_tile(S0);
    if (nt > 2) load_tile(t_beg + 2, S0);
    __syncthreads();
    for (int i = 0; i < nt; i += 2) {          // one barrier per tile, as in the consumer loop
      if (i + 1 < nt) store_tile(S1);
      if (i + 3 < nt) load_tile(t_beg + i + 3, S1);
      __syncthreads();
      if (i + 1 >= nt) break;
      if (i + 2 < nt) store_tile(S0);
      if (i + 4 < nt) load_tile(t_beg + i + 4, S0);
      __syncthreads();
    }
    // bias sums: thread (pixel column ptid >> 4, channels 4 (ptid & 15) ..) -> the 16 pixel columns of a channel quad sit
    // in four lane groups of each of the four staging waves: shuffles inside a wave, then LDS (both images are dead)
    bsum[0] += __shfl_xor(bsum[0], 16, 64); bsum[1] += __shfl_xor(bsum[1], 16, 64);
    bsum[2] += __shfl_xor(bsum[2], 16, 64); bsum[3] += __shfl_xor(bsum[3], 16, 64);
    bsum[0] += __shfl_xor(bsum[0], 32, 64); bsum[1] += __shfl_xor(bsum[1], 32, 64);
    bsum[2] += __shfl_xor(bsum[2], 32, 64); bsum[3] += __shfl_xor(bsum[3], 32, 64);
    float* bl = reinterpret_cast<float*>(sm);
    if (do_bias && lane < 16) *reinterpret_cast<f32x4*>(&bl[(wave - 4) * 64 + lane * 4]) = bsum;
    __syncthreads();                         // (matched by the consumers' barrier after their tile loop)
    if (do_bias && wave == 4 && lane < 16) {
      const int TSb = p.KH * p.KW + 1;
      float* slab = pslabs + (long)split * p.Cd * TSb * p.Cg;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ai = (lane < 8 ? aA : aB - 32) + lane * 4 + j;
        const float v = (bl[lane * 4 + j] + bl[64 + lane * 4 + j]) + (bl[128 + lane * 4 + j] + bl[192 + lane * 4 + j]);
        if (ai < p.Cd) slab[((long)ai * TSb + p.KH * p.KW) * p.Cg] = v;
      }
    }
    return;
  }

  // ---- waves 0-3: MFMAs only ------------------------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  const bool active = ((quads >> (wm * 2 + wn)) & 1u) != 0;      // (wave-uniform: a listed tile may not want this wave's product)
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3, g1 = (lane >> 4) & 1, lh = lane >> 5;
  const int a_off = (8 * lh + q) * LD + wm * 32 + 16 * g1 + 4 * pp;
  const int b_off = (8 * lh + q) * LD + wn * 32 + 16 * g1 + 4 * pp;

  __syncthreads();
  for (int t = t_beg; t < t_end; ++t) {
    const __bf16* sD = sm + ((t - t_beg) & 1) * BUF_E;
    const __bf16* sG = sD + D_E;
    // Walk the PATCH rows: the G fragments of patch row r serve the pixel rows r, r-1, .. r-KHB+1
    // (kernel rows 0 .. KHB-1), so each is read from LDS once per tile instead of KHB times, and the
    // D fragments of the last KHB pixel rows stay in registers.  (Fragment reads were the bound of
    // the pixel-row walk: 40 ds_read_b64_tr per 27 MFMAs and four waves on one LDS port.)
    // (Requesting row r + 1 before the MFMAs of row r -- a fourth D slot and a second G set -- was
    // measured and gained nothing: 1.87 vs 1.89 ms on the 256 x 256 x 128 x 128 layer.)
    bf16x8 fa[KHB][NS];
    if (active) {
#pragma unroll
    for (int r = 0; r < GH; ++r) {
      bf16x8 fb[KS][NS];
#pragma unroll
      for (int kw = 0; kw < KS; ++kw)
#pragma unroll
        for (int s = 0; s < NS; ++s) fb[kw][s] = tr_frag(sG + s * GP * LD, b_off + (r * GW + kw) * LD, LD);
      if (r < TR) {
#pragma unroll
        for (int s = 0; s < DT; ++s) fa[r % KHB][s] = tr_frag(sD + s * PX * LD, a_off + r * 16 * LD, LD);
      }
#pragma unroll
      for (int kh = 0; kh < KHB; ++kh) {
        const int ks = r - kh;                 // pixel row this patch row feeds through kernel row kh
        if (ks >= 0 && ks < TR) {
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            if constexpr (DT == 2) {
              mfma_splitf<NS, FMT>(fa[ks % KHB], fb[kw], acc[kh * KS + kw]);
            } else if constexpr (FMT == 0) {
              acc[kh * KS + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks % KHB][0], fb[kw][1], acc[kh * KS + kw], 0, 0, 0);
              acc[kh * KS + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks % KHB][0], fb[kw][0], acc[kh * KS + kw], 0, 0, 0);
            } else {
              const half8 a0 = __builtin_bit_cast(half8, fa[ks % KHB][0]);
              acc[kh * KS + kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(half8, fb[kw][1]), acc[kh * KS + kw], 0, 0, 0);
              acc[kh * KS + kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(half8, fb[kw][0]), acc[kh * KS + kw], 0, 0, 0);
            }
          }
        }
      }
    }
    }
    __syncthreads();
  }
  __syncthreads();                           // (the staging waves' bias sums pass through LDS behind this barrier)
  if (!active) return;

  const int T = p.KH * p.KW, li = lane & 31;
  const int TS = T + (with_bias ? 1 : 0);
  float* slab = pslabs + (long)split * p.Cd * TS * p.Cg;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tap = (kh0 + t / KS) * KS + (t % KS);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ai = (wm ? aB : aA) + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int bi = (wn ? bB : bA) + li;
      if (ai < p.Cd && bi < p.Cg) slab[((long)ai * TS + tap) * p.Cg + bi] = acc[t][r] * inv;
    }
  }
}


// ---- 16 gathered channels (FlowEstimator's 10 -> 128 entry, RRDBNet's 9 -> 64 entry) ---------------------------
// With G only 16 channels wide a 64-column B tile of the kernel above is three quarters padding, and a 5 x 5 layer
// pays that for 25 taps in three launches.  Here the 64 columns of the B tile are FOUR TAPS x 16 channels: column
// block s of accumulator j belongs to tap 4 j + s, i.e. each 16-lane column block of a transposed fragment read
// simply starts at a different pixel of the G patch.  25 taps -> 7 accumulators (9 -> 3), all kernel rows in one
// launch, a quarter of the MFMAs, and the G patch staged per tile is 16 channels instead of 64.
template <int KS, int FMT>
__global__ __launch_bounds__(512, 2) void wgrad_thin_kernel(WgK p, int tiles_y, int tiles_x, int tiles_per_split, int ntiles,
                                                           int with_bias) {
  constexpr int NS = 2, TR = 4, TW = 16, PX = TR * TW, CH = 64, LD = CH + MT_WGPC_PAD, CG = 16, LDG = CG + 8;
  constexpr int GH = TR - 1 + KS, GW = TW - 1 + KS, GP = GH * GW, PADK = KS / 2;
  constexpr int ND4 = PX * (CH / 4), RD = ND4 / 256;
  constexpr int NG4 = GP * (CG / 4), RG = (NG4 + 255) / 256;
  constexpr int T = KS * KS, NG = (T + 3) / 4;
  constexpr int D_E = NS * PX * LD, G_E = NS * GP * LDG, BUF_E = D_E + G_E;
  __shared__ __attribute__((aligned(16))) __bf16 sm[2 * BUF_E];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  int at, split, bz_;
  if (!wg_block(p, at, split, bz_)) return;
  const int a0 = at * CH;
  const int t_beg = split * tiles_per_split, t_end = min(ntiles, t_beg + tiles_per_split);
  const float scD = FMT ? pow2_scale(*p.damax) : 1.f, scG = FMT ? pow2_scale(*p.gamax) : 1.f;
  const float inv = 1.f / (scD * scG), inv_d = 1.f / scD;

  if (producer) {
    const int ptid = tid - 256;
    f32x4 rd[2][RD], rg[2][RG];
    const int rxD = ptid >> 4, c4 = (ptid & 15) * 4;
    const bool chD = a0 + c4 < p.Cd;
    const unsigned dsh = (unsigned)p.dsh, dsw = (unsigned)p.dsw, gsh = (unsigned)p.gsh, gsw = (unsigned)p.gsw;
    const unsigned offD0 = rxD * dsw + a0 + c4;
    // G: float4 idx = ptid + 256 i -> patch pixel idx >> 2, channels 4 (idx & 3)
    const int cg4 = (ptid & 3) * 4;
    int gyx[RG];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
      const int gpx = (ptid >> 2) + 64 * i, gy = gpx / GW, gx = gpx - gy * GW;
      gyx[i] = gy << 8 | gx;
    }
    auto load_tile = [&](int t, auto set_c) {
      constexpr int S = decltype(set_c)::value;
      const int tx = t % tiles_x, r = t / tiles_x, ty = r % tiles_y, n = r / tiles_y;
      const int ty0 = ty * TR, tx0 = tx * TW, gy0 = ty0 - PADK, gx0 = tx0 - PADK;
      const float* dp = p.d + (long)(n / p.dni) * p.dsg + (long)(n % p.dni) * p.dsn + (long)ty0 * p.dsh + (long)tx0 * p.dsw;
      const float* gimg = p.g + (long)(n / p.gni) * p.gsg + (long)(n % p.gni) * p.gsn;
#pragma unroll
      for (int i = 0; i < RD; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ty0 + i < p.Hd && tx0 + rxD < p.Wd && chD) v = *reinterpret_cast<const f32x4*>(dp + i * dsh + offD0);
        rd[S][i] = v;
      }
#pragma unroll
      for (int i = 0; i < RG; ++i) {
        const int ly = gy0 + (gyx[i] >> 8), lx = gx0 + (gyx[i] & 255);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ptid + 256 * i < NG4 && (unsigned)ly < (unsigned)p.Hg && (unsigned)lx < (unsigned)p.Wg)
          v = *reinterpret_cast<const f32x4*>(gimg + (__umul24(ly, gsh) + __umul24(lx, gsw) + cg4));
        rg[S][i] = v;
      }
    };
    auto store_tile = [&](auto set_c) {
      constexpr int S = decltype(set_c)::value;
      __bf16* sD = sm + S * BUF_E;
      __bf16* sG = sD + D_E;
#pragma unroll
      for (int i = 0; i < RD; ++i) {
        const int idx = ptid + 256 * i, px = idx / (CH / 4), c = (idx % (CH / 4)) * 4;
        bf16x4 parts[NS];
        split4f<NS, FMT, MT_WGPC_SHADOW>(rd[S][i], scD, parts);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&sD[s * PX * LD + px * LD + c]) = parts[s];
      }
#pragma unroll
      for (int i = 0; i < RG; ++i) {
        const int idx = ptid + 256 * i;
        if (idx < NG4) {
          bf16x4 parts[NS];
          split4f<NS, FMT, MT_WGPC_SHADOW>(rg[S][i], scG, parts);
#pragma unroll
          for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&sG[s * GP * LDG + (idx >> 2) * LDG + cg4]) = parts[s];
        }
      }
    };
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    const int nt = t_end - t_beg;
    if (nt > 0) load_tile(t_beg, S0);
    if (nt > 1) load_tile(t_beg + 1, S1);
    if (nt > 0) store_tile(S0);
    if (nt > 2) load_tile(t_beg + 2, S0);
    __syncthreads();
    for (int i = 0; i < nt; i += 2) {
      if (i + 1 < nt) store_tile(S1);
      if (i + 3 < nt) load_tile(t_beg + i + 3, S1);
      __syncthreads();
      if (i + 1 >= nt) break;
      if (i + 2 < nt) store_tile(S0);
      if (i + 4 < nt) load_tile(t_beg + i + 4, S0);
      __syncthreads();
    }
    return;
  }

  // ---- waves 0-3: MFMAs only ------------------------------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const bool do_bias = with_bias != 0;
  f32x16 acc_bias;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_bias[r] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = FMT ? __builtin_bit_cast(__bf16, (_Float16)1.0f) : (__bf16)1.0f;
  const int l16 = lane & 15, q = l16 >> 2, pp = l16 & 3, g1 = (lane >> 4) & 1, lh = lane >> 5;
  const int a_off = (8 * lh + q) * LD + wm * 32 + 16 * g1 + 4 * pp;
  // column block (wn, g1) of accumulator j = tap 4 j + 2 wn + g1: its fragment rows start at that tap's pixel
  int b_off[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    int tap = 4 * j + 2 * wn + g1;
    if (tap >= T) tap = T - 1;                 // (an unused column block: any valid address, its sums are dropped)
    const int kh = tap / KS, kw = tap - kh * KS;
    b_off[j] = (kh * GW + kw + 8 * lh + q) * LDG + 4 * pp;
  }

  __syncthreads();
  for (int t = t_beg; t < t_end; ++t) {
    const __bf16* sD = sm + ((t - t_beg) & 1) * BUF_E;
    const __bf16* sG = sD + D_E;
#pragma unroll
    for (int ks = 0; ks < TR; ++ks) {
      bf16x8 fa[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) fa[s] = tr_frag(sD + s * PX * LD, a_off + ks * 16 * LD, LD);
      if (do_bias) {
#pragma unroll
        for (int s = NS - 1; s >= 0; --s) {
          if constexpr (FMT == 0) acc_bias = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s], ones, acc_bias, 0, 0, 0);
          else acc_bias = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa[s]), __builtin_bit_cast(half8, ones), acc_bias, 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        bf16x8 fb[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) fb[s] = tr_frag(sG + s * GP * LDG, b_off[j] + ks * GW * LDG, LDG);
        mfma_splitf<NS, FMT>(fa, fb, acc[j]);
      }
    }
    __syncthreads();
  }

  const int li = lane & 31;
  const int TS = T + (with_bias ? 1 : 0);
  float* slab = p.slabs + (long)split * p.Cd * TS * p.Cg;
  if (do_bias && wn == 0 && li == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ai = a0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ai < p.Cd) slab[((long)ai * TS + T) * p.Cg] = acc_bias[r] * inv_d;
    }
  }
  const int tap_s = 2 * wn + (li >> 4), bi = li & 15;
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int tap = 4 * j + tap_s;
    if (tap >= T) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ai = a0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ai < p.Cd) slab[((long)ai * TS + tap) * p.Cg + bi] = acc[j][r] * inv;
    }
  }
}

}  // namespace

// Called by mt_wgrad_run: returns 1 if the launch was taken by the halo kernel.
// `splits` blocks along y each own a contiguous range of pixel tiles.
int mt_wgrad_halo_try_launch(const mt_wgrad* d, void* stream) {
  const int prec = d->precision;
  const MtWgradSel sel = mt_wgrad_select(d);          // (dispatch.cpp: the one place that chooses)
  if (!sel.halo) return 0;                            // (a tile list was checked by mt_wgrad_run: mt_wgrad_tiles_check)
  if (prec == MT_PREC_F16X3 && (!d->d_amax || !d->g_amax)) return 0;
  WgK k;
  k.d = d->d.p; k.dsn = d->d.sn; k.dsh = d->d.sh; k.dsw = d->d.sw; k.Hd = d->d.H; k.Wd = d->d.W; k.Cd = d->d.C;
  k.g = d->g.p; k.gsn = d->g.sn; k.gsh = d->g.sh; k.gsw = d->g.sw; k.Hg = d->g.H; k.Wg = d->g.W; k.Cg = d->g.C;
  k.dni = d->d.n_in; k.dsg = d->d.sg; k.gni = d->g.n_in; k.gsg = d->g.sg;
  k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.ups = d->ups;
  k.M = d->d.N * d->d.H * d->d.W; k.per_split = 0;
  k.atiles = (k.Cd + 63) / 64; k.btiles = (k.Cg + 63) / 64;
  k.slabs = d->slabs;
  k.damax = d->d_amax; k.gamax = d->g_amax;
  k.skip = d->skip_tiles;
  wgk_tiles(d, &k);
  const bool pc = sel.pc, tall = sel.tall, thin = sel.thin;
  const unsigned ctiles = k.ntl > 0 ? k.ntl : k.atiles * k.btiles;      // channel tiles of a producer / consumer launch
  dim3 gpc;                                                              // its grid (wg_grid also fills k's block-order fields: before any launch reads k)
  const int trows = sel.trows, dterms = sel.d_terms;
  const int tiles_y = (k.Hd + trows - 1) / trows, tiles_x = (k.Wd + 15) / 16;
  const int ntiles = d->d.N * tiles_y * tiles_x;
  const int tps = (ntiles + d->splits - 1) / d->splits;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define MT_WGH(KS, KHB, NS, FMT, OCC, GRID, KH_BASE) \
  hipLaunchKernelGGL((wgrad_halo_kernel<KS, KHB, NS, FMT, OCC>), GRID, dim3(256), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, KH_BASE)
#define MT_WGH2(KS, KHB, FMT, GRID, KH_BASE) \
  do { if (dterms == 1) hipLaunchKernelGGL((wgrad_halo_kernel<KS, KHB, 2, FMT, 1, 2, 4, 1>), GRID, dim3(256), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, KH_BASE); \
       else hipLaunchKernelGGL((wgrad_halo_kernel<KS, KHB, 2, FMT, 1, 2>), GRID, dim3(256), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, KH_BASE); } while (0)
#define MT_WGPC(KS, KHB, FMT, KH_BASE) \
  do { if (dterms == 1) hipLaunchKernelGGL((wgrad_pc_kernel<KS, KHB, FMT, false, 1>), gpc, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, KH_BASE, WgOne{}); \
       else hipLaunchKernelGGL((wgrad_pc_kernel<KS, KHB, FMT>), gpc, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, KH_BASE, WgOne{}); } while (0)
  if (thin) {
    const dim3 grid(wg_grid(&k, k.atiles, d->splits));
    if (d->KH == 3) {
      if (prec == MT_PREC_F16X3) hipLaunchKernelGGL((wgrad_thin_kernel<3, 1>), grid, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias);
      else hipLaunchKernelGGL((wgrad_thin_kernel<3, 0>), grid, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias);
    } else {
      if (prec == MT_PREC_F16X3) hipLaunchKernelGGL((wgrad_thin_kernel<5, 1>), grid, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias);
      else hipLaunchKernelGGL((wgrad_thin_kernel<5, 0>), grid, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias);
    }
    return 1;
  }
  if (pc) {
    gpc = dim3(wg_grid(&k, ctiles, d->splits));
    const bool f16 = prec == MT_PREC_F16X3;
    if (d->KH == 3 && f16 && trows == 8)
      hipLaunchKernelGGL((wgrad_pc_kernel<3, 3, 1, false, 1, 8>), gpc, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, 0, WgOne{});
    else if (d->KH == 3) { if (f16) MT_WGPC(3, 3, 1, 0); else MT_WGPC(3, 3, 0, 0); }
    // 5 x 5: kernel rows {0,1} | {2,3} | {4} (15 accumulator tiles would not leave the MFMA waves
    // their operand registers at two waves per SIMD; the staging is off the critical path here)
    else if (f16) { MT_WGPC(5, 2, 1, 0); MT_WGPC(5, 2, 1, 2); MT_WGPC(5, 1, 1, 4); }
    else { MT_WGPC(5, 2, 0, 0); MT_WGPC(5, 2, 0, 2); MT_WGPC(5, 1, 0, 4); }
    return 1;
  }
#undef MT_WGPC
  if (d->stride == 2) {      // stride-2 gather: all 9 taps, or kernel rows {0,1,2} | {3,4}
    dim3 grid(wg_grid(&k, k.atiles * k.btiles, d->splits));
    const bool f16 = prec == MT_PREC_F16X3;
    if (d->KH == 3) { if (f16) MT_WGH2(3, 3, 1, grid, 0); else MT_WGH2(3, 3, 0, grid, 0); }
    else if (f16) { MT_WGH2(5, 3, 1, grid, 0); MT_WGH2(5, 2, 1, grid, 3); }
    else { MT_WGH2(5, 3, 0, grid, 0); MT_WGH2(5, 2, 0, grid, 3); }
  } else if (d->KH == 3) {   // all nine taps per block, one block per CU (the 9 accumulator tiles need the registers)
    dim3 grid(wg_grid(&k, k.atiles * k.btiles, d->splits));
    if (prec == MT_PREC_BF16X6) MT_WGH(3, 3, 3, 0, 1, grid, 0);
    else if (tall && prec == MT_PREC_BF16X3)
      hipLaunchKernelGGL((wgrad_halo_kernel<3, 3, 2, 0, 1, 1, 8>), grid, dim3(256), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, 0);
    else if (tall)
      hipLaunchKernelGGL((wgrad_halo_kernel<3, 3, 2, 1, 1, 1, 8>), grid, dim3(256), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, 0);
    else if (prec == MT_PREC_BF16X3) MT_WGH(3, 3, 2, 0, 1, grid, 0);
    else MT_WGH(3, 3, 2, 1, 1, grid, 0);
  } else if (sel.rows5) {   // one kernel row (5 taps) per block
    dim3 grid(wg_grid(&k, k.atiles * k.btiles * 5, d->splits));
    if (prec == MT_PREC_BF16X6) MT_WGH(5, 1, 3, 0, 2, grid, 0);
    else if (prec == MT_PREC_BF16X3) MT_WGH(5, 1, 2, 0, 2, grid, 0);
    else MT_WGH(5, 1, 2, 1, 2, grid, 0);
  } else {            // kernel rows {0,1,2} and {3,4}: 15 / 10 taps per block, one block per CU; each
                      // pixel tile is staged twice instead of five times
    dim3 grid(wg_grid(&k, k.atiles * k.btiles, d->splits));
    if (prec == MT_PREC_BF16X3) { MT_WGH(5, 3, 2, 0, 1, grid, 0); MT_WGH(5, 2, 2, 0, 1, grid, 3); }
    else { MT_WGH(5, 3, 2, 1, 1, grid, 0); MT_WGH(5, 2, 2, 1, 1, grid, 3); }
  }
#undef MT_WGH
#undef MT_WGH2
  return 1;
}



// Several weight gradients of IDENTICAL geometry as one grid (blockIdx.z = problem): 3 x 3, stride 1,
// two-term split precisions, i.e. what the producer / consumer kernel takes.  The descriptors may
// differ only in the tensor pointers, the slab pointer and the max|.| scalars.
extern "C" int mt_wgrad_run_batch(const mt_wgrad* descs, int32_t n, void* stream) {
  MT_REQUIRE(descs && n >= 1 && n <= MT_WGRAD_BATCH_MAX, "mt_wgrad_run_batch: 1..%d problems (got %d)", MT_WGRAD_BATCH_MAX, n);
  const mt_wgrad* d = &descs[0];
  const int prec = d->precision;
  MT_REQUIRE(mt_view_ok(&d->d) && mt_view_ok(&d->g) && d->splits >= 1, "mt_wgrad_run_batch: invalid views / splits");
  MT_REQUIRE(d->stride == 1 && d->KH == 3 && d->KW == 3 && d->pad == 1 && d->ups == 0 &&
             (prec == MT_PREC_F16X3 || prec == MT_PREC_BF16X3) && d->algo == MT_ALGO_AUTO,
             "mt_wgrad_run_batch: only stride-1 3x3 layers in BF16X3 / F16X3 (use mt_wgrad_run)");
  MT_REQUIRE(d->d.N == d->g.N && d->d.H == d->g.H && d->d.W == d->g.W, "mt_wgrad_run_batch: D and G geometry differ");
  MT_REQUIRE(d->d.sh < (1 << 24) && d->d.sw < (1 << 24) && d->g.sh < (1 << 24) && d->g.sw < (1 << 24) &&
             (long)d->d.H * d->d.sh < (1L << 31) && (long)d->g.H * d->g.sh < (1L << 31), "mt_wgrad_run_batch: image too large for 32-bit offsets");
  WgBatch tab;
  for (int i = 0; i < n; ++i) {
    const mt_wgrad* e = &descs[i];
    const bool same = e->d.N == d->d.N && e->d.H == d->d.H && e->d.W == d->d.W && e->d.C == d->d.C && e->d.sn == d->d.sn &&
                      e->d.sh == d->d.sh && e->d.sw == d->d.sw && e->d.n_in == d->d.n_in && e->d.sg == d->d.sg &&
                      e->g.N == d->g.N && e->g.H == d->g.H && e->g.W == d->g.W && e->g.C == d->g.C && e->g.sn == d->g.sn &&
                      e->g.sh == d->g.sh && e->g.sw == d->g.sw && e->g.n_in == d->g.n_in && e->g.sg == d->g.sg &&
                      e->KH == d->KH && e->KW == d->KW && e->stride == d->stride && e->pad == d->pad && e->ups == d->ups &&
                      e->splits == d->splits && e->precision == d->precision && e->algo == d->algo &&
                      e->with_bias == d->with_bias && e->skip_tiles == d->skip_tiles && e->d_terms == d->d_terms &&
                      e->n_tiles == d->n_tiles && memcmp(e->tiles, d->tiles, sizeof(d->tiles)) == 0;
    MT_REQUIRE(same, "mt_wgrad_run_batch: problem %d differs from problem 0 in geometry", i);
    MT_REQUIRE(e->d.p && e->g.p && e->slabs && (reinterpret_cast<uintptr_t>(e->d.p) & 15) == 0 && (reinterpret_cast<uintptr_t>(e->g.p) & 15) == 0,
               "mt_wgrad_run_batch: problem %d: null / unaligned pointers", i);
    MT_REQUIRE(prec != MT_PREC_F16X3 || (e->d_amax && e->g_amax), "mt_wgrad_run_batch: F16X3 needs d_amax and g_amax (problem %d)", i);
    tab.d[i] = e->d.p; tab.g[i] = e->g.p; tab.slabs[i] = e->slabs; tab.damax[i] = e->d_amax; tab.gamax[i] = e->g_amax;
  }
  for (int i = n; i < MT_WGRAD_BATCH_MAX; ++i) { tab.d[i] = tab.g[i] = nullptr; tab.slabs[i] = nullptr; tab.damax[i] = tab.gamax[i] = nullptr; }
  WgK k;
  k.d = d->d.p; k.dsn = d->d.sn; k.dsh = d->d.sh; k.dsw = d->d.sw; k.Hd = d->d.H; k.Wd = d->d.W; k.Cd = d->d.C;
  k.g = d->g.p; k.gsn = d->g.sn; k.gsh = d->g.sh; k.gsw = d->g.sw; k.Hg = d->g.H; k.Wg = d->g.W; k.Cg = d->g.C;
  k.dni = d->d.n_in; k.dsg = d->d.sg; k.gni = d->g.n_in; k.gsg = d->g.sg;
  k.KH = d->KH; k.KW = d->KW; k.stride = d->stride; k.pad = d->pad; k.ups = d->ups;
  k.M = d->d.N * d->d.H * d->d.W; k.per_split = 0;
  k.atiles = (k.Cd + 63) / 64; k.btiles = (k.Cg + 63) / 64;
  k.slabs = d->slabs; k.damax = d->d_amax; k.gamax = d->g_amax; k.skip = d->skip_tiles;
  wgk_tiles(d, &k);
  const MtWgradSel sel = mt_wgrad_select(d);
  { const int rc = mt_wgrad_tiles_check(d, sel); if (rc) return rc; }
  MT_REQUIRE(sel.halo && sel.pc, "mt_wgrad_run_batch: the producer / consumer kernel does not take this layer (mt_wgrad_plan_query)");
  const int trb = prec == MT_PREC_F16X3 ? wgpc_tr(sel.d_terms, 3, k.Hd) : 4, dterms = sel.d_terms;
  const int tiles_y = (k.Hd + trb - 1) / trb, tiles_x = (k.Wd + 15) / 16;
  const int ntiles = d->d.N * tiles_y * tiles_x;
  const int tps = (ntiles + d->splits - 1) / d->splits;
  const dim3 grid(wg_grid(&k, k.ntl > 0 ? k.ntl : k.atiles * k.btiles, d->splits, n));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (prec == MT_PREC_F16X3 && dterms == 1 && trb == 8)
    hipLaunchKernelGGL((wgrad_pc_kernel<3, 3, 1, true, 1, 8>), grid, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, 0, tab);
  else if (prec == MT_PREC_F16X3 && dterms == 1)
    hipLaunchKernelGGL((wgrad_pc_kernel<3, 3, 1, true, 1>), grid, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, 0, tab);
  else if (prec == MT_PREC_F16X3)
    hipLaunchKernelGGL((wgrad_pc_kernel<3, 3, 1, true>), grid, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, 0, tab);
  else if (dterms == 1)
    hipLaunchKernelGGL((wgrad_pc_kernel<3, 3, 0, true, 1>), grid, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, 0, tab);
  else
    hipLaunchKernelGGL((wgrad_pc_kernel<3, 3, 0, true>), grid, dim3(512), 0, s, k, tiles_y, tiles_x, tps, ntiles, d->with_bias, 0, tab);
  return mt_check_launch("mt_wgrad_run_batch");
}
