// 3x3 stride-1 layers with at most 64 output channels on launches that fill the chip: the xi-split Winograd F(2,3)
// kernel of conv_halo_wino.hip (igemm_winox_kernel) with the software pipeline of igemm_wino16_kernel.
//
// Why (block timelines from wall-clock stamps, DESIGN.md "What the small launches wait for"): in igemm_winox_kernel a
// 32-channel chunk costs a block 1.8 us on a CU it shares with two others, although its MFMAs, its input transform
// (VALU) and its halo loads (TA) each need about 0.5 us of their unit: the image is single-buffered, so the three
// run one after the other between two barriers, in lock-step across the blocks of the launch.  Here ONE 512-thread
// block per CU owns a 16 x 16-pixel tile: waves = 2 row halves x 4 xi, the transformed image of the 18 x 18 halo
//   V[plane][row 18][xi 4][pair 8][32 ch]      (64-byte entries, rows padded by 16 B)
// is double-buffered (148 KB), and a wave interleaves, per kernel row of its xi,
//   * the 12 TN MFMAs of the tap (A fragments from LDS, B fragments from global memory one or two taps ahead),
//   * the transform + split + LDS stores of ONE of the three item rounds of the NEXT chunk (into the other buffer),
//   * the global loads of the same round for the chunk AFTER that (a whole chunk of latency to hide in),
// with one barrier per chunk.  The VALU work runs in the shadow of the matrix pipe and the other wave of the SIMD.
// Weight fragments: the layout of igemm_winox_kernel (mt_weight_pack_halo, fragment order 2) -- the packs are shared.
#include <cstdlib>
#include "conv_common.h"

namespace {

#ifndef MT_WINOX_SHADOW
#define MT_WINOX_SHADOW 2
#endif
#if MT_WINOX_TRACE            // (lab: wall-clock stamps of the stages of every block, read back by mt_lab_winoy_trace)
__device__ unsigned long long winoy_trace[4096 * 16];
#define WY_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) winoy_trace[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define WY_STAMP(i) do {} while (0)
#endif

template <int FMT, int TN>
__global__ __launch_bounds__(512, 1) void igemm_winoy_kernel(ConvK p, int tiles_y, int tiles_x) {
  constexpr int NS = 2, CK = 32, TH = 16, TW = 16, BN = 32 * TN, TM = 2;
  constexpr int NTX = TW / 2, ROWS = TH + 2;
  constexpr int ENT = CK, ROW_E = 4 * NTX * ENT + 8, PLANE_E = ROWS * ROW_E, BUF_E = NS * PLANE_E;
  constexpr int NIT = ROWS * NTX * (CK / 4), RI = (NIT + 511) / 512;
  constexpr int KSTEPS = CK / 16;
  constexpr int BAHEAD = TN == 1 ? 2 : 1;            // weight fragments: taps ahead (three register sets: slot = kernel row)
  static_assert(RI == 3, "one transform round per kernel row");
  static_assert(2 * BUF_E * 2 <= 160 * 1024, "LDS");
  static_assert(2 * 4 * TM * TN * 16 * 64 * 4 <= 2 * BUF_E * 2, "the epilogue exchange must fit the image buffers");
  __shared__ __attribute__((aligned(16))) __bf16 sV[2 * BUF_E];

  WY_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm2 = wave >> 2, xi = wave & 3;
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
      for (int x4 = 0; x4 < 4; ++x4) {
        bf16x4 parts[NS];
        split4f<NS, FMT, MT_WINOX_SHADOW>(v[x4], scA, parts);
#pragma unroll
        for (int s = 0; s < NS; ++s)
          *reinterpret_cast<bf16x4*>(dst + s * PLANE_E + x4 * NTX * ENT) = parts[s];
      }
    }
  };
  // weight fragments: [tap 12 = ky * 4 + xi][chunk][ntile][wn TN][kstep][plane][lane] x 16 B (mt_weight_pack_halo, order 2)
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK, ntile_w = (p.Co + BN - 1) / BN;
  bf16x8 fb[3][TN][KSTEPS][NS];
  auto load_B = [&](bf16x8 (&f)[TN][KSTEPS][NS], int ch, int ky) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const bf16x8* g = wfr + ((((long)(ky * 4 + xi) * nchunk + ch) * ntile_w + blockIdx.y) * TN + tn) * (KSTEPS * NS * 64) +
                          ks * NS * 64 + lane;
#pragma unroll
        for (int s = 0; s < NS; ++s) f[tn][ks][s] = g[s * 64];
      }
  };

  // ---- prologue: the whole first image, the second chunk's items on their way --------------------------------
  f32x4 rh[RI][4];
#pragma unroll
  for (int i = 0; i < RI; ++i) load_item(rh[i], i, 0);
  load_B(fb[0], 0, 0);
  if (BAHEAD == 2) load_B(fb[1], 0, 1);
  // what the epilogue reads that does not depend on the result: requested here, so that no load follows a store
  float bv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) bv[tn] = (p.bias && n0 + tn * 32 + li < p.Co) ? p.bias[n0 + tn * 32 + li] : 0.f;
  WY_STAMP(1);
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    store_item(rh[i], i, 0);
    if (nchunk > 1) load_item(rh[i], i, CK);
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][tn][r] = 0.f;
  // MFMA row i of M tile a = (output row 8 wm2 + 4a + (i >> 3), column pair i & 7); this wave's xi plane of the image
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = (8 * wm2 + 4 * a + (li >> 3)) * ROW_E + (li & 7) * ENT + 8 * lh + xi * NTX * ENT;
  WY_STAMP(2);
  __syncthreads();
  WY_STAMP(3);

  for (int ch = 0; ch < nchunk; ++ch) {
    const bool more = ch + 1 < nchunk, more2 = ch + 2 < nchunk;
    const __bf16* vb = sV + (ch & 1) * BUF_E;
    const int nb = (ch & 1) ^ 1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      // the weight fragments of the tap BAHEAD taps on (slot = its kernel row)
      {
        constexpr int ahead = BAHEAD;
        const int k2 = (ky + ahead) % 3;
        if (ky + ahead < 3) load_B(fb[k2], ch, k2);
        else if (more) load_B(fb[k2], ch + 1, k2);
      }
      const int toff = ky * ROW_E;
      // The MFMAs of the tap, product by product (lo*hi, hi*lo, hi*hi for every accumulator), and between the product
      // groups one quarter (one xi) of the transform + split + LDS stores of round ky of the NEXT chunk's image (loaded a
      // chunk ago; into the other buffer): a wave issues in order, so the VALU work has to sit BETWEEN the MFMAs in
      // program order to run in their shadow.  The round's registers then take the same round of the chunk after that.
      const bool item = more && (it_ok[ky] & 16u);
      f32x4 tv[4];
      if (item) {
        const f32x4 (&d)[4] = rh[ky];
        tv[0] = d[0] - d[2]; tv[1] = d[1] + d[2]; tv[2] = d[2] - d[1]; tv[3] = d[1] - d[3];
      }
      auto piece = [&](int x4) {
        if (item) {
          bf16x4 parts[NS];
          split4f<NS, FMT, MT_WINOX_SHADOW>(tv[x4], scA, parts);
          __bf16* dst = sV + nb * BUF_E + it_dst[ky] + x4 * NTX * ENT;
#pragma unroll
          for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(dst + s * PLANE_E) = parts[s];
        }
      };
      auto prod = [&](int pi, const bf16x8 (&a)[NS], const bf16x8 (&b)[NS], f32x16& c) {
        const int ia = pi == 0 ? 1 : 0, ib = pi == 1 ? 1 : 0;
        if constexpr (FMT == 1)
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a[ia]), __builtin_bit_cast(half8, b[ib]), c, 0, 0, 0);
        else
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], b[ib], c, 0, 0, 0);
      };
      bf16x8 fa[KSTEPS][TM][NS];
      auto load_A = [&](int ks) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int s = 0; s < NS; ++s)
            fa[ks][a][s] = *reinterpret_cast<const bf16x8*>(&vb[s * PLANE_E + a_base[a] + toff + ks * 16]);
      };
      load_A(0);
      if (TN == 1) load_A(1);                    // (64 accumulator registers: one k step of A fragments at a time)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        if (TN != 1 && ks == 1) load_A(1);
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int a = 0; a < TM; ++a) prod(pi, fa[ks][a], fb[ky][tn][ks], acc[a][tn]);
          const int g = ks * 3 + pi;             // product group 0 .. 5: quarters after groups 0, 2, 3, 5
          if (g == 0) piece(0);
          if (g == 2) piece(1);
          if (g == 3) piece(2);
          if (g == 5) piece(3);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (more && more2) load_item(rh[ky], ky, (ch + 2) * CK);
    }
    if (ch == 0) WY_STAMP(4);
    if (more) __syncthreads();      // the other image is complete, nobody reads this one any more
    if (ch == 0) WY_STAMP(5);
    if (ch == 1) WY_STAMP(6);
  }
  WY_STAMP(8);

  // ---- epilogue: the four xi of a row half meet through LDS (both M tiles at once); wave (wm2, xi) finishes output
  // rows 8 wm2 + 4a + xi.  Two phases (conv_common.h: halo_epi_load / halo_epi_store): no load follows a store.
  float* xch = reinterpret_cast<float*>(sV);       // [wm2 2][xi 4][a TM][tn][16][64 lanes]
  __syncthreads();                                 // the images are dead
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[((((wm2 * 4 + xi) * TM + a) * TN + tn) * 16 + r) * 64 + lane] = acc[a][tn][r];
  __syncthreads();
  WY_STAMP(11);
  float vals[TN][8 * TM];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * xi + q;                  // C/D rows (r & 3) + 8 (r >> 2) + 4 lh: output row r >> 2 = xi of the M tile
        float m[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) m[x] = xch[((((wm2 * 4 + x) * TM + a) * TN + tn) * 16 + r) * 64 + lane];
        vals[tn][8 * a + 2 * q] = (m[0] + m[1] + m[2]) * inv;
        vals[tn][8 * a + 2 * q + 1] = (m[1] - m[2] - m[3]) * inv;
      }
  auto pix = [&](int e, int& oy, int& ox) {
    oy = ty0 + 8 * wm2 + 4 * (e >> 3) + xi;
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
  WY_STAMP(9);
  finish_amax(p, vmax, sV);
  WY_STAMP(10);
}

}  // namespace

// MT355_WINOY=0 keeps igemm_winox_kernel on every layer it takes (A/B switch, read once).
static bool halo_winoy_enabled() {
  static const bool on = [] { const char* e = getenv("MT355_WINOY"); return !(e && e[0] == '0'); }();
  return on;
}

// A layer of the xi-split kernel goes to the pipelined form when its 16 x 16 tiles give every CU a block (one block
// per CU: a launch with fewer tiles is better off with the 8 x 16 tiles of igemm_winox_kernel, three blocks per CU).
bool halo_winoy_wanted(int N, int Ho, int Wo) {
  static const int min_tiles = [] { const char* e = getenv("MT355_WINOY_TILES"); return e ? atoi(e) : 256; }();
  return halo_winoy_enabled() && Ho >= 16 && (long)N * ((Ho + 15) / 16) * ((Wo + 15) / 16) >= min_tiles;
}

void mt_halo_winoy_launch(const void* kp, int fmt, void* stream) {
  const ConvK& k = *static_cast<const ConvK*>(kp);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_y = (k.Ho + 15) / 16, tiles_x = (k.Wo + 15) / 16;
  const unsigned nt = (unsigned)k.N * tiles_y * tiles_x;
  if (k.Co > 32) {
    if (fmt) hipLaunchKernelGGL((igemm_winoy_kernel<1, 2>), dim3(nt, 1), dim3(512), 0, s, k, tiles_y, tiles_x);
    else hipLaunchKernelGGL((igemm_winoy_kernel<0, 2>), dim3(nt, 1), dim3(512), 0, s, k, tiles_y, tiles_x);
  } else {
    if (fmt) hipLaunchKernelGGL((igemm_winoy_kernel<1, 1>), dim3(nt, 1), dim3(512), 0, s, k, tiles_y, tiles_x);
    else hipLaunchKernelGGL((igemm_winoy_kernel<0, 1>), dim3(nt, 1), dim3(512), 0, s, k, tiles_y, tiles_x);
  }
}

#if MT_WINOX_TRACE
extern "C" int mt_lab_winoy_trace(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(winoy_trace), sizeof(unsigned long long) * n);
}
#endif
