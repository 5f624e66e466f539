// Stride-1 3x3 "same" convolutions with 128-wide output-channel tiles on a 1-D Winograd F(4, 3) transform along W
// (round 3): the layers that carry most of the DFPN / CHN FLOPs -- SeparableConv4d and FlowEstimator's 3 x 3 layers
// (model_dfpn.py:596-605,686-711), and with tap-flipped, channel-swapped weights their backward-data.
//
//   y[oy][4t + j] = sum_ky  A^T ( (G g[ky][:]) .* (B^T x[oy + ky - 1][4t - 1 .. 4t + 4]) ),   j = 0 .. 3
//
// with the interpolation points 0, 1, -1, 2, -2, inf (the six points of the F(2,5) kernel, conv_halo_wino25.hip):
//   B^T = [ 4  0 -5  0  1  0 ]      G = 1/24 [  6   0   0 ]      A^T = [ 1  1  1  1  1  0 ]
//         [ 0 -4 -4  1  1  0 ]               [ -4  -4  -4 ]            [ 0  1 -1  2 -2  0 ]
//         [ 0  4 -4 -1  1  0 ]               [ -4   4  -4 ]            [ 0  1  1  4  4  0 ]
//         [ 0 -2 -1  2  1  0 ]               [  1   2   4 ]            [ 0  1 -1  8 -8  1 ]
//         [ 0  2 -1 -2  1  0 ]               [  1  -2   4 ]
//         [ 0  4  0 -5  0  1 ]               [  0   0  24 ]
// Four output pixels of a row cost 3 kernel rows x 6 products instead of 3 x 3 x 4: HALF the MFMAs of the direct form,
// three quarters of F(2,3)'s.  The kernels of this family run at a power-limited clock (1.75 GHz under sustained MFMA
// load), where halving the LDS and L1 operand traffic bought nothing (igemm_wino16x_kernel: a
// tie) while every MFMA that is not issued does: F(2,5) gained 25 - 33 % on the 5 x 5 layers.  In the engine's
// arithmetic (operands split into two fp16 terms after the transform, exact products, fp32 accumulation) the six-point
// transform costs a factor ~3 in error against the direct form and stays at fp32 rounding level
// (scratch/r3_wino43_cond.py; tests/test_gpu_kernels.py::test_conv3x3_f16x3_is_fp32_grade).  fp16 split only: the 16
// bits of a two-term bf16 split are not enough for six points (the never-differentiated VGG passes keep F(2,3)).
//
// Structure = igemm_wino25_kernel: one 512-thread block per CU owns 16 x 16 output pixels x 128 channels; the image
//   V[plane][row 18][xi 6][quad 4][32 ch]      (64-byte entries, rows padded by 16 B)
// of the 18 x 22 halo is double-buffered in LDS (2 x 55 KB); the eight waves split the XI axis: waves = 2 xi halves x 4
// channel quarters, a wave owns all 64 (row, quad) GEMM rows x 32 channels x three xi (96 accumulator registers) and
// walks its 9 of the 18 (ky, xi) taps; weight fragments come straight from global memory two taps ahead; the next
// chunk's image is built in two rounds under the current chunk's taps; one barrier per 32-channel chunk; the two xi
// halves exchange their shares of the output transform through LDS in the epilogue.
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"

namespace {

#if MT_W43_TRACE              // (lab: wall-clock stamps of the stages of every block, read back by mt_lab_w43_trace)
__device__ unsigned long long w43_trace[8192 * 8];
#define W43_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) w43_trace[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define W43_STAMP(i) do {} while (0)
#endif

#ifndef MT_W43_PREFETCH
#define MT_W43_PREFETCH 0            // (1: lab build with the successor's L2 prefetch -- measured and rejected, see below)
#endif
#ifndef MT_W43_SKIP_R1
#define MT_W43_SKIP_R1 1              // (0: A/B builds that always run the second transform round)
#endif
#ifndef MT_W43_SCALAR_XFORM
#define MT_W43_SCALAR_XFORM 0         // (1: the input transform without packed-fp32 instructions; needs -fno-slp-vectorize)
#endif
#ifndef MT_W43_SPLIT
#define MT_W43_SPLIT MT_SHADOW_DEFAULT   // split form of the transformed values (conv_common.h: split4f SHADOW)
#endif
#ifndef MT_W43_NT_LOAD
#define MT_W43_NT_LOAD 0              // (lab: halo loads as non-temporal streams)
#endif
#ifndef MT_W43_NT_STORE
#define MT_W43_NT_STORE 0             // (lab: output stores as non-temporal streams)
#endif
#ifndef MT_W43_NTILE_MAJOR
#define MT_W43_NTILE_MAJOR 0          // (lab: block order n-tile major inside an XCD)
#endif
#ifndef MT_W43_STAGGER
#define MT_W43_STAGGER 0              // (lab: the two xi halves transform at different taps)
#endif
#ifndef MT_W43_ABL
#define MT_W43_ABL 0                  // (lab ablations, bit mask: 1 no weight-fragment reloads, 2 no per-tap image reads, 4 no in-loop transform)
#endif
#ifndef MT_W43_PREFETCH_AHEAD
#define MT_W43_PREFETCH_AHEAD 32     // blocks of one XCD in flight: 32 CUs x one block
#endif

template <int FMT>
__global__ __launch_bounds__(512, 2) void igemm_wino43_kernel(ConvK p, int tiles_y, int tiles_x) {
  constexpr int NS = 2, CK = 32, TH = 16, TW = 16, BN = 128, TM = 2, NXI = 6;
  constexpr int NQ = TW / 4, ROWS = TH + 2;
  constexpr int ENT = CK, ROW_E = NXI * NQ * ENT + 8, PLANE_E = ROWS * ROW_E, BUF_E = NS * PLANE_E;
  constexpr int NIT = ROWS * NQ * (CK / 4), RI = (NIT + 511) / 512;       // 576 items: a full round and an eighth of one
  constexpr int TW_ = 9, KSTEPS = CK / 16;                                // taps of one wave: 3 kernel rows x 3 xi
  constexpr int XCH_E = 8 * 64 * 64 * 2;                                  // epilogue exchange (bf16 units): 8 waves x 64 floats x 64 lanes
  constexpr int SM_E = 2 * BUF_E > XCH_E ? 2 * BUF_E : XCH_E;
  static_assert(RI == 2, "two transform rounds per chunk");
  static_assert(SM_E * 2 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) __bf16 sV[SM_E];

  W43_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xh = wave >> 2, wn = wave & 3;                 // xi half, channel quarter
  const int li = lane & 31, lh = lane >> 5;
  // 1-D grid over (pixel tile, n-tile), as igemm_wino16_kernel
  const int ntn = (p.Co + BN - 1) / BN, npt = p.N * tiles_y * tiles_x;
  const int lid = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  int bid, nty;
  {
    const int q8 = npt >> 3, r8 = npt & 7;
    const int mine = q8 + (xcd < r8 ? 1 : 0), start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
#if MT_W43_NTILE_MAJOR
    // (lab) all pixel tiles of an XCD for n-tile 0, then for n-tile 1: the weight fragments in flight are ONE n-tile's
    // (2.4 MB for a 256 -> 256 layer: fits the 4 MB L2) at the price of reading every halo from HBM / Infinity Cache twice
    const int per = gridDim.x / 8 / ntn;            // block slots of this XCD per n-tile (the grid is padded to 8 * ceil(nt / 8) * ntn)
    nty = lid / per;
    const int tl = lid - nty * per;
#else
    const int tl = lid / ntn;
    nty = lid - tl * ntn;
#endif
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
  // the transformed image carries a further 2^-4 (|B^T d| <= 10 max|x|)
  const float scA = pow2_scale(*p.xamax) * 0.0625f;
  const float inv = 1.f / (scA * pow2_scale(*p.wamax));
  const float bv = (p.bias && n0 + wn * 32 + li < p.Co) ? p.bias[n0 + wn * 32 + li] : 0.f;    // (read here: no load after a store)
  const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;
  const int xsw = (int)p.xsw;

  // transform items (halo row r, column quad t, channel quad c4), c4 fastest; round i owns q = tid + 512 i.
  // Round 0 = the 16 halo rows 1 .. 16 (the tile's own rows: 512 items, every thread one), round 1 = halo rows 0 and 17
  // (64 items: wave 0 alone -- the wave every other wave of the block then waits for at the chunk's barrier).  When both
  // of those rows lie outside the image -- every tile of a 16 x 16 image: the SeparableConv4d layers, 80 % of this
  // kernel's work -- round 1 is all zeros: the two rows are zero-filled once and the round is skipped (round 4).
  const bool r1_live = !MT_W43_SKIP_R1 || ty0 > 0 || ty0 + TH < HL;
  int it_row[RI], it_lx[RI], it_dst[RI];
  unsigned it_ok[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int q = tid + 512 * i;
    const int c4 = q & 7, t = (q >> 3) & 3, rq = q >> 5;
    const int r = rq < TH ? rq + 1 : (rq == TH ? 0 : ROWS - 1);
    const int ly = ty0 - 1 + r;
    it_ok[i] = 0;
    it_lx[i] = tx0 - 1 + 4 * t;
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
      if (((it_ok[i] >> b) & 1u) && cin) {
#if MT_W43_NT_LOAD
        v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xin + (it_row[i] + ((it_lx[i] + b) >> p.ups) * xsw + c0)));
#else
        v = *reinterpret_cast<const f32x4*>(xin + (it_row[i] + ((it_lx[i] + b) >> p.ups) * xsw + c0));
#endif
      }
      d[b] = v;
    }
  };
  auto store_item = [&](const f32x4 (&d)[6], int i, int buf) {
    if (it_ok[i] & 64u) {
      // B^T d with shared terms: 12 multiply-adds for the six xi
#if MT_W43_SCALAR_XFORM
      // (lab, round 4) element by element in plain fp32 instructions: on f32x4 values hipcc emits v_pk_fma_f32 /
      // v_pk_add_f32 / v_pk_mul_f32, which run on the pipe the MFMAs of the SIMD's other wave need (scratch/lab/coissue.hip)
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
      for (int x6 = 0; x6 < NXI; ++x6) {
        bf16x4 parts[NS];
        split4f<NS, FMT, MT_W43_SPLIT>(v[x6], scA, parts);
#pragma unroll
        for (int s = 0; s < NS; ++s)
          *reinterpret_cast<bf16x4*>(dst + s * PLANE_E + x6 * NQ * ENT) = parts[s];
      }
    }
  };
  // weight fragments: [tap 18 = ky * 6 + xi][chunk][ntile][wn 4][kstep][plane][lane] x 16 B
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK, ntile_w = (p.Co + BN - 1) / BN;
  bf16x8 fb[2][KSTEPS][NS];
  // wave tap w (0 .. 8) = kernel row w / 3, xi 3 xh + w % 3
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
    if (r1_live) load_item(r1, 1, 0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) { load_B(fb[0][ks], 0, 0, ks); load_B(fb[1][ks], 0, 1, ks); }
    store_item(r0, 0, 0);
    if (r1_live) {
      store_item(r1, 1, 0);
    } else {
      // halo rows 0 and ROWS - 1 of both images and both planes stay zero for the whole block
      constexpr int ROW8 = ROW_E / 4;                        // 8-byte units per row
      for (int e = tid; e < 2 * NS * 2 * ROW8; e += 512) {
        const int u = e % ROW8, w_ = e / ROW8;               // w_: (buffer, plane, which row)
        const int row = (w_ & 1) ? ROWS - 1 : 0, plane = (w_ >> 1) % NS, buf = w_ / (2 * NS);
        *reinterpret_cast<bf16x4*>(sV + buf * BUF_E + plane * PLANE_E + row * ROW_E + 4 * u) = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      }
    }
  }
  f32x16 acc[TM][3];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int xl = 0; xl < 3; ++xl)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][xl][r] = 0.f;
  // MFMA row i of M tile a = (output row 8 a + (i >> 2), column quad i & 3); this wave's xi half of the image
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = (8 * a + (li >> 2)) * ROW_E + (li & 3) * ENT + 8 * lh + 3 * xh * NQ * ENT;
  __syncthreads();

  f32x4 rh[6];
  // A wave walks 9 taps per chunk -- an odd number -- so the fragment slot of a tap alternates from chunk to chunk:
  // the chunk body is instantiated for both parities (PAR = chunk & 1) and the slot index stays a compile-time constant.
  auto chunk = [&](int ch, auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    const bool more = ch + 1 < nchunk;
    const __bf16* vb = sV + PAR * BUF_E;
    const int nb = PAR ^ 1;
#if MT_W43_ABL & 2
    bf16x8 fa[KSTEPS][TM][NS];
#endif
#pragma unroll
    for (int w = 0; w < TW_; ++w) {
      const int ky = w / 3, xl = w % 3;
      const int slot = (w + PAR) & 1;                  // global tap counter 9 ch + w, modulo 2
      const int toff = ky * ROW_E + xl * NQ * ENT;
      // the next chunk's image: round 0 is loaded at tap 0 and written at tap 4, round 1 at taps 5 / 8
#if MT_W43_ABL & 4
      (void)more;
#elif MT_W43_STAGGER
      // (lab) the two waves of a SIMD are the two xi halves of one channel quarter: give them different taps for their VALU burst
      if (((w == (xh ? 2 : 0)) || (w == (xh ? 6 : 4) && r1_live)) && more) load_item(rh, w == (xh ? 2 : 0) ? 0 : 1, (ch + 1) * CK);
#else
      if ((w == 0 || (w == 5 && r1_live)) && more) load_item(rh, w == 0 ? 0 : 1, (ch + 1) * CK);
#endif
#if MT_W43_ABL & 2             // (lab ablation bit 1: the image fragments are read for the first tap of a chunk only)
      if (w == 0)
#else
      bf16x8 fa[KSTEPS][TM][NS];
#endif
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
#if !(MT_W43_ABL & 1)          // (lab ablation bit 0: the weight fragments are never reloaded -- wrong results, timing only)
        if (w + 2 < TW_) load_B(fb[slot][ks], ch, w + 2, ks);
        else if (more) load_B(fb[slot][ks], ch + 1, w + 2 - TW_, ks);
#endif
      }
#if MT_W43_ABL & 4
#elif MT_W43_STAGGER
      if (((w == (xh ? 5 : 3)) || (w == 8 && r1_live)) && more) store_item(rh, w == 8 ? 1 : 0, nb);
#else
      if ((w == 4 || (w == 8 && r1_live)) && more) store_item(rh, w == 4 ? 0 : 1, nb);
#endif
    }
    if (more) __syncthreads();      // the other image is complete, nobody reads this one any more
  };
  W43_STAMP(1);
  for (int ch = 0; ch < nchunk; ch += 2) {
    chunk(ch, std::integral_constant<int, 0>{});
    if (ch + 1 < nchunk) chunk(ch + 1, std::integral_constant<int, 1>{});
  }

  W43_STAMP(2);
#if MT_W43_PREFETCH
  // ---- L2 prefetch for the block that takes this CU next (round 4) -----------------------------------------------------
  // One 512-thread block per CU and 32 CUs per XCD: the block that starts when this one retires is, to a good
  // approximation, lid + 32 of the same XCD -- the same L2.  Its prologue waits for the first chunk of its halo with
  // nothing to overlap (4.9 of a block's 38 us): touch that chunk now (one dword per pixel = per
  // 128-byte line), so that the loads of its prologue are L2 hits.  The value is consumed after the last store.
  // MEASURED AND REJECTED (round 4, alternating builds on one box, scratch/r4_pf43.sh): igemm_wino43_kernel per DFPN step
  // 33.17 / 33.32 ms without, 33.91 / 33.82 ms with (32 ahead), 33.65 (64 ahead), 33.45 (16 ahead): the prologue's latency is
  // not an L2-miss latency -- the touch only adds traffic in front of the stores.  Off by default.
  float pf_touch = 0.f;
  {
    const int q8 = npt >> 3, r8 = npt & 7;
    const int mine = q8 + (xcd < r8 ? 1 : 0), start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int tl2 = (lid + MT_W43_PREFETCH_AHEAD) / ntn;
    const int r = tid / 22, c = tid - r * 22;
    if (tl2 < mine && tl2 != lid / ntn && r < ROWS) {
      int b2 = start + tl2;
      const int tx2 = b2 % tiles_x; b2 /= tiles_x;
      const int ty2 = b2 % tiles_y;
      const int n2 = b2 / tiles_y;
      const int ly = ty2 * TH - 1 + r, lx = tx2 * TW - 1 + c;
      if ((unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL)
        pf_touch = p.x[(long)(n2 / p.xni) * p.xsg + (long)(n2 % p.xni) * p.xsn + (long)(ly >> p.ups) * p.xsh + (long)(lx >> p.ups) * p.xsw];
    }
  }
#endif
  // ---- epilogue ---------------------------------------------------------------------------------------------
  // Output transform  y0 = m0 + m1 + m2 + m3 + m4,  y1 = m1 - m2 + 2 m3 - 2 m4,  y2 = m1 + m2 + 4 m3 + 4 m4,
  // y3 = m1 - m2 + 8 m3 - 8 m4 + m5.  xi half 0 holds m0 m1 m2, half 1 m3 m4 m5: each reduces them to its shares of
  // (y0 .. y3); half 0 finishes M tile 0, half 1 finishes M tile 1, and each hands the other the four shares of the
  // tile it does not finish: 64 floats per lane through LDS, lane-contiguous (wave-uniform branches on the half).
  auto share = [&](const f32x16 (&m)[3], int r, float (&s4)[4]) {
    if (xh == 0) {
      const float e = m[1][r] + m[2][r], o = m[1][r] - m[2][r];
      s4[0] = m[0][r] + e; s4[1] = o; s4[2] = e; s4[3] = o;
    } else {
      const float e = m[0][r] + m[1][r], o = m[0][r] - m[1][r];
      s4[0] = e; s4[1] = 2.f * o; s4[2] = 4.f * e; s4[3] = 8.f * o + m[2][r];
    }
  };
  __syncthreads();                    // every wave is done with the last image: the LDS is free
  float* xch = reinterpret_cast<float*>(sV);
  {
    float* mine = xch + (wave * 64) * 64 + lane;            // [wave][4 shares x 16][64 lanes]
    auto put = [&](const f32x16 (&m)[3]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s4[4];
        share(m, r, s4);
#pragma unroll
        for (int j = 0; j < 4; ++j) mine[(j * 16 + r) * 64] = s4[j];
      }
    };
    if (xh == 0) put(acc[1]); else put(acc[0]);             // the M tile the OTHER half finishes
  }
  __syncthreads();
  const float* theirs = xch + ((wave ^ 4) * 64) * 64 + lane;
  W43_STAMP(3);
  const int o = n0 + wn * 32 + li;
  float vmax = 0.f;
  auto finish = [&](const f32x16 (&m)[3]) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {      // 8 accumulator registers = 32 output pixels at a time
      float vals[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = 8 * hh + q;
        float s4[4];
        share(m, r, s4);
#pragma unroll
        for (int j = 0; j < 4; ++j) vals[q * 4 + j] = (s4[j] + theirs[(j * 16 + r) * 64]) * inv;
      }
      // element (q, j): C/D row (r & 3) + 8 (r >> 2) + 4 lh -> output row 8 xh + 2 (r >> 2) + lh, column quad r & 3
      const int oyb = ty0 + 8 * xh + lh;
      vmax = fmaxf(vmax, halo_epilogue<32, MT_W43_NT_STORE != 0>(p, n, o, vals, [&](int e, int& oy, int& ox) {
        const int r = 8 * hh + (e >> 2), j = e & 3;
        oy = oyb + 2 * (r >> 2);
        ox = tx0 + 4 * (r & 3) + j;
      }, bv));
    }
  };
  if (xh == 0) finish(acc[0]); else finish(acc[1]);
  W43_STAMP(4);
  finish_amax(p, vmax, sV);
  W43_STAMP(5);
#if MT_W43_PREFETCH
  if (pf_touch == 1.2345e-30f) p.y[0] = pf_touch;       // (never true for real data: keeps the touch alive)
#endif
}

__global__ void weight_pack_halo_wino43_kernel(const float* src, int A, int B, long sa, long sb, long st, __bf16* dst,
                                               int nchunk, int ntile, int flip, const float* wamax, int nchunk_total,
                                               int ch_off) {
  pack_halo_wino43_body(src, A, B, sa, sb, st, dst, nchunk, ntile, flip, wamax, nchunk_total, ch_off);
}

}  // namespace

void mt_halo_wino43_launch(const void* kp, void* stream) {
  const ConvK& k = *static_cast<const ConvK*>(kp);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_y = (k.Ho + 15) / 16, tiles_x = (k.Wo + 15) / 16;
  const unsigned nt = (unsigned)k.N * tiles_y * tiles_x, ntn = (k.Co + 127) / 128;
  const unsigned grid = 8 * ((nt + 7) / 8) * ntn;       // (padded so that every XCD gets the same number of block slots)
  ConvK kk = k;
  kk.amax_deferred = kk.yamax && kk.ascr && grid >= MT_AMAX_DEFER_BLOCKS;   // (several rounds of blocks: fold the slots afterwards)
  hipLaunchKernelGGL((igemm_wino43_kernel<1>), dim3(grid), dim3(512), 0, s, kk, tiles_y, tiles_x);
  if (kk.amax_deferred) mt_amax_fold_launch(kk.ascr, kk.yamax, s);
}

void mt_pack_halo_wino43_launch(const float* src, int A, int B, long sa, long sb, long st, void* dst, int nchunk, int ntile,
                                int flip, const float* wamax, int nchunk_total, int ch_off, void* stream) {
  const long tot = (long)18 * nchunk * ntile * 4 * 1024;
  hipLaunchKernelGGL(weight_pack_halo_wino43_kernel, dim3(mt_grid_1d(tot, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     src, A, B, sa, sb, st, reinterpret_cast<__bf16*>(dst), nchunk, ntile, flip, wamax, nchunk_total, ch_off);
}

#if MT_W43_TRACE
extern "C" __attribute__((visibility("default"))) int mt_lab_w43_trace(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(w43_trace), sizeof(unsigned long long) * n);
}
#endif
