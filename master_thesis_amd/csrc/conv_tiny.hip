// Layers with a handful of channels on one side (model_vgg.py:40 conv1_1 3 -> 64, model_dfpn.py:596-605 the
// 1 -> 128 / 128 -> 1 ends of SeparableConv4d, model_dfpn.py:686-711 the 128 -> 2 end of FlowEstimator and
// the backward-data / weight gradients that mirror them).  On the matrix-core kernels these pad 1 / 2 / 3
// channels to a 32-channel K chunk (or a 32-wide N tile) and spend their time staging zeros; their real
// arithmetic is tiny (K = 9 .. 27) and the launches are bound by the 1 GB they write or read.  Here they
// run as plain fp32 FMAs on the VALU -- exact fp32, no operand split, no packed weights -- laid out so
// that the wide tensor streams through once in 256 / 512-byte pixel rows.
#include <cstdlib>
#include "conv_common.h"
#include "dispatch.h"

namespace {

// ---- forward / backward-data: stride-1 3x3 "same", <= 4 real input channels, 64 or 128 output channels ----
// A block owns ROWS x 16 output pixels (ROWS = 256 / LG pixel rows, one per thread group) x all output
// channels: thread (g, l) keeps the 16 pixels of tile row g x output channels 4l .. 4l+3 in registers.  The
// halo of the (<= 4 channel) input sits in LDS as fp32 planes, the weights as [k = (tap, ci)][Co]; a thread
// reads an 18-float input row once per (ci, dy) (every lane of its group reads the same addresses:
// broadcasts) and one float4 of weights per k.  A group's stores are whole 256 / 512-byte pixel rows.
template <int LG>
__global__ __launch_bounds__(256) void conv_tiny_in_kernel(ConvK p, const float* __restrict__ w, int cin, long so, long si,
                                                         long st, int flip, int tiles_y, int tiles_x, int ntiles) {
  constexpr int ROWS = 256 / LG, TW = 16, HR = ROWS + 2, HC = TW + 2, HSTR = 20;
  __shared__ __attribute__((aligned(16))) float sX[4][HR * HSTR];
  __shared__ __attribute__((aligned(16))) float sW[36][4 * LG];
  const int tid = threadIdx.x, g = tid / LG, l = tid % LG;

  // weights: k = tap * cin + ci, tap = kh * 3 + kw (flip: the tap-reversed kernel of a backward-data pass)
  for (int idx = tid; idx < 9 * cin * 4 * LG; idx += 256) {
    const int k = idx / (4 * LG), co = idx - k * (4 * LG);
    const int tap = k / cin, ci = k - tap * cin;
    sW[k][co] = w[co * so + ci * si + (flip ? 8 - tap : tap) * st];
  }
  float4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + 4 * l);
  float vmax = 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % tiles_x, r_ = tile / tiles_x, ty = r_ % tiles_y, n = r_ / tiles_y;
    const int ty0 = ty * ROWS, tx0 = tx * TW;
    const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;
    __syncthreads();                       // (weights staged / the previous tile's halo is no longer read)
    for (int idx = tid; idx < HR * HC; idx += 256) {
      const int r = idx / HC, c = idx - r * HC;
      const int ly = ty0 - 1 + r, lx = tx0 - 1 + c;
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)ly < (unsigned)p.H && (unsigned)lx < (unsigned)p.W)
        v = *reinterpret_cast<const float4*>(xin + (long)ly * p.xsh + (long)lx * p.xsw);
      sX[0][r * HSTR + c] = v.x; sX[1][r * HSTR + c] = v.y; sX[2][r * HSTR + c] = v.z; sX[3][r * HSTR + c] = v.w;
    }
    __syncthreads();

    float4 acc[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) acc[i] = float4{0.f, 0.f, 0.f, 0.f};
    for (int ci = 0; ci < cin; ++ci) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        float row[HSTR];
        const float4* rp = reinterpret_cast<const float4*>(&sX[ci][(g + dy) * HSTR]);
#pragma unroll
        for (int q = 0; q < HSTR / 4; ++q) {
          const float4 t = rp[q];
          row[4 * q] = t.x; row[4 * q + 1] = t.y; row[4 * q + 2] = t.z; row[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float4 wv = *reinterpret_cast<const float4*>(&sW[(dy * 3 + dx) * cin + ci][4 * l]);
#pragma unroll
          for (int i = 0; i < TW; ++i) {
            const float xv = row[i + dx];
            acc[i].x = fmaf(wv.x, xv, acc[i].x); acc[i].y = fmaf(wv.y, xv, acc[i].y);
            acc[i].z = fmaf(wv.z, xv, acc[i].z); acc[i].w = fmaf(wv.w, xv, acc[i].w);
          }
        }
      }
    }

    // ---- epilogue: bias, activation, alpha, gate by act'(saved output), max|y| ----------------------------
    const int oy = ty0 + g;
    if (oy < p.Ho) {
      float* yb = p.y + ((long)(n / p.yni) * p.ysg + (long)(n % p.yni) * p.ysn + (long)oy * p.ysh + 4 * l);
      const float* db = p.ds ? p.ds + ((long)n * p.dsn + (long)oy * p.dsh + 4 * l) : nullptr;
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        const int ox = tx0 + i;
        if (ox >= p.Wo) continue;          // (no break: the accumulators must keep static register indices)
        float4 v;
        v.x = mt_act_fn(acc[i].x + bias4.x, p.act) * p.alpha; v.y = mt_act_fn(acc[i].y + bias4.y, p.act) * p.alpha;
        v.z = mt_act_fn(acc[i].z + bias4.z, p.act) * p.alpha; v.w = mt_act_fn(acc[i].w + bias4.w, p.act) * p.alpha;
        if (db) {
          const float4 s = *reinterpret_cast<const float4*>(db + (long)ox * p.dsw);
          v.x *= mt_dact_fn(s.x, p.dact); v.y *= mt_dact_fn(s.y, p.dact);
          v.z *= mt_dact_fn(s.z, p.dact); v.w *= mt_dact_fn(s.w, p.dact);
        }
        *reinterpret_cast<float4*>(yb + (long)ox * p.ysw) = v;
        vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
    }
  }
  finish_amax(p, vmax, sX);
}

// ---- weight gradient: stride-1 3x3 "same", one side with <= 4 real channels, the other 64 or 128 wide ------
//   dW[a][tap][b] = sum_p D[p][a] * G[p + tap - 1][b]
// The WIDE tensor is streamed once (a float4 of channels per thread and pixel, 256 / 512-byte rows per thread
// group), the 3x3 neighbourhood of the TINY tensor comes out of an LDS halo as broadcasts:
//   wide = D (tiny input channels):   acc[tap][t] += D[p] * G_t[p + tap - 1]
//   wide = G (tiny output channels):  dW[t][tap][w] = sum_q G[q][w] * D_t[q - (tap - 1)]: the same loop with the
//                                     taps reversed (FLIP).
// A block walks tiles blockIdx.x, + gridDim.x, ...; its partial sums go to slab blockIdx.x (= split) in the
// layout of the other weight-gradient kernels ([a][tap (+ bias)][b]), so mt_wgrad_reduce finishes the job.
template <int LG, int NT, bool WIDE_D>
__global__ __launch_bounds__(256) void wgrad_tiny_kernel(WgK p, int with_bias, int tiles_y, int tiles_x, int ntiles) {
  constexpr int ROWS = 256 / LG, TW = 16, HR = ROWS + 2, HC = TW + 2, HSTR = 20;
  __shared__ __attribute__((aligned(16))) float sT[NT][HR * HSTR];
  __shared__ __attribute__((aligned(16))) float sR[ROWS][4 * LG + 4];       // cross-group reduction, one (tap, t) at a time
  const int tid = threadIdx.x, g = tid / LG, l = tid % LG;
  const float* wp = WIDE_D ? p.d : p.g;
  const float* tp = WIDE_D ? p.g : p.d;
  const long wsn = WIDE_D ? p.dsn : p.gsn, wsh = WIDE_D ? p.dsh : p.gsh, wsw = WIDE_D ? p.dsw : p.gsw;
  const long tsn = WIDE_D ? p.gsn : p.dsn, tsh = WIDE_D ? p.gsh : p.dsh, tsw = WIDE_D ? p.gsw : p.dsw;
  const int wni = WIDE_D ? p.dni : p.gni, tni = WIDE_D ? p.gni : p.dni;
  const long wsg = WIDE_D ? p.dsg : p.gsg, tsg = WIDE_D ? p.gsg : p.dsg;
  const int H = p.Hd, W = p.Wd;            // (stride 1, 'same': both tensors have the same extent)

  float4 acc[9][NT];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[k][t] = float4{0.f, 0.f, 0.f, 0.f};
  float4 bsum = {0.f, 0.f, 0.f, 0.f};      // bias sums: wide D -> per wide channel; tiny D -> lane 0 keeps NT sums in .x .. .w

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % tiles_x, r_ = tile / tiles_x, ty = r_ % tiles_y, n = r_ / tiles_y;
    const int ty0 = ty * ROWS, tx0 = tx * TW;
    const float* wimg = wp + (long)(n / wni) * wsg + (long)(n % wni) * wsn;
    const float* timg = tp + (long)(n / tni) * tsg + (long)(n % tni) * tsn;
    // the wide tensor's 16 pixels of this thread's row (issued before the halo is staged)
    float4 wv[TW];
    const int oy = ty0 + g;
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if (oy < H && tx0 + i < W) v = *reinterpret_cast<const float4*>(wimg + (long)oy * wsh + (long)(tx0 + i) * wsw + 4 * l);
      wv[i] = v;
    }
    __syncthreads();                       // the previous tile's halo is no longer read
    for (int idx = tid; idx < HR * HC; idx += 256) {
      const int r = idx / HC, c = idx - r * HC;
      const int ly = ty0 - 1 + r, lx = tx0 - 1 + c;
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)ly < (unsigned)H && (unsigned)lx < (unsigned)W)
        v = *reinterpret_cast<const float4*>(timg + (long)ly * tsh + (long)lx * tsw);
      sT[0][r * HSTR + c] = v.x;
      if constexpr (NT > 1) sT[1][r * HSTR + c] = v.y;
      if constexpr (NT > 2) sT[2][r * HSTR + c] = v.z;
      if constexpr (NT > 3) sT[3][r * HSTR + c] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        float row[HSTR];
        const float4* rp = reinterpret_cast<const float4*>(&sT[t][(g + dy) * HSTR]);
#pragma unroll
        for (int q = 0; q < HSTR / 4; ++q) {
          const float4 u = rp[q];
          row[4 * q] = u.x; row[4 * q + 1] = u.y; row[4 * q + 2] = u.z; row[4 * q + 3] = u.w;
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          float4 a = acc[dy * 3 + dx][t];
#pragma unroll
          for (int i = 0; i < TW; ++i) {
            const float xv = row[i + dx];
            a.x = fmaf(wv[i].x, xv, a.x); a.y = fmaf(wv[i].y, xv, a.y); a.z = fmaf(wv[i].z, xv, a.z); a.w = fmaf(wv[i].w, xv, a.w);
          }
          acc[dy * 3 + dx][t] = a;
          if (!WIDE_D && with_bias && dy == 1 && dx == 1 && l == 0) {      // sum of the tiny D over this row's pixels
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < TW; ++i) sum += row[i + 1];
            if (t == 0) bsum.x += sum; else if (t == 1) bsum.y += sum; else if (t == 2) bsum.z += sum; else bsum.w += sum;
          }
        }
      }
    }
    if (WIDE_D && with_bias) {
#pragma unroll
      for (int i = 0; i < TW; ++i) { bsum.x += wv[i].x; bsum.y += wv[i].y; bsum.z += wv[i].z; bsum.w += wv[i].w; }
    }
  }

  // ---- sum over the block's pixel-row groups and write the slab ---------------------------------------------
  const int T = 9, TS = T + (with_bias ? 1 : 0);
  float* slab = p.slabs + (long)blockIdx.x * p.Cd * TS * p.Cg;
  auto reduce_store = [&](float4 v, int tap, int t) {
    __syncthreads();
    *reinterpret_cast<float4*>(&sR[g][4 * l]) = v;
    __syncthreads();
    if (g == 0) {
      float4 s = *reinterpret_cast<const float4*>(&sR[0][4 * l]);
#pragma unroll
      for (int r = 1; r < ROWS; ++r) {
        const float4 u = *reinterpret_cast<const float4*>(&sR[r][4 * l]);
        s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
      }
      const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int wch = 4 * l + j;
        if (WIDE_D) slab[((long)wch * TS + tap) * p.Cg + t] = sv[j];              // a = wide, b = tiny
        else slab[((long)t * TS + tap) * p.Cg + wch] = sv[j];                     // a = tiny, b = wide
      }
    }
  };
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int t = 0; t < NT; ++t) reduce_store(acc[k][t], WIDE_D ? k : 8 - k, t);
  if (with_bias) {
    if (WIDE_D) {
      reduce_store(bsum, T, 0);            // column 0 of the extra tap carries the bias sums
    } else {
      // lane 0 of every group holds NT sums: add the groups
      __syncthreads();
      if (l == 0) *reinterpret_cast<float4*>(&sR[g][0]) = bsum;
      __syncthreads();
      if (tid == 0) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < ROWS; ++r) { s[0] += sR[r][0]; s[1] += sR[r][1]; s[2] += sR[r][2]; s[3] += sR[r][3]; }
        for (int t = 0; t < NT; ++t) slab[((long)t * TS + T) * p.Cg] = s[t];
      }
    }
  }
}

// ---- forward / backward-data: stride-1 3x3 / 5x5 "same", 128 input channels, ONE or TWO output channels ------
// (the 128 -> 2 exit of FlowEstimator, the 128 -> 1 exit of SeparableConv4d, and the backward-data of
// FlowEstimator's 10 -> 128 entry, of which only the two flow channels are ever read: ops.PackFlowEstimatorFn.)
// A thread group (32 lanes = the 128 channels, a float4 each) walks DOWN a strip of 8 output columns: every input
// row of the strip is loaded once (8 + KS - 1 pixels, whole 512-byte pixel rows per group; the next row's loads
// are in flight while this one is multiplied) and scattered into the KS output rows it belongs to -- a ring of
// KS x NO x 8 partial sums per lane.  When an output row has seen its KS input rows, the 32 lanes are added by a
// reduce-scatter of shuffles and the row is written.  No input pixel is read twice inside a strip.
template <int KS, int NO>
__global__ __launch_bounds__(256) void conv_tiny_out_kernel(ConvK p, const float* __restrict__ w, long so, long si, long st,
                                                          int flip, int co_off, int zero_fill, int strips_x, int segs_y,
                                                          int seg_rows, int nwork) {
  constexpr int LG = 32, NGRP = 256 / LG, SW = 8, XW = SW + KS - 1, PAD = KS / 2, T = KS * KS, NV = NO * SW;
  __shared__ __attribute__((aligned(16))) float sW[T * NO][4 * LG];
  const int tid = threadIdx.x, g = tid / LG, l = tid % LG;
  for (int idx = tid; idx < T * NO * 4 * LG; idx += 256) {
    const int k = idx / (4 * LG), ci = idx - k * (4 * LG);
    const int tap = k / NO, co = k - tap * NO;
    sW[k][ci] = w[co * so + ci * si + (flip ? T - 1 - tap : tap) * st];
  }
  __syncthreads();
  float vmax = 0.f;
  // one work item = (image, vertical segment, strip); a group takes items g + NGRP * (blockIdx.x + gridDim.x * k)
  for (int item = blockIdx.x * NGRP + g; item < nwork; item += gridDim.x * NGRP) {
    const int sx = item % strips_x, r_ = item / strips_x, sy = r_ % segs_y, n = r_ / segs_y;
    const int x0 = sx * SW, y0 = sy * seg_rows, y1 = min(p.Ho, y0 + seg_rows);       // output rows y0 .. y1-1
    const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn + 4 * l;
    float* yimg = p.y + (long)(n / p.yni) * p.ysg + (long)(n % p.yni) * p.ysn;
    float acc[KS][NV];                      // acc[dy]: the output row this input row feeds through kernel row dy
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < NV; ++j) acc[s][j] = 0.f;
    auto load_row = [&](float4 (&xr)[XW], int iy) {
#pragma unroll
      for (int c = 0; c < XW; ++c) {
        const int ix = x0 + c - PAD;
        float4 t = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          t = *reinterpret_cast<const float4*>(xin + (long)iy * p.xsh + (long)ix * p.xsw);
        xr[c] = t;
      }
    };
    const int iy_end = y1 + PAD;            // input rows y0 - PAD .. y1 - 1 + PAD
    // one input row: the next row's loads are issued first, then this row is scattered into its KS output rows
    auto row = [&](float4 (&cur)[XW], float4 (&nxt)[XW], int iy) {
      if (iy + 1 < iy_end) load_row(nxt, iy + 1);
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
        for (int dx = 0; dx < KS; ++dx)
#pragma unroll
          for (int co = 0; co < NO; ++co) {
            const float4 wv = *reinterpret_cast<const float4*>(&sW[(dy * KS + dx) * NO + co][4 * l]);
#pragma unroll
            for (int i = 0; i < SW; ++i) {
              const float4 x = cur[i + dx];
              acc[dy][co * SW + i] = fmaf(x.x, wv.x, fmaf(x.y, wv.y, fmaf(x.z, wv.z, fmaf(x.w, wv.w, acc[dy][co * SW + i]))));
            }
          }
        __builtin_amdgcn_sched_barrier(0);   // (keeps the weight reads of one kernel row at a time in registers)
      }
      // output row oy = iy - PAD has now seen all its kernel rows (it sits in acc[KS - 1])
      const int oy = iy - PAD;
      if (oy >= y0 && oy < y1) {
        float v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = acc[KS - 1][j];
        int cnt = NV;
#pragma unroll
        for (int m = LG / 2; m >= 1; m >>= 1) {
          if (cnt > 1) {
            const int half = cnt / 2;
            const bool up = (l & m) != 0;
#pragma unroll
            for (int j = 0; j < half; ++j) {
              const float send = up ? v[j] : v[j + half], keep = up ? v[j + half] : v[j];
              v[j] = keep + __shfl_xor(send, m, 64);
            }
            cnt = half;
          } else {
            v[0] += __shfl_xor(v[0], m, 64);
          }
        }
        // NV = 16: value index = l >> 1 (both lanes of a pair hold it); NV = 8: l >> 2
        const int idx = NV == 16 ? (l >> 1) : (l >> 2);
        const int co = idx / SW, i = idx - co * SW;
        float val = v[0];
        if (p.bias) val += p.bias[co];
        val = mt_act_fn(val, p.act) * p.alpha;
        float val1 = 0.f;
        if (NO == 2) val1 = __shfl_xor(val, 16, 64);        // the lane of (0, i) fetches output (1, i)
        const bool writer = NO == 2 ? (l < 16 && (l & 1) == 0) : ((l & 3) == 0);
        const int ox = x0 + i;
        if (writer && ox < p.Wo) {
          float* yp = yimg + (long)oy * p.ysh + (long)ox * p.ysw;
          if (zero_fill) {
            *reinterpret_cast<float4*>(yp) = float4{val, val1, 0.f, 0.f};
            for (int c = 4; c < p.Co; c += 4) *reinterpret_cast<float4*>(yp + c) = float4{0.f, 0.f, 0.f, 0.f};
          } else {
            yp[co_off] = val;
            if (NO == 2) yp[co_off + 1] = val1;
          }
          vmax = fmaxf(vmax, fmaxf(fabsf(val), fabsf(val1)));
        }
      }
      // the next input row sees every output row one kernel row later
#pragma unroll
      for (int s = KS - 1; s > 0; --s)
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[s][j] = acc[s - 1][j];
#pragma unroll
      for (int j = 0; j < NV; ++j) acc[0][j] = 0.f;
    };
    float4 xa[XW], xb[XW];
    load_row(xa, y0 - PAD);
    for (int iy = y0 - PAD; iy < iy_end; iy += 2) {
      row(xa, xb, iy);
      if (iy + 1 < iy_end) row(xb, xa, iy + 1);
    }
  }
  finish_amax(p, vmax, sW);
}

}  // namespace

// w[(co) * so + (ci) * si + (tap) * st] = weight of output channel co, input channel ci, tap kh * 3 + kw of
// the convolution this launch computes; flip = 1 reverses the taps (backward-data of a Conv2d run as a
// forward convolution: co / ci strides swapped by the caller).  d->wt / wt_layout / precision are not used.
extern "C" int mt_conv_tiny_in_run(const mt_conv* d, const float* w, int32_t cin, int64_t so, int64_t si, int64_t st,
                                   int32_t flip, void* stream) {
  MT_REQUIRE(d && w && cin >= 1 && cin <= 4, "mt_conv_tiny_in_run: 1..4 real input channels (got %d)", cin);
  mt_conv e = *d;
  e.wt = d->x.p;                           // (fill_convk checks a packed-weights pointer this path does not have)
  ConvK k;
  const int rc = fill_convk(&e, &k);
  if (rc) return rc;
  MT_REQUIRE(mt_switches().tiny, "mt_conv_tiny_in_run: disabled (MT355_TINY=0)");
  MT_REQUIRE(d->stride == 1 && d->KH == 3 && d->KW == 3 && d->pad == 1 && d->ups == 0 && !d->transposed,
             "mt_conv_tiny_in_run: stride-1 3x3 'same' layers only");
  MT_REQUIRE(d->x.C >= 4 && (d->y.C == 64 || d->y.C == 128), "mt_conv_tiny_in_run: x.C >= 4 and y.C in {64, 128} (got %d, %d)", d->x.C, d->y.C);
  MT_REQUIRE(!d->res.p && !d->accumulate && d->dact_c0 == 0, "mt_conv_tiny_in_run: no residual / accumulate / partial gating");
  MT_REQUIRE(!d->bias || (reinterpret_cast<uintptr_t>(d->bias) & 15) == 0, "mt_conv_tiny_in_run: bias must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k.yamax) k.ascr = mt_amax_scratch(s);
  const int rows = d->y.C == 128 ? 8 : 16;
  const int tiles_y = (k.Ho + rows - 1) / rows, tiles_x = (k.Wo + 15) / 16;
  const long nt = (long)k.N * tiles_y * tiles_x;
  MT_REQUIRE(nt < (1L << 31), "mt_conv_tiny_in_run: too many tiles");
  const int grid = (int)(nt < 4096 ? nt : 4096);
  if (d->y.C == 128)
    hipLaunchKernelGGL((conv_tiny_in_kernel<32>), dim3(grid), dim3(256), 0, s, k, w, cin, (long)so, (long)si, (long)st, flip, tiles_y, tiles_x, (int)nt);
  else
    hipLaunchKernelGGL((conv_tiny_in_kernel<16>), dim3(grid), dim3(256), 0, s, k, w, cin, (long)so, (long)si, (long)st, flip, tiles_y, tiles_x, (int)nt);
  return mt_check_launch("mt_conv_tiny_in_run");
}

// Weight (+ bias) gradient of a stride-1 3x3 "same" Conv2d with `tiny` <= 4 real channels on one side and 64 / 128
// on the other, in fp32 on the VALU.  d->splits = number of blocks = number of slabs; the slabs have the layout
// mt_wgrad_run writes, so mt_wgrad_reduce / the callers' code is unchanged.  precision / amax fields are ignored.
extern "C" int mt_wgrad_tiny_run(const mt_wgrad* d, int32_t tiny, void* stream) {
  MT_REQUIRE(d && mt_view_ok(&d->d) && mt_view_ok(&d->g) && d->slabs && d->splits >= 1, "mt_wgrad_tiny_run: invalid descriptor");
  MT_REQUIRE(mt_switches().tiny, "mt_wgrad_tiny_run: disabled (MT355_TINY=0)");
  MT_REQUIRE(d->stride == 1 && d->KH == 3 && d->KW == 3 && d->pad == 1 && d->ups == 0, "mt_wgrad_tiny_run: stride-1 3x3 'same' layers only");
  MT_REQUIRE(d->d.N == d->g.N && d->d.H == d->g.H && d->d.W == d->g.W, "mt_wgrad_tiny_run: D and G geometry differ");
  MT_REQUIRE(tiny >= 1 && tiny <= 4, "mt_wgrad_tiny_run: 1..4 channels on the narrow side (got %d)", tiny);
  const bool wide_d = d->d.C > d->g.C;
  const int wc = wide_d ? d->d.C : d->g.C, tc = wide_d ? d->g.C : d->d.C;
  MT_REQUIRE((wc == 64 || wc == 128) && tc >= 4 && tc >= tiny, "mt_wgrad_tiny_run: wide side 64 / 128 channels, narrow side padded to >= 4 (got %d, %d)", wc, tc);
  MT_REQUIRE(!d->skip_tiles, "mt_wgrad_tiny_run: no tile mask");
  WgK k;
  k.d = d->d.p; k.dsn = d->d.sn; k.dsh = d->d.sh; k.dsw = d->d.sw; k.Hd = d->d.H; k.Wd = d->d.W; k.Cd = d->d.C;
  k.g = d->g.p; k.gsn = d->g.sn; k.gsh = d->g.sh; k.gsw = d->g.sw; k.Hg = d->g.H; k.Wg = d->g.W; k.Cg = d->g.C;
  k.dni = d->d.n_in; k.dsg = d->d.sg; k.gni = d->g.n_in; k.gsg = d->g.sg;
  k.KH = 3; k.KW = 3; k.stride = 1; k.pad = 1; k.ups = 0;
  k.M = d->d.N * d->d.H * d->d.W; k.per_split = 0; k.atiles = k.btiles = 1;
  k.slabs = d->slabs; k.damax = k.gamax = nullptr; k.skip = 0; k.ntl = 0;
  const int rows = wc == 128 ? 8 : 16;
  const int tiles_y = (k.Hd + rows - 1) / rows, tiles_x = (k.Wd + 15) / 16;
  const long nt = (long)d->d.N * tiles_y * tiles_x;
  MT_REQUIRE(nt < (1L << 31) && d->splits <= nt, "mt_wgrad_tiny_run: splits (%d) must not exceed the tile count", d->splits);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(d->splits);
#define MT_WT(LG, NT, WD) hipLaunchKernelGGL((wgrad_tiny_kernel<LG, NT, WD>), grid, dim3(256), 0, s, k, d->with_bias, tiles_y, tiles_x, (int)nt)
#define MT_WT_NT(LG, WD) do { if (tiny == 1) MT_WT(LG, 1, WD); else if (tiny == 2) MT_WT(LG, 2, WD); else if (tiny == 3) MT_WT(LG, 3, WD); else MT_WT(LG, 4, WD); } while (0)
  if (wc == 128) { if (wide_d) MT_WT_NT(32, true); else MT_WT_NT(32, false); }
  else { if (wide_d) MT_WT_NT(16, true); else MT_WT_NT(16, false); }
#undef MT_WT_NT
#undef MT_WT
  return mt_check_launch("mt_wgrad_tiny_run");
}

// Stride-1 3x3 / 5x5 "same" convolution of a 128-channel tensor to `cout` = 1 or 2 output channels, fp32 on the
// VALU.  w[co * so + ci * si + tap * st]; flip as in mt_conv_tiny_in_run.  zero_fill = 1: every pixel of y is
// written whole (channels co_off = 0 .. cout-1 hold the result, the padding channels zeros); zero_fill = 0: only
// channels co_off .. co_off + cout - 1 of y are written (a window of a wider gradient tensor whose other channels
// nobody reads).  Supported of mt_conv: x / y views, bias, act, alpha, y_amax; no residual / gating / accumulate.
extern "C" int mt_conv_tiny_out_run(const mt_conv* d, const float* w, int32_t cout, int64_t so, int64_t si, int64_t st,
                                    int32_t flip, int32_t co_off, int32_t zero_fill, void* stream) {
  MT_REQUIRE(d && w && (cout == 1 || cout == 2), "mt_conv_tiny_out_run: 1 or 2 output channels (got %d)", cout);
  mt_conv e = *d;
  e.wt = d->x.p;
  ConvK k;
  const int rc = fill_convk(&e, &k);
  if (rc) return rc;
  MT_REQUIRE(mt_switches().tiny, "mt_conv_tiny_out_run: disabled (MT355_TINY=0)");
  MT_REQUIRE(d->stride == 1 && d->KH == d->KW && (d->KH == 3 || d->KH == 5) && d->pad == d->KH / 2 && d->ups == 0 && !d->transposed,
             "mt_conv_tiny_out_run: stride-1 3x3 / 5x5 'same' layers only");
  MT_REQUIRE(d->x.C == 128, "mt_conv_tiny_out_run: 128 input channels (got %d)", d->x.C);
  MT_REQUIRE(co_off >= 0 && co_off + cout <= d->y.C && (!zero_fill || co_off == 0), "mt_conv_tiny_out_run: channel window outside y");
  MT_REQUIRE(!d->res.p && !d->accumulate && !d->dsrc.p, "mt_conv_tiny_out_run: no residual / accumulate / gating");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k.yamax) k.ascr = mt_amax_scratch(s);
  // vertical segments: tall enough that the KS - 1 halo rows are a small share, short enough for >= 4096 work items
  const int strips_x = (k.Wo + 7) / 8;
  // (measured at 256^2: 8 .. 128 rows per segment all within 10 %; at 64^2 only the short segments fill the chip)
  int seg_rows = 64;
  while (seg_rows > 16 && (long)k.N * strips_x * ((k.Ho + seg_rows - 1) / seg_rows) < 4096) seg_rows >>= 1;
  if (seg_rows > k.Ho) seg_rows = k.Ho;
  const int segs_y = (k.Ho + seg_rows - 1) / seg_rows;
  const long nwork = (long)k.N * strips_x * segs_y;
  MT_REQUIRE(nwork < (1L << 31), "mt_conv_tiny_out_run: too many strips");
  const long nblk = (nwork + 7) / 8;
  const dim3 grid((unsigned)(nblk < 1024 ? nblk : 1024));
#define MT_TO(KS, NO) hipLaunchKernelGGL((conv_tiny_out_kernel<KS, NO>), grid, dim3(256), 0, s, k, w, (long)so, (long)si, (long)st, flip, co_off, zero_fill, strips_x, segs_y, seg_rows, (int)nwork)
  if (d->KH == 3) { if (cout == 1) MT_TO(3, 1); else MT_TO(3, 2); }
  else { if (cout == 1) MT_TO(5, 1); else MT_TO(5, 2); }
#undef MT_TO
  return mt_check_launch("mt_conv_tiny_out_run");
}
