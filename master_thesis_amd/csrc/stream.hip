// HBM-bound streaming kernels of the DFPN / CHN path for gfx950: layout
// conversion (+normalisation), pooling, warps, resizes, L1-type reductions,
// Adam.  Built with -ffp-contract=off so that the coordinate arithmetic of the
// warps and resizes keeps the reference's (ATen's) operation order and the
// integer sample indices come out bit-exact.
#include <cstdlib>
#include "common.h"
#include <limits.h>

namespace {

// ---------------------------------------------------------------- layout ---
__global__ void nchw_to_nhwc4_kernel(const float* src, long sn, long sc, long sh, long sw, int N, int C, int H, int W,
                                     const float* scale, const float* shift, float* dst, long dn, long dh, long dw,
                                     int DC, int c_off) {
  // one thread per (pixel, group of 4 channels of the destination window)
  const int G = (C + 3) / 4;
  const long total = (long)N * H * W * G;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int g = (int)(e % G);
    long pix = e / G;
    int x = (int)(pix % W);
    long r = pix / W;
    int y = (int)(r % H);
    int n = (int)(r / H);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int c = g * 4 + j;
      if (c < C) {
        float t = src[n * sn + c * sc + y * sh + x * sw];
        if (scale) t = t * scale[c] + shift[c];
        v[j] = t;
      }
    }
    float* d = dst + n * dn + y * dh + x * dw + c_off + g * 4;
    if ((c_off & 3) == 0) {
      *reinterpret_cast<f32x4*>(d) = v;
    } else {
      for (int j = 0; j < 4; ++j)
        if (g * 4 + j < C) d[j] = v[j];
    }
  }
}

__global__ void nhwc4_to_nchw_kernel(const float* src, long sn, long sh, long sw, int N, int H, int W, int c_off,
                                     float* dst, long dn, long dc, long dh, long dw, int C, const float* scale,
                                     const float* shift, int accumulate) {
  const long total = (long)N * C * H * W;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int x = (int)(e % W);
    long r = e / W;
    int y = (int)(r % H);
    r /= H;
    int c = (int)(r % C);
    int n = (int)(r / C);
    float t = src[n * sn + y * sh + x * sw + c_off + c];
    if (scale) t = t * scale[c] + shift[c];
    float* d = dst + n * dn + c * dc + y * dh + x * dw;
    *d = accumulate ? *d + t : t;
  }
}

// CHN output stage (model_chn.py:81-85): de-normalise + clamp + compose, NHWC -> (B,3,F,H,W)
__global__ void compose_clamp_fwd_kernel(const float* o, long osn, long osh, long osw, const float* xt, const float* v,
                                         const float* mean, const float* stdv, int B, int F, int H, int W,
                                         float* yhat, float* ycomp) {
  const long total = (long)B * F * H * W;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int x = (int)(e % W);
    long r = e / W;
    const int y = (int)(r % H); r /= H;
    const int f = (int)(r % F);
    const int b = (int)(r / F);
    const float* op = o + ((long)b * F + f) * osn + (long)y * osh + (long)x * osw;
    const long hw = (long)H * W, pix = (long)y * W + x;
    const float vis = v[b * hw + pix];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float yh = fminf(fmaxf(op[c] * stdv[c] + mean[c], 0.f), 1.f);
      const long oi = (((long)b * 3 + c) * F + f) * hw + pix;
      yhat[oi] = yh;
      ycomp[oi] = vis * xt[((long)b * 3 + c) * hw + pix] + (1.f - vis) * yh;
    }
  }
}

__global__ void compose_clamp_bwd_kernel(const float* o, long osn, long osh, long osw, const float* v, const float* mean,
                                         const float* stdv, int B, int F, int H, int W, const float* dyh,
                                         const float* dyc, float* d_o, long dsn, long dsh, long dsw, int Cd) {
  const long total = (long)B * F * H * W;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int x = (int)(e % W);
    long r = e / W;
    const int y = (int)(r % H); r /= H;
    const int f = (int)(r % F);
    const int b = (int)(r / F);
    const float* op = o + ((long)b * F + f) * osn + (long)y * osh + (long)x * osw;
    float* dp = d_o + ((long)b * F + f) * dsn + (long)y * dsh + (long)x * dsw;
    const long hw = (long)H * W, pix = (long)y * W + x;
    const float keep = 1.f - v[b * hw + pix];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float pre = op[c] * stdv[c] + mean[c];
      const long oi = (((long)b * 3 + c) * F + f) * hw + pix;
      float g = (dyh ? dyh[oi] : 0.f) + (dyc ? keep * dyc[oi] : 0.f);
      dp[c] = (pre >= 0.f && pre <= 1.f) ? g * stdv[c] : 0.f;      // clamp passes the gradient on [0, 1] (ATen)
    }
    for (int c = 3; c < Cd; ++c) dp[c] = 0.f;
  }
}

__global__ void maxpool2x2_kernel(const float* x, long xn, long xh, long xw, float* y, long yn, long yh, long yw, int N,
                                  int Ho, int Wo, int C4) {
  const long total = (long)N * Ho * Wo * C4;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int c = (int)(e % C4) * 4;
    long r = e / C4;
    int ox = (int)(r % Wo);
    r /= Wo;
    int oy = (int)(r % Ho);
    int n = (int)(r / Ho);
    const float* p = x + n * xn + (2 * oy) * xh + (2 * ox) * xw + c;
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + xw);
    f32x4 c0 = *reinterpret_cast<const f32x4*>(p + xh), d = *reinterpret_cast<const f32x4*>(p + xh + xw);
    f32x4 m;
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(c0[j], d[j]));
    *reinterpret_cast<f32x4*>(y + n * yn + oy * yh + ox * yw + c) = m;
  }
}

__global__ void sumpool_kernel(const float* x, long xn, long xh, long xw, float* y, long yn, long yh, long yw, int N,
                               int Ho, int Wo, int C4, int f) {
  const long total = (long)N * Ho * Wo * C4;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int c = (int)(e % C4) * 4;
    long r = e / C4;
    int ox = (int)(r % Wo);
    r /= Wo;
    int oy = (int)(r % Ho);
    int n = (int)(r / Ho);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx)
        s += *reinterpret_cast<const f32x4*>(x + n * xn + (long)(f * oy + dy) * xh + (long)(f * ox + dx) * xw + c);
    *reinterpret_cast<f32x4*>(y + n * yn + oy * yh + ox * yw + c) = s;
  }
}

struct V3 { const float* p; long sn, sh, sw; };

// Iteration of the element-wise kernels over an NHWC view: a thread keeps ONE channel quad (c) and walks pixels
// with a block-wide stride -- the quotient / remainder chain per element (64-bit, four divisions) made these
// kernels ALU-bound at ~1.1 TB/s.  Pixel -> (n, y, x) is only decoded for views whose pixels are not evenly
// spaced (32-bit arithmetic).
struct PxWalk {
  int c, ppb, pl;           // channel offset in floats, pixels per block, this thread's pixel slot (-1: idle)
  __device__ PxWalk(int C4) {
    ppb = blockDim.x / C4;
    const int q = threadIdx.x / C4;
    c = ((int)threadIdx.x - q * C4) * 4;
    pl = q < ppb ? q : -1;
  }
};
__device__ __forceinline__ long px_off(long sn, long sh, long sw, int H, int W, unsigned p) {
  if (sh == (long)W * sw && sn == (long)H * sh) return (long)p * sw;        // evenly spaced pixels
  const unsigned r = p / (unsigned)W, xx = p - r * (unsigned)W, n = r / (unsigned)H, yy = r - n * (unsigned)H;
  return (long)n * sn + (long)yy * sh + (long)xx * sw;
}
// wave maximum raised into a device scalar (values >= 0 order like their bit patterns; the plain
// read filters out almost every wave once the scalar is near its final value)
__device__ __forceinline__ void wave_raise_amax(float v, float* out) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0 && v > __builtin_nontemporal_load(out))
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(v));
}
// block maximum raised into a device scalar: two levels through the stream's scratch (common.h: raise_amax_block --
// same-address atomics cost 11 ns each, a 2048-block launch spent 22 us in them) when there is one.  All threads call it.
__device__ __forceinline__ void block_raise_amax(float v, float* out, AmaxScratch* scr) {
  __shared__ float red[16];
  if (scr) raise_amax_block(out, scr, v, red, blockIdx.x, gridDim.x);
  else wave_raise_amax(v, out);
}

__global__ void axpby_kernel(V3 x, float a, V3 z, float b, float* y, long yn, long yh, long yw, int N, int H, int W,
                             int C4, float* yamax, AmaxScratch* scr) {
  const PxWalk w(C4);
  const unsigned P = (unsigned)N * H * W;
  float m = 0.f;
  if (w.pl >= 0)
    for (unsigned p = blockIdx.x * w.ppb + w.pl; p < P; p += gridDim.x * w.ppb) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x.p + px_off(x.sn, x.sh, x.sw, H, W, p) + w.c) * a;
      if (z.p) v += *reinterpret_cast<const f32x4*>(z.p + px_off(z.sn, z.sh, z.sw, H, W, p) + w.c) * b;
      *reinterpret_cast<f32x4*>(y + px_off(yn, yh, yw, H, W, p) + w.c) = v;
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
  if (yamax) block_raise_amax(m, yamax, scr);
}

__global__ void dact_kernel(V3 x, V3 s, int act, float* y, long yn, long yh, long yw, int N, int H, int W, int C4) {
  const PxWalk w(C4);
  const unsigned P = (unsigned)N * H * W;
  if (w.pl < 0) return;
  for (unsigned p = blockIdx.x * w.ppb + w.pl; p < P; p += gridDim.x * w.ppb) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x.p + px_off(x.sn, x.sh, x.sw, H, W, p) + w.c);
    const f32x4 t = *reinterpret_cast<const f32x4*>(s.p + px_off(s.sn, s.sh, s.sw, H, W, p) + w.c);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= mt_dact_fn(t[j], act);
    *reinterpret_cast<f32x4*>(y + px_off(yn, yh, yw, H, W, p) + w.c) = v;
  }
}

// ------------------------------------------------------------------ warp ---
// grid_sample coordinate un-normalisation (ATen grid_sampler_unnormalize)
__device__ __forceinline__ float unnorm(float g, int size, int align_corners) {
  if (align_corners) return ((g + 1.f) / 2.f) * (float)(size - 1);
  return ((g + 1.f) * (float)size - 1.f) / 2.f;
}

struct WarpL {   // 5-D (B, C, F, H, W) addressing without materialising the F-fold
  int B, F, C, H, W;
  long xb, xc, xf;     // x strides (elements); H, W contiguous
  long vb, vf;         // v (B,1,F,H,W)
  long ob, oc, of;     // xo
  long wb, wf;         // vo
};

// float -> int of a corner / nearest coordinate: defined for every input (huge flows, NaN: the
// result is far outside any image and fails the bounds tests; NaN maps to the lower clamp), and
// leaves room for the "+ 1" of the second corner.  Values inside +-2^30 convert unchanged.
__device__ __forceinline__ int coord_int(float f) { return (int)fminf(fmaxf(f, -1073741824.f), 1073741824.f); }

__global__ void warp_fwd_kernel(const float* x, const float* v, const float* grid, WarpL L, int ac, float* xo, float* vo,
                                int* idx) {
  const int H = L.H, W = L.W;
  const long HW = (long)H * W;
  const long total = (long)L.B * L.F * HW;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e / HW);
    const long pix = e - n * HW;
    const int b = n / L.F, f = n - b * L.F;
    const f32x2 g = *reinterpret_cast<const f32x2*>(grid + e * 2);
    const float ix = unnorm(g[0], W, ac), iy = unnorm(g[1], H, ac);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = coord_int(fx), y0 = coord_int(fy), x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy;
    const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
    const bool bx0 = x0 >= 0 && x0 < W, bx1 = x1 >= 0 && x1 < W, by0 = y0 >= 0 && y0 < H, by1 = y1 >= 0 && y1 < H;
    const float* xp = x + b * L.xb + f * L.xf;
    float* op = xo + b * L.ob + f * L.of + pix;
    for (int c = 0; c < L.C; ++c) {
      const float* pl = xp + c * L.xc;
      float o = 0.f;
      if (by0 && bx0) o += pl[(long)y0 * W + x0] * nw;
      if (by0 && bx1) o += pl[(long)y0 * W + x1] * ne;
      if (by1 && bx0) o += pl[(long)y1 * W + x0] * sw;
      if (by1 && bx1) o += pl[(long)y1 * W + x1] * se;
      op[c * L.oc] = o;
    }
    // nearest: nearbyint (ties to even), zeros outside
    const float rx = nearbyintf(ix), ry = nearbyintf(iy);
    const bool nin = rx >= 0.f && rx < (float)W && ry >= 0.f && ry < (float)H;
    const int nx = nin ? coord_int(rx) : -1, ny = nin ? coord_int(ry) : -1;
    if (vo) vo[b * L.wb + f * L.wf + pix] = nin ? v[b * L.vb + f * L.vf + (long)ny * W + nx] : 0.f;
    if (idx) {
      const bool cin = fabsf(fx) < 1e9f && fabsf(fy) < 1e9f;
      idx[e * 4 + 0] = cin ? y0 : INT_MIN;
      idx[e * 4 + 1] = cin ? x0 : INT_MIN;
      idx[e * 4 + 2] = ny;
      idx[e * 4 + 3] = nx;
    }
  }
}

// Four output pixels per thread, INTERLEAVED over the wave: a wave owns 256 consecutive pixels of one image and lane l
// takes pixels l, l + 64, l + 128, l + 192 of them.  Every memory instruction of a wave then touches consecutive
// addresses (the flow: 8 bytes per lane; a gather: the source pixels under 64 neighbouring output pixels; a store: 4
// bytes per lane), which is what the address coalescer wants, and the gathers of a thread's four pixels -- 24 eight-byte
// loads: the two corners of a bilinear row that are both inside the image come as ONE load -- are all in flight together.
// (History, 1024 pairs at 256 x 256: one pixel per thread, 18 instructions per 64 pixels: 1.98 TB/s; four CONSECUTIVE
// pixels per thread with 16-byte flow loads / stores: 3.0 TB/s, bound by the texture addresser -- a gather whose lanes
// sit 16 bytes apart costs it a cycle per lane (PMC: 57 % of the wave cycles waiting at 71 % L2 hits and no over-fetch).)
// The coordinate arithmetic is that of warp_fwd_kernel expression for expression (stream.hip is built with
// -ffp-contract=off): indices stay bit-exact.
template <int CC>
__global__ __launch_bounds__(256) void warp_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                        const float* __restrict__ grid, WarpL L, int ac,
                                                        float* __restrict__ xo, float* __restrict__ vo, int* __restrict__ idx) {
  const int H = L.H, W = L.W;
  const long HW = (long)H * W, HW4 = HW >> 2;
  const long total4 = (long)L.B * L.F * HW4;
  const int C = CC > 0 ? CC : L.C;
  const int lane = threadIdx.x & 63;
  // Blocks are dealt round-robin over the 8 XCDs (each with its own L2).  A row of output pixels gathers along a
  // slanted line of the source (a 0.2 rad rotation crosses 50 source rows in 256 pixels), so a cache line it touches is
  // shared with the output rows above and below: every XCD therefore takes ONE contiguous eighth of the pixels and its
  // blocks walk it side by side -- the rows that share lines are in flight on the same L2 at the same time.
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;      // (the grid is a multiple of 8)
  const long per = (((total4 + 7) >> 3) + 63) & ~63L, q_end = min(total4, (xcd + 1) * per);     // (whole waves)
  for (long q = xcd * per + lb * (long)blockDim.x + threadIdx.x; q < q_end; q += (long)nlb * blockDim.x) {
    const long qw = q - lane;                       // the wave's first group of four: 256 pixels from pixel 4 qw on
    const int n = (int)(qw / HW4);                  // (HW % 256 == 0: a wave never straddles two images)
    const long pix0 = ((qw - n * HW4) << 2) + lane;
    const long e0 = n * HW + pix0;
    const int b = n / L.F, f = n - b * L.F;
    f32x2 g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(grid + (e0 + 64 * k) * 2));
    int x0[4], y0[4], nx[4], ny[4];
    float nw[4], ne[4], sw[4], se[4];
    unsigned ok[4];                        // bit 0: x0, 1: x1, 2: y0, 3: y1 inside; bit 4: nearest inside; bit 5: corners finite
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float ix = unnorm(g[k][0], W, ac), iy = unnorm(g[k][1], H, ac);
      const float fx = floorf(ix), fy = floorf(iy);
      x0[k] = coord_int(fx); y0[k] = coord_int(fy);
      const int x1 = x0[k] + 1, y1 = y0[k] + 1;
      const float wx1 = ix - fx, wy1 = iy - fy, wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy;
      nw[k] = wx0 * wy0; ne[k] = wx1 * wy0; sw[k] = wx0 * wy1; se[k] = wx1 * wy1;
      ok[k] = (x0[k] >= 0 && x0[k] < W ? 1u : 0u) | (x1 >= 0 && x1 < W ? 2u : 0u) | (y0[k] >= 0 && y0[k] < H ? 4u : 0u) |
              (y1 >= 0 && y1 < H ? 8u : 0u);
      const float rx = nearbyintf(ix), ry = nearbyintf(iy);
      const bool nin = rx >= 0.f && rx < (float)W && ry >= 0.f && ry < (float)H;
      nx[k] = nin ? coord_int(rx) : -1; ny[k] = nin ? coord_int(ry) : -1;
      ok[k] |= nin ? 16u : 0u;
      ok[k] |= (fabsf(fx) < 1e9f && fabsf(fy) < 1e9f) ? 32u : 0u;
    }
    const float* xp = x + b * L.xb + f * L.xf;
    float* op = xo + b * L.ob + f * L.of + pix0;
    // One bilinear row = ONE unconditional 8-byte gather at a clamped address (columns xc, xc + 1 of a clamped row):
    // the corner that lies inside the image is one of the two values, a corner outside adds nothing to the sum.
    // No divergent control flow: the 24 gathers of a thread's four pixels are all in flight together.
    int xc[4], rt[4], rb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xc[k] = min(max(x0[k], 0), W - 2);
      rt[k] = min(max(y0[k], 0), H - 1) * W + xc[k];
      rb[k] = min(max(y0[k] + 1, 0), H - 1) * W + xc[k];
    }
#pragma unroll CC > 0 ? CC : 1
    for (int cc = 0; cc < C; ++cc) {
      const float* pl = xp + cc * L.xc;
      f32x2 top[4], bot[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_memcpy(&top[k], pl + rt[k], 8);
        __builtin_memcpy(&bot[k], pl + rb[k], 8);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool lo = x0[k] == xc[k];          // x0 is the first value of the pair (else x0 = W - 1: the second; x0 = -1: x1 is the first)
        const float tl = lo ? top[k][0] : top[k][1], tr = lo ? top[k][1] : top[k][0];
        const float bl = lo ? bot[k][0] : bot[k][1], br = lo ? bot[k][1] : bot[k][0];
        // (the same sum, in the same order, as warp_fwd_kernel)
        float t = 0.f;
        if ((ok[k] & 5u) == 5u) t += tl * nw[k];
        if ((ok[k] & 6u) == 6u) t += tr * ne[k];
        if ((ok[k] & 9u) == 9u) t += bl * sw[k];
        if ((ok[k] & 10u) == 10u) t += br * se[k];
        __builtin_nontemporal_store(t, op + cc * L.oc + 64 * k);
      }
    }
    if (vo) {
      const float* vp = v + b * L.vb + f * L.vf;
      float* wp = vo + b * L.wb + f * L.wf + pix0;
      float t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = vp[(long)max(ny[k], 0) * W + max(nx[k], 0)];
#pragma unroll
      for (int k = 0; k < 4; ++k) __builtin_nontemporal_store((ok[k] & 16u) ? t[k] : 0.f, wp + 64 * k);
    }
    if (idx) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool cin = (ok[k] & 32u) != 0;
        idx[(e0 + 64 * k) * 4 + 0] = cin ? y0[k] : INT_MIN;
        idx[(e0 + 64 * k) * 4 + 1] = cin ? x0[k] : INT_MIN;
        idx[(e0 + 64 * k) * 4 + 2] = ny[k];
        idx[(e0 + 64 * k) * 4 + 3] = nx[k];
      }
    }
  }
}

__global__ void warp_bwd_grid_kernel(const float* x, const float* grid, const float* dxo, WarpL L, int ac,
                                     float* dgrid) {
  const int H = L.H, W = L.W;
  const long HW = (long)H * W;
  const long total = (long)L.B * L.F * HW;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e / HW);
    const long pix = e - n * HW;
    const int b = n / L.F, f = n - b * L.F;
    const f32x2 g = *reinterpret_cast<const f32x2*>(grid + e * 2);
    const float ix = unnorm(g[0], W, ac), iy = unnorm(g[1], H, ac);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = coord_int(fx), y0 = coord_int(fy), x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = (fx + 1.f) - ix, wy0 = (fy + 1.f) - iy;
    const bool bx0 = x0 >= 0 && x0 < W, bx1 = x1 >= 0 && x1 < W, by0 = y0 >= 0 && y0 < H, by1 = y1 >= 0 && y1 < H;
    float gx = 0.f, gy = 0.f;
    const float* xp = x + b * L.xb + f * L.xf;
    const float* gp = dxo + b * L.ob + f * L.of + pix;
    for (int c = 0; c < L.C; ++c) {
      const float* pl = xp + c * L.xc;
      const float go = gp[c * L.oc];
      const float vnw = (by0 && bx0) ? pl[(long)y0 * W + x0] : 0.f;
      const float vne = (by0 && bx1) ? pl[(long)y0 * W + x1] : 0.f;
      const float vsw = (by1 && bx0) ? pl[(long)y1 * W + x0] : 0.f;
      const float vse = (by1 && bx1) ? pl[(long)y1 * W + x1] : 0.f;
      gx -= vnw * wy0 * go; gy -= vnw * wx0 * go;
      gx += vne * wy0 * go; gy -= vne * wx1 * go;
      gx -= vsw * wy1 * go; gy += vsw * wx0 * go;
      gx += vse * wy1 * go; gy += vse * wx1 * go;
    }
    const float mx = ac ? (float)(W - 1) / 2.f : (float)W / 2.f;
    const float my = ac ? (float)(H - 1) / 2.f : (float)H / 2.f;
    f32x2 o = {mx * gx, my * gy};
    *reinterpret_cast<f32x2*>(dgrid + e * 2) = o;
  }
}

// ---------------------------------------------------------------- resize ---
__device__ __forceinline__ void bilinear_src(float scale, int dst, int in, int* i0, int* i1, float* l0, float* l1) {
  // ATen area_pixel_compute_source_index, align_corners = false
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  int a = (int)s;
  if (a > in - 1) a = in - 1;
  *i0 = a;
  *i1 = a + (a < in - 1 ? 1 : 0);
  *l1 = s - (float)a;
  *l0 = 1.f - *l1;
}

__global__ void resize_fwd_kernel(const float* src, long sp, long sy, long sx, int NP, int H, int W, float* dst, long dp,
                                  long dy, long dx, int Ho, int Wo, int CI, int mode) {
  const float scy = (float)H / (float)Ho, scx = (float)W / (float)Wo;
  const long total = (long)NP * Ho * Wo * CI;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int c = (int)(e % CI);
    long r = e / CI;
    int ox = (int)(r % Wo);
    r /= Wo;
    int oy = (int)(r % Ho);
    long pl = r / Ho;
    const float* s = src + pl * sp + c;
    float o;
    if (mode == 0) {
      // ATen nearest_neighbor_compute_source_index
      int iy = min((int)floorf((float)oy * scy), H - 1), ix = min((int)floorf((float)ox * scx), W - 1);
      o = s[iy * sy + ix * sx];
    } else {
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      bilinear_src(scy, oy, H, &y0, &y1, &ly0, &ly1);
      bilinear_src(scx, ox, W, &x0, &x1, &lx0, &lx1);
      o = ly0 * (lx0 * s[y0 * sy + x0 * sx] + lx1 * s[y0 * sy + x1 * sx]) +
          ly1 * (lx0 * s[y1 * sy + x0 * sx] + lx1 * s[y1 * sy + x1 * sx]);
    }
    dst[pl * dp + oy * dy + ox * dx + c] = o;
  }
}

// Backward of the bilinear mode in GATHER form (round 4): a thread owns ONE source element and
// adds, in a fixed order, what every destination pixel whose stencil touches it contributes -- no float atomics (the
// scatter form's atomicAdd order made two runs of the same step differ, and ran at 0.7 TB/s), no zero-fill of dsrc.
// The destination rows that can touch source row sy are those whose source coordinate falls in [sy - 1, sy + 1)
// (plus everything clamped onto the first / last row); the candidates of that range re-evaluate bilinear_src itself, so
// the weights are the forward's bit for bit.
__global__ void resize_bilinear_bwd_kernel(const float* dd, long dp, long dy, long dx, int NP, int Ho, int Wo, float* ds,
                                           long sp, long sy_, long sx_, int H, int W, int CI) {
  const float scy = (float)H / (float)Ho, scx = (float)W / (float)Wo;
  const long total = (long)NP * H * W * CI;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % CI);
    long r = e / CI;
    const int sx = (int)(r % W);
    r /= W;
    const int sy = (int)(r % H);
    const long pl = r / H;
    // candidate ranges (one destination pixel of slack on either side for the float arithmetic)
    int oy_lo = sy == 0 ? 0 : max(0, (int)floorf(((float)sy - 0.5f) / scy - 0.5f) - 1);
    int oy_hi = sy == H - 1 ? Ho - 1 : min(Ho - 1, (int)ceilf(((float)sy + 1.5f) / scy - 0.5f) + 1);
    int ox_lo = sx == 0 ? 0 : max(0, (int)floorf(((float)sx - 0.5f) / scx - 0.5f) - 1);
    int ox_hi = sx == W - 1 ? Wo - 1 : min(Wo - 1, (int)ceilf(((float)sx + 1.5f) / scx - 0.5f) + 1);
    const float* g = dd + pl * dp + c;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly0, ly1;
      bilinear_src(scy, oy, H, &y0, &y1, &ly0, &ly1);
      const float wy = (y0 == sy ? ly0 : 0.f) + (y1 == sy ? ly1 : 0.f);
      if (y0 != sy && y1 != sy) continue;
      float row = 0.f;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx0, lx1;
        bilinear_src(scx, ox, W, &x0, &x1, &lx0, &lx1);
        if (x0 != sx && x1 != sx) continue;
        const float wx = (x0 == sx ? lx0 : 0.f) + (x1 == sx ? lx1 : 0.f);
        row += wx * g[oy * dy + ox * dx];
      }
      acc += wy * row;
    }
    ds[pl * sp + sy * sy_ + sx * sx_ + c] = acc;
  }
}

struct PackSrcs { mt_pack_src s[6]; int n; };
__global__ void pack_pairs_kernel(PackSrcs P, int B, int R, int H, int W, float* dst, long dn, long dh, long dw, int DC) {
  // one thread per (pair, pixel, destination float4)
  const int G = DC / 4;
  const long total = (long)B * R * H * W * G;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int g = (int)(e % G);
    long r0 = e / G;
    int x = (int)(r0 % W);
    r0 /= W;
    int y = (int)(r0 % H);
    int n = (int)(r0 / H);
    int b = n / R, r = n - b * R;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = g * 4 + j;
      for (int i = 0; i < P.n; ++i) {
        const mt_pack_src& q = P.s[i];
        if (c >= q.c_off && c < q.c_off + q.C) {
          const int cc = c - q.c_off;
          float t = q.p[b * q.sb + r * q.sr + cc * q.sc + y * q.sy + x * q.sx];
          if (q.scale) t = t * q.scale[cc] + q.shift[cc];
          v[j] = t;
        }
      }
    }
    *reinterpret_cast<f32x4*>(dst + n * dn + y * dh + x * dw + g * 4) = v;
  }
}

// ------------------------------------------------------------ reductions ---
__device__ __forceinline__ void block_add(float v, float* target) {
  __shared__ float part[8];
  v = mt_wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) part[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += part[i];
    atomicAdd(target, s);
  }
  __syncthreads();
}

__global__ void l1_fwd_kernel(const float* a, const float* b, const float* mask, const float* w, int B, int C, int MC,
                              long S, int count_mask, float* acc, AmaxScratch* scr) {
  const long total = (long)B * C * S;
  float s = 0.f, ms = 0.f;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long sidx = e % S;
    long r = e / S;
    int c = (int)(r % C);
    int bi = (int)(r / C);
    float m = mask ? mask[((long)bi * MC + (MC == 1 ? 0 : c)) * S + sidx] : 1.f;
    float wt = w ? w[bi] : 1.f;
    s += wt * fabsf(a[e] * m - b[e] * m);
    if (count_mask && (MC != 1 || c == 0)) ms += wt * m;
  }
  if (scr) {
    __shared__ float red[32];
    block_sum2(s, ms, acc, count_mask ? acc + 1 : nullptr, scr, red, blockIdx.x, gridDim.x);
    return;
  }
  block_add(s, acc);
  if (count_mask) block_add(ms, acc + 1);
}

__global__ void l1_bwd_kernel(const float* a, const float* b, const float* mask, const float* w, int B, int C, int MC,
                              long S, const float* scale, float* da, float* db, int accumulate) {
  const long total = (long)B * C * S;
  const float sc = scale[0];
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long sidx = e % S;
    long r = e / S;
    int c = (int)(r % C);
    int bi = (int)(r / C);
    float m = mask ? mask[((long)bi * MC + (MC == 1 ? 0 : c)) * S + sidx] : 1.f;
    float wt = w ? w[bi] : 1.f;
    float d = a[e] * m - b[e] * m;
    float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    float gval = sc * wt * sg * m;
    if (da) da[e] = accumulate ? da[e] + gval : gval;
    if (db) db[e] = accumulate ? db[e] - gval : -gval;
  }
}

__device__ __forceinline__ float pix(const float* p, int y, int x, int H, int W) {
  return (y >= 0 && y < H && x >= 0 && x < W) ? p[(long)y * W + x] : 0.f;
}
// Sobel responses of d = a - b at (y, x); cross-correlation with
// H: [[1,0,-1],[2,0,-2],[1,0,-1]], V: [[1,2,1],[0,0,0],[-1,-2,-1]], zero padding
__device__ __forceinline__ void sobel_at(const float* a, const float* b, int y, int x, int H, int W, float* gh, float* gv) {
  float d[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) d[i][j] = pix(a, y + i - 1, x + j - 1, H, W) - pix(b, y + i - 1, x + j - 1, H, W);
  *gh = (d[0][0] - d[0][2]) + 2.f * (d[1][0] - d[1][2]) + (d[2][0] - d[2][2]);
  *gv = (d[0][0] + 2.f * d[0][1] + d[0][2]) - (d[2][0] + 2.f * d[2][1] + d[2][2]);
}

__global__ void sobel_l1_fwd_kernel(const float* a, const float* b, int NP, int H, int W, float* acc, AmaxScratch* scr) {
  const long HW = (long)H * W, total = (long)NP * HW;
  float s = 0.f;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long pl = e / HW;
    int r = (int)(e - pl * HW), y = r / W, x = r - y * W;
    float gh, gv;
    sobel_at(a + pl * HW, b + pl * HW, y, x, H, W, &gh, &gv);
    s += fabsf(gh) + fabsf(gv);
  }
  if (scr) {
    __shared__ float red[32];
    block_sum2(s, 0.f, acc, nullptr, scr, red, blockIdx.x, gridDim.x);
    return;
  }
  block_add(s, acc);
}

__global__ void sobel_l1_bwd_kernel(const float* a, const float* b, int NP, int H, int W, const float* scale, float* da) {
  const float KHm[3][3] = {{1.f, 0.f, -1.f}, {2.f, 0.f, -2.f}, {1.f, 0.f, -1.f}};
  const float KVm[3][3] = {{1.f, 2.f, 1.f}, {0.f, 0.f, 0.f}, {-1.f, -2.f, -1.f}};
  const long HW = (long)H * W, total = (long)NP * HW;
  const float sc = scale[0];
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long pl = e / HW;
    int r = (int)(e - pl * HW), y = r / W, x = r - y * W;
    float g = 0.f;
    // out[q] depends on in[p] with kernel index (i, j) = p - q + 1
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        int qy = y - (i - 1), qx = x - (j - 1);
        if (qy < 0 || qy >= H || qx < 0 || qx >= W) continue;
        float gh, gv;
        sobel_at(a + pl * HW, b + pl * HW, qy, qx, H, W, &gh, &gv);
        float sh = gh > 0.f ? 1.f : (gh < 0.f ? -1.f : 0.f), sv = gv > 0.f ? 1.f : (gv < 0.f ? -1.f : 0.f);
        g += KHm[i][j] * sh + KVm[i][j] * sv;
      }
    da[e] = sc * g;
  }
}

__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long n, float b1, float b2, float step,
                            float ibc2, float eps, float gs) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gs;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step * (mi / (sqrtf(vi) * ibc2 + eps));
  }
}

// Adam whose step counter and learning rate live on the device, so that the
// update can be replayed from a captured hipGraph: state = {step, lr}
__global__ void adam_tick_kernel(float* state) { state[0] += 1.f; }

__global__ void adam_dev_kernel(float* p, const float* g, float* m, float* v, long n, float b1, float b2, float eps,
                                float gs, const float* state) {
  const float t = state[0], lr = state[1];
  const float step = lr / (1.f - powf(b1, t)), ibc2 = 1.f / sqrtf(1.f - powf(b2, t));
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gs;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step * (mi / (sqrtf(vi) * ibc2 + eps));
  }
}

// max |x| into a device scalar (non-negative floats order like their bit patterns)
__device__ __forceinline__ void block_amax(float v, float* out, AmaxScratch* scr) {
  __shared__ float part[16];
  if (scr) {
    raise_amax_block(out, scr, v, part, blockIdx.x, gridDim.x);
    return;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, part[i]);
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));
  }
}

// One wave behind a launch whose waves only raised the 64 slots (conv_common.h: finish_amax, amax_deferred): folds them
// into the scalar and leaves them zero.  For launches of many rounds of blocks, where the in-kernel election would cost
// every block two atomic round trips (2.3 us of a 42 us block of igemm_wino43_kernel).
__global__ void amax_fold_kernel(AmaxScratch* scr, float* dst) {
  float m = __uint_as_float(atomicExch(&scr->max[threadIdx.x * 64], 0u));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (threadIdx.x == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(dst), __float_as_uint(m));
}
__global__ void absmax_view_kernel(const float* x, long sn, long sh, long sw, int N, int H, int W, int C4, float* out,
                                   AmaxScratch* scr) {
  const PxWalk w(C4);
  const unsigned P = (unsigned)N * H * W;
  float m = 0.f;
  if (w.pl >= 0)
    for (unsigned p = blockIdx.x * w.ppb + w.pl; p < P; p += gridDim.x * w.ppb) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + px_off(sn, sh, sw, H, W, p) + w.c);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
  block_amax(m, out, scr);
}

__global__ void absmax_flat_kernel(const float* x, long n, float* out, AmaxScratch* scr) {
  float m = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(x[i]));
  block_amax(m, out, scr);
}

// 3D softmax of the correlation volume (model_dfpn.py:567-583): row = (clip b, target position pt),
// its R * P values live in R slabs, P contiguous floats each.  One wave per row: every lane keeps
// its values in registers between the two reductions when the row is short enough.
__global__ __launch_bounds__(256) void softmax3d_kernel(const float* x, int R, int P, float* out, long rows) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long b = row / P;
  const int pt = (int)(row - b * P);
  const long PP = (long)P * P;
  const float* src = x + (b * R) * PP + (long)pt * P;
  float* dst = out + (b * R) * PP + (long)pt * P;
  float m = -INFINITY;
  for (int r = 0; r < R; ++r)
    for (int c = lane * 4; c < P; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * PP + c);
      m = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), m);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float sum = 0.f;
  for (int r = 0; r < R; ++r)
    for (int c = lane * 4; c < P; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + r * PP + c);
      sum += (expf(v[0] - m) + expf(v[1] - m)) + (expf(v[2] - m) + expf(v[3] - m));
    }
  sum = mt_wave_sum(sum);
  const float inv = 1.f / sum;
  for (int r = 0; r < R; ++r)
    for (int c = lane * 4; c < P; c += 256) {
      f32x4 v = *reinterpret_cast<const f32x4*>(src + r * PP + c);
      v[0] = expf(v[0] - m) * inv; v[1] = expf(v[1] - m) * inv; v[2] = expf(v[2] - m) * inv; v[3] = expf(v[3] - m) * inv;
      *reinterpret_cast<f32x4*>(dst + r * PP + c) = v;
    }
}

// row-wise masked L2 normalisation of NHWC4 features (correlation prologue):
// out[p][c] = x[p][c]*v[p] / (||x[p]*v[p]||_2 + 1e-9); one wave per pixel row
__global__ void rownorm_kernel(const float* x, long xn, long xh, long xw, int H, int W, int C, const float* v,
                               float* out, long rows) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int HW = H * W;
  const long n = row / HW;
  const int rem = (int)(row - n * HW), y = rem / W, xx = rem - y * W;
  const float* src = x + n * xn + y * xh + xx * xw;
  const float vis = v ? v[row] : 1.f;
  float ss = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    f32x4 t = *reinterpret_cast<const f32x4*>(src + c) * vis;
    ss += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
  }
  ss = mt_wave_sum(ss);
  const float inv = 1.f / (sqrtf(ss) + 1e-9f);
  for (int c = lane * 4; c < C; c += 256) {
    f32x4 t = *reinterpret_cast<const f32x4*>(src + c) * vis;
    *reinterpret_cast<f32x4*>(out + row * C + c) = t * inv;
  }
}

V3 v3(const mt_view* v) { V3 r; r.p = v ? v->p : nullptr; r.sn = v ? v->sn : 0; r.sh = v ? v->sh : 0; r.sw = v ? v->sw : 0; return r; }
bool same_geom(const mt_view* a, const mt_view* b) { return a->N == b->N && a->H == b->H && a->W == b->W; }

}  // namespace

void mt_amax_fold_launch(AmaxScratch* scr, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(amax_fold_kernel, dim3(1), dim3(64), 0, s, scr, dst);
}


#define S_(x) static_cast<hipStream_t>(x)

extern "C" int mt_softmax3d(const float* x, int32_t B, int32_t R, int32_t P, float* out, void* stream) {
  MT_REQUIRE(x && out && B > 0 && R > 0 && P > 0 && (P & 3) == 0, "mt_softmax3d: bad arguments");
  const long rows = (long)B * P;
  hipLaunchKernelGGL(softmax3d_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_(stream), x, R, P, out, rows);
  return mt_check_launch("mt_softmax3d");
}

extern "C" int mt_rownorm(const mt_view* x, const float* v, float* out, void* stream) {
  MT_REQUIRE(mt_view_flat(x) && out, "mt_rownorm: bad arguments");
  const long rows = (long)x->N * x->H * x->W;
  hipLaunchKernelGGL(rownorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_(stream), x->p, (long)x->sn,
                     (long)x->sh, (long)x->sw, x->H, x->W, x->C, v, out, rows);
  return mt_check_launch("mt_rownorm");
}

extern "C" int mt_nchw_to_nhwc4(const float* src, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, int32_t N,
                                int32_t C, int32_t H, int32_t W, const float* scale, const float* shift,
                                const mt_view* dst, int32_t c_off, void* stream) {
  MT_REQUIRE(src && mt_view_flat(dst), "mt_nchw_to_nhwc4: bad views");
  MT_REQUIRE(dst->N == N && dst->H == H && dst->W == W && c_off >= 0 && c_off + C <= dst->C,
             "mt_nchw_to_nhwc4: geometry mismatch");
  MT_REQUIRE((scale == nullptr) == (shift == nullptr), "mt_nchw_to_nhwc4: scale/shift must come together");
  const long total = (long)N * H * W * ((C + 3) / 4);
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(mt_grid_1d(total, 256)), dim3(256), 0, S_(stream), src, (long)s_n,
                     (long)s_c, (long)s_h, (long)s_w, N, C, H, W, scale, shift, dst->p, (long)dst->sn, (long)dst->sh,
                     (long)dst->sw, dst->C, c_off);
  return mt_check_launch("mt_nchw_to_nhwc4");
}

extern "C" int mt_nhwc4_to_nchw(const mt_view* src, int32_t c_off, float* dst, int64_t d_n, int64_t d_c, int64_t d_h,
                                int64_t d_w, int32_t C, const float* scale, const float* shift, int32_t accumulate,
                                void* stream) {
  MT_REQUIRE(mt_view_flat(src) && dst && c_off >= 0 && c_off + C <= src->C, "mt_nhwc4_to_nchw: bad arguments");
  MT_REQUIRE((scale == nullptr) == (shift == nullptr), "mt_nhwc4_to_nchw: scale/shift must come together");
  const long total = (long)src->N * C * src->H * src->W;
  hipLaunchKernelGGL(nhwc4_to_nchw_kernel, dim3(mt_grid_1d(total, 256)), dim3(256), 0, S_(stream), src->p,
                     (long)src->sn, (long)src->sh, (long)src->sw, src->N, src->H, src->W, c_off, dst, (long)d_n,
                     (long)d_c, (long)d_h, (long)d_w, C, scale, shift, accumulate);
  return mt_check_launch("mt_nhwc4_to_nchw");
}

extern "C" int mt_compose_clamp_fwd(const mt_view* o, const float* x_t, const float* v, const float* mean3,
                                    const float* std3, int32_t B, int32_t F, float* y_hat, float* y_comp,
                                    void* stream) {
  MT_REQUIRE(mt_view_flat(o) && x_t && v && mean3 && std3 && y_hat && y_comp && B > 0 && F > 0 && o->N == B * F &&
                 o->C >= 3, "mt_compose_clamp_fwd: bad arguments");
  const long total = (long)B * F * o->H * o->W;
  hipLaunchKernelGGL(compose_clamp_fwd_kernel, dim3(mt_grid_1d(total, 256)), dim3(256), 0, S_(stream), o->p, (long)o->sn,
                     (long)o->sh, (long)o->sw, x_t, v, mean3, std3, B, F, o->H, o->W, y_hat, y_comp);
  return mt_check_launch("mt_compose_clamp_fwd");
}

extern "C" int mt_compose_clamp_bwd(const mt_view* o, const float* v, const float* mean3, const float* std3, int32_t B,
                                    int32_t F, const float* d_y_hat, const float* d_y_comp, const mt_view* d_o,
                                    void* stream) {
  MT_REQUIRE(mt_view_flat(o) && mt_view_flat(d_o) && v && mean3 && std3 && B > 0 && F > 0 && o->N == B * F &&
                 o->C >= 3 && d_o->N == o->N && d_o->H == o->H && d_o->W == o->W && d_o->C >= 3,
             "mt_compose_clamp_bwd: bad arguments");
  const long total = (long)B * F * o->H * o->W;
  hipLaunchKernelGGL(compose_clamp_bwd_kernel, dim3(mt_grid_1d(total, 256)), dim3(256), 0, S_(stream), o->p, (long)o->sn,
                     (long)o->sh, (long)o->sw, v, mean3, std3, B, F, o->H, o->W, d_y_hat, d_y_comp, d_o->p,
                     (long)d_o->sn, (long)d_o->sh, (long)d_o->sw, d_o->C);
  return mt_check_launch("mt_compose_clamp_bwd");
}

extern "C" int mt_maxpool2x2(const mt_view* x, const mt_view* y, void* stream) {
  MT_REQUIRE(mt_view_flat(x) && mt_view_flat(y), "mt_maxpool2x2: bad views");
  MT_REQUIRE(y->N == x->N && y->C == x->C && y->H == x->H / 2 && y->W == x->W / 2, "mt_maxpool2x2: geometry mismatch");
  const long total = (long)y->N * y->H * y->W * (y->C / 4);
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(mt_grid_1d(total, 256)), dim3(256), 0, S_(stream), x->p, (long)x->sn,
                     (long)x->sh, (long)x->sw, y->p, (long)y->sn, (long)y->sh, (long)y->sw, y->N, y->H, y->W,
                     y->C / 4);
  return mt_check_launch("mt_maxpool2x2");
}

extern "C" int mt_sumpool(const mt_view* dxup, const mt_view* dx, int32_t ups, void* stream) {
  MT_REQUIRE(mt_view_flat(dxup) && mt_view_flat(dx) && ups >= 1 && ups <= 2, "mt_sumpool: bad arguments");
  MT_REQUIRE(dx->N == dxup->N && dx->C == dxup->C && (dx->H << ups) == dxup->H && (dx->W << ups) == dxup->W,
             "mt_sumpool: geometry mismatch");
  const long total = (long)dx->N * dx->H * dx->W * (dx->C / 4);
  hipLaunchKernelGGL(sumpool_kernel, dim3(mt_grid_1d(total, 256)), dim3(256), 0, S_(stream), dxup->p, (long)dxup->sn,
                     (long)dxup->sh, (long)dxup->sw, dx->p, (long)dx->sn, (long)dx->sh, (long)dx->sw, dx->N, dx->H,
                     dx->W, dx->C / 4, 1 << ups);
  return mt_check_launch("mt_sumpool");
}

extern "C" int mt_axpby(const mt_view* x, float a, const mt_view* z, float b, const mt_view* y, float* y_amax,
                        void* stream) {
  MT_REQUIRE(mt_view_flat(x) && mt_view_flat(y) && same_geom(x, y) && x->C >= y->C, "mt_axpby: bad views");
  if (z && z->p) MT_REQUIRE(mt_view_flat(z) && same_geom(z, y) && z->C >= y->C, "mt_axpby: bad z view");
  MT_REQUIRE(y->C <= 1024 && (int64_t)y->N * y->H * y->W < (1ll << 31), "mt_axpby: more than 1024 channels or 2^31 pixels");
  const long total = (long)y->N * y->H * y->W * (y->C / 4);
  hipLaunchKernelGGL(axpby_kernel, dim3(mt_grid_1d(total, 256)), dim3(256), 0, S_(stream), v3(x), a,
                     v3(z && z->p ? z : nullptr), b, y->p, (long)y->sn, (long)y->sh, (long)y->sw, y->N, y->H, y->W,
                     y->C / 4, y_amax, y_amax ? mt_amax_scratch(S_(stream)) : nullptr);
  return mt_check_launch("mt_axpby");
}

extern "C" int mt_dact(const mt_view* x, const mt_view* s, int32_t act, const mt_view* y, void* stream) {
  MT_REQUIRE(mt_view_flat(x) && mt_view_flat(s) && mt_view_flat(y) && same_geom(x, y) && same_geom(s, y) && x->C >= y->C &&
                 s->C >= y->C, "mt_dact: bad views");
  MT_REQUIRE(y->C <= 1024 && (int64_t)y->N * y->H * y->W < (1ll << 31), "mt_dact: more than 1024 channels or 2^31 pixels");
  const long total = (long)y->N * y->H * y->W * (y->C / 4);
  hipLaunchKernelGGL(dact_kernel, dim3(mt_grid_1d(total, 256)), dim3(256), 0, S_(stream), v3(x), v3(s), act, y->p,
                     (long)y->sn, (long)y->sh, (long)y->sw, y->N, y->H, y->W, y->C / 4);
  return mt_check_launch("mt_dact");
}

static int warp_layout(const mt_warp_layout* l, WarpL* L) {
  MT_REQUIRE(l && l->B > 0 && l->F > 0 && l->C > 0 && l->H > 0 && l->W > 0, "mt_warp: bad layout");
  MT_REQUIRE((int64_t)l->B * l->F * l->H * l->W < (1ll << 31), "mt_warp: too many pixels");
  L->B = l->B; L->F = l->F; L->C = l->C; L->H = l->H; L->W = l->W;
  L->xb = l->xb; L->xc = l->xc; L->xf = l->xf; L->vb = l->vb; L->vf = l->vf;
  L->ob = l->ob; L->oc = l->oc; L->of = l->of; L->wb = l->wb; L->wf = l->wf;
  return MT_OK;
}

extern "C" int mt_warp_fwd(const float* x, const float* v, const float* grid, const mt_warp_layout* l,
                           int32_t align_corners, float* xo, float* vo, int32_t* idx, void* stream) {
  WarpL L;
  int rc = warp_layout(l, &L);
  if (rc) return rc;
  MT_REQUIRE(x && grid && xo, "mt_warp_fwd: null pointer");
  MT_REQUIRE((v == nullptr) == (vo == nullptr), "mt_warp_fwd: v and vo must come together");
  MT_REQUIRE((reinterpret_cast<uintptr_t>(grid) & 7) == 0, "mt_warp_fwd: grid must be 8-byte aligned");
  // four pixels per thread (interleaved over the wave) when an image is a whole number of 256-pixel wave loads
  const bool vec4 = L.W >= 2 && ((long)L.H * L.W & 255) == 0 && (long)L.H * L.W < (1L << 31);
  if (vec4) {
    const int g4 = (mt_grid_1d((long)L.B * L.F * L.H * L.W / 4, 256, 8192) + 7) & ~7;
    if (L.C == 3) hipLaunchKernelGGL((warp_fwd4_kernel<3>), dim3(g4), dim3(256), 0, S_(stream), x, v, grid, L, align_corners, xo, vo, idx);
    else if (L.C == 1) hipLaunchKernelGGL((warp_fwd4_kernel<1>), dim3(g4), dim3(256), 0, S_(stream), x, v, grid, L, align_corners, xo, vo, idx);
    else hipLaunchKernelGGL((warp_fwd4_kernel<0>), dim3(g4), dim3(256), 0, S_(stream), x, v, grid, L, align_corners, xo, vo, idx);
    return mt_check_launch("mt_warp_fwd");
  }
  hipLaunchKernelGGL(warp_fwd_kernel, dim3(mt_grid_1d((long)L.B * L.F * L.H * L.W, 256)), dim3(256), 0, S_(stream), x,
                     v, grid, L, align_corners, xo, vo, idx);
  return mt_check_launch("mt_warp_fwd");
}

extern "C" int mt_warp_bwd_grid(const float* x, const float* grid, const float* dxo, const mt_warp_layout* l,
                                int32_t align_corners, float* dgrid, void* stream) {
  WarpL L;
  int rc = warp_layout(l, &L);
  if (rc) return rc;
  MT_REQUIRE(x && grid && dxo && dgrid, "mt_warp_bwd_grid: null pointer");
  hipLaunchKernelGGL(warp_bwd_grid_kernel, dim3(mt_grid_1d((long)L.B * L.F * L.H * L.W, 256)), dim3(256), 0,
                     S_(stream), x, grid, dxo, L, align_corners, dgrid);
  return mt_check_launch("mt_warp_bwd_grid");
}

extern "C" int mt_resize_fwd(const float* src, int64_t s_plane, int64_t s_y, int64_t s_x, int32_t NP, int32_t H,
                             int32_t W, float* dst, int64_t d_plane, int64_t d_y, int64_t d_x, int32_t Ho, int32_t Wo,
                             int32_t CI, int32_t mode, void* stream) {
  MT_REQUIRE(src && dst && NP > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && CI > 0 && (mode == 0 || mode == 1),
             "mt_resize_fwd: bad arguments");
  hipLaunchKernelGGL(resize_fwd_kernel, dim3(mt_grid_1d((long)NP * Ho * Wo * CI, 256)), dim3(256), 0, S_(stream), src,
                     (long)s_plane, (long)s_y, (long)s_x, NP, H, W, dst, (long)d_plane, (long)d_y, (long)d_x, Ho, Wo,
                     CI, mode);
  return mt_check_launch("mt_resize_fwd");
}

extern "C" int mt_resize_bilinear_bwd(const float* ddst, int64_t d_plane, int64_t d_y, int64_t d_x, int32_t NP,
                                      int32_t Ho, int32_t Wo, float* dsrc, int64_t s_plane, int64_t s_y, int64_t s_x,
                                      int32_t H, int32_t W, int32_t CI, void* stream) {
  MT_REQUIRE(ddst && dsrc && NP > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && CI > 0,
             "mt_resize_bilinear_bwd: bad arguments");
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(mt_grid_1d((long)NP * H * W * CI, 256)), dim3(256), 0,
                     S_(stream), ddst, (long)d_plane, (long)d_y, (long)d_x, NP, Ho, Wo, dsrc, (long)s_plane, (long)s_y,
                     (long)s_x, H, W, CI);
  return mt_check_launch("mt_resize_bilinear_bwd");
}

extern "C" int mt_l1_fwd(const float* a, const float* b, const float* mask, const float* w, int32_t B, int32_t C,
                         int32_t MC, int64_t S, int32_t count_mask, float* acc, void* stream) {
  MT_REQUIRE(a && b && acc && B > 0 && C > 0 && S > 0 && (MC == 1 || MC == C), "mt_l1_fwd: bad arguments");
  hipLaunchKernelGGL(l1_fwd_kernel, dim3(mt_grid_1d((long)B * C * S, 256, 1024)), dim3(256), 0, S_(stream), a, b, mask,
                     w, B, C, MC, (long)S, count_mask, acc, mt_amax_scratch(S_(stream)));
  return mt_check_launch("mt_l1_fwd");
}

extern "C" int mt_l1_bwd(const float* a, const float* b, const float* mask, const float* w, int32_t B, int32_t C,
                         int32_t MC, int64_t S, const float* scale, float* da, float* db, int32_t accumulate,
                         void* stream) {
  MT_REQUIRE(a && b && scale && (da || db) && B > 0 && C > 0 && S > 0 && (MC == 1 || MC == C),
             "mt_l1_bwd: bad arguments");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(mt_grid_1d((long)B * C * S, 256)), dim3(256), 0, S_(stream), a, b, mask, w, B,
                     C, MC, (long)S, scale, da, db, accumulate);
  return mt_check_launch("mt_l1_bwd");
}

extern "C" int mt_sobel_l1_fwd(const float* a, const float* b, int32_t NP, int32_t H, int32_t W, float* acc,
                               void* stream) {
  MT_REQUIRE(a && b && acc && NP > 0 && H > 0 && W > 0, "mt_sobel_l1_fwd: bad arguments");
  hipLaunchKernelGGL(sobel_l1_fwd_kernel, dim3(mt_grid_1d((long)NP * H * W, 256, 1024)), dim3(256), 0, S_(stream), a,
                     b, NP, H, W, acc, mt_amax_scratch(S_(stream)));
  return mt_check_launch("mt_sobel_l1_fwd");
}

extern "C" int mt_sobel_l1_bwd(const float* a, const float* b, int32_t NP, int32_t H, int32_t W, const float* scale,
                               float* da, void* stream) {
  MT_REQUIRE(a && b && scale && da && NP > 0 && H > 0 && W > 0, "mt_sobel_l1_bwd: bad arguments");
  hipLaunchKernelGGL(sobel_l1_bwd_kernel, dim3(mt_grid_1d((long)NP * H * W, 256)), dim3(256), 0, S_(stream), a, b, NP,
                     H, W, scale, da);
  return mt_check_launch("mt_sobel_l1_bwd");
}

extern "C" int mt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2,
                            float step_size, float inv_bc2_sqrt, float eps, float grad_scale, void* stream) {
  MT_REQUIRE(p && g && m && v && n > 0, "mt_adam_step: bad arguments");
  hipLaunchKernelGGL(adam_kernel, dim3(mt_grid_1d(n, 256)), dim3(256), 0, S_(stream), p, g, m, v, (long)n, beta1,
                     beta2, step_size, inv_bc2_sqrt, eps, grad_scale);
  return mt_check_launch("mt_adam_step");
}

extern "C" int mt_pack_pairs(const mt_pack_src* srcs, int32_t nsrc, int32_t B, int32_t R, const mt_view* dst,
                             void* stream) {
  MT_REQUIRE(srcs && nsrc >= 1 && nsrc <= 6 && mt_view_flat(dst) && dst->N == B * R, "mt_pack_pairs: bad arguments");
  PackSrcs P;
  P.n = nsrc;
  for (int i = 0; i < nsrc; ++i) {
    MT_REQUIRE(srcs[i].p && srcs[i].C > 0 && srcs[i].c_off >= 0 && srcs[i].c_off + srcs[i].C <= dst->C,
               "mt_pack_pairs: source %d does not fit the destination channels", i);
    MT_REQUIRE((srcs[i].scale == nullptr) == (srcs[i].shift == nullptr), "mt_pack_pairs: scale/shift must come together");
    P.s[i] = srcs[i];
  }
  const long total = (long)dst->N * dst->H * dst->W * (dst->C / 4);
  hipLaunchKernelGGL(pack_pairs_kernel, dim3(mt_grid_1d(total, 256)), dim3(256), 0, S_(stream), P, B, R, dst->H, dst->W,
                     dst->p, (long)dst->sn, (long)dst->sh, (long)dst->sw, dst->C);
  return mt_check_launch("mt_pack_pairs");
}

extern "C" int mt_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2,
                                float eps, float grad_scale, float* state, void* stream) {
  MT_REQUIRE(p && g && m && v && state && n > 0, "mt_adam_step_dev: bad arguments");
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, S_(stream), state);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(mt_grid_1d(n, 256)), dim3(256), 0, S_(stream), p, g, m, v, (long)n, beta1,
                     beta2, eps, grad_scale, state);
  return mt_check_launch("mt_adam_step_dev");
}

extern "C" int mt_absmax(const mt_view* x, float* out, void* stream) {
  MT_REQUIRE(mt_view_flat(x) && out, "mt_absmax: bad arguments");
  MT_REQUIRE(x->C <= 1024 && (int64_t)x->N * x->H * x->W < (1ll << 31), "mt_absmax: more than 1024 channels or 2^31 pixels");
  const long total = (long)x->N * x->H * x->W * (x->C / 4);
  hipLaunchKernelGGL(absmax_view_kernel, dim3(mt_grid_1d(total, 256, 2048)), dim3(256), 0, S_(stream), x->p,
                     (long)x->sn, (long)x->sh, (long)x->sw, x->N, x->H, x->W, x->C / 4, out, mt_amax_scratch(S_(stream)));
  return mt_check_launch("mt_absmax");
}

extern "C" int mt_absmax_flat(const float* p, int64_t n, float* out, void* stream) {
  MT_REQUIRE(p && out && n > 0, "mt_absmax_flat: bad arguments");
  hipLaunchKernelGGL(absmax_flat_kernel, dim3(mt_grid_1d(n, 256, 1024)), dim3(256), 0, S_(stream), p, (long)n, out,
                     mt_amax_scratch(S_(stream)));
  return mt_check_launch("mt_absmax_flat");
}
