// Shared device/host pieces of the convolution engine (conv.hip, conv_halo.hip).
#pragma once
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;

struct ConvK {
  const float* x; long xsn, xsh, xsw; int H, W, C;   // physical dims of the gathered tensor
  int xni; long xsg; int yni; long ysg;              // two-level image index of x and y
  const float* wt; const float* bias;
  float* y; long ysn, ysh, ysw; int Ho, Wo, Co;
  int N, KH, KW, stride, pad, transposed, ups;
  int M, K;
  int act; float alpha;
  const float* res; long rsn, rsh, rsw; float beta;
  const float* ds; long dsn, dsh, dsw; int dact, dact_c0;
  int accumulate;
  int zb, zdiv; long xz, wz, yz;   // batched GEMM over blockIdx.z (correlation)
  int Kp;                          // split paths: padded row length of the packed weights
  int par;                         // 1: stride-2 transposed gather decomposed into 4 parity classes (blockIdx.z)
  const float* xamax; const float* wamax;   // F16X3: device scalars max|X|, max|W| (power-of-two scaling)
  float* yamax;                              // optional: raised to max|y| over the values this launch writes
  AmaxScratch* ascr;                         // kernels that reduce max|y| in two levels (raise_amax_block): the stream's scratch
  int amax_deferred;                         // 1: waves only raise slots; the launcher folds them with mt_amax_fold_launch
};

// Raise the device scalar *dst to at least v (v >= 0; non-negative floats order like their
// bit patterns).  A plain read filters out almost every block: the scalar only grows, so
// a stale copy can only cause a redundant atomic, never a lost maximum.
__device__ __forceinline__ void raise_amax(float* dst, float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0 && v > __builtin_nontemporal_load(dst))
    atomicMax(reinterpret_cast<unsigned*>(dst), __float_as_uint(v));
}

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Give
// each XCD a contiguous band of pixel tiles so that the rows two neighbouring
// tiles both gather (3x3 / 5x5 halos) are served by one L2.  Bijective for
// any tile count; placement is a speed matter only.
__device__ __forceinline__ int xcd_remap(int bid, int nt) {
  const int q = nt >> 3, r = nt & 7, xcd = bid & 7, local = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}


// Output-pixel enumeration of one launch: the plain case walks all Ho x Wo
// pixels; a parity class (py, px) of a stride-2 transposed gather walks the
// pixels (2*yy + py, 2*xx + px) only.
struct PixMap { int Hc, Wc, M, ostep, py, px; };

template <int TM, int TN>
__device__ __forceinline__ void igemm_epilogue_map(const ConvK& p, const PixMap& pm, f32x16 (&acc)[TM][TN], int m0,
                                                   int n0, int wm, int wn, int li, int lh) {
  const int HWo = pm.Hc * pm.Wc;
  float vmax = 0.f;
  // ---- epilogue -----------------------------------------------------------
  // C/D layout of the 32x32 tile: col (o) = lane & 31, row (pixel) =
  // (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * TM * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= pm.M) continue;
      const int n = m / HWo;
      const int rem = m - n * HWo;
      const int yy = rem / pm.Wc, xx = rem - yy * pm.Wc;
      const int oy = yy * pm.ostep + pm.py, ox = xx * pm.ostep + pm.px;
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
        vmax = fmaxf(vmax, fabsf(v));
      }
    }
  }
  if (p.yamax) raise_amax(p.yamax, vmax);
}

template <int TM, int TN>
__device__ __forceinline__ void igemm_epilogue(const ConvK& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                               int li, int lh) {
  const PixMap pm = {p.Ho, p.Wo, p.M, 1, 0, 0};
  igemm_epilogue_map<TM, TN>(p, pm, acc, m0, n0, wm, wn, li, lh);
}

// Fused epilogue of the halo-tile kernels for ONE lane: its output channel ``o`` is fixed, its NE
// values v[e] (accumulator * 1/scale) belong to the pixels (oy[e], ox[e]) = pix(e) of image n.
//   v = act(v + bias[o]) * alpha;  v += beta * R;  v += Y (accumulate);  v *= act'(D) for o >= dact_c0
// Everything that does not depend on the element is hoisted (the bias value, the image bases, the
// kernel-uniform switches), element offsets are 32-bit, and the optional operands are read in
// batches (all loads of a pass are in flight together) instead of one dependent load per element.
// Returns max |v| over what was written.
template <bool PRED, int NE, bool NTS = false, class PixFn>
__device__ __forceinline__ float halo_epilogue_body(const ConvK& p, int n, int o, float (&v)[NE], PixFn pix,
                                                    unsigned long long okm) {
  float* yb = p.y + ((long)(n / p.yni) * p.ysg + (long)(n % p.yni) * p.ysn + o);
  const int ysh = (int)p.ysh, ysw = (int)p.ysw;
  auto ok = [&](int e) { return !PRED || ((okm >> e) & 1); };
  if (p.res) {
    const float* rb = p.res + ((long)n * p.rsn + o);
    const int rsh = (int)p.rsh, rsw = (int)p.rsw;
    float t[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      int oy, ox;
      pix(e, oy, ox);
      t[e] = ok(e) ? rb[oy * rsh + ox * rsw] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] += p.beta * t[e];
  }
  if (p.accumulate) {
    float t[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      int oy, ox;
      pix(e, oy, ox);
      t[e] = ok(e) ? yb[oy * ysh + ox * ysw] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] += t[e];
  }
  if (p.ds && o >= p.dact_c0) {
    const float* db = p.ds + ((long)n * p.dsn + o);
    const int dsh = (int)p.dsh, dsw = (int)p.dsw;
    float t[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      int oy, ox;
      pix(e, oy, ox);
      t[e] = ok(e) ? db[oy * dsh + ox * dsw] : 1.f;
    }
    if (p.dact == MT_ACT_RELU) {                   // (kernel-uniform switch, decided once)
#pragma unroll
      for (int e = 0; e < NE; ++e) v[e] *= t[e] > 0.f ? 1.f : 0.f;
    } else if (p.dact == MT_ACT_LRELU02) {
#pragma unroll
      for (int e = 0; e < NE; ++e) v[e] *= t[e] > 0.f ? 1.f : 0.2f;
    }
  }
  float vmax = 0.f;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    int oy, ox;
    pix(e, oy, ox);
    if (ok(e)) {
      if constexpr (NTS) __builtin_nontemporal_store(v[e], yb + (oy * ysh + ox * ysw));      // (streamed: read by a later launch only)
      else yb[oy * ysh + ox * ysw] = v[e];
      vmax = fmaxf(vmax, fabsf(v[e]));
    }
  }
  return vmax;
}

// (bv: the lane's bias value, 0 without one -- kernels that finish several tiles per lane read it before their main loop,
// so that layers without residual / accumulate / gate operands issue no load after a store)
template <int NE, bool NTS = false, class PixFn>
__device__ __forceinline__ float halo_epilogue(const ConvK& p, int n, int o, float (&v)[NE], PixFn pix, float bv) {
  static_assert(NE <= 64, "one mask word");
  const bool lane_ok = o < p.Co;
  unsigned long long okm = 0;                      // bit e: element e is inside the output
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    int oy, ox;
    pix(e, oy, ox);
    if (lane_ok && oy < p.Ho && ox < p.Wo) okm |= 1ull << e;
  }
  // (the activation is a kernel-uniform switch: decided once, not per element)
  const float alpha = p.alpha;
  if (p.act == MT_ACT_NONE) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = (v[e] + bv) * alpha;
  } else if (p.act == MT_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = fmaxf(v[e] + bv, 0.f) * alpha;
  } else {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = mt_act_fn(v[e] + bv, p.act) * alpha;
  }
  constexpr unsigned long long FULL = NE == 64 ? ~0ull : ((1ull << (NE & 63)) - 1);
  // interior tiles (almost all of them): no per-element predicate anywhere
  if (__all(okm == FULL)) return halo_epilogue_body<false, NE, NTS>(p, n, o, v, pix, okm);
  return halo_epilogue_body<true, NE, NTS>(p, n, o, v, pix, okm);
}
template <int NE, class PixFn>
__device__ __forceinline__ float halo_epilogue(const ConvK& p, int n, int o, float (&v)[NE], PixFn pix) {
  return halo_epilogue<NE>(p, n, o, v, pix, (p.bias && o < p.Co) ? p.bias[o] : 0.f);
}

// The same epilogue, one element at a time (no batches: few live registers).  For the small-tile
// kernels, which are latency bound and live on occupancy (104 VGPRs = 4 blocks per CU).
template <int NE, class PixFn>
__device__ __forceinline__ float halo_epilogue_lean(const ConvK& p, int n, int o, float (&v)[NE], PixFn pix) {
  if (o >= p.Co) return 0.f;
  const float bv = p.bias ? p.bias[o] : 0.f;
  float* yb = p.y + ((long)(n / p.yni) * p.ysg + (long)(n % p.yni) * p.ysn + o);
  const float* rb = p.res ? p.res + ((long)n * p.rsn + o) : nullptr;
  const float* db = (p.ds && o >= p.dact_c0) ? p.ds + ((long)n * p.dsn + o) : nullptr;
  const int ysh = (int)p.ysh, ysw = (int)p.ysw, rsh = (int)p.rsh, rsw = (int)p.rsw, dsh = (int)p.dsh, dsw = (int)p.dsw;
  float vmax = 0.f;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    int oy, ox;
    pix(e, oy, ox);
    if (oy >= p.Ho || ox >= p.Wo) continue;
    float x = mt_act_fn(v[e] + bv, p.act) * p.alpha;
    if (rb) x += p.beta * rb[oy * rsh + ox * rsw];
    if (p.accumulate) x += yb[oy * ysh + ox * ysw];
    if (db) x *= mt_dact_fn(db[oy * dsh + ox * dsw], p.dact);
    yb[oy * ysh + ox * ysw] = x;
    vmax = fmaxf(vmax, fabsf(x));
  }
  return vmax;
}

// The same epilogue in TWO phases, for kernels that finish several tiles per lane: every optional operand of every
// tile is requested (halo_epi_load) before the first store of any tile (halo_epi_store), and the bias value comes
// from the caller, loaded before the main loop.  Why: on this ISA loads and stores share one counter (vmcnt), so
// a load issued after a store waits for the store's acknowledgement from L2 -- a bias / residual load per tile
// after the previous tile's stores costs 2 us or more per tile when every block of the launch writes at once
// (measured with wall-clock stamps, DESIGN.md "Epilogue stalls").
template <int NE>
struct EpiOps { float r[NE], y[NE], d[NE]; };

template <int NE, class PixFn>
__device__ __forceinline__ unsigned halo_epi_load(const ConvK& p, int n, int o, PixFn pix, EpiOps<NE>& t) {
  static_assert(NE <= 32, "one mask word");
  unsigned okm = 0;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    int oy, ox;
    pix(e, oy, ox);
    if (o < p.Co && oy < p.Ho && ox < p.Wo) okm |= 1u << e;
  }
  if (p.res) {
    const float* rb = p.res + ((long)n * p.rsn + o);
    const int rsh = (int)p.rsh, rsw = (int)p.rsw;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      int oy, ox;
      pix(e, oy, ox);
      t.r[e] = ((okm >> e) & 1) ? rb[oy * rsh + ox * rsw] : 0.f;
    }
  }
  if (p.accumulate) {
    const float* yb = p.y + ((long)(n / p.yni) * p.ysg + (long)(n % p.yni) * p.ysn + o);
    const int ysh = (int)p.ysh, ysw = (int)p.ysw;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      int oy, ox;
      pix(e, oy, ox);
      t.y[e] = ((okm >> e) & 1) ? yb[oy * ysh + ox * ysw] : 0.f;
    }
  }
  if (p.ds && o >= p.dact_c0) {
    const float* db = p.ds + ((long)n * p.dsn + o);
    const int dsh = (int)p.dsh, dsw = (int)p.dsw;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      int oy, ox;
      pix(e, oy, ox);
      t.d[e] = ((okm >> e) & 1) ? db[oy * dsh + ox * dsw] : 1.f;
    }
  }
  return okm;
}

// v[e] = accumulator * 1/scale; bv = bias[o] (0 without one).  Same operation order as halo_epilogue.
template <int NE, class PixFn>
__device__ __forceinline__ float halo_epi_store(const ConvK& p, int n, int o, float (&v)[NE], PixFn pix, float bv,
                                                const EpiOps<NE>& t, unsigned okm) {
  float* yb = p.y + ((long)(n / p.yni) * p.ysg + (long)(n % p.yni) * p.ysn + o);
  const int ysh = (int)p.ysh, ysw = (int)p.ysw;
  const bool dact = p.ds && o >= p.dact_c0;
  // (activation, residual, accumulate and gate are kernel-uniform switches: decided once, not per element)
  const float alpha = p.alpha;
  if (p.act == MT_ACT_NONE) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = (v[e] + bv) * alpha;
  } else if (p.act == MT_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = fmaxf(v[e] + bv, 0.f) * alpha;
  } else {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = mt_act_fn(v[e] + bv, p.act) * alpha;
  }
  if (p.res) {
    const float beta = p.beta;
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] += beta * t.r[e];
  }
  if (p.accumulate) {
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] += t.y[e];
  }
  if (dact) {
    if (p.dact == MT_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < NE; ++e) v[e] *= t.d[e] > 0.f ? 1.f : 0.f;
    } else if (p.dact == MT_ACT_LRELU02) {
#pragma unroll
      for (int e = 0; e < NE; ++e) v[e] *= t.d[e] > 0.f ? 1.f : 0.2f;
    }
  }
  float vmax = 0.f;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    int oy, ox;
    pix(e, oy, ox);
    if ((okm >> e) & 1) {
      yb[oy * ysh + ox * ysw] = v[e];
      vmax = fmaxf(vmax, fabsf(v[e]));
    }
  }
  return vmax;
}

// max|y| of a launch: two levels through the stream's scratch (common.h: raise_amax_block) where the launch function set
// one, else one filtered atomic per wave.  Every thread of every block of the launch must call it; ``lds`` = 64 bytes
// of shared memory that nobody reads any more.
__device__ __forceinline__ void finish_amax(const ConvK& p, float vmax, void* lds) {
  if (!p.yamax) return;
  if (p.ascr && p.amax_deferred) {     // long launches: no election, no wait -- a one-wave launch behind this one folds the slots
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if ((threadIdx.x & 63) == 0 && vmax > 0.f) {
      const unsigned w = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * (blockDim.x >> 6) + (threadIdx.x >> 6);
      atomicMax(&p.ascr->max[(w & 63u) * 64], __float_as_uint(vmax));
    }
    return;
  }
  if (p.ascr)
    raise_amax_block(p.yamax, p.ascr, vmax, reinterpret_cast<float*>(lds), blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                     gridDim.x * gridDim.y * gridDim.z);
  else raise_amax(p.yamax, vmax);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int NS>
__device__ __forceinline__ void split4(const f32x4& v, bf16x4 (&out)[NS]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float r = v[j];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const __bf16 h = (__bf16)r;
      out[s][j] = h;
      r -= (float)h;
    }
  }
}

// acc += sum over the retained partial products, smallest first
template <int NS>
__device__ __forceinline__ void mfma_split(const bf16x8 (&a)[NS], const bf16x8 (&b)[NS], f32x16& acc) {
  if constexpr (NS == 3) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// ---- split-fp16 with per-tensor power-of-two scaling ("f16x3") -------------
// fp16 carries 11 significant bits, so TWO terms (hi = fp16(a*s), lo =
// fp16(a*s - hi)) hold 22 bits and three partial products (lo*hi, hi*lo, hi*hi)
// reproduce the fp32 product to ~7e-8 relative -- below the fp32 accumulation
// noise -- at HALF the MFMA work of bf16x6.  fp16's narrow exponent range is
// handled by one power-of-two scale per tensor, s = 2^(14 - e) with max|a| =
// m * 2^e: the scaled tensor tops out in [2^13, 2^14), elements more than ~2^27
// below the maximum degrade gracefully towards fixed point.  max|a| comes from
// mt_absmax (a device scalar: nothing syncs with the host).
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float pow2_scale(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(amax, &e);
  int k = 14 - e;
  k = k > 126 ? 126 : (k < -126 ? -126 : k);
  return ldexpf(1.f, k);
}

// SHADOW = 1: for code that runs beside MFMAs of the same SIMD (the staging waves of the producer /
// consumer weight-gradient kernel).  Left to itself the compiler builds the split from v_pk_mul_f32
// and v_fma_mixlo/hi_f16; packed-fp32, fma_mix*_f16 and dot2 instructions execute on the matrix
// pipe's side of the SIMD and stall it -- measured (scratch/lab/coissue.hip): four of them per
// MFMA, from the same or from another wave, cost 30-40 % of the MFMA rate, while v_mul_f32,
// v_cvt_pk_f16_f32, v_cvt_f32_f16 and v_sub_f32 run in the MFMAs' shadow for free.  The shadow
// form spends twice the instructions, so code that is NOT overlapped with MFMAs keeps SHADOW = 0.
#ifndef MT_SHADOW_DEFAULT
#define MT_SHADOW_DEFAULT 0
#endif
template <int NS, int FMT, int SHADOW = MT_SHADOW_DEFAULT>
__device__ __forceinline__ void split4f(const f32x4& v, float scale, bf16x4 (&out)[NS]) {
  if constexpr (FMT == 0) {
    split4<NS>(v, out);
  } else if constexpr (SHADOW == 3) {
    // (lab) FOUR instructions per value pair: the scale rides in the mixed-precision FMAs -- hi = fp16(v * scale + 0),
    // lo = fp16(v * scale - hi) (v * scale is exact: a power of two) -- instead of two v_mul_f32 and a v_cvt_pk_f16_f32
    // in front of them.  Bit-identical to the other forms; all four instructions share the matrix pipe.
    static_assert(NS == 2, "f16x3 uses two fp16 terms");
    unsigned hi[2], lo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi[j]) : "v"(v[2 * j]), "v"(scale));
      asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi[j]) : "v"(v[2 * j + 1]), "v"(scale));
      asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo[j]) : "v"(v[2 * j]), "v"(scale), "v"(hi[j]));
      asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo[j]) : "v"(v[2 * j + 1]), "v"(scale), "v"(hi[j]));
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    out[0] = __builtin_bit_cast(bf16x4, u32x2{hi[0], hi[1]});
    out[1] = __builtin_bit_cast(bf16x4, u32x2{lo[0], lo[1]});
  } else if constexpr (SHADOW == 2) {
    // The fewest instructions (5 per value pair; hipcc's own form mixes both of the others and spends ~10): for code
    // whose bound is VALU issue, not the matrix pipe (the xi-split kernels: 32 output channels per transformed value).
    // lo = fp16(r - hi) in ONE mixed-precision FMA (r * 1.0 - hi with the fp16 operand read in place; r - hi is exact in
    // fp32, so this rounds once, like the other forms: bit-identical).  v_fma_mix* shares the matrix pipe (see SHADOW).
    static_assert(NS == 2, "f16x3 uses two fp16 terms");
    unsigned hi[2], lo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float r0, r1;
      asm("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(v[2 * j]), "v"(scale));
      asm("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(v[2 * j + 1]), "v"(scale));
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi[j]) : "v"(r0), "v"(r1));
      asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo[j]) : "v"(r0), "v"(hi[j]));
      asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo[j]) : "v"(r1), "v"(hi[j]));
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    out[0] = __builtin_bit_cast(bf16x4, u32x2{hi[0], hi[1]});
    out[1] = __builtin_bit_cast(bf16x4, u32x2{lo[0], lo[1]});
  } else if constexpr (SHADOW) {
    static_assert(NS == 2, "f16x3 uses two fp16 terms");
    unsigned hi[2], lo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float r0, r1, h0, h1, d0, d1;
      asm("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(v[2 * j]), "v"(scale));
      asm("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(v[2 * j + 1]), "v"(scale));
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi[j]) : "v"(r0), "v"(r1));
      asm("v_cvt_f32_f16 %0, %1" : "=v"(h0) : "v"(hi[j]));
      asm("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(h1) : "v"(hi[j]));
      asm("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(r0), "v"(h0));
      asm("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(r1), "v"(h1));
      asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo[j]) : "v"(d0), "v"(d1));
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    out[0] = __builtin_bit_cast(bf16x4, u32x2{hi[0], hi[1]});
    out[1] = __builtin_bit_cast(bf16x4, u32x2{lo[0], lo[1]});
  } else {
    static_assert(NS == 2, "f16x3 uses two fp16 terms");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float r = v[j] * scale;
      const _Float16 h = (_Float16)r;
      const _Float16 l = (_Float16)(r - (float)h);
      out[0][j] = __builtin_bit_cast(__bf16, h);
      out[1][j] = __builtin_bit_cast(__bf16, l);
    }
  }
}

template <int NS, int FMT>
__device__ __forceinline__ void mfma_splitf(const bf16x8 (&a)[NS], const bf16x8 (&b)[NS], f32x16& acc) {
  if constexpr (FMT == 0) {
    mfma_split<NS>(a, b, acc);
  } else {
    const half8 a0 = __builtin_bit_cast(half8, a[0]), a1 = __builtin_bit_cast(half8, a[1]);
    const half8 b0 = __builtin_bit_cast(half8, b[0]), b1 = __builtin_bit_cast(half8, b[1]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
  }
}

struct WgK {
  const float* d; long dsn, dsh, dsw; int Hd, Wd, Cd;
  const float* g; long gsn, gsh, gsw; int Hg, Wg, Cg;
  int dni; long dsg; int gni; long gsg;
  int KH, KW, stride, pad, ups;
  int M, per_split;   // pixels, pixels per split (multiple of 32)
  int atiles, btiles;
  float* slabs;
  const float* damax; const float* gamax;   // F16X3: device scalars max|D|, max|G|
  unsigned skip;                            // halo path: bit (at*btiles+bt) = tile not wanted
  // wgrad_pc_kernel (3 x 3): ntl > 0 = the launch computes the tiles of this list instead of the 64-aligned grid
  // (mt_wgrad.tiles, one 64-bit word each: a[0] | a[1] << 8 | b[0] << 16 | b[1] << 24 | quads << 32 | bias << 40)
  int ntl;
  unsigned long long tl[MT_WGRAD_TILES_MAX];
  // block order of the halo-tile weight-gradient grids (wg_block): gx = channel tiles (x kernel-row groups) per pixel
  // range, gy = pixel ranges (splits) per problem, nrg = pixel ranges of the launch (gy x problems of a batch); the launch
  // is 1-D with gx * gyp blocks, gyp = nrg rounded up to 8
  int gx, gy, gyp, nrg, xcd;      // xcd = 1: the XCD-aware order (launches with at least 8 pixel ranges)
};
#ifndef MT_WG_XCD
#define MT_WG_XCD 1      // (0: the plain order -- channel tile fastest, then pixel range: A/B)
#endif
inline unsigned wg_grid(WgK* k, int gx, int splits, int problems = 1) {
  const int ranges = splits * problems;                // pixel ranges of the launch (every problem of a batch has `splits`)
  k->xcd = MT_WG_XCD && ranges >= 8;
  k->gx = gx; k->gy = splits; k->nrg = ranges; k->gyp = k->xcd ? (ranges + 7) & ~7 : ranges;
  return (unsigned)(gx * k->gyp);
}
#ifdef __HIPCC__
// Blocks are dealt round-robin over the 8 XCDs in launch order, each XCD with its own L2.  The channel tiles of ONE
// pixel range read the same pixels of D and G (a 2 x 2 tiling reads every byte twice, the dense block's four listed
// tiles 2 - 2.4 times): they become consecutive blocks of ONE XCD -- in flight together, so that all but the first
// find the pixels in that L2 -- instead of neighbours in launch order, which land on different XCDs.
// Returns false for a padding block (pixel range beyond the last).  bx: channel tile, by: pixel range, bz: problem.
__device__ __forceinline__ bool wg_block(const WgK& p, int& bx, int& by, int& bz) {
  const int id = blockIdx.x;
  int rg;                                              // pixel range of the launch: problem * splits + split
  if (p.xcd) {
    const int slot = id >> 3;
    bx = slot % p.gx;
    rg = (slot / p.gx) * 8 + (id & 7);
  } else {
    bx = id % p.gx;
    rg = id / p.gx;
  }
  bz = rg / p.gy;
  by = rg - bz * p.gy;
  return rg < p.nrg;
}
#endif
inline void wgk_tiles(const mt_wgrad* d, WgK* k) {
  k->ntl = d->n_tiles;
  for (int i = 0; i < MT_WGRAD_TILES_MAX; ++i) {
    const mt_wgrad_tile& t = d->tiles[i];
    k->tl[i] = i < d->n_tiles ? ((unsigned long long)t.a[0] | (unsigned long long)t.a[1] << 8 | (unsigned long long)t.b[0] << 16 |
                                 (unsigned long long)t.b[1] << 24 | (unsigned long long)(t.quads & 15) << 32 |
                                 (unsigned long long)(t.bias ? 1 : 0) << 40) : 0ull;
  }
}

inline int fill_convk(const mt_conv* d, ConvK* k) {
  MT_REQUIRE(d, "mt_conv: null descriptor");
  MT_REQUIRE(mt_view_ok(&d->x) && mt_view_ok(&d->y), "mt_conv: x/y view invalid (need 16B-aligned pointer, C%%4==0, strides%%4==0)");
  MT_REQUIRE(d->wt && (reinterpret_cast<uintptr_t>(d->wt) & 15) == 0, "mt_conv: weights null/unaligned");
  MT_REQUIRE(d->stride == 1 || d->stride == 2, "mt_conv: stride must be 1 or 2 (got %d)", d->stride);
  MT_REQUIRE(d->KH > 0 && d->KW > 0 && d->pad >= 0 && d->ups >= 0 && d->ups <= 2, "mt_conv: bad kernel/pad/ups");
  MT_REQUIRE(d->ups == 0 || (d->stride == 1 && !d->transposed), "mt_conv: ups needs stride 1, non-transposed");
  MT_REQUIRE(d->x.N == d->y.N, "mt_conv: batch mismatch %d vs %d", d->x.N, d->y.N);
  const int HL = d->x.H << d->ups, WL = d->x.W << d->ups;
  if (!d->transposed) {
    int ho = (HL + 2 * d->pad - d->KH) / d->stride + 1, wo = (WL + 2 * d->pad - d->KW) / d->stride + 1;
    MT_REQUIRE(ho == d->y.H && wo == d->y.W, "mt_conv: output %dx%d does not match conv geometry %dx%d", d->y.H, d->y.W, ho, wo);
  } else {
    // y is the larger tensor: conv(y) must map back onto x's logical size
    int hi = (d->y.H + 2 * d->pad - d->KH) / d->stride + 1, wi = (d->y.W + 2 * d->pad - d->KW) / d->stride + 1;
    MT_REQUIRE(hi == HL && wi == WL, "mt_conv(transposed): y %dx%d is not a pre-image of x %dx%d", d->y.H, d->y.W, HL, WL);
  }
  MT_REQUIRE((int64_t)d->y.N * d->y.H * d->y.W < (1ll << 31), "mt_conv: too many pixels");
  if (d->res.p) MT_REQUIRE(mt_view_flat(&d->res) && d->res.C >= d->y.C, "mt_conv: residual view invalid");
  if (d->dsrc.p) MT_REQUIRE(mt_view_flat(&d->dsrc) && d->dsrc.C >= d->y.C, "mt_conv: dsrc view invalid");
  k->xni = d->x.n_in; k->xsg = d->x.sg; k->yni = d->y.n_in; k->ysg = d->y.sg;
  k->x = d->x.p; k->xsn = d->x.sn; k->xsh = d->x.sh; k->xsw = d->x.sw; k->H = d->x.H; k->W = d->x.W; k->C = d->x.C;
  k->wt = d->wt; k->bias = d->bias;
  k->y = d->y.p; k->ysn = d->y.sn; k->ysh = d->y.sh; k->ysw = d->y.sw; k->Ho = d->y.H; k->Wo = d->y.W; k->Co = d->y.C;
  k->N = d->x.N; k->KH = d->KH; k->KW = d->KW; k->stride = d->stride; k->pad = d->pad; k->transposed = d->transposed; k->ups = d->ups;
  k->M = d->y.N * d->y.H * d->y.W; k->K = d->KH * d->KW * d->x.C;
  k->act = d->act; k->alpha = d->alpha;
  k->res = d->res.p; k->rsn = d->res.sn; k->rsh = d->res.sh; k->rsw = d->res.sw; k->beta = d->beta;
  k->ds = d->dsrc.p; k->dsn = d->dsrc.sn; k->dsh = d->dsrc.sh; k->dsw = d->dsrc.sw; k->dact = d->dact; k->dact_c0 = d->dact_c0;
  k->accumulate = d->accumulate;
  k->zb = 1; k->zdiv = 1; k->xz = k->wz = k->yz = 0; k->par = 0; k->Kp = 0;
  k->xamax = d->x_amax; k->wamax = d->w_amax; k->yamax = d->y_amax;
  k->ascr = nullptr;                 // (set by the launch functions of the kernels that use it)
  k->amax_deferred = 0;
  return MT_OK;
}


}  // namespace

// Winograd F(2,3) weights of igemm_halo_wino_kernel in MFMA fragment order:
// dst[tap 12 = ky*4 + xi][chunk][ntile][wn WN][kstep 2][plane 2][lane 64][8], lane (li, lh) holds
// U[ky][xi][a = (ntile*WN + wn)*32 + li][b = chunk*32 + kstep*16 + 8*lh + j],  U = G g[ky][:]:
// (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2) over the kernel row ky (taps flipped first when `flip`).
__device__ __forceinline__ void pack_halo_wino_body(const float* src, int A, int B, long sa, long sb, long st, __bf16* dst,
                                                    int nchunk, int ntile, int flip, const float* wamax, int nchunk_total,
                                                    int ch_off, int WN) {
  const float sc = wamax ? pow2_scale(*wamax) : 1.f;
  const long total = (long)12 * nchunk * ntile * WN * 2 * 64 * 8;      // elements per plane
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), ks = (int)((e >> 9) & 1);
    long r = e >> 10;
    const int wv = (int)(r % WN); r /= WN;
    const int nt = (int)(r % ntile); r /= ntile;
    const int ch = (int)(r % nchunk);
    const int t = (int)(r / nchunk);
    const int ky = t >> 2, xi = t & 3;
    const int a = (nt * WN + wv) * 32 + (lane & 31);
    const int b = ch * 32 + ks * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (a < A && b < B) {
      const float* g = src + a * sa + b * sb;
      float g0 = g[(flip ? 8 - (ky * 3 + 0) : ky * 3 + 0) * st];
      float g1 = g[(flip ? 8 - (ky * 3 + 1) : ky * 3 + 1) * st];
      float g2 = g[(flip ? 8 - (ky * 3 + 2) : ky * 3 + 2) * st];
      v = xi == 0 ? g0 : (xi == 3 ? g2 : (xi == 1 ? 0.5f * ((g0 + g2) + g1) : 0.5f * ((g0 + g2) - g1)));
    }
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

// Winograd F(4,3) weights of igemm_wino43_kernel (conv_halo_wino43.hip), U = G g[ky][:], in MFMA fragment order:
// dst[tap 18 = ky*6 + xi][chunk][ntile][wn 4][kstep 2][plane 2][lane 64][8], lane (li, lh) holds
// U[ky][xi][a = (ntile*4 + wn)*32 + li][b = chunk*32 + kstep*16 + 8*lh + j]   (taps flipped first when `flip`)
__device__ __forceinline__ void pack_halo_wino43_body(const float* src, int A, int B, long sa, long sb, long st, __bf16* dst,
                                                      int nchunk, int ntile, int flip, const float* wamax, int nchunk_total,
                                                      int ch_off) {
  const float sc = wamax ? pow2_scale(*wamax) : 1.f;
  constexpr int WN = 4;
  const long total = (long)18 * nchunk * ntile * WN * 2 * 64 * 8;      // elements per plane
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), ks = (int)((e >> 9) & 1);
    long r = e >> 10;
    const int wv = (int)(r % WN); r /= WN;
    const int nt = (int)(r % ntile); r /= ntile;
    const int ch = (int)(r % nchunk);
    const int t = (int)(r / nchunk);
    const int ky = t / 6, xi = t - ky * 6;
    const int a = (nt * WN + wv) * 32 + (lane & 31);
    const int b = ch * 32 + ks * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (a < A && b < B) {
      const float* g = src + a * sa + b * sb;
      const float g0 = g[(flip ? 8 - (ky * 3 + 0) : ky * 3 + 0) * st];
      const float g1 = g[(flip ? 8 - (ky * 3 + 1) : ky * 3 + 1) * st];
      const float g2 = g[(flip ? 8 - (ky * 3 + 2) : ky * 3 + 2) * st];
      switch (xi) {
        case 0: v = 0.25f * g0; break;
        case 1: v = -(1.f / 6.f) * ((g0 + g2) + g1); break;
        case 2: v = -(1.f / 6.f) * ((g0 + g2) - g1); break;
        case 3: v = (1.f / 24.f) * ((g0 + 4.f * g2) + 2.f * g1); break;
        case 4: v = (1.f / 24.f) * ((g0 + 4.f * g2) - 2.f * g1); break;
        default: v = g2; break;
      }
    }
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

// Winograd F(2,5) weights of igemm_wino25_kernel (conv_halo_wino25.hip), U = G g[ky][:], in MFMA fragment order:
// dst[tap 30 = ky*6 + xi][chunk][ntile][wn 4][kstep 2][plane 2][lane 64][8], lane (li, lh) holds
// U[ky][xi][a = (ntile*4 + wn)*32 + li][b = chunk*32 + kstep*16 + 8*lh + j]   (taps flipped first when `flip`)
__device__ __forceinline__ void pack_halo_wino25_body(const float* src, int A, int B, long sa, long sb, long st, __bf16* dst,
                                                      int nchunk, int ntile, int flip, const float* wamax, int nchunk_total,
                                                      int ch_off) {
  const float sc = wamax ? pow2_scale(*wamax) : 1.f;
  constexpr int WN = 4;
  const long total = (long)30 * nchunk * ntile * WN * 2 * 64 * 8;      // elements per plane
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), ks = (int)((e >> 9) & 1);
    long r = e >> 10;
    const int wv = (int)(r % WN); r /= WN;
    const int nt = (int)(r % ntile); r /= ntile;
    const int ch = (int)(r % nchunk);
    const int t = (int)(r / nchunk);
    const int ky = t / 6, xi = t - ky * 6;
    const int a = (nt * WN + wv) * 32 + (lane & 31);
    const int b = ch * 32 + ks * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (a < A && b < B) {
      const float* g = src + a * sa + b * sb;
      float gk[5];
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) gk[kx] = g[(flip ? 24 - (ky * 5 + kx) : ky * 5 + kx) * st];
      const float e02 = gk[0] + gk[2], e4 = gk[4], o13 = gk[1] + gk[3];
      switch (xi) {
        case 0: v = 0.25f * gk[0]; break;
        case 1: v = -(1.f / 6.f) * ((e02 + e4) + o13); break;
        case 2: v = -(1.f / 6.f) * ((e02 + e4) - o13); break;
        case 3: v = (1.f / 24.f) * ((gk[0] + 4.f * gk[2] + 16.f * gk[4]) + (2.f * gk[1] + 8.f * gk[3])); break;
        case 4: v = (1.f / 24.f) * ((gk[0] + 4.f * gk[2] + 16.f * gk[4]) - (2.f * gk[1] + 8.f * gk[3])); break;
        default: v = gk[4]; break;
      }
    }
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


// ---- stride-2 layers as window convolutions over "units" (conv_halo_planes.hip) -----------------
// Unit u = parity plane (a, b) = (u >> 1, u & 1) of X for the strided gather, output parity class
// (A, B) for the transposed gather.  Window position (dy + 1) * 3 + (dx + 1), dy, dx in {-1, 0, 1},
// stands for the kernel tap  kh = 2 dy + a + pad  (strided)  /  kh = A + pad - 2 dy  (transposed).
struct UnitTaps { unsigned mask[4]; signed char src[4][9]; int ntaps[4]; };

__host__ __device__ inline UnitTaps unit_taps(int transposed, int K, int pad) {
  UnitTaps t = {};
  for (int u = 0; u < 4; ++u) {
    const int a = u >> 1, b = u & 1;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int kh = transposed ? a + pad - 2 * dy : 2 * dy + a + pad;
        const int kw = transposed ? b + pad - 2 * dx : 2 * dx + b + pad;
        const int wpos = (dy + 1) * 3 + (dx + 1);
        t.src[u][wpos] = -1;
        if (kh >= 0 && kh < K && kw >= 0 && kw < K) {
          t.mask[u] |= 1u << wpos;
          t.src[u][wpos] = (signed char)(kh * K + kw);
          ++t.ntaps[u];
        }
      }
  }
  return t;
}

// Weight fragments in the order igemm_halo_planes_kernel consumes them:
// dst[frag = (unit, chunk, tap of the unit's mask)][ntile][wn WN][kstep 2][plane 2][lane 64][8]
__device__ __forceinline__ void pack_planes_body(const float* src, int A, int B, long sa, long sb, long st, __bf16* dst,
                                                 int nchunk, int ntile, int WN, const float* wamax, const UnitTaps& t) {
  const float sc = wamax ? pow2_scale(*wamax) : 1.f;
  const long nfrag = (long)(t.ntaps[0] + t.ntaps[1] + t.ntaps[2] + t.ntaps[3]) * nchunk;
  const long total = nfrag * ntile * WN * 2 * 64 * 8;      // elements per plane
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63), ks = (int)((e >> 9) & 1);
    long r = e >> 10;
    const int wv = (int)(r % WN); r /= WN;
    const int nt = (int)(r % ntile); r /= ntile;
    const long fidx = r;
    int u = 0;
    long f = fidx;
    while (u < 3 && f >= (long)t.ntaps[u] * nchunk) { f -= (long)t.ntaps[u] * nchunk; ++u; }
    const int ch = (int)(f / t.ntaps[u]), ti = (int)(f % t.ntaps[u]);
    unsigned m = t.mask[u];
    for (int s = 0; s < ti; ++s) m &= m - 1;
    const int tap = t.src[u][__ffs((int)m) - 1];
    const int a = (nt * WN + wv) * 32 + (lane & 31), b = ch * 32 + ks * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (a < A && b < B) v = src[a * sa + b * sb + tap * st];
    __bf16* o = dst + ((((fidx * ntile + nt) * WN + wv) * 2 + ks) * 2) * 512 + lane * 8 + j;
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

// packed weights for the generic split kernels: dst[plane][a][Kp] (see mt_weight_pack_split)
__device__ __forceinline__ void pack_split_body(const float* src, int A, int B, int T, long sa, long sb, long st, __bf16* dst,
                                         int AP, int BP, int Kp, int NS, int flip, const float* wamax) {
  const long total = (long)AP * Kp;
  const float sc = wamax ? pow2_scale(*wamax) : 1.f;    // F16X3: fp16 planes of W * 2^k
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % Kp);
    const int a = (int)(e / Kp);
    float v = 0.f;
    if (k < T * BP) {
      const int t = k / BP, b = k - t * BP;
      if (a < A && b < B) v = src[a * sa + b * sb + (flip ? T - 1 - t : t) * st];
    }
    if (wamax) {
      v *= sc;
      const _Float16 h = (_Float16)v;
      dst[e] = __builtin_bit_cast(__bf16, h);
      dst[total + e] = __builtin_bit_cast(__bf16, (_Float16)(v - (float)h));
      continue;
    }
    for (int s = 0; s < NS; ++s) {
      const __bf16 h = (__bf16)v;
      dst[s * total + e] = h;
      v -= (float)h;
    }
  }
}
