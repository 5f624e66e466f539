// Shared host/device helpers for the mt355 kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mt355.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void mt_set_error(const char* fmt, ...);
int mt_check_launch(const char* what);

#define MT_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      mt_set_error(__VA_ARGS__);         \
      return MT_ERR_ARG;                 \
    }                                    \
  } while (0)

static inline bool mt_view_ok(const mt_view* v) {
  return v && v->p && (reinterpret_cast<uintptr_t>(v->p) & 15) == 0 &&
         v->C > 0 && (v->C & 3) == 0 && (v->sn & 3) == 0 && (v->sh & 3) == 0 &&
         (v->sw & 3) == 0 && (v->sg & 3) == 0 && v->N > 0 && v->H > 0 && v->W > 0 && v->n_in > 0;
}
// entry points that do not implement the two-level image index
static inline bool mt_view_flat(const mt_view* v) { return mt_view_ok(v) && v->n_in >= v->N; }

static inline int mt_grid_1d(int64_t work, int block, int cap = 4096) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

__device__ __forceinline__ float mt_act_fn(float v, int act) {
  if (act == MT_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == MT_ACT_LRELU02) return v > 0.f ? v : 0.2f * v;
  return v;
}
// derivative expressed on the SAVED POST-activation value s (sign(s) == sign
// of the pre-activation for both ReLU and LeakyReLU(0.2)); torch uses
// grad * (s > 0) for ReLU and (s > 0 ? 1 : 0.2) for LeakyReLU.
__device__ __forceinline__ float mt_dact_fn(float s, int act) {
  if (act == MT_ACT_RELU) return s > 0.f ? 1.f : 0.f;
  if (act == MT_ACT_LRELU02) return s > 0.f ? 1.f : 0.2f;
  return 1.f;
}

__device__ __forceinline__ float mt_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
