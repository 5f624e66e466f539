// Does VALU work overlap with v_mfma_f32_32x32x16_f16 on one SIMD?  One 512-thread block per CU:
// mode A (cross): waves 0-3 run MFMA triples (dependent through the accumulator, as the conv kernels'
// split products are), waves 4-7 run K fp32 FMAs per MFMA of their SIMD partner.
// mode B (same): waves 0-3 only, K independent FMAs written after every MFMA in the same stream.
// mode C: the VALU work alone (waves 4-7 only) / mode D: the MFMA work alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef OP
#define OP 0
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int K>
__device__ __forceinline__ void valu(float (&v)[8], float m) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 1) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 2) asm volatile("v_fma_mixlo_f16 %0, %0, %1, %0" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 3) { f32x2& p = *reinterpret_cast<f32x2*>(&v[(k & 3) * 2]); asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p)); }
    if (OP == 4) asm volatile("v_cvt_f32_f16_e32 %0, %0" : "+v"(v[k & 7]));
    if (OP == 5) { f32x2& p = *reinterpret_cast<f32x2*>(&v[(k & 3) * 2]); asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p)); }
    if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 8) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(v[k & 7]));
    if (OP == 9) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 10) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 11) asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 12) { f32x2& p = *reinterpret_cast<f32x2*>(&v[(k & 3) * 2]); asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p)); }
    if (OP == 13) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v[k & 7]));
    if (OP == 14) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 15) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 16) asm volatile("v_fma_mix_f32 %0, %0, %1, %0" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 17) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 18) asm volatile("v_dot2_f32_f16 %0, %0, %1, %0" : "+v"(v[k & 7]) : "v"(m));
    if (OP == 7) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(*reinterpret_cast<f32x2*>(&v[(k & 3) * 2])));
  }
}

template <int MODE, int K>
__global__ __launch_bounds__(512, 2) void co(const _Float16* src, float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = (float)src[(threadIdx.x + k * 77) & 16383];
  const float m = 1.0000001f;
  if (wave >= 4) {
    if (MODE == 0 || MODE == 2) {     // cross partner / VALU alone
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 24; ++u) valu<K>(v, m);
      float s = 0.f; for (int k = 0; k < 8; ++k) s += v[k];
      if (s == 12345.678f) out[threadIdx.x] = s;
    }
    return;
  }
  if (MODE == 2) return;
  half8 a[4], b[2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = src[(threadIdx.x * 8 + j + i * 2048) & 16383];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = src[(threadIdx.x * 8 + j + i * 4099 + 7) & 16383];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[1], acc[i], 0, 0, 0);
        if (MODE == 1) valu<K>(v, m);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 1) & 3], b[0], acc[i], 0, 0, 0);
        if (MODE == 1) valu<K>(v, m);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[0], acc[i], 0, 0, 0);
        if (MODE == 1) valu<K>(v, m);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int k = 0; k < 8; ++k) s += v[k];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE, int K>
void run(const char* name, const _Float16* d, float* o) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 8000, blocks = 256;
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((co<MODE, K>), dim3(blocks), dim3(512), 0, 0, d, o, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * iters * 24 * 32768.0;
  printf("%-10s K=%2d: %8.3f ms  (MFMA work alone would be %.0f TFLOP/s at this time)\n", name, K, best, flops / best / 1e9);
}

int main() {
  const int N = 16384;
  _Float16* h = (_Float16*)malloc(N * 2);
  _Float16* d; float* o;
  hipMalloc(&d, N * 2); hipMalloc(&o, 4096);
  for (int i = 0; i < N; ++i) h[i] = (_Float16)(((rand() % 2001) - 1000) / 500.f);
  hipMemcpy(d, h, N * 2, hipMemcpyHostToDevice);
  printf("OP %d\n", OP);
  run<3, 0>("mfma only", d, o);
  run<2, 4>("valu only", d, o);
  run<0, 2>("cross", d, o); run<0, 4>("cross", d, o);
  run<1, 4>("same", d, o);
  return 0;
}
