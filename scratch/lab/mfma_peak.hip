// How fast can v_mfma_f32_32x32x16_f16 run on this chip with operands already in registers?
// mode 0: random operands; mode 1: zero operands.  2 blocks of 4 waves per CU (= the conv kernels' occupancy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void peak(const _Float16* src, float* out, int iters) {
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
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 1) & 3], b[0], acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[0], acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
int main() {
  const int N = 16384;
  _Float16* h = (_Float16*)malloc(N * 2);
  _Float16* d; float* o;
  hipMalloc(&d, N * 2); hipMalloc(&o, 4096);
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < N; ++i) h[i] = mode ? (_Float16)0.f : (_Float16)(((rand() % 2001) - 1000) / 500.f);
    hipMemcpy(d, h, N * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 512;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(peak, dim3(blocks), dim3(256), 0, 0, d, o, iters);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)blocks * 4 * iters * 24 * 32768.0;
      printf("mode %d (%s) rep %d: %.3f ms  %.0f TFLOP/s\n", mode, mode ? "zeros" : "random", rep, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
