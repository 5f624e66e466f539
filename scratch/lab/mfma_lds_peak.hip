// MFMA rate when every fragment comes from LDS at the conv kernels' ratio (8 ds_read_b128 per 12 MFMAs,
// conflict-free rows of 80 bytes), 2 blocks of 4 waves per CU, random fp16 data, no global traffic in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int READS>
__global__ __launch_bounds__(256, 2) void peak(const _Float16* src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 sm[2][180 * 40];
  for (int i = threadIdx.x; i < 2 * 180 * 40; i += 256) (&sm[0][0])[i] = src[(i * 7 + blockIdx.x) & 16383];
  __syncthreads();
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  half8 b[2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) b[i][j] = src[(threadIdx.x * 8 + j + i * 4099 + 7) & 16383];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int off = 0;
  for (int it = 0; it < iters; ++it) {
    half8 a[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (i * 2 + s < READS) a[i][s] = *reinterpret_cast<const half8*>(&sm[s][((i * 32 + li + off) % 160) * 40 + 8 * lh]);
        else a[i][s] = b[s];
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[0], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[1], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[0], acc[i], 0, 0, 0);
    }
    off = (off + 19) % 20;
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int READS> void run(const _Float16* d, float* o) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 512;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(peak<READS>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 12 * 32768.0;
    if (rep == 2) printf("%d ds_read_b128 per 12 MFMA: %.3f ms  %.0f TFLOP/s\n", READS, ms, flops / ms / 1e9);
  }
}
int main() {
  const int N = 16384;
  _Float16* h = (_Float16*)malloc(N * 2);
  for (int i = 0; i < N; ++i) h[i] = (_Float16)(((rand() % 2001) - 1000) / 500.f);
  _Float16* d; float* o;
  hipMalloc(&d, N * 2); hipMalloc(&o, 4096);
  hipMemcpy(d, h, N * 2, hipMemcpyHostToDevice);
  run<0>(d, o); run<4>(d, o); run<8>(d, o);
  return 0;
}
