// How long does the write path take to absorb a conv epilogue?  1024 blocks (8 x 16-pixel tiles of 32 images of 64 x 64,
// 32 channels) each write their 16 KB of output the way igemm_winox_kernel does (mode 0: one dword per lane, the 32 lanes of
// a half-wave = the 32 channels of a pixel) or as 16 bytes per lane (mode 1: 8 lanes = the 32 channels of a pixel), after
// optionally streaming `rd` KB of input per block through the L2.  Reports the launch time (events) and, per block, the
// time from the first store to the acknowledgement of the last one (s_waitcnt vmcnt(0)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 3) void wr(const f32x4* in, float* y, int rd16, long long* stamps, int C, unsigned* amax, int atom) {
  const int tid = threadIdx.x, lane = tid & 63, xi = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int bid = blockIdx.x, tx = bid & 3, ty = (bid >> 2) & 7, n = bid >> 5;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < rd16; ++i) s += in[((long)bid * rd16 + i) * 256 + tid];
  float v = s[0] + s[1] + s[2] + s[3] + 1.f;
  __syncthreads();
  long long t0 = wall_clock64();
  float* yb = y + (long)n * 64 * 64 * C;
  if (MODE == 0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int oy = ty * 8 + 4 * a + xi, ox = tx * 16 + 8 * lh + e;
        yb[((long)oy * 64 + ox) * C + li] = v + e;
      }
  } else {
    const int c4 = lane & 7, cp = lane >> 3;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int oy = ty * 8 + 4 * a + xi, ox = tx * 16 + 2 * cp + e;
        f32x4 o = {v, v + 1, v + 2, v + e};
        *reinterpret_cast<f32x4*>(&yb[((long)oy * 64 + ox) * C + 4 * c4]) = o;
      }
  }
  // atom 1: one atomicMax per wave on ONE address (what raise_amax does when its filter read is stale), 2: the same on 64
  // addresses 256 bytes apart (slot = block & 63), 3: one per wave preceded by the filter read (values grow with the block id)
  if (atom == 1 && lane == 0) atomicMax(amax, __float_as_uint(v + bid + xi));
  if (atom == 2 && lane == 0) atomicMax(amax + 64 * (bid & 63), __float_as_uint(v + bid + xi));
  if (atom == 3 && lane == 0 && v + bid + xi > __builtin_nontemporal_load(reinterpret_cast<float*>(amax))) atomicMax(amax, __float_as_uint(v + bid + xi));
  long long t1 = wall_clock64();
  __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0) expcnt(0) lgkmcnt(0): the stores are acknowledged
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t2 = wall_clock64();
  if (tid == 0) { stamps[bid * 3] = t0; stamps[bid * 3 + 1] = t1; stamps[bid * 3 + 2] = t2; }
}
int main() {
  const int NB = 1024, C = 32;
  float* y; f32x4* in; long long* st;
  const size_t in_bytes = (size_t)NB * 96 * 256 * 16;      // up to 96 x 4 KB = 384 KB per block
  hipMalloc(&y, (size_t)32 * 64 * 64 * C * 4); hipMalloc(&in, in_bytes); hipMalloc(&st, NB * 3 * 8); unsigned* am; hipMalloc(&am, 64 * 64 * 4); hipMemset(am, 0, 64 * 64 * 4);
  hipMemset(in, 0, in_bytes);
  std::vector<long long> h(NB * 3);
  for (int rd16 : {0, 20})
   for (int atom = 0; atom < 4; ++atom)
    for (int mode = 0; mode < 1; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e9f, sum = 0.f;
      const int reps = 20;
      for (int rep = 0; rep < reps + 3; ++rep) {
        hipMemsetAsync(am, 0, 64 * 64 * 4, 0);
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(wr<0>, dim3(NB), dim3(256), 0, 0, in, y, rd16, st, C, am, atom);
        else hipLaunchKernelGGL(wr<1>, dim3(NB), dim3(256), 0, 0, in, y, rd16, st, C, am, atom);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3) { best = std::min(best, ms); sum += ms; }
      }
      hipMemcpy(h.data(), st, NB * 3 * 8, hipMemcpyDeviceToHost);
      std::vector<double> issue(NB), ack(NB);
      long long tmin = h[0], tmax = 0;
      for (int b = 0; b < NB; ++b) { issue[b] = (h[b * 3 + 1] - h[b * 3]) * 0.01; ack[b] = (h[b * 3 + 2] - h[b * 3]) * 0.01; tmin = std::min(tmin, h[b * 3]); tmax = std::max(tmax, h[b * 3 + 2]); }
      std::sort(issue.begin(), issue.end()); std::sort(ack.begin(), ack.end());
      printf("atomics %d, read %3d KB/block, %s stores: launch %.1f us avg (%.1f best); per block issue p50 %.2f p90 %.2f us, first store -> ack p50 %.2f p90 %.2f max %.2f us; first store .. last ack of the launch %.1f us\n",
             atom, rd16 * 4, mode ? "16-byte" : "dword  ", sum / reps * 1e3, best * 1e3, issue[NB / 2], issue[NB * 9 / 10], ack[NB / 2], ack[NB * 9 / 10], ack[NB - 1], (tmax - tmin) * 0.01);
    }
  return 0;
}
