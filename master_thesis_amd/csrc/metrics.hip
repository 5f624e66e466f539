// The kernels either side of the training step:
//   * mask dilation + masked fill of the dataset (utils.py:618-640 TransformsUtils.dilate,
//     dataset.py:158-169): cv2.dilate with a MORPH_CROSS element, `iterations` times, then
//     x = (1 - m) * y + m * fill -- one fused pass per 32 x 32 tile, the iterations run in LDS;
//   * validation metrics (utils.py:407-448 MeasuresUtils.psnr / ssim, consumed at
//     model_chn.py:408-422): skimage 0.18.3's peak_signal_noise_ratio / structural_similarity
//     restated as per-image reductions that stay on the device.
// All three are HBM-bound streaming kernels: one read of the operands, one write / a few atomics.
#include <float.h>
#include "common.h"

namespace {

constexpr int DT = 32;           // output tile edge of the dilation kernel
constexpr int DMAX_HALO = 16;    // iterations * radius the LDS tile can hold per side

// m, m_out: [NP][H][W] planes (NP = B * F for a (B,1,F,H,W) tensor); y, x: (B,3,F,H,W) or null.
__global__ __launch_bounds__(256) void dilate_fill_kernel(const float* __restrict__ m, const float* __restrict__ y,
                                                          const float* __restrict__ fill, int F, int H, int W, int ry,
                                                          int rx, int iters, float* __restrict__ m_out,
                                                          float* __restrict__ x) {
  __shared__ float sm[2][(DT + 2 * DMAX_HALO) * (DT + 2 * DMAX_HALO)];
  const int hy = iters * ry, hx = iters * rx;
  const int th = DT + 2 * hy, tw = DT + 2 * hx;
  const int plane = blockIdx.z, ty0 = blockIdx.y * DT, tx0 = blockIdx.x * DT;
  const float* mp = m + (long)plane * H * W;
  // cv2.dilate's default border (BORDER_CONSTANT, morphologyDefaultBorderValue) never wins a maximum
  for (int i = threadIdx.x; i < th * tw; i += 256) {
    const int r = i / tw, c = i - r * tw;
    const int gy = ty0 - hy + r, gx = tx0 - hx + c;
    sm[0][i] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? mp[(long)gy * W + gx] : -FLT_MAX;
  }
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    // after iteration `it` the ring of width (it + 1) * r around the tile is no longer exact: it is
    // never read by the iterations that remain for the DT x DT centre
    const int oy = (it + 1) * ry, ox = (it + 1) * rx;
    const int vh = th - 2 * oy, vw = tw - 2 * ox;
    for (int i = threadIdx.x; i < vh * vw; i += 256) {
      const int r = oy + i / vw, c = ox + i % vw;
      float v = sm[cur][r * tw + c];
      for (int d = 1; d <= ry; ++d) v = fmaxf(v, fmaxf(sm[cur][(r - d) * tw + c], sm[cur][(r + d) * tw + c]));
      for (int d = 1; d <= rx; ++d) v = fmaxf(v, fmaxf(sm[cur][r * tw + c - d], sm[cur][r * tw + c + d]));
      // a pixel outside the image stays "border": the next iteration must not see a value there
      const int gy = ty0 - hy + r, gx = tx0 - hx + c;
      sm[cur ^ 1][r * tw + c] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? v : -FLT_MAX;
    }
    __syncthreads();
    cur ^= 1;
  }
  const int b = plane / F, f = plane - b * F;
  const float f0 = fill ? fill[0] : 0.f, f1 = fill ? fill[1] : 0.f, f2 = fill ? fill[2] : 0.f;
  for (int i = threadIdx.x; i < DT * DT; i += 256) {
    const int r = i / DT, c = i - r * DT;
    const int gy = ty0 + r, gx = tx0 + c;
    if (gy >= H || gx >= W) continue;
    const float v = sm[cur][(hy + r) * tw + hx + c];
    const long o = (long)gy * W + gx;
    m_out[(long)plane * H * W + o] = v;
    if (x) {
      const long hw = (long)H * W;
      const long base = ((long)b * 3 * F + f) * hw + o;            // channel 0 of (b, :, f)
      x[base] = (1.f - v) * y[base] + v * f0;
      x[base + F * hw] = (1.f - v) * y[base + F * hw] + v * f1;
      x[base + 2 * F * hw] = (1.f - v) * y[base + 2 * F * hw] + v * f2;
    }
  }
}

// out[n] = {sum (a-b)^2, min(b), max(b)} per image
__global__ void psnr_init_kernel(double* out, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) { out[3 * n] = 0.0; out[3 * n + 1] = DBL_MAX; out[3 * n + 2] = -DBL_MAX; }
}

__global__ __launch_bounds__(256) void psnr_sums_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       long per, double* out) {
  const int n = blockIdx.y;
  const float* pa = a + (long)n * per;
  const float* pb = b + (long)n * per;
  double s = 0.0;
  float lo = FLT_MAX, hi = -FLT_MAX;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < per; i += stride) {
    // float32 difference and square, float64 mean: skimage's _as_floats keeps float32 inputs,
    // mean_squared_error is np.mean((a - b) ** 2, dtype=np.float64)
    const float d = pa[i] - pb[i];
    s += (double)(d * d);
    lo = fminf(lo, pb[i]);
    hi = fmaxf(hi, pb[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  __shared__ double ss[4];
  __shared__ float sl[4], sh[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { ss[wave] = s; sl[wave] = lo; sh[wave] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) { s += ss[w]; lo = fminf(lo, sl[w]); hi = fmaxf(hi, sh[w]); }
    atomicAdd(&out[3 * n], s);
    atomicMin(&out[3 * n + 1], (double)lo);
    atomicMax(&out[3 * n + 2], (double)hi);
  }
}

constexpr int ST = 32;           // SSIM output tile edge
constexpr int SWMAX = 11;        // largest window

// out[plane] += sum over the tile's interior pixels of the SSIM map S (skimage 0.18.3
// structural_similarity, uniform 7 x 7 window by default, sample covariance, float64 arithmetic)
__global__ __launch_bounds__(256) void ssim_sums_kernel(const float* __restrict__ a, const float* __restrict__ b, int H,
                                                       int W, int win, double C1, double C2, double* out) {
  constexpr int TWM = ST + SWMAX - 1;
  __shared__ float ta[TWM * TWM], tb[TWM * TWM];
  __shared__ double hs[5][TWM * ST];
  const int pad = (win - 1) / 2, tw = ST + win - 1;
  const int plane = blockIdx.z;
  const int oy0 = pad + blockIdx.y * ST, ox0 = pad + blockIdx.x * ST;      // first output pixel of the tile
  const float* pa = a + (long)plane * H * W;
  const float* pb = b + (long)plane * H * W;
  for (int i = threadIdx.x; i < tw * tw; i += 256) {
    const int r = i / tw, c = i - r * tw;
    const int gy = oy0 - pad + r, gx = ox0 - pad + c;
    const bool ok = gy < H && gx < W;
    ta[i] = ok ? pa[(long)gy * W + gx] : 0.f;
    tb[i] = ok ? pb[(long)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tw * ST; i += 256) {          // horizontal window sums
    const int r = i / ST, c = i - r * ST;
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int k = 0; k < win; ++k) {
      const double x = ta[r * tw + c + k], y = tb[r * tw + c + k];
      sx += x; sy += y; sxx += x * x; syy += y * y; sxy += x * y;
    }
    hs[0][i] = sx; hs[1][i] = sy; hs[2][i] = sxx; hs[3][i] = syy; hs[4][i] = sxy;
  }
  __syncthreads();
  const double np_ = (double)win * win, cov_norm = np_ / (np_ - 1.0);
  double acc = 0.0;
  for (int i = threadIdx.x; i < ST * ST; i += 256) {
    const int r = i / ST, c = i - r * ST;
    if (oy0 + r >= H - pad || ox0 + c >= W - pad) continue;
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int k = 0; k < win; ++k) {
      const int j = (r + k) * ST + c;
      sx += hs[0][j]; sy += hs[1][j]; sxx += hs[2][j]; syy += hs[3][j]; sxy += hs[4][j];
    }
    const double ux = sx / np_, uy = sy / np_, uxx = sxx / np_, uyy = syy / np_, uxy = sxy / np_;
    const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
    const double A1 = 2 * ux * uy + C1, A2 = 2 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
    acc += (A1 * A2) / (B1 * B2);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&out[plane], red[0] + red[1] + red[2] + red[3]);
}

__global__ void zero_f64_kernel(double* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.0;
}

}  // namespace

extern "C" int mt_mask_dilate_fill(const float* m, const float* y, const float* fill3, int32_t B, int32_t F, int32_t H,
                                   int32_t W, int32_t kh, int32_t kw, int32_t iterations, float* m_out, float* x,
                                   void* stream) {
  MT_REQUIRE(m && m_out && m != m_out && B > 0 && F > 0 && H > 0 && W > 0, "mt_mask_dilate_fill: bad arguments");
  MT_REQUIRE((y == nullptr) == (x == nullptr) && (x == nullptr || fill3 != nullptr),
             "mt_mask_dilate_fill: y, x and fill3 go together");
  MT_REQUIRE(kh >= 1 && kw >= 1 && (kh & 1) && (kw & 1) && iterations >= 0,
             "mt_mask_dilate_fill: the cross element needs odd sizes (got %d x %d)", kh, kw);
  MT_REQUIRE(iterations * (kh / 2) <= DMAX_HALO && iterations * (kw / 2) <= DMAX_HALO,
             "mt_mask_dilate_fill: iterations * radius must be <= %d", DMAX_HALO);
  MT_REQUIRE((long)B * F < 65536, "mt_mask_dilate_fill: too many planes");
  hipLaunchKernelGGL(dilate_fill_kernel, dim3((W + DT - 1) / DT, (H + DT - 1) / DT, B * F), dim3(256), 0,
                     static_cast<hipStream_t>(stream), m, y, fill3, F, H, W, kh / 2, kw / 2, iterations, m_out, x);
  return mt_check_launch("mt_mask_dilate_fill");
}

extern "C" int mt_psnr_sums(const float* y_hat, const float* y, int32_t N, int64_t per_image, double* out,
                            void* stream) {
  MT_REQUIRE(y_hat && y && out && N > 0 && N < 65536 && per_image > 0, "mt_psnr_sums: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(psnr_init_kernel, dim3((N + 255) / 256), dim3(256), 0, s, out, N);
  hipLaunchKernelGGL(psnr_sums_kernel, dim3(mt_grid_1d(per_image, 256 * 8, 256), N), dim3(256), 0, s, y_hat, y,
                     (long)per_image, out);
  return mt_check_launch("mt_psnr_sums");
}

extern "C" int mt_ssim_sums(const float* a, const float* b, int32_t planes, int32_t H, int32_t W, int32_t win,
                            double data_range, double K1, double K2, double* out, void* stream) {
  MT_REQUIRE(a && b && out && planes > 0 && planes < 65536, "mt_ssim_sums: bad arguments");
  MT_REQUIRE(win >= 3 && win <= SWMAX && (win & 1), "mt_ssim_sums: window must be odd, 3..%d", SWMAX);
  MT_REQUIRE(H >= win && W >= win, "mt_ssim_sums: win_size exceeds image extent");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(zero_f64_kernel, dim3((planes + 255) / 256), dim3(256), 0, s, out, planes);
  const int oh = H - win + 1, ow = W - win + 1;
  const double C1 = (K1 * data_range) * (K1 * data_range);
  const double C2 = (K2 * data_range) * (K2 * data_range);
  hipLaunchKernelGGL(ssim_sums_kernel, dim3((ow + ST - 1) / ST, (oh + ST - 1) / ST, planes), dim3(256), 0, s, a, b, H, W,
                     win, C1, C2, out);
  return mt_check_launch("mt_ssim_sums");
}
