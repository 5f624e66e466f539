// v4: 256 threads, 4 waves (2 x 2), wave tile 128 x 128, two prefetch register sets.
template <int MODE>
__global__ __launch_bounds__(256, 1) void corr_v4_kernel(const float* ft, long tsn, long tsh, long tsw,
                                                         const float* fr, long rsn, long rsh, long rsw, int W,
                                                         int C, const float* vt, const float* vr, int R,
                                                         float* out) {
  constexpr int P = 256, CKC = 32, LD = CKC + 8, NS = 2, NI = 8;
  __shared__ __attribute__((aligned(16))) __bf16 sT[NS][P * LD];
  __shared__ __attribute__((aligned(16))) __bf16 sR[NS][P * LD];
  __shared__ float nT[P], nR[P];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int pair = xcd_remap(blockIdx.x, gridDim.x);
  const int clip = pair / R;
  const float* tb = ft + (long)clip * tsn;
  const float* rb = fr + (long)pair * rsn;
  const int col4 = tid & 7, row0 = tid >> 3;
  int toff[NI], roff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int pix = row0 + 32 * i, y = pix / W, x = pix - y * W;
    toff[i] = (int)((long)y * tsh + (long)x * tsw) + col4 * 4;
    roff[i] = (int)((long)y * rsh + (long)x * rsw) + col4 * 4;
  }
  f32x4 rt[2][NI], rr[2][NI];
  float qt[NI], qr[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) qt[i] = qr[i] = 0.f;
  auto load = [&](f32x4 (&t)[NI], f32x4 (&r)[NI], int c0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) t[i] = *reinterpret_cast<const f32x4*>(tb + toff[i] + c0);
#pragma unroll
    for (int i = 0; i < NI; ++i) r[i] = *reinterpret_cast<const f32x4*>(rb + roff[i] + c0);
  };
  auto store = [&](const f32x4 (&t)[NI], const f32x4 (&r)[NI]) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      qt[i] += t[i][0] * t[i][0] + t[i][1] * t[i][1] + t[i][2] * t[i][2] + t[i][3] * t[i][3];
      bf16x4 p_[NS];
      split4<NS>(t[i], p_);
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&sT[s][(row0 + 32 * i) * LD + col4 * 4]) = p_[s];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      qr[i] += r[i][0] * r[i][0] + r[i][1] * r[i][1] + r[i][2] * r[i][2] + r[i][3] * r[i][3];
      bf16x4 p_[NS];
      split4<NS>(r[i], p_);
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&sR[s][(row0 + 32 * i) * LD + col4 * 4]) = p_[s];
    }
  };
  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int ao = (wm * 128 + li) * LD + 8 * lh, bo = (wn * 128 + li) * LD + 8 * lh;
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < CKC / 16; ++ks) {
      bf16x8 fa[4][NS];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int s = 0; s < NS; ++s) fa[a][s] = *reinterpret_cast<const bf16x8*>(&sT[s][ao + a * 32 * LD + ks * 16]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        bf16x8 fb[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) fb[s] = *reinterpret_cast<const bf16x8*>(&sR[s][bo + b * 32 * LD + ks * 16]);
#pragma unroll
        for (int a = 0; a < 4; ++a) mfma_split<NS>(fa[a], fb, acc[a][b]);
      }
    }
  };
  const int nchunk = C / CKC;   // even
  int cc = (2 * clip) % nchunk;
  auto next = [&]() { const int c = cc * CKC; cc = cc + 1 == nchunk ? 0 : cc + 1; return c; };
  load(rt[0], rr[0], next());
  load(rt[1], rr[1], next());
  for (int ch = 0; ch < nchunk - 2; ch += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
      store(rt[h], rr[h]);
      if (MODE != 2) load(rt[h], rr[h], next());
      __syncthreads();
      if (MODE != 1) compute();
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
    store(rt[h], rr[h]);
    __syncthreads();
    if (MODE != 1) compute();
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float a = qt[i], b = qr[i];
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if (col4 == 0) {
      const int pix = row0 + 32 * i;
      const float v1 = vt ? vt[(long)clip * P + pix] : 1.f, v2 = vr ? vr[(long)pair * P + pix] : 1.f;
      nT[pix] = v1 / (fabsf(v1) * sqrtf(a) + 1e-9f);
      nR[pix] = v2 / (fabsf(v2) * sqrtf(b) + 1e-9f);
    }
  }
  __syncthreads();
  float* ob = out + (long)pair * P * P;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pt = wm * 128 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float f1 = nT[pt];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int pr = wn * 128 + b * 32 + li;
        if (MODE != 3 || acc[a][b][r] == 12345.f) ob[(long)pt * P + pr] = acc[a][b][r] * f1 * nR[pr];
      }
    }
}
