// v5: 512 threads = two groups of 4 waves (one wave of each group per SIMD).  While one group
// runs the MFMAs of chunk k, the other splits + stores its share of chunk k+1 into the second
// LDS image; then the roles swap.  Two prefetch register sets keep loads in flight all the time.
template <int MODE, int PRIO = 0>
__global__ __launch_bounds__(512, 1) void corr_v7_kernel(const float* ft, long tsn, long tsh, long tsw,
                                                         const float* fr, long rsn, long rsh, long rsw, int W,
                                                         int C, const float* vt, const float* vr, int R,
                                                         float* out) {
  constexpr int P = 256, CKC = 32, LD = CKC + 8, NS = 2, NI = 4, PL = P * LD;
  // [buf][T|R][plane][P*LD]  = 2*2*2*20480 B = 163840 B (all of the CU's LDS)
  __shared__ __attribute__((aligned(16))) __bf16 sm[2 * 2 * NS * PL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wm = wave & 3, wn = g, li = lane & 31, lh = lane >> 5;
  const int pair = xcd_remap(blockIdx.x, gridDim.x);
  const int clip = pair / R;
  // buffer descriptors (wave-uniform) of the two feature slabs: loads use a 32-bit lane byte
  // offset + a scalar chunk offset, so no 64-bit address registers are held per load
  const int H = P / W;
  const unsigned tbytes = (unsigned)(((long)(H - 1) * tsh + (long)(W - 1) * tsw + C) * 4);
  const unsigned rbytes = (unsigned)(((long)(H - 1) * rsh + (long)(W - 1) * rsw + C) * 4);
  const auto srT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ft + (long)clip * tsn), 0, tbytes, 0x00020000);
  const auto srR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fr + (long)pair * rsn), 0, rbytes, 0x00020000);
  const int col4 = tid & 7, row0 = tid >> 3;
  unsigned toff[NI], roff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int pix = row0 + 64 * i, y = pix / W, x = pix - y * W;
    toff[i] = ((unsigned)((long)y * tsh + (long)x * tsw) + col4 * 4) * 4u;   // bytes
    roff[i] = ((unsigned)((long)y * rsh + (long)x * rsw) + col4 * 4) * 4u;
  }
  f32x4 rt[2][NI], rr[2][NI];
  float qt[NI], qr[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) qt[i] = qr[i] = 0.f;
  auto load = [&](f32x4 (&t)[NI], f32x4 (&r)[NI], int c0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) t[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srT, toff[i], c0 * 4, 0));
#pragma unroll
    for (int i = 0; i < NI; ++i) r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srR, roff[i], c0 * 4, 0));
  };
  auto store = [&](const f32x4 (&t)[NI], const f32x4 (&r)[NI], int buf) {
    __bf16* bT = sm + buf * (2 * NS * PL);
    __bf16* bR = bT + NS * PL;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      qt[i] += t[i][0] * t[i][0] + t[i][1] * t[i][1] + t[i][2] * t[i][2] + t[i][3] * t[i][3];
      bf16x4 p_[NS];
      split4<NS>(t[i], p_);
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&bT[s * PL + (row0 + 64 * i) * LD + col4 * 4]) = p_[s];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      qr[i] += r[i][0] * r[i][0] + r[i][1] * r[i][1] + r[i][2] * r[i][2] + r[i][3] * r[i][3];
      bf16x4 p_[NS];
      split4<NS>(r[i], p_);
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&bR[s * PL + (row0 + 64 * i) * LD + col4 * 4]) = p_[s];
    }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int ao = (wm * 64 + li) * LD + 8 * lh, bo = (wn * 128 + li) * LD + 8 * lh;
  auto compute = [&](int buf) {
    const __bf16* bT = sm + buf * (2 * NS * PL);
    const __bf16* bR = bT + NS * PL;
#pragma unroll
    for (int ks = 0; ks < CKC / 16; ++ks) {
      bf16x8 fa[2][NS];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int s = 0; s < NS; ++s) fa[a][s] = *reinterpret_cast<const bf16x8*>(&bT[s * PL + ao + a * 32 * LD + ks * 16]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        bf16x8 fb[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) fb[s] = *reinterpret_cast<const bf16x8*>(&bR[s * PL + bo + b * 32 * LD + ks * 16]);
#pragma unroll
        for (int a = 0; a < 2; ++a) mfma_split<NS>(fa[a], fb, acc[a][b]);
      }
    }
  };
  const int nchunk = C / CKC;   // even, >= 4
  const int c_start = (2 * clip) % nchunk;
  // channel offset of chunk c (rotated start); past the last chunk: beyond the descriptor's
  // range, where a buffer load returns zeros without touching memory
  auto choff = [&](int c) { int q = c + c_start; q = q >= nchunk ? q - nchunk : q; return c < nchunk ? q * CKC : (1 << 28); };
  // One barrier per chunk.  In round k every wave runs the MFMAs of chunk k (image k&1) and
  // stores its share of chunk k+1 (register set (k+1)&1) into the other image: group 1 BEFORE
  // its MFMAs, group 0 AFTER them, so on each SIMD one wave's split runs under the other's MFMAs.
  load(rt[0], rr[0], choff(0));
  load(rt[1], rr[1], choff(1));
  store(rt[0], rr[0], 0);
  load(rt[0], rr[0], choff(2));
  __syncthreads();
  for (int k = 0; k < nchunk; k += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool st = k + h + 1 < nchunk && MODE != 4;
      if (g == 1) {
        if (st) store(rt[h ^ 1], rr[h ^ 1], h ^ 1);
        if (MODE != 2) load(rt[h ^ 1], rr[h ^ 1], choff(k + h + 3));
      }
      if (MODE != 1) compute(h);
      if (g == 0) {
        if (st) store(rt[h ^ 1], rr[h ^ 1], h ^ 1);
        if (MODE != 2) load(rt[h ^ 1], rr[h ^ 1], choff(k + h + 3));
      }
      __syncthreads();
    }
  }
  float* nT = reinterpret_cast<float*>(sm);
  float* nR = nT + P;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float a = qt[i], b = qr[i];
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if (col4 == 0) {
      const int pix = row0 + 64 * i;
      const float v1 = vt ? vt[(long)clip * P + pix] : 1.f, v2 = vr ? vr[(long)pair * P + pix] : 1.f;
      nT[pix] = v1 / (fabsf(v1) * sqrtf(a) + 1e-9f);
      nR[pix] = v2 / (fabsf(v2) * sqrtf(b) + 1e-9f);
    }
  }
  __syncthreads();
  float* ob = out + (long)pair * P * P;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pt = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float f1 = nT[pt];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int pr = wn * 128 + b * 32 + li;
        if (MODE != 3 || acc[a][b][r] == 12345.f) ob[(long)pt * P + pr] = acc[a][b][r] * f1 * nR[pr];
      }
    }
}
