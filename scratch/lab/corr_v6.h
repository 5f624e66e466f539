// v5: 512 threads = two groups of 4 waves (one wave of each group per SIMD).  While one group
// runs the MFMAs of chunk k, the other splits + stores its share of chunk k+1 into the second
// LDS image; then the roles swap.  Two prefetch register sets keep loads in flight all the time.
template <int MODE>
__global__ __launch_bounds__(512, 1) void corr_v6_kernel(const float* ft, long tsn, long tsh, long tsw,
                                                         const float* fr, long rsn, long rsh, long rsw, int W,
                                                         int C, const float* vt, const float* vr, int R,
                                                         float* out, int npairs) {
  constexpr int P = 256, CKC = 32, LD = CKC + 8, NS = 2, NI = 4, PL = P * LD;
  // [buf][T|R][plane][P*LD]  = 2*2*2*20480 B = 163840 B (all of the CU's LDS)
  __shared__ __attribute__((aligned(16))) __bf16 sm[2 * 2 * NS * PL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wm = wave & 3, wn = g, li = lane & 31, lh = lane >> 5;
  // persistent: block b walks pairs remap(b) + j * gridDim.x
  int pair = xcd_remap(blockIdx.x, gridDim.x);
  int clip = pair / R;
  const int H = P / W;
  const unsigned tbytes = (unsigned)(((long)(H - 1) * tsh + (long)(W - 1) * tsw + C) * 4);
  const unsigned rbytes = (unsigned)(((long)(H - 1) * rsh + (long)(W - 1) * rsw + C) * 4);
  auto srT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ft + (long)clip * tsn), 0, tbytes, 0x00020000);
  auto srR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fr + (long)pair * rsn), 0, rbytes, 0x00020000);
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
  int c_start = (2 * clip) % nchunk;
  // channel offset of chunk c (rotated start); past the last chunk: beyond the descriptor's
  // range, where a buffer load returns zeros without touching memory
  auto choff = [&](int c) { int q = c + c_start; q = q >= nchunk ? q - nchunk : q; return c < nchunk ? q * CKC : (1 << 28); };
  // Group g keeps chunk c in register set (c + g) & 1 and runs one phase behind/ahead of the
  // other group:   g0: C0 | S1 | C1 | S2 | ...      g1: S1 | C0 | S2 | C1 | ...
  // (Ck = MFMAs of chunk k from image k&1, Sk = split+store of the group's share of chunk k)
  load(rt[0], rr[0], choff(g));
  load(rt[1], rr[1], choff(1 - g));
  for (;;) {
#pragma unroll
  for (int i = 0; i < NI; ++i) qt[i] = qr[i] = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  if (g == 0) { store(rt[0], rr[0], 0); load(rt[0], rr[0], choff(2)); }
  else        { store(rt[1], rr[1], 0); load(rt[1], rr[1], choff(2)); }
  __syncthreads();
  if (g == 1) {
    store(rt[0], rr[0], 1);
    load(rt[0], rr[0], choff(3));
    __syncthreads();
  }
  for (int k = 0; k < nchunk; k += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cs = k + h + 1 + g;        // chunk this group stores in this step (register set h^1)
      if (MODE != 1) compute(h);
      __syncthreads();
      if (cs < nchunk) {
        store(rt[h ^ 1], rr[h ^ 1], (h ^ 1) ^ g);
      }
      if (MODE != 2) load(rt[h ^ 1], rr[h ^ 1], choff(cs + 2));
      __syncthreads();
    }
  }
  if (g == 0) __syncthreads();
  // the next pair's first two chunks travel while this pair's output is written
  const int cur_pair = pair, cur_clip = clip;
  pair += gridDim.x;
  const bool more = pair < npairs;
  {
    const int np_ = more ? pair : cur_pair;     // (clamped: the loads are dropped when there is no next pair)
    clip = np_ / R;
    c_start = (2 * clip) % nchunk;
    srT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ft + (long)clip * tsn), 0, tbytes, 0x00020000);
    srR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fr + (long)np_ * rsn), 0, rbytes, 0x00020000);
    load(rt[0], rr[0], more ? choff(g) : (1 << 28));
    load(rt[1], rr[1], more ? choff(1 - g) : (1 << 28));
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
      const float v1 = vt ? vt[(long)cur_clip * P + pix] : 1.f, v2 = vr ? vr[(long)cur_pair * P + pix] : 1.f;
      nT[pix] = v1 / (fabsf(v1) * sqrtf(a) + 1e-9f);
      nR[pix] = v2 / (fabsf(v2) * sqrtf(b) + 1e-9f);
    }
  }
  __syncthreads();
  // out[pair][pt][pr]: buffer stores (lane byte offset + a scalar per store, no 64-bit addresses)
  const auto srO = __builtin_amdgcn_make_buffer_rsrc(out + (long)cur_pair * P * P, 0, P * P * 4, 0x00020000);
  const int obase = (((wm * 64 + 4 * lh) * P) + wn * 128 + li) * 4;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ptl = a * 32 + (r & 3) + 8 * (r >> 2);
      const float f1 = nT[wm * 64 + 4 * lh + ptl];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float v = acc[a][b][r] * f1 * nR[wn * 128 + b * 32 + li];
        if (MODE != 3 || v == 12345.f)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), srO, obase, (ptl * P + b * 32) * 4, 0);
      }
    }
  if (!more) break;
  __syncthreads();   // nT / nR (aliased on image 0) are read before the next pair's stores
  }
}
