// (was in master_thesis_amd/csrc/conv_halo.hip, behind MT355_HALO_RB32=1; removed from the library in round 3)
// ---- register-B variant for <= 32 output channels (RRDBNet's growth convs and the 32-channel windows of
// the gather-form dense-block backward: model_chn.py:602-622) ------------------------------------------
// With one 32-wide channel tile every wave of a block wants the SAME weight fragments, so the waves
// split the pixels instead: a block owns 16 x 16 pixels, a wave 4 rows of them (two 32-pixel MFMA tiles:
// 12 MFMAs per tap instead of the 6 of igemm_halo_kernel<..,4,1,1,1>), and the fragments come from
// global memory in MFMA operand order -- one L2 fetch per CU, the other 11 waves of the CU's three
// blocks hit L1 (every pixel tile of a layer uses the same weights).  No per-tap barrier: the halo
// image is single-buffered (51.8 KB: three blocks per CU), two barriers per 32-channel K chunk.
template <int FMT>
__global__ __launch_bounds__(256, 3) void igemm_halo_rb32_kernel(ConvK p, int tiles_y, int tiles_x) {
  constexpr int KS = 3, NS = 2, CK = 32, LD = CK + 8, TH = 16, TW = 16, TM = 2;
  constexpr int HW_ = TW + KS - 1, HH = TH + KS - 1, HP = HH * HW_;
  constexpr int F4 = CK / 4, NH4 = HP * F4, RH = (NH4 + 255) / 256;
  constexpr int T = KS * KS, KSTEPS = CK / 16;
  constexpr int HALO_E = NS * HP * LD;
  __shared__ __attribute__((aligned(16))) __bf16 sH[HALO_E];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int n = bid / tiles_y;
  const int ty0 = ty * TH, tx0 = tx * TW, n0 = blockIdx.y * 32;
  const int HL = p.H << p.ups, WL = p.W << p.ups;
  const float scA = FMT ? pow2_scale(*p.xamax) : 1.f;
  const float inv = FMT ? 1.f / (scA * pow2_scale(*p.wamax)) : 1.f;
  const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;

  // (staging coordinates are re-derived per load: three waves per SIMD leave no registers for tables)
  const int c4s = tid % F4, hp0 = tid / F4;
  f32x4 rh[RH];
  auto load_halo = [&](int c0) {
    const bool cin = c0 + 4 * c4s < p.C;
#pragma unroll
    for (int i = 0; i < RH; ++i) {
      const int hp = hp0 + (256 / F4) * i;
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int ly = ty0 - p.pad + hy, lx = tx0 - p.pad + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (hp < HP && (unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL && cin)
        v = *reinterpret_cast<const f32x4*>(xin + ((long)(ly >> p.ups) * p.xsh + (long)(lx >> p.ups) * p.xsw + 4 * c4s + c0));
      rh[i] = v;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < RH; ++i) {
      const int idx = tid + 256 * i;
      if (idx < NH4) {
        const int hp = idx / F4, c4 = idx % F4;
        bf16x4 parts[NS];
        split4f<NS, FMT>(rh[i], scA, parts);
#pragma unroll
        for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x4*>(&sH[s * HP * LD + hp * LD + 4 * c4]) = parts[s];
      }
    }
  };
  // weight fragments: [tap][chunk][ntile][kstep][plane][lane] x 16 B (one 32-channel tile per ntile)
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK, ntile_w = (p.Co + 31) / 32;
  bf16x8 fb[KSTEPS][NS];
  auto load_B = [&](int ch, int tap, int ks) {
    const bf16x8* f = wfr + (((long)tap * nchunk + ch) * ntile_w + blockIdx.y) * (KSTEPS * NS * 64) + ks * NS * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) fb[ks][s] = f[s * 64];
  };

  f32x16 acc[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  const int apx = li < 16 ? li : ((li - HW_) & 15);
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = ((2 * (wm * TM + a) + (li >> 4)) * HW_ + apx) * LD + 8 * lh;

  load_halo(0);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) load_B(0, 0, ks);
  store_halo();
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const bool more = ch + 1 < nchunk;
    int kh = 0, kw = 0;
    for (int tap = 0; tap < T; ++tap) {
      const bool last = tap == T - 1;
      const int toff = (kh * HW_ + kw) * LD;
      // (fragments of one k-step at a time: three waves per SIMD cover the LDS latency, and the registers
      // of a second fragment set are what keeps the third wave resident)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        bf16x8 fa[TM][NS];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int s = 0; s < NS; ++s)
            fa[a][s] = *reinterpret_cast<const bf16x8*>(&sH[s * HP * LD + a_base[a] + toff + ks * 16]);
#pragma unroll
        for (int a = 0; a < TM; ++a) mfma_splitf<NS, FMT>(fa[a], fb[ks], acc[a]);
        if (!last) load_B(ch, tap + 1, ks);
        else if (more) load_B(ch + 1, 0, ks);
      }
      if (tap == 2 && more) load_halo((ch + 1) * CK);        // in flight under the remaining taps
      if (++kw == KS) { kw = 0; ++kh; }
    }
    if (more) {
      __syncthreads();            // every wave is done reading this chunk's image
      store_halo();
      __syncthreads();
    }
  }

  float vmax = 0.f;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    float vals[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vals[r] = acc[a][r] * inv;
    vmax = fmaxf(vmax, halo_epilogue_lean<16>(p, n, n0 + li, vals, [&](int r, int& oy, int& ox) {
      const int row = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int i32 = row & 31;
      oy = ty0 + (row >> 4);
      ox = tx0 + (i32 < 16 ? i32 : ((i32 - HW_) & 15));
    }));
  }
  if (p.yamax) raise_amax(p.yamax, vmax);
}

// 3 x 3 layers with more than 32 input and at most 32 output channels.  OFF by default (MT355_HALO_RB32=1
// enables it): in the CHN step it LOSES to the LDS-staged kernel at the 64 x 64 trunk of config 3 (28.4 vs
// 23.3 ms per step over the 485 launches: 512 blocks fill only two of the three block slots per CU, and two
// waves per SIMD do not cover the fragment fetches) and ties at the 128 x 128 trunk of config 5 (60.3 vs 61.3).
bool halo_rb32_wanted(int cin_p, int cout_p, int T) {
  static const bool on = [] { const char* e = getenv("MT355_HALO_RB32"); return e && e[0] == '1'; }();
  return on && T == 9 && cin_p > 32 && cout_p <= 32;
}

