// (was in master_thesis_amd/csrc/conv_halo_wino.hip behind MT355_WINO16X during round 3; a tie with igemm_wino16_kernel:
// 1.598 vs 1.586 ms (128 -> 128 at 256^2), 4.80 vs 4.91 ms (256 -> 256, 8192 x 16^2), DFPN step 107.8 vs 108.4 ms)
// ---- 16 x 16 tiles, 128 channels, the XI axis split over the waves (round 3) -----------------------------------
// igemm_wino16_kernel is balanced on two operand paths at once: per MFMA triple its 64 x 32 x
// (four xi) wave tile reads 2/3 KB of A fragments from LDS and 1/3 KB of B fragments from L1 -- 67 % and 65 % of what
// the two paths deliver -- and no wave tile with all four xi is larger in both directions at 128 accumulator
// registers.  Giving every wave ONE xi buys a 128 x 64 tile for the same registers: waves = 4 xi x 2 channel halves,
// each owns all 128 (row, pair) GEMM rows of the 16 x 16 tile x 64 channels, every A fragment pair feeds two MFMA
// triples and every B pair four -- half the LDS reads AND half the fragment loads per MFMA -- and a wave walks 3 of the
// 12 taps.  The price is the epilogue: the four xi of a channel half meet through LDS (two rounds of 128 KB: the
// images are dead by then), after which wave xi finishes output row 4a + xi of every M tile a.
template <int FMT>
__global__ __launch_bounds__(512, 2) void igemm_wino16x_kernel(ConvK p, int tiles_y, int tiles_x) {
  constexpr int NS = 2, CK = 32, TH = 16, TW = 16, BN = 128, TM = 4, TN = 2;
  constexpr int NTX = TW / 2, ROWS = TH + 2;
  constexpr int ENT = CK, ROW_E = 4 * NTX * ENT + 8, PLANE_E = ROWS * ROW_E, BUF_E = NS * PLANE_E;
  constexpr int NIT = ROWS * NTX * (CK / 4), RI = (NIT + 511) / 512;      // 1152 items, 3 rounds
  constexpr int KSTEPS = CK / 16;
  static_assert(RI == 3, "three transform rounds per chunk, one per tap of a wave");
  static_assert(2 * 4 * 2 * TN * 16 * 64 * 4 <= 2 * BUF_E * 2, "one exchange round must fit the two images");
  __shared__ __attribute__((aligned(16))) __bf16 sV[2 * BUF_E];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xi = wave & 3, nh = wave >> 2;               // this wave's xi, channel half
  const int li = lane & 31, lh = lane >> 5;
  // 1-D grid over (pixel tile, n-tile), as igemm_wino16_kernel
  const int ntn = (p.Co + BN - 1) / BN, npt = p.N * tiles_y * tiles_x;
  const int lid = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  int bid, nty;
  {
    const int q8 = npt >> 3, r8 = npt & 7;
    const int mine = q8 + (xcd < r8 ? 1 : 0), start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int tl = lid / ntn;
    nty = lid - tl * ntn;
    if (tl >= mine) return;
    bid = start + tl;
  }
  const int tx_ = bid % tiles_x; bid /= tiles_x;
  const int ty_ = bid % tiles_y;
  const int n = bid / tiles_y;
  const int ty0 = ty_ * TH, tx0 = tx_ * TW, n0 = nty * BN;
  const int HL = p.H << p.ups, WL = p.W << p.ups;
  const float scA = FMT ? pow2_scale(*p.xamax) : 1.f;
  const float inv = FMT ? 1.f / (scA * pow2_scale(*p.wamax)) : 1.f;
  const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;
  const int xsw = (int)p.xsw;

  // transform items (halo row r, column pair t, channel quad c4), c4 fastest; round i owns q = tid + 512 i
  int it_row[RI], it_lx[RI], it_dst[RI];
  unsigned it_ok[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int q = tid + 512 * i;
    const int c4 = q & 7, t = (q >> 3) & 7, r = q >> 6;
    const int ly = ty0 - 1 + r;
    it_ok[i] = 0;
    it_lx[i] = tx0 - 1 + 2 * t;
    it_row[i] = (int)((long)(ly >> p.ups) * p.xsh) + 4 * c4;
    it_dst[i] = r * ROW_E + t * ENT + 4 * c4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int lx = it_lx[i] + b;
      const bool ok = q < NIT && (unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL;
      it_ok[i] |= ok ? (1u << b) : 0u;
    }
    if (q < NIT) it_ok[i] |= 16u;                       // bit 4: the item exists
  }
  auto load_item = [&](f32x4 (&d)[4], int i, int c0) {
    const int c4 = (tid + 512 * i) & 7;
    const bool cin = c0 + 4 * c4 < p.C;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (((it_ok[i] >> b) & 1u) && cin)
        v = *reinterpret_cast<const f32x4*>(xin + (it_row[i] + ((it_lx[i] + b) >> p.ups) * xsw + c0));
      d[b] = v;
    }
  };
  auto store_item = [&](const f32x4 (&d)[4], int i, int buf) {
    if (it_ok[i] & 16u) {
      const f32x4 v[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
      __bf16* dst = sV + buf * BUF_E + it_dst[i];
#pragma unroll
      for (int x4 = 0; x4 < 4; ++x4) {
        bf16x4 parts[NS];
        split4f<NS, FMT>(v[x4], scA, parts);
#pragma unroll
        for (int s = 0; s < NS; ++s)
          *reinterpret_cast<bf16x4*>(dst + s * PLANE_E + x4 * NTX * ENT) = parts[s];
      }
    }
  };
  // weight fragments: [tap 12 = ky * 4 + xi][chunk][ntile][wn 4][kstep][plane][lane] x 16 B; this wave: wn = 2 nh + tn
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK, ntile_w = (p.Co + BN - 1) / BN;
  bf16x8 fb[2][TN][NS];                                  // [slot = k-step parity][tn][plane]
  auto load_B = [&](bf16x8 (&f)[TN][NS], int ch, int ky, int ks) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const bf16x8* g = wfr + ((((long)(ky * 4 + xi) * nchunk + ch) * ntile_w + nty) * 4 + 2 * nh + tn) * (KSTEPS * NS * 64) +
                        ks * NS * 64 + lane;
#pragma unroll
      for (int s = 0; s < NS; ++s) f[tn][s] = g[s * 64];
    }
  };

  // ---- prologue: the whole first image (the accumulators are not live yet) ---------------------
  {
    f32x4 r0[4], r1[4], r2[4];
    load_item(r0, 0, 0);
    load_item(r1, 1, 0);
    load_item(r2, 2, 0);
    load_B(fb[0], 0, 0, 0);
    load_B(fb[1], 0, 0, 1);
    store_item(r0, 0, 0);
    store_item(r1, 1, 0);
    store_item(r2, 2, 0);
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][tn][r] = 0.f;
  // MFMA row i of M tile a = (output row 4a + (i >> 3), column pair i & 7); this wave's xi plane of the image
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = (4 * a + (li >> 3)) * ROW_E + (li & 7) * ENT + 8 * lh + xi * NTX * ENT;
  __syncthreads();

  f32x4 rh[4];
  for (int ch = 0; ch < nchunk; ++ch) {
    const bool more = ch + 1 < nchunk;
    const __bf16* vb = sV + (ch & 1) * BUF_E;
    const int nb = (ch & 1) ^ 1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      // the next chunk's image: round ky is loaded at the start of tap ky, transformed and written at its end
      if (more) load_item(rh, ky, (ch + 1) * CK);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        bf16x8 fa[TM][NS];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int s = 0; s < NS; ++s)
            fa[a][s] = *reinterpret_cast<const bf16x8*>(&vb[s * PLANE_E + a_base[a] + ky * ROW_E + ks * 16]);
        __builtin_amdgcn_sched_barrier(0);     // (the eight fragment reads of the k-step are requested ahead of its 24 MFMAs)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int a = 0; a < TM; ++a) mfma_splitf<NS, FMT>(fa[a], fb[ks][tn], acc[a][tn]);
        // this slot's next contents: the same k-step of the next tap
        if (ky < 2) load_B(fb[ks], ch, ky + 1, ks);
        else if (more) load_B(fb[ks], ch + 1, 0, ks);
      }
      if (more) store_item(rh, ky, nb);
    }
    if (more) __syncthreads();      // the other image is complete, nobody reads this one any more
  }

  // ---- epilogue: the four xi of a channel half meet through LDS, two M tiles per round --------------------------
  float* xch = reinterpret_cast<float*>(sV);       // [nh 2][xi 4][a2 2][tn 2][16][64 lanes]
  float vmax = 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();                               // the images (h == 0) / the previous round's products are dead
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          xch[((((nh * 4 + xi) * 2 + a2) * TN + tn) * 16 + r) * 64 + lane] = acc[2 * h + a2][tn][r];
    __syncthreads();
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        float vals[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 4 * xi + q;                // C/D rows (r & 3) + 8 (r >> 2) + 4 lh: output row r >> 2 = xi of the M tile
          float m[4];
#pragma unroll
          for (int x4 = 0; x4 < 4; ++x4) m[x4] = xch[((((nh * 4 + x4) * 2 + a2) * TN + tn) * 16 + r) * 64 + lane];
          vals[2 * q] = (m[0] + m[1] + m[2]) * inv;
          vals[2 * q + 1] = (m[1] - m[2] - m[3]) * inv;
        }
        const int oy_ = ty0 + 4 * (2 * h + a2) + xi, oxb = tx0 + 8 * lh;
        vmax = fmaxf(vmax, halo_epilogue<8>(p, n, n0 + (2 * nh + tn) * 32 + li, vals, [&](int e, int& oy, int& ox) {
          oy = oy_;
          ox = oxb + 2 * (e >> 1) + (e & 1);       // column pair (r & 3) + 4 lh
        }));
      }
  }
  if (p.yamax) raise_amax(p.yamax, vmax);
}

