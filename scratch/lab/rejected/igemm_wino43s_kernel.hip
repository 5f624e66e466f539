// REJECTED (round 4) -- kept as source for reference, not compiled into libmt355.so.
// igemm_wino43s_kernel: igemm_wino43_kernel (csrc/conv_halo_wino43.hip) with ONE wave per SIMD: 256-thread blocks, 512
// registers per wave, both 32-channel quarters of a channel half per wave (twelve accumulator tiles in AGPRs), three taps of
// weight fragments in flight, double-buffered image fragments.  To build it: paste into conv_halo_wino43.hip in front of
// weight_pack_halo_wino43_kernel and launch it with 256 threads from mt_halo_wino43_launch (same grid, same pack).
// Correct (bit-identical: 238 conv / amax / up-sampling tests), never faster -- SeparableConv4d shape, 8192 x 16 x 16, 256 -> 256,
// one box (scratch/r4_single43.sh): shipped kernel 3.96 ms; this one 5.74 ms as written (image reads + transform, fence,
// MFMAs), 5.86 ms without the fence, 4.08 ms with a sched_group_barrier pattern (1 MFMA : 12 VALU : 2 + 2 DS : 1 VMEM);
// without the in-loop transform 3.27 ms (shipped kernel: 3.11), with neither transform nor fragment reloads 2.74 (2.45).
// What it showed: requesting the weight fragments three taps ahead instead of two does not make them cheaper (0.53 ms
// against 0.49) -- their cost is not a latency stall.  See DESIGN.md section 3, "Measured in round 4 and not adopted".
#ifndef MT_W43S_ABL
#define MT_W43S_ABL 0                 // (lab ablations: 1 no fragment reloads, 4 no in-loop transform)
#endif
#ifndef MT_W43S_SCHED
#define MT_W43S_SCHED 0               // 0: image reads + transform, fence, MFMAs; 1: no fence; 2: sched_group_barrier interleave
#endif
// ---- the same kernel with ONE wave per SIMD (round 4) ---------------------------------------------------------------------
// Why: ablations of igemm_wino43_kernel on the SeparableConv4d shape (scratch/r4_abl43.sh; 3.94 ms as shipped) -- without the
// in-loop halo loads + transform 3.11 ms, without the weight-fragment reloads 3.45 ms, without the per-tap image reads 3.81 ms,
// MFMAs only 2.45 ms.  Vector-memory loads return IN ORDER (one vmcnt): a wave that has just requested the next chunk's halo
// (HBM latency) waits that latency at the next s_waitcnt for a weight fragment (an L2 hit requested two taps ahead), and with
// 256 registers per wave there is no room to request fragments further ahead.  Here a block is 256 threads -- four waves, one
// per SIMD, 512 registers each: a wave owns BOTH 32-channel quarters of its channel half (twelve accumulator tiles in
// AGPRs), keeps THREE taps of weight fragments in flight (requested three taps = ~2300 clocks ahead: past the halo's latency)
// and double-buffers the image fragments (the next tap's LDS reads are issued before this tap's MFMAs).  Same image, same
// pack, same per-accumulator order of products: bit-identical results.
#ifndef MT_W43S_ABL
#define MT_W43S_ABL 0                 // (lab ablations of the single-wave kernel: 1 no fragment reloads, 4 no in-loop transform)
#endif
#ifndef MT_W43S_SCHED
#define MT_W43S_SCHED 0               // 0: image reads + transform, fence, MFMAs; 1: no fence; 2: sched_group_barrier interleave
#endif
#ifndef MT_W43_SINGLE
#define MT_W43_SINGLE 0               // 1: igemm_wino43s_kernel takes the launches
#endif
template <int FMT>
__global__ __launch_bounds__(256, 1) void igemm_wino43s_kernel(ConvK p, int tiles_y, int tiles_x) {
  constexpr int NS = 2, CK = 32, TH = 16, TW = 16, BN = 128, TM = 2, NQ2 = 2, NXI = 6;
  constexpr int NQ = TW / 4, ROWS = TH + 2;
  constexpr int ENT = CK, ROW_E = NXI * NQ * ENT + 8, PLANE_E = ROWS * ROW_E, BUF_E = NS * PLANE_E;
  constexpr int NIT = ROWS * NQ * (CK / 4), RI = (NIT + 255) / 256;       // 576 items: two full rounds (the tile's rows) + 64
  constexpr int TW_ = 9, KSTEPS = CK / 16, SLOTS = 3;
  constexpr int XCH_E = 4 * NQ2 * 64 * 64 * 2;                            // epilogue exchange (bf16 units): 4 waves x 2 quarters x 64 floats x 64 lanes
  constexpr int SM_E = 2 * BUF_E > XCH_E ? 2 * BUF_E : XCH_E;
  static_assert(RI == 3 && TW_ % SLOTS == 0, "three transform rounds; the fragment slot of a tap must not depend on the chunk");
  static_assert(SM_E * 2 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) __bf16 sV[SM_E];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xh = wave >> 1, wq = wave & 1;                 // xi half, channel half (quarters 2 wq, 2 wq + 1)
  const int li = lane & 31, lh = lane >> 5;
  const int ntn = (p.Co + BN - 1) / BN, npt = p.N * tiles_y * tiles_x;
  const int lid = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  int bid, nty;
  {
    const int q8 = npt >> 3, r8 = npt & 7;
    const int mine = q8 + (xcd < r8 ? 1 : 0), start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int tl = lid / ntn;
    nty = lid - tl * ntn;
    if (tl >= mine) {            // (a padding block: it still takes part in the launch's max reduction)
      finish_amax(p, 0.f, sV);
      return;
    }
    bid = start + tl;
  }
  const int tx_ = bid % tiles_x; bid /= tiles_x;
  const int ty_ = bid % tiles_y;
  const int n = bid / tiles_y;
  const int ty0 = ty_ * TH, tx0 = tx_ * TW, n0 = nty * BN;
  const int HL = p.H << p.ups, WL = p.W << p.ups;
  const float scA = pow2_scale(*p.xamax) * 0.0625f;
  const float inv = 1.f / (scA * pow2_scale(*p.wamax));
  float bv[NQ2];
#pragma unroll
  for (int nq = 0; nq < NQ2; ++nq) {
    const int o = n0 + (2 * wq + nq) * 32 + li;
    bv[nq] = (p.bias && o < p.Co) ? p.bias[o] : 0.f;
  }
  const float* xin = p.x + (long)(n / p.xni) * p.xsg + (long)(n % p.xni) * p.xsn;
  const int xsw = (int)p.xsw;

  // transform items as in igemm_wino43_kernel: rounds 0, 1 = the tile's own 16 rows, round 2 = halo rows 0 and 17
  const bool r1_live = ty0 > 0 || ty0 + TH < HL;
  int it_row[RI], it_lx[RI], it_dst[RI];
  unsigned it_ok[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int q = tid + 256 * i;
    const int c4 = q & 7, t = (q >> 3) & 3, rq = q >> 5;
    const int r = rq < TH ? rq + 1 : (rq == TH ? 0 : ROWS - 1);
    const int ly = ty0 - 1 + r;
    it_ok[i] = 0;
    it_lx[i] = tx0 - 1 + 4 * t;
    it_row[i] = (int)((long)(ly >> p.ups) * p.xsh) + 4 * c4;
    it_dst[i] = r * ROW_E + t * ENT + 4 * c4;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const int lx = it_lx[i] + b;
      const bool ok = q < NIT && (unsigned)ly < (unsigned)HL && (unsigned)lx < (unsigned)WL;
      it_ok[i] |= ok ? (1u << b) : 0u;
    }
    if (q < NIT) it_ok[i] |= 64u;
  }
  auto load_item = [&](f32x4 (&d)[6], int i, int c0) {
    const int c4 = (tid + 256 * i) & 7;
    const bool cin = c0 + 4 * c4 < p.C;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (((it_ok[i] >> b) & 1u) && cin)
        v = *reinterpret_cast<const f32x4*>(xin + (it_row[i] + ((it_lx[i] + b) >> p.ups) * xsw + c0));
      d[b] = v;
    }
  };
  auto store_item = [&](const f32x4 (&d)[6], int i, int buf) {
    if (it_ok[i] & 64u) {
      const f32x4 t0 = d[4] - 4.f * d[2], t1 = d[3] - 4.f * d[1], t2 = d[4] - d[2], t3 = d[3] - d[1];
      const f32x4 v[6] = {4.f * d[0] - 5.f * d[2] + d[4], t0 + t1, t0 - t1, t2 + 2.f * t3, t2 - 2.f * t3,
                          4.f * d[1] - 5.f * d[3] + d[5]};
      __bf16* dst = sV + buf * BUF_E + it_dst[i];
#pragma unroll
      for (int x6 = 0; x6 < NXI; ++x6) {
        bf16x4 parts[NS];
        split4f<NS, FMT, MT_W43_SPLIT>(v[x6], scA, parts);
#pragma unroll
        for (int s = 0; s < NS; ++s)
          *reinterpret_cast<bf16x4*>(dst + s * PLANE_E + x6 * NQ * ENT) = parts[s];
      }
    }
  };
  // weight fragments: [tap 18 = ky * 6 + xi][chunk][ntile][wn 4][kstep][plane][lane] x 16 B
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK, ntile_w = (p.Co + BN - 1) / BN;
  bf16x8 fb[SLOTS][NQ2][KSTEPS][NS];
  // wave tap w (0 .. 8) = kernel row w / 3, xi 3 xh + w % 3; slot = w % 3 in every chunk (9 taps, 3 slots)
  auto load_B = [&](bf16x8 (&f)[NQ2][KSTEPS][NS], int ch, int w) {
    const int tap = (w / 3) * NXI + 3 * xh + (w % 3);
#pragma unroll
    for (int nq = 0; nq < NQ2; ++nq) {
      const bf16x8* g = wfr + ((((long)tap * nchunk + ch) * ntile_w + nty) * 4 + 2 * wq + nq) * (KSTEPS * NS * 64) + lane;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int s = 0; s < NS; ++s) f[nq][ks][s] = g[(ks * NS + s) * 64];
    }
  };

  // ---- prologue: the whole first image, the fragments of the first three taps ---------------------------------------
  f32x4 ra[6], rb[6];
  load_item(ra, 0, 0);
  load_item(rb, 1, 0);
#pragma unroll
  for (int w = 0; w < SLOTS; ++w) load_B(fb[w], 0, w);
  store_item(ra, 0, 0);
  if (r1_live) load_item(ra, 2, 0);
  store_item(rb, 1, 0);
  if (r1_live) {
    store_item(ra, 2, 0);
  } else {
    constexpr int ROW8 = ROW_E / 4;
    for (int e = tid; e < 2 * NS * 2 * ROW8; e += 256) {
      const int u = e % ROW8, w_ = e / ROW8;
      const int row = (w_ & 1) ? ROWS - 1 : 0, plane = (w_ >> 1) % NS, buf = w_ / (2 * NS);
      *reinterpret_cast<bf16x4*>(sV + buf * BUF_E + plane * PLANE_E + row * ROW_E + 4 * u) = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    }
  }
  f32x16 acc[TM][NQ2][3];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int nq = 0; nq < NQ2; ++nq)
#pragma unroll
      for (int xl = 0; xl < 3; ++xl)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][nq][xl][r] = 0.f;
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = (8 * a + (li >> 2)) * ROW_E + (li & 3) * ENT + 8 * lh + 3 * xh * NQ * ENT;
  __syncthreads();

  bf16x8 fa[2][KSTEPS][TM][NS];
  auto read_A = [&](bf16x8 (&f)[KSTEPS][TM][NS], const __bf16* vb, int w) {
    const int toff = (w / 3) * ROW_E + (w % 3) * NQ * ENT;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          f[ks][a][s] = *reinterpret_cast<const bf16x8*>(&vb[s * PLANE_E + a_base[a] + toff + ks * 16]);
  };
  for (int ch = 0; ch < nchunk; ++ch) {
    const bool more = ch + 1 < nchunk;
    const __bf16* vb = sV + (ch & 1) * BUF_E;
    const int nb = (ch & 1) ^ 1;
    read_A(fa[0], vb, 0);
#pragma unroll
    for (int w = 0; w < TW_; ++w) {
      const int xl = w % 3;
      // the next tap's image fragments (the other register set), the next chunk's halo (two register sets: rounds 0 / 2
      // travel through ra, round 1 through rb), then this tap's 24 MFMAs
      if (w + 1 < TW_) read_A(fa[(w + 1) & 1], vb, w + 1);
      if (more && !(MT_W43S_ABL & 4)) {
        if (w == 0) load_item(ra, 0, (ch + 1) * CK);
        if (w == 1) load_item(rb, 1, (ch + 1) * CK);
        if (w == 3) store_item(ra, 0, nb);
        if (w == 4 && r1_live) load_item(ra, 2, (ch + 1) * CK);
        if (w == 5) store_item(rb, 1, nb);
        if (w == 8 && r1_live) store_item(ra, 2, nb);
      }
#if MT_W43S_SCHED == 0
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        // per accumulator the same order of products as mfma_splitf (lo x hi, hi x lo, hi x hi); the four accumulators of a
        // tap alternate, so that no MFMA waits for the one issued right before it
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int nq = 0; nq < NQ2; ++nq) {
              const half8 av = __builtin_bit_cast(half8, fa[w & 1][ks][a][pr == 0 ? 1 : 0]);
              const half8 bw = __builtin_bit_cast(half8, fb[w % SLOTS][nq][ks][pr == 1 ? 1 : 0]);
              acc[a][nq][xl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bw, acc[a][nq][xl], 0, 0, 0);
            }
      }
      // this slot's next contents: three taps on
#if !(MT_W43S_ABL & 1)
      if (w + SLOTS < TW_) load_B(fb[w % SLOTS], ch, w + SLOTS);
      else if (more) load_B(fb[w % SLOTS], ch + 1, w + SLOTS - TW_);
#endif
#if MT_W43S_SCHED == 2
      // one MFMA, then up to 12 VALU / 2 LDS / 1 memory instruction, 24 times: the transform runs in the MFMAs' shadow
#pragma unroll
      for (int g_ = 0; g_ < 24; ++g_) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);   // VALU
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // DS read
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);    // DS write
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // VMEM read
      }
#endif
    }
    if (more) __syncthreads();      // the other image is complete, nobody reads this one any more
  }

  // ---- epilogue: as igemm_wino43_kernel, for both channel quarters of the wave ----------------------------------------
  auto share = [&](const f32x16 (&m)[3], int r, float (&s4)[4]) {
    if (xh == 0) {
      const float e = m[1][r] + m[2][r], o = m[1][r] - m[2][r];
      s4[0] = m[0][r] + e; s4[1] = o; s4[2] = e; s4[3] = o;
    } else {
      const float e = m[0][r] + m[1][r], o = m[0][r] - m[1][r];
      s4[0] = e; s4[1] = 2.f * o; s4[2] = 4.f * e; s4[3] = 8.f * o + m[2][r];
    }
  };
  __syncthreads();
  float* xch = reinterpret_cast<float*>(sV);
#pragma unroll
  for (int nq = 0; nq < NQ2; ++nq) {
    float* mine = xch + ((wave * NQ2 + nq) * 64) * 64 + lane;            // [wave][quarter][4 shares x 16][64 lanes]
    const f32x16 (&m)[3] = acc[xh == 0 ? 1 : 0][nq];                     // the M tile the OTHER half finishes
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s4[4];
      share(m, r, s4);
#pragma unroll
      for (int j = 0; j < 4; ++j) mine[(j * 16 + r) * 64] = s4[j];
    }
  }
  __syncthreads();
  float vmax = 0.f;
#pragma unroll
  for (int nq = 0; nq < NQ2; ++nq) {
    const float* theirs = xch + (((wave ^ 2) * NQ2 + nq) * 64) * 64 + lane;
    const int o = n0 + (2 * wq + nq) * 32 + li;
    const f32x16 (&m)[3] = acc[xh == 0 ? 0 : 1][nq];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float vals[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = 8 * hh + q;
        float s4[4];
        share(m, r, s4);
#pragma unroll
        for (int j = 0; j < 4; ++j) vals[q * 4 + j] = (s4[j] + theirs[(j * 16 + r) * 64]) * inv;
      }
      const int oyb = ty0 + 8 * xh + lh;
      vmax = fmaxf(vmax, halo_epilogue<32>(p, n, o, vals, [&](int e, int& oy, int& ox) {
        const int r = 8 * hh + (e >> 2), j = e & 3;
        oy = oyb + 2 * (r >> 2);
        ox = tx0 + 4 * (r & 3) + j;
      }, bv[nq]));
    }
  }
  finish_amax(p, vmax, sV);
}

