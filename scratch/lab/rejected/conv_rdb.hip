// (round 3: built, correct -- bit-identical to the layer-by-layer launches on every size tried -- and rejected: see
// scratch/lab/rejected/README.md.  It was exported as mt_rdb_fwd / mt_rdb_scratch_floats for one evening; the host side lived in
// engine.rdb_forward / model_chn.RRDBNetFn._trunk_forward_chains (git history: "Fused dense-block forward kernel").)
// One ResidualDenseBlock5C forward (model_chn.py:588-623: four growth convs 64+32(i-1) -> 32 with LeakyReLU, then
// conv5 192 -> 64 with out = 0.2 conv5 + x) as ONE launch (round 3; SURVEY 8b `rdb_block_fwd`).
//
// The five layers of a block are a dependency chain of ~30 us launches, each of which leaves the matrix pipe idle four
// fifths of the time (profiles/r03_pmc_chn_mfma_util.txt).  A persistent kernel with barriers between the layers needs
// every tile of an image co-resident; this kernel needs NO communication between thread blocks instead: a block owns a
// band of 8 rows x 64 columns... (all columns) of one image and RECOMPUTES the rows of its neighbours' bands that its later
// layers read -- conv5 on rows R needs y4 on R +- 1, y3 on R +- 2, y2 on R +- 3, y1 on R +- 4 -- 1.29x the MFMAs of the
// layer-by-layer form, on layers that use a tenth of the matrix pipe.  Rows inside the band go to the block buffer the
// other kernels read (the next dense block, the weight gradients, the backward); rows outside it -- valid values, but
// owned by the neighbour -- go to a private scratch strip of this block and are read back from there, so no two blocks
// ever write the same address and the result does not depend on their timing.  The fp16 scale of each layer's input
// comes from the global max|x| of the block input and a block-local running maximum of what this block has produced
// (everything a block reads beyond x it has produced itself).
//
// The per-tile work is igemm_winox_kernel's (conv_halo_wino.hip): 1-D Winograd F(2,3), the four xi on four waves, weight
// fragments from global memory (mt_weight_pack_halo, fragment order 2), single-buffered image, the xi meet in the epilogue.
#include <cstdlib>
#include "conv_common.h"

namespace {

struct RdbK {
  ConvK L[5];            // the five layers as mt_conv_run would see them (x = channel prefix of the block buffer)
  float* scratch;        // [blocks][SROWS][W][128]: y1 .. y4 on the rows outside a block's band
  int bands;             // row bands per image (H / 8)
  float* amax_out;       // the block buffer's running max|.| (raised by what the growth convs write)
};

constexpr int RDB_BAND = 8, RDB_HALO = 4, RDB_SROWS = RDB_BAND + 2 * RDB_HALO, RDB_GC = 128;

// One 8 x 16 output tile of one layer.  Rows [band0, band1) of the gathered channels >= 64 and of the output live in the
// block buffer, the others in this block's scratch strip (virtual row = row - srow0).
template <int FMT, int TN>
__device__ __forceinline__ float rdb_tile(const ConvK& p, __bf16* sV, int n, int ty0, int tx0, int band0, int band1,
                                          float* strip, int srow0, int row_lo, int row_hi, float scA, float inv,
                                          float* xch) {
  constexpr int NS = 2, CK = 32, TH = 8, TW = 16, TM = 2;
  constexpr int NTX = TW / 2, ROWS = TH + 2;
  constexpr int ENT = CK, ROW_E = 4 * NTX * ENT + 8, PLANE_E = ROWS * ROW_E;
  constexpr int NIT = ROWS * NTX * (CK / 4), RI = (NIT + 255) / 256;
  constexpr int KSTEPS = CK / 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63, xi = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int HL = p.H, WL = p.W;
  const float* xin = p.x + (long)n * p.xsn;
  const int W128 = WL * RDB_GC;

  // transform items (halo row r, column pair t, channel quad c4)
  int it_row[RI], it_srow[RI], it_lx[RI];
  unsigned it_ok[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int q = tid + 256 * i;
    const int c4 = q & 7, t = (q >> 3) & 7, r = q >> 6;
    const int ly = ty0 - 1 + r;
    it_ok[i] = 0;
    it_lx[i] = tx0 - 1 + 2 * t;
    it_row[i] = (int)((long)ly * p.xsh) + 4 * c4;
    it_srow[i] = (ly - srow0) * W128 + 4 * c4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int lx = it_lx[i] + b;
      // rows outside [row_lo - 1, row_hi + 1) feed only outputs this layer does not keep: not loaded (they may lie
      // outside the scratch strip)
      const bool ok = q < NIT && ly >= max(0, row_lo - 1) && ly < min(HL, row_hi + 1) && (unsigned)lx < (unsigned)WL;
      it_ok[i] |= ok ? (1u << b) : 0u;
    }
    if (ly >= band0 && ly < band1) it_ok[i] |= 16u;      // bit 4: the row lies in this block's band (block buffer for every channel)
  }
  const int xsw = (int)p.xsw;
  f32x4 rh[RI][4];
  auto load_raw = [&](int c0) {
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int c4 = (tid + 256 * i) & 7;
      const bool cin = c0 + 4 * c4 < p.C;
      // channels < 64 (the block input) and band rows come from the block buffer, the rest from the scratch strip
      const bool shared = c0 < 64 || (it_ok[i] & 16u);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (((it_ok[i] >> b) & 1u) && cin) {
          if (shared) v = *reinterpret_cast<const f32x4*>(xin + (it_row[i] + (it_lx[i] + b) * xsw + c0));
          else v = *reinterpret_cast<const f32x4*>(strip + (it_srow[i] + (it_lx[i] + b) * RDB_GC + (c0 - 64)));
        }
        rh[i][b] = v;
      }
    }
  };
  auto store_v = [&]() {
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int q = tid + 256 * i;
      if (q < NIT) {
        const int c4 = q & 7, t = (q >> 3) & 7, r = q >> 6;
        const f32x4 d0 = rh[i][0], d1 = rh[i][1], d2 = rh[i][2], d3 = rh[i][3];
        const f32x4 v[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
        __bf16* dst = sV + r * ROW_E + t * ENT + 4 * c4;
#pragma unroll
        for (int x4 = 0; x4 < 4; ++x4) {
          bf16x4 parts[NS];
          split4f<NS, FMT>(v[x4], scA, parts);
#pragma unroll
          for (int s = 0; s < NS; ++s)
            *reinterpret_cast<bf16x4*>(dst + s * PLANE_E + x4 * NTX * ENT) = parts[s];
        }
      }
    }
  };
  const bf16x8* wfr = reinterpret_cast<const bf16x8*>(p.wt);
  const int nchunk = (p.C + CK - 1) / CK;
  bf16x8 fb[TN][KSTEPS][NS];
  auto load_B = [&](int ch, int ky, int tn, int ks) {
    const bf16x8* f = wfr + (((long)(ky * 4 + xi) * nchunk + ch) * TN + tn) * (KSTEPS * NS * 64) + ks * NS * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) fb[tn][ks][s] = f[s * 64];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][tn][r] = 0.f;
  int a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) a_base[a] = (4 * a + (li >> 3)) * ROW_E + (li & 7) * ENT + 8 * lh + xi * NTX * ENT;

  load_raw(0);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) load_B(0, 0, tn, ks);
  __syncthreads();                    // the previous tile's epilogue is done with the LDS
  store_v();
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const bool more = ch + 1 < nchunk;
    if (more) load_raw((ch + 1) * CK);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int toff = ky * ROW_E;
      bf16x8 fa[KSTEPS][TM][NS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int s = 0; s < NS; ++s)
            fa[ks][a][s] = *reinterpret_cast<const bf16x8*>(&sV[s * PLANE_E + a_base[a] + toff + ks * 16]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
          for (int a = 0; a < TM; ++a) mfma_splitf<NS, FMT>(fa[ks][a], fb[tn][ks], acc[a][tn]);
          if (ky < 2) load_B(ch, ky + 1, tn, ks);
          else if (more) load_B(ch + 1, 0, tn, ks);
        }
      }
    }
    if (more) {
      __syncthreads();
      store_v();
      __syncthreads();
    }
  }

  // ---- epilogue: the four xi meet through LDS; wave w finishes output row 4a + w; band rows -> block buffer, the
  // others (inside [row_lo, row_hi)) -> scratch strip
  float vmax = 0.f;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[((xi * TN + tn) * 16 + r) * 64 + lane] = acc[a][tn][r];
    __syncthreads();
    const int oy_ = ty0 + 4 * a + xi;
    if (oy_ < row_lo || oy_ >= row_hi) continue;         // (wave-uniform) a row this layer does not need / outside the image
    const bool inband = oy_ >= band0 && oy_ < band1;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      float vals[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * xi + q;
        const float m0 = xch[((0 * TN + tn) * 16 + r) * 64 + lane], m1 = xch[((1 * TN + tn) * 16 + r) * 64 + lane];
        const float m2 = xch[((2 * TN + tn) * 16 + r) * 64 + lane], m3 = xch[((3 * TN + tn) * 16 + r) * 64 + lane];
        vals[2 * q] = (m0 + m1 + m2) * inv;
        vals[2 * q + 1] = (m1 - m2 - m3) * inv;
      }
      const int oxb = tx0 + 8 * lh;
      auto pix = [&](int e, int& oy, int& ox) {
        oy = oy_;
        ox = oxb + 2 * (e >> 1) + (e & 1);
      };
      const int o = tn * 32 + li;
      if (inband) {
        vmax = fmaxf(vmax, halo_epilogue_lean<8>(p, n, o, vals, pix));
      } else if (o < p.Co) {
        // the scratch strip [virtual row][x][y channel]: growth convs only (bias + activation, no residual / gate)
        const float bv = p.bias ? p.bias[o] : 0.f;
        float* dst = strip + (long)(oy_ - srow0) * W128 + (int)(p.y - (p.x + 64)) + o;      // (+ this layer's window inside the y channels)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ox = oxb + 2 * (e >> 1) + (e & 1);
          if (ox < p.Wo) {
            const float x = mt_act_fn(vals[e] + bv, p.act) * p.alpha;
            dst[ox * RDB_GC] = x;
            vmax = fmaxf(vmax, fabsf(x));
          }
        }
      }
    }
  }
  return vmax;
}

template <int FMT>
__global__ __launch_bounds__(256, 1) void rdb_fwd_kernel(RdbK k) {      // (one block per CU anyway: N x H / 8 blocks)
  constexpr int PLANE_E = 10 * (4 * 8 * 32 + 8);
  __shared__ __attribute__((aligned(16))) __bf16 sV[2 * PLANE_E];
  __shared__ float s_red[4];
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int n = bid / k.bands, band = bid - n * k.bands;
  const int band0 = band * RDB_BAND, band1 = band0 + RDB_BAND;
  const int srow0 = band0 - RDB_HALO;
  const int H = k.L[0].H, W = k.L[0].W;
  float* strip = k.scratch + (long)bid * RDB_SROWS * W * RDB_GC;
  float* xch = reinterpret_cast<float*>(sV);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float amax = FMT ? *k.L[0].xamax : 0.f;          // max|x| of the block input: complete (an earlier launch wrote it)
  const int tiles_x = (W + 15) / 16;
#pragma unroll 1
  for (int layer = 0; layer < 5; ++layer) {
    const ConvK& p = k.L[layer];
    const int hh = RDB_HALO - layer;                                 // rows beyond the band this layer's output is needed on: y1 4, y2 3, y3 2, y4 1, conv5 0
    const int row_lo = max(0, band0 - hh), row_hi = min(H, band1 + hh);
    const float scA = FMT ? pow2_scale(amax) : 1.f;
    const float inv = FMT ? 1.f / (scA * pow2_scale(*p.wamax)) : 1.f;
    const int r0 = band0 - hh;                                       // first tile row starts here (may be negative)
    const int trows = (RDB_BAND + 2 * hh + 7) / 8;
    float vmax = 0.f;
#pragma unroll 1
    for (int tr = 0; tr < trows; ++tr) {
      const int ty0 = r0 + 8 * tr;
      if (ty0 >= row_hi || ty0 + 8 <= row_lo) continue;              // (block-uniform)
#pragma unroll 1
      for (int tc = 0; tc < tiles_x; ++tc) {
        if (layer < 4) vmax = fmaxf(vmax, rdb_tile<FMT, 1>(p, sV, n, ty0, tc * 16, band0, band1, strip, srow0, row_lo, row_hi, scA, inv, xch));
        else vmax = fmaxf(vmax, rdb_tile<FMT, 2>(p, sV, n, ty0, tc * 16, band0, band1, strip, srow0, row_lo, row_hi, scA, inv, xch));
      }
    }
    // block-wide maximum of what this layer produced: the next layer's scale, and the buffer's running max for the others
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    __syncthreads();                                // every wave's stores of this layer are issued and (vmcnt) complete
    if (lane == 0) s_red[wave] = vmax;
    __syncthreads();
    const float lm = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    if (layer < 4) {
      amax = fmaxf(amax, lm);
      if (FMT && k.amax_out && threadIdx.x == 0 && lm > __builtin_nontemporal_load(k.amax_out))
        atomicMax(reinterpret_cast<unsigned*>(k.amax_out), __float_as_uint(lm));
    } else if (FMT && p.yamax && threadIdx.x == 0 && lm > __builtin_nontemporal_load(p.yamax)) {
      atomicMax(reinterpret_cast<unsigned*>(p.yamax), __float_as_uint(lm));
    }
    __threadfence_block();
  }
}

}  // namespace

bool halo_winox_wanted(int cin_p, int cout_p, int T);      // conv_halo_wino.hip

extern "C" int64_t mt_rdb_scratch_floats(int32_t N, int32_t H, int32_t W) {
  return (int64_t)N * ((H + RDB_BAND - 1) / RDB_BAND) * RDB_SROWS * W * RDB_GC;
}

// descs[0..3]: the growth convs (x = channels [0, 64 + 32 i) of the block buffer, y = channels [64 + 32 i, +32), LeakyReLU),
// descs[4]: conv5 (x = all 192 channels, y = the next block's first 64 channels, res = the block input, alpha / beta).
// Weights in mt_weight_pack_halo fragment order 2 (what mt_conv_run uses for these shapes).  Returns 1 when the launch was
// taken, 0 when the geometry is not this kernel's (the caller then runs the five layers one by one), < 0 on errors.
extern "C" int mt_rdb_fwd(const mt_conv* descs, float* scratch, int64_t scratch_floats, float* buf_amax, void* stream) {
  MT_REQUIRE(descs && scratch, "mt_rdb_fwd: null pointer");
  RdbK k;
  const mt_conv& d0 = descs[0];
  const int N = d0.x.N, H = d0.x.H, W = d0.x.W;
  if ((H % RDB_BAND) != 0 || (W % 16) != 0 || H < RDB_BAND) return 0;
  const int prec = d0.precision;
  if (prec != MT_PREC_F16X3 && prec != MT_PREC_BF16X3) return 0;
  for (int i = 0; i < 5; ++i) {
    const mt_conv& d = descs[i];
    const int cin = 64 + 32 * i, cout = i < 4 ? 32 : 64;
    if (d.x.C != cin || d.y.C != cout || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.ups != 0 || d.transposed ||
        d.precision != prec || d.wt_layout != MT_WT_HALO || d.x.N != N || d.x.H != H || d.x.W != W || d.x.p != d0.x.p ||
        d.x.n_in < N || d.y.n_in < N || d.dsrc.p || d.accumulate || !halo_winox_wanted(cin, cout, 9))
      return 0;
    if (i < 4 && (d.y.p != d0.x.p + cin || d.y.sn != d0.x.sn || d.y.sh != d0.x.sh || d.y.sw != d0.x.sw || d.res.p)) return 0;
    if (prec == MT_PREC_F16X3 && (!d.x_amax || !d.w_amax)) return 0;
    if (fill_convk(&d, &k.L[i]) != MT_OK) return -1;
  }
  MT_REQUIRE(scratch_floats >= mt_rdb_scratch_floats(N, H, W), "mt_rdb_fwd: scratch too small");
  k.scratch = scratch;
  k.bands = H / RDB_BAND;
  k.amax_out = buf_amax;
  const int blocks = N * k.bands;
  if (prec == MT_PREC_F16X3) hipLaunchKernelGGL((rdb_fwd_kernel<1>), dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), k);
  else hipLaunchKernelGGL((rdb_fwd_kernel<0>), dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), k);
  int rc = mt_check_launch("mt_rdb_fwd");
  return rc ? rc : 1;
}
