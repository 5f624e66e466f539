// Kernel selection of the conv engine (see dispatch.h) and the two plan queries of the C ABI.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "dispatch.h"

// ---- switches ---------------------------------------------------------------------------------------------------------
const MtSwitches& mt_switches() {
  static const MtSwitches sw = [] {
    auto on = [](const char* name) { const char* e = getenv(name); return !(e && e[0] == '0'); };      // default on
    auto off = [](const char* name) { const char* e = getenv(name); return e && e[0] == '1'; };        // default off
    MtSwitches s;
    s.halo_rb = on("MT355_HALO_RB");
    s.wino = on("MT355_WINO");
    s.winox = on("MT355_WINOX");
    s.wino16 = on("MT355_WINO16");
    s.wino25 = on("MT355_WINO25");
    s.wino43 = on("MT355_WINO43");
    s.planes = on("MT355_PLANES");
    s.tiny = on("MT355_TINY");
    s.tiny_out = on("MT355_TINY_OUT");
    s.wgrad_pc = on("MT355_WGRAD_PC");
    s.wgrad_pc5 = on("MT355_WGRAD_PC5");
    s.wgrad_s2_halo = on("MT355_WGRAD_S2_HALO");
    s.wgrad3_tall = on("MT355_WGRAD3_TALL");
    s.wgrad5_rows = off("MT355_WGRAD5_ROWS");
    s.wgrad_thin = on("MT355_WGRAD_THIN");
    return s;
  }();
  return sw;
}

// ---- stride-1 halo-tile family ------------------------------------------------------------------------------------------
// The register-B kernels are used for layers with at least two K chunks (with a single chunk the exposed first weight
// fetch costs more than the barriers it saves) and more than 32 output channels (a 32-wide tile leaves a wave 6 MFMAs
// per tap, too few to cover the fragment fetch: measured 129 vs 166 TFLOP/s for the LDS-staged kernel).
bool halo_rb_wanted(int cin_p, int cout_p) { return mt_switches().halo_rb && cin_p > 32 && cout_p > 32; }

// 3 x 3 layers of the xi-split kernel: more than one 32-channel K chunk's worth of gathered channels, <= 64 outputs
// (64-wide tiles from 128 gathered channels on: with two K chunks the exchange of the epilogue costs more than the
// halved operand traffic saves -- 0.64 vs 0.57 ms for 64 -> 64 at 256 x 256)
bool halo_winox_wanted(int cin_p, int cout_p, int T) {
  const MtSwitches& sw = mt_switches();
  return T == 9 && sw.wino && sw.winox && cin_p > 32 && (cout_p <= 32 || (cout_p <= 64 && cin_p >= 128));
}

// 1-D Winograd F(2,3) on the 3 x 3 layers the register-B kernel would take, and on the xi-split kernel's layers
bool halo_wino_wanted(int cin_p, int cout_p, int T) {
  return T == 9 && ((halo_rb_wanted(cin_p, cout_p) && mt_switches().wino) || halo_winox_wanted(cin_p, cout_p, T));
}

// F(4,3): fp16 split, 128-wide output-channel tiles, more than one K chunk of gathered channels
bool halo_wino43_wanted(int cin_p, int cout_p, int T, int f16) {
  return T == 9 && f16 && mt_switches().wino43 && cin_p > 32 && cout_p > 64;
}

// F(2,5): fp16 split, 128-wide tiles, whole 32-channel chunks, more than one of them
bool halo_wino25_wanted(int cin_p, int cout_p, int T, int f16) {
  return T == 25 && f16 && mt_switches().wino25 && cin_p > 32 && (cin_p & 31) == 0 && cout_p > 64;
}

bool planes_shape_ok(int K, int pad, int cin_p, int cout_p, int precision) {
  return (K == 3 || K == 5) && pad == K / 2 && (cin_p & 31) == 0 && cin_p >= 64 && cout_p >= 64 &&
         (precision == MT_PREC_BF16X3 || precision == MT_PREC_F16X3);
}

MtHaloSel mt_halo_select(int KS, int NS, int fmt, int C, int Co, int Ho, bool x32) {
  MtHaloSel s;
  s.bn = Co > 64 ? 128 : (Co > 32 ? 64 : 32);
  s.ck = 32;
  if (NS == 2) {
    if (KS == 3 && halo_wino43_wanted(C, Co, 9, fmt)) { s.kernel = MT_HK_WINO43; s.bn = 128; return s; }
    if (halo_wino_wanted(C, Co, KS * KS)) {
      if (halo_winox_wanted(C, Co, 9) && x32) { s.kernel = MT_HK_WINOX; return s; }
      // 16 x 16 tiles for 128-wide layers (an 8-row image would waste half of one)
      if (Co > 64 && mt_switches().wino16 && Ho > 8) { s.kernel = MT_HK_WINO16; return s; }
      s.kernel = MT_HK_HALO_WINO;
      if (s.bn < 64) s.bn = 64;
      return s;
    }
    // (the pack is chosen from (channels, taps, precision) alone -- mt_weight_pack_halo_frag_order -- so nothing else
    // may be looked at here: every geometry, up-sampled gathers included, goes to the kernel its pack was made for)
    if (KS == 5 && halo_wino25_wanted(C, Co, 25, fmt)) { s.kernel = MT_HK_WINO25; s.bn = 128; return s; }
    if (halo_rb_wanted(C, Co)) { s.kernel = MT_HK_RB; return s; }
    s.ck = halo_ck(C, NS);
  } else {
    s.ck = halo_ck(C, NS);
  }
  s.kernel = MT_HK_LDS;
  return s;
}

// ---- weight gradients ---------------------------------------------------------------------------------------------------
MtWgradSel mt_wgrad_select(const mt_wgrad* d) {
  const MtSwitches& sw = mt_switches();
  MtWgradSel s = {};
  s.trows = 4; s.d_terms = 2; s.launches = 1;
  const int prec = d->precision;
  if (d->algo != MT_ALGO_AUTO) return s;
  if (prec != MT_PREC_BF16X3 && prec != MT_PREC_BF16X6 && prec != MT_PREC_F16X3) return s;
  if ((d->stride != 1 && d->stride != 2) || d->KH != d->KW || (d->KH != 3 && d->KH != 5) || d->pad != d->KH / 2) return s;
  if (d->stride == 2 && (d->ups != 0 || prec == MT_PREC_BF16X6 || !sw.wgrad_s2_halo)) return s;
  // 32-bit offsets inside an image (24-bit strides for v_mul_u32_u24): larger tensors take the generic kernel
  if (!(d->d.sh < (1 << 24) && d->d.sw < (1 << 24) && d->g.sh < (1 << 24) && d->g.sw < (1 << 24) &&
        (long)d->d.H * d->d.sh < (1L << 31) && (long)d->g.H * d->g.sh < (1L << 31))) return s;
  s.halo = 1;
  // stride 1, two-term splits: the producer / consumer kernel (two LDS images); 5 x 5: kernel rows {0,1} | {2,3} | {4}
  s.pc = d->stride == 1 && prec != MT_PREC_BF16X6 && sw.wgrad_pc && (d->KH == 3 || sw.wgrad_pc5);
  // single-role 3 x 3 stride 1 (two-term splits): 8-row pixel tiles (+6 %; the 5 x 5 gradient loses 3 % with them)
  s.tall = !s.pc && d->stride == 1 && d->KH == 3 && prec != MT_PREC_BF16X6 && sw.wgrad3_tall;
  // 16 gathered channels (and no up-sampled gather): four taps share a B tile
  s.thin = s.pc && sw.wgrad_thin && d->g.C == 16 && d->ups == 0 && !d->skip_tiles && !d->n_tiles;
  // one term of D: fp16 terms only (the leading bf16 term has 8 bits), on the kernels that have the form
  // (producer / consumer; the stride-2 gathers of the single-role kernel)
  s.d_terms = (prec == MT_PREC_F16X3 && d->d_terms == 1 && ((s.pc && !s.thin) || d->stride == 2)) ? 1 : 2;
  s.trows = s.tall ? 8 : ((s.pc && !s.thin && prec == MT_PREC_F16X3) ? wgpc_tr(s.d_terms, d->KH, d->d.H) : 4);
  s.rows5 = !s.pc && d->stride == 1 && d->KH == 5 && (prec == MT_PREC_BF16X6 || sw.wgrad5_rows);
  if (d->KH == 5) s.launches = s.thin ? 1 : (s.pc ? 3 : (s.rows5 ? 1 : 2));
  // a tile list (mt_wgrad.tiles): the 3 x 3 producer / consumer kernel only
  s.tile_list = s.pc && !s.thin && d->KH == 3;
  return s;
}

int mt_wgrad_tiles_check(const mt_wgrad* d, const MtWgradSel& sel) {
  MT_REQUIRE(d->n_tiles >= 0 && d->n_tiles <= MT_WGRAD_TILES_MAX, "mt_wgrad: n_tiles must be 0 .. %d (got %d)", MT_WGRAD_TILES_MAX, d->n_tiles);
  if (d->n_tiles == 0) return MT_OK;
  MT_REQUIRE(sel.halo && sel.tile_list, "mt_wgrad: this layer's kernel takes no tile list (mt_wgrad_plan.tile_list_ok)");
  MT_REQUIRE(d->skip_tiles == 0, "mt_wgrad: a tile list and skip_tiles exclude one another");
  for (int i = 0; i < d->n_tiles; ++i) {
    const mt_wgrad_tile& t = d->tiles[i];
    MT_REQUIRE(32 * t.a[0] < d->d.C && 32 * t.a[1] < d->d.C && 32 * t.b[0] < d->g.C && 32 * t.b[1] < d->g.C,
               "mt_wgrad: tile %d names a 32-channel group outside D (%d channels) or G (%d)", i, d->d.C, d->g.C);
    MT_REQUIRE(t.quads >= 1 && t.quads <= 15, "mt_wgrad: tile %d wants no product (quads = %d)", i, t.quads);
  }
  return MT_OK;
}

// ---- plan queries -------------------------------------------------------------------------------------------------------
namespace {
const char* fmt_name(int prec) { return prec == MT_PREC_F16X3 ? "f16" : "bf16"; }
const char* htag(int prec) { return prec == MT_PREC_BF16X6 ? "3,bf16" : (prec == MT_PREC_BF16X3 ? "2,bf16" : "2,f16"); }
const char* ktag(int prec) { return prec == MT_PREC_F32 ? "" : (prec == MT_PREC_BF16X3 ? ",2" : (prec == MT_PREC_BF16X6 ? ",3" : ",2,f16")); }
const char* tile_cfg(int co) { return co > 64 ? "2,2,2,2" : (co > 32 ? "2,2,2,1" : "4,1,1,1"); }
int products_of(int prec) { return prec == MT_PREC_F32 ? 1 : (prec == MT_PREC_BF16X6 ? 6 : 3); }
}  // namespace

extern "C" int mt_conv_plan_query(const mt_conv* d, int32_t cin_real, int32_t cout_real, mt_conv_plan* out) {
  MT_REQUIRE(d && out, "mt_conv_plan_query: null pointer");
  MT_REQUIRE(d->x.C > 0 && d->y.C > 0 && d->KH > 0 && d->KH == d->KW, "mt_conv_plan_query: channel counts / square kernel");
  MT_REQUIRE(d->precision >= MT_PREC_F32 && d->precision <= MT_PREC_F16X3, "mt_conv_plan_query: unknown precision %d", d->precision);
  const MtSwitches& sw = mt_switches();
  memset(out, 0, sizeof(*out));
  const int prec = d->precision, K = d->KH;
  out->products = products_of(prec);
  out->wino_num = out->wino_den = 1;
  out->path = MT_PATH_IGEMM;
  out->wt_layout = MT_WT_ROWS;
  // the fp32 VALU kernels (separate entry points): stride-1 'same' layers with a handful of real channels on one side
  const bool same1 = !d->transposed && d->stride == 1 && d->pad == K / 2 && d->ups == 0 && d->algo == MT_ALGO_AUTO;
  out->tiny_in_ok = sw.tiny && same1 && K == 3 && cin_real >= 1 && cin_real <= 4 && (cout_real == 64 || cout_real == 128) &&
                    d->x.C >= 4 && d->y.C == cout_real;
  out->tiny_out_ok = sw.tiny && sw.tiny_out && same1 && (K == 3 || K == 5) && cin_real == 128 && d->x.C == 128 &&
                     (cout_real == 1 || cout_real == 2);
  const bool split = prec == MT_PREC_BF16X3 || prec == MT_PREC_BF16X6 || prec == MT_PREC_F16X3;
  const bool halo = d->algo == MT_ALGO_AUTO && split && !d->transposed && d->stride == 1 && (K == 3 || K == 5) &&
                    d->pad == K / 2 && (d->x.C & 7) == 0;
  if (halo) {
    const int NS = prec == MT_PREC_BF16X6 ? 3 : 2, fmt = prec == MT_PREC_F16X3 ? 1 : 0;
    const bool x32 = ((long)(d->x.H - 1) * d->x.sh + (long)(d->x.W - 1) * d->x.sw + d->x.C + 4) * 4 < (1L << 31);
    const MtHaloSel s = mt_halo_select(K, NS, fmt, d->x.C, d->y.C, d->y.H, x32);
    out->path = MT_PATH_HALO;
    out->wt_layout = MT_WT_HALO;
    switch (s.kernel) {
      case MT_HK_WINO43: snprintf(out->kernel, sizeof(out->kernel), "igemm_wino43_kernel<%s>", fmt_name(prec)); out->wino_num = 1; out->wino_den = 2; break;
      case MT_HK_WINOX: snprintf(out->kernel, sizeof(out->kernel), "igemm_winox_kernel<%s,%d>", fmt_name(prec), s.bn > 32 ? 2 : 1); out->wino_num = 2; out->wino_den = 3; break;
      case MT_HK_WINO16: snprintf(out->kernel, sizeof(out->kernel), "igemm_wino16_kernel<%s>", fmt_name(prec)); out->wino_num = 2; out->wino_den = 3; break;
      case MT_HK_HALO_WINO: snprintf(out->kernel, sizeof(out->kernel), "igemm_halo_wino_kernel<%s,%d>", fmt_name(prec), s.bn > 64 ? 1 : 2); out->wino_num = 2; out->wino_den = 3; break;
      case MT_HK_WINO25: snprintf(out->kernel, sizeof(out->kernel), "igemm_wino25_kernel<%s>", fmt_name(prec)); out->wino_num = 3; out->wino_den = 5; break;
      case MT_HK_RB: snprintf(out->kernel, sizeof(out->kernel), "igemm_halo_rb_kernel<%d,%s,%d>", K, fmt_name(prec), s.bn > 64 ? 1 : 2); break;
      default: snprintf(out->kernel, sizeof(out->kernel), "igemm_halo_kernel<%d,%s,%s%s>", K, htag(prec), tile_cfg(d->y.C), s.ck == 16 ? ",16" : ""); break;
    }
    return MT_OK;
  }
  if (d->algo == MT_ALGO_AUTO && sw.planes && d->stride == 2 && d->ups == 0 && planes_shape_ok(K, d->pad, d->x.C, d->y.C, prec)) {
    out->path = MT_PATH_PLANES;
    out->wt_layout = MT_WT_PLANES;
    snprintf(out->kernel, sizeof(out->kernel), "igemm_halo_planes_kernel<%d,s2,%s>", K, d->transposed ? "t" : "c");
    return MT_OK;
  }
  snprintf(out->kernel, sizeof(out->kernel), "%s<%s%s>", prec == MT_PREC_F32 ? "igemm_kernel" : "igemm_split_kernel", tile_cfg(d->y.C), ktag(prec));
  return MT_OK;
}

extern "C" int mt_wgrad_plan_query(const mt_wgrad* d, int32_t a_real, int32_t b_real, mt_wgrad_plan* out) {
  MT_REQUIRE(d && out, "mt_wgrad_plan_query: null pointer");
  MT_REQUIRE(d->d.C > 0 && d->g.C > 0 && d->KH > 0 && d->KH == d->KW && d->d.N > 0 && d->d.H > 0 && d->d.W > 0,
             "mt_wgrad_plan_query: geometry");
  MT_REQUIRE(d->precision >= MT_PREC_F32 && d->precision <= MT_PREC_F16X3, "mt_wgrad_plan_query: unknown precision %d", d->precision);
  MT_REQUIRE(d->d_terms >= 0 && d->d_terms <= 2, "mt_wgrad_plan_query: d_terms must be 0, 1 or 2 (got %d)", d->d_terms);
  memset(out, 0, sizeof(*out));
  const int prec = d->precision, K = d->KH, ap = d->d.C, bp = d->g.C, T = K * K;
  const MtWgradSel s = mt_wgrad_select(d);
  { const int rc = mt_wgrad_tiles_check(d, s); if (rc) return rc; }
  out->halo = s.halo;
  out->pc = s.pc; out->thin = s.thin; out->d_terms = s.d_terms; out->tile_rows = s.trows;
  out->products = products_of(prec);
  const int lo = a_real < b_real ? a_real : b_real, hi = a_real < b_real ? b_real : a_real;
  out->tiny_ok = mt_switches().tiny && s.halo && d->stride == 1 && K == 3 && d->pad == 1 && d->ups == 0 && !d->skip_tiles && lo >= 1 &&
                 lo <= 4 && (hi == 64 || hi == 128) && (ap > bp ? ap : bp) == hi && d->d.H == d->g.H && d->d.W == d->g.W;
  if (s.halo) {
    out->ntiles = d->d.N * ((d->d.H + s.trows - 1) / s.trows) * ((d->d.W + 15) / 16);
    const int tiles64 = d->n_tiles > 0 ? d->n_tiles : ((ap + 63) / 64) * ((bp + 63) / 64);
    out->blocks = s.thin ? (ap + 63) / 64 : tiles64 * (s.pc ? 1 : (K == 3 ? 1 : (s.rows5 ? 5 : 2)));
    out->tile_list_ok = s.tile_list;
    out->bias_tap_ok = 1;
    if (s.pc && s.d_terms == 1 && s.trows == 4) out->products = 2;
    if (s.pc && s.d_terms == 1 && s.trows == 8) out->products = 2;
    if (!s.pc && d->stride == 2 && s.d_terms == 1) out->products = 2;
    out->batchable = s.pc && !s.thin && K == 3 && d->ups == 0 && (prec == MT_PREC_F16X3 || prec == MT_PREC_BF16X3) &&
                     d->d.N == d->g.N && d->d.H == d->g.H && d->d.W == d->g.W;
    if (s.thin) snprintf(out->kernel, sizeof(out->kernel), "wgrad_thin_kernel<%d,%s>", K, fmt_name(prec));
    else if (s.pc) snprintf(out->kernel, sizeof(out->kernel), "wgrad_pc_kernel<%d,%s,%s>", K, K == 3 ? "3" : "2+2+1", fmt_name(prec));
    else snprintf(out->kernel, sizeof(out->kernel), "wgrad_halo_kernel<%d,%s,%s>", K, K == 3 ? "3" : (s.rows5 ? "1" : "3+2"), htag(prec));
    return MT_OK;
  }
  // the generic gather kernels: one block per (channel tile, tap)
  int tiles;
  const char* cfg;
  if (ap > 64 && bp > 64) { tiles = ((ap + 127) / 128) * ((bp + 127) / 128); cfg = "2,2,2,2"; }
  else if (ap <= 32) { tiles = (bp + 127) / 128; cfg = "1,4,1,1"; }
  else if (bp <= 32) { tiles = (ap + 127) / 128; cfg = "4,1,1,1"; }
  else { tiles = ((ap + 63) / 64) * ((bp + 63) / 64); cfg = "2,2,1,1"; }
  out->blocks = tiles * T;
  out->d_terms = 2;
  snprintf(out->kernel, sizeof(out->kernel), "%s<%s%s>", prec == MT_PREC_F32 ? "wgrad_kernel" : "wgrad_split_kernel", cfg, ktag(prec));
  return MT_OK;
}
