/* mt355 -- C ABI of the MI355X (gfx950) kernels behind the DFPN / CHN hot path.
 *
 * The reference (davidalvarezdlt/master_thesis) has NO FFI: every op on its hot
 * path is a torch.nn.functional call inside model_dfpn.py / model_chn.py /
 * utils.py / model_vgg.py.  Each entry point below replaces one such call-site
 * group (cited as reference file:line, relative to master_thesis/) and is what
 * a binding for this path would bind (see INTEGRATION.md for the ctypes stub).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all device pointers are HBM
 *     addresses on the current HIP device; `stream` is a hipStream_t (as void*).
 *   - return 0 on success, a negative mt_status otherwise; no exceptions cross
 *     the ABI; mt_last_error() gives a thread-local message.
 *   - nothing synchronises, and the caller passes workspaces; the one allocation inside the library is a 4 MB pool of
 *     reduction scratch (one area per stream), made by mt_init().
 *   - activations inside the engine are "NHWC" engine tensors: float32,
 *     channels innermost and contiguous, the channel count padded with ZEROS.
 *     The ABI itself accepts any C % 4 == 0 (mt_view); the LDS halo-tile
 *     kernels (stride-1 3x3 / 5x5 forward, backward-data and weight gradient,
 *     i.e. > 90 % of the FLOPs) additionally need C % 8 == 0 -- one 32-byte run
 *     = one pixel's 8 channels of one tap = one MFMA K group -- and a tensor
 *     with C % 8 != 0 silently takes the slower implicit-GEMM gather kernel.
 *     The host side (master_thesis_amd/engine.py: cpad) therefore pads every
 *     engine tensor to a multiple of 8 (1 -> 8, 3 -> 8, 10 -> 16).
 *     Image / row / pixel strides are explicit (in floats) so that channel
 *     windows of a wider buffer (dense blocks) and transposed pixel orders
 *     (SeparableConv4d's permute) are addressed in place.
 */
#ifndef MT355_H
#define MT355_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libmt355.so is built with -fvisibility=hidden: the declarations of this header are its whole export list. */
#pragma GCC visibility push(default)

enum mt_status {
  MT_OK = 0,
  MT_ERR_ARG = -1,      /* bad shape / alignment / null pointer */
  MT_ERR_LAUNCH = -2,   /* hipLaunchKernel failed (message in mt_last_error) */
  MT_ERR_WORKSPACE = -3 /* workspace too small */
};

enum mt_act { MT_ACT_NONE = 0, MT_ACT_RELU = 1, MT_ACT_LRELU02 = 2 };
/* Arithmetic of the MFMA contraction (inputs and outputs are fp32 either way):
 * MT_PREC_F32    v_mfma_f32_32x32x2_f32, exact fp32 products (157 TFLOP/s peak);
 * MT_PREC_BF16X3 every fp32 operand split into 2 bf16 terms, 3 partial products
 *                on v_mfma_f32_32x32x16_bf16, fp32 accumulate: 17-18 significant
 *                bits (~4e-6 relative per layer), 5.3x the fp32-MFMA rate;
 * MT_PREC_BF16X6 3 bf16 terms, 6 partial products: >= 24 significant bits =
 *                fp32-equivalent, 2.7x the fp32-MFMA rate;
 * MT_PREC_F16X3  2 fp16 terms (22 significant bits) of the operand scaled by a
 *                per-tensor power of two, 3 partial products on
 *                v_mfma_f32_32x32x16_f16: ~7e-8 relative per product (below the
 *                fp32 accumulation noise) at the BF16X3 cost.  Needs the
 *                device scalars max|X|, max|W| (or max|D|, max|G|) from
 *                mt_absmax; not available on the fp32 kernels' weight layout
 *                (pack with mt_weight_pack_split / _halo and w_amax). */
enum mt_precision { MT_PREC_F32 = 0, MT_PREC_BF16X3 = 1, MT_PREC_BF16X6 = 2, MT_PREC_F16X3 = 3 };
/* Kernel selection of mt_conv_run: AUTO picks the LDS halo-tile direct
 * convolution for stride-1 3x3/5x5 "same" layers (split precisions, X.C % 8 ==
 * 0) and the implicit-GEMM gather kernel otherwise; IGEMM forces the latter. */
enum mt_algo { MT_ALGO_AUTO = 0, MT_ALGO_IGEMM = 1 };
/* Layout of mt_conv.wt: ROWS = [Cout_p][KH*KW][Cin_p] rows (mt_weight_pack /
 * mt_weight_pack_split: every kernel but the halo one); HALO = the per-tile LDS
 * images of mt_weight_pack_halo (required by, and only valid for, layers that
 * MT_ALGO_AUTO sends to the halo-tile kernel); PLANES = the fragment stream of
 * mt_weight_pack_planes (stride-2 3x3 / 5x5 layers with >= 64 channels on both sides,
 * two-term split precisions: mt_weight_pack_planes_wanted). */
enum mt_wt_layout { MT_WT_ROWS = 0, MT_WT_HALO = 1, MT_WT_PLANES = 2 };

/* ABI revision: 2 since round 4 (mt_wgrad gained d_terms in round 3 without a bump; the plan queries). */
int mt_version(void);
const char* mt_last_error(void);
/* Allocates the library's one piece of device memory -- a 4 MB pool of reduction scratch, one area per stream (two-level
 * max|y| of the conv epilogues, fixed-order loss sums) -- on the current device.  Call it once after the device is chosen
 * and outside any graph capture; a launch that finds no pool falls back to allocating it itself (outside captures) or to
 * same-address atomics.  mt_amax_scratch_reset re-zeroes the pool on `stream` (after an aborted launch; not capturable). */
int mt_init(void);
int mt_amax_scratch_reset(void* stream);

/* A strided NHWC4 view: element (n, y, x, c) lives at
 * p[(n / n_in)*sg + (n % n_in)*sn + y*sh + x*sw + c], c < C (C % 4 == 0,
 * p 16-byte aligned, strides % 4 == 0).  The two-level image index (groups of
 * n_in images, group stride sg) expresses SeparableConv4d's
 * (target position <-> reference position) permute, model_dfpn.py:621-623,
 * as pure addressing; plain tensors use n_in = N, sg = 0.  mt_conv honours it
 * on x and y, mt_wgrad on d and g; every other entry point needs n_in >= N. */
typedef struct {
  float* p;
  int64_t sn, sh, sw;
  int32_t N, H, W, C;
  int32_t n_in;
  int64_t sg;
} mt_view;

/* One convolution-shaped contraction  (replaces every F.conv2d /
 * F.conv_transpose2d call-site and its autograd backward-data:
 * model_dfpn.py:596-605,638-660,686-711; model_chn.py:522-535,602-606;
 * model_vgg.py:40-41):
 *
 *   acc[n,y,x,o] = sum_{kh,kw,c} X[n, iy, ix, c] * Wt[o][kh][kw][c]
 *   gather (transposed == 0):  iy = y*stride - pad + kh            (conv fwd,
 *                                                     conv_transpose bwd-data)
 *   gather (transposed == 1):  iy = (y + pad - kh) / stride, taken only when
 *                              divisible                (conv bwd-data,
 *                                                        conv_transpose fwd)
 *   `ups` : the gathered tensor is a nearest x2^ups up-sampling of X
 *           (model_chn.py:550,553): logical size (H<<ups, W<<ups), physical
 *           index = logical >> ups.
 *   epilogue:  v = act(acc + bias[o]) * alpha
 *              v += beta * R[n,y,x,o]                    (R optional)
 *              v += Y[n,y,x,o]                           (when accumulate)
 *              v *= act'(D[n,y,x,o])  for o >= dact_c0   (D optional: the
 *                     saved activation whose derivative gates this gradient)
 *              Y[n,y,x,o] = v
 * Wt is [Y.C][KH*KW*X.C] (see mt_weight_pack); bias is [Y.C] or NULL.
 */
typedef struct {
  mt_view x;          /* gathered input  */
  mt_view y;          /* output          */
  const float* wt;
  const float* bias;
  int32_t KH, KW, stride, pad, transposed, ups;
  int32_t act;
  float alpha;
  mt_view res;        /* res.p == NULL: none */
  float beta;
  mt_view dsrc;       /* dsrc.p == NULL: none */
  int32_t dact, dact_c0;
  int32_t accumulate;
  int32_t precision;  /* mt_precision; split precisions need wt from mt_weight_pack_split */
  int32_t algo;       /* mt_algo */
  int32_t wt_layout;  /* mt_wt_layout */
  const float* x_amax; /* MT_PREC_F16X3: device scalar >= max|X| (mt_absmax) */
  const float* w_amax; /* MT_PREC_F16X3: the scalar given to mt_weight_pack_halo / _split */
  float* y_amax;       /* optional device scalar (>= 0): raised atomically to max|y| over the
                          values this launch writes, so that the next layer's x_amax needs no
                          extra pass (zero it before the first launch that feeds it) */
} mt_conv;

int mt_conv_run(const mt_conv* d, void* stream);
/* The same call for stride-1 3x3 "same" layers whose gathered tensor has at most FOUR real channels and whose
 * output has 64 or 128 (model_vgg.py:40 conv1_1, the 1 -> 128 entry of SeparableConv4d model_dfpn.py:596-605,
 * and the backward-data of the 128 -> 1 / 128 -> 2 exits, model_dfpn.py:604,710): plain fp32 on the VALU, no
 * operand split and no packed weights -- the launch is bound by the tensor it writes.  `w` are the parameter's
 * own fp32 weights, addressed as w[co * so + ci * si + tap * st] (tap = kh * 3 + kw); flip = 1 reverses the
 * taps (a Conv2d's backward-data run as a forward convolution, with the two channel strides swapped).
 * Supported of mt_conv: x / y views (two-level image index included), bias, act, alpha, dsrc / dact
 * (dact_c0 = 0), y_amax.  wt, wt_layout, precision, x_amax, w_amax are ignored; res / accumulate / ups /
 * transposed must be off. */
int mt_conv_tiny_in_run(const mt_conv* d, const float* w, int32_t cin, int64_t so, int64_t si, int64_t st,
                        int32_t flip, void* stream);
/* The mirror case: 128 gathered channels, ONE or TWO output channels, stride-1 3x3 / 5x5 "same" (the 128 -> 2 exit
 * of FlowEstimator model_dfpn.py:710, the 128 -> 1 exit of SeparableConv4d model_dfpn.py:604, and the backward-data
 * of FlowEstimator's 10 -> 128 entry model_dfpn.py:688, of which only the two flow channels are read).  zero_fill = 1
 * writes whole pixels of y (result in channels 0 .. cout-1, zeros in the padding); zero_fill = 0 writes only channels
 * co_off .. co_off+cout-1.  Supported of mt_conv: x / y views, bias, act, alpha, y_amax. */
int mt_conv_tiny_out_run(const mt_conv* d, const float* w, int32_t cout, int64_t so, int64_t si, int64_t st,
                         int32_t flip, int32_t co_off, int32_t zero_fill, void* stream);
/* slow one-thread-per-output restatement of mt_conv_run (self-check only) */
int mt_conv_run_naive(const mt_conv* d, void* stream);

/* What mt_conv_run (algo AUTO) does with a layer of this geometry and precision -- asked BEFORE packing the weights:
 * which pack the launch must be given, which kernel family runs, and the label / arithmetic facts a profile needs.
 * The library decides this in one place (csrc/dispatch.cpp); the host side holds no copy of the rules.
 * Read of `d`: x.{H,W,C,sh,sw}, y.{H,C}, KH, KW, stride, pad, transposed, ups, precision, algo -- no pointer is
 * dereferenced, nothing is launched.  cin_real / cout_real: channel counts before padding (the fp32 VALU kernels
 * are chosen by them). */
enum mt_conv_path { MT_PATH_IGEMM = 0, MT_PATH_HALO = 1, MT_PATH_PLANES = 2 };
typedef struct {
  int32_t path;          /* mt_conv_path */
  int32_t wt_layout;     /* mt_wt_layout the launch needs: mt_weight_pack_split / _halo / _planes */
  int32_t tiny_in_ok;    /* 1: the geometry is one mt_conv_tiny_in_run serves (no residual / accumulate on that path) */
  int32_t tiny_out_ok;   /* 1: ... mt_conv_tiny_out_run */
  int32_t products;      /* MFMA partial products per multiply-add of this precision: 1 (fp32), 3, 6 */
  int32_t wino_num, wino_den;   /* executed / algorithmic products: 1/1 direct, 2/3 F(2,3), 3/5 F(2,5), 1/2 F(4,3) */
  char kernel[96];       /* the kernel, as rocprofv3 names it (fp16 / bf16 terms spelled f16 / bf16) */
} mt_conv_plan;
int mt_conv_plan_query(const mt_conv* d, int32_t cin_real, int32_t cout_real, mt_conv_plan* out);

/* Weight gradient (autograd of the same call-sites):
 *   dW[a][kh][kw][b] = sum_{n,y,x} D[n,y,x,a] * G[n, y*stride-pad+kh, x*..., b]
 * D is iterated densely, G is gathered (conv: D = dY, G = X;
 * conv_transpose: D = X, G = dY).  Partial sums of `splits` pixel ranges go to
 * `slabs` ([splits][D.C][KH*KW][G.C] floats) and are reduced by
 * mt_wgrad_reduce, which also un-packs into the parameter's own layout.
 */
/* One 64 x 64 channel tile of a weight-gradient launch made of FOUR 32-channel groups chosen freely: D channels
 * 32*a[0].. and 32*a[1].., G channels 32*b[0].. and 32*b[1]...  `quads`: bit (2 i + j) set = the 32 x 32 product of
 * a[i] with b[j] is wanted (the others are neither computed nor written); `bias` = 1: this tile also writes the bias
 * sums of its two D groups (with_bias launches: exactly one tile per D group should).  For launches shared by
 * several layers whose useful products do not fill 64-aligned tiles: the four growth convolutions of a dense block
 * (model_chn.py:588-623, 32 outputs each from 64 / 96 / 128 / 160 inputs) need 14 such 32 x 32 products -- four
 * tiles instead of the five 64-aligned ones. */
typedef struct {
  uint8_t a[2], b[2];
  uint8_t quads, bias;
  uint8_t reserved[2];
} mt_wgrad_tile;
#define MT_WGRAD_TILES_MAX 8
typedef struct {
  mt_view d;
  mt_view g;
  int32_t KH, KW, stride, pad, ups;
  int32_t splits;
  float* slabs;
  int32_t precision;  /* mt_precision */
  int32_t algo;       /* mt_algo */
  int32_t with_bias;  /* halo-tile path only: also sum D over the pixels (the
                         bias gradient when D = dY); the slabs then carry
                         KH*KW + 1 taps, tap KH*KW / column 0 = those sums */
  const float* d_amax; /* MT_PREC_F16X3: device scalars >= max|D|, max|G| */
  const float* g_amax;
  uint32_t skip_tiles; /* halo-tile path: bit (at * btiles + bt) set = the 64 x 64 channel tile (D channels
                          64*at.., G channels 64*bt..) is not wanted and its slab entries stay undefined (a
                          launch shared by several layers forms a block-triangular product); 0 = all */
  int32_t d_terms;     /* (must be 0, 1 or 2) fp16 split, stride-1 3x3 / 5x5 producer / consumer kernel and the stride-2 gathers
                          -- anywhere else the field is ignored (mt_wgrad_plan.d_terms tells): 0 / 2 = both 16-bit terms of D
                          (three partial products, fp32-grade); 1 = D's leading term only (two partial products: D
                          rounded to 11 / 8 significant bits, independent rounding errors per pixel, ~1e-4 relative on
                          a gradient summed over >= 10^5 pixels) -- a speed / accuracy choice of the caller */
  int32_t n_tiles;     /* 0: the regular grid of 64-aligned channel tiles (skip_tiles applies).  1 .. MT_WGRAD_TILES_MAX:
                          the launch computes exactly tiles[0 .. n_tiles) (skip_tiles must be 0) -- only where
                          mt_wgrad_plan.tile_list_ok says so (the stride-1 3x3 producer / consumer kernel) */
  mt_wgrad_tile tiles[MT_WGRAD_TILES_MAX];
} mt_wgrad;

int mt_wgrad_run(const mt_wgrad* d, void* stream);
/* What mt_wgrad_run does with this geometry (read of `d`: the views' shapes and strides, KH, KW, stride, pad, ups,
 * precision, algo, skip_tiles, n_tiles, d_terms; `splits` and the pointers are not looked at): the caller sizes `splits` and
 * the slabs from it.  a_real / b_real: D's and G's channel counts before padding. */
typedef struct {
  int32_t halo;          /* 1: the halo-tile family takes it (with_bias available), 0: the generic gather kernels */
  int32_t pc;            /* producer / consumer kernel: 512-thread blocks, one per CU */
  int32_t thin;          /* 16 gathered channels: four taps per B tile, all kernel rows in one launch */
  int32_t batchable;     /* 1: mt_wgrad_run_batch takes it */
  int32_t tile_rows;     /* pixel tile = tile_rows x 16 */
  int32_t ntiles;        /* pixel tiles of the whole tensor (halo family) */
  int32_t blocks;        /* thread blocks per pixel range, summed over the launches of one call */
  int32_t bias_tap_ok;   /* 1: with_bias may be set */
  int32_t d_terms;       /* 16-bit terms of D that will enter: 1 only where asked for AND the kernel has the form */
  int32_t tiny_ok;       /* 1: the geometry is one mt_wgrad_tiny_run serves (Conv2d layers only) */
  int32_t products;      /* MFMA partial products per multiply-add */
  int32_t tile_list_ok;  /* 1: the kernel takes a tile list (mt_wgrad.n_tiles > 0) */
  char kernel[96];
} mt_wgrad_plan;
int mt_wgrad_plan_query(const mt_wgrad* d, int32_t a_real, int32_t b_real, mt_wgrad_plan* out);
/* `n` weight gradients of identical geometry (everything but d.p, g.p, slabs, d_amax, g_amax) as ONE
 * grid: stride-1 3x3 layers in MT_PREC_BF16X3 / MT_PREC_F16X3 only.  Each problem has its own slabs
 * ([splits][...], reduced per problem with mt_wgrad_reduce / _multi).  For the dense blocks of the
 * RRDBNet trunk (model_chn.py:588-623): their weight gradients do not depend on one another, so they
 * run after the backward chain as one launch with few, long pixel ranges per block. */
/* The weight (+ bias) gradient of a stride-1 3x3 "same" Conv2d with `tiny` <= 4 real channels on one side and 64 or
 * 128 on the other (the layers mt_conv_tiny_in_run serves, and the 128 -> 1 / 128 -> 2 exits): fp32 on the VALU,
 * the wide tensor is read once.  d->splits blocks each write one slab in mt_wgrad_run's layout; reduce as usual.
 * precision, d_amax, g_amax are ignored; skip_tiles must be 0. */
int mt_wgrad_tiny_run(const mt_wgrad* d, int32_t tiny, void* stream);
#define MT_WGRAD_BATCH_MAX 64
int mt_wgrad_run_batch(const mt_wgrad* descs, int32_t n, void* stream);
int64_t mt_wgrad_slab_floats(const mt_wgrad* d);
/* dst[a*sa + b*sb + t*st] (+)= scale * sum_s slabs[s][a][t][b], a < A, b < B,
 * t < T; slabs are [splits][AP][TS][BP] with TS = T, or T + 1 when the launch
 * ran with_bias: then db[a] (+)= scale * sum_s slabs[s][a][T][0].
 * BP must be a multiple of 4 and slabs 16-byte aligned (16-byte loads). */
int mt_wgrad_reduce(const float* slabs, int32_t splits, int32_t AP, int32_t T,
                    int32_t TS, int32_t BP, float* dst, int32_t A, int32_t B,
                    int64_t sa, int64_t sb, int64_t st, int32_t accumulate,
                    float scale, float* db, void* stream);
/* The same for up to 8 sub-blocks of one launch's slabs (row ranges [a0, a0 + A) x [0, B):
 * the layers that shared a weight-gradient launch) in ONE reduce launch. */
typedef struct {
  int32_t a0, A, B;
  int64_t sa, sb, st;
  float* dst;
  float* db;           /* bias sums of rows a0.. (slabs with the extra tap) or NULL */
} mt_reduce_sub;
int mt_wgrad_reduce_multi(const float* slabs, int32_t splits, int32_t AP, int32_t T,
                          int32_t TS, int32_t BP, const mt_reduce_sub* subs,
                          int32_t nsub, int32_t accumulate, float scale, void* stream);
/* db[c] (+)= sum_pixels D[n,y,x,c], c < C (ws: >= 256*D.C floats) */
int mt_bias_grad(const mt_view* d, float* db, int32_t C, float* ws,
                 int32_t accumulate, void* stream);

/* dst[a][t][b] = src[a*sa + b*sb + t*st] for a < A, b < B, else 0;
 * dst is [AP][T][BP].  Conv2d weight (O,I,kh,kw) -> fwd: a=O, b=I;
 * bwd-data: a=I, b=O.  ConvTranspose2d weight (I,O,kh,kw) likewise. */
int mt_weight_pack(const float* src, int32_t A, int32_t B, int32_t T,
                   int64_t sa, int64_t sb, int64_t st, float* dst, int32_t AP,
                   int32_t BP, void* stream);
/* same selection, split into 16-bit planes for the split precisions:
 * dst = [NS][AP][Kp] (plane s = s-th term; NS = 2 bf16 planes for BF16X3, 3 for
 * BF16X6, 2 fp16 planes of W * 2^k for F16X3), Kp = T*BP rounded up to a
 * multiple of 32.  flip = 1 reverses the tap order (t -> T-1-t): with a and b
 * swapped this turns the backward-data of a stride-1 "same" convolution into a
 * plain forward convolution.  w_amax: device scalar >= max|W| (mt_absmax_flat)
 * for F16X3, NULL otherwise. */
int mt_weight_pack_split(const float* src, int32_t A, int32_t B, int32_t T,
                         int64_t sa, int64_t sb, int64_t st, void* dst,
                         int32_t AP, int32_t BP, int32_t precision, int32_t flip,
                         const float* w_amax, void* stream);
/* Weight tiles of the halo-tile kernel: dst = [tap][chunk][ntile][plane][BN][CK+8]
 * bf16, every tile the exact LDS image the kernel copies (CK = 32 / 16 input
 * channels per chunk for BF16X3 / BF16X6, BN = 128 / 64 / 32 by AP). */
int64_t mt_weight_pack_halo_elems(int32_t AP, int32_t BP, int32_t T, int32_t precision);
int mt_weight_pack_halo(const float* src, int32_t A, int32_t B, int32_t T,
                        int64_t sa, int64_t sb, int64_t st, void* dst,
                        int32_t AP, int32_t BP, int32_t precision, int32_t flip,
                        const float* w_amax, void* stream);
/* ---- batched re-pack: every weight image of a model in two launches ---------
 * The packs are a pure function of the parameters, and the set of packs a model
 * needs is the same every step; re-packing after the optimizer step layer by
 * layer costs ~1300 tiny launches for the CHN step.  The host keeps two tables
 * in DEVICE memory and replays them once per step. */
typedef struct {
  const float* p;      /* segment start */
  int64_t n;           /* floats */
  float* out;          /* device scalar, raised to max|p[0..n)| (zero it first); segments may share it */
} mt_amax_seg;
int mt_absmax_segments(const mt_amax_seg* segs_dev, int32_t nseg, void* stream);

enum mt_pack_kind { MT_PACK_SPLIT = 0, MT_PACK_HALO = 1, MT_PACK_PLANES = 2 /* flip holds `transposed` */ };
typedef struct {
  const float* src;
  void* dst;
  const float* w_amax;  /* F16X3 scale scalar or NULL */
  int64_t sa, sb, st;
  int32_t A, B, T, AP, BP;
  int32_t precision;    /* mt_precision (not F32) */
  int32_t flip;
  int32_t kind;         /* mt_pack_kind: the arguments of mt_weight_pack_split / mt_weight_pack_halo_part */
  int32_t BP_total, b_off;   /* MT_PACK_HALO: stacked K parts (BP_total == BP, b_off == 0 for a plain pack) */
  int32_t frag_order;   /* MT_PACK_HALO: mt_weight_pack_halo_frag_order(AP, BP_total, precision) */
} mt_pack_job;
/* Layout mt_weight_pack_halo writes for this shape: 0 = LDS tile images, 1 = MFMA-fragment
 * order (the register-B kernel), 2 = fragment order of the 1-D Winograd F(2,3) transformed
 * weights G g (3 x 3 layers, T = 9: 12 (kernel row, xi) "taps"), 3 = fragment order with one
 * 32-channel tile per n-tile (3 x 3 layers with > 32 input and <= 32 output channels), 4 = the
 * Winograd weights as v_mfma_f32_16x16x32 fragments (128-wide tiles); the kernels choose by the
 * same rule. */
int32_t mt_weight_pack_halo_frag_order(int32_t AP, int32_t BP_total, int32_t T, int32_t precision);
/* jobs_dev: njobs entries in device memory; equivalent to calling the single-pack
 * entry point of every job.  jobs_host: the same table in host memory (argument checks). */
int mt_weight_pack_jobs(const mt_pack_job* jobs_dev, const mt_pack_job* jobs_host, int32_t njobs, void* stream);

/* Stride-2 layers (Conv2d stride 2 / ConvTranspose2d stride 2 and their backward-data:
 * model_dfpn.py:689-708, model_chn.py:525,527) on the halo-tile kernel: the strided gather
 * (transposed == 0) walks the four parity planes of X as extra K chunks, the transposed gather
 * (transposed == 1) runs one stride-1 window convolution per output parity class; either way only
 * the taps that exist are multiplied.  The pack is the weight-fragment stream in the order the
 * kernel consumes it; `transposed` must be the mt_conv.transposed of the launch that uses it. */
int32_t mt_weight_pack_planes_wanted(int32_t AP, int32_t BP, int32_t T, int32_t precision);
int64_t mt_weight_pack_planes_elems(int32_t AP, int32_t BP, int32_t T, int32_t transposed);
int mt_weight_pack_planes(const float* src, int32_t A, int32_t B, int32_t T, int64_t sa, int64_t sb,
                          int64_t st, void* dst, int32_t AP, int32_t BP, int32_t precision,
                          int32_t transposed, const float* w_amax, void* stream);

/* One K-part of a STACKED weight image: the B (input-channel) axis of the packed
 * tiles is the concatenation of several sources (e.g. the layers whose output
 * gradients a dense block's backward gathers in one launch, model_chn.py:618-622);
 * this call writes channels [b_off, b_off + BP) of BP_total (b_off and every
 * part but the last a multiple of the K chunk: 32, 16 for BF16X6).  All parts
 * share AP, the precision and (F16X3) one w_amax >= max|W| over the stack. */
int mt_weight_pack_halo_part(const float* src, int32_t A, int32_t B, int32_t T,
                             int64_t sa, int64_t sb, int64_t st, void* dst,
                             int32_t AP, int32_t BP, int32_t precision,
                             int32_t flip, const float* w_amax, int32_t BP_total,
                             int32_t b_off, void* stream);
/* out[0] = max(out[0], max |x|) over an NHWC4 view / a flat array (atomic max on
 * the device scalar: zero it first; several calls may share one scalar). */
int mt_absmax(const mt_view* x, float* out, void* stream);
int mt_absmax_flat(const float* p, int64_t n, float* out, void* stream);
int mt_pad_vector(const float* src, int32_t n, float* dst, int32_t np, void* stream);

/* ---- layout / streaming kernels --------------------------------------- */
/* NCHW (arbitrary strides) <-> NHWC4, with optional per-channel affine
 * (x*scale[c] + shift[c]) (normalisation, model_dfpn.py:71-72,
 * model_chn.py:73-74,83, model_vgg.py:37-38) */
int mt_nchw_to_nhwc4(const float* src, int64_t s_n, int64_t s_c, int64_t s_h,
                     int64_t s_w, int32_t N, int32_t C, int32_t H, int32_t W,
                     const float* scale, const float* shift, const mt_view* dst,
                     int32_t c_off, void* stream);
int mt_nhwc4_to_nchw(const mt_view* src, int32_t c_off, float* dst, int64_t d_n,
                     int64_t d_c, int64_t d_h, int64_t d_w, int32_t C,
                     const float* scale, const float* shift, int32_t accumulate,
                     void* stream);
/* Build the per-pair network input (FlowEstimator's 10-channel pack,
 * model_dfpn.py:733-741; CHN's 9-channel pack, model_chn.py:70-80) straight
 * into an NHWC4 tensor of B*R images: source i contributes channels
 * [c_off, c_off+C) read from p[b*sb + r*sr + c*sc + y*sy + x*sx], optionally
 * as v*scale[c] + shift[c] (the mean/std normalisation). sr = 0 repeats a
 * per-clip tensor over the R references. */
typedef struct {
  const float* p;
  int64_t sb, sr, sc, sy, sx;
  int32_t C, c_off;
  const float* scale;
  const float* shift;
} mt_pack_src;
int mt_pack_pairs(const mt_pack_src* srcs, int32_t nsrc, int32_t B, int32_t R,
                  const mt_view* dst, void* stream);
/* 2x2/2 max pool (model_vgg.py:40-43) */
int mt_maxpool2x2(const mt_view* x, const mt_view* y, void* stream);
/* dX[n,y,x,c] = sum of the 2^ups x 2^ups block of dXup (backward of the
 * nearest up-sampling folded into mt_conv.ups) */
int mt_sumpool(const mt_view* dxup, const mt_view* dx, int32_t ups, void* stream);
/* elementwise on NHWC4 views with equal geometry: y = a*x (+ b*z); y_amax: optional
 * device scalar raised to max|y| (as mt_conv.y_amax), or NULL */
int mt_axpby(const mt_view* x, float a, const mt_view* z, float b,
             const mt_view* y, float* y_amax, void* stream);
/* y = x * act'(s)  (in place allowed) */
int mt_dact(const mt_view* x, const mt_view* s, int32_t act, const mt_view* y,
            void* stream);

/* Output stage of CHN.forward (model_chn.py:81-85) straight from the network's NHWC output:
 *   y_hat[b,c,f,y,x]  = clamp(o[(b*F+f),y,x,c] * std[c] + mean[c], 0, 1)
 *   y_comp[b,c,f,y,x] = v[b,0,y,x] * x_t[b,c,y,x] + (1 - v[b,0,y,x]) * y_hat[b,c,f,y,x]
 * o: NHWC view (B*F, H, W, >= 3); x_t (B,3,H,W), v (B,1,H,W), y_hat / y_comp (B,3,F,H,W) contiguous. */
int mt_compose_clamp_fwd(const mt_view* o, const float* x_t, const float* v,
                         const float* mean3, const float* std3, int32_t B, int32_t F,
                         float* y_hat, float* y_comp, void* stream);
/* and its gradient wrt o (channels >= 3 of d_o are zeroed):
 *   d_o = (d_y_hat + (1 - v) * d_y_comp) * std[c] * [0 <= o*std+mean <= 1]
 * d_y_hat / d_y_comp: (B,3,F,H,W) contiguous, either may be NULL. */
int mt_compose_clamp_bwd(const mt_view* o, const float* v, const float* mean3,
                         const float* std3, int32_t B, int32_t F, const float* d_y_hat,
                         const float* d_y_comp, const mt_view* d_o, void* stream);

/* Masked cosine-similarity correlation volume
 * (CorrelationVGG.correlation_masked_4d, model_dfpn.py:534-565).
 * ft: NHWC4 (B, h, w, C) target features; fr: NHWC4 (B*R, h, w, C), image
 * b*R+r is reference r of clip b; P = h*w (P % 4 == 0);
 * vt: [B][P] visibility or NULL; vr: [B*R][P] or NULL;
 * out: [B*R][P_t][P_r] floats (== (B,R,h,w,h,w) contiguous);
 * ws: (B + B*R) * P * C floats of scratch (the normalised features); only the
 * general path uses it and it may be NULL otherwise: for P = 256, C % 64 == 0,
 * C >= 128 and per-image extents below 256 MiB (the DFPN case) one fused
 * kernel streams both feature slabs once (bf16x3 MFMA, norms and masks in the
 * epilogue). */
int mt_corr_masked(const mt_view* ft, const mt_view* fr, const float* vt,
                   const float* vr, int32_t B, int32_t R, float* out, float* ws,
                   void* stream);
/* CorrelationVGG.softmax_3d (model_dfpn.py:567-583) on the volume mt_corr_masked
 * writes: for every (clip b, target position pt) a softmax over the R * P values
 * x[(b*R + r)][pt][0..P) of all R reference frames jointly (P % 4 == 0).  One wave
 * per row, maximum and sum by wave shuffles; out may alias x. */
int mt_softmax3d(const float* x, int32_t B, int32_t R, int32_t P, float* out, void* stream);
/* out[row][c] = x[row][c]*v[row] / (||x[row]*v[row]||_2 + 1e-9), rows = pixels */
int mt_rownorm(const mt_view* x, const float* v, float* out, void* stream);

/* FlowsUtils.align_set (utils.py:77-104): bilinear+zeros warp of x and nearest
 * warp of v by an absolute normalised sampling grid.  The 5-D (B,C,F,H,W)
 * tensors of the reference are addressed in place (the reference first
 * materialises transpose(1,2).reshape(B*F,C,H,W)); H and W are contiguous.
 * grid is (B,F,H,W,2) contiguous, last dim (x, y). */
typedef struct {
  int32_t B, F, C, H, W;
  int64_t xb, xc, xf; /* x  (B,C,F,H,W) strides in floats           */
  int64_t vb, vf;     /* v  (B,1,F,H,W)                              */
  int64_t ob, oc, of; /* xo (and dxo in the backward)                */
  int64_t wb, wf;     /* vo                                          */
} mt_warp_layout;
/* align_corners=1: DFPN convention (utils.py:96,101); 0: the CPN convention
 * (model_cpn.py:81-88).  idx (optional, int32 [B*F*H*W][4]) receives per
 * output pixel (y0, x0) = the top-left bilinear corner and (ny, nx) = the
 * nearest sample (-1,-1 when outside): the integer indices that must be
 * bit-exact against the reference. */
int mt_warp_fwd(const float* x, const float* v, const float* grid,
                const mt_warp_layout* l, int32_t align_corners, float* xo,
                float* vo, int32_t* idx, void* stream);
/* gradient of the bilinear warp wrt the grid: dgrid (B,F,H,W,2) */
int mt_warp_bwd_grid(const float* x, const float* grid, const float* dxo,
                     const mt_warp_layout* l, int32_t align_corners,
                     float* dgrid, void* stream);

/* F.interpolate on planes [NP][H][W] -> [NP][Ho][Wo]
 * (utils.py:122-124,536-547,583-593; model_dfpn.py:246-250,521-526).
 * mode 0: nearest; 1: bilinear, align_corners=False.  Every sample is a
 * vector of CI contiguous floats (1 for planar images, 2 for (..,H,W,2)
 * flows); s_x / s_y / s_plane are strides in floats. */
int mt_resize_fwd(const float* src, int64_t s_plane, int64_t s_y, int64_t s_x,
                  int32_t NP, int32_t H, int32_t W, float* dst, int64_t d_plane,
                  int64_t d_y, int64_t d_x, int32_t Ho, int32_t Wo, int32_t CI,
                  int32_t mode, void* stream);
/* backward of the bilinear mode: every element of dsrc is WRITTEN (gather form, no atomics: bit-reproducible) */
int mt_resize_bilinear_bwd(const float* ddst, int64_t d_plane, int64_t d_y,
                           int64_t d_x, int32_t NP, int32_t Ho, int32_t Wo,
                           float* dsrc, int64_t s_plane, int64_t s_y, int64_t s_x,
                           int32_t H, int32_t W, int32_t CI, void* stream);

/* Masked / weighted L1 reductions (LossesUtils.masked_l1 utils.py:138-169,
 * F.l1_loss model_dfpn.py:257, utils.py:188-189).
 *   acc[0] += sum_i w[b(i)] * | a_i*m_i - b_i*m_i |
 *   acc[1] += sum over the mask tensor of w[b]*m      (if count_mask)
 * data is [B][C][S], mask [B][MC][S] with MC in {1, C} (or NULL = ones),
 * w is [B] (or NULL = ones).  `acc` must be zeroed by the caller. */
int mt_l1_fwd(const float* a, const float* b, const float* mask, const float* w,
              int32_t B, int32_t C, int32_t MC, int64_t S, int32_t count_mask,
              float* acc, void* stream);
/* da_i = sa * scale[0] * w[b] * sign(a_i*m_i - b_i*m_i) * m_i   (if da)
 * db_i = -(same with sb)                                      (if db)  */
int mt_l1_bwd(const float* a, const float* b, const float* mask, const float* w,
              int32_t B, int32_t C, int32_t MC, int64_t S, const float* scale,
              float* da, float* db, int32_t accumulate, void* stream);
/* Sobel-gradient L1 (LossesUtils.grad utils.py:193-224) on planes [NP][H][W]:
 * acc[0] += sum |sobel_h(a)-sobel_h(b)| + |sobel_v(a)-sobel_v(b)| */
int mt_sobel_l1_fwd(const float* a, const float* b, int32_t NP, int32_t H,
                    int32_t W, float* acc, void* stream);
int mt_sobel_l1_bwd(const float* a, const float* b, int32_t NP, int32_t H,
                    int32_t W, const float* scale, float* da, void* stream);

/* Adam (torch.optim.Adam defaults; model_dfpn.py:302, model_chn.py:385) over a
 * flat parameter range. step_size = lr/bias_correction1,
 * inv_bc2_sqrt = 1/sqrt(bias_correction2). */
int mt_adam_step(float* p, const float* g, float* m, float* v, int64_t n,
                 float beta1, float beta2, float step_size, float inv_bc2_sqrt,
                 float eps, float grad_scale, void* stream);

/* The same update with the step counter and the learning rate on the device
 * (state = {step, lr}: step is incremented first), so that a captured hipGraph
 * of the whole training step can be replayed. */
int mt_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n,
                     float beta1, float beta2, float eps, float grad_scale,
                     float* state, void* stream);

/* ---- either side of the step: dataset synthesis and validation metrics ----
 * TransformsUtils.dilate (utils.py:618-640: cv2.dilate, MORPH_CROSS element of kh x kw, odd,
 * `iterations` times, default border) on the mask planes m (B,1,F,H,W), fused with the masked
 * fill of dataset.py:168-169, x = (1 - m') * y + m' * fill3[c] on (B,3,F,H,W) tensors
 * (y = x = fill3 = NULL: dilation only).  iterations * (k / 2) <= 16.  m_out must not alias m. */
int mt_mask_dilate_fill(const float* m, const float* y, const float* fill3, int32_t B, int32_t F,
                        int32_t H, int32_t W, int32_t kh, int32_t kw, int32_t iterations,
                        float* m_out, float* x, void* stream);
/* MeasuresUtils.psnr (utils.py:407-429 -> skimage 0.18.3 peak_signal_noise_ratio): per image n of
 * N contiguous images of per_image floats, out[3n..3n+2] = {sum (y_hat - y)^2, min(y), max(y)}
 * (float32 differences, float64 sum, as skimage); the caller derives data_range from the range of
 * y and forms 10 log10(data_range^2 / mse).  out: 3 N doubles (initialised here). */
int mt_psnr_sums(const float* y_hat, const float* y, int32_t N, int64_t per_image, double* out,
                 void* stream);
/* MeasuresUtils.ssim (utils.py:431-448 -> skimage 0.18.3 structural_similarity, multichannel):
 * per plane (image, channel) of H x W floats, out[plane] = sum of the SSIM map over the
 * (H - win + 1) x (W - win + 1) pixels whose window lies inside the image (skimage crops the rest),
 * uniform win x win window, sample covariance, float64 arithmetic, C1 = (K1 R)^2, C2 = (K2 R)^2.
 * out: `planes` doubles (zeroed here). */
int mt_ssim_sums(const float* a, const float* b, int32_t planes, int32_t H, int32_t W, int32_t win,
                 double data_range, double K1, double K2, double* out, void* stream);

/* ---- data-parallel gradient exchange (SURVEY.md 5 "distributed comm backend", 8e) ----
 * The reference has no collective call site: with `--gpus N` Lightning's DDP would
 * all-reduce the gradients after loss.backward() (model_dfpn.py:135-158,
 * model_chn.py:256-280 are the steps it wraps).  Here the exchange is one in-place fp32
 * SUM all-reduce of the flat gradient buffer over an RCCL communicator (xGMI) that the
 * library owns: one process per GPU, rank 0 calls mt_comm_unique_id and hands the 128
 * bytes to every rank out of band (torch.distributed's store: bootstrap only), every
 * rank calls mt_comm_init with its HIP device current.  RCCL is bound at run time from
 * `rccl_path` (NULL / "" = the loader's default librccl.so.1), so the library and every
 * other entry point work without it.  mt_allreduce_sum_f32 enqueues on `stream` and
 * returns; nothing synchronises with the host. */
int mt_comm_unique_id(const char* rccl_path, void* id128);
int mt_comm_init(const char* rccl_path, const void* id128, int32_t rank, int32_t world, void** comm);
int mt_comm_info(void* comm, int32_t* rank, int32_t* world);
int mt_comm_destroy(void* comm);
int mt_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* MT355_H */
