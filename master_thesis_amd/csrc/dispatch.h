// Kernel selection of the conv engine, decided in ONE place (dispatch.cpp).
//
// Every launch path (mt_conv_run, mt_wgrad_run, the weight-pack routines and their size queries) and the two plan
// queries of the C ABI (mt_conv_plan_query, mt_wgrad_plan_query: what the host side packs for, sizes its workspaces
// from and labels its profiles with) ask the functions below; nothing else in the library tests a shape or an
// environment switch to choose a kernel.  Rounds 1-3 kept a Python mirror of these rules in the host package
// (VERDICT round 3, weak #11 / #12): pack layout and kernel agreed only because both sides read the same variable.
#pragma once
#include <hip/hip_runtime.h>
#include "mt355.h"

// Lab A/B switches, read from the environment ONCE, at first use (the only getenv calls of the library).
struct MtSwitches {
  bool halo_rb;        // MT355_HALO_RB=0: LDS-staged weight tiles everywhere (no register-B kernels)
  bool wino;           // MT355_WINO=0: no 1-D Winograd F(2,3) (direct register-B kernel on the 3 x 3 layers)
  bool winox;          // MT355_WINOX=0: layers with <= 64 output channels stay on the pixel-split kernels
  bool wino16;         // MT355_WINO16=0: 8 x 16 tiles for the 128-wide F(2,3) kernel
  bool wino25;         // MT355_WINO25=0: direct kernel on the 5 x 5 layers
  bool wino43;         // MT355_WINO43=0: F(2,3) instead of F(4,3) on the 128-wide 3 x 3 layers
  bool planes;         // MT355_PLANES=0: stride-2 layers on the generic gather kernel
  bool tiny;           // MT355_TINY=0: no fp32 VALU kernels for layers with <= 4 channels on one side
  bool tiny_out;       // MT355_TINY_OUT=0: ... for the 128 -> 1 / 2 channel layers only
  bool wgrad_pc;       // MT355_WGRAD_PC=0: stride-1 weight gradients on the single-role kernel
  bool wgrad_pc5;      // MT355_WGRAD_PC5=0: 5 x 5 weight gradients on the single-role kernel (rows {0,1,2} | {3,4})
  bool wgrad_s2_halo;  // MT355_WGRAD_S2_HALO=0: stride-2 weight gradients on the generic kernel
  bool wgrad3_tall;    // MT355_WGRAD3_TALL=0: 4-row pixel tiles for the single-role 3 x 3 weight gradient
  bool wgrad5_rows;    // MT355_WGRAD5_ROWS=1: one kernel row per block for the single-role 5 x 5 weight gradient
  bool wgrad_thin;     // MT355_WGRAD_THIN=0: no four-taps-per-tile kernel for 16 gathered channels
};
const MtSwitches& mt_switches();

// Input channels per LDS stage of igemm_halo_kernel and of its weight tiles: 16 for the three-term split and for
// layers whose (padded) input has at most 16 channels, else 32.
#ifndef MT_HALO_THIN_CK
#define MT_HALO_THIN_CK 1      // (0: 32-channel stages for every two-term layer -- A/B builds)
#endif
__host__ __device__ inline int halo_ck(int cin_p, int ns) { return (ns == 3 || (MT_HALO_THIN_CK && cin_p <= 16)) ? 16 : 32; }

// ---- shape rules of the stride-1 halo-tile family (the weight pack is chosen by the same rules: they may look at
// channel counts, taps and precision only, never at the image size) ----------------------------------------------
bool halo_rb_wanted(int cin_p, int cout_p);                          // register-B family: >= 2 K chunks, > 32 outputs
bool halo_winox_wanted(int cin_p, int cout_p, int T);                // xi-split F(2,3): <= 32 outputs (64 from 128 inputs on)
bool halo_wino_wanted(int cin_p, int cout_p, int T);                 // F(2,3)-transformed weights (any of its kernels)
bool halo_wino43_wanted(int cin_p, int cout_p, int T, int f16);      // F(4,3): 128-wide tiles, fp16 split
bool halo_wino25_wanted(int cin_p, int cout_p, int T, int f16);      // F(2,5): 128-wide tiles, whole chunks, fp16 split
bool planes_shape_ok(int K, int pad, int cin_p, int cout_p, int precision);   // stride-2 window kernels

enum MtHaloKernel {
  MT_HK_NONE = 0, MT_HK_WINO43, MT_HK_WINOX, MT_HK_WINO16, MT_HK_HALO_WINO, MT_HK_WINO25, MT_HK_RB, MT_HK_LDS
};
struct MtHaloSel {
  int kernel;   // MtHaloKernel
  int bn;       // output channels per block tile: 128, 64 or 32
  int ck;       // MT_HK_LDS: channels per LDS stage (16 / 32)
};
// The kernel a stride-1 3 x 3 / 5 x 5 "same" layer runs on.  NS = terms of the split (2 / 3), fmt = 1 for fp16 terms;
// x32: the input image is addressable with 32-bit byte offsets (the xi-split kernel's buffer descriptor).
MtHaloSel mt_halo_select(int KS, int NS, int fmt, int cin_p, int cout_p, int Ho, bool x32);

// ---- weight gradients -----------------------------------------------------------------------------------------------
struct MtWgradSel {
  int halo;       // 1: the halo-tile family takes it (else the generic gather kernels)
  int pc;         // producer / consumer kernel (512-thread blocks)
  int thin;       // 16 gathered channels, four taps per B tile
  int tall;       // single-role kernel with 8-row tiles
  int rows5;      // single-role 5 x 5: one kernel row per block (five times the blocks)
  int trows;      // pixel-tile rows (4 / 8)
  int d_terms;    // 16-bit terms of D that enter (1 / 2): the caller's wish where the kernel has the form, else 2
  int launches;   // kernel launches per call (5 x 5: 3 for pc, 2 for the single-role kernel)
  int tile_list;  // 1: the kernel takes mt_wgrad.tiles (3 x 3 producer / consumer)
};
inline int wgpc_tr(int d_terms, int KS, int Hd) { return (d_terms == 1 && KS == 3 && Hd <= 16 && (Hd & 7) == 0) ? 8 : 4; }
MtWgradSel mt_wgrad_select(const mt_wgrad* d);
int mt_wgrad_tiles_check(const mt_wgrad* d, const MtWgradSel& sel);   // MT_OK, or the error of a malformed / unsupported tile list
