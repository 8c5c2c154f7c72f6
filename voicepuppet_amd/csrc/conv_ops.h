// Host-side construction of the implicit-GEMM argument blocks for one convolution layer:
// forward, backward-data, backward-weight, and the weight-packing descriptors that go with them.
// Geometry follows the TF ops the reference calls:
//   conv   : tf.layers.conv2d, kernel HWIO [k,k,Cin,Cout], symmetric pad      (pixrefer.py:61-74)
//   deconv : tf.layers.conv2d_transpose k4 s2 "same", kernel HWOI [4,4,Cout,Cin] (pixrefer.py:85)
//            out[n,2i+kh-1,2j+kw-1,co] += in[n,i,j,ci] * W[kh,kw,co,ci]
#pragma once
#include <stdlib.h>
#include <string.h>

#include "launch.h"

namespace vp {

struct ConvGeomX : ConvGeom {
  int CoutT;          // channel count of the output-side tensors (Cout, or 8 when Cout < 8)
};

inline ConvGeomX make_geom(int kind, int ks, int stride, int pad, int N, int Hin, int Win, int Cin, int Cin_real, int Cout) {
  ConvGeomX g;
  g.kind = kind; g.ks = ks; g.stride = stride; g.pad = pad;
  g.N = N; g.Hin = Hin; g.Win = Win;
  if (kind == 0) { g.Hout = (Hin + 2 * pad - ks) / stride + 1; g.Wout = (Win + 2 * pad - ks) / stride + 1; }
  else { g.Hout = 2 * Hin; g.Wout = 2 * Win; }
  g.Cin = Cin; g.Cin_real = Cin_real; g.Cout = Cout;
  g.CoutT = Cout < 8 ? 8 : Cout;
  return g;
}

inline int kc_elems(int is_bf16) { return is_bf16 ? 32 : 16; }

// vp_tune("igemm_small_grid", n): a launch whose 128 x 128 tiling has at most n blocks per class takes the 64-row x 128-pixel tile (twice
// the blocks: a grid below the CU count runs at the per-CU L2 -> LDS fill rate of the CUs it occupies; 1.5 x the fill on 2 x the CUs)
// cost model of the weight-gradient K split (plan_wgrad): [0] fixed cost of a block in K iterations x 10, [1] cost of a slab x 100
inline int& wgrad_cost_knob(int i) { static int v[3] = {80, 15, 150}; return v[i]; }    // [2]: round 6 (0 before): bs 32 -0.2 ms, bs 8 -0.1, bs 4 -0.1 (profiles/r06_ab_wgrad_split_cost.txt)
// [0] few-pixel kernel: blocks its K split aims at; [1] / [2] implicit GEMM: most K splits, fewest K chunks per split; [3] weight gradient: resident blocks per round of its cost model
inline int& plan_misc_knob(int i) { static int v[4] = {384, 8, 4, 512}; return v[i]; }
inline int& igemm_small_grid_knob() { static int v = 128; return v; }
// grid caps of the thin-layer tile kernels (conv3x3_cout8_tile / deconv_cout8_tile / deconv_cout4_tile): vp_tune("thin_blocks_cout8" /
// "thin_blocks_dcout8" / "thin_blocks_cout4") and of conv_cin8_kernel ("thin_blocks_cin8").  The blocks are persistent and build a
// fragment-ordered weight image in LDS first: with the round-5 caps (4096 / 4096 / 2048) a block of layer_1's backward-data launch
// owned two tiles and the prologue was most of its time (EXPERIMENTS.md 0.8)
inline int& thin_blocks_knob(int i) { static int v[4] = {1024, 512, 512, 512}; return v[i]; }

// tile choice for an igemm producing `rows` channels over P pixels
inline int pick_igemm_cfg(int rows, int P, int Kpad = 0) {
  if (P >= 96) {
    if (rows % 128 == 0 && igemm_small_grid_knob() > 0 && (long long)((P + 127) / 128) * (rows / 128) <= igemm_small_grid_knob()) return 1;
    // 256x256 (8-wave tile) runs one block per CU: its prologue / epilogue are exposed, only long K loops amortise them
    if (rows % 256 == 0 && P >= 256 * 256 && Kpad >= 4096) return 7;
    if (rows % 128 == 0 && P >= 256 * 512) return 6;           // 128x256, 8 waves
    if (rows % 128 == 0) return 0;
    if (rows % 64 == 0) return 1;
    if (rows <= 16) return 2;
    return rows > 64 ? 0 : 1;
  }
  if (rows <= 64) return 5;
  return P > 16 ? 3 : 4;
}

constexpr int IGEMM_SPLITK_TARGET_DEFAULT = 64;   // rounds 2-5: 128
inline int& igemm_splitk_target_knob() { static int v = IGEMM_SPLITK_TARGET_DEFAULT; return v; }   // vp_tune("igemm_splitk_target"): resident blocks a K split aims at
inline int pick_igemm_splitk(int blocks, int nchunk) {
  // round-2 sweep (scripts/ab.sh with VP_SPLITK_TARGET / _MAX / _MINCHUNK): target 128 / cap 8 / at least 4 chunks per split: 8.79 vs 8.97 ms at batch 32, 2.84 vs 3.00 ms at
  // batch 4 against the round-1 setting 512 / 32 / 2 - the slab reduce and the short blocks cost more than the idle CUs
  const int target = igemm_splitk_target_knob();      // resident blocks aimed at
  const int cap = plan_misc_knob(1);           // vp_tune("igemm_splitk_cap"), default 8
  const int minchunk = plan_misc_knob(2);      // K chunks per split at least: vp_tune("igemm_splitk_minchunk"), default 4
  if (blocks >= 256 || blocks >= target) return 1;
  int s = (target + blocks - 1) / blocks;
  if (s > nchunk / minchunk) s = nchunk / minchunk;
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

inline void set_single_src(PixSrc& x, const void* p, int C, const float* a, const float* b, int act, int group_n) {
  x.ptr[0] = p; x.ptr[1] = nullptr; x.C[0] = C; x.C[1] = 0;
  x.aff_a[0] = a; x.aff_b[0] = b; x.aff_a[1] = nullptr; x.aff_b[1] = nullptr;
  x.act = act; x.group_n = group_n > 0 ? group_n : (1 << 30);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
struct IgemmPlan {
  IgemmArgs a;
  int cfg;
  PackDesc pack;
  size_t pack_elems;      // elements of the packed weight block
  size_t partial_bytes;
};

inline void finish_igemm(IgemmPlan& p, int rows, int is_bf16) {
  IgemmArgs& a = p.a;
  const int P = a.N * a.Hg * a.Wg;
  a.Kpad = round_up(a.ntaps * a.Cin, kc_elems(is_bf16));
  p.cfg = pick_igemm_cfg(rows, P, a.Kpad);
  // float32 matrix products (one tap: BFMNet's 1x1 convolutions, the DFT): a 128x128 block is 4x the MFMA time of a bf16 one, so a grid
  // of 257-511 blocks costs two full rounds of the 256 CUs.  64x128 tiles (two resident per CU, half the time each) quantise finer:
  // rounds x rows 3 x 64 against 2 x 128 for 300 blocks (scripts/mm_bench.py: 19200 x 1152 x 256 forward 140 -> 109 us)
  if (!is_bf16 && a.ntaps == 1 && p.cfg == 0) {
    const long long b128 = (long long)((P + 127) / 128) * (rows / 128), b64 = 2 * b128;
    if (((b64 + 255) / 256) * 64 < ((b128 + 255) / 256) * 128) p.cfg = 1;
  }
  int bc, bp;
  igemm_tile(p.cfg, &bc, &bp);
  a.CoutPad = round_up(rows, bc);
  if (a.ntaps == 1) { a.log2Cin = 30; a.cin_mask = 0x3fffffff; }   // 1x1: any channel count
  else { a.log2Cin = ilog2(a.Cin); a.cin_mask = a.Cin - 1; }
  const int blocks = ((P + bp - 1) / bp) * (a.CoutPad / bc) * a.nclass;
  a.splitk = pick_igemm_splitk(blocks, a.Kpad / kc_elems(is_bf16));
  p.partial_bytes = a.splitk > 1 ? (size_t)a.nclass * a.splitk * P * a.CoutPad * sizeof(float) : 0;
  // row permutation inside 64-row blocks (IgemmArgs::rowperm) where the tile's waves own whole 64-row blocks
  {
    const bool tc4 = p.cfg == 0 || p.cfg == 1 || p.cfg == 6 || p.cfg == 7 || p.cfg == 8 || p.cfg == 9;
    a.rowperm = (tc4 && rows % 64 == 0) ? 1 : 0;
    p.pack.perm = a.rowperm;
  }
  // packed rows are padded to the largest channel tile so the tile choice may vary with the batch
  a.wp_rows = round_up(rows, 128);
  p.pack.rows_pad = a.wp_rows;
  p.pack.Kpad = a.Kpad;
  p.pack.kc = kc_elems(is_bf16);
  p.pack_elems = (size_t)a.nclass * a.wp_rows * a.Kpad;
}

// Patch kernel (conv_patch.hip) for stride-1 convolutions and their backward-data passes: a plan-time decision because the packed
// weights are stored with PackDesc::kswap.  single_src: the layer reads one tensor (no virtual concat).
// tuning knobs that tests and experiments may change at run time (vp_tune): which patch tiles are allowed (bit 0: 256-row, bit 1:
// 128-row, bit 2: 64-row) and the smallest grid worth one 8-wave block per CU
inline int& patch_tiles_knob() { static int v = 7; return v; }
inline int& patch_small_knob() { static int v = 3; return v; }   // two-blocks-per-CU tiles (see patch_tile_pixels)
inline int& patch_longk_knob() { static int v = 1; return v; }      // 1: long-K 512-row layers stay on the 256x256 tile
inline int& patch3_knob() { static int v = 1; return v; }   // 3x3 layers on the unrolled patch kernel (conv_patch3.hip)
inline int& wgrad_big_knob() { static int v = 1; return v; }   // 256 x 256 tile of wgrad_tr.hip where the grid still fills the CUs
inline int& patch4_knob() { static int v = 1; return v; }   // 4x4 stride-1 backward-data passes on the unrolled patch kernel (conv_patch3.hip, KW = 4)
inline int& c64_knob() { static int v = 1; return v; }   // 64 -> 64 channel 3x3 layers on the register-resident-weights kernel (conv_c64.hip)
inline int& cout1_knob() { static int v = 256; return v; }  // most blocks (= partial rows, <= 256: the tables plan_net carves) per batch-norm group of the one-output-channel backward-data kernel (conv_cout1.hip); 0: the generic kernels
inline int& wgrad1_rows_knob() { static int v = 512; return v; }  // most blocks (= slabs) of cout1_wgrad_kernel: vp_tune("cout1_wgrad_rows")
inline int& dc64_knob() { static int v = 1; return v; }  // 128 -> 64 channel transposed-conv classes on conv_dc64.hip
inline int& s2c64_knob() { static int v = 512; return v; }   // 64 -> 128 channel 4x4 / stride-2 convolutions on conv_s2c64.hip from this many 4 x 16-pixel tiles (0: off)
inline int& s2c64_pair_knob() { static int v = 1; return v; }   // the two-output backward-data of merged2_decoder_2 on conv_s2c64.hip (0: the generic two-output GEMM)
constexpr int PATCH_MIN_BLOCKS_DEFAULT = 192;     // round 6 (with the K-split target below): 4 frames 2.34-2.37 -> 2.28 ms, 8 frames 3.00 -> 2.97, 32 frames +-0 (profiles/r06_ab_plan_heuristics.txt); rounds 2-5: 384
inline int& patch_minblk_knob() { static int v = PATCH_MIN_BLOCKS_DEFAULT; return v; }

// pixel tile of a patch-kernel plan: bp = 512 -> 16 x 32, 256 -> 16 x 16, 128 -> 8 x 16
inline void patch_tile_hw(int bp, int* th, int* tw) { *th = bp == 128 ? 8 : 16; *tw = bp == 512 ? 32 : 16; }
// pixels of the tile chosen for `bc` channel rows: knob bit 0: two-blocks-per-CU 16 x 16 tiles for the 128- / 64-row variants
// (else 16 x 32, one block per CU); bit 1: 8 x 16 tiles, two blocks per CU, for the 256-row variant (else 16 x 16, one block)
inline int patch_tile_pixels(int bc) {
  const int k = patch_small_knob();
  if (bc == 256) return (k & 2) ? 128 : 256;
  return (k & 1) ? 256 : 512;
}

// channel rows of the patch tile for a layer of `rows` output channels: the largest enabled tile that divides it (0: none)
inline int patch_tile_rows(int rows, int on) {
  if ((on & 1) && rows % 256 == 0) return 256;
  if ((on & 2) && rows % 128 == 0) return 128;
  if ((on & 4) && rows % 64 == 0) return 64;
  return 0;
}

// any_grid: skip the minimum-block rule (a half-batch launch follows the kernel choice of the full-batch plan of its layer)
inline bool plan_patch_eligible(const IgemmPlan& p, int rows, int is_bf16, bool single_src, bool any_grid = false) {
  const int on = patch_tiles_knob();
  const int minblk = any_grid ? 0 : patch_minblk_knob();
  const IgemmArgs& a = p.a;
  const int kc = kc_elems(is_bf16);
  if (!on || !single_src || a.nclass != 1 || a.sh != 1 || a.sw != 1 || a.os != 1 || a.ntaps < 9 || a.Cin % kc || a.Cin < kc) return false;
  if (rows % 64 || a.Cout % 8 || a.ldY % 8 || a.Hg < 16 || a.Wg < 16 || a.Hof != a.Hg || a.Wof != a.Wg) return false;
  // >= 512 output channels with a long K loop (discriminator layer_4 forward): measured faster on the wave-specialised 256x256 tile
  // (0.32 vs 0.36 ms at N = 32), whose exposed prologue / epilogue that loop amortises
  // (round 5: 4x4 taps run the unrolled kernel - conv_patch3.hip, KW = 4: 1250 TF on the same layer - and stay here)
  const bool p4 = patch4_knob() && is_bf16 && a.ntaps == 16 && rows % 128 == 0;
  if (!p4 && patch_longk_knob() && rows >= 512 && rows % 256 == 0 && a.ntaps * a.Cin >= 4096 && (long long)a.N * a.Hg * a.Wg >= 256 * 256) return false;
  const int bc = p4 ? 128 : patch_tile_rows(rows, on);
  if (!bc) return false;
  int th, tw;
  patch_tile_hw(p4 ? 256 : patch_tile_pixels(bc), &th, &tw);
  const long long blocks = (long long)a.N * ((a.Hg + th - 1) / th) * ((a.Wg + tw - 1) / tw) * (rows / bc);
  // 4x4 taps: from 512 blocks (batch 8: 384 blocks measured 0.01-0.02 ms slower than the 256 x 256 wave-specialised tile); from 256 - one per
  // CU - where K >= 8192 amortises the block's prologue (layer_4's generator-loss pass at batch 32)
  const long long p4min = (long long)a.ntaps * a.Cin >= 8192 ? 256 : 512;
  if (p4 && !any_grid && patch_minblk_knob() > 1) {
    if (blocks < p4min) return false;
  } else if (blocks < minblk) return false;
  if ((size_t)a.N * a.Hin * a.Win * a.Cin * (is_bf16 ? 2 : 4) >= 0x70000000ull) return false;     // lane offsets of the buffer loads
  // taps on a regular grid
  int ks = 0;
  while (ks * ks < a.ntaps) ++ks;
  if (ks * ks != a.ntaps) return false;
  const int sh = a.taps[0].dh[ks] - a.taps[0].dh[0], sw = a.taps[0].dw[1] - a.taps[0].dw[0];
  if ((sh != 1 && sh != -1) || (sw != 1 && sw != -1)) return false;
  for (int t = 0; t < a.ntaps; ++t)
    if (a.taps[0].dh[t] != a.taps[0].dh[0] + (t / ks) * sh || a.taps[0].dw[t] != a.taps[0].dw[0] + (t % ks) * sw) return false;
  return true;
}
inline void plan_make_patch(IgemmPlan& p, int rows, int is_bf16) {
  IgemmArgs& a = p.a;
  // 4x4 taps (bf16): the unrolled kernel's 128-row x 16 x 16-pixel tile (conv_patch3.hip patch4_eligible); the plan names the tile the
  // batch statistics are chunked by
  const bool p4 = patch4_knob() && is_bf16 && a.ntaps == 16 && rows % 128 == 0;
  const int bc = p4 ? 128 : patch_tile_rows(rows, patch_tiles_knob());
  int ks = 0;
  while (ks * ks < a.ntaps) ++ks;
  a.patch = 1; a.p_kw = ks;
  a.p_dhf = a.taps[0].dh[0]; a.p_dwf = a.taps[0].dw[0];
  a.p_dhs = a.taps[0].dh[ks] - a.taps[0].dh[0]; a.p_dws = a.taps[0].dw[1] - a.taps[0].dw[0];
  const int bp = p4 ? 256 : patch_tile_pixels(bc);
  p.cfg = bc == 256 ? (bp == 128 ? 15 : 10) : (bc == 128 ? (bp == 256 ? 13 : 11) : (bp == 256 ? 14 : 12));
  a.CoutPad = round_up(rows, bc);
  a.splitk = 1;
  p.partial_bytes = 0;
  a.rowperm = 1;
  p.pack.perm = 1;
  p.pack.kswap = 1;
}

// The four parity classes of a 4x4 stride-2 transposed conv (deconv forward, conv backward-data) on the unrolled 2x2-tap patch
// kernel (conv_patch2.hip).  c0 / c1: channels of the one or two source tensors of the GEMM's pixel operand.
inline int& patch_xcd_knob() { static int v = 0; return v; }
inline int& patch2_knob() { static int v = 1; return v; }
inline bool plan_patch2_eligible(const IgemmPlan& p, int rows, int is_bf16, int c0, int c1) {
  const IgemmArgs& a = p.a;
  const int kc = kc_elems(is_bf16);
  if (!patch2_knob() || a.nclass != 4 || a.ntaps != 4 || a.os != 2 || a.sh != 1 || a.sw != 1) return false;
  if (c0 + c1 != a.Cin || c0 % (2 * kc) || c1 % (2 * kc) || c0 < 2 * kc) return false;
  if (rows % 64 || rows > 128 || a.Cout % 8 || a.ldY % 8 || a.Hg < 16 || a.Wg < 16 || a.Hof != 2 * a.Hg || a.Wof != 2 * a.Wg) return false;
  const long long blocks = (long long)a.N * ((a.Hg + 15) / 16) * ((a.Wg + 15) / 16) * 4;
  if (blocks < patch_minblk_knob()) return false;
  const size_t cmax = c0 > c1 ? c0 : c1;
  if ((size_t)a.N * a.Hin * a.Win * cmax * (is_bf16 ? 2 : 4) >= 0x70000000ull) return false;     // lane offsets of the buffer loads
  for (int cls = 0; cls < 4; ++cls)
    for (int t = 0; t < 4; ++t)          // patch origin = tap 3, tap t one step up / left per bit (conv_patch2.hip)
      if (a.taps[cls].dh[t] != a.taps[cls].dh[3] + 1 - (t >> 1) || a.taps[cls].dw[t] != a.taps[cls].dw[3] + 1 - (t & 1)) return false;
  return true;
}
inline void plan_make_patch2(IgemmPlan& p, int rows, int is_bf16) {
  IgemmArgs& a = p.a;
  const int bc = rows % 128 == 0 ? 128 : 64;
  a.patch = 2; a.p_kw = 2;
  p.cfg = bc == 128 ? 13 : 14;
  a.CoutPad = round_up(rows, bc);
  a.splitk = 1;
  p.partial_bytes = 0;
  a.rowperm = 1;
  p.pack.perm = 1;
  p.pack.kswap = 1;
}

// Few-pixel kernel (conv_smallp.hip): layers whose pixel count per parity class is so small that the layer is a stream of weights
// (the generator's 1x1 .. 16x16 bottleneck).  A plan-time decision: the packed rows are not permuted, the slab / counter sizes differ.
// c0 / c1: channels of the one or two source tensors.
inline int& smallp_knob() { static int v = 256; return v; }   // largest pixel count per class (0: off)
inline bool plan_smallp_eligible(const IgemmPlan& p, int rows, int is_bf16, int c0, int c1) {
  const IgemmArgs& a = p.a;
  const int kc = kc_elems(is_bf16);
  const long long Pc = (long long)a.N * a.Hg * a.Wg;
  if (smallp_knob() <= 0 || Pc > smallp_knob() || a.ntaps > 16 || a.nclass > 4) return false;
  if (rows % 32 || a.Cout != rows || a.ldY % 8) return false;
  if (a.Cin < kc || (a.Cin & (a.Cin - 1)) || c0 + c1 != a.Cin || c0 % kc || c1 % kc) return false;
  if ((long long)a.N * a.Hin * a.Win * (c0 > c1 ? c0 : c1) >= (1ll << 30)) return false;       // 32-bit element offsets in the loader
  return true;
}
// counters a smallp plan needs: one per tile + one per channel tile
inline int smallp_counters(const IgemmArgs& a) {
  const int PT = a.sp_npt * 16;
  const int Pc = a.N * a.Hg * a.Wg;
  return (a.CoutPad / 32) * a.nclass * ((Pc + PT - 1) / PT) + a.CoutPad / 32;
}
inline void plan_make_smallp(IgemmPlan& p, int rows, int is_bf16) {
  IgemmArgs& a = p.a;
  const int kc = kc_elems(is_bf16);
  const int Pc = a.N * a.Hg * a.Wg;
  a.patch = 3;
  a.sp_npt = Pc <= 16 ? 1 : (Pc <= 32 ? 2 : 4);
  p.cfg = a.sp_npt == 1 ? 16 : (a.sp_npt == 2 ? 17 : 18);
  a.sp_lcpt = ilog2(a.Cin / kc);
  a.CoutPad = round_up(rows, 32);
  a.rowperm = 0; p.pack.perm = 0; p.pack.kswap = 0;
  // taps that reach the image for at least one pixel of the class (rows and columns are independent)
  int minchunks = 1 << 30;
  for (int cls = 0; cls < a.nclass; ++cls) {
    unsigned m = 0;
    for (int t = 0; t < a.ntaps; ++t) {
      bool okh = false, okw = false;
      for (int q = 0; q < a.Hg && !okh; ++q) { const int ih = q * a.sh + a.taps[cls].dh[t]; okh = ih >= 0 && ih < a.Hin; }
      for (int r = 0; r < a.Wg && !okw; ++r) { const int iw = r * a.sw + a.taps[cls].dw[t]; okw = iw >= 0 && iw < a.Win; }
      if (okh && okw) m |= 1u << t;
    }
    a.sp_mask[cls] = (unsigned short)m;
    const int chunks = __builtin_popcount(m) * (a.Cin / kc);
    if (chunks < minchunks) minchunks = chunks;
  }
  // K splits over blocks: aim at >= 384 blocks, keep >= 2 chunks per wave (4 waves per block share a block's K range)
  const int PT = a.sp_npt * 16;
  const int tiles = (a.CoutPad / 32) * a.nclass * ((Pc + PT - 1) / PT);
  const int target = plan_misc_knob(0);          // vp_tune("smallp_split_target"), default 384
  int s = (target + tiles - 1) / tiles;
  if (s > minchunks / 8) s = minchunks / 8;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  a.splitk = s;
  p.partial_bytes = s > 1 ? (size_t)s * tiles * PT * 32 * sizeof(float) : 0;
}

// 4x4 / stride-2 / pad-1 convolutions from ONE 64-channel tensor to 128 channels on conv_s2c64.hip (weights resident in registers, batch
// statistics per block).  A plan-time decision: no K split; the generic packing (rows permuted inside 64-row blocks) is what it reads.
// c0 / c1: channels of the source tensors.
// rows = 128 (forward layers) or 256 (the two-output backward-data of a 256 -> 64 transposed convolution: split at 128, pixel stride 128)
inline bool plan_s2c64_eligible(const IgemmPlan& p, int rows, int is_bf16, int c0, int c1) {
  const IgemmArgs& a = p.a;
  if (s2c64_knob() <= 0 || !is_bf16 || a.nclass != 1 || a.ntaps != 16 || a.os != 1 || a.sh != 2 || a.sw != 2) return false;
  if ((rows != 128 && rows != 256) || a.Cout != rows || a.ldY != 128 || a.Cin != 64 || c0 != 64 || c1 != 0 || a.Kpad != 1024) return false;
  if (a.Hg % 4 || a.Wg % 16 || a.Hin != 2 * a.Hg || a.Win != 2 * a.Wg || a.Hof != a.Hg || a.Wof != a.Wg) return false;
  for (int t = 0; t < 16; ++t)
    if (a.taps[0].dh[t] != (t >> 2) - 1 || a.taps[0].dw[t] != (t & 3) - 1) return false;
  // (default: two 4 x 16-pixel tiles per block - below, the 256 KB weight load of each of the <= 256 blocks is not amortised)
  if (a.N * (a.Hg / 4) * (a.Wg / 16) < s2c64_knob()) return false;
  return (size_t)a.N * a.Hin * a.Win * 64 * 2 < 0x70000000ull;
}
inline void plan_make_s2c64(IgemmPlan& p) {
  IgemmArgs& a = p.a;
  a.patch = 4;
  p.cfg = 0;                 // (128 x 128: the class name's tile; the kernel has its own)
  a.CoutPad = a.Cout; a.wp_rows = a.Cout; p.pack.rows_pad = a.Cout;
  p.pack_elems = (size_t)a.wp_rows * a.Kpad;
  a.splitk = 1; p.partial_bytes = 0;
  a.rowperm = 1; p.pack.perm = 1; p.pack.kswap = 0;
}

// x (PixSrc, total channels g.Cin) -> y [N,Hout,Wout,ldY]
inline IgemmPlan plan_fwd(const ConvGeomX& g, size_t w_off, int is_bf16) {
  IgemmPlan p;
  memset(&p, 0, sizeof(p));
  IgemmArgs& a = p.a;
  a.N = g.N; a.Hin = g.Hin; a.Win = g.Win; a.Cin = g.Cin; a.cin_real = g.Cin_real;
  a.Cout = g.Cout; a.ldY = g.Cout; a.Hof = g.Hout; a.Wof = g.Wout;
  a.ref_group_n = 1 << 30;
  PackDesc& d = p.pack;
  d.src_off = w_off; d.rows_real = g.Cout; d.C = g.Cin; d.C_real = g.Cin_real;
  if (g.kind == 0) {
    a.Hg = g.Hout; a.Wg = g.Wout; a.sh = a.sw = g.stride; a.os = 1;
    a.ntaps = g.ks * g.ks; a.nclass = 1;
    for (int t = 0; t < a.ntaps; ++t) {
      a.taps[0].dh[t] = (int8_t)(t / g.ks - g.pad); a.taps[0].dw[t] = (int8_t)(t % g.ks - g.pad);
      d.kh[0][t] = (int8_t)(t / g.ks); d.kw[0][t] = (int8_t)(t % g.ks);
    }
    d.s_kh = g.ks * g.Cin_real * g.Cout; d.s_kw = g.Cin_real * g.Cout; d.s_ch = g.Cout; d.s_row = 1;
  } else {
    a.Hg = g.Hin; a.Wg = g.Win; a.sh = a.sw = 1; a.os = 2;
    a.ntaps = 4; a.nclass = 4;
    for (int cls = 0; cls < 4; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      a.o0h[cls] = ph; a.o0w[cls] = pw;
      for (int t = 0; t < 4; ++t) {
        const int ta = t >> 1, tb = t & 1;
        a.taps[cls].dh[t] = (int8_t)(ph - ta); a.taps[cls].dw[t] = (int8_t)(pw - tb);
        d.kh[cls][t] = (int8_t)((1 - ph) + 2 * ta); d.kw[cls][t] = (int8_t)((1 - pw) + 2 * tb);
      }
    }
    d.s_kh = 4 * g.Cout * g.Cin_real; d.s_kw = g.Cout * g.Cin_real; d.s_row = g.Cin_real; d.s_ch = 1;
  }
  d.nclass = a.nclass; d.ntaps = a.ntaps;
  finish_igemm(p, g.Cout, is_bf16);
  return p;
}

// ---------------------------------------------------------------------------------------------
// backward-data for input channels [row0, row0+rows): dY [N,Hout,Wout,CoutT] -> dX [N,Hin,Win,ldX]
// ---------------------------------------------------------------------------------------------
inline IgemmPlan plan_bwd_data(const ConvGeomX& g, size_t w_off, int row0, int rows, int rows_real, int ldX, int is_bf16) {
  IgemmPlan p;
  memset(&p, 0, sizeof(p));
  IgemmArgs& a = p.a;
  a.N = g.N; a.Hin = g.Hout; a.Win = g.Wout; a.Cin = g.CoutT; a.cin_real = g.Cout;
  a.Cout = rows; a.ldY = ldX; a.Hof = g.Hin; a.Wof = g.Win;
  a.ref_group_n = 1 << 30;
  PackDesc& d = p.pack;
  d.rows_real = rows_real; d.C = g.CoutT; d.C_real = g.Cout;
  if (g.kind == 0) {
    d.s_kh = g.ks * g.Cin_real * g.Cout; d.s_kw = g.Cin_real * g.Cout; d.s_row = g.Cout; d.s_ch = 1;
    if (g.stride == 1) {
      a.Hg = g.Hin; a.Wg = g.Win; a.sh = a.sw = 1; a.os = 1;
      a.ntaps = g.ks * g.ks; a.nclass = 1;
      for (int t = 0; t < a.ntaps; ++t) {
        const int kh = t / g.ks, kw = t % g.ks;
        a.taps[0].dh[t] = (int8_t)(g.pad - kh); a.taps[0].dw[t] = (int8_t)(g.pad - kw);
        d.kh[0][t] = (int8_t)kh; d.kw[0][t] = (int8_t)kw;
      }
    } else {   // k4 s2 p1: four input-parity classes with 2x2 taps each
      a.Hg = g.Hout; a.Wg = g.Wout; a.sh = a.sw = 1; a.os = 2;
      a.ntaps = 4; a.nclass = 4;
      for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        a.o0h[cls] = ph; a.o0w[cls] = pw;
        for (int t = 0; t < 4; ++t) {
          const int ta = t >> 1, tb = t & 1;
          a.taps[cls].dh[t] = (int8_t)(ph - ta); a.taps[cls].dw[t] = (int8_t)(pw - tb);
          d.kh[cls][t] = (int8_t)((1 - ph) + 2 * ta); d.kw[cls][t] = (int8_t)((1 - pw) + 2 * tb);
        }
      }
    }
  } else {     // deconv: dX = conv k4 s2 p1 over dY with W[kh,kw,co,ci] read as [co -> K, ci -> rows]
    d.s_kh = 4 * g.Cout * g.Cin_real; d.s_kw = g.Cout * g.Cin_real; d.s_row = 1; d.s_ch = g.Cin_real;
    a.Hg = g.Hin; a.Wg = g.Win; a.sh = a.sw = 2; a.os = 1;
    a.ntaps = 16; a.nclass = 1;
    for (int t = 0; t < 16; ++t) {
      a.taps[0].dh[t] = (int8_t)(t / 4 - 1); a.taps[0].dw[t] = (int8_t)(t % 4 - 1);
      d.kh[0][t] = (int8_t)(t / 4); d.kw[0][t] = (int8_t)(t % 4);
    }
  }
  d.src_off = w_off + (size_t)row0 * d.s_row;
  d.nclass = a.nclass; d.ntaps = a.ntaps;
  finish_igemm(p, rows, is_bf16);
  return p;
}

// ---------------------------------------------------------------------------------------------
// backward-weight
// ---------------------------------------------------------------------------------------------
struct WgradPlan {
  WgradArgs a;
  int cfg;
  size_t partial_bytes;
};

inline int& wgrad_tr_knob() { static int v = 3; return v; }   // LDS-DMA + transpose-read weight gradient (wgrad_tr.hip)

// plain_operands: both tensors are read as stored (no deferred affine / activation): required by the LDS-DMA kernel
inline WgradPlan plan_wgrad(const ConvGeomX& g, int is_bf16, bool plain_operands = true) {
  WgradPlan p;
  memset(&p, 0, sizeof(p));
  WgradArgs& a = p.a;
  a.N = g.N;
  a.ntaps = g.ks * g.ks;
  for (int t = 0; t < a.ntaps; ++t) { a.taps.dh[t] = (int8_t)(t / g.ks - g.pad); a.taps.dw[t] = (int8_t)(t % g.ks - g.pad); }
  if (g.kind == 0) {   // G = layer input, D = dY
    a.Gc = g.Cin; a.Greal = g.Cin_real; a.Hgin = g.Hin; a.Wgin = g.Win;
    a.Dc = g.CoutT; a.Dreal = g.Cout;
    a.Hb = g.Hout; a.Wb = g.Wout; a.s = g.stride;
  } else {             // G = dY, D = layer input
    a.Gc = g.CoutT; a.Greal = g.Cout; a.Hgin = g.Hout; a.Wgin = g.Wout;
    a.Dc = g.Cin; a.Dreal = g.Cin_real;
    a.Hb = g.Hin; a.Wb = g.Win; a.s = 2;
  }
  a.log2Gc = ilog2(a.Gc);
  a.gc_mask = a.Gc - 1;
  if (a.ntaps == 1) { a.log2Gc = 30; a.gc_mask = 0x3fffffff; }   // 1x1: rows are channels, any count (BFMNet's 192 .. 1536-wide layers)
  p.cfg = (a.Dc % 128 == 0) ? 0 : (a.Dc % 64 == 0 ? 1 : 2);
  // (256-row 8-wave tiles for the register-transposing kernel measured no faster - one 240-register block per CU, EXPERIMENTS.md 3)
  a.lw = ilog2(a.Wb); a.lh = ilog2(a.Hb);
  const bool tr = is_bf16 && plain_operands && (wgrad_tr_knob() & 1) && (a.ntaps * a.Gc) % 256 == 0 && a.Dc % 128 == 0 &&
                  (((long long)a.N << (a.lw + a.lh)) % 32) == 0;
  if (tr) p.cfg = 5;
  // thin layers (image inputs: 16 taps x 8 padded channels = 128 rows): the 128 x 128 form of the same kernel, single-source operands
  const bool tr_thin = !tr && is_bf16 && plain_operands && (wgrad_tr_knob() & 2) && (a.ntaps * a.Gc) % 128 == 0 && a.Dc % 8 == 0 && a.Dc <= 128 &&
                       (((long long)a.N << (a.lw + a.lh)) % 32) == 0 && a.lw + a.lh >= 5;
  if (tr_thin) p.cfg = 6;
  int bm, bn;
  wgrad_tile(p.cfg, &bm, &bn);
  a.Mpad = round_up(a.ntaps * a.Gc, bm);
  a.Dpad = round_up(a.Dc, bn);
  const int kiter = (tr || tr_thin) ? 32 : kc_elems(is_bf16) * (is_bf16 ? 2 : 1);     // pixels per loop iteration of the kernel
  // padded-grid K walk: rows of 2^lw slots, at least one 16-byte pixel group and at most one iteration long
  a.fastw = (!tr && !tr_thin && (1 << a.lw) >= (is_bf16 ? 8 : 4) && (1 << a.lw) <= kiter) ? 1 : 0;
  const int P = (a.fastw || tr || tr_thin) ? (a.N << (a.lw + a.lh)) : a.N * a.Hb * a.Wb;
  const int nchunk = (P + kiter - 1) / kiter;
  const int tiles = (a.Mpad / bm) * (a.Dpad / bn);
  // K split: minimise (rounds of the ~512 resident blocks) x (iterations per block + fixed per-block cost),
  // plus a small penalty per slab for the reduce pass
  int s = 1;
  // vp_tune("wgrad_fixed_x10" / "wgrad_slab_x100" / "wgrad_slab_tile_x1000"): a slab costs a constant + a term per tile (its reduce moves the
  // whole slab through HBM twice: 2 x 128 KB per 256 x 128 tile)
  const double wg_fixed = wgrad_cost_knob(0) * 0.1, wg_slab = wgrad_cost_knob(1) * 0.01 + wgrad_cost_knob(2) * 0.001 * tiles;
  {
    const int smax = nchunk / 4 < 1 ? 1 : (nchunk / 4 > 512 ? 512 : nchunk / 4);
    double best = 1e30;
    for (int c = 1; c <= smax; ++c) {
      const int rounds = (tiles * c + plan_misc_knob(3) - 1) / plan_misc_knob(3);
      const double cost = rounds * ((double)nchunk / c + wg_fixed) + wg_slab * c;
      if (cost < best - 1e-9) { best = cost; s = c; }
    }
  }
  // wgrad_tr pins K splits to XCDs (the blocks of a split share their pixels in one L2): a multiple of 8 splits when there are 8 or more
  if ((tr || tr_thin) && s >= 8 && (s & 7)) {
    const int smax = nchunk / 4 < 1 ? 1 : (nchunk / 4 > 512 ? 512 : nchunk / 4);
    auto cost = [&](int c) { return ((tiles * c + plan_misc_knob(3) - 1) / plan_misc_knob(3)) * ((double)nchunk / c + wg_fixed) + wg_slab * c; };
    const int lo = s & ~7, hi = lo + 8;
    s = (hi <= smax && cost(hi) <= cost(lo)) ? hi : lo;
  }
  a.splitk = s;
  p.partial_bytes = (size_t)s * a.Mpad * a.Dpad * sizeof(float);
  return p;
}

}  // namespace vp
