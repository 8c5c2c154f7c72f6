// Argument blocks of the implicit-GEMM convolution kernels (host + device).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace vp {

struct TapTable {
  int8_t dh[16];
  int8_t dw[16];
};

// A pixel-major (NHWC) tensor, optionally the virtual concat of two tensors along C, read through
// the deferred batch-norm affine of its producer (z = a*y + b per channel and BN group) followed by
// the consumer's activation.  This is how "BN -> act -> conv" costs no extra pass over HBM.
struct PixSrc {
  const void* ptr[2];
  int C[2];                 // channels per source (C[1] = 0: single source)
  const float* aff_a[2];    // [groups][C] or null (identity)
  const float* aff_b[2];
  int act;                  // vp::Act applied after the affine
  int group_n;              // samples per BN group
};

// Y[pixel, co] = epilogue( sum_{tap, ci} X~[pixel (+) tap, ci] * Wp[co][tap*Cin + ci] )
// Covers conv fwd, conv bwd-data, transposed-conv fwd (4 output-parity classes) and its bwd-data.
struct IgemmArgs {
  PixSrc x;
  int N, Hin, Win;
  int Cin, log2Cin;         // total channels of x (power of two unless ntaps == 1)
  int cin_mask;             // Cin-1 (ntaps > 1) or all-ones (1x1: k is the channel index itself)
  int cin_real;             // channels that carry data (accounting only)
  int Hg, Wg;               // GEMM pixel grid per class: pixel = (n, q, r)
  int sh, sw;               // ih = q*sh + dh[tap], iw = r*sw + dw[tap]
  int ntaps, nclass;
  TapTable taps[4];
  const void* Wp;           // [nclass][CoutPad][Kpad], element type T
  int Kpad, CoutPad;        // CoutPad: rows rounded up to the channel tile (grid / slab stride)
  int wp_rows;              // rows per class in Wp (>= CoutPad, zero padded)
  void* Y;
  int y_f32;                // store Y as float even when T is bf16
  int Cout, ldY;            // real output channels (store mask), channel stride of one Y pixel
  int Hof, Wof, os;         // Y pixel = (n, q*os + o0h[cls], r*os + o0w[cls]) in an Hof x Wof image
  int o0h[4], o0w[4];
  const float* bias;        // [Cout] or null
  int out_act;
  // backward-data epilogue: Y = [Y +] acc * act'(ref_a*ref + ref_b); ref has Y's geometry
  const void* ref;
  const float* ref_a;
  const float* ref_b;
  int ref_act, ref_group_n, accumulate;
  // Two-output form (split_c > 0; backward-data of a layer that reads the virtual concat of two tensors of split_c channels each -
  // the decoders' skip connections): ONE GEMM over the rows of both sources; output channels [0, split_c) go to Y / ref / accumulate,
  // channels [split_c, 2 * split_c) to Y2 / ref2 / accumulate2 (same pixel stride ldY = split_c).  Both data gradients share the dY
  // operand and, at 4-8 frames per GPU, one launch on the step's critical chain instead of two
  int split_c;
  void* Y2;
  const void* ref2;
  int accumulate2;
  int y2_f32;               // the second output is float32 (a float32 few-pixel tensor's gradient accumulator) although T is bf16
  int splitk;
  float* partial;           // [nclass][splitk][P][CoutPad] when splitk > 1
  int vec_epi;              // staged (LDS) epilogue with 16-byte row stores (set by the launcher)
  int fastk;                // buffer-descriptor loader with scalar K stepping (set by the launcher)
  int rowperm;              // packed weight rows are permuted inside every 64-row block (PackDesc::perm): MFMA tile t, row 4q+e of a
                            // block holds channel 32*(t>>1) + 8q + 4*(t&1) + e, so a lane ends with 2 x 8 consecutive channels per pixel
  int xcd_remap;            // conv_patch3.hip: pixel tiles dealt to the XCDs in contiguous runs (set by the launcher)
  double* bn_part;          // staged epilogue also writes batch-norm partials [group][bn_nchunk][2][Cout] (null: no)
  int bn_tpg, bn_nchunk;    // pixel tiles per BN group (per class), partial chunks per group = nclass * bn_tpg
  // Backward sums of a batch-normalised tensor in the epilogue of the launch that completes its gradient (staged epilogue, STATS == 2):
  // bst_y = that tensor's raw forward output (Y's geometry and channel count); the partial rows [group][bn_nchunk][2][C] = RAW moments
  // (sum dz, sum dz * y) go to bn_part (bn_tpg / bn_nchunk as for the forward statistics; bn_bwd_finalize_kernel with BnArgs::raw turns
  // them into sum dz * zhat).  Two-output launches: bst_y2 / bn_part2 belong to the second output; either side may be null (that output
  // is not complete yet / has no batch-norm)
  const void* bst_y;
  const void* bst_y2;
  double* bn_part2;
  double* colsum_part;      // conv_dc64_kernel only: also write the column sums of the stored output, one row [2][64] per block (null: no)
  const void* zeros;        // >= 16 bytes of zeros (padding source of the LDS-DMA loader); null: register loader
  // patch kernel (conv_patch.hip; plan-time decision, the packed weights carry PackDesc::kswap): stride-1 taps on a regular grid,
  // tap t = r * p_kw + c  ->  (dh, dw) = (p_dhf + r * p_dhs, p_dwf + c * p_dws)
  int patch, p_kw, p_dhf, p_dhs, p_dwf, p_dws;
  // first layers without batch-norm (encoder_1, encoder_fg_1, discriminator layer_1; conv_cin8_kernel only - conv_cin8_eligible): the
  // epilogue also writes the activations the consumers read - what act_apply would materialise from Y in a pass of its own (Y itself
  // is stored only when the caller passes Y != null: the rounding-aware oracle tests teacher-force on it, a step does not read it)
  void* xa_lrelu;
  void* xa_relu;
  void* pool_out;           // patch kernel, 16 x 16-pixel tiles: also write the 2x2 max-pooled output [N][Hg/2][Wg/2][ldY] (null: no)
  int pool_only;            // with pool_out: write ONLY the pooled output (nobody reads the full-resolution tensor: the real half of the VGG trunk)
  // few-pixel kernel (conv_smallp.hip; patch == 3, plan-time decision: packed rows unpermuted): 32 channels x sp_npt * 16 pixels per tile,
  // splitk = K splits over blocks, partial = their slabs [split][tile][pixels][32]
  unsigned* sp_cnt;         // [tiles + channel tiles] arrival counters: zero before the launch, left zero by it
  unsigned short sp_mask[4];   // per class: taps inside the image for at least one pixel
  int sp_npt, sp_lcpt;      // MFMA pixel tiles per block tile (1, 2, 4); log2(64-byte K chunks per tap)
};

// dW[tap][g][d] = sum_{pixels} G~[pixel (+) tap, g] * D~[pixel, d]
//   conv  : G = layer input (ci), D = dY (co)       -> HWIO kernel gradient
//   deconv: G = dY (co),          D = layer input    -> HWOI kernel gradient
struct WgradArgs {
  PixSrc g;
  int Gc, log2Gc;           // padded channels of the gathered operand (power of two; any count with one tap: log2Gc = 30)
  int gc_mask;              // Gc - 1 (or all-ones with one tap): row m of the gradient = (tap = m >> log2Gc, channel = m & gc_mask)
  int Hgin, Wgin;           // spatial size of the gathered tensor
  PixSrc d;
  int Dc;                   // channels of the dense operand
  int N, Hb, Wb;            // base grid: dense pixel (n,q,r); gathered pixel (n, q*s+dh, r*s+dw)
  int s, ntaps;
  TapTable taps;
  int Mpad, Dpad;           // slab dims
  int splitk;
  float* partial;           // [splitk][Mpad][Dpad]
  float* dW;                // [ntaps][Greal][Dreal]
  int Greal, Dreal, accumulate;
  const void* zeros;        // >= 16 zero bytes: enables the branch-free loader for prologue-free operands
  int lw, lh;               // log2 of Wb, Hb rounded up to powers of two
  int fastw;                // K walks the padded grid [N][2^lh][2^lw] (division-free loader)
  int xcd_remap;            // wgrad_tr.hip: K splits pinned to XCDs (set by the launcher)
  int fast_tr;              // wgrad_tr.hip: buffer-descriptor DMAs with scalar per-chunk offsets (set by the launcher: tensors below 2 GiB)
};

// conv_cout1.hip: backward-data of a 4x4 stride-1 convolution with ONE output channel over 512 input channels (the discriminator's
// layer_5), with the act'(reference) product and - optionally - the two raw moments of the producing layer's batch-norm backward
struct Cout1Args {
  const void* dy;       // [N, Ho, Wo, ld_dy] bf16: channel 0 is the gradient of the one output channel
  int ld_dy;
  const float* w;       // [ks * ks][C] float32 master weights (HWIO with O = 1); rounded to bf16 in the kernel as pack_weights rounds them
  const void* ref;      // [N, H, W, C] bf16: the materialised activated input (lrelu'(ref) product)
  int ref_act;
  const void* y;        // [N, H, W, C] bf16 raw (pre-norm) output of the producer: only read when `part` is set
  double* part;         // null, or [groups * rows][2][C]: one partial row (sum dx, sum dx * y of the gradient as stored) per block
  void* dx;             // [N, H, W, C] bf16 out
  int N, H, W, C, Ho, Wo, ks, pad;
  int groups, rows;     // batch-norm groups of the N images; blocks (= partial rows) per group
  float* slabs;         // weight-gradient form (cout1_wgrad_kernel): [rows][ks * ks][C] float32 scratch, one slab per block
  float* dW;            // ... and the gradient [ks * ks][C] float32 (HWIO with O = 1), overwritten
  int pix_per_group, tiles_per_group;  // (filled by the launcher) pixels and 16-pixel tiles per group (the last tile of a group may be ragged)
};

}  // namespace vp
