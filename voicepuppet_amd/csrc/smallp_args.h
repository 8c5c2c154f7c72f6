// Argument block of the few-pixel convolution kernel (conv_smallp.hip): the 1x1 .. 16x16 bottleneck of the generator
// (pixrefer.py:215-257 merged_encoder_* / merged_decoder_*), where a layer is a stream of 4-16 MB of weights against a
// few hundred pixels.  One launch does what the general path needs three to five launches for:
//   K split over blocks AND over the four waves of a block, slabs combined by the last-arriving block of a tile (in-launch,
//   fixed summation order), then - by the last-arriving tile of a channel group - the batch-norm statistics, the affine
//   and the materialised activations (forward), or the batch-norm backward of the tensor that receives the data gradient.
#pragma once
#include "conv_args.h"

namespace vp {

enum SmallPMode { SP_PLAIN = 0, SP_FWD_BN = 1, SP_BWD_BN = 2 };

struct SmallPArgs {
  IgemmArgs g;              // geometry, operands, plain epilogue (bias / activation / act'(ref) product / accumulate); g.splitk = K splits over blocks
  unsigned short tap_mask[4];   // per class: taps that fall inside the image for at least one pixel (the others are never read)
  float* slab;              // [K split][tile][PT][32] f32 partial tiles (K splits > 1)
  unsigned* cnt;            // [tiles] arrival counters of the K splits + [channel tiles] of the pixel tiles; zero before the launch, left zero
  double* part;             // SP_FWD_BN / SP_BWD_BN: [channel tile][pixel tile][2][32] statistics partials
  int mode;
  // SP_FWD_BN: batch-norm statistics of the layer's own output (pixrefer.py:99-101) + act(scale * y + shift) for the consumers
  const float* gamma;
  const float* beta;
  float* aff_a; float* aff_b; float* mu; float* rstd;
  void* out_lrelu; void* out_relu;
  float eps;
  // SP_BWD_BN: the tensor that receives dX is batch-normalised: y = its raw forward output, mu / rstd / gamma of its BN;
  // the launch leaves dL/dy (through the BN) in g.Y, and dgamma / dbeta / c1 / c2 (+ the analytically zero conv bias gradient)
  const void* bn_y;
  const float* bn_mu; const float* bn_rstd; const float* bn_gamma;
  float* c1; float* c2; float* dgamma; float* dbeta; float* dbias_zero;
  // f32 storage of a few-pixel tensor on the bf16 path (hi != 0; DESIGN.md "few-pixel tensors stay float32"): a batch-norm over N*1*1 ..
  // N*8*8 values per channel subtracts two projections from the tensor, and bf16 rounding of the tensor (2^-9) is then of the order of
  // what is left.  SP_FWD_BN: g.Y (the raw output) is float, statistics of the unrounded values, the consumers' activations stay T.
  // SP_BWD_BN: g.Y (the accumulated dz) and bn_y are float; the batch-norm backward's result dL/dy goes to dy_out as T (the MFMA operand
  // of the producer's weight / data gradients).  SP_PLAIN contributions to such a tensor use g.y_f32.
  int hi;
  void* dy_out;
};

}  // namespace vp
