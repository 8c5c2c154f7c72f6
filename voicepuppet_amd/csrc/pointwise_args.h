// Argument blocks of the HBM-bound kernels (weight packing, BN statistics, compositing, losses, Adam).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace vp {

// W[cls][row][k = tap*C + c] = src[kh*s_kh + kw*s_kw + row*s_row + c*s_ch]   (0 in the padding), stored CHUNK-MAJOR:
// element (cls, row, k) lives at ((cls*(Kpad/kc) + k/kc)*rows_pad + row)*kc + k%kc, so the 64-byte pieces a block needs
// for one K chunk are contiguous (a [row][K] image strides by K*2 bytes = a power of two: every row of a tile
// would sit on the same L2 channel)
struct PackDesc {
  size_t src_off;           // floats into the fp32 master arena
  size_t dst_off;           // elements into the packed arena
  int nclass, rows_real, rows_pad, ntaps, C, C_real, Kpad;
  int kc;                   // elements per 64-byte K chunk; packed layout is [class][K chunk][row][kc]
  int perm;                 // rows permuted inside every 64-row block (see IgemmArgs::rowperm): packed row of channel c = perm_row(c)
  int kswap;                // odd 16-byte k-pieces of every 64-byte chunk are stored with their two halves swapped (conv_patch.hip reads
                            // the pixel operand that way; a permutation of k common to both operands leaves the product unchanged)
  int s_kh, s_kw, s_row, s_ch;
  int8_t kh[4][16];
  int8_t kw[4][16];
};

// what PixReferNet.execute fetches from a finished forward pass (pixrefer.py:279-290, 414-438), formed on the device in one launch
struct FetchArgs {
  const float* raw3;     // Outputs in [-1, 1], [npix][3]
  const float* fg3;      // Outputs_FG, [npix][3]
  const float* o4;       // generator output after tanh, [npix][4] (channel 3 = alpha in [-1, 1])
  void* dst;             // [npix][3] float32 (uint8 for mode 1)
  size_t npix;
  int mode;              // 0 Outputs = (raw + 1) / 2; 1 the same as uint8 (clamp, * 255, truncate); 2 Alphas = (alpha + 1) / 2 tiled x 3;
                         // 3 Outputs_FG of build_inference_op: ((fg + (alpha + 1) / 2 - 1) + 1) / 2 (pixrefer.py:436)
};

struct BnArgs {
  const void* y;            // [G*Pg][C] raw conv output
  const void* dz;           // bwd: gradient w.r.t. the normalised tensor
  void* dy;                 // bwd apply: output (may alias dz)
  int C, G, Pg;             // channels, BN groups, pixels per group
  int nchunk;               // pixel chunks per group (partials)
  double* partial;          // [G][nchunk][2][C]
  const float* gamma;
  const float* beta;
  float* aff_a;             // [G][C]  z = a*y + b
  float* aff_b;
  float* mu;                // [G][C]
  float* rstd;
  float* c1;                // [G][C] mean(dz), mean(dz*zhat)
  float* c2;
  float* dgamma;            // [C]
  float* dbeta;
  int accumulate;           // dgamma/dbeta +=
  float eps;
  float* dbias_zero;        // bwd: [C] bias gradient of the conv in front of this BN, set to its analytic value 0 (may be null)
  int raw;                  // bwd finalize: the partial rows hold RAW moments (sum dz, sum dz * y) from a conv epilogue (IgemmArgs::bst_y)
};

struct PackInputsArgs {
  const float* inputs;      // [N,H,W,6] in [0,1]
  const float* fg_inputs;   // [N,H,W,fg_c]
  void* gin;                // [N,H,W,8]   generator input  (inputs*2-1, 0, 0)
  void* gfg;                // [N,H,W,8]   fg branch input  (fg[...,:3]*2-1, 0 x5)
  void* din;                // [3N,H,W,8]  discriminator batch: real1 | real2 | fake(cond only)
  void* vin;                // [2N,H,W,8]  VGG batch: real fg | (fake, written by composite)
  int N, HW, train;
  int fg_c;                 // channels of fg_inputs: 6 (training graphs) or 3 (infer_bfmvid.py:203 feeds [N,H,W,3]); only 0:3 feed the generator
};

// On-device form of PixReferDataGenerator.iterator (generator/generator.py:956-1019): per sample two decoded jpg triptychs
// (target | 3-D face | matte, S x 3S, uint8 BGR as cv2.imread returns them), each with its own random square crop
// (rx rows, ry columns, rsize), bilinear-resized back to S x S (cv2.resize INTER_LINEAR on float data) and packed.
struct FramePackArgs {
  const unsigned char* ex;    // [N][S][3S][3]  example frame
  const unsigned char* cur;   // [N][S][3S][3]  current frame
  const int* crops;           // [N][2][3]      (rx, ry, rsize) of the example and of the current frame
  float* inputs;              // [N][S][S][6]   3-D face of (example, current)
  float* fg_inputs;           // [N][S][S][6]   target * matte of (example, current)
  float* targets;             // [N][S][S][3]   target of the current frame
  float* masks;               // [N][S][S][3]   matte of the current frame
  int N, S;
};

struct CompositeArgs {
  const float* y4;          // [N,H,W,4] decoder_1 output (pre-tanh, bias included)
  const float* targets;     // [N,H,W,3] in [0,1]
  const float* masks;       // [N,H,W,3] in [0,1] (train only)
  float* o4;                // [N,H,W,4] tanh(y4)
  float* outputs;           // [N,H,W,3] in [-1,1]
  float* outputs_fg;        // [N,H,W,3]
  void* din;                // fake group image channels 3:6 (train)
  void* vin;                // fake half channels 0:3 (train)
  double* partial;          // [nblocks][2]  sum|tgt-out|, sum|mask-alpha|
  int N, HW, train;
  // backward
  const void* d_din;        // [N,H,W,8] grad of D layer_1 input (G loss), channels 3:6
  const void* d_vin;        // [N,H,W,8] grad of VGG input (fake half), channels 0:3
  void* dy4;                // [N,H,W,8] grad w.r.t. y4 (pre-tanh), channels 4:8 zero
  float l1_weight;
};

struct GanLossArgs {
  const float* logits;      // [3][M] D layer_5 output (bias included): real1 | real2 | fake
  void* dl_d;               // [3][M][8] seed of the D loss w.r.t. logits (channel 0)
  void* dl_g;               // [M][8]    seed of the G loss w.r.t. the fake logits
  float* predict;           // [2][M] predict_real, predict_fake
  float* losses;            // [0] Discrim_loss [1] Gen_loss_GAN
  int M;
  float gan_weight;
};

struct PerceptualArgs {
  const void* f3;           // [2N*hw][C] conv3_3 output (post-relu): real half | fake half
  void* df3;                // [N*hw][C] grad w.r.t. the pre-relu conv3_3 output of the fake half
  double* partial;          // [nblocks]
  size_t half;              // N*hw*C
  float l1_weight;
};

struct LossFinalArgs {
  const double* comp_partial; int n_comp;     // composite partials [n][2]
  const double* perc_partial; int n_perc;
  double n_out;             // N*H*W*3
  double n_feat;            // N*hw*C
  float* losses;            // [0] D [1] G_GAN -> writes [2] G_L1 [3] G_loss [4] perceptual
  float l1_weight, gan_weight;
};

// Single-output-channel stride-1 convolution as "GEMM over taps" (discriminator layer_5, pixrefer.py:128-131):
//   S[q][t] = sum_c x[q][c] * W[t][c]      one 1x1 GEMM over the INPUT pixels q (x is read once instead of once per tap)
//   y[p]    = bias + sum_t S[p + tap_t][t] gather of ks*ks = 16 partial sums
// and for the weight gradient  dW[t][c] = sum_q dyS[q][t] * x[q][c]  with  dyS[q][t] = dy[q - tap_t].
struct TapArgs {
  const float* S;           // [N,Hin,Win,16] f32 partial sums (fwd)
  const float* bias;        // [1]
  float* y;                 // [N,Hout,Wout] (fwd)
  const void* dy;           // [N,Hout,Wout,ld_dy] channel 0 (bwd)
  void* dyS;                // [N,Hin,Win,16] (bwd)
  int N, Hin, Win, Hout, Wout, ks, pad, ld_dy;
};

struct AdamArgs {
  float* p; const float* g; float* m; float* v;
  size_t n;
  float lr_t, beta1, beta2, eps;
};

}  // namespace vp
