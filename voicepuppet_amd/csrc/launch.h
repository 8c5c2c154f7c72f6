// Host-side launchers implemented in conv_kernels.hip / pointwise.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_args.h"
#include "pointwise_args.h"

namespace vp {

hipError_t launch_igemm(const IgemmArgs& a, int is_bf16, int cfg, hipStream_t st);
void igemm_tile(int cfg, int* bc, int* bp);
hipError_t launch_igemm_patch(const IgemmArgs& a, int is_bf16, int bc, int bp, hipStream_t st);                    // conv_patch.hip
bool patch3_eligible(const IgemmArgs& a, int is_bf16);                                                               // conv_patch3.hip
hipError_t launch_igemm_patch3(const IgemmArgs& a, int is_bf16, int bc, int bp, hipStream_t st);                   // conv_patch3.hip
bool patch4_eligible(const IgemmArgs& a, int is_bf16);                                                               // conv_patch3.hip (KW = 4)
hipError_t launch_igemm_patch4(const IgemmArgs& a, hipStream_t st);
bool conv_cin8_eligible(const IgemmArgs& a, int is_bf16);                                                           // conv_kernels.hip
bool conv_c64_eligible(const IgemmArgs& a, int is_bf16);                                                            // conv_c64.hip
hipError_t launch_conv_c64(const IgemmArgs& a, hipStream_t st);
bool conv_dc64_eligible(const IgemmArgs& a, int is_bf16);                                                           // conv_dc64.hip
hipError_t launch_conv_dc64(const IgemmArgs& a, hipStream_t st);
int conv_dc64_grid(const IgemmArgs& a);                        // its blocks = partial rows of IgemmArgs::colsum_part
bool conv_dc256_eligible(const IgemmArgs& a, int is_bf16);      // conv_dc64.hip: the forward form for 2 x 128 input channels (merged2_decoder_2)
int conv_dc256_grid(const IgemmArgs& a);                       // its blocks; batch-norm partial rows = 2 per block
hipError_t launch_conv_dc256(const IgemmArgs& a, hipStream_t st);
bool conv_s2c64_eligible(const IgemmArgs& a, int is_bf16);                                                          // conv_s2c64.hip
int conv_s2c64_grid(const IgemmArgs& a);                   // blocks of the launch = partial rows per batch-norm group
int conv_s2c64_tiles_per_image(const IgemmArgs& a);
hipError_t launch_conv_s2c64(const IgemmArgs& a, hipStream_t st);
bool conv_cout1_bwd_eligible(const Cout1Args& a);                                                                 // conv_cout1.hip
hipError_t launch_conv_cout1_bwd(const Cout1Args& a, hipStream_t st);
bool conv_cout1_wgrad_eligible(const Cout1Args& a);
hipError_t launch_conv_cout1_wgrad(const Cout1Args& a, hipStream_t st);
hipError_t launch_cout1_wgrad_prof(const Cout1Args& a, hipStream_t st);
hipError_t launch_cout1_bwd_prof(const Cout1Args& a, hipStream_t st);                                               // ... with the per-launch profile record (conv_kernels.hip)
hipError_t launch_igemm_patch2(const IgemmArgs& a, int is_bf16, int bc, int bp, hipStream_t st);                   // conv_patch2.hip
hipError_t launch_igemm_smallp(const IgemmArgs& a, int is_bf16, hipStream_t st);                                   // conv_smallp.hip (plain epilogue)
struct SmallPArgs;
hipError_t launch_smallp(const SmallPArgs& s, int is_bf16, hipStream_t st);                                         // conv_smallp.hip (fused batch-norm forms)
hipError_t launch_smallp_fused(const SmallPArgs& s, int is_bf16, hipStream_t st);                                   // ... with the per-launch profile record (conv_kernels.hip)
hipError_t launch_wgrad(const WgradArgs& a, int is_bf16, int cfg, hipStream_t st);
bool wgrad_mm_eligible(const WgradArgs& a, int cfg);                                                                                      // wgrad_mm.hip
hipError_t launch_wgrad_mm(const WgradArgs& a, int cfg, hipStream_t st);
hipError_t launch_wgrad_tr(const WgradArgs& a, hipStream_t st, const char** variant = nullptr, int* tile_cols = nullptr);                                          // wgrad_tr.hip
void wgrad_tile(int cfg, int* bm, int* bn);

void profile_enable(int on);          // 1: per kernel kind, 2: per layer tag
void profile_tag(const char* tag);
size_t profile_collect(char* out, size_t cap);

hipError_t launch_pack_weights(const PackDesc* d_descs, int ndesc, const float* master, void* packed, int is_bf16, hipStream_t st);
hipError_t launch_pack_weights_one(const PackDesc& d, const float* master, void* packed, int is_bf16, hipStream_t st);
int bn_nchunk(int Pg, int C, int G, int is_bf16);
hipError_t launch_bn_stats(const BnArgs& a, int is_bf16, hipStream_t st);
hipError_t launch_bn_finalize(const BnArgs& a, hipStream_t st);
bool bn_small(const BnArgs& a);     // few pixels per group: statistics + finalize + per-pixel pass in one launch
hipError_t launch_bn_small_fwd(const BnArgs& a, void* out_lrelu, void* out_relu, int is_bf16, hipStream_t st);
hipError_t launch_bn_small_bwd(const BnArgs& a, int is_bf16, hipStream_t st);
hipError_t launch_bn_bwd(const BnArgs& a, int is_bf16, hipStream_t st);
hipError_t launch_bn_bwd_tail(const BnArgs& a, int is_bf16, hipStream_t st);     // finalize + apply from partial rows a conv epilogue wrote (IgemmArgs::bst_*)
hipError_t launch_colsum(const BnArgs& a, int creal, float* out, int accumulate, int is_bf16, hipStream_t st);
hipError_t launch_colsum_tail(const BnArgs& a, int creal, float* out, int accumulate, hipStream_t st);   // the finalize alone: partial rows from a conv epilogue (IgemmArgs::colsum_part)
hipError_t launch_act_apply(const void* y, const float* sc, const float* sh, int C, int Pg, size_t npix,
                            void* out_lrelu, void* out_relu, int is_bf16, hipStream_t st);
hipError_t launch_tap_gather(const TapArgs& a, hipStream_t st);
hipError_t launch_tap_spread(const TapArgs& a, int is_bf16, hipStream_t st);
hipError_t launch_frame_pack(const FramePackArgs& a, hipStream_t st);
hipError_t launch_pack_inputs(const PackInputsArgs& a, int is_bf16, hipStream_t st);
int composite_nblocks(int N, int HW);
hipError_t launch_composite_fwd(const CompositeArgs& a, int is_bf16, hipStream_t st);
hipError_t launch_composite_bwd(const CompositeArgs& a, int is_bf16, hipStream_t st);
hipError_t launch_gan_loss(const GanLossArgs& a, int is_bf16, hipStream_t st);
int perceptual_nblocks(size_t half, int is_bf16);
hipError_t launch_perceptual(const PerceptualArgs& a, int is_bf16, hipStream_t st);
hipError_t launch_loss_final(const LossFinalArgs& a, hipStream_t st);
hipError_t launch_maxpool_fwd(const void* x, void* y, int B, int H, int W, int C, int is_bf16, hipStream_t st);
hipError_t launch_maxpool_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int is_bf16, hipStream_t st);
hipError_t launch_relu_bwd(const void* y, void* d, size_t n, int is_bf16, hipStream_t st);
hipError_t launch_adam(const AdamArgs& a, hipStream_t st);
hipError_t launch_fetch(const FetchArgs& a, hipStream_t st);
hipError_t launch_grad_pack_bf16(const float* src, void* dst, size_t n, hipStream_t st);
hipError_t launch_grad_unpack_bf16(const void* src, float* dst, size_t n, float scale, hipStream_t st);

// ---- shape / tiling helpers shared by the plan and the single-op entry points ----
struct ConvGeom {
  int kind;           // 0 conv (HWIO), 1 deconv k4 s2 (HWOI)
  int ks, stride, pad;
  int N, Hin, Win, Hout, Wout;
  int Cin, Cin_real;  // Cin: padded (multiple of 8, power of two); Cin_real: channels in the TF kernel
  int Cout;           // real
};

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace vp
