// Launchers of audio_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace vp {
hipError_t launch_frame_window(const float* pcm, const float* window, float* frames, int B, int L, int F, int win, int hop, hipStream_t st);
// 512-sample frames, <= 80 mel bins: the whole log-mel front-end in one launch (FFT form)
hipError_t launch_logmel512(const float* pcm, const float* window, const float* w256, const float* w512, const float* mel, float* out, int B, int L, int F, int hop,
                            int nmel, hipStream_t st);
hipError_t launch_mag_mel_log(const float* spec, int ld, int nb, const float* mel, int nmel, float* out, int nframes, hipStream_t st);
hipError_t launch_conv_first(const float* x, const float* w, const float* bias, void* y, int out_bf16, int B, int H, int W, int Wo, int Cout, int pt, int pl, hipStream_t st);
hipError_t launch_dwconv7x3(const void* x, const float* w, const float* bias, void* y, int is_bf16, int B, int H, int W, int C, hipStream_t st, int rev = 0);
// bfm_dwproj.hip: depthwise 7x3 + ReLU6 + 1x1 projection (+ residual) of an inverted-residual block in one kernel (float32, mel widths <= 20);
// wdw22: the 21 folded taps followed by the folded bias row [22][Ce]; Wp: PackDesc-packed projection weights WITHOUT row permutation
inline int& bfm_dwproj_knob() { static int v = 1; return v; }
bool dwproj_eligible(int W, int Ce, int cout);
hipError_t launch_dwproj(const float* ex, const float* wdw22, const float* Wp, int rows_pad, const float* bias, float* y, int add, int B, int H, int W,
                         int Ce, int cout, hipStream_t st);
hipError_t launch_cvt_f32_bf16(const float* x, void* y, size_t n, hipStream_t st);
hipError_t launch_maxpool_same(const void* x, void* y, int in_bf16, int out_bf16, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo, hipStream_t st);
hipError_t launch_fold_bn(const float* w, const float* beta, const float* mean, const float* var, float eps, size_t n, int C, float* wf, float* bf, hipStream_t st);
hipError_t launch_gru_seq(const float* xg, const float* xc, const float* whg, const float* whc, const int* seq_len, float* out, int B, int T, hipStream_t st);
hipError_t launch_add_ears(float* out, const float* ears, int n, hipStream_t st);
hipError_t launch_mul_inplace(float* x, const float* m, size_t n, hipStream_t st);
}  // namespace vp
