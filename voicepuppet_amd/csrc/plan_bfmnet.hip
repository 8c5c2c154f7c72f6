// Audio front-end executors: log-mel (DataGenerator.extract_mfcc, generator/generator.py:60-80) and
// BFMNet inference (voicepuppet/bfmnet/bfmnet.py:189-213,325-333 + tinynet.py:159-212), f32.
//   * STFT = framing kernel + ONE f32-MFMA GEMM against a [512 x 514] cos/-sin matrix (the DFT of a 512
//     window is 0.26 MMAC per frame; a GEMM keeps it on the matrix cores and bit-stable), then a fused
//     |.| -> mel(257x80) -> log kernel.
//   * MfccNet: inference batch-norm (contrib, no gamma, eps 1e-3) is folded into the conv weights on the
//     device whenever the parameters change; every 1x1 conv / dense layer is the implicit-GEMM kernel with
//     bias + activation (+ residual accumulate) in its epilogue; depthwise 7x3, the 9x5 stem and the SAME
//     max-pools are VALU kernels; the GRU is one persistent block per sequence.
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#include "audio_args.h"
#include "conv_ops.h"
#include "errors.h"
#include "launch.h"
#include "vp_common.h"

using namespace vp;

namespace {

struct Bump {
  char* base; size_t off;
  void* alloc(size_t bytes) { off = (off + 255) & ~(size_t)255; void* p = base ? base + off : nullptr; off += bytes; return p; }
};

void same_pad(int size, int k, int s, int* pb, int* out) {
  const int o = (size + s - 1) / s;
  int total = (o - 1) * s + k - size;
  if (total < 0) total = 0;
  *pb = total / 2; *out = o;
}

// a 1x1 conv / dense layer as an igemm over `pixels` rows
struct Gemm {
  int cin = 0, cout = 0;
  size_t w_src = 0;      // float offset of the [cin, cout] matrix in its source arena
  int src_ld = 0;        // row stride (cout of the full TF kernel; GRU kernels are read partially)
  int which = 0;         // 0: folded arena, 1: raw parameter arena
  size_t bias = 0;       // float offset of the bias (folded arena or raw arena, same `which`)
  size_t pk = 0;         // BYTE offset in the packed arena
  int bf = 0;            // bf16 operands / output (the MfccNet trunk in bf16-storage mode), f32 accumulation
  IgemmPlan plan;
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// log-mel
// ------------------------------------------------------------------------------------------------
struct vp_logmel {
  vp_logmel_desc d;
  int frames, nb, ncol;
  float *window, *dft, *mel, *frames_buf, *spec;
  float *w256, *w512;      // FFT twiddles of the one-launch form (audio_kernels.hip logmel512_kernel)
  bool fused;              // 512-sample frames, <= 80 mel bins: one launch; other shapes: framing + DFT matrix product + mel (three)
  char* packed;
  char* scratch;
  void* zeros;
  IgemmPlan plan;
};

static size_t logmel_carve(vp_logmel* h, char* base) {
  Bump ar{base, 0};
  const vp_logmel_desc& d = h->d;
  h->frames = 1 + (d.samples - d.win_length) / d.hop_step;
  h->nb = d.fft_length / 2 + 1;
  h->ncol = round_up(2 * h->nb, 8);
  const size_t P = (size_t)d.batch * h->frames;
  h->window = (float*)ar.alloc(d.win_length * sizeof(float));
  h->dft = (float*)ar.alloc((size_t)d.win_length * h->ncol * sizeof(float));
  h->mel = (float*)ar.alloc((size_t)h->nb * d.num_mel_bins * sizeof(float));
  h->w256 = (float*)ar.alloc(256 * 2 * sizeof(float));
  h->w512 = (float*)ar.alloc(257 * 2 * sizeof(float));
  h->fused = d.win_length == 512 && d.num_mel_bins <= 80;
  h->frames_buf = (float*)ar.alloc(P * d.win_length * sizeof(float));
  h->spec = (float*)ar.alloc(P * h->ncol * sizeof(float));
  ConvGeomX g = make_geom(0, 1, 1, 0, 1, (int)P, 1, d.win_length, d.win_length, h->ncol);
  h->plan = plan_fwd(g, 0, 0);
  h->packed = (char*)ar.alloc(h->plan.pack_elems * sizeof(float));
  h->scratch = (char*)ar.alloc(h->plan.partial_bytes + 256);
  h->zeros = ar.alloc(256);
  return ar.off + 256;
}

static bool logmel_ok(const vp_logmel_desc* d) {
  return d && d->batch >= 1 && d->win_length == d->fft_length && d->win_length % 16 == 0 && d->win_length <= 4096 &&
         d->hop_step >= 1 && d->samples >= d->win_length && d->num_mel_bins >= 1 && d->num_mel_bins <= 128 && d->sample_rate > 0;
}

extern "C" {

size_t vp_logmel_workspace_bytes(const vp_logmel_desc* d) {
  if (!logmel_ok(d)) return 0;
  vp_logmel h{};
  h.d = *d;
  return logmel_carve(&h, nullptr);
}

int vp_logmel_frames(const vp_logmel_desc* d) { return logmel_ok(d) ? 1 + (d->samples - d->win_length) / d->hop_step : 0; }

int vp_logmel_create(const vp_logmel_desc* d, void* workspace, size_t bytes, void* stream, vp_logmel_t** out) {
  if (!logmel_ok(d) || !workspace || !out) { set_err("vp_logmel_create: bad argument"); return VP_ERR_ARG; }
  vp_logmel* h = new vp_logmel{};
  h->d = *d;
  if (bytes < logmel_carve(h, nullptr)) { delete h; set_err("vp_logmel_create: workspace too small"); return VP_ERR_WORKSPACE; }
  logmel_carve(h, (char*)workspace);
  hipStream_t st = (hipStream_t)stream;
  const int n = d->win_length, nb = h->nb;
  // constants in double, stored as f32: periodic Hann, DFT matrix [n][cos 0..nb-1 | -sin 0..nb-1 | 0], HTK mel matrix
  std::vector<float> win(n), dft((size_t)n * h->ncol, 0.f), mel((size_t)nb * d->num_mel_bins, 0.f);
  const double PI = 3.14159265358979323846;
  for (int i = 0; i < n; ++i) win[i] = (float)(0.5 - 0.5 * cos(2.0 * PI * i / n));
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < nb; ++k) {
      const double ang = 2.0 * PI * (double)(((long long)i * k) % n) / n;
      dft[(size_t)i * h->ncol + k] = (float)cos(ang);
      dft[(size_t)i * h->ncol + nb + k] = (float)(-sin(ang));
    }
  // tf.signal.linear_to_mel_weight_matrix (HTK mel, DC row zero, triangles in the mel domain)
  auto hz2mel = [](double f) { return 1127.0 * log1p(f / 700.0); };
  const double lo = hz2mel(d->lower_hz), hi = hz2mel(d->upper_hz);
  for (int k = 1; k < nb; ++k) {
    const double m = hz2mel((double)k * (d->sample_rate / 2.0) / (nb - 1));
    for (int j = 0; j < d->num_mel_bins; ++j) {
      const double e0 = lo + (hi - lo) * j / (d->num_mel_bins + 1), e1 = lo + (hi - lo) * (j + 1) / (d->num_mel_bins + 1),
                   e2 = lo + (hi - lo) * (j + 2) / (d->num_mel_bins + 1);
      const double v = fmin((m - e0) / (e1 - e0), (e2 - m) / (e2 - e1));
      mel[(size_t)k * d->num_mel_bins + j] = (float)(v > 0 ? v : 0);
    }
  }
  std::vector<float> t256(512), t512(514);
  for (int i = 0; i < 256; ++i) { t256[2 * i] = (float)cos(2.0 * PI * i / 256); t256[2 * i + 1] = (float)(-sin(2.0 * PI * i / 256)); }
  for (int i = 0; i < 257; ++i) { t512[2 * i] = (float)cos(2.0 * PI * i / 512); t512[2 * i + 1] = (float)(-sin(2.0 * PI * i / 512)); }
  VP_HIP_CHECK(hipMemcpyAsync(h->w256, t256.data(), t256.size() * 4, hipMemcpyHostToDevice, st));
  VP_HIP_CHECK(hipMemcpyAsync(h->w512, t512.data(), t512.size() * 4, hipMemcpyHostToDevice, st));
  VP_HIP_CHECK(hipMemcpyAsync(h->window, win.data(), win.size() * 4, hipMemcpyHostToDevice, st));
  VP_HIP_CHECK(hipMemcpyAsync(h->dft, dft.data(), dft.size() * 4, hipMemcpyHostToDevice, st));
  VP_HIP_CHECK(hipMemcpyAsync(h->mel, mel.data(), mel.size() * 4, hipMemcpyHostToDevice, st));
  VP_HIP_CHECK(hipMemsetAsync(h->zeros, 0, 256, st));
  VP_HIP_CHECK(hipStreamSynchronize(st));   // host staging vectors die here
  VP_HIP_CHECK(launch_pack_weights_one(h->plan.pack, h->dft, h->packed, 0, st));
  *out = h;
  return VP_OK;
}

void vp_logmel_destroy(vp_logmel_t* h) { delete h; }

int vp_logmel_forward(vp_logmel_t* h, const float* pcm, float* out, void* stream) {
  if (!h || !pcm || !out) { set_err("vp_logmel_forward: null argument"); return VP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const vp_logmel_desc& d = h->d;
  const int P = d.batch * h->frames;
  if (h->fused) {
    VP_HIP_CHECK(launch_logmel512(pcm, h->window, h->w256, h->w512, h->mel, out, d.batch, d.samples, h->frames, d.hop_step, d.num_mel_bins, st));
    return VP_OK;
  }
  VP_HIP_CHECK(launch_frame_window(pcm, h->window, h->frames_buf, d.batch, d.samples, h->frames, d.win_length, d.hop_step, st));
  IgemmArgs a = h->plan.a;
  set_single_src(a.x, h->frames_buf, d.win_length, nullptr, nullptr, ACT_NONE, 0);
  a.Wp = h->packed; a.Y = h->spec; a.ldY = h->ncol; a.partial = (float*)h->scratch; a.zeros = h->zeros;
  VP_HIP_CHECK(launch_igemm(a, 0, h->plan.cfg, st));
  VP_HIP_CHECK(launch_mag_mel_log(h->spec, h->ncol, h->nb, h->mel, d.num_mel_bins, out, P, st));
  return VP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// BFMNet inference
// ------------------------------------------------------------------------------------------------
namespace {

struct PInfo { std::string name; size_t off; int ndim; int64_t shape[4]; };

struct ConvBN {            // conv (or depthwise) + contrib batch_norm
  size_t w, beta, mean, var;   // offsets in the parameter arena
  size_t wn;                   // number of weights
  int C;                       // output channels
  size_t wf, bf;               // offsets in the folded arena
};

struct Block {
  int cin, cexp, cout;
  bool pool, shortcut;
  bool fusable;          // depthwise + projection in one kernel (bfm_dwproj.hip): float32 trunk, mel width <= 20
  ConvBN expand, dw, project, sc;
  Gemm g_expand, g_project, g_sc;
};

struct BfmModel {
  std::vector<PInfo> manifest;
  size_t nparams = 0, nfolded = 0;
  ConvBN stem, last;
  Gemm g_last;
  std::vector<Block> blocks;
  std::vector<ConvBN*> folds;
  size_t enc_w, enc_b, rnn_w, rnn_b, gk, gb, ck, cb, d0w, d0b, d1w, d1b, d2w, d2b;
  Gemm g_enc, g_rnn, g_xg, g_xc, g_d0, g_d1, g_d2;
};

size_t add_p(BfmModel& m, const std::string& name, std::initializer_list<int64_t> shape) {
  PInfo p;
  p.name = name; p.off = m.nparams; p.ndim = (int)shape.size();
  size_t cnt = 1; int i = 0;
  for (auto s : shape) { p.shape[i++] = s; cnt *= (size_t)s; }
  for (; i < 4; ++i) p.shape[i] = 1;
  m.nparams += cnt;
  m.manifest.push_back(p);
  return p.off;
}

ConvBN add_conv_bn(BfmModel& m, const std::string& scope, const char* wname, std::initializer_list<int64_t> wshape, int C) {
  ConvBN c{};
  c.w = add_p(m, scope + "/" + wname, wshape);
  c.wn = 1;
  for (auto s : wshape) c.wn *= (size_t)s;
  const std::string bn = scope.substr(0, scope.rfind('/')) + "/BatchNorm/";
  c.beta = add_p(m, bn + "beta", {C});
  c.mean = add_p(m, bn + "moving_mean", {C});
  c.var = add_p(m, bn + "moving_variance", {C});
  c.C = C;
  c.wf = m.nfolded; m.nfolded += c.wn;
  c.bf = m.nfolded; m.nfolded += (size_t)C;
  return c;
}

void build_model(BfmModel& m) {
  const std::string P = "mfcc_encoder/MfccNet/";
  m.stem = add_conv_bn(m, P + "block0_0/conv2d/conv2d", "kernel", {9, 5, 1, 32}, 32);
  struct Spec { const char* scope; int cout, exp; bool pool; };
  static const Spec specs[] = {{"block1_0", 64, 1, false}, {"block2_0", 64, 6, true}, {"block2_1", 64, 6, false},
                               {"block3_0", 128, 6, true}, {"block3_1", 128, 6, false}, {"block3_2", 128, 6, false},
                               {"block4_0", 192, 6, true}, {"block4_1", 192, 6, false}, {"block4_2", 192, 6, false}, {"block4_3", 192, 6, false},
                               {"block5_0", 256, 6, false}, {"block5_1", 256, 6, false}, {"block5_2", 256, 6, false},
                               {"block6_0", 256, 6, true}, {"block6_1", 256, 6, false}, {"block6_2", 256, 6, false},
                               {"block7_0", 256, 6, false}};
  int cin = 32;
  m.blocks.reserve(32);
  for (const Spec& s : specs) {
    Block b{};
    b.cin = cin; b.cexp = cin * s.exp; b.cout = s.cout; b.pool = s.pool; b.shortcut = s.cout != cin;
    const std::string B = P + s.scope;
    b.expand = add_conv_bn(m, B + "/expansion_1x1_conv2d/conv2d", "kernel", {1, 1, cin, b.cexp}, b.cexp);
    // depthwise: variable scope .../depthwise_conv2d/SeparableConv2d/depthwise_weights, BN at .../depthwise_conv2d/BatchNorm
    b.dw = add_conv_bn(m, B + "/depthwise_conv2d/SeparableConv2d", "depthwise_weights", {7, 3, b.cexp, 1}, b.cexp);
    b.project = add_conv_bn(m, B + "/projection_1x1_conv2d/conv2d", "kernel", {1, 1, b.cexp, s.cout}, s.cout);
    if (b.shortcut) b.sc = add_conv_bn(m, B + "/1x1_conv2d/conv2d", "kernel", {1, 1, cin, s.cout}, s.cout);
    m.blocks.push_back(b);
    cin = s.cout;
  }
  m.last = add_conv_bn(m, P + "block8_0/conv2d/conv2d", "kernel", {1, 1, cin, 256}, 256);
  m.enc_w = add_p(m, "mfcc_encoder/dense/kernel", {256, 256}); m.enc_b = add_p(m, "mfcc_encoder/dense/bias", {256});
  m.rnn_w = add_p(m, "rnn_module/dense/kernel", {256, 256}); m.rnn_b = add_p(m, "rnn_module/dense/bias", {256});
  const std::string G = "rnn_module/rnn/multi_rnn_cell/cell_0/gru_cell/";
  m.gk = add_p(m, G + "gates/kernel", {512, 512}); m.gb = add_p(m, G + "gates/bias", {512});
  m.ck = add_p(m, G + "candidate/kernel", {512, 256}); m.cb = add_p(m, G + "candidate/bias", {256});
  m.d0w = add_p(m, "bfm_coeff_decoder/dense/kernel", {256, 128}); m.d0b = add_p(m, "bfm_coeff_decoder/dense/bias", {128});
  m.d1w = add_p(m, "bfm_coeff_decoder/dense_1/kernel", {128, 64}); m.d1b = add_p(m, "bfm_coeff_decoder/dense_1/bias", {64});
  m.d2w = add_p(m, "bfm_coeff_decoder/dense_2/kernel", {64, 64}); m.d2b = add_p(m, "bfm_coeff_decoder/dense_2/bias", {64});
}

}  // namespace

struct vp_bfmnet {
  vp_bfmnet_desc d;
  BfmModel m;
  const float* params;
  float* folded;
  char* packed;
  size_t packed_elems;
  void* nb16;            // bf16 copy of the current narrow trunk tensor (bf16-trunk mode: operand of the expansion / shortcut convs)
  float *mf, *n0, *n1, *ex, *dwb, *pooled, *enc, *c1, *xg, *xc, *rnn, *dd0, *dd1;
  char* scratch;
  size_t scratch_bytes;
  void* zeros;
  bool dirty;
  int T5, Wm[6];
  const float *drop0, *drop1;   // opt-in: the reference's unconditional decoder dropout (bfmnet.py:114,116) as explicit masks
};

namespace {

void plan_gemm(vp_bfmnet* h, Gemm& g, int pixels, int cin, int cout, int which, size_t w_src, int src_ld, size_t bias, int bf = 0) {
  g.cin = cin; g.cout = cout; g.which = which; g.w_src = w_src; g.src_ld = src_ld; g.bias = bias; g.bf = bf;
  ConvGeomX geo = make_geom(0, 1, 1, 0, 1, pixels, 1, cin, cin, cout);
  g.plan = plan_fwd(geo, w_src, bf);
  g.plan.pack.s_ch = src_ld;      // row stride of the [cin, cout] source matrix (HWIO with H = W = 1)
  g.pk = h->packed_elems;         // (bytes: the arena mixes bf16 and f32 blocks; every block is packed relative to its own base)
  g.plan.pack.dst_off = 0;
  h->packed_elems += (g.plan.pack_elems * (bf ? 2 : 4) + 255) & ~(size_t)255;
  if (g.plan.partial_bytes > h->scratch_bytes) h->scratch_bytes = g.plan.partial_bytes;
}

size_t bfm_carve(vp_bfmnet* h, char* base) {
  Bump ar{base, 0};
  const int B = h->d.batch, T = h->d.frames;
  h->T5 = 5 * T;
  BfmModel& m = h->m;
  h->packed_elems = 0; h->scratch_bytes = 0;
  // mel widths: 80 -> 40 (stem stride 2) -> 20 -> 10 -> 5 -> 3 (SAME pools after block2_0, 3_0, 4_0, 6_0)
  int W = (h->d.num_mel_bins + 1) / 2;
  size_t max_net = 0, max_exp = 0;
  const int tb = h->d.trunk_dtype == VP_BF16 ? 1 : 0;      // MfccNet activations / 1x1-conv operands in bf16 (f32 accumulation)
  const size_t tes = tb ? 2 : 4;
  for (Block& b : m.blocks) {
    const int P = B * h->T5 * W;
    plan_gemm(h, b.g_expand, P, b.cin, b.cexp, 0, b.expand.wf, b.cexp, b.expand.bf, tb);
    plan_gemm(h, b.g_project, P, b.cexp, b.cout, 0, b.project.wf, b.cout, b.project.bf, tb);
    // (the fused kernel addresses the expanded tensor with 32-bit lane offsets - launch_dwproj refuses 0xF0000000 bytes and more: very long
    // clips fall back to the depthwise kernel + the GEMM here, at plan time, instead of failing at run time: ADVICE r5)
    b.fusable = !tb && dwproj_eligible(W, b.cexp, b.cout) && (size_t)P * b.cexp * sizeof(float) < 0xF0000000ull;
    if (b.fusable) { b.g_project.plan.pack.perm = 0; b.g_project.plan.a.rowperm = 0; }     // (the fused kernel reads plain packed rows; the GEMM kernels take either)
    if (b.shortcut) plan_gemm(h, b.g_sc, P, b.cin, b.cout, 0, b.sc.wf, b.cout, b.sc.bf, tb);
    if ((size_t)P * b.cexp > max_exp) max_exp = (size_t)P * b.cexp;
    if ((size_t)P * (b.cin > b.cout ? b.cin : b.cout) > max_net) max_net = (size_t)P * (b.cin > b.cout ? b.cin : b.cout);
    if (b.pool) W = (W + 1) / 2;
  }
  plan_gemm(h, m.g_last, B * h->T5 * W, 256, 256, 0, m.last.wf, 256, m.last.bf, tb);
  const int BT = B * T;
  plan_gemm(h, m.g_enc, BT, 256, 256, 1, m.enc_w, 256, m.enc_b);
  plan_gemm(h, m.g_rnn, BT, 256, 256, 1, m.rnn_w, 256, m.rnn_b);
  plan_gemm(h, m.g_xg, BT, 256, 512, 1, m.gk, 512, m.gb);       // rows 0..255 of the [512,512] gate kernel (the x part)
  plan_gemm(h, m.g_xc, BT, 256, 256, 1, m.ck, 256, m.cb);
  plan_gemm(h, m.g_d0, BT, 256, 128, 1, m.d0w, 128, m.d0b);
  plan_gemm(h, m.g_d1, BT, 128, 64, 1, m.d1w, 64, m.d1b);
  plan_gemm(h, m.g_d2, BT, 64, 64, 1, m.d2w, 64, m.d2b);
  h->folded = (float*)ar.alloc(m.nfolded * sizeof(float));
  h->packed = (char*)ar.alloc(h->packed_elems);
  h->n0 = (float*)ar.alloc(max_net * sizeof(float));      // the residual stream stays f32 in both modes
  h->n1 = (float*)ar.alloc(max_net * sizeof(float));
  h->nb16 = ar.alloc(max_net * 2);
  h->ex = (float*)ar.alloc(max_exp * tes);
  h->dwb = (float*)ar.alloc(max_exp * tes);
  h->pooled = (float*)ar.alloc((size_t)BT * 256 * 4);
  h->enc = (float*)ar.alloc((size_t)BT * 256 * 4);
  h->c1 = (float*)ar.alloc((size_t)BT * 256 * 4);
  h->xg = (float*)ar.alloc((size_t)BT * 512 * 4);
  h->xc = (float*)ar.alloc((size_t)BT * 256 * 4);
  h->rnn = (float*)ar.alloc((size_t)BT * 256 * 4);
  h->dd0 = (float*)ar.alloc((size_t)BT * 128 * 4);
  h->dd1 = (float*)ar.alloc((size_t)BT * 64 * 4);
  h->scratch = (char*)ar.alloc(h->scratch_bytes + 256);
  h->zeros = ar.alloc(256);
  return ar.off + 256;
}

bool bfm_ok(const vp_bfmnet_desc* d) {
  return d && d->batch >= 1 && d->frames >= 1 && d->num_mel_bins == 80 && (d->trunk_dtype == VP_F32 || d->trunk_dtype == VP_BF16);
}

int run_gemm(vp_bfmnet* h, Gemm& g, const void* x, void* y, int act, int accumulate, hipStream_t st, int y_f32 = 0) {
  IgemmArgs a = g.plan.a;
  a.y_f32 = (g.bf && y_f32) ? 1 : 0;           // bf16 operands, f32 result (residual stream / head inputs)
  set_single_src(a.x, x, g.cin, nullptr, nullptr, ACT_NONE, 0);
  a.Wp = h->packed + g.pk;
  a.Y = y; a.ldY = g.cout;
  a.bias = (g.which == 0 ? h->folded : h->params) + g.bias;
  a.out_act = act; a.accumulate = accumulate;
  a.partial = (float*)h->scratch; a.zeros = h->zeros;
  VP_HIP_CHECK(launch_igemm(a, g.bf, g.plan.cfg, st));
  return VP_OK;
}

int prepare_weights(vp_bfmnet* h, hipStream_t st) {
  BfmModel& m = h->m;
  auto fold = [&](const ConvBN& c) -> hipError_t {
    return launch_fold_bn(h->params + c.w, h->params + c.beta, h->params + c.mean, h->params + c.var, 1e-3f, c.wn, c.C,
                          h->folded + c.wf, h->folded + c.bf, st);
  };
  VP_HIP_CHECK(fold(m.stem));
  for (Block& b : m.blocks) {
    VP_HIP_CHECK(fold(b.expand)); VP_HIP_CHECK(fold(b.dw)); VP_HIP_CHECK(fold(b.project));
    if (b.shortcut) VP_HIP_CHECK(fold(b.sc));
  }
  VP_HIP_CHECK(fold(m.last));
  auto pack = [&](Gemm& g) -> hipError_t {
    return launch_pack_weights_one(g.plan.pack, g.which == 0 ? h->folded : h->params, h->packed + g.pk, g.bf, st);
  };
  for (Block& b : m.blocks) {
    VP_HIP_CHECK(pack(b.g_expand)); VP_HIP_CHECK(pack(b.g_project));
    if (b.shortcut) VP_HIP_CHECK(pack(b.g_sc));
  }
  for (Gemm* g : {&m.g_last, &m.g_enc, &m.g_rnn, &m.g_xg, &m.g_xc, &m.g_d0, &m.g_d1, &m.g_d2}) VP_HIP_CHECK(pack(*g));
  h->dirty = false;
  return VP_OK;
}

}  // namespace

extern "C" {

size_t vp_bfmnet_param_count(void) { BfmModel m; build_model(m); return m.nparams; }

int vp_bfmnet_param_info(int index, char* name, int name_cap, size_t* offset, int* ndim, int64_t shape[4]) {
  BfmModel m;
  build_model(m);
  if (index < 0 || index >= (int)m.manifest.size()) return VP_ERR_ARG;
  const PInfo& p = m.manifest[index];
  if (name && name_cap > 0) { strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (offset) *offset = p.off;
  if (ndim) *ndim = p.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  return VP_OK;
}

size_t vp_bfmnet_workspace_bytes(const vp_bfmnet_desc* d) {
  if (!bfm_ok(d)) return 0;
  vp_bfmnet* h = new vp_bfmnet{};
  h->d = *d;
  build_model(h->m);
  const size_t n = bfm_carve(h, nullptr);
  delete h;
  return n;
}

int vp_bfmnet_create(const vp_bfmnet_desc* d, void* workspace, size_t bytes, const float* params, void* stream, vp_bfmnet_t** out) {
  if (!bfm_ok(d) || !workspace || !params || !out) { set_err("vp_bfmnet_create: bad argument"); return VP_ERR_ARG; }
  vp_bfmnet* h = new vp_bfmnet{};
  h->d = *d;
  build_model(h->m);
  if (bytes < bfm_carve(h, nullptr)) { delete h; set_err("vp_bfmnet_create: workspace too small"); return VP_ERR_WORKSPACE; }
  bfm_carve(h, (char*)workspace);
  h->params = params;
  h->dirty = true;
  VP_HIP_CHECK(hipMemsetAsync(h->zeros, 0, 256, (hipStream_t)stream));
  *out = h;
  return VP_OK;
}

void vp_bfmnet_destroy(vp_bfmnet_t* h) { delete h; }

int vp_bfmnet_params_changed(vp_bfmnet_t* h) { if (!h) return VP_ERR_ARG; h->dirty = true; return VP_OK; }

int vp_bfmnet_forward(vp_bfmnet_t* h, const float* ears, const float* mfccs, const int* seq_len, float* out, void* stream) {
  if (!h || !ears || !mfccs || !seq_len || !out) { set_err("vp_bfmnet_forward: null argument"); return VP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (h->dirty && (rc = prepare_weights(h, st))) return rc;
  BfmModel& m = h->m;
  const int B = h->d.batch, T = h->d.frames, H = h->T5;
  int W = h->d.num_mel_bins, pl, Wo;
  // stem: conv [9,5] stride [1,2] SAME on [B, 5T, 80, 1]
  same_pad(W, 5, 2, &pl, &Wo);
  // bf16-trunk mode: the 6x-expanded tensors (ex, dwb: the bytes of this network) and every 1x1-conv operand are bf16; the narrow
  // residual stream (block inputs / outputs) stays f32 so that rounding does not compound over the 17 blocks - an expansion conv
  // reads a bf16 COPY of it (nb16), the projection conv adds its f32 result into it
  const int tb = h->d.trunk_dtype == VP_BF16 ? 1 : 0;
  VP_HIP_CHECK(launch_conv_first(mfccs, h->folded + m.stem.wf, h->folded + m.stem.bf, h->n0, 0, B, H, W, Wo, 32, 4, pl, st));
  W = Wo;
  float* cur = h->n0;
  float* alt = h->n1;
  auto operand = [&](const float* t, int c) -> const void* {      // what a trunk GEMM reads for the narrow tensor t
    if (!tb) return t;
    (void)launch_cvt_f32_bf16(t, h->nb16, (size_t)B * H * W * c, st);
    return h->nb16;
  };
  for (Block& b : m.blocks) {
    const void* xin = operand(cur, b.cin);
    if ((rc = run_gemm(h, b.g_expand, xin, h->ex, ACT_RELU6, 0, st))) return rc;
    // depthwise + projection: one kernel where the block is fusable (the depthwise result never reaches HBM), else two
    const bool fused = b.fusable && bfm_dwproj_knob();
    auto dw_project = [&](float* dst) -> int {
      if (fused) {
        VP_HIP_CHECK(launch_dwproj(h->ex, h->folded + b.dw.wf, (const float*)(h->packed + b.g_project.pk), b.g_project.plan.a.wp_rows, h->folded + b.project.bf, dst, 1,
                                   B, H, W, b.cexp, b.cout, st));
        return VP_OK;
      }
      VP_HIP_CHECK(launch_dwconv7x3(h->ex, h->folded + b.dw.wf, h->folded + b.dw.bf, h->dwb, tb, B, H, W, b.cexp, st));
      return run_gemm(h, b.g_project, h->dwb, dst, ACT_NONE, 1, st, 1);
    };
    if (b.shortcut) {
      if ((rc = run_gemm(h, b.g_sc, xin, alt, ACT_NONE, 0, st, 1))) return rc;
      if ((rc = dw_project(alt))) return rc;
      float* t = cur; cur = alt; alt = t;
    } else {
      if ((rc = dw_project(cur))) return rc;   // residual add in place
    }
    if (b.pool) {   // max_pooling2d([2,2], strides [1,2], 'same'): time pad (0,1), mel pad (0, W odd)
      const int Wn = (W + 1) / 2;
      VP_HIP_CHECK(launch_maxpool_same(cur, alt, 0, 0, B, H, W, b.cout, 2, 2, 1, 2, 0, 0, H, Wn, st));
      float* t = cur; cur = alt; alt = t;
      W = Wn;
    }
  }
  if ((rc = run_gemm(h, m.g_last, operand(cur, 256), alt, ACT_RELU, 0, st, 1))) return rc;
  // MfccEncoder pool [5,3] stride [5,3] SAME -> [B, T, 1, 256]  (bfmnet.py:35)
  {
    int pt, ph, pw2, wo2;
    same_pad(H, 5, 5, &pt, &ph);
    same_pad(W, 3, 3, &pw2, &wo2);
    if (ph != T || wo2 != 1) { set_err("vp_bfmnet_forward: unexpected pooled size %dx%d", ph, wo2); return VP_ERR_STATE; }
    VP_HIP_CHECK(launch_maxpool_same(alt, h->pooled, 0, 0, B, H, W, 256, 5, 3, 5, 3, pt, pw2, ph, wo2, st));
  }
  if ((rc = run_gemm(h, m.g_enc, h->pooled, h->enc, ACT_LEAKY, 0, st))) return rc;
  if ((rc = run_gemm(h, m.g_rnn, h->enc, h->c1, ACT_LEAKY, 0, st))) return rc;
  if ((rc = run_gemm(h, m.g_xg, h->c1, h->xg, ACT_NONE, 0, st))) return rc;
  if ((rc = run_gemm(h, m.g_xc, h->c1, h->xc, ACT_NONE, 0, st))) return rc;
  VP_HIP_CHECK(launch_gru_seq(h->xg, h->xc, h->params + m.gk + (size_t)256 * 512, h->params + m.ck + (size_t)256 * 256, seq_len, h->rnn, B, T, st));
  if ((rc = run_gemm(h, m.g_d0, h->rnn, h->dd0, ACT_LEAKY, 0, st))) return rc;
  if (h->drop0) VP_HIP_CHECK(launch_mul_inplace(h->dd0, h->drop0, (size_t)B * T * 128, st));
  if ((rc = run_gemm(h, m.g_d1, h->dd0, h->dd1, ACT_LEAKY, 0, st))) return rc;
  if (h->drop1) VP_HIP_CHECK(launch_mul_inplace(h->dd1, h->drop1, (size_t)B * T * 64, st));
  if ((rc = run_gemm(h, m.g_d2, h->dd1, out, ACT_NONE, 0, st))) return rc;
  VP_HIP_CHECK(launch_add_ears(out, ears, B * T, st));
  return VP_OK;
}

int vp_bfmnet_set_decoder_dropout(vp_bfmnet_t* h, const float* mask0, const float* mask1) {
  if (!h) { set_err("vp_bfmnet_set_decoder_dropout: null handle"); return VP_ERR_ARG; }
  h->drop0 = mask0; h->drop1 = mask1;
  return VP_OK;
}

int vp_bfmnet_tensor(vp_bfmnet_t* h, const char* name, void** ptr, int64_t shape[4]) {
  if (!h || !name || !ptr) return VP_ERR_ARG;
  const std::string s(name);
  const int B = h->d.batch, T = h->d.frames;
  float* p = nullptr; int c = 256;
  if (s == "MfccEncoder") p = h->enc;
  else if (s == "RNNModule") p = h->rnn;
  else if (s == "pooled") p = h->pooled;
  else return VP_ERR_ARG;
  *ptr = p;
  if (shape) { shape[0] = B; shape[1] = T; shape[2] = c; shape[3] = 1; }
  return VP_OK;
}

}  // extern "C"
