// PixReferNet step executor: the whole G + 3xD + VGG + losses forward, and both backward passes, as a
// fixed sequence of kernel launches on one HIP stream (no host sync, no allocation: hipGraph-capturable).
//
// Follows voicepuppet/pixrefer/pixrefer.py:59-438 and vgg_simple.py:96-162.
// Design (MI355X-first, not the TF graph):
//   * every conv output is stored ONCE, raw (pre-BN); act(scale*y + shift) is materialised once per activation kind the
//     consumers need (the LDS-DMA loaders move plain bytes), the skip concats are virtual (two source pointers);
//   * the three discriminator applications run as ONE batch of 3N with three batch-norm groups;
//   * the VGG trunk runs on the 2N batch [real | fake]; its backward only on the fake half;
//   * weights live as f32 masters in flat arenas (also the all-reduce buffers) and are re-packed
//     into chunk-major [class][K chunk][row][32 k] blocks of the compute dtype once per parameter update.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "conv_ops.h"
#include "errors.h"
#include "launch.h"
#include "smallp_args.h"
#include "vp_common.h"

namespace vp {

thread_local char g_err[512] = "";
void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct Arena {
  char* base;
  size_t off, cap;
  std::vector<std::pair<size_t, size_t>>* track = nullptr;   // (offset, bytes) of every carve: vp_pixrefer_validate_plan
  void* alloc(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    void* p = base ? base + off : nullptr;
    if (track) track->push_back({off, bytes});
    off += bytes;
    return p;
  }
};

struct BnBuf {
  float *a, *b, *mu, *rstd, *c1, *c2, *c1g, *c2g;   // c1g / c2g: the generator-loss pass through the discriminator
  // backward sums written by the conv epilogue that completes this tensor's gradient (IgemmArgs::bst_*): partial rows [group][chunks][2][C]
  // of this tensor's OWN buffer (the launch and the batch-norm backward may sit on different streams with other layers' work in between)
  double *pb, *pbg;
};

struct Tens {
  std::string name;
  void* y = nullptr;      // raw conv output (or packed network input)
  void* dz = nullptr;     // gradient buffer (w.r.t. the normalised tensor, then in place w.r.t. y)
  int N = 0, H = 0, W = 0, C = 0;
  bool is_f32 = false;
  bool has_bn = false;
  bool is_input = false;
  bool dz_written = false;
  void* dz2 = nullptr;    // discriminator tensors: gradient buffer of the generator-loss pass (fake group only), so that pass can run
  bool dz2_written = false;   // concurrently with the discriminator-loss pass, which owns dz
  void* xa[3] = {nullptr, nullptr, nullptr};   // materialised act(bn(y)) per vp::Act (lrelu, relu) for the consumers
  bool need_act[3] = {false, false, false};
  BnBuf bn{};
  int producer = -1;          // layer that writes this tensor (-1: network input / pool)
  int n_bwd_consumers = 0;    // layers whose backward-data pass writes into dz
  int dz_writes = 0;          // ... of which have run in the current backward pass
  bool bn_bwd_done = false;   // the batch-norm backward of this tensor ran inside the last writer's launch (conv_smallp.hip, SP_BWD_BN)
  // bf16 plans: a batch-normalised tensor whose producer and every gradient contribution run on the few-pixel kernel (<= 256 pixels per
  // launch class, i.e. N*1*1 .. N*8*8 values per channel at small batches) is kept in FLOAT32: y (raw output) is a float buffer and the
  // gradient accumulates in dz32; dz stays the bf16 dL/dy the producer's weight / data gradients read (smallp_args.h `hi`).  The
  // batch-norm backward over so few values subtracts two projections from the tensor - with bf16 storage the rounding of the tensor was
  // of the size of the result (element-wise gradient errors of 0.35-0.75 against the float64 oracle).
  bool hi = false;
  void* dz32 = nullptr;
  double* cs_part = nullptr;        // bias gradient of the producing layer (no batch-norm) from the epilogue of the launch that completes dz (conv_dc64.hip CSUM): [512][2][64]
  int cs_chunks = 0;                // run time: rows the last launch wrote in this backward pass (0: the column-sum pass runs)
  int pb_cap = 0, pbg_cap = 0;      // rows per group the buffers bn.pb / bn.pbg hold (plan time: the largest table a gradient-completing launch writes; 0: none)
  int bst_chunks = 0, bst_chunks_g = 0;   // run time: rows per group the last launch DID write in this backward pass (0: the reduce kernel runs)
  size_t elems() const { return (size_t)N * H * W * C; }
};

struct Layer {
  std::string scope;       // TF variable scope
  ConvGeomX g;
  int src[2] = {-1, -1};
  int nsrc = 1;
  int in_act = ACT_NONE;
  int out_act = ACT_NONE;  // VGG: relu in the epilogue
  int out = -1;
  bool has_bn = false;
  size_t w_off = 0, b_off = 0, gamma_off = 0, beta_off = 0;
  size_t pk_fwd = 0, pk_bwd[2] = {0, 0}, pk_bwd_alt[2] = {0, 0};   // pk_bwd_alt == pk_bwd unless the two batch sizes run different kernel families
  bool need_bwd[2] = {false, false};
  IgemmPlan fwd, bwd[2], bwd_alt[2];   // bwd_alt: discriminator G-loss pass (batch N)
  // both data gradients of a two-source layer (a decoder and its skip connection, merged_encoder_2) as ONE two-output GEMM over the
  // shared dY (IgemmArgs::split_c): one launch on the critical chain of the backward pass instead of two, dY read once
  IgemmPlan bwd_pair;
  size_t pk_bwd_pair = 0;
  bool has_pair = false;
  IgemmPlan fwd_half;                  // VGG: forward of one half (N) of the [real | fake] batch, so the real half can run early on the side stream
  size_t pk_fwd_half = 0;
  bool has_fwd_half = false;
  WgradPlan wg;
  // one-output-channel stride-1 conv (D layer_5) run as a GEMM over taps (TapArgs): x is read once per pass, not 16 times
  size_t desc0 = 0, desc1 = 0;   // this layer's pack descriptors: Net::descs[desc0, desc1)
  bool tapgemm = false;
  float* tap_S = nullptr;    // [N,Hin,Win,16] f32
  void* tap_dyS = nullptr;   // [N,Hin,Win,16] T
  unsigned* sp_cnt_fwd = nullptr;               // few-pixel kernel (conv_smallp.hip): arrival counters of this layer's launches
  unsigned* sp_cnt_bwd[2] = {nullptr, nullptr};
  unsigned* sp_cnt_pair = nullptr;
};

struct ParamInfo { std::string name; size_t off; int ndim; int64_t shape[4]; };

struct Net {
  std::vector<Tens> t;
  std::vector<Layer> l;
  std::vector<ParamInfo> manifest;
  size_t nparams = 0;
  int batch = 0, groups = 1;
  float* params = nullptr;
  float* grads = nullptr;
  char* packed = nullptr;       // packed weights of this net (compute dtype)
  size_t packed_elems = 0;
  std::vector<PackDesc> descs;
  PackDesc* d_descs = nullptr;
  unsigned* sp_cnt = nullptr;   // arrival counters of the few-pixel launches of this net (zeroed at create, left zero by every launch)
  size_t sp_cnt_n = 0;
};

}  // namespace vp

using namespace vp;

struct vp_pixrefer {
  vp_pixrefer_desc d;
  int bf16, es;
  Net G, D, V;
  // shared device buffers
  void *gin, *gfg, *din, *vin;
  float *y4, *o4, *outputs, *outputs_fg, *logits, *predict, *losses;
  void *dl_d, *dl_g, *d_din, *d_vin, *dy4, *df3;
  void *vpool1, *vpool2, *d_vpool1, *d_vpool2;
  double *comp_partial, *perc_partial, *bn_partial;
  char* scratch;
  char* scratch2;             // split-K / slab scratch and batch-norm partials of the side stream (backward_d when overlapped)
  double* bn_partial2;
  char* scratch3;             // ... and of the branch stream (the generator's foreground encoder branch, forward and backward)
  double* bn_partial3;
  char* scratch4;             // ... and of the second branch stream (the foreground encoder branch of the BACKWARD pass, see branch2)
  double* bn_partial4;
  hipStream_t side, branch;
  hipStream_t branch2;             // backward: the foreground encoder chain, so that it does not queue behind the weight-gradient backlog of `branch`
                                   // (== branch with vp_pixrefer_desc::streams = 3 / vp_pixrefer_use_streams(h, 3): a host that brings streams of its own wants three)
  hipEvent_t ev_b2join;
  bool use_b2;                     // vp_pixrefer_use_streams: false = the backward pass keeps to three streams (the host runs a stream of its own)
  hipEvent_t ev_fork, ev_join, ev_bfork, ev_bjoin;
  hipEvent_t ev_upd_b, ev_upd_m;   // fused update: a generator bucket's weight gradients (branch stream) / data gradients (caller's stream) are done
  bool overlap, forked;
  int bst_count;              // launches since vp_pixrefer_create that carried such sums (vp_pixrefer_counter("bwd_sums_launches"): tests)
  int bst_on;                 // vp_pixrefer_set_option("bwd_sums_in_epilogue", default 1; 2 = every launch that can): the batch-norm backward's sums from the epilogue of the launch that completes the gradient
  int vgg_fork_layer;         // vp_pixrefer_set_option("vgg_real_fork"): generator layer (TF scope order) in front of which the side stream starts the VGG pass of the real half (0: at once)
  bool store_first_raw;       // vp_pixrefer_set_option("store_first_raw"): the first layers also store their raw output (tests / debugging; see first_layer_acts_fused)
  bool ov_on;                 // vp_pixrefer_set_option("overlap"): spread the step over the executor's streams (false: everything on the caller's)
  int dfork_point;            // ... ("d_backward_fork"): where vp_pixrefer_backward starts the discriminator-loss pass on the side stream (0 / 1 / 2)
  bool dsplit_on;             // ... ("d_beside_vgg"): discriminator passes on the branch stream beside the VGG passes
  // vp_tune("phase_marks", 1): HIP events on the caller's stream at the phase boundaries of a step (vp_pixrefer_phase_ms)
  hipEvent_t mark[64];
  int nmark;
  bool marks_made;
  int dfork_pending;          // vp_pixrefer_backward only: where the generator-loss pass still has to start the discriminator-loss pass on the side stream (0: nowhere)
  void* zeros;
  size_t scratch_bytes;
  int n_comp, n_perc;
  // vp_pixrefer_backward_update: Adam state of the two optimisers + this step's hyper-parameters; active only inside that call
  struct { bool active, d_done; float *m_g, *v_g, *m_d, *v_d; float lr_t_g, lr_t_d, beta1, beta2, eps; } upd;
  int upd_mask;               // vp_pixrefer_update_bucket: buckets of the current step already updated + re-packed (bits 0-2 generator, 3 discriminator)
  bool params_dirty;
  bool vgg_dirty;             // set by vp_pixrefer_params_changed (a host wrote the arenas), cleared by the next forward; vp_pixrefer_optimizer_stepped leaves it
  const float *in_targets, *in_masks;
};

namespace vp {

// ------------------------------------------------------------------------------------------------
// network construction
// ------------------------------------------------------------------------------------------------
static int add_tensor(Net& n, const std::string& name, int N, int H, int W, int C, bool has_bn, bool is_input) {
  Tens t;
  t.name = name; t.N = N; t.H = H; t.W = W; t.C = C; t.has_bn = has_bn; t.is_input = is_input;
  n.t.push_back(t);
  return (int)n.t.size() - 1;
}

static void add_param(Net& n, const std::string& name, std::initializer_list<int64_t> shape, size_t* off_out) {
  ParamInfo p;
  p.name = name; p.off = n.nparams; p.ndim = (int)shape.size();
  size_t cnt = 1;
  int i = 0;
  for (auto s : shape) { p.shape[i++] = s; cnt *= (size_t)s; }
  for (; i < 4; ++i) p.shape[i] = 1;
  *off_out = n.nparams;
  n.nparams += cnt;
  n.manifest.push_back(p);
}

// conv/deconv layer reading tensors src[] and producing a new tensor
static int add_layer(Net& n, const std::string& prefix, const std::string& scope, int kind, int ks, int stride, int pad,
                     std::initializer_list<int> srcs, int cin_real, int cout, bool has_bn, int in_act, int out_act,
                     const char* wname, const char* bname) {
  Layer L;
  L.scope = scope;
  L.nsrc = 0;
  int cin = 0;
  for (int s : srcs) {
    L.src[L.nsrc++] = s; cin += n.t[s].C;
    if (in_act != ACT_NONE) n.t[s].need_act[in_act] = true;
  }
  const Tens& t0 = n.t[L.src[0]];
  L.g = make_geom(kind, ks, stride, pad, n.batch, t0.H, t0.W, cin, cin_real, cout);
  L.in_act = in_act; L.out_act = out_act; L.has_bn = has_bn;
  const std::string base = prefix + scope + "/";
  if (kind == 0) add_param(n, base + wname, {ks, ks, cin_real, cout}, &L.w_off);
  else add_param(n, base + wname, {4, 4, cout, cin_real}, &L.w_off);
  add_param(n, base + bname, {cout}, &L.b_off);
  if (has_bn) {
    add_param(n, base + "batch_normalization/gamma", {cout}, &L.gamma_off);
    add_param(n, base + "batch_normalization/beta", {cout}, &L.beta_off);
  }
  L.out = add_tensor(n, scope, n.batch, L.g.Hout, L.g.Wout, L.g.CoutT, has_bn, false);
  n.t[L.out].producer = (int)n.l.size();
  n.l.push_back(L);
  return L.out;
}

static void build_generator(Net& n, int N, int H, int ngf, int groups = 1) {
  n.batch = N; n.groups = groups;
  const int tin = add_tensor(n, "inputs", N, H, H, 8, false, true);
  const int tfg = add_tensor(n, "fg_inputs", N, H, H, 8, false, true);
  const char* CK = "conv2d/kernel"; const char* CB = "conv2d/bias";
  const char* DK = "conv2d_transpose/kernel"; const char* DB = "conv2d_transpose/bias";
  const std::string P = "generator/";
  // encoders (pixrefer.py:169-213): act -> conv -> BN
  int e1 = add_layer(n, P, "encoder_1", 0, 4, 2, 1, {tin}, 6, ngf, false, ACT_NONE, ACT_NONE, CK, CB);
  int e2 = add_layer(n, P, "encoder_2", 0, 4, 2, 1, {e1}, ngf, ngf * 2, true, ACT_LRELU, ACT_NONE, CK, CB);
  int e3 = add_layer(n, P, "encoder_3", 0, 4, 2, 1, {e2}, ngf * 2, ngf * 2, true, ACT_LRELU, ACT_NONE, CK, CB);
  int e4 = add_layer(n, P, "encoder_4", 0, 4, 2, 1, {e3}, ngf * 2, ngf * 4, true, ACT_LRELU, ACT_NONE, CK, CB);
  int f1 = add_layer(n, P, "encoder_fg_1", 0, 4, 2, 1, {tfg}, 3, ngf, false, ACT_NONE, ACT_NONE, CK, CB);
  int f2 = add_layer(n, P, "encoder_fg_2", 0, 4, 2, 1, {f1}, ngf, ngf * 2, true, ACT_LRELU, ACT_NONE, CK, CB);
  int f3 = add_layer(n, P, "encoder_fg_3", 0, 4, 2, 1, {f2}, ngf * 2, ngf * 2, true, ACT_LRELU, ACT_NONE, CK, CB);
  int f4 = add_layer(n, P, "encoder_fg_4", 0, 4, 2, 1, {f3}, ngf * 2, ngf * 4, true, ACT_LRELU, ACT_NONE, CK, CB);
  // merged encoder on concat(e4, f4) (pixrefer.py:215-232)
  int m2 = add_layer(n, P, "merged_encoder_2", 0, 4, 2, 1, {e4, f4}, ngf * 8, ngf * 4, true, ACT_LRELU, ACT_NONE, CK, CB);
  int m3 = add_layer(n, P, "merged_encoder_3", 0, 4, 2, 1, {m2}, ngf * 4, ngf * 8, true, ACT_LRELU, ACT_NONE, CK, CB);
  int m4 = add_layer(n, P, "merged_encoder_4", 0, 4, 2, 1, {m3}, ngf * 8, ngf * 8, true, ACT_LRELU, ACT_NONE, CK, CB);
  int m5 = add_layer(n, P, "merged_encoder_5", 0, 4, 2, 1, {m4}, ngf * 8, ngf * 8, true, ACT_LRELU, ACT_NONE, CK, CB);
  // decoders (pixrefer.py:234-276): relu -> deconv -> BN, skip = virtual concat
  int d5 = add_layer(n, P, "merged_decoder_5", 1, 4, 2, 1, {m5}, ngf * 8, ngf * 8, true, ACT_RELU, ACT_NONE, DK, DB);
  int d4 = add_layer(n, P, "merged_decoder_4", 1, 4, 2, 1, {d5, m4}, ngf * 16, ngf * 8, true, ACT_RELU, ACT_NONE, DK, DB);
  int d3 = add_layer(n, P, "merged_decoder_3", 1, 4, 2, 1, {d4, m3}, ngf * 16, ngf * 4, true, ACT_RELU, ACT_NONE, DK, DB);
  int d2 = add_layer(n, P, "merged_decoder_2", 1, 4, 2, 1, {d3, m2}, ngf * 8, ngf * 4, true, ACT_RELU, ACT_NONE, DK, DB);
  int c4 = add_layer(n, P, "merged2_decoder_4", 1, 4, 2, 1, {d2, e4}, ngf * 8, ngf * 2, true, ACT_RELU, ACT_NONE, DK, DB);
  int c3 = add_layer(n, P, "merged2_decoder_3", 1, 4, 2, 1, {c4, e3}, ngf * 4, ngf * 2, true, ACT_RELU, ACT_NONE, DK, DB);
  int c2 = add_layer(n, P, "merged2_decoder_2", 1, 4, 2, 1, {c3, e2}, ngf * 4, ngf, true, ACT_RELU, ACT_NONE, DK, DB);
  add_layer(n, P, "decoder_1", 1, 4, 2, 1, {c2, e1}, ngf * 2, 4, false, ACT_RELU, ACT_NONE, DK, DB);
}

static void build_discriminator(Net& n, int N, int H, int ndf) {
  n.batch = 3 * N; n.groups = 3;
  const int tin = add_tensor(n, "d_inputs", 3 * N, H, H, 8, false, true);
  const char* CK = "conv2d/kernel"; const char* CB = "conv2d/bias";
  const std::string P = "discriminator/";
  // pixrefer.py:103-134: conv -> BN -> lrelu; the lrelu is applied by the next layer's loader
  int l1 = add_layer(n, P, "layer_1", 0, 4, 2, 1, {tin}, 6, ndf, false, ACT_NONE, ACT_NONE, CK, CB);
  int l2 = add_layer(n, P, "layer_2", 0, 4, 2, 1, {l1}, ndf, ndf * 2, true, ACT_LRELU, ACT_NONE, CK, CB);
  int l3 = add_layer(n, P, "layer_3", 0, 4, 2, 1, {l2}, ndf * 2, ndf * 4, true, ACT_LRELU, ACT_NONE, CK, CB);
  int l4 = add_layer(n, P, "layer_4", 0, 4, 1, 1, {l3}, ndf * 4, ndf * 8, true, ACT_LRELU, ACT_NONE, CK, CB);
  add_layer(n, P, "layer_5", 0, 4, 1, 1, {l4}, ndf * 8, 1, false, ACT_LRELU, ACT_NONE, CK, CB);
}

static void build_vgg(Net& n, int N, int H) {
  n.batch = 2 * N; n.groups = 1;
  const int tin = add_tensor(n, "v_inputs", 2 * N, H, H, 8, false, true);
  const std::string P = "vgg_16/";
  // vgg_simple.py:138-151: conv3x3 s1 SAME + bias + relu; pools are separate ops between the blocks
  int c11 = add_layer(n, P, "conv1/conv1_1", 0, 3, 1, 1, {tin}, 3, 64, false, ACT_NONE, ACT_RELU, "weights", "biases");
  add_layer(n, P, "conv1/conv1_2", 0, 3, 1, 1, {c11}, 64, 64, false, ACT_NONE, ACT_RELU, "weights", "biases");
  const int p1 = add_tensor(n, "pool1", 2 * N, H / 2, H / 2, 64, false, false);
  int c21 = add_layer(n, P, "conv2/conv2_1", 0, 3, 1, 1, {p1}, 64, 128, false, ACT_NONE, ACT_RELU, "weights", "biases");
  add_layer(n, P, "conv2/conv2_2", 0, 3, 1, 1, {c21}, 128, 128, false, ACT_NONE, ACT_RELU, "weights", "biases");
  const int p2 = add_tensor(n, "pool2", 2 * N, H / 4, H / 4, 128, false, false);
  int c31 = add_layer(n, P, "conv3/conv3_1", 0, 3, 1, 1, {p2}, 128, 256, false, ACT_NONE, ACT_RELU, "weights", "biases");
  int c32 = add_layer(n, P, "conv3/conv3_2", 0, 3, 1, 1, {c31}, 256, 256, false, ACT_NONE, ACT_RELU, "weights", "biases");
  add_layer(n, P, "conv3/conv3_3", 0, 3, 1, 1, {c32}, 256, 256, false, ACT_NONE, ACT_RELU, "weights", "biases");
}

static int epi_stat_chunks(const IgemmPlan& p, int batch, int groups);
constexpr int BST_MAX_CHUNKS = 8192;      // rows per group of a backward-sums table (the finalize walks them 512 at a time)

// plans + packed-weight layout for one net.  alt_batch > 0: also plan bwd-data for that batch (D, G-loss pass)
static void plan_net(Net& n, int bf16, bool training, bool want_wgrad, int alt_batch, size_t* scratch_max, bool fwd_halves = false) {
  size_t pk = 0;
  auto take = [&](IgemmPlan& p) {
    p.pack.dst_off = pk;
    pk += (p.pack_elems + 63) & ~(size_t)63;
    n.descs.push_back(p.pack);
    if (p.partial_bytes > *scratch_max) *scratch_max = p.partial_bytes;
  };
  for (Layer& L : n.l) {
    L.desc0 = n.descs.size();
    L.tapgemm = training && L.g.kind == 0 && L.g.stride == 1 && L.g.ks == 4 && L.g.Cout == 1 && L.nsrc == 1 && !L.has_bn &&
                L.g.Cin == L.g.Cin_real;
    if (L.tapgemm) {
      // S = x . W^T as a 1x1 conv with 16 output channels: row t of the packed matrix is W[kh,kw,:,0] (HWIO, Cout = 1)
      ConvGeomX g1 = make_geom(0, 1, 1, 0, L.g.N, L.g.Hin, L.g.Win, L.g.Cin, L.g.Cin, 16);
      L.fwd = plan_fwd(g1, L.w_off, bf16);
      L.fwd.pack.s_row = L.g.Cin; L.fwd.pack.s_ch = 1; L.fwd.pack.s_kh = 0; L.fwd.pack.s_kw = 0;
    } else {
      L.fwd = plan_fwd(L.g, L.w_off, bf16);
      // the generator's few-pixel bottleneck: conv + K-split combine + batch-norm + activations in one launch (conv_smallp.hip)
      if (n.groups == 1 && alt_batch == 0 && plan_smallp_eligible(L.fwd, L.g.Cout, bf16, n.t[L.src[0]].C, L.nsrc > 1 ? n.t[L.src[1]].C : 0))
        plan_make_smallp(L.fwd, L.g.Cout, bf16);
      else if (plan_patch_eligible(L.fwd, L.g.Cout, bf16, L.nsrc == 1 && n.t[L.src[0]].C == L.g.Cin)) plan_make_patch(L.fwd, L.g.Cout, bf16);
      else if (plan_patch2_eligible(L.fwd, L.g.Cout, bf16, n.t[L.src[0]].C, L.nsrc > 1 ? n.t[L.src[1]].C : 0)) plan_make_patch2(L.fwd, L.g.Cout, bf16);
      else if (L.has_bn && L.out_act == ACT_NONE && !L.tapgemm && plan_s2c64_eligible(L.fwd, L.g.Cout, bf16, n.t[L.src[0]].C, L.nsrc > 1 ? n.t[L.src[1]].C : 0))
        plan_make_s2c64(L.fwd);
    }
    take(L.fwd);
    L.pk_fwd = L.fwd.pack.dst_off;
    if (fwd_halves && alt_batch > 0 && !L.tapgemm && !L.has_bn) {
      ConvGeomX g2 = L.g;
      g2.N = alt_batch;
      L.fwd_half = plan_fwd(g2, L.w_off, bf16);
      // the half-batch launches take the kernel family of the layer's full-batch plan (the minimum-grid rule would otherwise pick the
      // patch kernel for 2N images and the gather kernel for N at small batches: same result up to the order of the K sum, but then
      // the overlapped step and the single-stream step are no longer bit-identical - seen at N = 8, 256x256)
      if (L.fwd.a.patch == 1 && plan_patch_eligible(L.fwd_half, g2.Cout, bf16, L.nsrc == 1 && n.t[L.src[0]].C == g2.Cin, true))
        plan_make_patch(L.fwd_half, g2.Cout, bf16);
      const PackDesc &pa = L.fwd.pack, &pb = L.fwd_half.pack;
      if (L.fwd_half.a.splitk == 1) {
        if (pa.kswap != pb.kswap || pa.perm != pb.perm || pa.kc != pb.kc || pa.Kpad != pb.Kpad || pa.rows_pad != pb.rows_pad) {
          take(L.fwd_half);
          L.pk_fwd_half = L.fwd_half.pack.dst_off;
        } else {
          L.pk_fwd_half = L.pk_fwd;
        }
        L.has_fwd_half = true;
      }
    }
    if (!training) continue;
    int row0 = 0;
    for (int s = 0; s < L.nsrc; ++s) {
      const Tens& ts = n.t[L.src[s]];
      const int rows = ts.C;
      int rows_real = rows;
      if (ts.is_input) rows_real = L.g.Cin_real;
      // gradient w.r.t. network inputs is only needed for the discriminator (G loss) and VGG inputs
      L.need_bwd[s] = !ts.is_input || (n.groups == 3) || (n.l[0].g.ks == 3);
      if (L.need_bwd[s]) {
        L.bwd[s] = plan_bwd_data(L.g, L.w_off, row0, rows, rows_real, ts.C, bf16);
        if (alt_batch > 0) {
          ConvGeomX g2 = L.g;
          g2.N = alt_batch;
          L.bwd_alt[s] = plan_bwd_data(g2, L.w_off, row0, rows, rows_real, ts.C, bf16);
          // both batch sizes read ONE packed block: they must agree on its row order
          if (L.bwd_alt[s].a.rowperm != L.bwd[s].a.rowperm) {
            L.bwd[s].a.rowperm = L.bwd_alt[s].a.rowperm = 0;
            L.bwd[s].pack.perm = L.bwd_alt[s].pack.perm = 0;
          }
        }
        if (n.groups == 1 && alt_batch == 0 && !ts.is_input && plan_smallp_eligible(L.bwd[s], rows, bf16, L.g.CoutT, 0)) {
          plan_make_smallp(L.bwd[s], rows, bf16);
        } else {
          // backward-data of a stride-1 conv is a stride-1 conv over dY (one tensor of CoutT channels): patch kernel, per batch size
          if (plan_patch_eligible(L.bwd[s], rows, bf16, true)) plan_make_patch(L.bwd[s], rows, bf16);
          else if (plan_patch2_eligible(L.bwd[s], rows, bf16, L.g.CoutT, 0)) plan_make_patch2(L.bwd[s], rows, bf16);
          if (alt_batch > 0 && plan_patch_eligible(L.bwd_alt[s], rows, bf16, true)) plan_make_patch(L.bwd_alt[s], rows, bf16);
          else if (alt_batch > 0 && plan_patch2_eligible(L.bwd_alt[s], rows, bf16, L.g.CoutT, 0)) plan_make_patch2(L.bwd_alt[s], rows, bf16);
        }
        take(L.bwd[s]);
        L.pk_bwd[s] = L.pk_bwd_alt[s] = L.bwd[s].pack.dst_off;
        if (alt_batch > 0) {
          const PackDesc &pa = L.bwd[s].pack, &pb = L.bwd_alt[s].pack;
          if (pa.kswap != pb.kswap || pa.perm != pb.perm || pa.kc != pb.kc || pa.Kpad != pb.Kpad) {
            take(L.bwd_alt[s]);                       // the two batch sizes run different kernel families: one packed block each
            L.pk_bwd_alt[s] = L.bwd_alt[s].pack.dst_off;
          } else {
            L.bwd_alt[s].pack.dst_off = L.pk_bwd[s];
            if (L.bwd_alt[s].partial_bytes > *scratch_max) *scratch_max = L.bwd_alt[s].partial_bytes;
          }
        }
      }
      row0 += rows;
    }
    {
      const int c0 = n.t[L.src[0]].C, c1 = L.nsrc > 1 ? n.t[L.src[1]].C : 0;
      if (L.nsrc == 2 && n.groups == 1 && alt_batch == 0 && L.need_bwd[0] && L.need_bwd[1] && c0 == c1 && !n.t[L.src[0]].is_input &&
          !n.t[L.src[1]].is_input && L.bwd[0].a.patch == L.bwd[1].a.patch && (L.bwd[0].a.patch == 0 || L.bwd[0].a.patch == 3) &&
          L.g.Cin_real == c0 + c1 && c0 % 32 == 0) {     // (the two-output epilogues split at a 32-channel tile boundary: conv_smallp.hip, epi_store8)
        L.bwd_pair = plan_bwd_data(L.g, L.w_off, 0, c0 + c1, c0 + c1, c0, bf16);
        bool ok = true;
        if (L.bwd[0].a.patch == 3) {      // few-pixel layers: the pair runs on the few-pixel kernel too (first output with its batch-norm backward)
          ok = plan_smallp_eligible(L.bwd_pair, c0 + c1, bf16, L.g.CoutT, 0);
          if (ok) plan_make_smallp(L.bwd_pair, c0 + c1, bf16);
        }
        // the 256 -> 64 transposed convolution's pair (merged2_decoder_2: relu inputs): register-resident weights, conv_s2c64.hip
        if (ok && s2c64_pair_knob() && L.bwd[0].a.patch == 0 && L.in_act == ACT_RELU && plan_s2c64_eligible(L.bwd_pair, c0 + c1, bf16, L.g.CoutT, 0)) plan_make_s2c64(L.bwd_pair);
        if (ok) {
          L.bwd_pair.a.split_c = c0;
          take(L.bwd_pair);
          L.pk_bwd_pair = L.bwd_pair.pack.dst_off;
          L.has_pair = true;
        }
      }
    }
    if (want_wgrad) {
      if (L.tapgemm) {
        // dW[t][c] = sum_q dyS[q][t] x[q][c]: a 1x1 weight gradient with G = dyS (16 "channels") and D = x
        ConvGeomX gw = make_geom(0, 1, 1, 0, L.g.N, L.g.Hin, L.g.Win, 16, 16, L.g.Cin);
        L.wg = plan_wgrad(gw, bf16);
      } else {
        L.wg = plan_wgrad(L.g, bf16);
      }
      if (L.wg.partial_bytes > *scratch_max) *scratch_max = L.wg.partial_bytes;
    }
  }
  n.packed_elems = pk;
  {
    for (Tens& t : n.t) t.hi = bf16 && t.has_bn && n.groups == 1 && alt_batch == 0 && t.producer >= 0 && n.l[t.producer].fwd.a.patch == 3;
    if (training)
      for (Layer& L : n.l)
        for (int s = 0; s < L.nsrc; ++s)
          if (L.need_bwd[s] && L.bwd[s].a.patch != 3) n.t[L.src[s]].hi = false;
    // a consumer without an input activation reads the tensor's raw storage as T (fill_src): such a tensor stays T (no consumer of a
    // batch-normalised tensor of the reference's nets does - a plan-time rule instead of a misread at run time)
    for (Layer& L : n.l)
      for (int s = 0; s < L.nsrc; ++s)
        if (L.in_act == ACT_NONE) n.t[L.src[s]].hi = false;
  }
  // backward sums from the epilogue of the launch that completes a batch-normalised tensor's gradient: the largest table any of the
  // tensor's gradient contributions could write (which one is last is decided by the backward walk at run time)
  for (Tens& t : n.t) t.pb_cap = t.pbg_cap = 0;
  if (training)
    for (Layer& L : n.l) {
      auto cap = [&](Tens& t, const IgemmPlan& p, int batch, int groups, int& c) {
        if (!t.has_bn || t.hi) return;
        const int k = epi_stat_chunks(p, batch, groups);
        if (k > 0 && k <= BST_MAX_CHUNKS && k > c) c = k;
      };
      for (int s = 0; s < L.nsrc; ++s) {
        if (!L.need_bwd[s]) continue;
        Tens& ts = n.t[L.src[s]];
        cap(ts, L.bwd[s], n.batch, n.groups, ts.pb_cap);
        if (alt_batch > 0) cap(ts, L.bwd_alt[s], alt_batch, 1, ts.pbg_cap);
        if (L.has_pair) cap(ts, L.bwd_pair, n.batch, n.groups, ts.pb_cap);
        // the one-output-channel layer's own backward-data kernel (conv_cout1.hip) leaves one row per block, whatever the generic plan
        // could do (its 128-pixel tiles straddle the three groups of the discriminator-loss pass: 961 pixels per image)
        if (L.tapgemm && bf16 && ts.has_bn && !ts.hi && L.g.ks == 4 && L.g.stride == 1 && L.g.Cin == 512) {
          if (ts.pb_cap < 256) ts.pb_cap = 256;
          if (alt_batch > 0 && ts.pbg_cap < 256) ts.pbg_cap = 256;
        }
      }
    }
  for (size_t i = 0; i < n.l.size(); ++i) n.l[i].desc1 = i + 1 < n.l.size() ? n.l[i + 1].desc0 : n.descs.size();
  for (Tens& t : n.t) t.n_bwd_consumers = 0;
  n.sp_cnt_n = 0;
  for (Layer& L : n.l) {
    if (L.fwd.a.patch == 3) n.sp_cnt_n += smallp_counters(L.fwd.a);
    for (int s = 0; s < L.nsrc && training; ++s) {
      if (!L.need_bwd[s]) continue;
      n.t[L.src[s]].n_bwd_consumers++;
      if (L.bwd[s].a.patch == 3) n.sp_cnt_n += smallp_counters(L.bwd[s].a);
    }
    if (training && L.has_pair && L.bwd_pair.a.patch == 3) n.sp_cnt_n += smallp_counters(L.bwd_pair.a);
  }
}

// carve device buffers of a net out of the workspace
static void carve_net(Net& n, Arena& ar, int es, bool training) {
  for (Tens& t : n.t) {
    if (t.name == "decoder_1" || t.name == "layer_5") {   // thin f32 outputs live in the plan-level buffers
      continue;
    }
    if (!t.is_input) t.y = ar.alloc(t.elems() * (t.hi ? sizeof(float) : (size_t)es));
    for (int k = 1; k < 3; ++k) if (t.need_act[k]) t.xa[k] = ar.alloc(t.elems() * es);
    if (t.has_bn) {
      const size_t gc = (size_t)n.groups * t.C * sizeof(float);
      t.bn.a = (float*)ar.alloc(gc); t.bn.b = (float*)ar.alloc(gc);
      t.bn.mu = (float*)ar.alloc(gc); t.bn.rstd = (float*)ar.alloc(gc);
      t.bn.c1 = (float*)ar.alloc(gc); t.bn.c2 = (float*)ar.alloc(gc);
      t.bn.c1g = (float*)ar.alloc(gc); t.bn.c2g = (float*)ar.alloc(gc);
      t.bn.pb = (training && t.pb_cap) ? (double*)ar.alloc((size_t)n.groups * t.pb_cap * 2 * t.C * sizeof(double)) : nullptr;
      t.bn.pbg = (training && t.pbg_cap) ? (double*)ar.alloc((size_t)t.pbg_cap * 2 * t.C * sizeof(double)) : nullptr;
    }
  }
  for (Layer& L : n.l) {
    if (!L.tapgemm) continue;
    const size_t q = (size_t)L.g.N * L.g.Hin * L.g.Win * 16;
    L.tap_S = (float*)ar.alloc(q * sizeof(float));
    L.tap_dyS = ar.alloc(q * es);
  }
  n.packed = (char*)ar.alloc(n.packed_elems * es);
  n.d_descs = (PackDesc*)ar.alloc(n.descs.size() * sizeof(PackDesc));
  n.sp_cnt = (unsigned*)ar.alloc((n.sp_cnt_n + 1) * sizeof(unsigned));
  {
    unsigned* c = n.sp_cnt;
    for (Layer& L : n.l) {
      if (L.fwd.a.patch == 3) { L.sp_cnt_fwd = c; c += c ? smallp_counters(L.fwd.a) : 0; }
      for (int s = 0; s < L.nsrc && training; ++s)
        if (L.need_bwd[s] && L.bwd[s].a.patch == 3) { L.sp_cnt_bwd[s] = c; c += c ? smallp_counters(L.bwd[s].a) : 0; }
      if (training && L.has_pair && L.bwd_pair.a.patch == 3) { L.sp_cnt_pair = c; c += c ? smallp_counters(L.bwd_pair.a) : 0; }
    }
  }
}

static size_t carve_all(vp_pixrefer* h, char* base, size_t cap, std::vector<std::pair<size_t, size_t>>* track = nullptr) {
  Arena ar{base, 0, cap};
  ar.track = track;
  const vp_pixrefer_desc& d = h->d;
  const int N = d.batch, H = d.height, es = h->es;
  const size_t px = (size_t)N * H * H;
  h->gin = ar.alloc(px * 8 * es);
  h->gfg = ar.alloc(px * 8 * es);
  h->y4 = (float*)ar.alloc(px * 4 * sizeof(float));
  h->o4 = (float*)ar.alloc(px * 4 * sizeof(float));
  h->outputs = (float*)ar.alloc(px * 3 * sizeof(float));
  h->outputs_fg = (float*)ar.alloc(px * 3 * sizeof(float));
  h->losses = (float*)ar.alloc(8 * sizeof(float));
  h->G.t[0].y = h->gin; h->G.t[1].y = h->gfg;
  carve_net(h->G, ar, es, d.training);
  if (d.training) {
    const int hd = H / 8 - 2;
    const size_t M = (size_t)N * hd * hd;
    h->din = ar.alloc(3 * px * 8 * es);
    h->vin = ar.alloc(2 * px * 8 * es);
    h->logits = (float*)ar.alloc(3 * M * sizeof(float));
    h->predict = (float*)ar.alloc(2 * M * sizeof(float));
    h->dl_d = ar.alloc(3 * M * 8 * es);
    h->dl_g = ar.alloc(M * 8 * es);
    h->d_din = ar.alloc(px * 8 * es);
    h->d_vin = ar.alloc(px * 8 * es);
    h->dy4 = ar.alloc(px * 8 * es);
    h->D.t[0].y = h->din; h->V.t[0].y = h->vin;
    carve_net(h->D, ar, es, true);
    carve_net(h->V, ar, es, true);
    // gradient buffers.  generator + discriminator: one per non-input tensor.
    for (Tens& t : h->G.t) if (!t.is_input && t.name != "decoder_1") { t.dz = ar.alloc(t.elems() * es); if (t.hi) t.dz32 = ar.alloc(t.elems() * sizeof(float)); }
    for (Tens& t : h->D.t) if (!t.is_input && t.name != "layer_5") { t.dz = ar.alloc(t.elems() * es); t.dz2 = ar.alloc(t.elems() / 3 * es); }
    // 64-channel tensors without batch-norm (encoder_1, encoder_fg_1, layer_1): partial rows of the producer's bias gradient (conv_dc64.hip CSUM)
    for (Net* n : {&h->G, &h->D})
      for (Tens& t : n->t) if (h->bf16 && !t.is_input && !t.has_bn && t.C == 64) t.cs_part = (double*)ar.alloc((size_t)512 * 2 * 64 * sizeof(double));
    // VGG backward runs on the fake half only
    for (Tens& t : h->V.t) if (!t.is_input) t.dz = ar.alloc(t.elems() / 2 * es);
    h->n_comp = composite_nblocks(N, H * H);
    h->n_perc = perceptual_nblocks((size_t)N * (H / 4) * (H / 4) * 256, h->bf16);
    h->comp_partial = (double*)ar.alloc((size_t)h->n_comp * 2 * sizeof(double));
    h->perc_partial = (double*)ar.alloc((size_t)h->n_perc * sizeof(double));
  }
  h->zeros = ar.alloc(256);
  h->bn_partial = (double*)ar.alloc((size_t)1024 * 2 * 512 * sizeof(double));
  h->scratch = (char*)ar.alloc(h->scratch_bytes);
  if (d.training) {
    h->bn_partial2 = (double*)ar.alloc((size_t)1024 * 2 * 512 * sizeof(double));
    h->scratch2 = (char*)ar.alloc(h->scratch_bytes);
    h->bn_partial3 = (double*)ar.alloc((size_t)1024 * 2 * 512 * sizeof(double));
    h->scratch3 = (char*)ar.alloc(h->scratch_bytes);
    h->bn_partial4 = (double*)ar.alloc((size_t)1024 * 2 * 512 * sizeof(double));
    h->scratch4 = (char*)ar.alloc(h->scratch_bytes);
  }
  return ar.off + 256;
}

static void init_handle(vp_pixrefer* h, const vp_pixrefer_desc* d) {
  h->d = *d;
  h->bf16 = d->dtype == VP_BF16;
  h->es = h->bf16 ? 2 : 4;
  build_generator(h->G, d->batch, d->height, d->ngf, (!d->training && d->per_sample_bn) ? d->batch : 1);
  size_t smax = 0;
  plan_net(h->G, h->bf16, d->training, d->training, 0, &smax);
  if (d->training) {
    build_discriminator(h->D, d->batch, d->height, d->ndf);
    plan_net(h->D, h->bf16, true, true, d->batch, &smax);
    build_vgg(h->V, d->batch, d->height);
    // VGG backward (dX only) is planned for the fake half: batch N
    for (Layer& L : h->V.l) L.g.N = 2 * d->batch;
    plan_net(h->V, h->bf16, true, false, d->batch, &smax, true);
  }
  h->scratch_bytes = smax + 256;
  h->params_dirty = true;
  h->vgg_dirty = true;
}

// The batch-norm partial-sum workspace (bn_partial: 1024 chunk rows x 2 x 512 channels of f64) bounds the widths and the
// number of batch-norm groups a plan may have: channels <= 8 * ngf <= 512, groups * chunks <= 1024.
static bool valid_desc(const vp_pixrefer_desc* d) {
  if (!d) { set_err("pixrefer descriptor: null"); return false; }
  if (d->batch < 1 || d->height < 256 || d->height % 256 != 0) { set_err("pixrefer descriptor: batch %d / height %d (height must be a multiple of 256)", d->batch, d->height); return false; }
  if (d->ngf < 8 || (d->ngf & (d->ngf - 1)) || d->ndf < 8 || (d->ndf & (d->ndf - 1))) { set_err("pixrefer descriptor: ngf %d / ndf %d must be powers of two >= 8", d->ngf, d->ndf); return false; }
  if (d->ngf > 64 || d->ndf > 64) { set_err("pixrefer descriptor: ngf %d / ndf %d > 64 exceed the batch-norm workspace (512 channels)", d->ngf, d->ndf); return false; }
  if (!d->training && d->per_sample_bn && d->batch > 1024) { set_err("pixrefer descriptor: per-sample batch-norm supports at most 1024 frames per call (got %d)", d->batch); return false; }
  if (d->dtype != VP_F32 && d->dtype != VP_BF16) { set_err("pixrefer descriptor: dtype %d", d->dtype); return false; }
  if ((d->streams != 0 && d->streams != 1 && d->streams != 3 && d->streams != 4) || d->d_backward_fork < 0 || d->d_backward_fork > 3 || d->d_beside_vgg < 0 ||
      d->d_beside_vgg > 2) {
    set_err("pixrefer descriptor: schedule fields streams %d / d_backward_fork %d / d_beside_vgg %d", d->streams, d->d_backward_fork, d->d_beside_vgg);
    return false;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// execution helpers
// ------------------------------------------------------------------------------------------------
static void fill_src(const Net& n, const Layer& L, PixSrc& x, int group_n, int sample0, int group0, int es) {
  for (int s = 0; s < 2; ++s) {
    x.ptr[s] = nullptr; x.C[s] = 0; x.aff_a[s] = nullptr; x.aff_b[s] = nullptr;
  }
  for (int s = 0; s < L.nsrc; ++s) {
    const Tens& t = n.t[L.src[s]];
    const void* base = (L.in_act != ACT_NONE) ? t.xa[L.in_act] : t.y;   // act(bn(y)) was materialised by the producer
    x.ptr[s] = (const char*)base + (size_t)sample0 * t.H * t.W * t.C * es;
    x.C[s] = t.C;
  }
  x.act = ACT_NONE;
  x.group_n = group_n;
  (void)group0;
}

static int run_pack(vp_pixrefer* h, Net& n, hipStream_t st) {
  if (n.descs.empty()) return VP_OK;
  VP_HIP_CHECK(launch_pack_weights(n.d_descs, (int)n.descs.size(), n.params, n.packed, h->bf16, st));
  return VP_OK;
}

// scratch set of a stream: 0 = the caller's stream, 1 = side stream, 2 = branch stream
static char* scratch_of(vp_pixrefer* h, int ss) { return ss == 3 ? h->scratch4 : ss == 2 ? h->scratch3 : ss == 1 ? h->scratch2 : h->scratch; }
static double* bnp_of(vp_pixrefer* h, int ss) { return ss == 3 ? h->bn_partial4 : ss == 2 ? h->bn_partial3 : ss == 1 ? h->bn_partial2 : h->bn_partial; }

// fused_chunks > 0: the conv epilogue already wrote that many partial chunks per group; only the finalize pass runs
static int run_bn_stats(vp_pixrefer* h, Net& n, Layer& L, int fused_chunks, hipStream_t st, int ss = 0) {
  Tens& t = n.t[L.out];
  BnArgs b;
  memset(&b, 0, sizeof(b));
  b.y = t.y; b.C = t.C; b.G = n.groups; b.Pg = (n.batch / n.groups) * t.H * t.W;
  b.nchunk = bn_nchunk(b.Pg, b.C, b.G, h->bf16);
  b.partial = bnp_of(h, ss);
  b.gamma = n.params + L.gamma_off; b.beta = n.params + L.beta_off;
  b.aff_a = t.bn.a; b.aff_b = t.bn.b; b.mu = t.bn.mu; b.rstd = t.bn.rstd;
  b.eps = 1e-5f;   // pixrefer.py:100
  if (fused_chunks > 0) { b.nchunk = fused_chunks; VP_HIP_CHECK(launch_bn_finalize(b, st)); return VP_OK; }
  if (bn_small(b)) {   // also materialises the activations: tell the caller to skip act_apply
    VP_HIP_CHECK(launch_bn_small_fwd(b, t.xa[ACT_LRELU], t.xa[ACT_RELU], h->bf16, st));
    return 1;
  }
  VP_HIP_CHECK(launch_bn_stats(b, h->bf16, st));
  return VP_OK;
}

// the patch kernel's 16-pixel-wide tiles can write the 2x2 max pool of their output from the epilogue (even image sizes)
static bool plan_can_pool(const IgemmPlan& p) {
  if (p.a.patch != 1 || (p.a.Hg & 1) || (p.a.Wg & 1)) return false;
  int bc, bp;
  igemm_tile(p.cfg, &bc, &bp);
  int th, tw;
  patch_tile_hw(bp, &th, &tw);
  return tw == 16 && th % 2 == 0;
}

// Rows per batch-norm group of the partial table a launch of plan `p` over `batch` samples in `groups` groups writes from its staged
// epilogue (forward statistics, backward sums): one row per pixel tile and parity class.  0: that launch has no such epilogue (K split,
// few-pixel / register-resident-weights kernels) or its pixel tiles straddle groups.
static int epi_stat_chunks(const IgemmPlan& p, int batch, int groups) {
  const IgemmArgs& a = p.a;
  if (a.splitk != 1 || a.patch == 3 || a.patch == 4 || batch % groups) return 0;
  int bc, bp;
  igemm_tile(p.cfg, &bc, &bp);
  const int pg = (batch / groups) * a.Hg * a.Wg;
  if (!(a.patch || groups == 1 || pg % bp == 0)) return 0;
  int pth = 16, ptw = 16;
  patch_tile_hw(bp, &pth, &ptw);
  const int tpg = a.patch ? (batch / groups) * ((a.Hg + pth - 1) / pth) * ((a.Wg + ptw - 1) / ptw) : (pg + bp - 1) / bp;
  return a.nclass * tpg;
}

// forward of one half of the batch of a plain conv layer (VGG: no batch-norm, activation in the epilogue): half 0 / 1
static int run_layer_fwd_half(vp_pixrefer* h, Net& n, Layer& L, int half, hipStream_t st, void* pool_out = nullptr, bool pool_only = false) {
  IgemmArgs a = L.fwd_half.a;
  a.pool_out = pool_out;
  a.pool_only = pool_out && pool_only ? 1 : 0;
  const int nb = a.N;
  fill_src(n, L, a.x, nb, half * nb, 0, h->es);
  a.Wp = n.packed + L.pk_fwd_half * h->es;
  Tens& to = n.t[L.out];
  a.Y = (char*)to.y + (size_t)half * nb * to.H * to.W * to.C * h->es; a.ldY = to.C; a.y_f32 = 0;
  a.bias = n.params + L.b_off;
  a.out_act = L.out_act;
  a.partial = nullptr;        // (no K split on this path)
  a.zeros = h->zeros;
  profile_tag((L.scope + (half ? ":fwd1" : ":fwd0")).c_str());
  VP_HIP_CHECK(launch_igemm(a, h->bf16, L.fwd_half.cfg, st));
  return VP_OK;
}

// First layers (8-channel image inputs, no batch-norm: encoder_1, encoder_fg_1, discriminator layer_1) on the bf16 path: the direct kernel's
// epilogue writes the activations the consumers read (what act_apply would materialise in a pass of its own over the 67 .. 201 MB output).
// Nothing in a step reads the RAW output of such a layer then (consumers read the activations, the chain rule needs their sign only), so
// it is not stored unless vp_pixrefer_set_option(h, "store_first_raw", 1) asks for it (vp_pixrefer_tensor refuses the tensor otherwise).
static bool first_layer_acts_fused(const vp_pixrefer* h, const Net& n, const Layer& L) {
  if (!h->bf16 || L.has_bn || L.tapgemm || L.out_act != ACT_NONE) return false;
  const Tens& to = n.t[L.out];
  if (!to.need_act[ACT_LRELU] || to.is_f32) return false;      // (the kernel's output combinations: lrelu copy [+ relu copy] [+ raw])
  IgemmArgs a = L.fwd.a;
  fill_src(n, L, a.x, n.batch / n.groups, 0, 0, h->es);
  a.ldY = to.C; a.y_f32 = 0; a.out_act = L.out_act;
  return conv_cin8_eligible(a, 1);
}

// forward of one conv layer (+ its batch statistics)
static int run_layer_fwd(vp_pixrefer* h, Net& n, Layer& L, hipStream_t st, void* pool_out = nullptr, int ss = 0) {
  IgemmArgs a = L.fwd.a;
  a.pool_out = pool_out;
  fill_src(n, L, a.x, n.batch / n.groups, 0, 0, h->es);
  a.Wp = n.packed + L.pk_fwd * h->es;
  Tens& to = n.t[L.out];
  a.Y = to.y; a.ldY = to.C; a.y_f32 = 0;
  if (to.is_f32) { a.y_f32 = 1; a.ldY = L.g.Cout; }
  a.bias = L.has_bn ? nullptr : n.params + L.b_off;   // a bias in front of batch-norm cancels exactly
  a.out_act = L.out_act;
  a.partial = (float*)scratch_of(h, ss);
  a.zeros = h->zeros;
  if (L.tapgemm) { a.Y = L.tap_S; a.y_f32 = 1; a.ldY = 16; a.bias = nullptr; }
  if (a.patch == 3) {
    // few-pixel layer: convolution, K-split combine, batch statistics, affine and the consumers' activations in ONE launch
    SmallPArgs sp;
    memset(&sp, 0, sizeof(sp));
    a.sp_cnt = L.sp_cnt_fwd;
    sp.g = a;
    for (int c = 0; c < 4; ++c) sp.tap_mask[c] = a.sp_mask[c];
    sp.slab = a.partial; sp.cnt = a.sp_cnt; sp.part = bnp_of(h, ss);
    sp.mode = L.has_bn ? SP_FWD_BN : SP_PLAIN;
    sp.hi = to.hi ? 1 : 0;               // float32 raw output + statistics of the unrounded values (Tens::hi)
    if (L.has_bn) {
      sp.gamma = n.params + L.gamma_off; sp.beta = n.params + L.beta_off;
      sp.aff_a = to.bn.a; sp.aff_b = to.bn.b; sp.mu = to.bn.mu; sp.rstd = to.bn.rstd;
      sp.out_lrelu = to.need_act[ACT_LRELU] ? to.xa[ACT_LRELU] : nullptr;
      sp.out_relu = to.need_act[ACT_RELU] ? to.xa[ACT_RELU] : nullptr;
      sp.eps = 1e-5f;   // pixrefer.py:100
    }
    profile_tag((L.scope + ":fwd").c_str());
    VP_HIP_CHECK(launch_smallp_fused(sp, h->bf16, st));
    if (!L.has_bn && (to.need_act[ACT_LRELU] || to.need_act[ACT_RELU]))
      VP_HIP_CHECK(launch_act_apply(to.y, nullptr, nullptr, to.C, (n.batch / n.groups) * to.H * to.W, (size_t)to.N * to.H * to.W,
                                    to.xa[ACT_LRELU], to.xa[ACT_RELU], h->bf16, st));
    return VP_OK;
  }
  // batch statistics from the conv epilogue (no re-read of the output) when every pixel tile lies inside one BN group
  bool fused_stats = false;
  int stat_chunks = 0;
  if (L.has_bn && n.groups == 1 && h->bf16 && !L.tapgemm && conv_dc256_eligible(a, 1)) {
    // conv_dc64.hip conv_dc256_kernel: one partial row per block and column parity
    stat_chunks = 2 * conv_dc256_grid(a);
    a.bn_part = bnp_of(h, ss); a.bn_tpg = 1 << 30; a.bn_nchunk = stat_chunks;
    fused_stats = true;
  } else if (L.has_bn && a.patch == 4) {
    // conv_s2c64.hip: one partial row per block and group
    const int grid = conv_s2c64_grid(a);
    if ((size_t)n.groups * grid * 2 * to.C <= (size_t)1024 * 2 * 512 && n.groups <= 32) {
      stat_chunks = grid;
      a.bn_part = bnp_of(h, ss); a.bn_tpg = (n.batch / n.groups) * conv_s2c64_tiles_per_image(a); a.bn_nchunk = grid;
      fused_stats = true;
    }
  } else if (L.has_bn && a.splitk == 1 && L.g.Cout % 8 == 0 && a.ldY % 8 == 0 && !a.y_f32) {
    int bc, bp;
    igemm_tile(L.fwd.cfg, &bc, &bp);
    const int pg = (n.batch / n.groups) * a.Hg * a.Wg;            // pixels of one group, per class
    if (a.patch || n.groups == 1 || pg % bp == 0) {
      // patch kernel: 2-D tiles of one image each (16 x 16 or 16 x 32 pixels), never across images or groups
      int pth = 16, ptw = 16;
      patch_tile_hw(bp, &pth, &ptw);
      const int tpg = a.patch ? (n.batch / n.groups) * ((a.Hg + pth - 1) / pth) * ((a.Wg + ptw - 1) / ptw) : (pg + bp - 1) / bp;
      stat_chunks = a.nclass * tpg;
      if ((size_t)n.groups * stat_chunks * 2 * to.C <= (size_t)1024 * 2 * 512) {
        a.bn_part = bnp_of(h, ss); a.bn_tpg = tpg; a.bn_nchunk = stat_chunks;
        fused_stats = true;
      }
    }
  }
  // first layers (8-channel image inputs, no batch-norm: encoder_1, encoder_fg_1, discriminator layer_1): the direct kernel's epilogue
  // writes the consumers' activations itself - no act_apply pass over the (67 .. 201 MB) output
  bool acts_fused = false;
  if (first_layer_acts_fused(h, n, L)) {
    a.xa_lrelu = to.need_act[ACT_LRELU] ? to.xa[ACT_LRELU] : nullptr;
    a.xa_relu = to.need_act[ACT_RELU] ? to.xa[ACT_RELU] : nullptr;
    if (!h->store_first_raw) a.Y = nullptr;
    acts_fused = true;
  }
  profile_tag((L.scope + ":fwd").c_str());
  VP_HIP_CHECK(launch_igemm(a, h->bf16, L.fwd.cfg, st));
  if (L.tapgemm) {
    TapArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.S = L.tap_S; ta.bias = n.params + L.b_off; ta.y = (float*)to.y;
    ta.N = L.g.N; ta.Hin = L.g.Hin; ta.Win = L.g.Win; ta.Hout = L.g.Hout; ta.Wout = L.g.Wout; ta.ks = L.g.ks; ta.pad = L.g.pad;
    VP_HIP_CHECK(launch_tap_gather(ta, st));
  }
  bool acts_done = acts_fused;
  if (L.has_bn) {
    const int rc = run_bn_stats(h, n, L, fused_stats ? stat_chunks : 0, st, ss);
    if (rc < 0) return rc;
    acts_done = rc == 1;
  }
  if (!acts_done && (to.need_act[ACT_LRELU] || to.need_act[ACT_RELU])) {
    VP_HIP_CHECK(launch_act_apply(to.y, L.has_bn ? to.bn.a : nullptr, L.has_bn ? to.bn.b : nullptr, to.C,
                                  (n.batch / n.groups) * to.H * to.W, (size_t)to.N * to.H * to.W,
                                  to.xa[ACT_LRELU], to.xa[ACT_RELU], h->bf16, st));
  }
  return VP_OK;
}

// backward of one layer given dL/dy in `dy` (dtype T, channel stride g.CoutT):
//   weight/bias/BN gradients (when grads != null) and dz of every source tensor.
// sample0/nb/group0/ng select a sub-batch (discriminator G-loss pass: the fake group only).
// ss: scratch set of the stream the call runs on; gpass: generator-loss pass through the discriminator (dz2 buffers)
static int run_layer_bwd(vp_pixrefer* h, Net& n, Layer& L, const void* dy, bool want_dw, bool alt,
                         int sample0, int nb, int group0, hipStream_t st, int ss = 0, bool gpass = false, int parts = 7) {
  // parts: bit 0 = weight / bias gradients, bit 1 = data gradient of source 0, bit 2 = data gradient of source 1 (so a host can put
  // them on different streams: only source 0's gradient - the decoder chain - is on the critical path of the generator backward)
  const int es = h->es;
  if (!(parts & 1)) want_dw = false;
  char* scratch = scratch_of(h, ss);
  double* bn_partial = bnp_of(h, ss);
  const int group_n = n.batch / n.groups;
  if (want_dw) {
    WgradArgs w = L.wg.a;
    PixSrc xs;
    fill_src(n, L, xs, group_n, sample0, group0, es);
    PixSrc ds;
    set_single_src(ds, dy, L.g.CoutT, nullptr, nullptr, ACT_NONE, 0);
    if (L.g.kind == 0) { w.g = xs; w.d = ds; } else { w.g = ds; w.d = xs; }
    bool own_wgrad = false;
    if (L.tapgemm && h->bf16 && cout1_knob() > 0 && L.nsrc == 1 && !n.t[L.src[0]].is_input && !n.t[L.src[0]].hi && n.t[L.src[0]].xa[L.in_act] &&
        L.g.ks == 4 && L.g.stride == 1) {
      // the one-output-channel layer's weight gradient on its own kernel (conv_cout1.hip): no tap spreading, no transposes
      Tens& tx = n.t[L.src[0]];
      Cout1Args c;
      memset(&c, 0, sizeof(c));
      c.dy = dy; c.ld_dy = L.g.CoutT;
      c.ref = (const char*)tx.xa[L.in_act] + (size_t)sample0 * tx.H * tx.W * tx.C * es;
      c.N = nb; c.H = tx.H; c.W = tx.W; c.C = tx.C; c.Ho = L.g.Hout; c.Wo = L.g.Wout; c.ks = L.g.ks; c.pad = L.g.pad;
      c.slabs = (float*)scratch; c.dW = n.grads + L.w_off;
      const long long px = (long long)nb * tx.H * tx.W;
      const size_t cap = h->scratch_bytes / ((size_t)16 * 512 * sizeof(float));
      c.rows = (int)(px / 96 < wgrad1_rows_knob() ? px / 96 : wgrad1_rows_knob());            // >= 24 pixels per wave
      if ((size_t)c.rows > cap) c.rows = (int)cap;
      if (c.rows >= 1 && L.g.Cin == 512 && conv_cout1_wgrad_eligible(c)) {
        profile_tag((L.scope + ":wgrad").c_str());
        VP_HIP_CHECK(launch_cout1_wgrad_prof(c, st));
        own_wgrad = true;
      }
    }
    if (L.tapgemm && !own_wgrad) {
      TapArgs ta;
      memset(&ta, 0, sizeof(ta));
      ta.dy = dy; ta.dyS = L.tap_dyS; ta.ld_dy = L.g.CoutT;
      ta.N = nb; ta.Hin = L.g.Hin; ta.Win = L.g.Win; ta.Hout = L.g.Hout; ta.Wout = L.g.Wout; ta.ks = L.g.ks; ta.pad = L.g.pad;
      VP_HIP_CHECK(launch_tap_spread(ta, h->bf16, st));
      set_single_src(w.g, L.tap_dyS, 16, nullptr, nullptr, ACT_NONE, 0);
      w.d = xs;
    }
    w.partial = (float*)scratch;
    w.dW = n.grads + L.w_off;
    w.accumulate = 0;
    w.zeros = h->zeros;
    if (!own_wgrad) {
      profile_tag((L.scope + ":wgrad").c_str());
      VP_HIP_CHECK(launch_wgrad(w, h->bf16, L.wg.cfg, st));
    }
    float* db = n.grads + L.b_off;
    if (L.has_bn) {
      // analytically zero: written by this layer's batch-norm backward (run_bn_bwd, dbias_zero) - no separate memset launch
    } else {
      BnArgs b;
      memset(&b, 0, sizeof(b));
      b.y = dy; b.C = L.g.CoutT; b.G = 1; b.Pg = nb * L.g.Hout * L.g.Wout;
      b.nchunk = bn_nchunk(b.Pg, b.C, 1, h->bf16);
      b.partial = bn_partial;
      Tens& to = n.t[L.out];
      if (to.cs_chunks > 0 && dy == to.dz) {     // the launch that completed dy left its column sums behind (conv_dc64.hip CSUM): finalize only
        b.partial = to.cs_part; b.nchunk = to.cs_chunks;
        to.cs_chunks = 0;
        VP_HIP_CHECK(launch_colsum_tail(b, L.g.Cout, db, 0, st));
      } else {
        VP_HIP_CHECK(launch_colsum(b, L.g.Cout, db, 0, h->bf16, st));
      }
    }
  }
  // Backward sums of a batch-normalised tensor from the epilogue of the launch that completes its gradient (IgemmArgs::bst_*, staged_epilogue
  // STATS == 2): the tensor's batch-norm backward then starts at its finalize - bn_reduce_kernel<T, 1> never re-reads y and dz.  `which`:
  // output 0 / 1 of a two-output launch.  Not for tensors the one-launch batch-norm backward handles (<= 2048 pixels per group), not for
  // launches on the few-pixel / register-resident-weights kernels (no such epilogue), not when a pixel tile would straddle two groups.
  auto try_bst = [&](IgemmArgs& a, const IgemmPlan& p, Tens& ts, int which, bool last) {
    if (!h->bst_on || !last || !ts.has_bn || ts.hi || ts.producer < 0 || ts.is_input) return;
    const int groups = (gpass || alt) ? 1 : n.groups;
    if ((nb / groups) * ts.H * ts.W <= 2048) return;                       // bn_small
    if (a.y_f32 || a.ldY % 8 || a.Cout % 8 || a.ldY != ts.C) return;
    // bwd_sums_in_epilogue = 1 (default): only where the A/B said it pays (profiles/r06_bwd_sums_per_layer.txt) - single-output launches of the
    // 2x2-tap patch kernel and the plain implicit GEMM.  The sixteen accumulators do not fit beside the staged tile's registers in the
    // 128-register kernels: on the two-output launches (decoder_1: 0.090 -> 0.178 ms), the 4x4 patch kernel (layer_4: +0.063 ms) and the
    // 16-deep tap product of the generator-loss pass (layer_5: +0.010 ms against a 0.010 ms reduce) the spills cost as much or more than the reduce pass they replace.  2: every launch that can.
    if (h->bst_on < 2 && (a.split_c || a.patch == 1 || (L.tapgemm && gpass))) return;
    if (a.patch == 1 && !patch4_eligible(a, h->bf16)) return;
    if (a.patch == 2 && (a.x.C[1] > 0 || (h->bf16 && conv_dc64_eligible(a, 1)))) return;
    const int chunks = epi_stat_chunks(p, nb, groups);
    if (chunks <= 0 || chunks > (gpass ? ts.pbg_cap : ts.pb_cap)) return;
    double* part = gpass ? ts.bn.pbg : ts.bn.pb;
    const void* y = (const char*)ts.y + (size_t)sample0 * ts.H * ts.W * ts.C * es;
    if (which == 0) { a.bn_part = part; a.bst_y = y; }
    else { a.bn_part2 = part; a.bst_y2 = y; }
    a.bn_tpg = chunks / a.nclass; a.bn_nchunk = chunks;
    (gpass ? ts.bst_chunks_g : ts.bst_chunks) = chunks;
    h->bst_count++;
  };
  // few-pixel pairs: only the usual case - this launch is the LAST contribution to the first tensor (its batch-norm backward runs in
  // the launch) and NOT the last one to the second (the skip tensor's own encoder consumer comes later)
  const bool pair_sp = L.has_pair && L.bwd_pair.a.patch == 3;
  const bool pair_sp_ok = pair_sp && n.t[L.src[0]].has_bn && n.t[L.src[0]].dz_writes + 1 == n.t[L.src[0]].n_bwd_consumers &&
                          n.t[L.src[1]].dz_writes + 1 < n.t[L.src[1]].n_bwd_consumers;
  if (L.has_pair && (parts & 6) == 6 && !alt && !gpass && (!pair_sp || pair_sp_ok)) {
    // both sources' data gradients in one two-output launch (Layer::bwd_pair)
    Tens &t0 = n.t[L.src[0]], &t1 = n.t[L.src[1]];
    IgemmArgs a = L.bwd_pair.a;
    if (pair_sp) {
      set_single_src(a.x, dy, L.g.CoutT, nullptr, nullptr, ACT_NONE, 0);
      a.Wp = n.packed + L.pk_bwd_pair * es;
      a.partial = (float*)scratch;
      a.zeros = h->zeros;
      a.ref = (const char*)t0.xa[L.in_act] + (size_t)sample0 * t0.H * t0.W * t0.C * es;
      a.ref2 = (const char*)t1.xa[L.in_act] + (size_t)sample0 * t1.H * t1.W * t1.C * es;
      a.ref_act = L.in_act;
      a.ref_group_n = group_n;
      a.Y = t0.hi ? t0.dz32 : t0.dz; a.accumulate = t0.dz_written ? 1 : 0;
      a.Y2 = t1.hi ? t1.dz32 : t1.dz; a.accumulate2 = t1.dz_written ? 1 : 0; a.y2_f32 = t1.hi ? 1 : 0;
      t0.dz_written = t1.dz_written = true;
      t0.dz_writes++; t1.dz_writes++;
      SmallPArgs sp;
      memset(&sp, 0, sizeof(sp));
      a.sp_cnt = L.sp_cnt_pair;
      sp.g = a;
      for (int c = 0; c < 4; ++c) sp.tap_mask[c] = a.sp_mask[c];
      sp.slab = a.partial; sp.cnt = a.sp_cnt; sp.part = bn_partial;
      const Layer& Lp = n.l[t0.producer];
      sp.mode = SP_BWD_BN;
      if (t0.hi) { sp.hi = 1; sp.dy_out = t0.dz; }
      sp.bn_y = t0.y; sp.bn_mu = t0.bn.mu; sp.bn_rstd = t0.bn.rstd; sp.bn_gamma = n.params + Lp.gamma_off;
      sp.c1 = t0.bn.c1; sp.c2 = t0.bn.c2;
      sp.dgamma = n.grads + Lp.gamma_off; sp.dbeta = n.grads + Lp.beta_off; sp.dbias_zero = n.grads + Lp.b_off;
      t0.bn_bwd_done = true;
      profile_tag((L.scope + ":bwd").c_str());
      VP_HIP_CHECK(launch_smallp_fused(sp, h->bf16, st));
      return VP_OK;
    }
    set_single_src(a.x, dy, L.g.CoutT, nullptr, nullptr, ACT_NONE, 0);
    a.Wp = n.packed + L.pk_bwd_pair * es;
    a.partial = (float*)scratch;
    a.Y = t0.dz; a.accumulate = t0.dz_written ? 1 : 0;
    a.Y2 = t1.dz; a.accumulate2 = t1.dz_written ? 1 : 0;
    t0.dz_written = t1.dz_written = true;
    t0.dz_writes++; t1.dz_writes++;
    a.ref = (const char*)t0.xa[L.in_act] + (size_t)sample0 * t0.H * t0.W * t0.C * es;
    a.ref2 = (const char*)t1.xa[L.in_act] + (size_t)sample0 * t1.H * t1.W * t1.C * es;
    a.ref_act = L.in_act;
    a.ref_group_n = group_n;
    a.zeros = h->zeros;
    try_bst(a, L.bwd_pair, t0, 0, t0.dz_writes == t0.n_bwd_consumers);
    try_bst(a, L.bwd_pair, t1, 1, t1.dz_writes == t1.n_bwd_consumers);
    profile_tag((L.scope + ":bwd").c_str());
    VP_HIP_CHECK(launch_igemm(a, h->bf16, L.bwd_pair.cfg, st));
    return VP_OK;
  }
  for (int s = 0; s < L.nsrc; ++s) {
    if (!(parts & (2 << s))) continue;
    if (!L.need_bwd[s]) continue;
    Tens& ts = n.t[L.src[s]];
    IgemmArgs a = alt ? L.bwd_alt[s].a : L.bwd[s].a;
    set_single_src(a.x, dy, L.g.CoutT, nullptr, nullptr, ACT_NONE, 0);
    a.Wp = n.packed + (alt ? L.pk_bwd_alt[s] : L.pk_bwd[s]) * es;
    a.partial = (float*)scratch;
    if (ts.is_input) {
      a.Y = (n.groups == 3) ? h->d_din : h->d_vin;
      a.accumulate = 0;
    } else {
      a.Y = gpass ? ts.dz2 : ts.dz;
      if (ts.hi) {
        if (gpass || alt || a.patch != 3) { set_err("%s: float32 few-pixel tensor %s reached by a launch that is not the few-pixel kernel", L.scope.c_str(), ts.name.c_str()); return VP_ERR_STATE; }
        a.Y = ts.dz32; a.y_f32 = 1;        // the gradient accumulates in float32 (Tens::hi)
      }
      bool& written = gpass ? ts.dz2_written : ts.dz_written;
      a.accumulate = written ? 1 : 0;
      written = true;
      if (!gpass) {
        ts.dz_writes++;
      }
      // chain rule through the consumer's activation and (for BN tensors) up to the normalised value
      // lrelu'/relu' only depend on the sign of the pre-activation == the sign of the materialised x~
      a.ref = (const char*)ts.xa[L.in_act] + (size_t)sample0 * ts.H * ts.W * ts.C * es;
      a.ref_act = L.in_act;
      a.ref_group_n = group_n;
    }
    a.zeros = h->zeros;
    profile_tag((L.scope + (alt ? ":bwdG" : ":bwd")).c_str());
    if (a.patch == 3 && !alt) {
      // few-pixel layer: K-split combine in the launch; when this is the last gradient contribution to a batch-normalised tensor, the
      // batch-norm backward of that tensor (sums, c1 / c2, dgamma / dbeta, dy in place) runs in the same launch
      SmallPArgs sp;
      memset(&sp, 0, sizeof(sp));
      a.sp_cnt = L.sp_cnt_bwd[s];
      sp.g = a;
      for (int c = 0; c < 4; ++c) sp.tap_mask[c] = a.sp_mask[c];
      sp.slab = a.partial; sp.cnt = a.sp_cnt; sp.part = bn_partial;
      const bool last = !ts.is_input && !gpass && ts.has_bn && ts.producer >= 0 && ts.dz_writes == ts.n_bwd_consumers && n.groups == 1;
      if (ts.hi && !last && ts.dz_writes == ts.n_bwd_consumers) { set_err("%s: float32 tensor %s needs its batch-norm backward in the launch", L.scope.c_str(), ts.name.c_str()); return VP_ERR_STATE; }
      if (last) {
        const Layer& Lp = n.l[ts.producer];
        sp.mode = SP_BWD_BN;
        if (ts.hi) { sp.hi = 1; sp.dy_out = ts.dz; sp.g.y_f32 = 0; }      // dz32 / y in float32, dL/dy out as T into dz
        sp.bn_y = ts.y; sp.bn_mu = ts.bn.mu; sp.bn_rstd = ts.bn.rstd; sp.bn_gamma = n.params + Lp.gamma_off;
        sp.c1 = ts.bn.c1; sp.c2 = ts.bn.c2;
        sp.dgamma = n.grads + Lp.gamma_off; sp.dbeta = n.grads + Lp.beta_off; sp.dbias_zero = n.grads + Lp.b_off;
        ts.bn_bwd_done = true;
      } else {
        sp.mode = SP_PLAIN;
      }
      VP_HIP_CHECK(launch_smallp_fused(sp, h->bf16, st));
      continue;
    }
    // the PatchGAN's last layer (4x4, stride 1, ONE output channel over 512): its own kernel (conv_cout1.hip), which also leaves the two
    // sums of the producer's batch-norm backward behind - when that backward is not the one-launch form and the tensor has a table
    if (L.tapgemm && h->bf16 && cout1_knob() > 0 && !ts.is_input && !ts.hi && !a.accumulate && L.in_act == ACT_LRELU && L.g.ks == 4 && L.g.stride == 1) {
      const int groups = (gpass || alt) ? 1 : n.groups;
      Cout1Args c;
      memset(&c, 0, sizeof(c));
      c.dy = dy; c.ld_dy = L.g.CoutT;
      c.w = n.params + L.w_off;
      c.ref = a.ref; c.ref_act = a.ref_act;
      c.dx = a.Y;
      c.N = nb; c.H = ts.H; c.W = ts.W; c.C = ts.C; c.Ho = L.g.Hout; c.Wo = L.g.Wout; c.ks = L.g.ks; c.pad = L.g.pad;
      c.groups = groups;
      const int tiles = nb % groups ? 0 : ((nb / groups) * ts.H * ts.W + 15) >> 4;
      const int cap = gpass ? ts.pbg_cap : ts.pb_cap;
      const bool sums = h->bst_on && ts.has_bn && ts.producer >= 0 && (nb / groups) * ts.H * ts.W > 2048 && cap > 0;
      // at least eight tiles per (persistent) block - its prologue loads the block's 32 KB of weights -, and not for launches below 16384
      // pixels (8 frames, generator-loss pass: 0.025 ms against 0.020 on the generic kernel)
      c.rows = tiles / 8 < cout1_knob() ? tiles / 8 : cout1_knob();
      if (c.rows > 512 / groups) c.rows = 512 / groups;          // 512 resident blocks in all (two per compute unit): three groups x 170 rows 0.059 ms, x 256 0.065
      if (sums && c.rows > cap) c.rows = cap;
      if ((long long)nb * ts.H * ts.W < 16384 || c.rows < 1) c.rows = 0;
      if (sums) { c.part = gpass ? ts.bn.pbg : ts.bn.pb; c.y = (const char*)ts.y + (size_t)sample0 * ts.H * ts.W * ts.C * es; }
      if (tiles > 0 && c.rows > 0 && conv_cout1_bwd_eligible(c)) {
        if (sums) { (gpass ? ts.bst_chunks_g : ts.bst_chunks) = c.rows; h->bst_count++; }
        VP_HIP_CHECK(launch_cout1_bwd_prof(c, st));
        continue;
      }
    }
    // the bias gradient of a producer without batch-norm (layer_1, encoder_1, encoder_fg_1) from this launch's epilogue when it completes
    // the tensor's gradient on conv_dc64.hip (the only kernel with that epilogue)
    if (h->bst_on && h->bf16 && !ts.is_input && !ts.has_bn && ts.cs_part && !gpass && !alt && ts.producer >= 0 && !n.l[ts.producer].tapgemm &&
        (n.groups != 1 || ts.dz_writes == ts.n_bwd_consumers) && dc64_knob() && conv_dc64_eligible(a, 1)) {
      a.colsum_part = ts.cs_part;
      ts.cs_chunks = conv_dc64_grid(a);
    }
    // (the discriminator's tensors have one consumer each and no write counter: every gradient launch completes its tensor)
    if (!ts.is_input) try_bst(a, alt ? L.bwd_alt[s] : L.bwd[s], ts, 0, (gpass || n.groups != 1) ? true : ts.dz_writes == ts.n_bwd_consumers);
    VP_HIP_CHECK(launch_igemm(a, h->bf16, alt ? L.bwd_alt[s].cfg : L.bwd[s].cfg, st));
  }
  return VP_OK;
}

// dz -> dy through the training-mode batch norm of tensor t (in place), + dgamma/dbeta
static int run_bn_bwd(vp_pixrefer* h, Net& n, Layer& L, bool want_dw, int sample0, int nb, int group0, int ng, hipStream_t st,
                      int ss = 0, bool gpass = false) {
  Tens& t = n.t[L.out];
  if (!gpass && t.bn_bwd_done) { t.bn_bwd_done = false; return VP_OK; }   // done inside the last gradient contribution's launch (conv_smallp.hip)
  if (t.hi) { set_err("%s: the batch-norm backward of a float32 few-pixel tensor runs inside the few-pixel launch only", L.scope.c_str()); return VP_ERR_STATE; }
  BnArgs b;
  memset(&b, 0, sizeof(b));
  b.y = (const char*)t.y + (size_t)sample0 * t.H * t.W * t.C * h->es;
  b.dy = gpass ? t.dz2 : t.dz; b.dz = b.dy;
  b.C = t.C; b.G = ng; b.Pg = (nb / ng) * t.H * t.W;
  b.nchunk = bn_nchunk(b.Pg, b.C, b.G, h->bf16);
  b.partial = bnp_of(h, ss);
  b.gamma = n.params + L.gamma_off;
  b.mu = t.bn.mu + (size_t)group0 * t.C; b.rstd = t.bn.rstd + (size_t)group0 * t.C;
  b.c1 = gpass ? t.bn.c1g : t.bn.c1; b.c2 = gpass ? t.bn.c2g : t.bn.c2;
  if (want_dw) { b.dgamma = n.grads + L.gamma_off; b.dbeta = n.grads + L.beta_off; b.dbias_zero = n.grads + L.b_off; }
  int& fused = gpass ? t.bst_chunks_g : t.bst_chunks;
  if (fused > 0) {      // the launch that completed the gradient left the partial sums in the tensor's own table: finalize + apply only
    b.partial = gpass ? t.bn.pbg : t.bn.pb;
    b.nchunk = fused;
    b.raw = 1;
    fused = 0;
    VP_HIP_CHECK(launch_bn_bwd_tail(b, h->bf16, st));
    return VP_OK;
  }
  if (bn_small(b)) { VP_HIP_CHECK(launch_bn_small_bwd(b, h->bf16, st)); return VP_OK; }
  VP_HIP_CHECK(launch_bn_bwd(b, h->bf16, st));
  return VP_OK;
}

}  // namespace vp

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

static bool g_phase_marks = false;   // vp_tune("phase_marks", 0 / 1 / 2): 2 = also one mark in front of every generator layer
static bool g_phase_detail = false;
void vp_phase_marks_enable(int on) { g_phase_marks = on != 0; g_phase_detail = on > 1; }
static void phase_mark(vp_pixrefer* h, hipStream_t st, int idx) {
  if (!g_phase_marks || idx >= 64) return;
  if (!h->marks_made) { for (int i = 0; i < 64; ++i) (void)hipEventCreate(&h->mark[i]); h->marks_made = true; }
  (void)hipEventRecord(h->mark[idx], st);
  if (idx + 1 > h->nmark) h->nmark = idx + 1;
}
// ms between consecutive marks of the last step: [0] generator forward (to the composite), [1] discriminator / VGG forward + losses,
// [2] generator-loss pass through D and VGG + composite backward, [3] generator backward stage 0, [4] stage 1, [5] stage 2 + joins
// ms from mark `from` to mark `to` of the last step (-1: one of them was not recorded)
float vp_pixrefer_mark_ms(vp_pixrefer_t* h, int from, int to) {
  if (!h || !h->marks_made || from < 0 || to < 0 || from >= 64 || to >= 64) return -1.f;
  float t = -1.f;
  if (hipEventSynchronize(h->mark[to]) != hipSuccess || hipEventElapsedTime(&t, h->mark[from], h->mark[to]) != hipSuccess) { (void)hipGetLastError(); return -1.f; }
  return t;
}
int vp_pixrefer_phase_ms(vp_pixrefer_t* h, float* ms, int cap) {
  if (!h || !ms) return 0;
  int n = 0;
  for (int i = 0; i + 1 < (h->nmark < 8 ? h->nmark : 8) && n < cap; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(h->mark[i + 1]) != hipSuccess || hipEventElapsedTime(&t, h->mark[i], h->mark[i + 1]) != hipSuccess) {
      (void)hipGetLastError();        // (a mark that was never recorded: not an error of the step)
      break;
    }
    ms[n++] = t;
  }
  return n;
}
int vp_version(void) { return 100; }
const char* vp_last_error(void) { return g_err; }

static Net* manifest_net(const vp_pixrefer_desc* d, int which, vp_pixrefer* tmp) {
  if (which == 0) { build_generator(tmp->G, d->batch, d->height, d->ngf); return &tmp->G; }
  if (which == 1) { build_discriminator(tmp->D, d->batch, d->height, d->ndf); return &tmp->D; }
  if (which == 2) { build_vgg(tmp->V, d->batch, d->height); return &tmp->V; }
  return nullptr;
}

size_t vp_pixrefer_param_count(const vp_pixrefer_desc* d, int which) {
  if (!valid_desc(d)) return 0;
  vp_pixrefer tmp{};
  Net* n = manifest_net(d, which, &tmp);
  return n ? n->nparams : 0;
}

int vp_pixrefer_param_info(const vp_pixrefer_desc* d, int which, int index, char* name, int name_cap,
                           size_t* offset, int* ndim, int64_t shape[4]) {
  if (!valid_desc(d)) return VP_ERR_ARG;
  vp_pixrefer tmp{};
  Net* n = manifest_net(d, which, &tmp);
  if (!n || index < 0 || index >= (int)n->manifest.size()) return VP_ERR_ARG;
  const ParamInfo& p = n->manifest[index];
  if (name && name_cap > 0) { strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (offset) *offset = p.off;
  if (ndim) *ndim = p.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  return VP_OK;
}

size_t vp_pixrefer_workspace_bytes(const vp_pixrefer_desc* d) {
  if (!valid_desc(d)) return 0;
  vp_pixrefer* h = new vp_pixrefer{};
  init_handle(h, d);
  const size_t n = carve_all(h, nullptr, 0);
  delete h;
  return n;
}

// Host-only self-check of a plan (no GPU call; the sanitizer build of the host layer runs it on the CPU, tests/test_host_logic.py):
// carves the workspace at a fake base and verifies that every buffer the kernels will be handed is a carved region of sufficient
// size, that regions are disjoint and inside the workspace, that every parameter / packed-weight range lies inside its arena and
// that every plan's split-K slab fits the scratch and its statistics the batch-norm partial buffer.
int vp_pixrefer_validate_plan(const vp_pixrefer_desc* d) {
  if (!valid_desc(d)) return VP_ERR_ARG;
  vp_pixrefer* h = new vp_pixrefer{};
  init_handle(h, d);
  const size_t need = carve_all(h, nullptr, 0);
  std::vector<std::pair<size_t, size_t>> regs;
  char* const base = reinterpret_cast<char*>((uintptr_t)1 << 40);        // never dereferenced
  carve_all(h, base, need, &regs);
  int rc = VP_OK;
  auto fail = [&](const char* fmt, auto... args) { if (rc == VP_OK) { set_err(fmt, args...); rc = VP_ERR_STATE; } };
  size_t prev_end = 0;
  for (size_t i = 0; i < regs.size(); ++i) {
    if (regs[i].first % 256) fail("region %zu at %zu: not 256-byte aligned", i, regs[i].first);
    if (regs[i].first < prev_end) fail("region %zu at %zu overlaps the previous one (ends %zu)", i, regs[i].first, prev_end);
    prev_end = regs[i].first + regs[i].second;
  }
  if (prev_end + 256 > need + 256 || prev_end > need) fail("carved %zu bytes of a %zu-byte workspace", prev_end, need);
  auto region_of = [&](const void* p, size_t bytes, const char* what) {
    if (!p) { fail("%s: null", what); return; }
    const size_t off = (size_t)((const char*)p - base);
    for (const auto& r : regs)
      if (off >= r.first && off + bytes <= r.first + r.second) return;
    fail("%s: [%zu, +%zu) is not inside a carved region", what, off, bytes);
  };
  const int es = h->es;
  for (Net* n : {&h->G, &h->D, &h->V}) {
    if (n->t.empty()) continue;
    for (const Tens& t : n->t) {
      const std::string nm = t.name;
      if (t.name == "decoder_1" || t.name == "layer_5") continue;       // thin f32 outputs live in handle-level buffers (checked below)
      region_of(t.y, t.elems() * (t.hi ? 4 : es), (nm + ".y").c_str());
      for (int k = 1; k < 3; ++k) if (t.need_act[k]) region_of(t.xa[k], t.elems() * es, (nm + ".xa").c_str());
      if (t.has_bn) {
        const size_t gc = (size_t)n->groups * t.C * sizeof(float);
        for (const float* p : {t.bn.a, t.bn.b, t.bn.mu, t.bn.rstd, t.bn.c1, t.bn.c2, t.bn.c1g, t.bn.c2g}) region_of(p, gc, (nm + ".bn").c_str());
        if (d->training && t.pb_cap) region_of(t.bn.pb, (size_t)n->groups * t.pb_cap * 2 * t.C * sizeof(double), (nm + ".bn.pb").c_str());
        if (d->training && t.pbg_cap) region_of(t.bn.pbg, (size_t)t.pbg_cap * 2 * t.C * sizeof(double), (nm + ".bn.pbg").c_str());
        if ((size_t)n->groups * t.C > (size_t)1024 * 512) fail("%s: %d groups x %d channels exceed the batch-norm partial rows", nm.c_str(), n->groups, t.C);
      }
      if (t.cs_part) region_of(t.cs_part, (size_t)512 * 2 * 64 * sizeof(double), (nm + ".cs_part").c_str());
      if (d->training && !t.is_input) {
        const size_t div = n == &h->V ? 2 : 1;
        region_of(t.dz, t.elems() / div * es, (nm + ".dz").c_str());
        if (t.hi) region_of(t.dz32, t.elems() * 4, (nm + ".dz32").c_str());
        if (n == &h->D) region_of(t.dz2, t.elems() / 3 * es, (nm + ".dz2").c_str());
      }
    }
    region_of(n->packed, n->packed_elems * es, "packed weights");
    region_of(n->d_descs, n->descs.size() * sizeof(PackDesc), "pack descriptors");
    for (const PackDesc& p : n->descs) {
      const size_t pe = (size_t)p.nclass * p.Kpad * p.rows_pad;
      if (p.dst_off + pe > n->packed_elems) fail("pack block [%zu, +%zu) outside the %zu-element packed arena", p.dst_off, pe, n->packed_elems);
      if (p.src_off >= n->nparams) fail("pack source %zu outside the %zu-float parameter arena", p.src_off, n->nparams);
    }
    size_t pend = 0;
    for (const ParamInfo& p : n->manifest) {
      size_t cnt = 1;
      for (int i = 0; i < 4; ++i) cnt *= (size_t)p.shape[i];
      if (p.off != pend) fail("parameter %s at %zu: arena not contiguous (expected %zu)", p.name.c_str(), p.off, pend);
      pend = p.off + cnt;
    }
    if (pend != n->nparams) fail("parameter manifest covers %zu of %zu floats", pend, n->nparams);
    for (const Layer& L : n->l) {
      auto plan_ok = [&](const IgemmPlan& p, const char* what) {
        if (p.partial_bytes > h->scratch_bytes) fail("%s %s: split-K slab %zu > scratch %zu", L.scope.c_str(), what, p.partial_bytes, h->scratch_bytes);
        if (p.a.splitk < 1 || p.a.CoutPad < p.a.Cout) fail("%s %s: splitk %d CoutPad %d", L.scope.c_str(), what, p.a.splitk, p.a.CoutPad);
      };
      plan_ok(L.fwd, "fwd");
      if (d->training && L.has_pair) plan_ok(L.bwd_pair, "bwd_pair");
      if (d->training)
        for (int s2 = 0; s2 < L.nsrc; ++s2) if (L.need_bwd[s2]) { plan_ok(L.bwd[s2], "bwd"); if (n == &h->D || n == &h->V) plan_ok(L.bwd_alt[s2], "bwd_alt"); }
      if (d->training && n != &h->V && L.wg.partial_bytes > h->scratch_bytes) fail("%s wgrad: slab %zu > scratch %zu", L.scope.c_str(), L.wg.partial_bytes, h->scratch_bytes);
      if (L.tapgemm) { region_of(L.tap_S, (size_t)L.g.N * L.g.Hin * L.g.Win * 16 * 4, "tap_S"); region_of(L.tap_dyS, (size_t)L.g.N * L.g.Hin * L.g.Win * 16 * es, "tap_dyS"); }
    }
  }
  const size_t px = (size_t)d->batch * d->height * d->height;
  region_of(h->gin, px * 8 * es, "gin"); region_of(h->gfg, px * 8 * es, "gfg");
  region_of(h->y4, px * 4 * 4, "y4"); region_of(h->o4, px * 4 * 4, "o4");
  region_of(h->outputs, px * 3 * 4, "outputs"); region_of(h->outputs_fg, px * 3 * 4, "outputs_fg");
  region_of(h->scratch, h->scratch_bytes, "scratch"); region_of(h->bn_partial, (size_t)1024 * 2 * 512 * 8, "bn_partial");
  if (d->training) {
    const int hd = d->height / 8 - 2;
    const size_t M = (size_t)d->batch * hd * hd;
    region_of(h->din, 3 * px * 8 * es, "din"); region_of(h->vin, 2 * px * 8 * es, "vin");
    region_of(h->logits, 3 * M * 4, "logits"); region_of(h->predict, 2 * M * 4, "predict");
    region_of(h->dl_d, 3 * M * 8 * es, "dl_d"); region_of(h->dl_g, M * 8 * es, "dl_g");
    region_of(h->d_din, px * 8 * es, "d_din"); region_of(h->d_vin, px * 8 * es, "d_vin"); region_of(h->dy4, px * 8 * es, "dy4");
    region_of(h->scratch2, h->scratch_bytes, "scratch2"); region_of(h->scratch3, h->scratch_bytes, "scratch3");
    region_of(h->scratch4, h->scratch_bytes, "scratch4"); region_of(h->bn_partial4, (size_t)1024 * 2 * 512 * 8, "bn_partial4");
    region_of(h->bn_partial2, (size_t)1024 * 2 * 512 * 8, "bn_partial2"); region_of(h->bn_partial3, (size_t)1024 * 2 * 512 * 8, "bn_partial3");
  }
  delete h;
  return rc;
}

// The executor's three extra streams are PROCESS-WIDE (per device), created once and never destroyed: every plan of the process uses the
// same ones.  Which hardware queues the HIP runtime gives a stream depends on what exists when it is created, and streams created behind
// other busy ones (an RCCL communicator's, an earlier plan's) run the step slow - a plan created after dist.init_process_group(device_id)
// measured 7.8 ms instead of 7.2 at 32 frames and 2.85 instead of 2.19 at 4 (scripts/exp_dp_order.py), a second engine beside a live
// first one 3.4 - 4.6 ms instead of 2.15 (scripts/exp_engine_sequence.py).  vp_reserve_streams() creates them NOW: a host calls it first
// thing (before it creates a communicator); otherwise the first training plan does.  Plans that share the streams serialise on them,
// which is what two plans stepping in one process do on the device anyway; fork / join events stay per plan.
struct StreamPool {
  int dev = -1;
  hipStream_t side = nullptr, branch = nullptr, branch2 = nullptr;
};
static StreamPool g_pool;
static std::mutex g_pool_mu;

static int pool_streams(bool want_b2, hipStream_t* side, hipStream_t* branch, hipStream_t* branch2) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  int dev = 0;
  VP_HIP_CHECK(hipGetDevice(&dev));
  if (g_pool.dev >= 0 && g_pool.dev != dev) { set_err("the step executor's streams were created on device %d, this plan is on device %d (one device per process)", g_pool.dev, dev); return VP_ERR_STATE; }
  if (!g_pool.side) {
    // lowest priority: the side stream only fills the CUs the main stream's (longer, critical-path) passes leave idle
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    VP_HIP_CHECK(hipStreamCreateWithPriority(&g_pool.side, hipStreamNonBlocking, prio_lo));
    VP_HIP_CHECK(hipStreamCreateWithFlags(&g_pool.branch, hipStreamNonBlocking));
    g_pool.dev = dev;
  }
  // the fourth stream on first use (the backward pass of a plan that still wants four streams, or vp_reserve_streams): hardware queues
  // are handed out as streams are created, and a host stream created later (an input prefetcher's) must not be pushed onto a shared one
  if (want_b2 && !g_pool.branch2) VP_HIP_CHECK(hipStreamCreateWithFlags(&g_pool.branch2, hipStreamNonBlocking));
  if (side) *side = g_pool.side;
  if (branch) *branch = g_pool.branch;
  if (branch2) *branch2 = g_pool.branch2;
  return VP_OK;
}

int vp_reserve_streams(void) { return pool_streams(true, nullptr, nullptr, nullptr); }

// The process's fourth executor stream for a host with ONE busy stream of its own (an input prefetcher) whose plans keep to three
// (vp_pixrefer_use_streams(h, 3)): a fifth stream, created behind the four, would get a shared hardware queue.
void* vp_host_stream(void) {
  hipStream_t b2 = nullptr;
  if (pool_streams(true, nullptr, nullptr, &b2)) return nullptr;
  return (void*)b2;
}

int vp_pixrefer_create(const vp_pixrefer_desc* d, void* workspace, size_t workspace_bytes,
                       float* params_g, float* params_d, const float* params_vgg,
                       float* grads_g, float* grads_d, void* stream, vp_pixrefer_t** out) {
  if (!valid_desc(d) || !workspace || !params_g || !out) { set_err("vp_pixrefer_create: bad argument"); return VP_ERR_ARG; }
  if (d->training && (!params_d || !params_vgg || !grads_g || !grads_d)) { set_err("training needs D/VGG params and grad arenas"); return VP_ERR_ARG; }
  vp_pixrefer* h = new vp_pixrefer{};
  init_handle(h, d);
  const size_t need = carve_all(h, nullptr, 0);
  if (workspace_bytes < need) { set_err("workspace too small: %zu < %zu", workspace_bytes, need); delete h; return VP_ERR_WORKSPACE; }
  carve_all(h, (char*)workspace, workspace_bytes);
  h->G.params = params_g; h->G.grads = grads_g;
  h->D.params = params_d; h->D.grads = grads_d;
  h->V.params = const_cast<float*>(params_vgg);
  // thin f32 outputs
  for (Tens& t : h->G.t) if (t.name == "decoder_1") { t.y = h->y4; t.is_f32 = true; t.dz = h->dy4; }
  hipStream_t st = (hipStream_t)stream;
  VP_HIP_CHECK(hipMemsetAsync(h->zeros, 0, 256, st));
  if (d->training) for (Tens& t : h->D.t) if (t.name == "layer_5") { t.y = h->logits; t.is_f32 = true; t.dz = h->dl_d; }
  for (Net* n : {&h->G, &h->D, &h->V}) {
    if (n->descs.empty()) continue;
    VP_HIP_CHECK(hipMemcpyAsync(n->d_descs, n->descs.data(), n->descs.size() * sizeof(PackDesc), hipMemcpyHostToDevice, st));
    VP_HIP_CHECK(hipMemsetAsync(n->sp_cnt, 0, (n->sp_cnt_n + 1) * sizeof(unsigned), st));
  }
  VP_HIP_CHECK(hipStreamSynchronize(st));   // descs are host vectors owned by the handle; copy is complete
  h->overlap = false; h->forked = false;
  // the schedule of this plan (vp_pixrefer_desc; vp_pixrefer_set_option changes it later)
  h->bst_on = 1; h->vgg_fork_layer = 3;
  h->ov_on = true; h->store_first_raw = false; h->dfork_point = d->d_backward_fork ? d->d_backward_fork - 1 : 2; h->dsplit_on = d->d_beside_vgg != 1;
  if (d->training && d->streams != 1) {
    // the process-wide executor streams (StreamPool above); the fourth one on first use unless vp_reserve_streams() made it already
    { const int rc = pool_streams(false, &h->side, &h->branch, &h->branch2); if (rc) { delete h; return rc; } }
    VP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    VP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    h->use_b2 = d->streams != 3;
    VP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_b2join, hipEventDisableTiming));
    VP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_bfork, hipEventDisableTiming));
    VP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_bjoin, hipEventDisableTiming));
    VP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_upd_b, hipEventDisableTiming));
    VP_HIP_CHECK(hipEventCreateWithFlags(&h->ev_upd_m, hipEventDisableTiming));
    h->overlap = true;
  }
  *out = h;
  return VP_OK;
}

void vp_pixrefer_destroy(vp_pixrefer_t* h) {
  if (!h) return;
  if (h->marks_made) for (int i = 0; i < 64; ++i) (void)hipEventDestroy(h->mark[i]);
  if (h->overlap) {
    // the streams belong to the process (StreamPool): this plan's work on them is waited for, they live on
    (void)hipStreamSynchronize(h->side);
    (void)hipEventDestroy(h->ev_fork);
    (void)hipEventDestroy(h->ev_join);
    (void)hipStreamSynchronize(h->branch);
    (void)hipEventDestroy(h->ev_bfork);
    (void)hipEventDestroy(h->ev_bjoin);
    (void)hipEventDestroy(h->ev_b2join);
    if (h->branch2) (void)hipStreamSynchronize(h->branch2);
    (void)hipEventDestroy(h->ev_upd_b);
    (void)hipEventDestroy(h->ev_upd_m);
  }
  delete h;
}

int vp_pixrefer_params_changed(vp_pixrefer_t* h) {
  if (!h) return VP_ERR_ARG;
  h->params_dirty = true;
  h->vgg_dirty = true;
  h->upd_mask = 0;
  return VP_OK;
}

int vp_pixrefer_optimizer_stepped(vp_pixrefer_t* h) {
  if (!h) return VP_ERR_ARG;
  h->params_dirty = true;       // generator* / discriminator* moved; vgg_16 is not an optimiser variable
  return VP_OK;
}

static int forward_impl(vp_pixrefer_t* h, const float* inputs, const float* fg_inputs, int fg_c, const float* targets, const float* masks, void* stream);

int vp_pixrefer_forward(vp_pixrefer_t* h, const float* inputs, const float* fg_inputs,
                        const float* targets, const float* masks, void* stream) {
  return forward_impl(h, inputs, fg_inputs, 6, targets, masks, stream);
}

// build_inference_op as infer_bfmvid.py:202-205 feeds it: fg_inputs [N,H,H,3] (the graph reads fg_inputs[..., :3] only, pixrefer.py:281)
int vp_pixrefer_forward_fg3(vp_pixrefer_t* h, const float* inputs, const float* fg_inputs3, const float* targets, void* stream) {
  if (h && h->d.training) { set_err("vp_pixrefer_forward_fg3: inference plans only (the training graph reads fg_inputs[..., 3:], pixrefer.py:297,321)"); return VP_ERR_STATE; }
  return forward_impl(h, inputs, fg_inputs3, 3, targets, nullptr, stream);
}

size_t vp_pixrefer_desc_size(void) { return sizeof(vp_pixrefer_desc); }

static int forward_impl(vp_pixrefer_t* h, const float* inputs, const float* fg_inputs, int fg_c, const float* targets, const float* masks, void* stream) {
  if (!h || !inputs || !fg_inputs || !targets) { set_err("vp_pixrefer_forward: null argument"); return VP_ERR_ARG; }
  const vp_pixrefer_desc& d = h->d;
  if (d.training && !masks) { set_err("vp_pixrefer_forward: masks required when training"); return VP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int N = d.batch, H = d.height, bf = h->bf16;
  int rc;
  if (h->params_dirty) {
    if ((rc = run_pack(h, h->G, st))) return rc;
    if (d.training) {
      if ((rc = run_pack(h, h->D, st))) return rc;
      // the perceptual trunk is frozen (pixrefer.py:325-327 restores it, no optimiser touches it): packed once per load, not per step
      if (h->vgg_dirty) { if ((rc = run_pack(h, h->V, st))) return rc; h->vgg_dirty = false; }
    }
    h->params_dirty = false;
  }
  h->in_targets = targets; h->in_masks = masks;
  PackInputsArgs pi;
  memset(&pi, 0, sizeof(pi));
  pi.inputs = inputs; pi.fg_inputs = fg_inputs; pi.gin = h->gin; pi.gfg = h->gfg; pi.din = h->din; pi.vin = h->vin;
  pi.N = N; pi.HW = H * H; pi.train = d.training; pi.fg_c = fg_c;
  phase_mark(h, st, 0);
  VP_HIP_CHECK(launch_pack_inputs(pi, bf, st));

  // the real half of the perceptual trunk only needs the packed inputs: side stream, under the generator forward
  bool split_vgg = d.training && h->overlap && h->ov_on;
  if (split_vgg) for (Layer& L : h->V.l) split_vgg = split_vgg && L.has_fwd_half;
  // (vp_pixrefer_set_option("vgg_real_fork", k): the side stream starts it in front of generator layer k instead of at once - the
  // encoders then share the device with one stream less and the real half fills the few-pixel section of the generator: EXPERIMENTS.md)
  auto vgg_real_half = [&]() -> int {
    VP_HIP_CHECK(hipEventRecord(h->ev_fork, st));
    VP_HIP_CHECK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
    Net& Vs = h->V;
    for (size_t i = 0; i < Vs.l.size(); ++i) {
      Layer& L = Vs.l[i];
      const bool pooled = L.scope == "conv1/conv1_2" || L.scope == "conv2/conv2_2";
      const bool fuse = pooled && plan_can_pool(L.fwd_half);
      // the real half has no backward pass: of conv1_2 / conv2_2 only the pooled image is ever read, the full-resolution store is skipped
      if ((rc = run_layer_fwd_half(h, Vs, L, 0, h->side, fuse ? Vs.t[L.out + 1].y : nullptr, true))) return rc;
      if (pooled && !fuse) {
        const Tens& ti = Vs.t[L.out];
        Tens& tp = Vs.t[L.out + 1];
        VP_HIP_CHECK(launch_maxpool_fwd(ti.y, tp.y, N, ti.H, ti.W, ti.C, bf, h->side));
      }
    }
    VP_HIP_CHECK(hipEventRecord(h->ev_join, h->side));
    return VP_OK;
  };
  bool vgg_real_started = false;
  if (split_vgg && h->vgg_fork_layer <= 0) { if ((rc = vgg_real_half())) return rc; vgg_real_started = true; }

  // generator: the two encoder branches (encoder_1..4 on `inputs`, encoder_fg_1..4 on `fg_inputs`, pixrefer.py:169-213) are
  // independent chains of small kernels until merged_encoder_2: the foreground branch runs on the branch stream
  const bool split_enc = d.training && h->overlap && h->ov_on;
  if (split_enc) {
    VP_HIP_CHECK(hipEventRecord(h->ev_bfork, st));
    VP_HIP_CHECK(hipStreamWaitEvent(h->branch, h->ev_bfork, 0));
    for (Layer& L : h->G.l) if (L.scope.rfind("encoder_fg_", 0) == 0) if ((rc = run_layer_fwd(h, h->G, L, h->branch, nullptr, 2))) return rc;
    VP_HIP_CHECK(hipEventRecord(h->ev_bjoin, h->branch));
  }
  for (Layer& L : h->G.l) {
    if (split_enc && L.scope.rfind("encoder_fg_", 0) == 0) continue;
    if (split_enc && L.scope == "merged_encoder_2") VP_HIP_CHECK(hipStreamWaitEvent(st, h->ev_bjoin, 0));
    if (split_vgg && !vgg_real_started && (int)(&L - &h->G.l[0]) >= h->vgg_fork_layer) { if ((rc = vgg_real_half())) return rc; vgg_real_started = true; }
    if (g_phase_detail) phase_mark(h, st, 8 + (int)(&L - &h->G.l[0]));          // per-layer marks: [8 + layer] = before the layer's forward
    if ((rc = run_layer_fwd(h, h->G, L, st))) return rc;
  }

  if (split_vgg && !vgg_real_started) { if ((rc = vgg_real_half())) return rc; vgg_real_started = true; }
  CompositeArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.y4 = h->y4; ca.targets = targets; ca.masks = masks; ca.o4 = h->o4; ca.outputs = h->outputs; ca.outputs_fg = h->outputs_fg;
  ca.din = h->din; ca.vin = h->vin; ca.partial = h->comp_partial; ca.N = N; ca.HW = H * H; ca.train = d.training;
  VP_HIP_CHECK(launch_composite_fwd(ca, bf, st));
  phase_mark(h, st, 1);
  if (!d.training) return VP_OK;

  // discriminator on [real1 | real2 | fake] (pixrefer.py:295-306).  It and the fake half of the VGG trunk both hang off the
  // composite only: the discriminator (conv + batch-norm glue, HBM-bound in between) runs on the branch stream under the VGG convs
  const bool split_d = h->overlap && h->ov_on && h->dsplit_on;
  hipStream_t sd = split_d ? h->branch : st;
  const int ssd = split_d ? 2 : 0;
  if (split_d) {
    VP_HIP_CHECK(hipEventRecord(h->ev_bfork, st));
    VP_HIP_CHECK(hipStreamWaitEvent(h->branch, h->ev_bfork, 0));
  }
  for (Layer& L : h->D.l) if ((rc = run_layer_fwd(h, h->D, L, sd, nullptr, ssd))) return rc;
  const int hd = H / 8 - 2;
  GanLossArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.logits = h->logits; ga.dl_d = h->dl_d; ga.dl_g = h->dl_g; ga.predict = h->predict; ga.losses = h->losses;
  ga.M = N * hd * hd; ga.gan_weight = d.gan_weight;
  VP_HIP_CHECK(launch_gan_loss(ga, bf, sd));
  if (split_d) VP_HIP_CHECK(hipEventRecord(h->ev_bjoin, h->branch));

  // VGG trunk on [real fg | Outputs_FG] (pixrefer.py:321).  The real half does not depend on the generator: when every layer has
  // a half-batch plan it was started on the side stream right after pack_inputs (below, `vgg_half`) and only the fake half runs here
  Net& V = h->V;
  auto vgg_half = [&](int half, hipStream_t s2) -> int {
    for (size_t i = 0; i < V.l.size(); ++i) {
      Layer& L = V.l[i];
      int rc2;
      const bool pooled = L.scope == "conv1/conv1_2" || L.scope == "conv2/conv2_2";
      const bool fuse = pooled && plan_can_pool(L.fwd_half);
      const Tens& ti = V.t[L.out];
      Tens& tp = V.t[pooled ? L.out + 1 : L.out];   // pool tensor follows in creation order
      const size_t oi = (size_t)half * N * ti.H * ti.W * ti.C * h->es, op = (size_t)half * N * tp.H * tp.W * tp.C * h->es;
      if ((rc2 = run_layer_fwd_half(h, V, L, half, s2, fuse ? (char*)tp.y + op : nullptr))) return rc2;
      if (pooled && !fuse) VP_HIP_CHECK(launch_maxpool_fwd((const char*)ti.y + oi, (char*)tp.y + op, N, ti.H, ti.W, ti.C, bf, s2));
    }
    return VP_OK;
  };
  if (split_vgg) {
    if ((rc = vgg_half(1, st))) return rc;
    VP_HIP_CHECK(hipStreamWaitEvent(st, h->ev_join, 0));      // the real half (side stream) is complete
  } else {
    for (size_t i = 0; i < V.l.size(); ++i) {
      Layer& L = V.l[i];
      const bool pooled = L.scope == "conv1/conv1_2" || L.scope == "conv2/conv2_2";
      const bool fuse = pooled && plan_can_pool(L.fwd);
      if ((rc = run_layer_fwd(h, V, L, st, fuse ? V.t[L.out + 1].y : nullptr))) return rc;
      if (pooled && !fuse) {
        const Tens& ti = V.t[L.out];
        Tens& tp = V.t[L.out + 1];   // pool tensor follows in creation order
        VP_HIP_CHECK(launch_maxpool_fwd(ti.y, tp.y, ti.N, ti.H, ti.W, ti.C, bf, st));
      }
    }
  }
  const Tens& f3 = V.t.back();
  PerceptualArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.f3 = f3.y; pa.df3 = f3.dz; pa.partial = h->perc_partial; pa.half = f3.elems() / 2; pa.l1_weight = d.l1_weight;
  VP_HIP_CHECK(launch_perceptual(pa, bf, st));

  if (split_d) VP_HIP_CHECK(hipStreamWaitEvent(st, h->ev_bjoin, 0));
  LossFinalArgs lf;
  memset(&lf, 0, sizeof(lf));
  lf.comp_partial = h->comp_partial; lf.n_comp = h->n_comp; lf.perc_partial = h->perc_partial; lf.n_perc = h->n_perc;
  lf.n_out = (double)N * H * H * 3; lf.n_feat = (double)pa.half; lf.losses = h->losses;
  lf.l1_weight = d.l1_weight; lf.gan_weight = d.gan_weight;
  VP_HIP_CHECK(launch_loss_final(lf, st));
  phase_mark(h, st, 2);
  return VP_OK;
}

static int backward_d_on(vp_pixrefer_t* h, hipStream_t st, bool side);
static int fork_d(vp_pixrefer_t* h, hipStream_t st);

// tf.train.AdamOptimizer on the arena range [off0, off1) of a net (= its layers l0 .. l1) followed by the re-pack of those layers'
// weights for the next step, both on stream s
static int update_range(vp_pixrefer_t* h, Net& n, float* m, float* v, float lr_t, size_t off0, size_t off1, int l0, int l1, hipStream_t s) {
  if (off1 > off0) {
    AdamArgs a;
    a.p = n.params + off0; a.g = n.grads + off0; a.m = m + off0; a.v = v + off0; a.n = off1 - off0;
    a.lr_t = lr_t; a.beta1 = h->upd.beta1; a.beta2 = h->upd.beta2; a.eps = h->upd.eps;
    VP_HIP_CHECK(launch_adam(a, s));
  }
  const size_t d0 = n.l[l0].desc0, d1 = n.l[l1].desc1;
  if (d1 > d0) VP_HIP_CHECK(launch_pack_weights(n.d_descs + d0, (int)(d1 - d0), n.params, n.packed, h->bf16, s));
  return VP_OK;
}
static int update_d(vp_pixrefer_t* h, hipStream_t s) {
  h->upd.d_done = true;
  return update_range(h, h->D, h->upd.m_d, h->upd.v_d, h->upd.lr_t_d, 0, h->D.nparams, 0, (int)h->D.l.size() - 1, s);
}

// Both gradient passes.  They are independent of each other (the discriminator-loss pass writes only the discriminator's gradient
// arena and its own dz / scratch buffers, the generator-loss pass the generator's), so the discriminator-loss pass runs on a second
// HIP stream: its large kernels fill the CUs that the generator's launch-bound bottleneck layers leave idle.  Fork / join by events;
// nothing here blocks the host.  Results are bit-identical to running the two passes one after the other.
int vp_pixrefer_backward(vp_pixrefer_t* h, void* stream) {
  if (!h || !h->d.training) { set_err("vp_pixrefer_backward: needs a training plan"); return VP_ERR_STATE; }
  hipStream_t st = (hipStream_t)stream;
  if (!h->overlap || !h->ov_on) {
    int rc = vp_pixrefer_backward_d(h, stream);
    if (rc) return rc;
    return vp_pixrefer_backward_g(h, stream);
  }
  int rc;
  // per handle, and cleared on every way out: a failed call must not leave a pending fork behind for a later stand-alone
  // vp_pixrefer_backward_g_stage (data-parallel path), which would start an un-joined discriminator pass
  h->dfork_pending = h->dfork_point;
  if (h->dfork_pending == 0 && (rc = fork_d(h, st))) { h->dfork_pending = 0; return rc; }
  rc = vp_pixrefer_backward_g(h, stream);
  h->dfork_pending = 0;
  if (rc) return rc;
  VP_HIP_CHECK(hipStreamWaitEvent(st, h->ev_join, 0));
  phase_mark(h, st, 7);
  return VP_OK;
}

static int fork_d(vp_pixrefer_t* h, hipStream_t st) {
  h->dfork_pending = 0;
  VP_HIP_CHECK(hipEventRecord(h->ev_fork, st));
  VP_HIP_CHECK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
  int rc = backward_d_on(h, h->side, true);
  if (rc) return rc;
  // forked behind the generator-loss pass through the discriminator (point 2): nobody reads the discriminator's weights any more
  if (h->upd.active && h->dfork_point == 2 && (rc = update_d(h, h->side))) return rc;
  VP_HIP_CHECK(hipEventRecord(h->ev_join, h->side));
  return VP_OK;
}

// Backward of both losses AND the two Adam updates + weight re-packs of the step (train_pixrefer.py:134-139 runs them in one
// sess.run as well).  A range of an arena is updated as soon as its gradients are final - the discriminator behind its loss pass on
// the side stream, the generator in the three buckets of vp_pixrefer_backward_g_stage on the branch stream - so the (HBM-bound)
// optimiser runs under the launch-bound tail of the generator backward instead of after it.  Same kernels on the same values as
// vp_pixrefer_backward + vp_adam_tf x 2 (+ the re-pack the next forward would do): bit-identical parameters.
int vp_pixrefer_backward_update(vp_pixrefer_t* h, float* m_g, float* v_g, float* m_d, float* v_d, int step_t_g, int step_t_d,
                                float lr, float beta1, float beta2, float eps, void* stream) {
  if (!h || !h->d.training) { set_err("vp_pixrefer_backward_update: needs a training plan"); return VP_ERR_STATE; }
  if (!m_g || !v_g || !m_d || !v_d || step_t_g < 1 || step_t_d < 1) { set_err("vp_pixrefer_backward_update: bad argument"); return VP_ERR_ARG; }
  auto lr_t = [&](int t) { return (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t))); };
  h->upd.active = true; h->upd.d_done = false;
  h->upd.m_g = m_g; h->upd.v_g = v_g; h->upd.m_d = m_d; h->upd.v_d = v_d;
  h->upd.lr_t_g = lr_t(step_t_g); h->upd.lr_t_d = lr_t(step_t_d); h->upd.beta1 = beta1; h->upd.beta2 = beta2; h->upd.eps = eps;
  int rc = vp_pixrefer_backward(h, stream);
  if (!rc && !h->upd.d_done) rc = update_d(h, (hipStream_t)stream);
  h->upd.active = false;
  if (rc) return rc;
  h->params_dirty = false;          // every layer was re-packed from the updated parameters
  return VP_OK;
}

// Data parallel: the Adam update + weight re-pack of ONE gradient bucket, for a host that has just all-reduced that bucket on `stream`
// (voicepuppet_amd/parallel.py GradExchange): which = 0 generator, bucket = the backward_g stage that completed it; which = 1
// discriminator (bucket 0 = the whole arena).  Same kernels on the same values as vp_adam_tf over the whole arena + the re-pack of
// the next forward, so parameters are bit-identical; once all four buckets of a step are through, the packed weights are current.
// The caller orders `stream` behind the stage that completed the bucket (the bucket's layers are no longer read by the backward pass).
int vp_pixrefer_update_bucket(vp_pixrefer_t* h, int which, int bucket, float* m, float* v, int step_t, float lr, float beta1, float beta2,
                              float eps, void* stream) {
  if (!h || !h->d.training || !m || !v || step_t < 1 || which < 0 || which > 1 || bucket < 0 || bucket > (which ? 0 : 2)) {
    set_err("vp_pixrefer_update_bucket: bad argument");
    return VP_ERR_ARG;
  }
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, step_t)) / (1.0 - pow((double)beta1, step_t)));
  h->upd.beta1 = beta1; h->upd.beta2 = beta2; h->upd.eps = eps;
  int rc;
  if (which == 1) {
    rc = update_range(h, h->D, m, v, lr_t, 0, h->D.nparams, 0, (int)h->D.l.size() - 1, (hipStream_t)stream);
    h->upd_mask |= 8;
  } else {
    Net& G = h->G;
    int i_md5 = -1, i_me2 = -1;
    for (int i = 0; i < (int)G.l.size(); ++i) {
      if (G.l[i].scope == "merged_decoder_5") i_md5 = i;
      if (G.l[i].scope == "merged_encoder_2") i_me2 = i;
    }
    const int l0 = bucket == 0 ? i_md5 : (bucket == 1 ? i_me2 : 0);
    const int l1 = bucket == 0 ? (int)G.l.size() - 1 : (bucket == 1 ? i_md5 - 1 : i_me2 - 1);
    const size_t off0 = G.l[l0].w_off, off1 = l1 + 1 < (int)G.l.size() ? G.l[l1 + 1].w_off : G.nparams;
    rc = update_range(h, G, m, v, lr_t, off0, off1, l0, l1, (hipStream_t)stream);
    h->upd_mask |= 1 << bucket;
  }
  if (rc) return rc;
  if (h->upd_mask == 15) { h->upd_mask = 0; h->params_dirty = false; }
  return VP_OK;
}

int vp_pixrefer_backward_d(vp_pixrefer_t* h, void* stream) {
  if (!h || !h->d.training) { set_err("vp_pixrefer_backward_d: needs a training plan"); return VP_ERR_STATE; }
  return backward_d_on(h, (hipStream_t)stream, false);
}

// The two halves of vp_pixrefer_backward for a host that has work of its own between them (data parallel: the staged generator
// backward with its bucketed all-reduces): _fork starts the discriminator-loss pass on the side stream behind everything already
// enqueued on `stream`; _join makes `stream` wait for it (the discriminator gradient arena is final after the join).
int vp_pixrefer_backward_d_fork(vp_pixrefer_t* h, void* stream) {
  if (!h || !h->d.training) { set_err("vp_pixrefer_backward_d_fork: needs a training plan"); return VP_ERR_STATE; }
  hipStream_t st = (hipStream_t)stream;
  h->forked = h->overlap && h->ov_on;
  if (!h->forked) return backward_d_on(h, st, false);
  // the pass starts where vp_pixrefer_backward starts it (vp_tune "d_backward_fork", default: inside stage 0 of the generator
  // backward, behind the generator-loss pass through D and the VGG trunk, i.e. under the generator's own layers); 0: at once
  h->dfork_pending = h->dfork_point;
  if (h->dfork_pending == 0) return fork_d(h, st);
  return VP_OK;
}

// The executor's low-priority side stream (the discriminator-loss pass and the fused optimiser run there), for a data-parallel host that
// wants to issue its collectives on it instead of creating one more stream: the GPU exposes a handful of hardware queues, and every
// extra stream beyond them shares one with another stream (false serialisation).  NULL when the plan does not overlap.
void* vp_pixrefer_side_stream(vp_pixrefer_t* h) { return (h && h->overlap) ? (void*)h->side : nullptr; }

// Streams the executor spreads a training step over: 4 (default: the caller's, side, branch and the foreground chain's own in the backward
// pass) or 3.  The device schedules a handful of hardware queues; a host that keeps a busy stream of its own next to the step (an input
// prefetcher copying and packing the next batch) asks for 3, or its stream shares a queue with one of the executor's (measured: a
// PCIe-fed step 8.5 -> 11.4 ms with five busy streams).  Takes effect from the next backward pass.
int vp_pixrefer_use_streams(vp_pixrefer_t* h, int n) {
  if (!h || (n != 3 && n != 4)) { set_err("vp_pixrefer_use_streams: 3 or 4"); return VP_ERR_ARG; }
  h->use_b2 = (n == 4) && h->overlap;
  return VP_OK;
}

// Schedule options of ONE plan (they used to be process globals: two engines in a process, or a profiling pass on one of them,
// changed each other's schedule).  Takes effect from the next forward / backward call of this handle; results are bit-identical
// under every setting (tests/test_gpu_soak.py).
//   "overlap"          0 / 1   the step on the caller's stream only / spread over the executor's streams (default 1)
//   "d_backward_fork"  0..2    where vp_pixrefer_backward starts the discriminator-loss pass on the side stream (default 2)
//   "d_beside_vgg"     0 / 1   discriminator forward / generator-loss backward on the branch stream beside the VGG passes (default 1)
int vp_pixrefer_set_option(vp_pixrefer_t* h, const char* key, int value) {
  if (!h || !key) { set_err("vp_pixrefer_set_option: null argument"); return VP_ERR_ARG; }
  const std::string k(key);
  if (k == "overlap") { h->ov_on = value != 0; return VP_OK; }
  if (k == "d_backward_fork") { h->dfork_point = value < 0 ? 0 : (value > 2 ? 2 : value); return VP_OK; }
  if (k == "d_beside_vgg") { h->dsplit_on = value != 0; return VP_OK; }
  if (k == "store_first_raw") { h->store_first_raw = value != 0; return VP_OK; }
  if (k == "bwd_sums_in_epilogue") { h->bst_on = value < 0 ? 0 : (value > 2 ? 2 : value); return VP_OK; }
  if (k == "vgg_real_fork") { h->vgg_fork_layer = value < 0 ? 0 : value; return VP_OK; }
  set_err("vp_pixrefer_set_option: unknown key %s", key);
  return VP_ERR_ARG;
}

// Counters of a plan since vp_pixrefer_create (tests): "bwd_sums_launches" = gradient launches whose epilogue also produced the sums of a
// batch-norm backward pass (IgemmArgs::bst_*).  -1: unknown key.
long long vp_pixrefer_counter(vp_pixrefer_t* h, const char* key) {
  if (!h || !key) return -1;
  if (std::string(key) == "bwd_sums_launches") return h->bst_count;
  return -1;
}

int vp_pixrefer_backward_d_join(vp_pixrefer_t* h, void* stream) {
  if (!h || !h->d.training) { set_err("vp_pixrefer_backward_d_join: needs a training plan"); return VP_ERR_STATE; }
  if (h->forked && h->dfork_pending) {   // (no stage 0 ran since the fork: start the pass now)
    const int rc = fork_d(h, (hipStream_t)stream);
    if (rc) return rc;
  }
  if (h->forked) VP_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, h->ev_join, 0));
  h->forked = false;
  return VP_OK;
}

static int backward_d_on(vp_pixrefer_t* h, hipStream_t st, bool side) {
  const int N = h->d.batch;
  int rc;
  Net& D = h->D;

  // ---- Discrim_loss -> discriminator* (pixrefer.py:396-400), all three applications at once ----
  for (Tens& t : D.t) { t.dz_written = false; t.bst_chunks = 0; t.cs_chunks = 0; }
  for (int i = (int)D.l.size() - 1; i >= 0; --i) {
    Layer& L = D.l[i];
    Tens& to = D.t[L.out];
    if (L.has_bn) if ((rc = run_bn_bwd(h, D, L, true, 0, 3 * N, 0, 3, st, side ? 1 : 0))) return rc;
    const bool save = L.need_bwd[0];
    if (i == 0) L.need_bwd[0] = false;          // no gradient w.r.t. the images for the D loss
    rc = run_layer_bwd(h, D, L, to.dz, true, false, 0, 3 * N, 0, st, side ? 1 : 0);
    L.need_bwd[0] = save;
    if (rc) return rc;
  }

  return VP_OK;
}

int vp_pixrefer_backward_g(vp_pixrefer_t* h, void* stream) { return vp_pixrefer_backward_g_stage(h, -1, stream); }

int vp_pixrefer_backward_g_stages(void) { return 3; }

// Stage s of the generator-loss backward pass; stage boundaries are where a contiguous range of the generator's flat
// gradient arena becomes final, so a data-parallel host can start that bucket's all-reduce while the next stage computes:
//   0: D(fake) + VGG + composite, then decoder_1 .. merged_decoder_5   -> arena [merged_decoder_5 .. end) final
//   1: merged_encoder_5 .. merged_encoder_2                              -> arena [merged_encoder_2 .. merged_decoder_5) final
//   2: encoder_fg_4 .. encoder_1                                         -> arena [0 .. merged_encoder_2) final
// stage < 0 runs all three.
int vp_pixrefer_backward_g_stage(vp_pixrefer_t* h, int stage, void* stream) {
  if (!h || !h->d.training) { set_err("vp_pixrefer_backward_g: needs a training plan"); return VP_ERR_STATE; }
  if (stage > 2) { set_err("vp_pixrefer_backward_g_stage: stage must be < 3"); return VP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const vp_pixrefer_desc& d = h->d;
  const int N = d.batch, H = d.height, bf = h->bf16, es = h->es;
  int rc;
  Net &G = h->G, &D = h->D, &V = h->V;
  // generator layers are walked last to first; l_hi / l_lo = the layer range of this stage
  int i_md5 = -1, i_me2 = -1;
  for (int i = 0; i < (int)G.l.size(); ++i) {
    if (G.l[i].scope == "merged_decoder_5") i_md5 = i;
    if (G.l[i].scope == "merged_encoder_2") i_me2 = i;
  }
  const int l_hi = stage <= 0 ? (int)G.l.size() - 1 : (stage == 1 ? i_md5 - 1 : i_me2 - 1);
  const int l_lo = stage < 0 ? 0 : (stage == 0 ? i_md5 : (stage == 1 ? i_me2 : 0));
  if (stage <= 0) {
  // ---- Gen_loss -> generator* (pixrefer.py:402-407) ----
  // (a) GAN term through the fake application of the discriminator (dX only, pre-update weights)
  // ... on the branch stream: independent of the VGG pass (b) until the composite (c) adds the two image gradients
  const bool split_d = h->overlap && h->ov_on && h->dsplit_on;
  hipStream_t sd = split_d ? h->branch : st;
  const int ssd = split_d ? 2 : 0;
  if (split_d) {
    VP_HIP_CHECK(hipEventRecord(h->ev_bfork, st));
    VP_HIP_CHECK(hipStreamWaitEvent(h->branch, h->ev_bfork, 0));
  }
  for (Tens& t : D.t) { t.dz2_written = false; t.bst_chunks_g = 0; }
  for (int i = (int)D.l.size() - 1; i >= 0; --i) {
    Layer& L = D.l[i];
    Tens& to = D.t[L.out];
    if (L.has_bn) if ((rc = run_bn_bwd(h, D, L, false, 2 * N, N, 2, 1, sd, ssd, true))) return rc;
    const void* dy = (i == (int)D.l.size() - 1) ? h->dl_g : to.dz2;
    if ((rc = run_layer_bwd(h, D, L, dy, false, true, 2 * N, N, 2, sd, ssd, true))) return rc;
  }
  if (split_d) VP_HIP_CHECK(hipEventRecord(h->ev_bjoin, h->branch));
  if (h->dfork_pending == 1 && (rc = fork_d(h, st))) return rc;
  // (b) perceptual term through the VGG trunk, fake half only (dX only: VGG is frozen)
  for (Tens& t : V.t) t.dz_written = false;
  for (int i = (int)V.l.size() - 1; i >= 0; --i) {
    Layer& L = V.l[i];
    Tens& to = V.t[L.out];
    Tens& ti = V.t[L.src[0]];
    // dz of a conv output here is already w.r.t. the pre-relu value (perceptual seed / epilogue / pool bwd)
    IgemmArgs a = L.bwd_alt[0].a;
    set_single_src(a.x, to.dz, L.g.CoutT, nullptr, nullptr, ACT_NONE, 0);
    a.Wp = V.packed + L.pk_bwd_alt[0] * es;
    a.partial = (float*)h->scratch;
    const bool from_pool = (ti.name == "pool1" || ti.name == "pool2");
    if (ti.is_input) {
      a.Y = h->d_vin;
    } else {
      a.Y = ti.dz;
      if (!from_pool) {   // producer is a conv+relu: multiply by relu'(stored output of the fake half)
        a.ref = (const char*)ti.y + ti.elems() / 2 * es;
        a.ref_act = ACT_RELU;
      }
    }
    a.zeros = h->zeros;
    profile_tag((L.scope + ":bwd").c_str());
    VP_HIP_CHECK(launch_igemm(a, bf, L.bwd_alt[0].cfg, st));
    if (from_pool) {
      // ti = pool output; its input is the conv tensor created just before it
      const int ci = L.src[0] - 1;
      Tens& tc = V.t[ci];
      const char* xin = (const char*)tc.y + tc.elems() / 2 * es;
      VP_HIP_CHECK(launch_maxpool_bwd(xin, ti.dz, tc.dz, N, tc.H, tc.W, tc.C, bf, st));
    }
  }
  if (split_d) VP_HIP_CHECK(hipStreamWaitEvent(st, h->ev_bjoin, 0));
  if (h->dfork_pending == 2 && (rc = fork_d(h, st))) return rc;
  // (c) composite + L1 / matte terms -> gradient w.r.t. the pre-tanh generator output
  CompositeArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.targets = h->in_targets; ca.masks = h->in_masks; ca.o4 = h->o4; ca.outputs = h->outputs;
  ca.d_din = h->d_din; ca.d_vin = h->d_vin; ca.dy4 = h->dy4; ca.N = N; ca.HW = H * H; ca.l1_weight = d.l1_weight;
  VP_HIP_CHECK(launch_composite_bwd(ca, bf, st));
  phase_mark(h, st, 3);
  for (Tens& t : G.t) { t.dz_written = false; t.dz_writes = 0; t.bn_bwd_done = false; t.bst_chunks = 0; t.cs_chunks = 0; }
  }
  // (d) generator, last layer first.  Below merged_encoder_2 the two encoder branches are independent again: the foreground
  // branch (encoder_fg_4 .. encoder_fg_1) runs on the branch stream, joined before this call returns control of `st`
  const bool split_enc = h->overlap && h->ov_on && l_lo == 0;
  const bool wsplit = h->overlap && h->ov_on;
  bool forked = false, fg_forked = false;
  bool b2_used = false;            // something of this call runs on the second branch stream (joined wherever `branch` is)
  for (int i = l_hi; i >= l_lo; --i) {
    Layer& L = G.l[i];
    Tens& to = G.t[L.out];
    const bool fg = split_enc && L.scope.rfind("encoder_fg_", 0) == 0;
    if (fg && !fg_forked) {      // the branch's first layer needs merged_encoder_2's data gradient, enqueued on `st` just before
      VP_HIP_CHECK(hipEventRecord(h->ev_bfork, st));
      if (h->use_b2 && !h->branch2) { const int rc2 = pool_streams(true, nullptr, nullptr, &h->branch2); if (rc2) return rc2; }
      VP_HIP_CHECK(hipStreamWaitEvent(h->use_b2 ? h->branch2 : h->branch, h->ev_bfork, 0));
      forked = fg_forked = true;
      b2_used = h->use_b2;
    }
    // (the foreground chain has a stream of its own: on `branch` it queued behind the weight gradients of every layer before it and
    // ended the step 0.2-0.6 ms after the caller's chain)
    hipStream_t s2 = fg ? (h->use_b2 ? h->branch2 : h->branch) : st;
    const int ss = fg ? (h->use_b2 ? 3 : 2) : 0;
    if (i == i_md5 - 1) phase_mark(h, st, 4);
    if (i == i_me2 - 1) phase_mark(h, st, 5);
    if (g_phase_detail && !fg) phase_mark(h, st, 32 + i);                         // [32 + layer] = before the layer's backward on `st`
    if (L.has_bn) if ((rc = run_bn_bwd(h, G, L, true, 0, N, 0, 1, s2, ss))) return rc;
    if (wsplit && !fg) {
      // the weight gradient of a layer hangs off the chain (only its data gradient feeds the next layer): branch stream
      // (alternating the weight gradients between the two branch streams was measured: slower - they then sit in front of the
      // foreground chain again; batch 4 2.55 vs 2.50 ms)
      VP_HIP_CHECK(hipEventRecord(h->ev_bfork, st));
      VP_HIP_CHECK(hipStreamWaitEvent(h->branch, h->ev_bfork, 0));
      forked = true;
      // (the data gradient of a decoder's skip-connection source on the branch stream too was measured slower - EXPERIMENTS.md - and
      // raced with the fourth stream: removed.  Stage 0's weight gradients alternating between `branch` and the then idle `branch2`:
      // +0.01 .. 0.05 ms at batch 4 / 8 / 32 - the weight gradients, the optimiser and the weight-streaming layers share HBM, a second
      // stream adds contention, not throughput; EXPERIMENTS.md 0.2)
      if ((rc = run_layer_bwd(h, G, L, to.dz, true, false, 0, N, 0, h->branch, 2, false, 1))) return rc;
      if ((rc = run_layer_bwd(h, G, L, to.dz, true, false, 0, N, 0, st, 0, false, 6))) return rc;
    } else {
      if ((rc = run_layer_bwd(h, G, L, to.dz, true, false, 0, N, 0, s2, ss))) return rc;
    }
    // a bucket of the gradient arena is final (stage boundaries above): its Adam update + re-pack, on the side stream
    // (holding the first bucket's update back until the end of stage 1, so that it does not run beside the weight-streaming few-pixel
    // layers of that stage, was measured: +0.02 .. 0.08 ms at batch 4, +-0 at 8 / 32 - EXPERIMENTS.md)
    if (h->upd.active && (i == i_md5 || i == i_me2 || i == 0)) {
      const int l0 = i, l1 = i == i_md5 ? (int)G.l.size() - 1 : (i == i_me2 ? i_md5 - 1 : i_me2 - 1);
      const size_t off0 = G.l[l0].w_off, off1 = l1 + 1 < (int)G.l.size() ? G.l[l1 + 1].w_off : G.nparams;
      hipStream_t su = st;
      const bool on_side = h->overlap && h->ov_on && h->dfork_pending == 0 && h->dfork_point >= 0;
      if (h->overlap && h->ov_on && on_side) {
        // the (HBM-bound) optimiser + re-pack of the bucket go to the SIDE stream, behind the discriminator-loss pass that runs there:
        // on the branch stream they sat between the weight gradients of the layers still to come and delayed the end of the step.
        // The bucket is final when its weight gradients (branch stream) and its data gradients (this stream: they read the packed
        // weights the re-pack overwrites) are done; vp_pixrefer_backward's closing wait on ev_join covers the updates.
        VP_HIP_CHECK(hipEventRecord(h->ev_upd_b, h->branch));
        VP_HIP_CHECK(hipEventRecord(h->ev_upd_m, st));
        VP_HIP_CHECK(hipStreamWaitEvent(h->side, h->ev_upd_b, 0));
        VP_HIP_CHECK(hipStreamWaitEvent(h->side, h->ev_upd_m, 0));
        if (b2_used) {                                   // weight gradients / the foreground encoders' chain on the second branch stream
          VP_HIP_CHECK(hipEventRecord(h->ev_b2join, h->branch2));
          VP_HIP_CHECK(hipStreamWaitEvent(h->side, h->ev_b2join, 0));
        }
        su = h->side;
      } else if (h->overlap && h->ov_on) {
        VP_HIP_CHECK(hipEventRecord(h->ev_bfork, st));
        VP_HIP_CHECK(hipStreamWaitEvent(h->branch, h->ev_bfork, 0));
        if (b2_used) {
          VP_HIP_CHECK(hipEventRecord(h->ev_b2join, h->branch2));
          VP_HIP_CHECK(hipStreamWaitEvent(h->branch, h->ev_b2join, 0));
        }
        su = h->branch; forked = true;
      }
      if ((rc = update_range(h, G, h->upd.m_g, h->upd.v_g, h->upd.lr_t_g, off0, off1, l0, l1, su))) return rc;
      if (h->overlap && h->ov_on && on_side) VP_HIP_CHECK(hipEventRecord(h->ev_join, h->side));
    }
  }
  if (forked) {
    if (g_phase_detail && (stage < 0 || stage == 2)) phase_mark(h, st, 60);      // [60] the caller's chain is done, [6] the branch stream's weight gradients too
    VP_HIP_CHECK(hipEventRecord(h->ev_bjoin, h->branch));
    VP_HIP_CHECK(hipStreamWaitEvent(st, h->ev_bjoin, 0));
    if (b2_used) {
      VP_HIP_CHECK(hipEventRecord(h->ev_b2join, h->branch2));
      VP_HIP_CHECK(hipStreamWaitEvent(st, h->ev_b2join, 0));
    }
  }
  if (stage < 0 || stage == 2) phase_mark(h, st, 6);
  return VP_OK;
}

int vp_profile_enable(int on) { profile_enable(on); return VP_OK; }
size_t vp_profile_collect(char* json, size_t cap) { return profile_collect(json, cap); }

int vp_pixrefer_tensor(vp_pixrefer_t* h, const char* name, void** ptr, int64_t shape[4], int* dtype) {
  if (!h || !name || !ptr) return VP_ERR_ARG;
  const int N = h->d.batch, H = h->d.height;
  auto ret = [&](void* p, int64_t a, int64_t b, int64_t c, int64_t e, int dt) {
    *ptr = p;
    if (shape) { shape[0] = a; shape[1] = b; shape[2] = c; shape[3] = e; }
    if (dtype) *dtype = dt;
    return p ? VP_OK : VP_ERR_STATE;
  };
  const std::string s(name);
  const int cd = h->bf16 ? VP_BF16 : VP_F32;
  if (s == "Outputs_raw") return ret(h->outputs, N, H, H, 3, VP_F32);
  if (s == "Outputs_FG") return ret(h->outputs_fg, N, H, H, 3, VP_F32);
  if (s == "gen_out4") return ret(h->o4, N, H, H, 4, VP_F32);
  if (s == "losses") return ret(h->losses, 8, 1, 1, 1, VP_F32);
  const int hd = H / 8 - 2;
  if (s == "Predict") return ret(h->predict, 2, N, hd, hd, VP_F32);
  if (s == "logits") return ret(h->logits, 3 * N, hd, hd, 1, VP_F32);
  if (s == "d_gen_out4") return ret(h->dy4, N, H, H, 8, cd);
  if (s == "d_din") return ret(h->d_din, N, H, H, 8, cd);
  if (s == "d_vin") return ret(h->d_vin, N, H, H, 8, cd);
  const size_t c = s.find('/');
  if (c == std::string::npos) return VP_ERR_ARG;
  const std::string net = s.substr(0, c);
  std::string rest = s.substr(c + 1);
  Net* n = net == "g" ? &h->G : net == "d" ? &h->D : net == "v" ? &h->V : nullptr;
  if (!n) return VP_ERR_ARG;
  std::string field;
  const size_t k = rest.find(':');
  if (k != std::string::npos) { field = rest.substr(k + 1); rest = rest.substr(0, k); }
  for (Tens& t : n->t) {
    if (t.name != rest) continue;
    if (field.empty() && !h->store_first_raw && t.producer >= 0 && first_layer_acts_fused(h, *n, n->l[t.producer])) {
      set_err("vp_pixrefer_tensor: the raw output of %s is not stored (its kernel writes the consumers' activations); vp_pixrefer_set_option(h, \"store_first_raw\", 1) before the forward pass", s.c_str());
      return VP_ERR_STATE;
    }
    if (field.empty()) return ret(t.y, t.N, t.H, t.W, t.is_f32 ? (t.name == "decoder_1" ? 4 : 1) : t.C, (t.is_f32 || t.hi) ? VP_F32 : cd);
    if (field == "dz32") return t.hi ? ret(t.dz32, t.N, t.H, t.W, t.C, VP_F32) : VP_ERR_ARG;
    if (field == "dy") return ret(t.dz, n == &h->V ? t.N / 2 : t.N, t.H, t.W, t.C, cd);
    if (!t.has_bn) return VP_ERR_ARG;
    if (field == "scale") return ret(t.bn.a, n->groups, t.C, 1, 1, VP_F32);
    if (field == "shift") return ret(t.bn.b, n->groups, t.C, 1, 1, VP_F32);
    if (field == "mean") return ret(t.bn.mu, n->groups, t.C, 1, 1, VP_F32);
    if (field == "rstd") return ret(t.bn.rstd, n->groups, t.C, 1, 1, VP_F32);
    return VP_ERR_ARG;
  }
  return VP_ERR_ARG;
}

// One of the reference's node values formed on the device from the last forward pass, into a caller-owned device buffer of N * H * H * 3
// elements: what = 0 Outputs (float32, deprocessed: (x + 1) / 2, pixrefer.py:424), 1 the same as uint8 frames (clamped, * 255, truncated:
// what infer_bfmvid.py:243 writes out), 2 Alphas (float32, tiled to three channels, pixrefer.py:284), 3 Outputs_FG as the graph of this
// plan defines it - build_train_op: the composite's tensor; build_inference_op: deprocess(Outputs_FG + Alphas - 1), the quirk of
// pixrefer.py:436.  No framework kernel, no synchronisation.
int vp_pixrefer_fetch(vp_pixrefer_t* h, int what, void* dst, void* stream) {
  if (!h || !dst || what < 0 || what > 3) { set_err("vp_pixrefer_fetch: bad argument"); return VP_ERR_ARG; }
  FetchArgs f;
  memset(&f, 0, sizeof(f));
  f.raw3 = h->outputs; f.fg3 = h->outputs_fg; f.o4 = h->o4; f.dst = dst;
  f.npix = (size_t)h->d.batch * h->d.height * h->d.height;
  f.mode = what;
  if (what == 3 && h->d.training) {          // the training graph's Outputs_FG is the stored tensor itself
    VP_HIP_CHECK(hipMemcpyAsync(dst, h->outputs_fg, f.npix * 3 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return VP_OK;
  }
  VP_HIP_CHECK(launch_fetch(f, (hipStream_t)stream));
  return VP_OK;
}

// bf16 transport of a gradient bucket: f32 range -> bf16 communication buffer, and back with the 1 / world scale (both need 32-byte
// aligned pointers: arena offsets of whole variables are, torch allocations are)
int vp_grad_pack_bf16(const float* src, void* dst_bf16, size_t n, void* stream) {
  if (!src || !dst_bf16 || n < 1 || ((uintptr_t)src & 31) || ((uintptr_t)dst_bf16 & 15)) { set_err("vp_grad_pack_bf16: bad argument"); return VP_ERR_ARG; }
  VP_HIP_CHECK(launch_grad_pack_bf16(src, dst_bf16, n, (hipStream_t)stream));
  return VP_OK;
}
int vp_grad_unpack_bf16(const void* src_bf16, float* dst, size_t n, float scale, void* stream) {
  if (!src_bf16 || !dst || n < 1 || ((uintptr_t)dst & 31) || ((uintptr_t)src_bf16 & 15)) { set_err("vp_grad_unpack_bf16: bad argument"); return VP_ERR_ARG; }
  VP_HIP_CHECK(launch_grad_unpack_bf16(src_bf16, dst, n, scale, (hipStream_t)stream));
  return VP_OK;
}

int vp_adam_tf(float* params, const float* grads, float* m, float* v, size_t n, int step_t,
               float lr, float beta1, float beta2, float eps, void* stream) {
  if (!params || !grads || !m || !v || step_t < 1) { set_err("vp_adam_tf: bad argument"); return VP_ERR_ARG; }
  AdamArgs a;
  a.p = params; a.g = grads; a.m = m; a.v = v; a.n = n;
  a.lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, step_t)) / (1.0 - pow((double)beta1, step_t)));
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  VP_HIP_CHECK(launch_adam(a, (hipStream_t)stream));
  return VP_OK;
}

}  // extern "C"
