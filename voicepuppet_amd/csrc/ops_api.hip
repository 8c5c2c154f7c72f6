// Stand-alone op entry points of the C ABI (parity tests drive the same kernels the step executor uses).
#include <string.h>

#include <string>

#include "conv_ops.h"
#include "errors.h"
#include "audio_args.h"
#include "launch.h"
#include "vp_common.h"

using namespace vp;

namespace {

bool conv_desc_ok(const vp_conv_desc* d) {
  if (!d || d->n < 1 || d->h < 1 || d->w < 1) return false;
  if (d->cin < 8 || (d->cin & (d->cin - 1))) return false;
  if (d->kind == 1 && (d->ksize != 4 || d->stride != 2 || d->pad != 1)) return false;
  if (d->kind != 0 && d->kind != 1) return false;
  if (d->ksize < 1 || d->ksize > 4 || d->stride < 1 || d->stride > 2) return false;
  if (d->kind == 0 && d->stride == 2 && (d->ksize != 4 || d->pad != 1 || (d->h & 1) || (d->w & 1))) return false;
  return d->dtype == VP_F32 || d->dtype == VP_BF16;
}

ConvGeomX geom_of(const vp_conv_desc* d) {
  return make_geom(d->kind, d->ksize, d->stride, d->pad, d->n, d->h, d->w, d->cin, d->cin, d->cout);
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// 256 zero bytes behind the op's workspace use (padding source of the LDS-DMA loader)
const void* zero_page(char* ws, size_t used, hipStream_t st) {
  char* z = ws + align256(used);
  (void)hipMemsetAsync(z, 0, 256, st);
  return z;
}

}  // namespace

extern "C" {

int vp_pixrefer_pack_frames(const unsigned char* example_frames, const unsigned char* current_frames, const int* crops,
                            int n, int img_size, float* inputs, float* fg_inputs, float* targets, float* masks, void* stream) {
  if (!example_frames || !current_frames || !crops || !inputs || !fg_inputs || !targets || !masks || n < 1 || img_size < 2) {
    set_err("vp_pixrefer_pack_frames: bad argument");
    return VP_ERR_ARG;
  }
  FramePackArgs a;
  a.ex = example_frames; a.cur = current_frames; a.crops = crops;
  a.inputs = inputs; a.fg_inputs = fg_inputs; a.targets = targets; a.masks = masks;
  a.N = n; a.S = img_size;
  VP_HIP_CHECK(launch_frame_pack(a, (hipStream_t)stream));
  return VP_OK;
}

extern "C" void vp_phase_marks_enable(int on);   // plan_pixrefer.hip

int vp_tune(const char* key, int value) {
  if (!key) return VP_ERR_ARG;
  const std::string k(key);
  if (k == "patch_tiles") { patch_tiles_knob() = value; return VP_OK; }
  if (k == "patch_min_blocks") { patch_minblk_knob() = value < 0 ? PATCH_MIN_BLOCKS_DEFAULT : value; return VP_OK; }      // (< 0: back to the default)
  if (k == "patch_small_tiles") { patch_small_knob() = value; return VP_OK; }
  if (k == "patch_long_k_on_256") { patch_longk_knob() = value; return VP_OK; }
  if (k == "wgrad_tr") { wgrad_tr_knob() = value; return VP_OK; }
  if (k == "patch3") { patch3_knob() = value; return VP_OK; }
  if (k == "c64") { c64_knob() = value; return VP_OK; }
  if (k == "dc64") { dc64_knob() = value; return VP_OK; }
  if (k == "cout1_wgrad_rows" && value > 0) { wgrad1_rows_knob() = value; return VP_OK; }
  if (k == "cout1_bwd") { cout1_knob() = value < 0 ? 256 : value; return VP_OK; }
  if (k == "s2c64") { s2c64_knob() = value; return VP_OK; }
  if (k == "wgrad_big") { wgrad_big_knob() = value; return VP_OK; }
  if (k == "patch4") { patch4_knob() = value; return VP_OK; }
  if (k == "s2c64_pair") { s2c64_pair_knob() = value; return VP_OK; }
  if (k == "bfm_dwproj") { bfm_dwproj_knob() = value; return VP_OK; }
  if (k == "patch2") { patch2_knob() = value; return VP_OK; }
  if (k == "patch_xcd") { patch_xcd_knob() = value; return VP_OK; }
  if (k == "smallp_max_pixels") { smallp_knob() = value; return VP_OK; }
  if (k == "igemm_splitk_target") { igemm_splitk_target_knob() = value < 0 ? IGEMM_SPLITK_TARGET_DEFAULT : value; return VP_OK; }
  if (k == "wgrad_fixed_x10") { wgrad_cost_knob(0) = value; return VP_OK; }
  if (k == "wgrad_slab_tile_x1000") { wgrad_cost_knob(2) = value < 0 ? 150 : value; return VP_OK; }      // (< 0: back to the default)
  if (k == "wgrad_slab_x100") { wgrad_cost_knob(1) = value; return VP_OK; }
  if (k == "smallp_split_target") { plan_misc_knob(0) = value; return VP_OK; }
  if (k == "igemm_splitk_cap") { plan_misc_knob(1) = value; return VP_OK; }
  if (k == "igemm_splitk_minchunk") { plan_misc_knob(2) = value; return VP_OK; }
  if (k == "wgrad_resident_blocks") { plan_misc_knob(3) = value; return VP_OK; }
  if (k == "igemm_small_grid") { igemm_small_grid_knob() = value; return VP_OK; }
  if (k == "thin_blocks_cout8" && value > 0) { thin_blocks_knob(0) = value; return VP_OK; }
  if (k == "thin_blocks_dcout8" && value > 0) { thin_blocks_knob(1) = value; return VP_OK; }
  if (k == "thin_blocks_cout4" && value > 0) { thin_blocks_knob(2) = value; return VP_OK; }
  if (k == "thin_blocks_cin8" && value > 0) { thin_blocks_knob(3) = value; return VP_OK; }
  if (k == "phase_marks") { vp_phase_marks_enable(value); return VP_OK; }
  set_err("vp_tune: unknown key %s", key);
  return VP_ERR_ARG;
}

// the few-pixel kernel's arrival counters live behind the zero page of the op's workspace, zeroed per call
static unsigned* smallp_counter_page(char* ws, size_t used, const IgemmArgs& a, hipStream_t st) {
  unsigned* c = (unsigned*)(ws + align256(used) + 256);
  (void)hipMemsetAsync(c, 0, (size_t)smallp_counters(a) * sizeof(unsigned), st);
  return c;
}

size_t vp_conv_workspace_bytes(const vp_conv_desc* d) {
  if (!conv_desc_ok(d)) return 0;
  const int bf = d->dtype == VP_BF16, es = bf ? 2 : 4;
  const ConvGeomX g = geom_of(d);
  size_t best = 0;
  {
    IgemmPlan p = plan_fwd(g, 0, bf);      // (the patch-kernel plan of the same layer needs no more: same packed block, no split-K slab)
    best = align256(p.pack_elems * es) + p.partial_bytes;
    if (plan_smallp_eligible(p, g.Cout, bf, d->cin, 0)) {
      plan_make_smallp(p, g.Cout, bf);
      const size_t b = align256(p.pack_elems * es) + align256(p.partial_bytes) + 512 + (size_t)smallp_counters(p.a) * sizeof(unsigned);
      if (b > best) best = b;
    }
  }
  if ((d->cout & (d->cout - 1)) == 0 && d->cout >= 8) {
    IgemmPlan p = plan_bwd_data(g, 0, 0, d->cin, d->cin, d->cin, bf);
    size_t b = align256(p.pack_elems * es) + p.partial_bytes;
    if (b > best) best = b;
    if (plan_smallp_eligible(p, d->cin, bf, d->cout, 0)) {
      plan_make_smallp(p, d->cin, bf);
      b = align256(p.pack_elems * es) + align256(p.partial_bytes) + 512 + (size_t)smallp_counters(p.a) * sizeof(unsigned);
      if (b > best) best = b;
    }
    for (int plain = 0; plain < 2; ++plain) {      // the LDS-DMA weight-gradient kernel (plain operands) tiles and splits differently
      WgradPlan w = plan_wgrad(g, bf, plain != 0);
      if (w.partial_bytes + 512 > best) best = w.partial_bytes + 512;
    }
  }
  return best + 1024;
}

int vp_conv_fwd(const vp_conv_desc* d, const void* x, const float* in_scale, const float* in_shift,
                const float* w, const float* bias, void* y, void* workspace, void* stream) {
  if (!conv_desc_ok(d) || !x || !w || !y || !workspace) { set_err("vp_conv_fwd: bad argument"); return VP_ERR_ARG; }
  const int bf = d->dtype == VP_BF16, es = bf ? 2 : 4;
  hipStream_t st = (hipStream_t)stream;
  const ConvGeomX g = geom_of(d);
  IgemmPlan p = plan_fwd(g, 0, bf);
  // the patch kernel moves plain bytes (LDS-DMA): only inputs that need no deferred affine / activation
  if (d->in_act == ACT_NONE && !in_scale && plan_smallp_eligible(p, g.Cout, bf, d->cin, 0)) plan_make_smallp(p, g.Cout, bf);
  else if (d->in_act == ACT_NONE && !in_scale && plan_patch_eligible(p, g.Cout, bf, true)) plan_make_patch(p, g.Cout, bf);
  else if (d->in_act == ACT_NONE && !in_scale && plan_patch2_eligible(p, g.Cout, bf, d->cin, 0)) plan_make_patch2(p, g.Cout, bf);
  else if (d->in_act == ACT_NONE && !in_scale && d->out_act == ACT_NONE && plan_s2c64_eligible(p, g.Cout, bf, d->cin, 0)) plan_make_s2c64(p);
  char* ws = (char*)workspace;
  VP_HIP_CHECK(launch_pack_weights_one(p.pack, w, ws, bf, st));
  IgemmArgs a = p.a;
  set_single_src(a.x, x, d->cin, in_scale, in_shift, d->in_act, 0);
  a.Wp = ws;
  a.partial = (float*)(ws + align256(p.pack_elems * es));
  a.Y = y; a.ldY = d->cout; a.bias = bias; a.out_act = d->out_act;
  a.zeros = zero_page(ws, align256(p.pack_elems * es) + p.partial_bytes, st);
  if (a.patch == 3) a.sp_cnt = smallp_counter_page(ws, align256(p.pack_elems * es) + p.partial_bytes, a, st);
  VP_HIP_CHECK(launch_igemm(a, bf, p.cfg, st));
  return VP_OK;
}

int vp_conv_bwd_data(const vp_conv_desc* d, const void* dy, const float* w, void* dx, void* workspace, void* stream) {
  if (!conv_desc_ok(d) || !dy || !w || !dx || !workspace) { set_err("vp_conv_bwd_data: bad argument"); return VP_ERR_ARG; }
  if (d->cout < 8 || (d->cout & (d->cout - 1))) { set_err("vp_conv_bwd_data: cout must be a power of two >= 8"); return VP_ERR_ARG; }
  const int bf = d->dtype == VP_BF16, es = bf ? 2 : 4;
  hipStream_t st = (hipStream_t)stream;
  const ConvGeomX g = geom_of(d);
  IgemmPlan p = plan_bwd_data(g, 0, 0, d->cin, d->cin, d->cin, bf);
  if (plan_smallp_eligible(p, d->cin, bf, d->cout, 0)) plan_make_smallp(p, d->cin, bf);
  else if (plan_patch_eligible(p, d->cin, bf, true)) plan_make_patch(p, d->cin, bf);
  else if (plan_patch2_eligible(p, d->cin, bf, d->cout, 0)) plan_make_patch2(p, d->cin, bf);
  char* ws = (char*)workspace;
  VP_HIP_CHECK(launch_pack_weights_one(p.pack, w, ws, bf, st));
  IgemmArgs a = p.a;
  set_single_src(a.x, dy, d->cout, nullptr, nullptr, ACT_NONE, 0);
  a.Wp = ws;
  a.partial = (float*)(ws + align256(p.pack_elems * es));
  a.Y = dx;
  a.zeros = zero_page(ws, align256(p.pack_elems * es) + p.partial_bytes, st);
  if (a.patch == 3) a.sp_cnt = smallp_counter_page(ws, align256(p.pack_elems * es) + p.partial_bytes, a, st);
  VP_HIP_CHECK(launch_igemm(a, bf, p.cfg, st));
  return VP_OK;
}

int vp_conv_bwd_weight(const vp_conv_desc* d, const void* x, const float* in_scale, const float* in_shift,
                       const void* dy, float* dw, void* workspace, void* stream) {
  if (!conv_desc_ok(d) || !x || !dy || !dw || !workspace) { set_err("vp_conv_bwd_weight: bad argument"); return VP_ERR_ARG; }
  if (d->cout < 8 || (d->cout & (d->cout - 1))) { set_err("vp_conv_bwd_weight: cout must be a power of two >= 8"); return VP_ERR_ARG; }
  const int bf = d->dtype == VP_BF16;
  hipStream_t st = (hipStream_t)stream;
  const ConvGeomX g = geom_of(d);
  WgradPlan p = plan_wgrad(g, bf, !in_scale && d->in_act == ACT_NONE);
  WgradArgs a = p.a;
  PixSrc xs, ds;
  set_single_src(xs, x, d->cin, in_scale, in_shift, d->in_act, 0);
  set_single_src(ds, dy, d->cout, nullptr, nullptr, ACT_NONE, 0);
  if (d->kind == 0) { a.g = xs; a.d = ds; } else { a.g = ds; a.d = xs; }
  a.partial = (float*)workspace;
  a.dW = dw;
  a.zeros = zero_page((char*)workspace, p.partial_bytes, st);
  VP_HIP_CHECK(launch_wgrad(a, bf, p.cfg, st));
  return VP_OK;
}

size_t vp_bn_workspace_bytes(int pixels, int c, int dtype) {
  if (pixels < 1 || c < 8) return 0;
  const int nch = bn_nchunk(pixels, c, 1, dtype == VP_BF16);
  return (size_t)nch * 2 * c * sizeof(double) + 4 * (size_t)c * sizeof(float) + 1024;
}

static bool bn_ok(int pixels, int c, int dtype) {
  const int e = dtype == VP_BF16 ? 8 : 4;
  return pixels >= 1 && c >= e && c % e == 0 && 256 % (c / e) == 0 && (dtype == VP_F32 || dtype == VP_BF16);
}

int vp_bn_stats(const void* y, int pixels, int c, int dtype, const float* gamma, const float* beta, float eps,
                float* scale, float* shift, float* mean, float* rstd, void* workspace, void* stream) {
  if (!bn_ok(pixels, c, dtype) || !y || !gamma || !beta || !scale || !shift || !mean || !rstd || !workspace) {
    set_err("vp_bn_stats: bad argument");
    return VP_ERR_ARG;
  }
  BnArgs b;
  memset(&b, 0, sizeof(b));
  b.y = y; b.C = c; b.G = 1; b.Pg = pixels;
  b.nchunk = bn_nchunk(pixels, c, 1, dtype == VP_BF16);
  b.partial = (double*)workspace;
  b.gamma = gamma; b.beta = beta; b.aff_a = scale; b.aff_b = shift; b.mu = mean; b.rstd = rstd; b.eps = eps;
  VP_HIP_CHECK(launch_bn_stats(b, dtype == VP_BF16, (hipStream_t)stream));
  return VP_OK;
}

int vp_bn_bwd(const void* y, const void* dz, void* dy, int pixels, int c, int dtype, const float* gamma,
              const float* mean, const float* rstd, float* dgamma, float* dbeta, void* workspace, void* stream) {
  if (!bn_ok(pixels, c, dtype) || !y || !dz || !dy || !gamma || !mean || !rstd || !dgamma || !dbeta || !workspace) {
    set_err("vp_bn_bwd: bad argument");
    return VP_ERR_ARG;
  }
  BnArgs b;
  memset(&b, 0, sizeof(b));
  b.y = y; b.dz = dz; b.dy = dy; b.C = c; b.G = 1; b.Pg = pixels;
  b.nchunk = bn_nchunk(pixels, c, 1, dtype == VP_BF16);
  b.partial = (double*)workspace;
  float* f = (float*)((char*)workspace + (size_t)b.nchunk * 2 * c * sizeof(double));
  b.c1 = f; b.c2 = f + c;
  b.gamma = gamma; b.mu = const_cast<float*>(mean); b.rstd = const_cast<float*>(rstd);
  b.dgamma = dgamma; b.dbeta = dbeta;
  VP_HIP_CHECK(launch_bn_bwd(b, dtype == VP_BF16, (hipStream_t)stream));
  return VP_OK;
}

// ---- single pointwise / audio ops behind the step and BFMNet executors (SURVEY.md 8b list), exported for parity tests and reuse ----

int vp_maxpool2x2_fwd(const void* x, void* y, int n, int h, int w, int c, int dtype, void* stream) {
  const int e = dtype == VP_BF16 ? 8 : 4;
  if (!x || !y || n < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || c < e || c % e) { set_err("vp_maxpool2x2_fwd: bad argument"); return VP_ERR_ARG; }
  VP_HIP_CHECK(launch_maxpool_fwd(x, y, n, h, w, c, dtype == VP_BF16, (hipStream_t)stream));
  return VP_OK;
}

int vp_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int n, int h, int w, int c, int dtype, void* stream) {
  const int e = dtype == VP_BF16 ? 8 : 4;
  if (!x || !dy || !dx || n < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || c < e || c % e) { set_err("vp_maxpool2x2_bwd: bad argument"); return VP_ERR_ARG; }
  VP_HIP_CHECK(launch_maxpool_bwd(x, dy, dx, n, h, w, c, dtype == VP_BF16, (hipStream_t)stream));
  return VP_OK;
}

int vp_composite_fwd(const float* gen_out4, const float* targets, float* out4, float* outputs, float* outputs_fg, int n, int hw,
                     void* stream) {
  if (!gen_out4 || !targets || !out4 || !outputs || !outputs_fg || n < 1 || hw < 1) { set_err("vp_composite_fwd: bad argument"); return VP_ERR_ARG; }
  CompositeArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.y4 = gen_out4; ca.targets = targets; ca.o4 = out4; ca.outputs = outputs; ca.outputs_fg = outputs_fg; ca.N = n; ca.HW = hw; ca.train = 0;
  VP_HIP_CHECK(launch_composite_fwd(ca, 0, (hipStream_t)stream));
  return VP_OK;
}

int vp_gan_loss(const float* logits, void* seed_d, void* seed_g, float* predict, float* losses, int m, float gan_weight, int dtype,
                void* stream) {
  if (!logits || !seed_d || !seed_g || !predict || !losses || m < 1) { set_err("vp_gan_loss: bad argument"); return VP_ERR_ARG; }
  GanLossArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.logits = logits; ga.dl_d = seed_d; ga.dl_g = seed_g; ga.predict = predict; ga.losses = losses; ga.M = m; ga.gan_weight = gan_weight;
  VP_HIP_CHECK(launch_gan_loss(ga, dtype == VP_BF16, (hipStream_t)stream));
  return VP_OK;
}

int vp_dwconv7x3_bn_act(const float* x, const float* w, const float* bias, float* y, int b, int h, int wd, int c, void* stream) {
  if (!x || !w || !bias || !y || b < 1 || h < 1 || wd < 1 || c < 4 || (c & 3)) { set_err("vp_dwconv7x3_bn_act: bad argument"); return VP_ERR_ARG; }
  VP_HIP_CHECK(launch_dwconv7x3(x, w, bias, y, 0, b, h, wd, c, (hipStream_t)stream));
  return VP_OK;
}

int vp_maxpool_hw(const float* x, float* y, int b, int h, int w, int c, int kh, int kw, int sh, int sw, void* stream) {
  if (!x || !y || b < 1 || h < 1 || w < 1 || c < 4 || (c & 3) || kh < 1 || kw < 1 || sh < 1 || sw < 1) { set_err("vp_maxpool_hw: bad argument"); return VP_ERR_ARG; }
  // TF 'same': out = ceil(in / stride); total pad = max((out-1)*stride + k - in, 0), the odd unit goes to the end
  const int ho = (h + sh - 1) / sh, wo = (w + sw - 1) / sw;
  const int ph = (ho - 1) * sh + kh - h, pw = (wo - 1) * sw + kw - w;
  const int pt = ph > 0 ? ph / 2 : 0, pl = pw > 0 ? pw / 2 : 0;
  VP_HIP_CHECK(launch_maxpool_same(x, y, 0, 0, b, h, w, c, kh, kw, sh, sw, pt, pl, ho, wo, (hipStream_t)stream));
  return VP_OK;
}

int vp_gru_seq(const float* xg, const float* xc, const float* whg, const float* whc, const int* seq_len, float* out, int b, int t,
               void* stream) {
  if (!xg || !xc || !whg || !whc || !seq_len || !out || b < 1 || t < 1) { set_err("vp_gru_seq: bad argument"); return VP_ERR_ARG; }
  VP_HIP_CHECK(launch_gru_seq(xg, xc, whg, whc, seq_len, out, b, t, (hipStream_t)stream));
  return VP_OK;
}

}  // extern "C"

// CRC-32C (Castagnoli) of a host buffer, continuing from `crc` (0 to start): what TensorFlow's checkpoint bundles checksum every tensor
// with.  The checkpoint reader / writer (voicepuppet_amd/utils/tf_checkpoint.py) verifies 160 MB per PixReferNet restore; its numpy
// lane-parallel form runs at 0.1 GB/s, the hardware instruction at several GB/s.  Host code only.
extern "C" __attribute__((target("sse4.2"))) unsigned vp_crc32c(const void* data, size_t n, unsigned crc) {
  const unsigned char* p = (const unsigned char*)data;
  unsigned long long c = (unsigned long long)(crc ^ 0xFFFFFFFFu);
  while (n && ((size_t)p & 7)) { c = __builtin_ia32_crc32qi((unsigned)c, *p++); --n; }
  while (n >= 8) { c = __builtin_ia32_crc32di(c, *(const unsigned long long*)p); p += 8; n -= 8; }
  while (n) { c = __builtin_ia32_crc32qi((unsigned)c, *p++); --n; }
  return (unsigned)c ^ 0xFFFFFFFFu;
}
