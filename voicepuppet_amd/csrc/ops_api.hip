// Stand-alone op entry points of the C ABI (parity tests drive the same kernels the step executor uses).
#include <string.h>

#include "conv_ops.h"
#include "errors.h"
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

size_t vp_conv_workspace_bytes(const vp_conv_desc* d) {
  if (!conv_desc_ok(d)) return 0;
  const int bf = d->dtype == VP_BF16, es = bf ? 2 : 4;
  const ConvGeomX g = geom_of(d);
  size_t best = 0;
  {
    IgemmPlan p = plan_fwd(g, 0, bf);
    best = align256(p.pack_elems * es) + p.partial_bytes;
  }
  if ((d->cout & (d->cout - 1)) == 0 && d->cout >= 8) {
    IgemmPlan p = plan_bwd_data(g, 0, 0, d->cin, d->cin, d->cin, bf);
    size_t b = align256(p.pack_elems * es) + p.partial_bytes;
    if (b > best) best = b;
    WgradPlan w = plan_wgrad(g, bf);
    if (w.partial_bytes + 512 > best) best = w.partial_bytes + 512;
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
  char* ws = (char*)workspace;
  VP_HIP_CHECK(launch_pack_weights_one(p.pack, w, ws, bf, st));
  IgemmArgs a = p.a;
  set_single_src(a.x, x, d->cin, in_scale, in_shift, d->in_act, 0);
  a.Wp = ws;
  a.partial = (float*)(ws + align256(p.pack_elems * es));
  a.Y = y; a.ldY = d->cout; a.bias = bias; a.out_act = d->out_act;
  a.zeros = zero_page(ws, align256(p.pack_elems * es) + p.partial_bytes, st);
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
  char* ws = (char*)workspace;
  VP_HIP_CHECK(launch_pack_weights_one(p.pack, w, ws, bf, st));
  IgemmArgs a = p.a;
  set_single_src(a.x, dy, d->cout, nullptr, nullptr, ACT_NONE, 0);
  a.Wp = ws;
  a.partial = (float*)(ws + align256(p.pack_elems * es));
  a.Y = dx;
  a.zeros = zero_page(ws, align256(p.pack_elems * es) + p.partial_bytes, st);
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
  WgradPlan p = plan_wgrad(g, bf);
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

}  // extern "C"
