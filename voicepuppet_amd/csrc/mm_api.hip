// Plain float32 matrix products behind the C ABI, on the repo's own MFMA kernels (conv_kernels.hip: the LDS-DMA implicit GEMM with
// one tap and the register-transposing weight-gradient kernel, v_mfma_f32_16x16x4_f32 - float32 products, float32 accumulation).
// The BFMNet training step (voicepuppet/bfmnet/bfmnet.py:215-323 over tinynet.py:12-142) is made of them: 1x1 convolutions on
// [pixels, channels] matrices with 32 .. 1536 channels (not powers of two), dense layers, the GRU's input / recurrent weight
// gradients and the [B*T, 64] x [64, 107127] face-shape products.  Row-major matrices, leading dimensions in floats:
//   vp_mm_fwd_f32         y[P,N]  = x[P,K] . w[K,N] (+ bias[N])     (w optionally stored transposed, [N,K])
//   vp_mm_bwd_data_f32    dx[P,K] (+)= dy[P,N] . w[K,N]^T
//   vp_mm_bwd_weight_f32  dw[K,N] = x[P,K]^T . dy[P,N]              (first k_real rows are written: the zero-padded stem)
// The contraction dimension of the first two (K, resp. N) must be a multiple of 16 floats (one 64-byte K chunk); padding columns
// must hold zeros.  Every call packs w into the kernels' chunk-major layout first (workspace), as the single-conv entry points do.
// Workspace contract: the FIRST 256 bytes must be zero on entry and are never written (the LDS-DMA loaders' padding source), so a
// chain of calls on one workspace needs no memset node per call.
#include <stdlib.h>
#include <string.h>

#include "conv_ops.h"
#include "errors.h"
#include "launch.h"
#include "vp_common.h"

using namespace vp;

namespace {
size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
}

extern "C" {

size_t vp_mm_workspace_bytes(int P, int K, int N) {
  if (P < 1 || K < 1 || N < 1) return 0;
  size_t best = 0;
  {
    ConvGeomX g = make_geom(0, 1, 1, 0, 1, P, 1, round_up(K, 16), K, N);
    IgemmPlan p = plan_fwd(g, 0, 0);
    best = al256(p.pack_elems * 4) + al256(p.partial_bytes);
  }
  {
    ConvGeomX g = make_geom(0, 1, 1, 0, 1, P, 1, K, K, round_up(N, 16));
    g.CoutT = round_up(N, 16);
    IgemmPlan p = plan_bwd_data(g, 0, 0, K, K, K, 0);
    const size_t b = al256(p.pack_elems * 4) + al256(p.partial_bytes);
    if (b > best) best = b;
    WgradPlan w = plan_wgrad(g, 0, true);
    if (al256(w.partial_bytes) > best) best = al256(w.partial_bytes);
  }
  return best + 1024 + 256;
}

// the plans of the two weight-consuming products (tile / split-K choice and the packed layout follow from P, K, N alone)
static IgemmPlan mm_fwd_plan(int P, int K, int N, int ldw, int w_transposed) {
  ConvGeomX g = make_geom(0, 1, 1, 0, 1, P, 1, K, K, N);
  IgemmPlan p = plan_fwd(g, 0, 0);
  // packed row n, element k = w[k][n]: source strides of the [K, N] matrix (or of its transpose [N, K])
  p.pack.s_ch = w_transposed ? 1 : ldw;
  p.pack.s_row = w_transposed ? ldw : 1;
  return p;
}
static IgemmPlan mm_bwd_plan(int P, int K, int N, int ldw, int w_transposed, int lddx) {
  ConvGeomX g = make_geom(0, 1, 1, 0, 1, P, 1, K, K, N);
  g.CoutT = round_up(N, 16);
  IgemmPlan p = plan_bwd_data(g, 0, 0, K, K, lddx, 0);
  // packed row k (a column of dx), element n = w[k][n]
  p.pack.s_row = w_transposed ? 1 : ldw;
  p.pack.s_ch = w_transposed ? ldw : 1;
  return p;
}

static int mm_fwd_run(const IgemmPlan& p, const float* x, int ldx, const void* packed, const float* bias, float* y, int ldy, void* workspace, hipStream_t st) {
  IgemmArgs a = p.a;
  set_single_src(a.x, x, ldx, nullptr, nullptr, ACT_NONE, 0);
  a.Wp = packed;
  a.partial = (float*)((char*)workspace + 256);
  a.Y = y; a.ldY = ldy; a.bias = bias; a.out_act = ACT_NONE;
  a.zeros = workspace;
  VP_HIP_CHECK(launch_igemm(a, 0, p.cfg, st));
  return VP_OK;
}
static int mm_bwd_run(const IgemmPlan& p, const float* dy, int lddy, const void* packed, float* dx, int lddx, int accumulate, void* workspace, hipStream_t st) {
  IgemmArgs a = p.a;
  set_single_src(a.x, dy, lddy, nullptr, nullptr, ACT_NONE, 0);
  a.Wp = packed;
  a.partial = (float*)((char*)workspace + 256);
  a.Y = dx; a.ldY = lddx; a.accumulate = accumulate ? 1 : 0;
  a.zeros = workspace;
  VP_HIP_CHECK(launch_igemm(a, 0, p.cfg, st));
  return VP_OK;
}

int vp_mm_fwd_f32(const float* x, int ldx, const float* w, int ldw, int w_transposed, const float* bias, float* y, int ldy,
                  int P, int K, int N, void* workspace, void* stream) {
  if (!x || !w || !y || !workspace || P < 1 || K < 16 || K % 16 || N < 1 || ldx < K || ldx % 4 || ldy < N) { set_err("vp_mm_fwd_f32: bad argument (K must be a multiple of 16)"); return VP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  IgemmPlan p = mm_fwd_plan(P, K, N, ldw, w_transposed);
  // workspace: zero page | split-K slabs | packed copy of w
  char* pk = (char*)workspace + 256 + al256(p.partial_bytes);
  VP_HIP_CHECK(launch_pack_weights_one(p.pack, w, pk, 0, st));
  return mm_fwd_run(p, x, ldx, pk, bias, y, ldy, workspace, st);
}

int vp_mm_bwd_data_f32(const float* dy, int lddy, const float* w, int ldw, int w_transposed, float* dx, int lddx, int accumulate,
                       int P, int K, int N, void* workspace, void* stream) {
  // contraction over N: the gradient matrix has lddy >= round_up(N, 16) columns, the ones beyond N hold zeros
  const int Np = round_up(N, 16);
  if (!dy || !w || !dx || !workspace || P < 1 || K < 1 || N < 1 || lddy < Np || lddy % 4 || lddx < K) { set_err("vp_mm_bwd_data_f32: bad argument"); return VP_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  IgemmPlan p = mm_bwd_plan(P, K, N, ldw, w_transposed, lddx);
  char* pk = (char*)workspace + 256 + al256(p.partial_bytes);
  VP_HIP_CHECK(launch_pack_weights_one(p.pack, w, pk, 0, st));
  return mm_bwd_run(p, dy, lddy, pk, dx, lddx, accumulate, workspace, st);
}

// ---- weights packed ahead of time: a training step packs every weight matrix ONCE (one launch over a descriptor table, behind the
// optimiser update) instead of once per product.  dir: 0 = the layout vp_mm_fwd_f32 reads, 1 = vp_mm_bwd_data_f32's. ----
size_t vp_mm_packed_bytes(int P, int K, int N, int dir) {
  if (P < 1 || K < 1 || N < 1) return 0;
  const IgemmPlan p = dir ? mm_bwd_plan(P, K, N, N, 0, K) : mm_fwd_plan(P, K, N, N, 0);
  return al256(p.pack_elems * sizeof(float));
}
size_t vp_mm_pack_desc_bytes(void) { return sizeof(PackDesc); }
// host: descriptor of one matrix for vp_mm_pack_table (w = master + w_off floats; packed block at packed + dst_off floats)
int vp_mm_pack_desc(size_t w_off, int ldw, int w_transposed, int P, int K, int N, int dir, size_t dst_off, void* desc) {
  if (!desc || P < 1 || K < 1 || N < 1) { set_err("vp_mm_pack_desc: bad argument"); return VP_ERR_ARG; }
  IgemmPlan p = dir ? mm_bwd_plan(P, K, N, ldw, w_transposed, K) : mm_fwd_plan(P, K, N, ldw, w_transposed);
  p.pack.src_off = w_off;
  p.pack.dst_off = dst_off;
  memcpy(desc, &p.pack, sizeof(PackDesc));
  return VP_OK;
}
int vp_mm_pack_table(const void* device_descs, int n, const float* master, void* packed, void* stream) {
  if (!device_descs || n < 1 || !master || !packed) { set_err("vp_mm_pack_table: bad argument"); return VP_ERR_ARG; }
  VP_HIP_CHECK(launch_pack_weights((const PackDesc*)device_descs, n, master, packed, 0, (hipStream_t)stream));
  return VP_OK;
}
int vp_mm_fwd_f32_packed(const float* x, int ldx, const void* packed_w, const float* bias, float* y, int ldy, int P, int K, int N,
                         void* workspace, void* stream) {
  if (!x || !packed_w || !y || !workspace || P < 1 || K < 16 || K % 16 || N < 1 || ldx < K || ldx % 4 || ldy < N) { set_err("vp_mm_fwd_f32_packed: bad argument"); return VP_ERR_ARG; }
  return mm_fwd_run(mm_fwd_plan(P, K, N, N, 0), x, ldx, packed_w, bias, y, ldy, workspace, (hipStream_t)stream);
}
int vp_mm_bwd_data_f32_packed(const float* dy, int lddy, const void* packed_w, float* dx, int lddx, int accumulate, int P, int K, int N,
                              void* workspace, void* stream) {
  const int Np = round_up(N, 16);
  if (!dy || !packed_w || !dx || !workspace || P < 1 || K < 1 || N < 1 || lddy < Np || lddy % 4 || lddx < K) { set_err("vp_mm_bwd_data_f32_packed: bad argument"); return VP_ERR_ARG; }
  return mm_bwd_run(mm_bwd_plan(P, K, N, N, 0, lddx), dy, lddy, packed_w, dx, lddx, accumulate, workspace, (hipStream_t)stream);
}

int vp_mm_bwd_weight_f32(const float* x, int ldx, const float* dy, int lddy, float* dw, int P, int K, int k_real, int N,
                         void* workspace, void* stream) {
  // dw [k_real, N] contiguous; x has ldx >= K columns (K a multiple of 4: 16-byte loader pieces), dy has lddy >= N (multiple of 4)
  if (!x || !dy || !dw || !workspace || P < 1 || K < 4 || K % 4 || k_real < 1 || k_real > K || N < 4 || N % 4 || ldx < K || ldx % 4 || lddy < N || lddy % 4) {
    set_err("vp_mm_bwd_weight_f32: bad argument");
    return VP_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  ConvGeomX g = make_geom(0, 1, 1, 0, 1, P, 1, K, k_real, N);
  WgradPlan p = plan_wgrad(g, 0, true);
  WgradArgs a = p.a;
  PixSrc xs, ds;
  set_single_src(xs, x, ldx, nullptr, nullptr, ACT_NONE, 0);
  set_single_src(ds, dy, lddy, nullptr, nullptr, ACT_NONE, 0);
  a.g = xs; a.d = ds;
  a.partial = (float*)((char*)workspace + 256);
  a.dW = dw;
  a.zeros = workspace;
  VP_HIP_CHECK(launch_wgrad(a, 0, p.cfg, st));
  return VP_OK;
}

}  // extern "C"
