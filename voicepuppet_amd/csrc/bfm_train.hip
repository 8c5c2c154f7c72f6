// Kernels of the BFMNet TRAINING step (voicepuppet/bfmnet/bfmnet.py:215-323 over tinynet.py:7-212), f32, NHWC with any
// channel count that is a multiple of 4 (MfccNet's expanded widths 192 ... 1536 are not powers of two, which the batch-norm / weight-
// gradient kernels of the PixReferNet executor assume).  Everything that is NOT a plain matrix product lives here:
//   training-mode contrib batch_norm (no gamma) forward statistics and backward, relu / relu6 / leaky-relu forward + backward,
//   depthwise 7x3 weight gradient (forward and backward-data reuse dwconv7x3_kernel with its raw flag), SAME max-pool backward,
//   the 9x5 stem as im2col, the GRU recurrence forward (with saved gates) and backward through time, the vertex-space loss with
//   its gradient, sums of squares (regulariser, global-norm clipping).
// The matrix products themselves (1x1 convolutions, dense layers, GRU input / recurrent weight gradients, the [B*T,64] x [64,3n]
// face-shape products) run on the repo's own float32-MFMA kernels (mm_api.hip: igemm / wgrad_mm; no vendor GEMM since round 3); the
// step's scalar arithmetic (partial sums, the reported losses, the clip factor) are the small kernels at the end of this file.
#include <math.h>

#include "audio_args.h"
#include "errors.h"
#include "gru_device.h"
#include "vp_common.h"

namespace vp {

static inline int tblk(size_t work, int cap = 4096) {
  size_t b = (work + 255) / 256;
  if (b > (size_t)cap) b = cap;
  return b < 1 ? 1 : (int)b;
}

// ------------------------------------------------------------------------------------------------
// per-channel sums over the P rows of a [P, C] tensor.  MODE 0: sum x, sum x^2.  MODE 1: sum dz, sum dz * (x - mean) * rstd, where
// dz = da * act'(rstd * x + shift) when an activation follows the batch-norm (its backward is folded in: no dz tensor exists).
// A thread owns one channel quad (float4 loads: a row group of QL = 2^ql lanes reads QL*16 contiguous bytes) and every RL-th row of
// its row chunk, four rows in flight; grid (nchunk, ceil(C/4 / QL)); f64 partials [nchunk][2][C].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_d(int act, float y);

struct ChanSumsArgs {
  const float* x; const float* dz; const float* mean; const float* rstd; const float* shift;
  size_t P; int C; int ql; int act; double* partial;
};

template <int MODE>
__global__ __launch_bounds__(256) void chan_sums_kernel(const ChanSumsArgs a) {
  const int QL = 1 << a.ql, RL = 256 >> a.ql;
  const int q = blockIdx.y * QL + (threadIdx.x & (QL - 1)), rl = threadIdx.x >> a.ql, cq = a.C >> 2;
  const size_t per = (a.P + gridDim.x - 1) / gridDim.x;
  const size_t p0 = blockIdx.x * per, p1 = p0 + per < a.P ? p0 + per : a.P;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (q < cq) {
    float4 mu = {0, 0, 0, 0}, rs = mu, sh = mu;
    if (MODE == 1) {
      mu = *reinterpret_cast<const float4*>(a.mean + q * 4); rs = *reinterpret_cast<const float4*>(a.rstd + q * 4);
      if (a.act) sh = *reinterpret_cast<const float4*>(a.shift + q * 4);
    }
    const int act = a.act;
    auto fold = [&](const float4 v, float4 d) {
      if (MODE == 0) {
        s0[0] += v.x; s0[1] += v.y; s0[2] += v.z; s0[3] += v.w;
        s1[0] += (double)v.x * v.x; s1[1] += (double)v.y * v.y; s1[2] += (double)v.z * v.z; s1[3] += (double)v.w * v.w;
      } else {
        if (act) {
          d.x *= act_d(act, fmaf(rs.x, v.x, sh.x)); d.y *= act_d(act, fmaf(rs.y, v.y, sh.y));
          d.z *= act_d(act, fmaf(rs.z, v.z, sh.z)); d.w *= act_d(act, fmaf(rs.w, v.w, sh.w));
        }
        s0[0] += d.x; s0[1] += d.y; s0[2] += d.z; s0[3] += d.w;
        s1[0] += (double)d.x * ((v.x - mu.x) * rs.x); s1[1] += (double)d.y * ((v.y - mu.y) * rs.y);
        s1[2] += (double)d.z * ((v.z - mu.z) * rs.z); s1[3] += (double)d.w * ((v.w - mu.w) * rs.w);
      }
    };
    const size_t rs4 = (size_t)a.C, step = (size_t)RL;
    size_t p = p0 + rl;
    for (; p + 3 * step < p1; p += 4 * step) {
      const float* xp = a.x + p * rs4 + q * 4;
      const float4 v0 = *reinterpret_cast<const float4*>(xp), v1 = *reinterpret_cast<const float4*>(xp + step * rs4),
                   v2 = *reinterpret_cast<const float4*>(xp + 2 * step * rs4), v3 = *reinterpret_cast<const float4*>(xp + 3 * step * rs4);
      float4 d0 = v0, d1 = v0, d2 = v0, d3 = v0;
      if (MODE == 1) {
        const float* dp = a.dz + p * rs4 + q * 4;
        d0 = *reinterpret_cast<const float4*>(dp); d1 = *reinterpret_cast<const float4*>(dp + step * rs4);
        d2 = *reinterpret_cast<const float4*>(dp + 2 * step * rs4); d3 = *reinterpret_cast<const float4*>(dp + 3 * step * rs4);
      }
      fold(v0, d0); fold(v1, d1); fold(v2, d2); fold(v3, d3);
    }
    for (; p < p1; p += step) {
      const float4 v = *reinterpret_cast<const float4*>(a.x + p * rs4 + q * 4);
      float4 d = v;
      if (MODE == 1) d = *reinterpret_cast<const float4*>(a.dz + p * rs4 + q * 4);
      fold(v, d);
    }
  }
  __shared__ double sm[256];
  for (int k = 0; k < 8; ++k) {
    __syncthreads();
    sm[threadIdx.x] = k < 4 ? s0[k] : s1[k - 4];
    __syncthreads();
    if (rl == 0 && q < cq) {
      double t = 0;
      for (int r = 0; r < RL; ++r) t += sm[threadIdx.x + (r << a.ql)];
      a.partial[((size_t)blockIdx.x * 2 + (k >> 2)) * a.C + q * 4 + (k & 3)] = t;
    }
  }
}

// sum of the nchunk partials of one channel: a block = 16 channels x 16 lanes (the partial rows are read 128 bytes at a time, sixteen
// rows in flight); valid in the lanes with threadIdx.x < 16 afterwards
__device__ __forceinline__ void chunk_sums(const double* __restrict__ partial, int nchunk, int C, int c, double& s0, double& s1) {
  __shared__ double sm[2][256];
  const int lane = threadIdx.x >> 4;
  double a = 0, b = 0;
  if (c < C)
    for (int k = lane; k < nchunk; k += 16) { a += partial[((size_t)k * 2) * C + c]; b += partial[((size_t)k * 2 + 1) * C + c]; }
  sm[0][threadIdx.x] = a; sm[1][threadIdx.x] = b;
  __syncthreads();
  s0 = 0; s1 = 0;
  if (threadIdx.x < 16)
    for (int l = 0; l < 16; ++l) { s0 += sm[0][threadIdx.x + 16 * l]; s1 += sm[1][threadIdx.x + 16 * l]; }
}

// forward finalize: mean, biased variance, rstd = 1/sqrt(var + eps), scale = rstd, shift = beta - mean * rstd
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const double* __restrict__ partial, int nchunk, size_t P, int C, const float* __restrict__ beta,
                                                              float eps, float* mean, float* var, float* rstd, float* scale, float* shift) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  double s0, s1;
  chunk_sums(partial, nchunk, C, c, s0, s1);
  if (c >= C || threadIdx.x >= 16) return;
  const double m = s0 / (double)P;
  double v = s1 / (double)P - m * m;
  if (v < 0) v = 0;
  const float r = (float)(1.0 / sqrt(v + (double)eps));
  mean[c] = (float)m; var[c] = (float)v; rstd[c] = r; scale[c] = r; shift[c] = (float)((double)beta[c] - m * (double)r);
}

// backward finalize: c1 = mean(dz), c2 = mean(dz * xhat); dbeta = sum dz
__global__ __launch_bounds__(256) void bn_bwd_finalize2_kernel(const double* __restrict__ partial, int nchunk, size_t P, int C, float* c1, float* c2,
                                                               float* dbeta) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  double s0, s1;
  chunk_sums(partial, nchunk, C, c, s0, s1);
  if (c >= C || threadIdx.x >= 16) return;
  c1[c] = (float)(s0 / (double)P); c2[c] = (float)(s1 / (double)P); dbeta[c] = (float)s0;
}

// dx = rstd * (dz - c1 - xhat * c2)       (no gamma: tf.contrib batch_norm scale=False); dz = da * act'(rstd * x + shift) when act != 0
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const float* __restrict__ x, const float* __restrict__ dz, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ shift, int act,
                                                            const float* __restrict__ c1, const float* __restrict__ c2, size_t P, int C,
                                                            float* __restrict__ dx) {
  const int cq = C >> 2;
  const size_t total = P * cq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % cq) * 4;
    const float4 xv = *reinterpret_cast<const float4*>(x + i * 4);
    float4 dv = *reinterpret_cast<const float4*>(dz + i * 4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs = *reinterpret_cast<const float4*>(rstd + c);
    const float4 a = *reinterpret_cast<const float4*>(c1 + c), b = *reinterpret_cast<const float4*>(c2 + c);
    if (act) {
      const float4 sh = *reinterpret_cast<const float4*>(shift + c);
      dv.x *= act_d(act, fmaf(rs.x, xv.x, sh.x)); dv.y *= act_d(act, fmaf(rs.y, xv.y, sh.y));
      dv.z *= act_d(act, fmaf(rs.z, xv.z, sh.z)); dv.w *= act_d(act, fmaf(rs.w, xv.w, sh.w));
    }
    float4 o;
    o.x = rs.x * (dv.x - a.x - (xv.x - mu.x) * rs.x * b.x); o.y = rs.y * (dv.y - a.y - (xv.y - mu.y) * rs.y * b.y);
    o.z = rs.z * (dv.z - a.z - (xv.z - mu.z) * rs.z * b.z); o.w = rs.w * (dv.w - a.w - (xv.w - mu.w) * rs.w * b.w);
    *reinterpret_cast<float4*>(dx + i * 4) = o;
  }
}

// act: 0 none, 1 leaky-relu(0.2), 2 relu, 5 relu6 (vp::Act numbering)
__device__ __forceinline__ float act_f(int act, float v) {
  if (act == ACT_LRELU || act == ACT_LEAKY) return v >= 0.f ? v : 0.2f * v;
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}
// derivative as a function of the OUTPUT y = act(v)
__device__ __forceinline__ float act_d(int act, float y) {
  if (act == ACT_LRELU || act == ACT_LEAKY) return y >= 0.f ? 1.f : 0.2f;
  if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == ACT_RELU6) return (y > 0.f && y < 6.f) ? 1.f : 0.f;
  return 1.f;
}

// y = act(scale[c] * x + shift[c]) * mask   (scale / shift / mask optional)
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mask, const float* __restrict__ add, size_t P, int C, int act,
                                                         float* __restrict__ y) {
  const int cq = C >> 2;
  const size_t total = P * cq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % cq) * 4;
    float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    if (scale) {
      const float4 s = *reinterpret_cast<const float4*>(scale + c), b = *reinterpret_cast<const float4*>(shift + c);
      v.x = fmaf(s.x, v.x, b.x); v.y = fmaf(s.y, v.y, b.y); v.z = fmaf(s.z, v.z, b.z); v.w = fmaf(s.w, v.w, b.w);
    }
    v.x = act_f(act, v.x); v.y = act_f(act, v.y); v.z = act_f(act, v.z); v.w = act_f(act, v.w);
    if (mask) { const float4 m = *reinterpret_cast<const float4*>(mask + i * 4); v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w; }
    if (add) { const float4 r = *reinterpret_cast<const float4*>(add + i * 4); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }   // residual branch
    *reinterpret_cast<float4*>(y + i * 4) = v;
  }
}

// dx = dy * mask * act'(y_pre_mask)   where ya = act output BEFORE the mask
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ya, const float* __restrict__ mask, size_t n4,
                                                      int act, float* __restrict__ dx) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 d = *reinterpret_cast<const float4*>(dy + i * 4);
    const float4 y = *reinterpret_cast<const float4*>(ya + i * 4);
    if (mask) { const float4 m = *reinterpret_cast<const float4*>(mask + i * 4); d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w; }
    d.x *= act_d(act, y.x); d.y *= act_d(act, y.y); d.z *= act_d(act, y.z); d.w *= act_d(act, y.w);
    *reinterpret_cast<float4*>(dx + i * 4) = d;
  }
}

// ------------------------------------------------------------------------------------------------
// depthwise 7x3 weight gradient: dW[kh*3+kw][c] = sum_{b,h,w} x[b, h+kh-3, w+kw-1, c] * dy[b,h,w,c]
// grid (G, ceil(C/64)); block = 4 waves x 64 channels; a wave walks work items (b, w, row segment of HS rows) down the column.  The
// 7 x 3 window of x lives in a ring of U = 10 row slots (the 7 rows of the current output row + PF = 3 rows already requested), the
// row loop is unrolled by U so every slot index is compile-time (no register shuffling), loads are branch-free (clamped address,
// multiplied by 0 outside the image) - four 256-byte loads per row and wave, PF rows in flight.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv7x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int H, int W, int C, int HS,
                                                              int nseg, float* __restrict__ partial /*[G][21][C]*/) {
  constexpr int U = 10, PF = 3;
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float acc[21];
#pragma unroll
  for (int k = 0; k < 21; ++k) acc[k] = 0.f;
  const int nitem = B * W * nseg;
  if (c < C) {
    for (int item = blockIdx.x * 4 + rl; item < nitem; item += gridDim.x * 4) {
      const int seg = item % nseg, col = item / nseg;
      const int b = col / W, w = col - b * W;
      const int h0 = seg * HS, hs = h0 + HS <= H ? HS : H - h0;
      const bool okl = w > 0, okr = w + 1 < W;
      const float* xb = x + ((size_t)b * H * W) * C + c;
      const float* db = dy + (((size_t)b * H + h0) * W + w) * C + c;
      const size_t ol = (size_t)(okl ? w - 1 : w) * C, oc = (size_t)w * C, orr = (size_t)(okr ? w + 1 : w) * C, rowstride = (size_t)W * C;
      float ring[U][3], dring[U];
      auto loadx = [&](int xrel, float (&r)[3]) {          // x row h0 - 3 + xrel
        int ih = h0 - 3 + xrel;
        const bool ok = (unsigned)ih < (unsigned)H;
        ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
        const float* rp = xb + (size_t)ih * rowstride;
        const float m = ok ? 1.f : 0.f;
        r[0] = rp[ol] * (okl ? m : 0.f); r[1] = rp[oc] * m; r[2] = rp[orr] * (okr ? m : 0.f);
      };
      auto loadd = [&](int rel) {                          // dy row h0 + rel (0 beyond the segment)
        const int rr = rel < hs ? rel : hs - 1;
        return db[(size_t)rr * rowstride] * (rel < hs ? 1.f : 0.f);
      };
#pragma unroll
      for (int j = 0; j < 6 + PF; ++j) loadx(j, ring[j]);
#pragma unroll
      for (int j = 0; j < PF; ++j) dring[j] = loadd(j);
      for (int k = 0; k < hs; k += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rel = k + u;
          loadx(rel + 6 + PF, ring[(u + 6 + PF) % U]);
          dring[(u + PF) % U] = loadd(rel + PF);
          const float d = dring[u];
#pragma unroll
          for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc[r * 3 + kw] = fmaf(ring[(u + r) % U][kw], d, acc[r * 3 + kw]);
        }
      }
    }
  }
  __shared__ float sm[4][64];
  for (int k = 0; k < 21; ++k) {
    __syncthreads();
    sm[rl][threadIdx.x & 63] = acc[k];
    __syncthreads();
    if (rl == 0 && c < C) partial[((size_t)blockIdx.x * 21 + k) * C + c] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
  }
}

// out[i] = sum_g partial[g][i]; a block = 64 outputs x 4 lanes over g
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ partial, int G, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int lane = threadIdx.x >> 6;
  float s = 0.f;
  if (i < n) {
    int g = lane;
    for (; g + 12 < G; g += 16) {
      const float a = partial[(size_t)g * n + i], b = partial[(size_t)(g + 4) * n + i], c = partial[(size_t)(g + 8) * n + i], d = partial[(size_t)(g + 12) * n + i];
      s += (a + b) + (c + d);
    }
    for (; g < G; g += 4) s += partial[(size_t)g * n + i];
  }
  __shared__ float sm[256];
  sm[threadIdx.x] = s;
  __syncthreads();
  if (lane == 0 && i < n) out[i] = (sm[threadIdx.x] + sm[threadIdx.x + 64]) + (sm[threadIdx.x + 128] + sm[threadIdx.x + 192]);
}

// ------------------------------------------------------------------------------------------------
// SAME max-pool backward (gather form): an input element collects dy of every window that contains it and whose FIRST maximum
// (row-major window scan, as TF's MaxPoolGrad) it is
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_same_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int B, int H,
                                                               int W, int C, int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo) {
  const int cq = C >> 2;
  const size_t total = (size_t)B * H * W * cq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % cq) * 4;
    size_t t = i / cq;
    const int iw = (int)(t % W); t /= W;
    const int ih = (int)(t % H);
    const int b = (int)(t / H);
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    // windows (oh, ow) with oh*sh - pt <= ih < oh*sh - pt + kh
    int oh0 = (ih + pt - kh + sh) / sh; if (oh0 < 0) oh0 = 0;      // ceil((ih + pt - kh + 1) / sh)
    int oh1 = (ih + pt) / sh; if (oh1 > Ho - 1) oh1 = Ho - 1;
    int ow0 = (iw + pl - kw + sw) / sw; if (ow0 < 0) ow0 = 0;
    int ow1 = (iw + pl) / sw; if (ow1 > Wo - 1) ow1 = Wo - 1;
    for (int oh = oh0; oh <= oh1; ++oh)
      for (int ow = ow0; ow <= ow1; ++ow) {
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {-1, -1, -1, -1};
        for (int a = 0; a < kh; ++a) {
          const int yh = oh * sh + a - pt;
          if ((unsigned)yh >= (unsigned)H) continue;
          for (int bb = 0; bb < kw; ++bb) {
            const int yw = ow * sw + bb - pl;
            if ((unsigned)yw >= (unsigned)W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * H + yh) * W + yw) * C + c4);
            const float vv[4] = {v.x, v.y, v.z, v.w};
            const int id = yh * W + yw;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (vv[e] > best[e]) { best[e] = vv[e]; arg[e] = id; }
          }
        }
        const float4 d = *reinterpret_cast<const float4*>(dy + (((size_t)b * Ho + oh) * Wo + ow) * C + c4);
        const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) if (arg[e] == ih * W + iw) g[e] += dd[e];
      }
    *reinterpret_cast<float4*>(dx + i * 4) = make_float4(g[0], g[1], g[2], g[3]);
  }
}

// stem 9x5 stride (1,2) SAME on [B,H,W,1]: col[p][kh*5+kw] (48 columns, the last 3 zero), p = (b, oh, ow)
__global__ __launch_bounds__(256) void im2col_9x5_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int H, int W, int Wo, int pt, int pl) {
  const size_t total = (size_t)B * H * Wo * 48;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % 48);
    size_t t = i / 48;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % H);
    const int b = (int)(t / H);
    float v = 0.f;
    if (k < 45) {
      const int ih = oh + k / 5 - pt, iw = ow * 2 + k % 5 - pl;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[((size_t)b * H + ih) * W + iw];
    }
    col[i] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// GRU (tf.contrib.rnn.GRUCell under dynamic_rnn, 256 units), training form: gru_device.h (gru_fwd_kernel<true>, gru_bwd_kernel).
//   forward : saves r, u, c and h_prev per step (what the backward needs); out = h past-the-end zero, state frozen
//   backward: d_ag [B,T,512] (gate pre-activations), d_ac [B,T,256] (candidate pre-activation) from d_out; the weight / input
//             gradients are GEMMs over these (train_engine.py)
// ------------------------------------------------------------------------------------------------
// vertex-space loss (bfmnet.py:215-262) on D[b,t,j] = face_shape(true) - face_shape(pred) (a GEMM of the expression difference):
//   loss = (1/B) sum_b [ sum_t fm[b,t] sum_j |D[b,t,j]| vm[j]  +  sum_{t<T-1} vd[b,t] sum_j |D[b,t+1,j] - D[b,t,j]| vm[j] ]
// thread = (b, j) walks t; writes gD = d loss / d D and one f64 partial per block
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vertex_loss_kernel(const float* __restrict__ D, const float* __restrict__ vmask, const int* __restrict__ seq_len, int B,
                                                          int T, int J, float* __restrict__ gD, double* __restrict__ partial) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  double acc = 0;
  if (i < (size_t)B * J) {
    const int b = (int)(i / J), j = (int)(i - (size_t)b * J);
    const int n = seq_len[b];
    const float vm = vmask[j], inv = 1.f / (float)B;
    const float* d = D + (size_t)b * T * J + j;
    float* g = gD + (size_t)b * T * J + j;
    float prev = 0.f, gprev = 0.f;                   // gprev: gradient already owed to D[t] by the difference (t-1, t)
    for (int t = 0; t < T; ++t) {
      const float v = d[(size_t)t * J];
      float gt = gprev;
      gprev = 0.f;
      if (t < n) { acc += (double)(fabsf(v) * vm); gt += (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * vm * inv; }
      if (t + 1 < T && t < n - 1) {                  // video term between t and t + 1
        const float w = d[(size_t)(t + 1) * J] - v;
        acc += (double)(fabsf(w) * vm);
        const float s = (w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f)) * vm * inv;
        gt -= s; gprev = s;
      }
      g[(size_t)t * J] = gt;
      prev = v;
    }
    (void)prev;
  }
  __shared__ double sm[256];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] / (double)B;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, double* __restrict__ partial) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += (double)x[i] * x[i];
  __shared__ double sm[256];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

// ------------------------------------------------------------------------------------------------
// tf.clip_by_global_norm + tf.train.AdamOptimizer over a flat arena with the step scalars on the DEVICE (so a captured hipGraph of the
// whole step replays with a new learning rate / step count): g *= clip / max(sqrt(*sumsq), clip), written back, then
// theta -= *lr_t * m / (sqrt(v) + eps)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n4,
                                                        const float* __restrict__ lr_t, const double* __restrict__ sumsq, float clip, float b1, float b2,
                                                        float eps) {
  const float scale = (float)((double)clip / fmax(sqrt(*sumsq), (double)clip)), lr = *lr_t;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 gv = *reinterpret_cast<float4*>(g + i * 4), mv = *reinterpret_cast<float4*>(m + i * 4), vv = *reinterpret_cast<float4*>(v + i * 4),
           pv = *reinterpret_cast<float4*>(p + i * 4);
    float* G = &gv.x; float* M = &mv.x; float* V = &vv.x; float* Pp = &pv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = G[k] * scale;
      G[k] = gg;
      M[k] = b1 * M[k] + (1.f - b1) * gg;
      V[k] = b2 * V[k] + (1.f - b2) * gg * gg;
      Pp[k] -= lr * M[k] / (sqrtf(V[k]) + eps);
    }
    *reinterpret_cast<float4*>(g + i * 4) = gv; *reinterpret_cast<float4*>(m + i * 4) = mv;
    *reinterpret_cast<float4*>(v + i * 4) = vv; *reinterpret_cast<float4*>(p + i * 4) = pv;
  }
}

// moving = decay * moving + factor[i] * batch[i]   (every batch-norm's moving mean / variance in one pass; factor folds 1 - decay and the
// n / (n - 1) of the variance slots)
__global__ __launch_bounds__(256) void moving_update_kernel(float* __restrict__ moving, const float* __restrict__ batch, const float* __restrict__ factor,
                                                            size_t n, float decay) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) moving[i] = decay * moving[i] + factor[i] * batch[i];
}

// g += scale * mask[i] * p[i]  and  partial sums of mask * p^2 (the l2 regulariser's gradient and value over the flat arena)
__global__ __launch_bounds__(256) void l2_reg_kernel(const float* __restrict__ p, const float* __restrict__ mask, float* __restrict__ g, size_t n, float scale,
                                                     double* __restrict__ partial) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float w = p[i] * mask[i];
    g[i] = fmaf(scale, w, g[i]);
    acc += (double)w * w;
  }
  __shared__ double sm[256];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

}  // namespace vp

using namespace vp;

// Scalars of a training step, on the device and in a fixed order (the framework's reductions / sqrt / stack used to do this):
//   sum_f64: out[0] = (add ? add[0] : 0) + scale * sum(partial[0 .. n)), one 256-thread block, float64
__global__ __launch_bounds__(256) void sum_f64_kernel(const double* __restrict__ partial, int n, double scale, const double* __restrict__ add, double* __restrict__ out) {
  __shared__ double sm[256];
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (add ? add[0] : 0.0) + scale * sm[0];
}
//   step_report: out3 = [loss_data + half_l2 * reg, loss_data, sqrt(sumsq)]  (the Loss / data loss / global norm a step returns)
__global__ void step_report_kernel(const double* loss_data, const double* reg, double half_l2, const double* sumsq, double* out3) {
  out3[0] = loss_data[0] + half_l2 * reg[0];
  out3[1] = loss_data[0];
  out3[2] = sqrt(sumsq[0]);
}
//   clip_scale: g *= clip / max(sqrt(sumsq), clip)  (tf.clip_by_global_norm without the optimiser: the gradient-only step)
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, size_t n, const double* __restrict__ sumsq, float clip) {
  const double gn = sqrt(sumsq[0]);
  const float sc = (float)((double)clip / (gn > (double)clip ? gn : (double)clip));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) g[i] *= sc;
}

extern "C" {

// launch shape of chan_sums_kernel for a [pixels, c] tensor: ql = log2 of the lanes across channel quads, nchunk row chunks
static inline void bn_shape(size_t pixels, int c, int& ql, int& nch) {
  const int cq = c / 4;
  ql = 3;
  while (ql < 6 && (1 << ql) < cq) ++ql;
  const size_t rl = 256 >> ql;
  size_t n = pixels / (rl * 8);
  nch = (int)(n < 1 ? 1 : (n > 256 ? 256 : n));
}

// workspace (bytes) of vp_bn_train_fwd / vp_bn_train_bwd for a [pixels, c] tensor
size_t vp_bn_train_workspace_bytes(size_t pixels, int c) {
  int ql, nch;
  bn_shape(pixels, c, ql, nch);
  return (size_t)nch * 2 * c * sizeof(double) + 2 * (size_t)c * sizeof(float) + 256;
}

// tf.contrib.layers.batch_norm(is_training=True, scale=False): batch statistics of x [pixels, c]; y = x * scale + shift normalises
int vp_bn_train_fwd(const float* x, size_t pixels, int c, const float* beta, float eps, float* mean, float* var, float* rstd, float* scale,
                    float* shift, void* workspace, void* stream) {
  if (!x || !beta || !mean || !var || !rstd || !scale || !shift || !workspace || pixels < 1 || c < 4 || c % 4) { set_err("vp_bn_train_fwd: bad argument"); return VP_ERR_ARG; }
  int ql, nch;
  bn_shape(pixels, c, ql, nch);
  hipStream_t st = (hipStream_t)stream;
  ChanSumsArgs a{x, nullptr, nullptr, nullptr, nullptr, pixels, c, ql, 0, (double*)workspace};
  hipLaunchKernelGGL((chan_sums_kernel<0>), dim3(nch, (c / 4 + (1 << ql) - 1) >> ql), dim3(256), 0, st, a);
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((c + 15) / 16), dim3(256), 0, st, (const double*)workspace, nch, pixels, c, beta, eps, mean, var, rstd, scale, shift);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// backward of batch_norm followed by an activation: da = d loss / d act(normalised + beta) -> dx (may alias da), dbeta.  act 0: da is the
// gradient at the batch-norm output itself (shift may be null); otherwise dz = da * act'(rstd * x + shift) is formed on the fly in both
// passes (shift = beta - mean * rstd as vp_bn_train_fwd returned it).
int vp_bn_act_train_bwd(const float* x, const float* da, size_t pixels, int c, const float* mean, const float* rstd, const float* shift, int act,
                        float* dx, float* dbeta, void* workspace, void* stream) {
  if (!x || !da || !mean || !rstd || !dx || !dbeta || !workspace || (act && !shift) || pixels < 1 || c < 4 || c % 4) { set_err("vp_bn_act_train_bwd: bad argument"); return VP_ERR_ARG; }
  int ql, nch;
  bn_shape(pixels, c, ql, nch);
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)workspace;
  float* c1 = (float*)((char*)workspace + (size_t)nch * 2 * c * sizeof(double));
  float* c2 = c1 + c;
  ChanSumsArgs a{x, da, mean, rstd, shift, pixels, c, ql, act, part};
  hipLaunchKernelGGL((chan_sums_kernel<1>), dim3(nch, (c / 4 + (1 << ql) - 1) >> ql), dim3(256), 0, st, a);
  hipLaunchKernelGGL(bn_bwd_finalize2_kernel, dim3((c + 15) / 16), dim3(256), 0, st, (const double*)part, nch, pixels, c, c1, c2, dbeta);
  hipLaunchKernelGGL(bn_bwd_apply2_kernel, dim3(tblk(pixels * (c / 4))), dim3(256), 0, st, x, da, mean, rstd, shift, act, c1, c2, pixels, c, dx);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// the same without a following activation
int vp_bn_train_bwd(const float* x, const float* dz, size_t pixels, int c, const float* mean, const float* rstd, float* dx, float* dbeta,
                    void* workspace, void* stream) {
  return vp_bn_act_train_bwd(x, dz, pixels, c, mean, rstd, nullptr, 0, dx, dbeta, workspace, stream);
}

// y = act(scale[c] * x + shift[c]) * mask   (scale / shift and mask may be null); act: 0 none, 1 leaky-relu(0.2), 2 relu, 5 relu6
int vp_affine_act_fwd(const float* x, const float* scale, const float* shift, const float* mask, size_t pixels, int c, int act, float* y, void* stream) {
  if (!x || !y || (scale && !shift) || pixels < 1 || c < 4 || c % 4) { set_err("vp_affine_act_fwd: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(affine_act_kernel, dim3(tblk(pixels * (c / 4))), dim3(256), 0, (hipStream_t)stream, x, scale, shift, mask, (const float*)nullptr, pixels, c,
                     act, y);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// y = act(scale[c] * x + shift[c]) * mask + add: the same pass with the residual branch of an inverted-residual block added (add [pixels, c])
int vp_affine_act_add_fwd(const float* x, const float* scale, const float* shift, const float* mask, const float* add, size_t pixels, int c, int act,
                          float* y, void* stream) {
  if (!x || !y || !add || (scale && !shift) || pixels < 1 || c < 4 || c % 4) { set_err("vp_affine_act_add_fwd: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(affine_act_kernel, dim3(tblk(pixels * (c / 4))), dim3(256), 0, (hipStream_t)stream, x, scale, shift, mask, add, pixels, c, act, y);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// dx = dy * mask * act'(ya): ya = the activation's output before the mask (n elements, n % 4 == 0)
int vp_act_bwd(const float* dy, const float* ya, const float* mask, size_t n, int act, float* dx, void* stream) {
  if (!dy || !ya || !dx || n < 4 || n % 4) { set_err("vp_act_bwd: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(act_bwd_kernel, dim3(tblk(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, ya, mask, n / 4, act, dx);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// depthwise [7,3] stride 1 SAME without bias / activation (training forward; backward-data = the same call with the taps reversed)
int vp_dwconv7x3_raw(const float* x, const float* w, float* y, int b, int h, int wd, int c, void* stream) {
  if (!x || !w || !y || b < 1 || h < 1 || wd < 1 || c < 4 || c % 4) { set_err("vp_dwconv7x3_raw: bad argument"); return VP_ERR_ARG; }
  VP_HIP_CHECK(launch_dwconv7x3(x, w, nullptr, y, 0, b, h, wd, c, (hipStream_t)stream));
  return VP_OK;
}

// backward-data of the same convolution: dx = dy convolved with the taps reversed (the kernel indexes them backwards, no flipped copy)
int vp_dwconv7x3_bwd_data(const float* dy, const float* w, float* dx, int b, int h, int wd, int c, void* stream) {
  if (!dy || !w || !dx || b < 1 || h < 1 || wd < 1 || c < 4 || c % 4) { set_err("vp_dwconv7x3_bwd_data: bad argument"); return VP_ERR_ARG; }
  VP_HIP_CHECK(launch_dwconv7x3(dy, w, nullptr, dx, 0, b, h, wd, c, (hipStream_t)stream, 1));
  return VP_OK;
}

// work split of dwconv7x3_wgrad_kernel: row segments so that about 2048 (b, w, segment) items exist, G blocks of 4 items each
static inline void dw_wgrad_shape(int b, int h, int wd, int& hs, int& nseg, int& g) {
  const int cols = b * wd;
  nseg = (2048 + cols - 1) / cols;
  const int most = h / 10 > 0 ? h / 10 : 1;
  if (nseg > most) nseg = most;
  hs = (h + nseg - 1) / nseg;
  hs = (hs + 9) / 10 * 10;               // the kernel's row loop is unrolled by 10
  nseg = (h + hs - 1) / hs;
  g = (cols * nseg + 3) / 4;
  if (g > 256) g = 256;
}

size_t vp_dwconv7x3_wgrad_workspace_bytes(int b, int h, int wd, int c) {
  int hs, nseg, g;
  dw_wgrad_shape(b, h, wd, hs, nseg, g);
  return (size_t)g * 21 * c * sizeof(float);
}
// dw [21][c]
int vp_dwconv7x3_wgrad(const float* x, const float* dy, float* dw, int b, int h, int wd, int c, void* workspace, void* stream) {
  if (!x || !dy || !dw || !workspace || b < 1 || h < 1 || wd < 1 || c < 4) { set_err("vp_dwconv7x3_wgrad: bad argument"); return VP_ERR_ARG; }
  int hs, nseg, g;
  dw_wgrad_shape(b, h, wd, hs, nseg, g);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dwconv7x3_wgrad_kernel, dim3(g, (c + 63) / 64), dim3(256), 0, st, x, dy, b, h, wd, c, hs, nseg, (float*)workspace);
  hipLaunchKernelGGL(sum_rows_kernel, dim3((21 * c + 63) / 64), dim3(256), 0, st, (const float*)workspace, g, (size_t)21 * c, dw);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// backward of vp_maxpool_hw (tf.layers.max_pooling2d 'same')
int vp_maxpool_hw_bwd(const float* x, const float* dy, float* dx, int b, int h, int w, int c, int kh, int kw, int sh, int sw, void* stream) {
  if (!x || !dy || !dx || b < 1 || c < 4 || c % 4 || kh < 1 || kw < 1 || sh < 1 || sw < 1) { set_err("vp_maxpool_hw_bwd: bad argument"); return VP_ERR_ARG; }
  const int ho = (h + sh - 1) / sh, wo = (w + sw - 1) / sw;
  int th = (ho - 1) * sh + kh - h; if (th < 0) th = 0;
  int tw = (wo - 1) * sw + kw - w; if (tw < 0) tw = 0;
  hipLaunchKernelGGL(maxpool_same_bwd_kernel, dim3(tblk((size_t)b * h * w * (c / 4))), dim3(256), 0, (hipStream_t)stream, x, dy, dx, b, h, w, c, kh, kw, sh,
                     sw, th / 2, tw / 2, ho, wo);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// col [b*h*wo][48] of the 9x5 stride-(1,2) SAME stem on x [b,h,w,1] (columns 45..47 zero)
int vp_stem_im2col(const float* x, float* col, int b, int h, int w, void* stream) {
  if (!x || !col || b < 1 || h < 1 || w < 2) { set_err("vp_stem_im2col: bad argument"); return VP_ERR_ARG; }
  const int wo = (w + 1) / 2;
  int tw = (wo - 1) * 2 + 5 - w; if (tw < 0) tw = 0;
  hipLaunchKernelGGL(im2col_9x5_kernel, dim3(tblk((size_t)b * h * wo * 48)), dim3(256), 0, (hipStream_t)stream, x, col, b, h, w, wo, 4, tw / 2);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

int vp_gru_train_fwd(const float* xg, const float* xc, const float* whg, const float* whc, const int* seq_len, float* out, float* r, float* u, float* c,
                     float* hprev, int b, int t, void* stream) {
  if (!xg || !xc || !whg || !whc || !seq_len || !out || !r || !u || !c || !hprev || b < 1 || t < 1) { set_err("vp_gru_train_fwd: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(gru_fwd_kernel<true>, dim3(b), dim3(1024), 0, (hipStream_t)stream, xg, xc, whg, whc, seq_len, out, r, u, c, hprev, t);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

int vp_gru_train_bwd(const float* dout, const float* whg_t, const float* whc_t, const int* seq_len, const float* r, const float* u, const float* c,
                     const float* hprev, float* dag, float* dac, int b, int t, void* stream) {
  if (!dout || !whg_t || !whc_t || !seq_len || !r || !u || !c || !hprev || !dag || !dac || b < 1 || t < 1) { set_err("vp_gru_train_bwd: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, dout, whg_t, whc_t, seq_len, r, u, c, hprev, dag, dac, t);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// ---- small glue of the training step that used to run as at::native kernels --------------------------------------
// out[c] = sum over rows of x[r][c] (bias gradients of the dense / GRU layers, bfmnet.py:194-211): one thread per column, rows in
// ascending order (deterministic); rows = B * T (96 .. 768), cols <= 512
__global__ __launch_bounds__(64) void colsum_rows_kernel(const float* __restrict__ x, int rows, int cols, float* __restrict__ out) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  int r = 0;
  for (; r + 4 <= rows; r += 4) {           // four loads in flight, added in row order
    const float a0 = x[(size_t)r * cols + c], a1 = x[(size_t)(r + 1) * cols + c], a2 = x[(size_t)(r + 2) * cols + c], a3 = x[(size_t)(r + 3) * cols + c];
    s += a0; s += a1; s += a2; s += a3;
  }
  for (; r < rows; ++r) s += x[(size_t)r * cols + c];
  out[c] = s;
}
int vp_colsum_f32(const float* x, int rows, int cols, float* out, void* stream) {
  if (!x || !out || rows < 1 || cols < 1) { set_err("vp_colsum_f32: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(colsum_rows_kernel, dim3((cols + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, rows, cols, out);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// The recurrent halves of tf.contrib.rnn.GRUCell's kernels (bfmnet.py:53: gates/kernel [512, 512] = [x | h] rows, candidate/kernel
// [512, 256]) as the four images the recurrence kernels read: rows 256.. as they are (forward) and transposed (backward), one launch
__global__ __launch_bounds__(256) void gru_split_kernel(const float* __restrict__ gk, const float* __restrict__ ck, float* __restrict__ whg,
                                                        float* __restrict__ whc, float* __restrict__ whg_t, float* __restrict__ whc_t) {
  const int i = blockIdx.x * 256 + threadIdx.x;                 // over 256 x (512 + 256)
  if (i >= 256 * 768) return;
  const int r = i / 768, c = i - r * 768;
  if (c < 512) { const float v = gk[(size_t)(256 + r) * 512 + c]; whg[r * 512 + c] = v; whg_t[c * 256 + r] = v; }
  else { const int cc = c - 512; const float v = ck[(size_t)(256 + r) * 256 + cc]; whc[r * 256 + cc] = v; whc_t[cc * 256 + r] = v; }
}
int vp_gru_split_recurrent(const float* gates_kernel, const float* cand_kernel, float* whg, float* whc, float* whg_t, float* whc_t, void* stream) {
  if (!gates_kernel || !cand_kernel || !whg || !whc || !whg_t || !whc_t) { set_err("vp_gru_split_recurrent: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(gru_split_kernel, dim3(768), dim3(256), 0, (hipStream_t)stream, gates_kernel, cand_kernel, whg, whc, whg_t, whc_t);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i];
}
int vp_mul_f32(const float* a, const float* b, float* out, size_t n, void* stream) {
  if (!a || !b || !out || n < 1) { set_err("vp_mul_f32: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(mul_kernel, dim3(tblk(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// out[r][16 + k] += ears[r] * (-2, -2, -2, -4)[k]: tf.pad(ears * [-2, -2, -2, -4], [[0,0],[0,0],[16,44]]) added to the decoder output
// (bfmnet.py:117,210), in place
__global__ __launch_bounds__(256) void add_ears_rows_kernel(float* __restrict__ out, const float* __restrict__ ears, int rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * 4) return;
  const int r = i >> 2, k = i & 3;
  out[(size_t)r * 64 + 16 + k] += ears[r] * (k == 3 ? -4.f : -2.f);
}
int vp_add_ears_f32(float* out, const float* ears, int rows, void* stream) {
  if (!out || !ears || rows < 1) { set_err("vp_add_ears_f32: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(add_ears_rows_kernel, dim3((rows * 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, ears, rows);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// number of f64 partials vp_bfm_vertex_loss / vp_sumsq write (sum them on the host or with one more reduction)
int vp_vertex_loss_partials(int b, int j) { return (int)(((size_t)b * j + 255) / 256); }
int vp_bfm_vertex_loss(const float* d, const float* vmask, const int* seq_len, int b, int t, int j, float* gd, double* partial, void* stream) {
  if (!d || !vmask || !seq_len || !gd || !partial || b < 1 || t < 1 || j < 1) { set_err("vp_bfm_vertex_loss: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(vertex_loss_kernel, dim3(vp_vertex_loss_partials(b, j)), dim3(256), 0, (hipStream_t)stream, d, vmask, seq_len, b, t, j, gd, partial);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

int vp_sumsq_partials(size_t n) { return tblk(n, 1024); }
int vp_sumsq(const float* x, size_t n, double* partial, void* stream) {
  if (!x || !partial || n < 1) { set_err("vp_sumsq: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(sumsq_kernel, dim3(tblk(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, n, partial);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// clip_by_global_norm + AdamOptimizer, step scalars on the device: *sumsq = the squared global norm, *lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t).
// grads are scaled in place (they hold the clipped gradients afterwards); n % 4 == 0 and 16-byte aligned arenas
int vp_adam_tf_clipped(float* params, float* grads, float* m, float* v, size_t n, const float* lr_t, const double* sumsq, float clip, float beta1,
                       float beta2, float eps, void* stream) {
  if (!params || !grads || !m || !v || !lr_t || !sumsq || n < 4 || n % 4 || !(clip > 0)) { set_err("vp_adam_tf_clipped: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(adam_clip_kernel, dim3(tblk(n / 4)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n / 4, lr_t, sumsq, clip, beta1, beta2, eps);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// moving[i] = decay * moving[i] + factor[i] * batch[i]
int vp_moving_update(float* moving, const float* batch, const float* factor, size_t n, float decay, void* stream) {
  if (!moving || !batch || !factor || n < 1) { set_err("vp_moving_update: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(moving_update_kernel, dim3(tblk(n)), dim3(256), 0, (hipStream_t)stream, moving, batch, factor, n, decay);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// l2 regulariser over a flat arena: grads += scale * mask * params; partial[vp_sumsq_partials(n)] (f64) sums to sum(mask * params^2)
int vp_l2_regulariser(const float* params, const float* mask, float* grads, size_t n, float scale, double* partial, void* stream) {
  if (!params || !mask || !grads || !partial || n < 1) { set_err("vp_l2_regulariser: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(l2_reg_kernel, dim3(tblk(n, 1024)), dim3(256), 0, (hipStream_t)stream, params, mask, grads, n, scale, partial);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// the step's scalar arithmetic (see the kernels): all arguments are device pointers, nothing synchronises
int vp_sum_f64(const double* partial, int n, double scale, const double* add, double* out, void* stream) {
  if (!partial || !out || n < 1) { set_err("vp_sum_f64: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(sum_f64_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, scale, add, out);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}
int vp_bfm_step_report(const double* loss_data, const double* reg, double half_l2, const double* sumsq, double* out3, void* stream) {
  if (!loss_data || !reg || !sumsq || !out3) { set_err("vp_bfm_step_report: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(step_report_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, loss_data, reg, half_l2, sumsq, out3);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}
int vp_clip_scale_f32(float* grads, size_t n, const double* sumsq, float clip, void* stream) {
  if (!grads || !sumsq || n < 1 || !(clip > 0)) { set_err("vp_clip_scale_f32: bad argument"); return VP_ERR_ARG; }
  hipLaunchKernelGGL(clip_scale_kernel, dim3(tblk(n, 1024)), dim3(256), 0, (hipStream_t)stream, grads, n, sumsq, clip);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

}  // extern "C"
