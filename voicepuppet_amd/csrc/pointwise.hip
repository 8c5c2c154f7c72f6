// HBM-bound kernels of the PixReferNet step: weight packing, training-mode batch-norm statistics
// (fwd and bwd), input packing, alpha compositing (fwd/bwd), GAN / L1 / perceptual losses, 2x2
// max-pool, TF-style Adam.  All reductions are two-stage with a fixed summation order, so a step
// is bit-reproducible.
#include "pointwise_args.h"
#include "vp_common.h"

namespace vp {

// block-wide sum of NV doubles per thread; result valid in thread 0
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* sm /* >= NV*4 doubles */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) sm[i * 16 + wave] = v[i];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double s = 0;
      for (int w = 0; w < nw; ++w) s += sm[i * 16 + w];
      v[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing (fp32 master, TF layouts HWIO / HWOI) -> [class][row][tap*C + c] in T
// ------------------------------------------------------------------------------------------------
// One thread per 16-byte output piece (E consecutive k of one row: always inside one tap, channels are padded to 8).
// The thread order follows the SOURCE layout: HWIO convolution kernels are contiguous along the packed row (Cout), so
// adjacent threads take adjacent rows; otherwise (HWOI, backward-data views) k is contiguous and adjacent threads take
// adjacent pieces of one row.
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackDesc* __restrict__ descs, const float* __restrict__ master, T* __restrict__ packed) {
  constexpr int E = Elem<T>::E;
  const PackDesc& d = descs[blockIdx.y];
  const int npiece = d.Kpad / E;
  const int total = d.nclass * d.rows_pad * npiece;
  T* dst = packed + d.dst_off;
  const float* src = master + d.src_off;
  const bool row_fast = d.s_row == 1;
  if (row_fast && d.kc == 4 * E) {
    // Round 6: a thread packs the WHOLE 64-byte (row, K chunk) block - four pieces - so adjacent threads (adjacent rows) write one
    // contiguous run; with one piece per thread the four 16-byte pieces of a 64-byte block were written by four threads a whole row
    // sweep apart (stores at a 64-byte stride: a quarter of every sector; 0.9 GB of HBM traffic per step for 0.33 GB of packed weights)
    const int nchk = d.Kpad / d.kc;
    const int total4 = d.nclass * d.rows_pad * nchk;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
      const int row = i % d.rows_pad, t = i / d.rows_pad, chk = t % nchk, cls = t / nchk;
      const int prow = d.perm ? perm_row(row) : row;
      T* o = dst + (((size_t)cls * nchk + chk) * d.rows_pad + prow) * d.kc;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k0 = chk * d.kc + j * E;
        const int tap = k0 / d.C, c0 = k0 - tap * d.C;
        float v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = 0.f;
        if (row < d.rows_real && tap < d.ntaps) {
          const float* p = src + (size_t)d.kh[cls][tap] * d.s_kh + (size_t)d.kw[cls][tap] * d.s_kw + (size_t)row * d.s_row + (size_t)c0 * d.s_ch;
#pragma unroll
          for (int e = 0; e < E; ++e) if (c0 + e < d.C_real) v[e] = p[(size_t)e * d.s_ch];
        }
        if (d.kswap && (j & 1)) {
#pragma unroll
          for (int e = 0; e < E / 2; ++e) { const float tmp = v[e]; v[e] = v[e + E / 2]; v[e + E / 2] = tmp; }
        }
        reinterpret_cast<uint4*>(o)[j] = Elem<T>::pack(v);
      }
    }
    return;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int row, piece, cls;
    if (row_fast) { row = i % d.rows_pad; const int t = i / d.rows_pad; piece = t % npiece; cls = t / npiece; }
    else { piece = i % npiece; const int t = i / npiece; row = t % d.rows_pad; cls = t / d.rows_pad; }
    const int k0 = piece * E;
    const int tap = k0 / d.C, c0 = k0 - tap * d.C;
    float v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = 0.f;
    if (row < d.rows_real && tap < d.ntaps) {
      const float* p = src + (size_t)d.kh[cls][tap] * d.s_kh + (size_t)d.kw[cls][tap] * d.s_kw + (size_t)row * d.s_row + (size_t)c0 * d.s_ch;
#pragma unroll
      for (int e = 0; e < E; ++e) if (c0 + e < d.C_real) v[e] = p[(size_t)e * d.s_ch];
    }
    const int prow = d.perm ? perm_row(row) : row;
    T* o = dst + (((size_t)cls * (d.Kpad / d.kc) + k0 / d.kc) * d.rows_pad + prow) * d.kc + k0 % d.kc;
    if (d.kswap && (piece & 1)) {
#pragma unroll
      for (int e = 0; e < E / 2; ++e) { const float tmp = v[e]; v[e] = v[e + E / 2]; v[e + E / 2] = tmp; }
    }
    *reinterpret_cast<uint4*>(o) = Elem<T>::pack(v);
  }
}

// single-descriptor variant (descriptor passed by value) for the stand-alone op entry points
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_one_kernel(const PackDesc d, const float* __restrict__ master, T* __restrict__ packed) {
  const size_t total = (size_t)d.nclass * d.rows_pad * d.Kpad;
  T* dst = packed + d.dst_off;
  const float* src = master + d.src_off;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % d.Kpad);
    const size_t t = i / d.Kpad;
    const int row = (int)(t % d.rows_pad);
    const int cls = (int)(t / d.rows_pad);
    const int tap = k / d.C, c = k - tap * d.C;
    float v = 0.f;
    if (row < d.rows_real && tap < d.ntaps && c < d.C_real) {
      // select the tap entry without dynamically indexing the by-value argument block
      int kh = 0, kw = 0;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int tt = 0; tt < 16; ++tt)
          if (cc == cls && tt == tap) { kh = d.kh[cc][tt]; kw = d.kw[cc][tt]; }
      v = src[(size_t)kh * d.s_kh + (size_t)kw * d.s_kw + (size_t)row * d.s_row + (size_t)c * d.s_ch];
    }
    int kk = k % d.kc;
    if (d.kswap) {   // odd pieces: halves swapped
      constexpr int E = Elem<T>::E;
      if ((kk / E) & 1) kk = (kk / E) * E + ((kk % E) + E / 2) % E;
    }
    Elem<T>::st(dst + (((size_t)cls * (d.Kpad / d.kc) + k / d.kc) * d.rows_pad + (d.perm ? perm_row(row) : row)) * d.kc + kk, v);
  }
}

// ------------------------------------------------------------------------------------------------
// batch-norm statistics.  MODE 0: sum y, sum y^2.  MODE 1: sum dz, sum dz*zhat.
// grid = (nchunk, G); thread -> (pixel lane, E-channel group); double accumulation.
// ------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const BnArgs a) {
  constexpr int E = Elem<T>::E;
  const int ncg = a.C / E;                       // channel groups (divides 256)
  const int cgi = threadIdx.x % ncg, prow = threadIdx.x / ncg, nprow = 256 / ncg;
  const int grp = blockIdx.y;
  const int per = (a.Pg + a.nchunk - 1) / a.nchunk;
  const int p0 = blockIdx.x * per, p1 = min(a.Pg, p0 + per);
  double s0[E], s1[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { s0[e] = 0; s1[e] = 0; }
  float mu[E], rs[E];
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < E; ++e) { mu[e] = a.mu[grp * a.C + cgi * E + e]; rs[e] = a.rstd[grp * a.C + cgi * E + e]; }
  }
  const T* y = reinterpret_cast<const T*>(a.y) + (size_t)grp * a.Pg * a.C + cgi * E;
  const T* dz = reinterpret_cast<const T*>(a.dz) + (size_t)grp * a.Pg * a.C + cgi * E;
  // four pixel rows per trip, all their loads issued before the first use (the pass is latency-bound otherwise); the sums still
  // run over p in ascending order, i.e. the result is the same bit for bit as a one-row-per-trip loop
  constexpr int U = 4;
  for (int p = p0 + prow; p < p1; p += U * nprow) {
    uint4 ry[U], rd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pu = p + u * nprow;
      const bool ok = pu < p1;
      ry[u] = ok ? *reinterpret_cast<const uint4*>(y + (size_t)pu * a.C) : make_uint4(0, 0, 0, 0);
      if (MODE == 1) rd[u] = ok ? *reinterpret_cast<const uint4*>(dz + (size_t)pu * a.C) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (p + u * nprow >= p1) break;
      float fy[E];
      Elem<T>::unpack(ry[u], fy);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) { s0[e] += fy[e]; s1[e] += (double)fy[e] * fy[e]; }
      } else {
        float fd[E];
        Elem<T>::unpack(rd[u], fd);
#pragma unroll
        for (int e = 0; e < E; ++e) { s0[e] += fd[e]; s1[e] += (double)fd[e] * ((fy[e] - mu[e]) * rs[e]); }
      }
    }
  }
  // reduce over the pixel lanes that share a channel group.  Fewer than 64 groups (<= 256 channels in bf16): a wave holds 64 / ncg lanes
  // per group - folded with xor shuffles first, then the four waves through LDS.  (The one-thread-per-group serial walk over all 256 /
  // ncg rows that this replaces was most of the kernel for narrow tensors: the 8-channel gradient of decoder_1, 4096 dependent LDS
  // reads, took 90 us for 33 MB.)
  __shared__ double sm[256 * 2];
  double* out = a.partial + ((size_t)(grp * a.nchunk + blockIdx.x) * 2) * a.C;
  if (ncg < 64) {
    for (int off = 32; off >= ncg; off >>= 1) {
#pragma unroll
      for (int e = 0; e < E; ++e) { s0[e] += __shfl_xor(s0[e], off); s1[e] += __shfl_xor(s1[e], off); }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // (per element e as below: sm is [2][4 waves][ncg])
#pragma unroll
    for (int e = 0; e < E; ++e) {
      __syncthreads();
      if (lane < ncg) { sm[wv * ncg + lane] = s0[e]; sm[256 + wv * ncg + lane] = s1[e]; }
      __syncthreads();
      if ((int)threadIdx.x < ncg) {
        const int c = threadIdx.x;
        out[c * E + e] = (sm[c] + sm[ncg + c]) + (sm[2 * ncg + c] + sm[3 * ncg + c]);
        out[a.C + c * E + e] = (sm[256 + c] + sm[256 + ncg + c]) + (sm[256 + 2 * ncg + c] + sm[256 + 3 * ncg + c]);
      }
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    __syncthreads();
    sm[threadIdx.x] = s0[e]; sm[256 + threadIdx.x] = s1[e];
    __syncthreads();
    if (prow == 0) {
      double t0 = 0, t1 = 0;
      for (int r = 0; r < nprow; ++r) { t0 += sm[r * ncg + cgi]; t1 += sm[256 + r * ncg + cgi]; }
      out[cgi * E + e] = t0; out[a.C + cgi * E + e] = t1;
    }
  }
}

// sum of the (s0, s1) partials of (grp, c) over the pixel chunks: one wave, lanes stride over chunks
__device__ __forceinline__ void chunk_sum(const BnArgs& a, int grp, int c, int lane, double& s0, double& s1) {
  // eight chunk rows (sixteen loads) in flight, added in the same order as a one-at-a-time loop: the finalize kernels sit on the
  // step's dependency chain (conv -> finalize -> activation -> next conv) and were pure load latency (13 us for 2048 chunks)
  s0 = 0; s1 = 0;
  for (int k = lane; k < a.nchunk; k += 64 * 8) {
    double x0[8], x1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = k + 64 * u;
      const bool ok = kk < a.nchunk;
      const double* p = a.partial + ((size_t)(grp * a.nchunk + (ok ? kk : k)) * 2) * a.C;
      x0[u] = p[c]; x1[u] = p[a.C + c];
      if (!ok) { x0[u] = 0; x1[u] = 0; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += x0[u]; s1 += x1[u]; }
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
}

// one wave per (group, channel): mean / biased variance -> affine (pixrefer.py:99-101, eps in sqrt)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const BnArgs a) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= a.G * a.C) return;
  const int grp = i / a.C, c = i - grp * a.C;
  double s0, s1;
  chunk_sum(a, grp, c, lane, s0, s1);
  if (lane != 0) return;
  const double mean = s0 / a.Pg;
  double var = s1 / a.Pg - mean * mean;
  if (var < 0) var = 0;
  const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  const float g = a.gamma[c], b = a.beta[c];
  // zero variance (N=1 at the 1x1 bottleneck): y - mean == 0 exactly, so z == beta exactly
  const float sc = (var == 0.0) ? 0.f : g * rstd;
  a.aff_a[i] = sc;
  a.aff_b[i] = (var == 0.0) ? b : (float)((double)b - mean * (double)sc);
  a.mu[i] = (float)mean;
  a.rstd[i] = rstd;
}

// bwd, one wave per channel: c1 = mean(dz), c2 = mean(dz*zhat) per group; dgamma/dbeta summed over groups
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const BnArgs a) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= a.C) return;
  double dg = 0, db = 0;
  for (int grp = 0; grp < a.G; ++grp) {
    double s0, s1;
    chunk_sum(a, grp, c, lane, s0, s1);
    if (a.raw) s1 = (double)a.rstd[grp * a.C + c] * (s1 - (double)a.mu[grp * a.C + c] * s0);      // sum dz * y -> sum dz * zhat
    if (lane == 0) {
      a.c1[grp * a.C + c] = (float)(s0 / a.Pg);
      a.c2[grp * a.C + c] = (float)(s1 / a.Pg);
    }
    db += s0; dg += s1;
  }
  if (lane == 0 && a.dgamma) {
    a.dgamma[c] = (float)dg + (a.accumulate ? a.dgamma[c] : 0.f);
    a.dbeta[c] = (float)db + (a.accumulate ? a.dbeta[c] : 0.f);
  }
  if (lane == 0 && a.dbias_zero) a.dbias_zero[c] = 0.f;
}

// dy = gamma*rstd*(dz - c1 - zhat*c2)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnArgs a) {
  constexpr int E = Elem<T>::E;
  const int ncg = a.C / E;
  const size_t total = (size_t)a.G * a.Pg * ncg;
  const T* y = reinterpret_cast<const T*>(a.y);
  const T* dz = reinterpret_cast<const T*>(a.dz);
  T* dy = reinterpret_cast<T*>(a.dy);
  // ncg divides 256: a thread stays on one channel group; its five per-channel coefficients change only with the BN group
  // (any other channel count: reloaded per item)
  const bool fixed = 256 % ncg == 0;
  int cur = -1;
  float frs[E], fmu[E], fgm[E], fc1[E], fc2[E];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c0 = (int)(i % ncg) * E;
    const size_t pix = i / ncg;
    const int grp = (int)(pix / a.Pg);
    if (grp != cur || !fixed) {
      cur = grp;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int gi = grp * a.C + c0 + e;
        frs[e] = a.rstd[gi]; fmu[e] = a.mu[gi]; fc1[e] = a.c1[gi]; fc2[e] = a.c2[gi]; fgm[e] = a.gamma[c0 + e];
      }
    }
    float fy[E], fd[E];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + pix * a.C + c0), fy);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(dz + pix * a.C + c0), fd);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float zh = (fy[e] - fmu[e]) * frs[e];
      fd[e] = fgm[e] * frs[e] * (fd[e] - fc1[e] - zh * fc2[e]);
    }
    *reinterpret_cast<uint4*>(dy + pix * a.C + c0) = Elem<T>::pack(fd);
  }
}

// ------------------------------------------------------------------------------------------------
// Small tensors (the 1x1 .. 16x16 bottleneck of the generator): statistics, finalize and the per-pixel pass in ONE launch
// per direction - these layers are launch-latency bound, three dependent kernels each.  One block per E-channel group walks
// all pixels of a BN group twice (the tensor slice is a few tens of KB: second pass hits L2).
//   forward : sum / sum^2 -> scale, shift, mean, rstd -> x~ = act(scale*y + shift) for the activations the consumers need
//   backward: sum dz, sum dz*zhat -> c1, c2 (+ dgamma, dbeta over all groups) -> dy = gamma*rstd*(dz - c1 - zhat*c2) in place
// ------------------------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void block_sum2(double (&s0)[E], double (&s1)[E], double* sm /* [4][2E] */) {
#pragma unroll
  for (int e = 0; e < E; ++e) { s0[e] = wave_sum(s0[e]); s1[e] = wave_sum(s1[e]); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) { sm[wave * 2 * E + e] = s0[e]; sm[wave * 2 * E + E + e] = s1[e]; }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    s0[e] = sm[e] + sm[2 * E + e] + sm[4 * E + e] + sm[6 * E + e];
    s1[e] = sm[E + e] + sm[3 * E + e] + sm[5 * E + e] + sm[7 * E + e];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(const BnArgs a, T* __restrict__ out_lrelu, T* __restrict__ out_relu) {
  constexpr int E = Elem<T>::E;
  __shared__ double sm[4 * 2 * E];
  const int c0 = blockIdx.x * E, grp = blockIdx.y;
  const T* y = reinterpret_cast<const T*>(a.y) + (size_t)grp * a.Pg * a.C + c0;
  double s0[E], s1[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { s0[e] = 0; s1[e] = 0; }
  for (int p = threadIdx.x; p < a.Pg; p += 256) {
    float f[E];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + (size_t)p * a.C), f);
#pragma unroll
    for (int e = 0; e < E; ++e) { s0[e] += f[e]; s1[e] += (double)f[e] * f[e]; }
  }
  block_sum2<E>(s0, s1, sm);
  float sc[E], sh[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const double mean = s0[e] / a.Pg;
    double var = s1[e] / a.Pg - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float g = a.gamma[c0 + e], b = a.beta[c0 + e];
    sc[e] = (var == 0.0) ? 0.f : g * rstd;                               // same zero-variance rule as bn_finalize_kernel
    sh[e] = (var == 0.0) ? b : (float)((double)b - mean * (double)sc[e]);
    if (threadIdx.x == 0) {
      const int i = grp * a.C + c0 + e;
      a.aff_a[i] = sc[e]; a.aff_b[i] = sh[e]; a.mu[i] = (float)mean; a.rstd[i] = rstd;
    }
  }
  if (!out_lrelu && !out_relu) return;
  const size_t base = (size_t)grp * a.Pg * a.C + c0;
  for (int p = threadIdx.x; p < a.Pg; p += 256) {
    float f[E], o[E];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + (size_t)p * a.C), f);
#pragma unroll
    for (int e = 0; e < E; ++e) f[e] = fmaf(sc[e], f[e], sh[e]);
    if (out_lrelu) {
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = act_apply(ACT_LRELU, f[e]);
      *reinterpret_cast<uint4*>(out_lrelu + base + (size_t)p * a.C) = Elem<T>::pack(o);
    }
    if (out_relu) {
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = act_apply(ACT_RELU, f[e]);
      *reinterpret_cast<uint4*>(out_relu + base + (size_t)p * a.C) = Elem<T>::pack(o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(const BnArgs a) {
  constexpr int E = Elem<T>::E;
  __shared__ double sm[4 * 2 * E];
  const int c0 = blockIdx.x * E;
  double dg[E], db[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { dg[e] = 0; db[e] = 0; }
  for (int grp = 0; grp < a.G; ++grp) {
    const size_t base = (size_t)grp * a.Pg * a.C + c0;
    const T* y = reinterpret_cast<const T*>(a.y) + base;
    const T* dz = reinterpret_cast<const T*>(a.dz) + base;
    T* dy = reinterpret_cast<T*>(a.dy) + base;
    float mu[E], rs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { mu[e] = a.mu[grp * a.C + c0 + e]; rs[e] = a.rstd[grp * a.C + c0 + e]; }
    double s0[E], s1[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { s0[e] = 0; s1[e] = 0; }
    for (int p = threadIdx.x; p < a.Pg; p += 256) {
      float fy[E], fd[E];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + (size_t)p * a.C), fy);
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(dz + (size_t)p * a.C), fd);
#pragma unroll
      for (int e = 0; e < E; ++e) { s0[e] += fd[e]; s1[e] += (double)fd[e] * ((fy[e] - mu[e]) * rs[e]); }
    }
    block_sum2<E>(s0, s1, sm);
    float c1[E], c2[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      c1[e] = (float)(s0[e] / a.Pg); c2[e] = (float)(s1[e] / a.Pg);
      db[e] += s0[e]; dg[e] += s1[e];
      if (threadIdx.x == 0) { a.c1[grp * a.C + c0 + e] = c1[e]; a.c2[grp * a.C + c0 + e] = c2[e]; }
    }
    for (int p = threadIdx.x; p < a.Pg; p += 256) {
      float fy[E], fd[E];
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + (size_t)p * a.C), fy);
      Elem<T>::unpack(*reinterpret_cast<const uint4*>(dz + (size_t)p * a.C), fd);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float zh = (fy[e] - mu[e]) * rs[e];
        fd[e] = a.gamma[c0 + e] * rs[e] * (fd[e] - c1[e] - zh * c2[e]);
      }
      *reinterpret_cast<uint4*>(dy + (size_t)p * a.C) = Elem<T>::pack(fd);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && a.dgamma) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      a.dgamma[c0 + e] = (float)dg[e] + (a.accumulate ? a.dgamma[c0 + e] : 0.f);
      a.dbeta[c0 + e] = (float)db[e] + (a.accumulate ? a.dbeta[c0 + e] : 0.f);
    }
  }
  if (threadIdx.x == 0 && a.dbias_zero) {
#pragma unroll
    for (int e = 0; e < E; ++e) a.dbias_zero[c0 + e] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// materialise the activated tensors the consumers read: x~ = act(scale*y + shift) (batch-norm affine of
// the producer, per BN group).  One read of y, one write per needed activation.  The MFMA kernels then
// move plain bytes (LDS-DMA) instead of re-doing this per tap and per consumer.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void act_apply_kernel(const T* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh,
                                                        int C, int Pg, size_t npix, T* __restrict__ out_lrelu, T* __restrict__ out_relu) {
  constexpr int E = Elem<T>::E;
  const int ncg = C / E;
  const size_t total = npix * ncg;
  // ncg divides 256: a thread stays on one channel group for all its items, its scale / shift change only with the BN group
  // (any other channel count: reloaded per item)
  const bool fixed = 256 % ncg == 0;
  int cur = -1;
  float fsc[E], fsh[E];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % ncg);
    const size_t pix = i / ncg;
    float f[E], o[E];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(y + pix * C + cg * E), f);
    if (sc) {
      const int grp = (int)(pix / Pg);
      if (grp != cur || !fixed) {
        cur = grp;
#pragma unroll
        for (int e = 0; e < E; ++e) { fsc[e] = sc[grp * C + cg * E + e]; fsh[e] = sh[grp * C + cg * E + e]; }
      }
#pragma unroll
      for (int e = 0; e < E; ++e) f[e] = fmaf(fsc[e], f[e], fsh[e]);
    }
    if (out_lrelu) {
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = act_apply(ACT_LRELU, f[e]);
      *reinterpret_cast<uint4*>(out_lrelu + pix * C + cg * E) = Elem<T>::pack(o);
    }
    if (out_relu) {
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = act_apply(ACT_RELU, f[e]);
      *reinterpret_cast<uint4*>(out_relu + pix * C + cg * E) = Elem<T>::pack(o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// "GEMM over taps" helpers for the one-channel stride-1 conv (see TapArgs)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tap_gather_kernel(TapArgs a) {
  const size_t total = (size_t)a.N * a.Hout * a.Wout;
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= total) return;
  const int j = (int)(p % a.Wout), i = (int)((p / a.Wout) % a.Hout), n = (int)(p / ((size_t)a.Wout * a.Hout));
  float acc = 0.f;
  for (int kh = 0; kh < a.ks; ++kh) {
    const int qi = i + kh - a.pad;
    if (qi < 0 || qi >= a.Hin) continue;
    for (int kw = 0; kw < a.ks; ++kw) {
      const int qj = j + kw - a.pad;
      if (qj < 0 || qj >= a.Win) continue;
      acc += a.S[(((size_t)n * a.Hin + qi) * a.Win + qj) * 16 + kh * a.ks + kw];
    }
  }
  a.y[p] = acc + a.bias[0];
}

template <typename T>
__global__ __launch_bounds__(256) void tap_spread_kernel(TapArgs a) {
  constexpr int E = Elem<T>::E;
  const size_t total = (size_t)a.N * a.Hin * a.Win;
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= total) return;
  const int qj = (int)(q % a.Win), qi = (int)((q / a.Win) % a.Hin), n = (int)(q / ((size_t)a.Win * a.Hin));
  const T* dy = (const T*)a.dy;
  float f[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int i = qi - t / 4 + a.pad, j = qj - t % 4 + a.pad;       // ks == 4
    const bool ok = i >= 0 && i < a.Hout && j >= 0 && j < a.Wout;
    f[t] = ok ? Elem<T>::ld(dy + (((size_t)n * a.Hout + i) * a.Wout + j) * a.ld_dy) : 0.f;
  }
  T* o = (T*)a.dyS + q * 16;
#pragma unroll
  for (int e = 0; e < 16; e += E) {
    float g[E];
#pragma unroll
    for (int k = 0; k < E; ++k) g[k] = f[e + k];
    *reinterpret_cast<uint4*>(o + e) = Elem<T>::pack(g);
  }
}

// ------------------------------------------------------------------------------------------------
// input packing: [0,1] -> [-1,1] (pixrefer.py:373-375), channel padding to 8, the real halves of the
// discriminator and VGG batches (pixrefer.py:295-306, 321)
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&f)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&f)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float (&f)[8]) {
  *reinterpret_cast<uint4*>(p) = Elem<bf16>::pack(f);
}

template <typename T>
__global__ __launch_bounds__(256) void pack_inputs_kernel(const PackInputsArgs a) {
  const size_t total = (size_t)a.N * a.HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    float in[6], fg[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { in[c] = a.inputs[i * 6 + c] * 2.f - 1.f; fg[c] = c < a.fg_c ? a.fg_inputs[i * a.fg_c + c] * 2.f - 1.f : -1.f; }
    { const float f[8] = {in[0], in[1], in[2], in[3], in[4], in[5], 0.f, 0.f}; store8<T>(reinterpret_cast<T*>(a.gin) + i * 8, f); }
    { const float f[8] = {fg[0], fg[1], fg[2], 0.f, 0.f, 0.f, 0.f, 0.f}; store8<T>(reinterpret_cast<T*>(a.gfg) + i * 8, f); }
    if (a.train) {
      T* din = reinterpret_cast<T*>(a.din);
      { const float f[8] = {in[3], in[4], in[5], fg[3], fg[4], fg[5], 0.f, 0.f}; store8<T>(din + i * 8, f); }
      { const float f[8] = {in[0], in[1], in[2], fg[0], fg[1], fg[2], 0.f, 0.f}; store8<T>(din + (total + i) * 8, f); }
      { const float f[8] = {in[3], in[4], in[5], 0.f, 0.f, 0.f, 0.f, 0.f}; store8<T>(din + (2 * total + i) * 8, f); }
      { const float f[8] = {fg[3], fg[4], fg[5], 0.f, 0.f, 0.f, 0.f, 0.f}; store8<T>(reinterpret_cast<T*>(a.vin) + i * 8, f); }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// frame_pack_kernel: the host input pipeline of generator.py:956-1019 as one kernel.  One thread per output pixel and frame
// role; source coordinate of cv2.resize(INTER_LINEAR): f = (d + 0.5) * (rsize / S) - 0.5, clamped at the crop borders;
// horizontal interpolation first, then vertical, in float32 on data already divided by 255 (ImageLoader: loader.py:85-89).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void resize_coord(int d, int rsize, int S, int& s0, int& s1, float& w1) {
  const float f = (float)(((double)d + 0.5) * ((double)rsize / (double)S) - 0.5);
  int s = (int)floorf(f);
  float w = f - (float)s;
  if (s < 0) { s = 0; w = 0.f; }
  if (s >= rsize - 1) { s = rsize - 1; w = 0.f; }
  s0 = s; s1 = min(s + 1, rsize - 1); w1 = w;
}

__global__ __launch_bounds__(256) void frame_pack_kernel(const FramePackArgs a) {
  const int S = a.S;
  const size_t total = (size_t)a.N * S * S * 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int role = (int)(i & 1);                  // 0: example frame, 1: current frame
    const size_t pix = i >> 1;
    const int x = (int)(pix % S), y = (int)((pix / S) % S), n = (int)(pix / ((size_t)S * S));
    const int* cr = a.crops + (n * 2 + role) * 3;
    const int rx = cr[0], ry = cr[1], rsize = cr[2];
    int y0, y1, x0, x1;
    float wy, wx;
    resize_coord(y, rsize, S, y0, y1, wy);
    resize_coord(x, rsize, S, x0, x1, wx);
    const unsigned char* f = (role ? a.cur : a.ex) + (size_t)n * S * 3 * S * 3;
    float v[3][3];                                  // [panel][rgb]
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const unsigned char* r0 = f + ((size_t)(rx + y0) * 3 * S + p * S + ry) * 3;
      const unsigned char* r1 = f + ((size_t)(rx + y1) * 3 * S + p * S + ry) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int b = 2 - c;                        // BGR -> RGB (cv2.cvtColor, generator.py:981)
        const float a00 = r0[x0 * 3 + b] / 255.0f, a01 = r0[x1 * 3 + b] / 255.0f;
        const float a10 = r1[x0 * 3 + b] / 255.0f, a11 = r1[x1 * 3 + b] / 255.0f;
        const float h0 = a00 * (1.f - wx) + a01 * wx, h1 = a10 * (1.f - wx) + a11 * wx;
        v[p][c] = h0 * (1.f - wy) + h1 * wy;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a.inputs[pix * 6 + role * 3 + c] = v[1][c];
      a.fg_inputs[pix * 6 + role * 3 + c] = v[0][c] * v[2][c];
      if (role) { a.targets[pix * 3 + c] = v[0][c]; a.masks[pix * 3 + c] = v[2][c]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// alpha composite (pixrefer.py:281-286) fused with tanh, the L1 / matte loss partial sums and the
// hand-over of Outputs_FG to the discriminator and VGG batches
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const CompositeArgs a) {
  const size_t total = (size_t)a.N * a.HW;
  double acc[2] = {0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const float4 y = reinterpret_cast<const float4*>(a.y4)[i];
    const float o[4] = {tanhf(y.x), tanhf(y.y), tanhf(y.z), tanhf(y.w)};
    reinterpret_cast<float4*>(a.o4)[i] = make_float4(o[0], o[1], o[2], o[3]);
    const float al = (o[3] + 1.f) * 0.5f;
    float ofg[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tg = a.targets[i * 3 + c] * 2.f - 1.f;
      const float out = o[c] * al + tg * (1.f - al);
      ofg[c] = o[c] * al + al - 1.f;
      a.outputs[i * 3 + c] = out;
      a.outputs_fg[i * 3 + c] = ofg[c];
      if (a.train) {
        acc[0] += fabsf(tg - out);
        acc[1] += fabsf(a.masks[i * 3 + c] - al);
      }
    }
    if (a.train) {
      // whole 8-channel rows with one store each (three 2-byte stores into the middle of a row are partial writes): the fake
      // group's discriminator row keeps its first three channels (the conditioning image pack_inputs put there), read back exactly
      T* d = reinterpret_cast<T*>(a.din) + (2 * total + i) * 8;
      T* v = reinterpret_cast<T*>(a.vin) + (total + i) * 8;
      float head[3];
      if (sizeof(T) == 2) {
        const uint2 h = *reinterpret_cast<const uint2*>(d);
        head[0] = __uint_as_float(h.x << 16); head[1] = __uint_as_float(h.x & 0xffff0000u); head[2] = __uint_as_float(h.y << 16);
      } else {
        const float4 h = *reinterpret_cast<const float4*>(d);
        head[0] = h.x; head[1] = h.y; head[2] = h.z;
      }
      const float fd[8] = {head[0], head[1], head[2], ofg[0], ofg[1], ofg[2], 0.f, 0.f};
      const float fv[8] = {ofg[0], ofg[1], ofg[2], 0.f, 0.f, 0.f, 0.f, 0.f};
      store8<T>(d, fd);
      store8<T>(v, fv);
    }
  }
  if (a.train) {
    __shared__ double sm[64];
    block_sum<2>(acc, sm);
    if (threadIdx.x == 0) { a.partial[blockIdx.x * 2] = acc[0]; a.partial[blockIdx.x * 2 + 1] = acc[1]; }
  }
}

// gradient of Gen_loss w.r.t. the pre-tanh generator output
template <typename T>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const CompositeArgs a) {
  const size_t total = (size_t)a.N * a.HW;
  const float s = a.l1_weight / (float)((double)total * 3.0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const float4 o4 = reinterpret_cast<const float4*>(a.o4)[i];
    const float o[4] = {o4.x, o4.y, o4.z, o4.w};
    const float al = (o[3] + 1.f) * 0.5f;
    const T* dd = reinterpret_cast<const T*>(a.d_din) + i * 8 + 3;
    const T* dv = reinterpret_cast<const T*>(a.d_vin) + i * 8;
    float dout[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float dal = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tg = a.targets[i * 3 + c] * 2.f - 1.f;
      const float out = a.outputs[i * 3 + c];
      const float df = tg - out, dm = a.masks[i * 3 + c] - al;
      const float d_out = -s * ((df > 0.f) - (df < 0.f));
      const float d_al = -s * ((dm > 0.f) - (dm < 0.f));
      const float d_fg = Elem<T>::ld(dd + c) + Elem<T>::ld(dv + c);
      dout[c] = (d_out + d_fg) * al;
      dal += d_out * (o[c] - tg) + d_fg * (o[c] + 1.f) + d_al;
    }
    dout[3] = dal * 0.5f;
#pragma unroll
    for (int c = 0; c < 4; ++c) dout[c] *= (1.f - o[c] * o[c]);
    store8<T>(reinterpret_cast<T*>(a.dy4) + i * 8, dout);
  }
}

// ------------------------------------------------------------------------------------------------
// GAN losses and their seeds (pixrefer.py:334-347).  One block; M = N*30*30 at 256x256.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void gan_loss_kernel(const GanLossArgs a) {
  const float eps = 1e-12f;
  double acc[2] = {0, 0};
  const float invM = 1.f / (float)a.M;
  T* dld = reinterpret_cast<T*>(a.dl_d);
  T* dlg = reinterpret_cast<T*>(a.dl_g);
  for (int i = threadIdx.x; i < a.M; i += 1024) {
    const float p0 = 1.f / (1.f + expf(-a.logits[i]));
    const float p1 = 1.f / (1.f + expf(-a.logits[a.M + i]));
    const float pf = 1.f / (1.f + expf(-a.logits[2 * a.M + i]));
    const float pr = (p0 + p1) * 0.5f;
    a.predict[i] = pr; a.predict[a.M + i] = pf;
    acc[0] += -(logf(pr + eps) * 2.f + logf(1.f - pf + eps));
    acc[1] += -logf(pf + eps);
    const float dpr = -2.f / (pr + eps) * invM * 0.5f;
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f[0] = dpr * p0 * (1.f - p0); store8<T>(dld + (size_t)i * 8, f);
    f[0] = dpr * p1 * (1.f - p1); store8<T>(dld + (size_t)(a.M + i) * 8, f);
    f[0] = 1.f / (1.f - pf + eps) * invM * pf * (1.f - pf); store8<T>(dld + (size_t)(2 * a.M + i) * 8, f);
    f[0] = a.gan_weight * (-1.f / (pf + eps)) * invM * pf * (1.f - pf); store8<T>(dlg + (size_t)i * 8, f);
  }
  __shared__ double sm[64];
  block_sum<2>(acc, sm);
  if (threadIdx.x == 0) { a.losses[0] = (float)(acc[0] / a.M); a.losses[1] = (float)(acc[1] / a.M); }
}

// perceptual loss on conv3_3 (pixrefer.py:321-323) + seed for the fake half (through its relu)
template <typename T>
__global__ __launch_bounds__(256) void perceptual_kernel(const PerceptualArgs a) {
  constexpr int E = Elem<T>::E;
  const T* f = reinterpret_cast<const T*>(a.f3);
  T* df = reinterpret_cast<T*>(a.df3);
  const float s = a.l1_weight / (float)a.half;
  double acc[1] = {0};
  const size_t nv = a.half / E;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
    float fa[E], fb[E], d[E];
    Elem<T>::unpack(reinterpret_cast<const uint4*>(f)[i], fa);
    Elem<T>::unpack(reinterpret_cast<const uint4*>(f + a.half)[i], fb);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float x = fa[e] - fb[e];
      acc[0] += (double)x * x;
      d[e] = (fb[e] > 0.f) ? -s * x : 0.f;
    }
    reinterpret_cast<uint4*>(df)[i] = Elem<T>::pack(d);
  }
  __shared__ double sm[64];
  block_sum<1>(acc, sm);
  if (threadIdx.x == 0) a.partial[blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(256) void loss_final_kernel(const LossFinalArgs a) {
  double v[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < a.n_comp; i += 256) { v[0] += a.comp_partial[2 * i]; v[1] += a.comp_partial[2 * i + 1]; }
  for (int i = threadIdx.x; i < a.n_perc; i += 256) v[2] += a.perc_partial[i];
  __shared__ double sm[64];
  block_sum<3>(v, sm);
  if (threadIdx.x != 0) return;
  const double content = v[2] / 2.0 / a.n_feat;
  const double gl1 = v[0] / a.n_out + v[1] / a.n_out + content;
  a.losses[2] = (float)gl1;
  a.losses[3] = (float)((double)a.losses[1] * a.gan_weight + gl1 * a.l1_weight);
  a.losses[4] = (float)content;
}

// ------------------------------------------------------------------------------------------------
// 2x2/s2 max-pool (vgg_simple.py:143,150) fwd, and bwd fused with the relu' of the producing conv
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
  constexpr int E = Elem<T>::E;
  const int ncg = C / E, Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)B * Ho * Wo * ncg;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % ncg);
    size_t t = i / ncg;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const T* p = x + (((size_t)n * H + 2 * oh) * W + 2 * ow) * C + cg * E;
    float m[E], v[E];
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(p), m);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(p + C), v);
#pragma unroll
    for (int e = 0; e < E; ++e) m[e] = fmaxf(m[e], v[e]);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(p + (size_t)W * C), v);
#pragma unroll
    for (int e = 0; e < E; ++e) m[e] = fmaxf(m[e], v[e]);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(p + (size_t)W * C + C), v);
#pragma unroll
    for (int e = 0; e < E; ++e) m[e] = fmaxf(m[e], v[e]);
    *reinterpret_cast<uint4*>(y + i * E) = Elem<T>::pack(m);
  }
}

// dx = (first arg-max of the window) ? dy : 0, times relu'(x) of the conv that produced x (x is post-relu)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C) {
  constexpr int E = Elem<T>::E;
  const int ncg = C / E, Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)B * Ho * Wo * ncg;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % ncg);
    size_t t = i / ncg;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const size_t base = (((size_t)n * H + 2 * oh) * W + 2 * ow) * C + cg * E;
    const size_t offs[4] = {0, (size_t)C, (size_t)W * C, (size_t)W * C + C};
    float v[4][E], g[E];
#pragma unroll
    for (int k = 0; k < 4; ++k) Elem<T>::unpack(*reinterpret_cast<const uint4*>(x + base + offs[k]), v[k]);
    Elem<T>::unpack(*reinterpret_cast<const uint4*>(dy + i * E), g);
    float o[4][E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      int best = 0;
      float m = v[0][e];
#pragma unroll
      for (int k = 1; k < 4; ++k) if (v[k][e] > m) { m = v[k][e]; best = k; }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][e] = (k == best && m > 0.f) ? g[e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<uint4*>(dx + base + offs[k]) = Elem<T>::pack(o[k]);
  }
}

// relu' on a stored post-relu tensor: d *= (y > 0)   (used where no bwd-data epilogue can do it)
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ y, T* __restrict__ d, size_t nvec) {
  constexpr int E = Elem<T>::E;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float fy[E], fd[E];
    Elem<T>::unpack(reinterpret_cast<const uint4*>(y)[i], fy);
    Elem<T>::unpack(reinterpret_cast<const uint4*>(d)[i], fd);
#pragma unroll
    for (int e = 0; e < E; ++e) fd[e] = fy[e] > 0.f ? fd[e] : 0.f;
    reinterpret_cast<uint4*>(d)[i] = Elem<T>::pack(fd);
  }
}

// ------------------------------------------------------------------------------------------------
// tf.train.AdamOptimizer over a flat arena (pixrefer.py:398,405): theta -= lr_t * m / (sqrt(v) + eps)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_tf_kernel(const AdamArgs a) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
    const float g = a.g[i];
    const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
    a.m[i] = m; a.v[i] = v;
    a.p[i] -= a.lr_t * m / (sqrtf(v) + a.eps);
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 transport of a gradient bucket (data parallel, optional): the f32 arena stays the master copy; a bucket is rounded to bf16
// (nearest even) into a communication buffer, all-reduced as bf16, and written back as f32 times 1 / world.  Halves the bytes on
// xGMI (152 -> 76 MB per step) at the price of one rounding of every gradient element per rank.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_pack_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, size_t n8, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    reinterpret_cast<uint4*>(dst)[i] = Elem<bf16>::pack(f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n8 * 8)) Elem<bf16>::st(dst + n8 * 8 + threadIdx.x, src[n8 * 8 + threadIdx.x]);
}

__global__ __launch_bounds__(256) void grad_unpack_bf16_kernel(const bf16* __restrict__ src, float* __restrict__ dst, size_t n8, size_t n, float scale) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float f[8];
    Elem<bf16>::unpack(reinterpret_cast<const uint4*>(src)[i], f);
    reinterpret_cast<float4*>(dst)[2 * i] = make_float4(f[0] * scale, f[1] * scale, f[2] * scale, f[3] * scale);
    reinterpret_cast<float4*>(dst)[2 * i + 1] = make_float4(f[4] * scale, f[5] * scale, f[6] * scale, f[7] * scale);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n8 * 8)) dst[n8 * 8 + threadIdx.x] = Elem<bf16>::ld(src + n8 * 8 + threadIdx.x) * scale;
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
static inline int nblocks(size_t work, int cap = 2048) {
  size_t b = (work + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b > (size_t)cap ? cap : b);
}

#define VP_DISPATCH(is_bf16, KERNEL, grid, block, stream, ...)                         \
  do {                                                                                 \
    if (is_bf16) hipLaunchKernelGGL((KERNEL<bf16>), grid, block, 0, stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<float>), grid, block, 0, stream, __VA_ARGS__);       \
  } while (0)

// ------------------------------------------------------------------------------------------------
// node fetches of PixReferNet.execute (deprocess, the uint8 frame, Alphas, the :436 quirk of build_inference_op): one thread per
// pixel, every float operation rounded on its own (no contraction) in the order the reference's graph applies them
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fetch_kernel(const FetchArgs a) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < a.npix; p += (size_t)gridDim.x * 256) {
    float v[3];
    if (a.mode == 0 || a.mode == 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = __fmul_rn(__fadd_rn(a.raw3[p * 3 + c], 1.f), 0.5f);
    } else {
      const float al = __fmul_rn(__fadd_rn(a.o4[p * 4 + 3], 1.f), 0.5f);
#pragma unroll
      for (int c = 0; c < 3; ++c)
        v[c] = a.mode == 2 ? al : __fmul_rn(__fadd_rn(__fsub_rn(__fadd_rn(a.fg3[p * 3 + c], al), 1.f), 1.f), 0.5f);
    }
    if (a.mode == 1) {
      unsigned char* o = reinterpret_cast<unsigned char*>(a.dst) + p * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = (unsigned char)(int)__fmul_rn(fminf(fmaxf(v[c], 0.f), 1.f), 255.f);
    } else {
      float* o = reinterpret_cast<float*>(a.dst) + p * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) o[c] = v[c];
    }
  }
}

hipError_t launch_fetch(const FetchArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(fetch_kernel, dim3(nblocks(a.npix, 4096)), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_pack_weights(const PackDesc* d_descs, int ndesc, const float* master, void* packed, int is_bf16, hipStream_t st) {
  dim3 grid(512, ndesc);
  if (is_bf16) hipLaunchKernelGGL((pack_weights_kernel<bf16>), grid, dim3(256), 0, st, d_descs, master, (bf16*)packed);
  else hipLaunchKernelGGL((pack_weights_kernel<float>), grid, dim3(256), 0, st, d_descs, master, (float*)packed);
  return hipGetLastError();
}

hipError_t launch_pack_weights_one(const PackDesc& d, const float* master, void* packed, int is_bf16, hipStream_t st) {
  if (is_bf16) hipLaunchKernelGGL((pack_weights_one_kernel<bf16>), dim3(512), dim3(256), 0, st, d, master, (bf16*)packed);
  else hipLaunchKernelGGL((pack_weights_one_kernel<float>), dim3(512), dim3(256), 0, st, d, master, (float*)packed);
  return hipGetLastError();
}

int bn_nchunk(int Pg, int C, int G, int is_bf16) {
  const int E = is_bf16 ? 8 : 4;
  const int nprow = 256 / (C / E);
  int n = (Pg + nprow * 8 - 1) / (nprow * 8);       // >= 8 pixel rows per thread
  const int cap = 1024 / G;                         // (bn_partial holds 1024 chunk rows)
  if (n > cap) n = cap;
  return n < 1 ? 1 : n;
}

hipError_t launch_bn_stats(const BnArgs& a, int is_bf16, hipStream_t st) {
  dim3 grid(a.nchunk, a.G);
  if (is_bf16) hipLaunchKernelGGL((bn_reduce_kernel<bf16, 0>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((bn_reduce_kernel<float, 0>), grid, dim3(256), 0, st, a);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((a.G * a.C + 3) / 4), dim3(256), 0, st, a);
  return hipGetLastError();
}

bool bn_small(const BnArgs& a) {
  return a.Pg <= 2048;          // (4096 .. 16384 pixels per group measured slower: EXPERIMENTS.md 0.2)
}

hipError_t launch_bn_small_fwd(const BnArgs& a, void* out_lrelu, void* out_relu, int is_bf16, hipStream_t st) {
  const int E = is_bf16 ? 8 : 4;
  if (is_bf16) hipLaunchKernelGGL((bn_small_fwd_kernel<bf16>), dim3(a.C / E, a.G), dim3(256), 0, st, a, (bf16*)out_lrelu, (bf16*)out_relu);
  else hipLaunchKernelGGL((bn_small_fwd_kernel<float>), dim3(a.C / E, a.G), dim3(256), 0, st, a, (float*)out_lrelu, (float*)out_relu);
  return hipGetLastError();
}

hipError_t launch_bn_small_bwd(const BnArgs& a, int is_bf16, hipStream_t st) {
  const int E = is_bf16 ? 8 : 4;
  if (is_bf16) hipLaunchKernelGGL((bn_small_bwd_kernel<bf16>), dim3(a.C / E), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((bn_small_bwd_kernel<float>), dim3(a.C / E), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_bn_finalize(const BnArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((a.G * a.C + 3) / 4), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_bn_bwd(const BnArgs& a, int is_bf16, hipStream_t st) {
  dim3 grid(a.nchunk, a.G);
  if (is_bf16) hipLaunchKernelGGL((bn_reduce_kernel<bf16, 1>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((bn_reduce_kernel<float, 1>), grid, dim3(256), 0, st, a);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((a.C + 3) / 4), dim3(256), 0, st, a);
  const size_t work = (size_t)a.G * a.Pg * (a.C / (is_bf16 ? 8 : 4));
  VP_DISPATCH(is_bf16, bn_bwd_apply_kernel, dim3(nblocks(work)), dim3(256), st, a);
  return hipGetLastError();
}

// the same without the reduce pass: the launch that completed the gradient left the partial rows (sum dz, sum dz * zhat) in a.partial
// (staged_epilogue STATS == 2, igemm_device.h; a.nchunk = that launch's rows per group)
hipError_t launch_bn_bwd_tail(const BnArgs& a, int is_bf16, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((a.C + 3) / 4), dim3(256), 0, st, a);
  const size_t work = (size_t)a.G * a.Pg * (a.C / (is_bf16 ? 8 : 4));
  VP_DISPATCH(is_bf16, bn_bwd_apply_kernel, dim3(nblocks(work)), dim3(256), st, a);
  return hipGetLastError();
}

// out[c] (+)= sum over pixels of x[p][c], c < creal  (bias gradients of the BN-free layers)
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const BnArgs a, int creal, float* out, int accumulate) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= creal) return;
  double s = 0;
  for (int grp = 0; grp < a.G; ++grp) {
    double s0, s1;
    chunk_sum(a, grp, c, lane, s0, s1);
    s += s0;
  }
  if (lane == 0) out[c] = (float)s + (accumulate ? out[c] : 0.f);
}

hipError_t launch_colsum(const BnArgs& a, int creal, float* out, int accumulate, int is_bf16, hipStream_t st) {
  dim3 grid(a.nchunk, a.G);
  if (is_bf16) hipLaunchKernelGGL((bn_reduce_kernel<bf16, 0>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((bn_reduce_kernel<float, 0>), grid, dim3(256), 0, st, a);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((creal + 3) / 4), dim3(256), 0, st, a, creal, out, accumulate);
  return hipGetLastError();
}

hipError_t launch_colsum_tail(const BnArgs& a, int creal, float* out, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((creal + 3) / 4), dim3(256), 0, st, a, creal, out, accumulate);
  return hipGetLastError();
}

hipError_t launch_act_apply(const void* y, const float* sc, const float* sh, int C, int Pg, size_t npix,
                            void* out_lrelu, void* out_relu, int is_bf16, hipStream_t st) {
  const size_t work = npix * (C / (is_bf16 ? 8 : 4));
  if (is_bf16) hipLaunchKernelGGL((act_apply_kernel<bf16>), dim3(nblocks(work)), dim3(256), 0, st, (const bf16*)y, sc, sh, C, Pg, npix, (bf16*)out_lrelu, (bf16*)out_relu);
  else hipLaunchKernelGGL((act_apply_kernel<float>), dim3(nblocks(work)), dim3(256), 0, st, (const float*)y, sc, sh, C, Pg, npix, (float*)out_lrelu, (float*)out_relu);
  return hipGetLastError();
}

hipError_t launch_tap_gather(const TapArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(tap_gather_kernel, dim3((unsigned)(((size_t)a.N * a.Hout * a.Wout + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_tap_spread(const TapArgs& a, int is_bf16, hipStream_t st) {
  VP_DISPATCH(is_bf16, tap_spread_kernel, dim3((unsigned)(((size_t)a.N * a.Hin * a.Win + 255) / 256)), dim3(256), st, a);
  return hipGetLastError();
}

hipError_t launch_pack_inputs(const PackInputsArgs& a, int is_bf16, hipStream_t st) {
  VP_DISPATCH(is_bf16, pack_inputs_kernel, dim3(nblocks((size_t)a.N * a.HW)), dim3(256), st, a);
  return hipGetLastError();
}

hipError_t launch_frame_pack(const FramePackArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(frame_pack_kernel, dim3(nblocks((size_t)a.N * a.S * a.S * 2)), dim3(256), 0, st, a);
  return hipGetLastError();
}

int composite_nblocks(int N, int HW) { return nblocks((size_t)N * HW, 1024); }

hipError_t launch_composite_fwd(const CompositeArgs& a, int is_bf16, hipStream_t st) {
  VP_DISPATCH(is_bf16, composite_fwd_kernel, dim3(composite_nblocks(a.N, a.HW)), dim3(256), st, a);
  return hipGetLastError();
}

hipError_t launch_composite_bwd(const CompositeArgs& a, int is_bf16, hipStream_t st) {
  VP_DISPATCH(is_bf16, composite_bwd_kernel, dim3(nblocks((size_t)a.N * a.HW)), dim3(256), st, a);
  return hipGetLastError();
}

hipError_t launch_gan_loss(const GanLossArgs& a, int is_bf16, hipStream_t st) {
  VP_DISPATCH(is_bf16, gan_loss_kernel, dim3(1), dim3(1024), st, a);
  return hipGetLastError();
}

int perceptual_nblocks(size_t half, int is_bf16) { return nblocks(half / (is_bf16 ? 8 : 4), 1024); }

hipError_t launch_perceptual(const PerceptualArgs& a, int is_bf16, hipStream_t st) {
  VP_DISPATCH(is_bf16, perceptual_kernel, dim3(perceptual_nblocks(a.half, is_bf16)), dim3(256), st, a);
  return hipGetLastError();
}

hipError_t launch_loss_final(const LossFinalArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_maxpool_fwd(const void* x, void* y, int B, int H, int W, int C, int is_bf16, hipStream_t st) {
  const size_t work = (size_t)B * (H / 2) * (W / 2) * (C / (is_bf16 ? 8 : 4));
  if (is_bf16) hipLaunchKernelGGL((maxpool_fwd_kernel<bf16>), dim3(nblocks(work)), dim3(256), 0, st, (const bf16*)x, (bf16*)y, B, H, W, C);
  else hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(nblocks(work)), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C);
  return hipGetLastError();
}

hipError_t launch_maxpool_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int is_bf16, hipStream_t st) {
  const size_t work = (size_t)B * (H / 2) * (W / 2) * (C / (is_bf16 ? 8 : 4));
  if (is_bf16) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16>), dim3(nblocks(work)), dim3(256), 0, st, (const bf16*)x, (const bf16*)dy, (bf16*)dx, B, H, W, C);
  else hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(nblocks(work)), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, B, H, W, C);
  return hipGetLastError();
}

hipError_t launch_relu_bwd(const void* y, void* d, size_t n, int is_bf16, hipStream_t st) {
  const size_t nvec = n / (is_bf16 ? 8 : 4);
  if (is_bf16) hipLaunchKernelGGL((relu_bwd_kernel<bf16>), dim3(nblocks(nvec)), dim3(256), 0, st, (const bf16*)y, (bf16*)d, nvec);
  else hipLaunchKernelGGL((relu_bwd_kernel<float>), dim3(nblocks(nvec)), dim3(256), 0, st, (const float*)y, (float*)d, nvec);
  return hipGetLastError();
}

hipError_t launch_grad_pack_bf16(const float* src, void* dst, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(grad_pack_bf16_kernel, dim3(nblocks(n / 8 + 1, 4096)), dim3(256), 0, st, src, (bf16*)dst, n / 8, n);
  return hipGetLastError();
}

hipError_t launch_grad_unpack_bf16(const void* src, float* dst, size_t n, float scale, hipStream_t st) {
  hipLaunchKernelGGL(grad_unpack_bf16_kernel, dim3(nblocks(n / 8 + 1, 4096)), dim3(256), 0, st, (const bf16*)src, dst, n / 8, n, scale);
  return hipGetLastError();
}

hipError_t launch_adam(const AdamArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(adam_tf_kernel, dim3(nblocks(a.n, 4096)), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace vp
