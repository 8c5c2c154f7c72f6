// conv_cout1.hip - the backward passes of a 4x4 stride-1 convolution with ONE output channel over 512 input channels: the PatchGAN's
// last layer (pixrefer.py:111-131, `layer_5`).  First the backward-data pass, in both of its uses (the discriminator-loss pass over the
// three applications, the generator-loss pass over the fake one); the weight gradient (cout1_wgrad_kernel) is at the end of the file.
//
// dx[n, ih, iw, c] = act'(ref[n, ih, iw, c]) * sum over (kh, kw) of dy[n, ih - kh + pad, iw - kw + pad] * w[kh, kw, c]
//
// As a GEMM this is K = 16: the generic implicit-GEMM kernel ran it with K padded to 128 (the 8-channel padded dy x 16 taps) under a 32 KB
// staged epilogue at 1.5 TB/s of its algorithmic bytes, and the batch-norm backward of the producing layer (`layer_4`) then re-read dx and
// the layer's raw output for its two sums (bn_reduce_kernel<T, 1>, 28 - 35 us).  Here a 16-pixel tile is ONE 16x16x32 MFMA step per 16
// channels: the B fragment (the 16 taps of dy around a pixel; lane groups 2, 3 are the zero half of K) is shared by all 32 channel tiles,
// the A fragments (the weights, rounded to bf16 exactly as pack_weights_kernel rounds them) live in registers - wave w owns channels
// 128 w .. 128 w + 127 - and with the row permutation of perm_row() a lane ends with 8 consecutive channels of its pixel per tile pair:
// one 16-byte load of the reference, one 16-byte store, and - the producer being batch-normalised - one 16-byte load of the raw output
// for the two raw moments sum dx and sum dx * y of the gradient AS STORED (what staged_epilogue STATS == 2 produces; the finalize
// converts them: BnArgs::raw).  Blocks are persistent over a contiguous range of tiles of ONE batch-norm group and leave one partial row
// each.  The tile loop is branch-free (clamped prefetch with out-of-range offsets: see conv3x3_cout8_tile_kernel).
#include <hip/hip_runtime.h>

#include "conv_args.h"
#include "igemm_device.h"
#include "launch.h"
#include "vp_common.h"

namespace vp {

template <bool STATS>
__global__ __launch_bounds__(256) void cout1_bwd_kernel(const Cout1Args a) {
  constexpr int C = 512;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  // A fragments: tile T = 2 q + hh of the wave's 128 channels; row i of it is channel 32 q + 8 (i >> 2) + 4 hh + (i & 3)
  uint4 af[8];
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    const int c = 128 * wv + 32 * (T >> 1) + 8 * (i >> 2) + 4 * (T & 1) + (i & 3);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = g < 2 ? a.w[(8 * g + j) * C + c] : 0.f;
    af[T] = Elem<bf16>::pack(v);
  }
  // a block's tiles: a contiguous range of the 16-pixel tiles of ONE batch-norm group (the last tile of a group may be ragged: its lanes
  // beyond the group's pixels load zeros and store nowhere)
  const int grp = blockIdx.x / a.rows, r = blockIdx.x - grp * a.rows;
  const int tpg = a.tiles_per_group, pg = a.pix_per_group;
  const int t0 = (int)((long long)tpg * r / a.rows), t1 = (int)((long long)tpg * (r + 1) / a.rows);
  const int ntile = t1 - t0;
  const size_t tensor_bytes = (size_t)a.N * a.H * a.W * C * 2;
  __amdgpu_buffer_rsrc_t rsD = make_rsrc(a.dy, (unsigned)((size_t)a.N * a.Ho * a.Wo * a.ld_dy * 2));
  __amdgpu_buffer_rsrc_t rsR = make_rsrc(a.ref, (unsigned)tensor_bytes);
  __amdgpu_buffer_rsrc_t rsY = make_rsrc(STATS ? a.y : a.ref, (unsigned)tensor_bytes);
  __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.dx, (unsigned)tensor_bytes);
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const unsigned chan_off = (unsigned)((128 * wv + 8 * g) * 2);        // + 64 q: the lane's 8 channels of tile pair q
  const int HW = a.H * a.W;

  float s0[4][8], s1[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[q][e] = 0.f; s1[q][e] = 0.f; }

  unsigned short dq[8];
  u32x4 rq[4], yq[4];
  auto load_tile = [&](int k) {               // tile t0 + k; beyond the block's last tile: every offset out of range, nothing moves
    const int pl = (t0 + k) * 16 + i;                 // pixel inside the group
    const bool live = k < ntile && pl < pg;
    const int p = grp * pg + (live ? pl : 0);
    const int n = p / HW, rem = p - n * HW;
    const int ih = rem / a.W, iw = rem - ih * a.W;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 8 * g + j, kh = t >> 2, kw = t & 3;
      const int oh = ih - kh + a.pad, ow = iw - kw + a.pad;
      const bool ok = live && g < 2 && (unsigned)oh < (unsigned)a.Ho && (unsigned)ow < (unsigned)a.Wo;
      const unsigned off = ok ? (unsigned)((((n * a.Ho + oh) * a.Wo + ow) * a.ld_dy) * 2) : DMA_OOB;
      dq[j] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsD, (int)off, 0, 0);
    }
    const unsigned base = live ? (unsigned)p * (unsigned)(C * 2) + chan_off : DMA_OOB;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      rq[q] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (int)(base + 64u * q), 0, 0);
      if constexpr (STATS) yq[q] = __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)(base + 64u * q), 0, 0);
    }
  };
  load_tile(0);
  // (a store that goes nowhere behind the first tile's loads: the first trip then looks like every other one to the s_waitcnt pass)
#pragma unroll
  for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, rsX, (int)DMA_OOB, 0, 0);
  for (int k = 0; k < ntile; ++k) {
    const uint4 bf = make_uint4((unsigned)dq[0] | ((unsigned)dq[1] << 16), (unsigned)dq[2] | ((unsigned)dq[3] << 16),
                                (unsigned)dq[4] | ((unsigned)dq[5] << 16), (unsigned)dq[6] | ((unsigned)dq[7] << 16));
    u32x4 rc[4], yc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { rc[q] = rq[q]; if constexpr (STATS) yc[q] = yq[q]; }
    load_tile(k + 1);
    const int pl = (t0 + k) * 16 + i;
    const unsigned base = pl < pg ? (unsigned)(grp * pg + pl) * (unsigned)(C * 2) + chan_off : DMA_OOB;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 lo = mma16<bf16>(af[2 * q], bf, (f32x4){0.f, 0.f, 0.f, 0.f});
      const f32x4 hi = mma16<bf16>(af[2 * q + 1], bf, (f32x4){0.f, 0.f, 0.f, 0.f});
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      {
        float z[8];
        Elem<bf16>::unpack(make_uint4(rc[q].x, rc[q].y, rc[q].z, rc[q].w), z);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= act_grad(ACT_LRELU, z[e]);
      }
      const uint4 pk = Elem<bf16>::pack(v);
      __builtin_amdgcn_raw_buffer_store_b128((u32x4){pk.x, pk.y, pk.z, pk.w}, rsX, (int)(base + 64u * q), 0, 0);
      if constexpr (STATS) {
        float d[8], yy[8];
        Elem<bf16>::unpack(pk, d);
        Elem<bf16>::unpack(make_uint4(yc[q].x, yc[q].y, yc[q].z, yc[q].w), yy);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[q][e] += d[e]; s1[q][e] = fmaf(d[e], yy[e], s1[q][e]); }
      }
    }
  }
  if constexpr (!STATS) return;
  // fold the 16 pixel columns of a lane group; lane 16 g then holds the sums of channels 128 wv + 32 q + 8 g + e
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s0[q][e] += __shfl_xor(s0[q][e], o, 64); s1[q][e] += __shfl_xor(s1[q][e], o, 64); }
    }
  if (i == 0) {
    double* row = a.part + (size_t)blockIdx.x * 2 * C;        // rows of a group are consecutive: (grp * rows + r) == blockIdx.x
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 128 * wv + 32 * q + 8 * g + e;
        row[c] = (double)s0[q][e];
        row[C + c] = (double)s1[q][e];
      }
  }
}

// what the kernel handles (the caller has checked the layer: 4x4, stride 1, one output channel)
bool conv_cout1_bwd_eligible(const Cout1Args& a) {
  const long long px = (long long)a.N * a.H * a.W;
  return a.C == 512 && a.ks == 4 && a.dy && a.w && a.dx && a.ref && a.ref_act == ACT_LRELU && (!a.part || a.y) && a.groups >= 1 && a.N % a.groups == 0 &&
         px * 512 * 2 < 0x70000000ll && (long long)a.N * a.Ho * a.Wo * a.ld_dy * 2 < 0x70000000ll && a.rows >= 1;
}

hipError_t launch_conv_cout1_bwd(const Cout1Args& a0, hipStream_t st) {
  Cout1Args a = a0;
  a.pix_per_group = (a.N / a.groups) * a.H * a.W;
  a.tiles_per_group = (a.pix_per_group + 15) >> 4;
  if (a.rows > a.tiles_per_group) return hipErrorInvalidValue;
  if (a.part) hipLaunchKernelGGL(cout1_bwd_kernel<true>, dim3(a.groups * a.rows), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(cout1_bwd_kernel<false>, dim3(a.groups * a.rows), dim3(256), 0, st, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// cout1_wgrad_kernel: the weight gradient of the same layer, dW[kh, kw, c] = sum over pixels of dy[n, ih - kh + pad, iw - kw + pad] *
// x[n, ih, iw, c].  As a GEMM it is 16 "rows" against 512 channels with K = all pixels, K being the strided dimension of both operands:
// the generic register-transposing kernel ran it at 1.4 TB/s behind a tap-spreading pass (tap_spread_kernel) and in front of a slab
// reduce.  Here a WAVE owns a pixel at a time: its 64 lanes hold the pixel's 512 channels (one coalesced 1 KB read, 8 channels per lane),
// the 16 values of dy around the pixel reach it as wave-uniform scalars, and every lane keeps 16 x 8 float32 sums in registers - 128 fused
// multiply-adds per 16 bytes read.  The four waves of a block fold through LDS and leave one [16][512] slab; cout1_wgrad_reduce_kernel
// adds the slabs in a fixed order.
__global__ __launch_bounds__(256) void cout1_wgrad_kernel(const Cout1Args a) {
  constexpr int C = 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 waves][64 sums][64 lanes] float
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int npix = a.N * a.H * a.W, HW = a.H * a.W;
  const int nw = gridDim.x * 4, w = blockIdx.x * 4 + wv;
  // the wave's pixels: a contiguous range
  const int p0 = (int)((long long)npix * w / nw), p1 = (int)((long long)npix * (w + 1) / nw);
  __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.ref, (unsigned)((size_t)npix * C * 2));
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  float acc[16][8];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
  // per pixel: one 16-byte load of x per lane and ONE 2-byte load of dy - lane t < 16 fetches tap t of the pixel's neighbourhood (an
  // out-of-range buffer offset where the tap falls into the padding: zero) -, both U pixels ahead; the sixteen values then reach the
  // multiply-adds as scalars through v_readlane.  (As sixteen scalar loads per pixel hipcc either sank each load into a branch on its
  // bounds test, waited for each one in front of its four multiply-adds, or - all of them hoisted - spilled the scalar registers:
  // 0.10 - 0.12 ms for the launch.)
  __amdgpu_buffer_rsrc_t rsD = make_rsrc(a.dy, (unsigned)((size_t)a.N * a.Ho * a.Wo * a.ld_dy * 2));
  constexpr int U = 4;                       // pixels in flight
  u32x4 xq[U];
  unsigned dq[U];
  const int tkh = (lane & 15) >> 2, tkw = lane & 3;
  // (pixels beyond the wave's range are clamped to its last one - always a valid address, no test around the loads - and their dy is
  // zeroed where it is used)
  auto fetch = [&](int p, u32x4& dst, unsigned& ddst) {
    const int pc = p < p1 ? p : p1 - 1;                           // wave-uniform
    dst = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)((unsigned)pc * (unsigned)(C * 2) + (unsigned)lane * 16u), 0, 0);
    const int n = pc / HW, rem = pc - n * HW;
    const int ih = rem / a.W, iw = rem - ih * a.W;
    const int oh = ih - tkh + a.pad, ow = iw - tkw + a.pad;
    const bool ok = lane < 16 && (unsigned)oh < (unsigned)a.Ho && (unsigned)ow < (unsigned)a.Wo;
    const unsigned doff = ok ? (unsigned)((((n * a.Ho + oh) * a.Wo + ow) * a.ld_dy) * 2) : DMA_OOB;
    ddst = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsD, (int)doff, 0, 0);
  };
#pragma unroll
  for (int u = 0; u < U; ++u) fetch(p0 + u, xq[u], dq[u]);
  for (int pb = p0; pb < p1; pb += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u32x4 xv = xq[u];
      const unsigned dv = pb + u < p1 ? dq[u] : 0u;
      fetch(pb + u + U, xq[u], dq[u]);
      float x[8];
      Elem<bf16>::unpack(make_uint4(xv.x, xv.y, xv.z, xv.w), x);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float d = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)dv, t) << 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = fmaf(d, x[e], acc[t][e]);
      }
    }
  }
  // fold the block's four waves (waves 1 .. 3 through LDS, in that order; two rounds of eight taps: 48 KB), one slab per block
  float* sm = reinterpret_cast<float*>(smem);
  float* slab = a.slabs + (size_t)blockIdx.x * 16 * C;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();
    if (wv > 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[((wv - 1) * 64 + t * 8 + e) * 64 + lane] = acc[8 * half + t][e];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = ((acc[8 * half + t][e] + sm[(t * 8 + e) * 64 + lane]) + sm[(64 + t * 8 + e) * 64 + lane]) + sm[(128 + t * 8 + e) * 64 + lane];
        float4* dst = reinterpret_cast<float4*>(slab + (8 * half + t) * C + lane * 8);
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        dst[1] = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
  }
}

// dW[i] = sum over slabs, in a fixed order: a block owns 16 consecutive elements, its 256 threads are 16 slab phases x 16 elements (a
// one-thread-per-element walk over 256 - 512 slabs was 100 us of load latency)
__global__ __launch_bounds__(256) void cout1_wgrad_reduce_kernel(const float* slabs, int nslab, float* dW, int n) {
  __shared__ float sm[16][17];
  const int j = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + j;
  float s = 0.f;
  for (int k = ph; k < nslab; k += 16 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (k + 16 * u < nslab && i < n) ? slabs[(size_t)(k + 16 * u) * n + i] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  sm[ph][j] = s;
  __syncthreads();
  if (ph == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sm[q][j];
    dW[i] = t;
  }
}

bool conv_cout1_wgrad_eligible(const Cout1Args& a) {
  const long long px = (long long)a.N * a.H * a.W;
  return a.C == 512 && a.ks == 4 && a.dy && a.ref && a.dW && a.slabs && a.rows >= 1 && (a.ld_dy & 1) == 0 && px * 512 * 2 < 0x70000000ll && px >= 4 * (long long)a.rows;
}

// a.ref = the layer's (materialised, activated) input, a.rows = blocks = slabs, a.slabs >= rows * 16 * 512 floats, a.dW [16][512]
hipError_t launch_conv_cout1_wgrad(const Cout1Args& a, hipStream_t st) {
  const int smem = 3 * 64 * 64 * 4;
  (void)hipFuncSetAttribute((const void*)cout1_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipLaunchKernelGGL(cout1_wgrad_kernel, dim3(a.rows), dim3(256), smem, st, a);
  hipLaunchKernelGGL(cout1_wgrad_reduce_kernel, dim3(16 * 512 / 16), dim3(256), 0, st, a.slabs, a.rows, a.dW, 16 * 512);
  return hipGetLastError();
}

}  // namespace vp
