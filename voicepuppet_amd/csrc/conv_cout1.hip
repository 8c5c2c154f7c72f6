// conv_cout1.hip - the backward-data pass of a 4x4 stride-1 convolution with ONE output channel over 512 input channels: the PatchGAN's
// last layer (pixrefer.py:111-131, `layer_5`), in both of its passes (the discriminator-loss pass over the three applications, the
// generator-loss pass over the fake one).
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

}  // namespace vp
