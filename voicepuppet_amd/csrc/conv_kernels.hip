// Implicit-GEMM convolution family for gfx950 (MI355X), hand-written MFMA kernels (16x16x32 bf16 / 16x16x4 f32 tiles).
//
// Kernels on the step path (DESIGN.md section 3 has the measurements and what bounds each):
//   igemm_dma_kernel   conv fwd / bwd-data, deconv fwd / bwd-data: both operands HBM/L2 -> LDS by LDS-DMA (rb_swz image: coalesced 64-byte
//                      segments, conflict-free ds_read_b128), 3-deep ring with counted vmcnt, one barrier per 64-byte K chunk,
//                      scalar-stepped buffer-descriptor loader (fastk), staged LDS epilogue with 16-byte row stores (+ BN statistics)
//   igemm_ws_kernel    the same GEMM with 4 producer waves issuing every DMA and consumer waves doing only ds_read + MFMA
//   conv_cin8_kernel   8-channel (padded image) inputs: pieces straight from global memory, weights as LDS fragments, no operand tiles
//   deconv_cout4_kernel  4-channel transposed conv: 4 parity classes x 4 channels = the 16 rows of one MFMA tile
//   wgrad_kernel       bwd-weight: K = pixels (the strided NHWC dim): register loader + 8x8 transposes into LDS planes,
//                      division-free padded-grid K walk, split-K slabs + deterministic reduces
//   igemm_splitk_reduce_kernel, wgrad_reduce_kernel, wgrad_reduce_wave_kernel
// The kernel variants that were measured slower (register-operand loader, resident-weight persistent tiles, 128-byte K chunks,
// register-double-buffered 256x256 tile, direct epilogue: EXPERIMENTS.md) were deleted in round 4; `git log -- voicepuppet_amd/csrc/conv_db.hip`
// and `.../experiments/igemm_experiments.inc` have them.  igemm_kernel (register loader with the deferred-BN prologue) serves the single-op API.
//
// LDS plane layout of the register-loader kernels (igemm_kernel, wgrad_kernel): a tile of ROWS rows x 64 bytes of K is stored as 4
// planes (one per 16-byte k-piece g), plane g = ROWS consecutive 16-byte slots.  MFMA lane (i = lane&15, g = lane>>4) reads
// slot(row0+i) of plane g with one ds_read_b128: the 16 lanes of every b128 service group hit 16 distinct 16-byte slots of a
// 256-byte bank row (conflict-free); writers use 8-lane contiguous (row loader) or XOR-swizzled (transposing loader) slots.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "conv_args.h"
#include "conv_ops.h"
#include "igemm_device.h"
#include "smallp_args.h"
#include "launch.h"
#include "vp_common.h"

#ifndef VP_RING
#define VP_RING 3      // LDS ring depth of the DMA GEMM loop (stages); 3 = one chunk in flight across the barrier
#endif
#ifndef VP_REGB_EPI_BYTES
#define VP_REGB_EPI_BYTES (36 * 1024)   // LDS budget of igemm_regb_kernel's staged epilogue (its weight ring is far smaller)
#endif
#ifndef VP_ABLATE
#define VP_ABLATE 0   // build-time ablation of the LDS-DMA GEMM loop: 1 = no MFMA, 2 = no DMA in the loop
#endif

namespace vp {

// ------------------------------------------------------------------------------------------------
// slot permutation inside a 16-row block (b = row >> 4)
//   SWZ 0: identity (row loader)
//   SWZ 1: transposing loader; a thread owns E consecutive rows and writes them in E instructions,
//          so lanes of one ds_write_b128 group are E rows apart -> spread them over the 8 slots.
// ------------------------------------------------------------------------------------------------
template <int SWZ, int E> __device__ __forceinline__ int lds_slot(int row) {
  if (SWZ == 0) return row;
  int r = row & 15, b = row >> 4;
  int s16;
  if (E == 4) s16 = (((r & 3) << 2) | (r >> 2)) ^ ((b & 1) << 2);
  else        s16 = (((r & 7) << 1) | (r >> 3)) ^ ((b & 3) << 1);
  return (row & ~15) | s16;
}

// acc[tc][tp] += A(tile tc) x B(tile tp) for one 64-byte K chunk
template <typename T, int TC, int TP, int BC, int BP, int SWZ>
__device__ __forceinline__ void mma_chunk(const uint4* __restrict__ ldsA, const uint4* __restrict__ ldsB,
                                          int rowA0, int rowB0, int lane, f32x4 (&acc)[TC][TP]) {
  constexpr int E = Elem<T>::E;
  const int i = lane & 15, g = lane >> 4;
  uint4 fa[TC], fb[TP];
#pragma unroll
  for (int t = 0; t < TC; ++t) fa[t] = ldsA[g * BC + lds_slot<SWZ, E>(rowA0 + t * 16 + i)];
#pragma unroll
  for (int t = 0; t < TP; ++t) fb[t] = ldsB[g * BP + lds_slot<SWZ, E>(rowB0 + t * 16 + i)];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) acc[tc][tp] = mma16<T>(fa[tc], fb[tp], acc[tc][tp]);
}

// deferred-BN affine + activation on one 16-byte piece (channels c .. c+E-1 of BN group grp)
template <typename T>
__device__ __forceinline__ uint4 prologue_piece(uint4 v, const float* __restrict__ pa, const float* __restrict__ pb, int act) {
  constexpr int E = Elem<T>::E;
  if (pa == nullptr && act == ACT_NONE) return v;
  float f[E];
  Elem<T>::unpack(v, f);
  if (pa != nullptr) {
#pragma unroll
    for (int e = 0; e < E; e += 4) {
      const float4 a4 = *reinterpret_cast<const float4*>(pa + e);
      const float4 b4 = *reinterpret_cast<const float4*>(pb + e);
      f[e + 0] = fmaf(a4.x, f[e + 0], b4.x); f[e + 1] = fmaf(a4.y, f[e + 1], b4.y);
      f[e + 2] = fmaf(a4.z, f[e + 2], b4.z); f[e + 3] = fmaf(a4.w, f[e + 3], b4.w);
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) f[e] = act_apply(act, f[e]);
  return Elem<T>::pack(f);
}

// Load channels [c, c+E) of pixel (n, ih, iw) of a PixSrc (after affine + act); zeros outside.
template <typename T>
__device__ __forceinline__ uint4 load_pix_piece(const PixSrc& x, int n, int ih, int iw, int H, int W, int c, bool ok) {
  if (!ok || (unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) return make_uint4(0, 0, 0, 0);
  // explicit selects (no dynamic indexing of the by-value argument block: that would go to scratch)
  const bool s = c >= x.C[0];
  const int cl = s ? c - x.C[0] : c;
  const int C = s ? x.C[1] : x.C[0];
  const T* p = reinterpret_cast<const T*>(s ? x.ptr[1] : x.ptr[0]) + ((size_t)(n * H + ih) * W + iw) * C + cl;
  uint4 v = *reinterpret_cast<const uint4*>(p);
  const float* pa = s ? x.aff_a[1] : x.aff_a[0];
  const float* pb = s ? x.aff_b[1] : x.aff_b[0];
  if (pa != nullptr) {
    const int off = (n / x.group_n) * C + cl;
    pa += off; pb += off;
  }
  return prologue_piece<T>(v, pa, pb, x.act);
}

// ------------------------------------------------------------------------------------------------
// epilogue shared by igemm_kernel and splitk_reduce_kernel: 4 consecutive output channels of one pixel
// ------------------------------------------------------------------------------------------------
template <typename T, bool DUAL = false>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, int cls, int pidx, int c0, float v[4]) {
  if (c0 >= a.Cout) return;
  const int hw = a.Hg * a.Wg;
  const int n = pidx / hw;
  const int rem = pidx - n * hw;
  const int q = rem / a.Wg;
  const int r = rem - q * a.Wg;
  const size_t pix = ((size_t)n * a.Hof + (q * a.os + a.o0h[cls])) * a.Wof + (r * a.os + a.o0w[cls]);
  const int nv = min(4, a.Cout - c0);
  void* Yp = a.Y;
  const void* refp = a.ref;
  int accu = a.accumulate, yf32 = a.y_f32;
  if constexpr (DUAL) {
    if (c0 >= a.split_c) { Yp = a.Y2; refp = a.ref2; accu = a.accumulate2; yf32 = a.y2_f32; c0 -= a.split_c; }     // two-output form (conv_args.h)
  }
  const size_t off = pix * a.ldY + c0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e < nv) {
      if (a.bias) v[e] += a.bias[c0 + e];
      v[e] = act_apply(a.out_act, v[e]);
    }
  }
  if (refp) {
    const T* rp = reinterpret_cast<const T*>(refp) + off;
    const int goff = (n / a.ref_group_n) * a.Cout + c0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e < nv) {
        float z = Elem<T>::ld(rp + e);
        if (a.ref_a) z = fmaf(a.ref_a[goff + e], z, a.ref_b[goff + e]);
        v[e] *= act_grad(a.ref_act, z);
      }
    }
  }
  if (yf32) {
    float* yp = reinterpret_cast<float*>(Yp) + off;
    if (accu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) v[e] += yp[e];
    }
    if (nv == 4 && (a.ldY & 3) == 0) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) yp[e] = v[e];
    }
  } else {
    T* yp = reinterpret_cast<T*>(Yp) + off;
    if (accu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) v[e] += Elem<T>::ld(yp + e);
    }
    if (nv == 4 && (a.ldY & 3) == 0) {
      if (sizeof(T) == 4) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
      else *reinterpret_cast<uint2*>(yp) = make_uint2(f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16),
                                                     f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) Elem<T>::st(yp + e, v[e]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// igemm_kernel: grid = (pixel tiles, channel tiles, nclass * splitk), 256 threads = 4 waves (WC x WP)
//   MFMA A operand = weights (rows -> output channels), B operand = pixels (cols), so one lane ends
//   up with 4 consecutive channels of one pixel: a single 16-byte (f32) / 8-byte (bf16) NHWC store.
// ------------------------------------------------------------------------------------------------
template <typename T, int WC, int WP, int TC, int TP>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs a) {
  constexpr int E = Elem<T>::E, KC = 4 * E;
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16;
  constexpr int PA = (BC + 63) / 64, PB = (BP + 63) / 64;   // staging passes (64 rows per pass)
  constexpr int BUF = 4 * (BC + BP);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  int* ltap = reinterpret_cast<int*>(lds + 2 * BUF);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cls = blockIdx.z / a.splitk, split = blockIdx.z - cls * a.splitk;
  const int P = a.N * a.Hg * a.Wg;
  const int p_base = blockIdx.x * BP, c_base = blockIdx.y * BC;

  if (tid < 16) ltap[tid] = (tid < a.ntaps) ? (((int)a.taps[cls].dh[tid] << 16) | ((int)a.taps[cls].dw[tid] & 0xffff)) : 0;

  // staging thread -> (row within 16, k-piece g): 8 consecutive lanes = 8 consecutive rows of one plane
  const int r16 = (lane & 7) + 8 * (lane >> 5);
  const int g = (lane >> 3) & 3;

  // per-thread pixel rows (fixed over the K loop)
  int pn[PB], pbh[PB], pbw[PB];
  bool pok[PB];
#pragma unroll
  for (int ps = 0; ps < PB; ++ps) {
    const int row = ps * 64 + wave * 16 + r16;
    const int pidx = p_base + row;
    pok[ps] = (row < BP) && (pidx < P);
    const int hw = a.Hg * a.Wg;
    const int pc = pok[ps] ? pidx : 0;
    const int n = pc / hw, rem = pc - n * hw, q = rem / a.Wg;
    pn[ps] = n; pbh[ps] = q * a.sh; pbw[ps] = (rem - q * a.Wg) * a.sw;
  }
  const T* wp = reinterpret_cast<const T*>(a.Wp) + (size_t)cls * a.wp_rows * a.Kpad;   // [K chunk][row][KC]

  const int nchunk = a.Kpad / KC;
  const int per = (nchunk + a.splitk - 1) / a.splitk;
  const int kc0 = split * per, kc1 = min(nchunk, kc0 + per);

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging registers as named scalars (arrays captured by the lambdas below end up in scratch)
  uint4 ra0, ra1, rb0, rb1;
  ra0 = ra1 = rb0 = rb1 = make_uint4(0, 0, 0, 0);
  __syncthreads();   // tap table visible

  auto load_a = [&](int kc, int ps) -> uint4 {
    const int row = ps * 64 + wave * 16 + r16;
    if (BC % 64 == 0 || row < BC) return *reinterpret_cast<const uint4*>(wp + ((size_t)kc * a.wp_rows + c_base + row) * KC + g * E);
    return make_uint4(0, 0, 0, 0);
  };
  auto load_b = [&](int kc, int ps, int n, int bh, int bw, bool ok) -> uint4 {
    const int k0 = kc * KC + g * E;
    const int tap = k0 >> a.log2Cin;
    const int ci = k0 & a.cin_mask;
    const bool tok = tap < a.ntaps;
    const int tv = ltap[tok ? tap : 0];
    const int dh = tv >> 16, dw = (int)(short)(tv & 0xffff);
    const int row = ps * 64 + wave * 16 + r16;
    if (BP % 64 == 0 || row < BP) return load_pix_piece<T>(a.x, n, bh + dh, bw + dw, a.Hin, a.Win, ci, ok && tok);
    return make_uint4(0, 0, 0, 0);
  };
  auto stage_load = [&](int kc) {
    ra0 = load_a(kc, 0);
    if (PA > 1) ra1 = load_a(kc, 1);
    rb0 = load_b(kc, 0, pn[0], pbh[0], pbw[0], pok[0]);
    if (PB > 1) rb1 = load_b(kc, 1, pn[PB - 1], pbh[PB - 1], pbw[PB - 1], pok[PB - 1]);
  };
  auto stage_store = [&](int buf) {
    uint4* la = lds + buf * BUF;
    uint4* lb = la + 4 * BC;
    const int row = wave * 16 + r16;
    if (BC % 64 == 0 || row < BC) la[g * BC + row] = ra0;
    if (PA > 1) la[g * BC + 64 + row] = ra1;
    if (BP % 64 == 0 || row < BP) lb[g * BP + row] = rb0;
    if (PB > 1) lb[g * BP + 64 + row] = rb1;
  };

  const int wc = wave / WP, wpi = wave - wc * WP;
  const int rowA0 = wc * TC * 16, rowB0 = wpi * TP * 16;

  if (kc0 < kc1) {
    stage_load(kc0);
    stage_store(0);
    __syncthreads();
    for (int kc = kc0; kc < kc1; ++kc) {
      const int buf = (kc - kc0) & 1;
      const bool more = kc + 1 < kc1;
      if (more) stage_load(kc + 1);
      const uint4* la = lds + buf * BUF;
      mma_chunk<T, TC, TP, BC, BP, 0>(la, la + 4 * BC, rowA0, rowB0, lane, acc);
      if (more) stage_store(buf ^ 1);
      __syncthreads();
    }
  }

  // epilogue: lane holds pixel (lane&15), channels 4*(lane>>4) .. +3 of each 16x16 tile
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) {
    const int pidx = p_base + rowB0 + tp * 16 + (lane & 15);
    if (pidx >= P) continue;
#pragma unroll
    for (int tc = 0; tc < TC; ++tc) {
      const int c0 = c_base + tile_chan0(a.rowperm, rowA0 / 16 + tc, lane >> 4);
      float v[4] = {acc[tc][tp][0], acc[tc][tp][1], acc[tc][tp][2], acc[tc][tp][3]};
      if (a.splitk > 1) {
        float* pp = a.partial + (((size_t)(cls * a.splitk + split) * P + pidx) * a.CoutPad + c0);
        *reinterpret_cast<float4*>(pp) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        igemm_epilogue<T>(a, cls, pidx, c0, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// igemm_dma_kernel: same GEMM, operands that need no prologue (materialised activations, gradients,
// packed weights) go HBM/L2 -> LDS directly with global_load_lds_dwordx4 (no VGPR round trip, no
// ds_write pass).  One wave instruction moves a 16-row block x 4 k-pieces = 64 consecutive 16-byte
// slots; the LDS image is [row block][16 rows][4 k-pieces] with the pieces of each row XOR-permuted
// (rb_swz) so that global reads are coalesced AND ds_read_b128 stays conflict-free.  Padding / ragged
// pixels read a 16-byte zero page instead (LDS-DMA cannot zero-fill).  One barrier per K chunk.
// ------------------------------------------------------------------------------------------------
template <typename T, int WC, int WP, int TC, int TP, bool STAGED, int STATS = 0, bool DUAL = false>
__global__ __launch_bounds__(WC * WP * 64) void igemm_dma_kernel(const IgemmArgs a) {
  constexpr int E = Elem<T>::E, KC = 4 * E;
  constexpr int NW = WC * WP, NT = NW * 64;             // 4 waves (256 threads) or 8 waves (512 threads: 128x256 / 256x256 tiles)
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16;
  constexpr int NBA = BC / 16, NBB = BP / 16;          // 16-row blocks per operand tile
  constexpr int JA = (NBA + NW - 1) / NW, JB = (NBB + NW - 1) / NW; // DMA instructions per wave per chunk
  constexpr int BUF = 4 * (BC + BP);
  // every wave issues the same number of DMAs per chunk -> a counted vmcnt can keep one chunk in flight
  // across the barrier (3-deep LDS ring); otherwise 2 buffers and a full drain per chunk
  constexpr bool RING = (NBA % NW == 0) && (NBB % NW == 0);
  constexpr int NST = RING ? VP_RING : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  int* ltap = reinterpret_cast<int*>(lds + NST * BUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cls = blockIdx.z / a.splitk, split = blockIdx.z - cls * a.splitk;
  const int P = a.N * a.Hg * a.Wg;
  // XCD-aware tile order (speed only): workgroup b is observed to run on XCD b % 8, each XCD has a private 4 MiB L2.
  // Give every XCD a contiguous range of logical tiles, ordered pixel-tile major / channel-tile minor, so the channel
  // tiles that re-read one activation tile run back to back on the same L2 and the weight slabs stay resident.
  int pt, ct;
  {
    const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = id & 7, slot = id >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    ct = logical % (int)gridDim.y; pt = logical / (int)gridDim.y;
  }
  const int p_base = pt * BP, c_base = ct * BC;
#if VP_ABLATE & 4
  // bandwidth experiment (results are garbage): a DMA instruction fetches 8 rows x 128 contiguous bytes instead of 16 x 64
  const int r = lane >> 3, g = lane & 7;
#else
  const int r = lane >> 2, g = (lane & 3) ^ rb_swz(lane >> 2);   // row in the 16-row block, k-piece fetched
#endif

  if (tid < 16) ltap[tid] = (tid < a.ntaps) ? (((int)a.taps[cls].dh[tid] << 16) | ((int)a.taps[cls].dw[tid] & 0xffff)) : 0;

  // fixed per-thread rows
  const T* wrow[JA];
#pragma unroll
  for (int j = 0; j < JA; ++j)
    wrow[j] = reinterpret_cast<const T*>(a.Wp) + (size_t)cls * a.wp_rows * a.Kpad + ((size_t)c_base + (wave + NW * j) * 16 + r) * KC + g * E;
  int pn[JB], pbh[JB], pbw[JB];
  bool pok[JB];
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int pidx = p_base + (wave + NW * j) * 16 + r;
    pok[j] = (wave + NW * j < NBB) && (pidx < P);
    const int hw = a.Hg * a.Wg;
    const int pc = pok[j] ? pidx : 0;
    const int n = pc / hw, rem = pc - n * hw, q = rem / a.Wg;
    pn[j] = n * a.Hin; pbh[j] = q * a.sh; pbw[j] = (rem - q * a.Wg) * a.sw;
  }
  const T* x0 = reinterpret_cast<const T*>(a.x.ptr[0]);
  const T* x1 = reinterpret_cast<const T*>(a.x.ptr[1]);
  const int C0 = a.x.C[0], C1 = a.x.C[1];

  const int nchunk = a.Kpad / KC;
  const int per = (nchunk + a.splitk - 1) / a.splitk;
  const int kc0 = split * per, kc1 = min(nchunk, kc0 + per);

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int kc, int buf) {
    uint4* la = lds + buf * BUF;
    uint4* lb = la + 4 * BC;
#pragma unroll
    for (int j = 0; j < JA; ++j)
      if (NBA % NW == 0 || wave + NW * j < NBA) dma16(wrow[j] + (size_t)kc * a.wp_rows * KC, la + (wave + NW * j) * 64);
    const int k0 = kc * KC + g * E;
    const int tap = k0 >> a.log2Cin;
    const int ci = k0 & a.cin_mask;
    const bool tok = tap < a.ntaps;
    const int tv = ltap[tok ? tap : 0];
    const int dh = tv >> 16, dw = (int)(short)(tv & 0xffff);
    const bool s1 = ci >= C0;
    const T* xb = s1 ? x1 : x0;
    const int Cs = s1 ? C1 : C0;
    const int cl = s1 ? ci - C0 : ci;
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      if (NBB % NW == 0 || wave + NW * j < NBB) {
        const int ih = pbh[j] + dh, iw = pbw[j] + dw;
        const bool ok = pok[j] && tok && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
        const void* src = ok ? (const void*)(xb + ((size_t)(pn[j] + ih) * a.Win + iw) * Cs + cl) : a.zeros;
        dma16(src, lb + (wave + NW * j) * 64);
      }
    }
  };

  // ---- fast loader (a.fastk: every 64-byte K chunk lies inside one tap and one source tensor) -------------------
  // K is walked as segments (tap, source); per segment each lane computes ONE byte offset per pixel row (or DMA_OOB for
  // padding), inside a segment a chunk only bumps a scalar offset.  The weight stream is chunk-major: scalar bump too.
  const unsigned es = sizeof(T);
  __amdgpu_buffer_rsrc_t rsW = make_rsrc(reinterpret_cast<const T*>(a.Wp) + (size_t)cls * a.wp_rows * a.Kpad, 0xFFFFFFFFu);
  __amdgpu_buffer_rsrc_t rsX0 = make_rsrc(x0, (unsigned)((size_t)a.N * a.Hin * a.Win * C0 * es));
  __amdgpu_buffer_rsrc_t rsX1 = make_rsrc(x1 ? (const void*)x1 : (const void*)x0, (unsigned)((size_t)a.N * a.Hin * a.Win * C1 * es));
  unsigned wvo[JA];
#pragma unroll
  for (int j = 0; j < JA; ++j) wvo[j] = (unsigned)(((c_base + (wave + NW * j) * 16 + r) * KC + g * E) * es);
  const unsigned wstep = (unsigned)(a.wp_rows * KC * es);
  unsigned f_wso = (unsigned)kc0 * wstep;          // scalar: weight chunk offset
  unsigned f_xso = 0;                              // scalar: channel offset inside the segment (bytes)
  int f_left = 0;                                  // chunks left in the current segment
  int f_tap = 0, f_src = 0;                        // next segment to open
  unsigned f_xvo[JB];
  bool f_use1 = false;
  if (a.fastk) {
    const int k0 = kc0 * KC;
    f_tap = k0 >> a.log2Cin;
    const int ci = k0 & a.cin_mask;
    f_src = ci >= C0 ? 1 : 0;
    // a split that starts inside a segment: open it now and skip the consumed chunks
    const int cl = f_src ? ci - C0 : ci;
    f_xso = (unsigned)cl * es;
    f_left = -(cl / KC);                            // negative: corrected when the segment opens below
  }
  auto open_segment = [&]() {
    const bool tok = f_tap < a.ntaps;
    const int tv = ltap[tok ? f_tap : 0];
    const int dh = tv >> 16, dw = (int)(short)(tv & 0xffff);
    f_use1 = f_src != 0;
    const int Cs = f_use1 ? C1 : C0;
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int ih = pbh[j] + dh, iw = pbw[j] + dw;
      const bool ok = pok[j] && tok && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      f_xvo[j] = ok ? (unsigned)((((pn[j] + ih) * a.Win + iw) * Cs + g * E) * es) : DMA_OOB;
    }
    f_left += Cs / KC;
    // next segment: the other source of the same tap, or the next tap
    if (!f_use1 && C1 > 0) f_src = 1; else { f_src = 0; ++f_tap; }
  };
  auto issue_fast = [&](int buf) {
    uint4* la = lds + buf * BUF;
    uint4* lb = la + 4 * BC;
    if (f_left <= 0) { const int skipped = -f_left; f_left = 0; open_segment(); f_left -= skipped; if (skipped == 0) f_xso = 0; }
#pragma unroll
    for (int j = 0; j < JA; ++j)
      if (NBA % NW == 0 || wave + NW * j < NBA) dma16_buf(rsW, wvo[j], f_wso, la + (wave + NW * j) * 64);
    const __amdgpu_buffer_rsrc_t rx = f_use1 ? rsX1 : rsX0;
#pragma unroll
    for (int j = 0; j < JB; ++j)
      if (NBB % NW == 0 || wave + NW * j < NBB) dma16_buf(rx, f_xvo[j], f_xso, lb + (wave + NW * j) * 64);
    f_wso += wstep;
    f_xso += KC * es;
    --f_left;
  };
  auto issue_any = [&](int kc, int buf) { if (a.fastk) issue_fast(buf); else issue(kc, buf); };

  const int wc = wave / WP, wpi = wave - wc * WP;
  const int blkA0 = wc * TC, blkB0 = wpi * TP;

  __syncthreads();   // tap table visible
  if (kc0 < kc1) {
    if (RING) {
      // NST-deep ring: chunks kc+1 .. kc+NST-2 stay in flight across the barrier while chunk kc is consumed
#pragma unroll
      for (int d = 0; d < NST - 1; ++d) if (kc0 + d < kc1) issue_any(kc0 + d, d);
      int st = 0;
      for (int kc = kc0; kc < kc1; ++kc) {
        if (kc + NST - 2 < kc1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (JA + JB)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail: fewer batches outstanding than the constant assumes
        __builtin_amdgcn_s_barrier();   // every wave's DMA of chunk kc landed; every wave is done reading chunk kc-1
        asm volatile("" ::: "memory");
        const int stn = st == 0 ? NST - 1 : st - 1;   // the buffer chunk kc-1 used
#if !(VP_ABLATE & 2)
        if (kc + NST - 1 < kc1) issue_any(kc + NST - 1, stn);
#endif
        const uint4* la = lds + st * BUF;
#if !(VP_ABLATE & 1)
        mma_chunk_rb<T, TC, TP, BC, BP>(la, la + 4 * BC, blkA0, blkB0, lane, acc);
#endif
        st = st == NST - 1 ? 0 : st + 1;
      }
    } else {
      issue_any(kc0, 0);
      for (int kc = kc0; kc < kc1; ++kc) {
        const int buf = (kc - kc0) & 1;
        __syncthreads();   // waits for this wave's DMA (vmcnt) and for every wave's reads of the other buffer
        if (kc + 1 < kc1) issue_any(kc + 1, buf ^ 1);
        const uint4* la = lds + buf * BUF;
        mma_chunk_rb<T, TC, TP, BC, BP>(la, la + 4 * BC, blkA0, blkB0, lane, acc);
      }
    }
  }

#if VP_ABLATE & 8
  {   // epilogue ablation: every accumulator stays live (no dead-code elimination of the MFMAs), nothing is stored
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
      for (int j = 0; j < TP; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 123.456f) reinterpret_cast<float*>(a.Y)[0] = 1.f;
    return;
  }
#endif
  if (STAGED) {
    constexpr int RINGB = NST * BUF * 16;
    constexpr int NPASS = epi_passes(BC, BP, WP, RINGB);
    staged_epilogue<T, TC, TP, BC, BP, NPASS, NT, STATS, 0, DUAL>(a, LinearPix{a, cls, p_base, P}, c_base, blkA0, blkB0, acc, smem, pt, cls);
    return;
  }
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) {
    const int pidx = p_base + (blkB0 + tp) * 16 + (lane & 15);
    if (pidx >= P) continue;
#pragma unroll
    for (int tc = 0; tc < TC; ++tc) {
      const int c0 = c_base + tile_chan0(a.rowperm, blkA0 + tc, lane >> 4);
      float v[4] = {acc[tc][tp][0], acc[tc][tp][1], acc[tc][tp][2], acc[tc][tp][3]};
      if (a.splitk > 1) {
        float* pp = a.partial + (((size_t)(cls * a.splitk + split) * P + pidx) * a.CoutPad + c0);
        *reinterpret_cast<float4*>(pp) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        igemm_epilogue<T, DUAL>(a, cls, pidx, c0, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// igemm_ws_kernel: wave-specialised form of igemm_dma_kernel.  An LDS-DMA costs its issuing wave 60-185 cycles of issue
// time (MI355X_MICROARCH.md), comparable to the 16 MFMAs of a K chunk, so in the kernels above the DMA issue and the
// matrix pipe take turns inside every wave.  Here NPW = 4 producer waves (one per SIMD) issue ALL the DMAs of a stage and
// the WC x WP consumer waves only do barrier -> ds_read -> MFMA; the producers' issue stalls overlap the consumers' MFMAs
// on the same SIMD.  One barrier per K chunk, NST-deep ring:
//   producer kc: wait vmcnt (its share of chunk kc landed) -> barrier kc -> issue chunk kc+NST-1 into the stage chunk kc-1 used
//   consumer kc: barrier kc -> fragments + MFMAs of chunk kc
// fastk operands only (scalar K stepping, hardware zero fill); epilogue = the staged 16-byte row stores, all waves storing.
// ------------------------------------------------------------------------------------------------
template <typename T, int WC, int WP, int TC, int TP, int NST, int STATS = 0, bool DUAL = false>
__global__ __launch_bounds__((WC * WP + 4) * 64) void igemm_ws_kernel(const IgemmArgs a) {
  constexpr int E = Elem<T>::E, KC = 4 * E;
  constexpr int NPW = 4;
  constexpr int NW = WC * WP, NT = (NW + NPW) * 64;
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16;
  constexpr int NBA = BC / 16, NBB = BP / 16, NB = NBA + NBB;
  static_assert(NB % NPW == 0, "every producer issues the same number of DMAs per chunk");
  constexpr int J = NB / NPW;
  constexpr int BUF = 4 * (BC + BP);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  int* ltap = reinterpret_cast<int*>(lds + NST * BUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= NW;
  const int cls = blockIdx.z;
  const int P = a.N * a.Hg * a.Wg;
  int pt, ct;
  {
    const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = id & 7, slot = id >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    ct = logical % (int)gridDim.y; pt = logical / (int)gridDim.y;
  }
  const int p_base = pt * BP, c_base = ct * BC;
  if (tid < 16) ltap[tid] = (tid < a.ntaps) ? (((int)a.taps[cls].dh[tid] << 16) | ((int)a.taps[cls].dw[tid] & 0xffff)) : 0;
  const int nchunk = a.Kpad / KC;
  const int wc = producer ? 0 : wave / WP, wpi = producer ? 0 : wave - wc * WP;
  const int blkA0 = wc * TC, blkB0 = producer ? (1 << 20) : wpi * TP;     // producers stage nothing in the epilogue

  f32x4 acc[TC][TP];   // consumers only: left undefined on the producer path so it holds no registers there

  __syncthreads();   // tap table visible

  if (producer) {
    const int pw = wave - NW;
    const unsigned es = sizeof(T);
    const T* x0 = reinterpret_cast<const T*>(a.x.ptr[0]);
    const T* x1 = reinterpret_cast<const T*>(a.x.ptr[1]);
    const int C0 = a.x.C[0], C1 = a.x.C[1];
    __amdgpu_buffer_rsrc_t rsW = make_rsrc(reinterpret_cast<const T*>(a.Wp) + (size_t)cls * a.wp_rows * a.Kpad, 0xFFFFFFFFu);
    __amdgpu_buffer_rsrc_t rsX0 = make_rsrc(x0, (unsigned)((size_t)a.N * a.Hin * a.Win * C0 * es));
    __amdgpu_buffer_rsrc_t rsX1 = make_rsrc(x1 ? (const void*)x1 : (const void*)x0, (unsigned)((size_t)a.N * a.Hin * a.Win * C1 * es));
    const int r = lane >> 2, g = (lane & 3) ^ rb_swz(lane >> 2);
    // row block b = pw + NPW*j: weight block b (b < NBA) or pixel block b - NBA
    unsigned wvo[J];
    int pn[J], pbh[J], pbw[J];
    bool pok[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int b = pw + NPW * j;
      wvo[j] = (unsigned)(((c_base + b * 16 + r) * KC + g * E) * es);
      const int pidx = p_base + (b - NBA) * 16 + r;
      pok[j] = b >= NBA && pidx < P;
      const int hw = a.Hg * a.Wg;
      const int pc = pok[j] ? pidx : 0;
      const int n = pc / hw, rem = pc - n * hw, q = rem / a.Wg;
      pn[j] = n * a.Hin; pbh[j] = q * a.sh; pbw[j] = (rem - q * a.Wg) * a.sw;
    }
    const unsigned wstep = (unsigned)(a.wp_rows * KC * es);
    unsigned wso = 0, xso = 0;
    int left = 0, tap = 0, src = 0;
    bool use1 = false;
    unsigned xvo[J];
    auto open_segment = [&]() {
      const bool tok = tap < a.ntaps;
      const int tv = ltap[tok ? tap : 0];
      const int dh = tv >> 16, dw = (int)(short)(tv & 0xffff);
      use1 = src != 0;
      const int Cs = use1 ? C1 : C0;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int ih = pbh[j] + dh, iw = pbw[j] + dw;
        const bool ok = pok[j] && tok && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
        xvo[j] = ok ? (unsigned)((((pn[j] + ih) * a.Win + iw) * Cs + g * E) * es) : DMA_OOB;
      }
      left = Cs / KC;
      xso = 0;
      if (!use1 && C1 > 0) src = 1; else { src = 0; ++tap; }
    };
    auto issue = [&](int buf) {
      uint4* la = lds + buf * BUF;
      uint4* lb = la + 4 * BC;
      if (left == 0) open_segment();
      const __amdgpu_buffer_rsrc_t rx = use1 ? rsX1 : rsX0;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int b = pw + NPW * j;
        if (b < NBA) dma16_buf(rsW, wvo[j], wso, la + b * 64);
        else dma16_buf(rx, xvo[j], xso, lb + (b - NBA) * 64);
      }
      wso += wstep;
      xso += KC * es;
      --left;
    };
#pragma unroll
    for (int d = 0; d < NST - 1; ++d) if (d < nchunk) issue(d);
    int st = 0;
    for (int kc = 0; kc < nchunk; ++kc) {
      if (kc + NST - 2 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * J) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int stn = st == 0 ? NST - 1 : st - 1;
      if (kc + NST - 1 < nchunk) issue(stn);
      st = st == NST - 1 ? 0 : st + 1;
    }
  } else {
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
      for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int st = 0;
    for (int kc = 0; kc < nchunk; ++kc) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const uint4* la = lds + st * BUF;
      if constexpr (TC <= 4) {
        mma_chunk_rb<T, TC, TP, BC, BP>(la, la + 4 * BC, blkA0, blkB0, lane, acc);
      } else {
        // 128-row wave tiles: channel fragments in groups of four (the 128 accumulator registers leave room for no more)
        const uint4* lb = la + 4 * BC;
        const int i = lane & 15, g = lane >> 4;
        const int so = i * 4 + (g ^ rb_swz(i));
        uint4 fb[TP];
#pragma unroll
        for (int t = 0; t < TP; ++t) fb[t] = lb[(blkB0 + t) * 64 + so];
#pragma unroll
        for (int h = 0; h < TC; h += 4) {
          uint4 fa[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) fa[t] = la[(blkA0 + h + t) * 64 + so];
#pragma unroll
          for (int tc = 0; tc < 4; ++tc)
#pragma unroll
            for (int tp = 0; tp < TP; ++tp) acc[h + tc][tp] = mma16<T>(fa[tc], fb[tp], acc[h + tc][tp]);
        }
      }
      st = st == NST - 1 ? 0 : st + 1;
    }
  }

#if VP_ABLATE & 8
  if (!producer) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
      for (int j = 0; j < TP; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 123.456f) reinterpret_cast<float*>(a.Y)[0] = 1.f;
  }
  return;
#endif
  constexpr int RINGB = NST * BUF * 16;
  constexpr int NPASS = epi_passes(BC, BP, WP, RINGB);
  staged_epilogue<T, TC, TP, BC, BP, NPASS, NT, STATS, 0, DUAL>(a, LinearPix{a, cls, p_base, P}, c_base, blkA0, blkB0, acc, smem, pt, cls);
}


// ------------------------------------------------------------------------------------------------
// conv_cin8_kernel: the first layers of the three nets (3- and 6-channel images padded to 8: VGG conv1_1, discriminator layer_1,
// encoder_1, encoder_fg_1; all 64 output channels, bf16).  K = taps x 8 is only 3-4 MFMA steps, so a tiled GEMM is all
// prologue and epilogue; these layers are bound by writing the output (8 channels in, 64 out).  Direct form, no LDS:
//   * a tap of a padded pixel is exactly one 16-byte piece = the 8 k values one lane feeds to mfma 16x16x32; lane (pixel i,
//     k group g) loads tap 4s+g of its pixel for MFMA step s straight from global memory (buffer load: padding reads zeros);
//   * the whole weight matrix (64 x K) sits in registers as A fragments for the life of the wave (S x 4 x 4 VGPRs);
//   * MFMA row (tile t, 4q+e) is channel 32*(t>>1) + 8q + 4*(t&1) + e, so after the 4 tiles a lane holds channels 8q..8q+7 and
//     32+8q..32+8q+7 of its pixel: two 16-byte stores per lane, and the four lanes of a pixel write 64 contiguous bytes per store;
//   * each wave walks 16-pixel tiles with the next tile's pieces in flight (double-buffered fragments).
// ------------------------------------------------------------------------------------------------
// Round 6: the tile loop is BRANCH-FREE.  The round-1 form guarded every load / finish / store by `tile < ntile`, a run-time switch on
// out_act and null checks of the three output pointers: 8175 lines of ISA, and - what cost the time - hipcc's s_waitcnt pass, which merges
// the pending-operation state at every join, put `vmcnt(2)` behind the loads of tile t + 2: a wave drained its previous tile's stores
// and the loads of tile t + 1 in every iteration (ablation, profiles/r06_cin8_ablation.txt: conv1_1 0.178 ms = 0.053 instruction stream +
// 0.12 stores, not overlapped: 3.1 TB/s with 16 waves x 2 KB of stores in flight per CU).  Now: tile indices are clamped to the wave's last
// tile (a wave past its end recomputes and re-stores that tile: same bytes), the pixel count is a multiple of 16 (eligibility), out_act is
// NONE or RELU as a floor value, which outputs exist is a template parameter (OUTS: 1 raw, 2 lrelu copy, 4 relu copy), and the four
// fragment sets rotate through a loop unrolled by four: the compiler's own counts come out exact (the loads of tile t wait with the
// stores of the previous tiles and the loads of three tiles still in flight).
// (Forcing five / six waves per SIMD with amdgpu_waves_per_eu - the kernel allocates 104-124 registers, four / three waves - spills: conv1_1 0.155
// -> 0.155 / 0.253 ms, the stride-2 layers 0.093 -> 0.117 / 0.183: profiles/r06_cin8_ablation.txt.)
template <int S, int OUTS>
__global__ __launch_bounds__(256) void conv_cin8_kernel(const IgemmArgs a, int lgW, int lgH) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, g = lane >> 4;
  const int P = a.N << (lgW + lgH);
  const int ntile = P >> 4;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;

  // A fragments: packed weights are [K chunk s][row][32 k]; this lane's row of tile t is channel 32*(t>>1) + 8*(i>>2) + 4*(t&1) + (i&3).
  // They live in LDS in fragment order [s][t][lane] (each lane re-reads its own 16 bytes: conflict-free, 12-16 KB per block),
  // which leaves the registers to occupancy and to the pixel pieces in flight.
  __shared__ uint4 wfrag[S * 4 * 64];
  __shared__ uint4 otile[4 * 16 * 144 / 16];
  // output rows are dense ([pixel][64]) and the pixel grid is the output grid: a tile's 16 pixels are one 2 KB run in every output
  // (measured: conv1_1, stride 1, 64 images: 0.209 -> 0.178 ms in round 5, where the stride-2 first layers lost 8 us each with it; on the
  // branch-free loop of round 6 they gain: layer_1 0.093 -> 0.089 ms, encoder_1 0.046 -> 0.044 - one store path for every output)
  constexpr bool ACTS = (OUTS & 6) != 0;
  {
    const bf16* wp = reinterpret_cast<const bf16*>(a.Wp);
    for (int idx = threadIdx.x; idx < S * 4 * 64; idx += 256) {
      const int l = idx & 63, t = (idx >> 6) & 3, s = idx >> 8;
      // with the global row permutation this is simply packed row 16t + i
      const int row = a.rowperm ? t * 16 + (l & 15) : (t >> 1) * 32 + 8 * ((l & 15) >> 2) + (t & 1) * 4 + (l & 3);
      wfrag[idx] = *reinterpret_cast<const uint4*>(wp + ((size_t)s * a.wp_rows + row) * 32 + (l >> 4) * 8);
    }
    __syncthreads();
  }
  float bias[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bias[e] = a.bias ? a.bias[(e >> 3) * 32 + 8 * g + (e & 7)] : 0.f;
  const float act_floor = a.out_act == ACT_RELU ? 0.f : -__builtin_inff();     // relu as a floor: no branch per element
  // this lane's tap of step s
  int tdh[S], tdw[S];
  bool tok[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int tap = 4 * s + g;
    tok[s] = tap < a.ntaps;
    int dh = 0, dw = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) if (t == tap) { dh = a.taps[0].dh[t]; dw = a.taps[0].dw[t]; }
    tdh[s] = dh; tdw[s] = dw;
  }
  __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * 16));
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  // a wave's tiles: wave_global, + nwave, ...; indices past its last one are clamped to it
  const int my_n = wave_global < ntile ? (ntile - 1 - wave_global) / nwave + 1 : 0;
  if (my_n == 0) return;
  const int tile_last = wave_global + (my_n - 1) * nwave;

  auto load_tile = [&](int tile_, uint4 (&fb)[S]) {
    const int tile = tile_ < tile_last ? tile_ : tile_last;
    const int p = tile * 16 + i;
    const int ow = p & ((1 << lgW) - 1), oh = (p >> lgW) & ((1 << lgH) - 1), n = p >> (lgW + lgH);
    const int bh = oh * a.sh, bw = ow * a.sw;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int ih = bh + tdh[s], iw = bw + tdw[s];
      const bool ok = tok[s] && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      const unsigned off = ok ? (unsigned)(((n * a.Hin + ih) * a.Win + iw) * 16) : DMA_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)off, 0, 0);
      fb[s] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  auto finish_tile = [&](int tile_, const uint4 (&fb)[S]) {
    const int tile = tile_ < tile_last ? tile_ : tile_last;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int wl = lane;
    asm volatile("" : "+v"(wl));      // opaque: keeps the weight fragments in LDS (hoisted into registers they cost 48-64 VGPRs)
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mma16<bf16>(wfrag[(s * 4 + t) * 64 + wl], fb[s], acc[t]);
    const int p = tile * 16 + i;
    float lo[8], hi[8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = fmaxf(acc[t][e] + bias[4 * t + e], act_floor);
        if (t < 2) lo[4 * t + e] = x; else hi[4 * (t - 2) + e] = x;
      }
    const uint4 plo = Elem<bf16>::pack(lo), phi = Elem<bf16>::pack(hi);
    // the 16 pixels of a tile are 2 KB of consecutive output (dense [pixel][64] rows, pixel grid == output grid: eligibility); a lane's own
    // two pieces are 64-byte segments 128 bytes apart (half cache lines per store instruction).  Transpose through a wave-private LDS
    // tile (pixel pitch 144 bytes: conflict-free both ways) so that each of the two store instructions writes one contiguous 1 KB run
    char* tb = reinterpret_cast<char*>(otile) + (threadIdx.x >> 6) * (16 * 144);
    auto store_run = [&](void* dst, const uint4& v0, const uint4& v1) {
      *reinterpret_cast<uint4*>(tb + i * 144 + g * 16) = v0;
      *reinterpret_cast<uint4*>(tb + i * 144 + 64 + g * 16) = v1;
      __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the wave's own LDS writes have landed (same-wave, no barrier)
      __builtin_amdgcn_wave_barrier();
      const uint4 q0 = *reinterpret_cast<const uint4*>(tb + (lane >> 3) * 144 + (lane & 7) * 16);
      const uint4 q1 = *reinterpret_cast<const uint4*>(tb + (8 + (lane >> 3)) * 144 + (lane & 7) * 16);
      bf16* yt = reinterpret_cast<bf16*>(dst) + (size_t)tile * 16 * 64;
      reinterpret_cast<uint4*>(yt)[lane] = q0;
      reinterpret_cast<uint4*>(yt)[64 + lane] = q1;
      __builtin_amdgcn_wave_barrier();
    };
    if constexpr (OUTS & 1) store_run(a.Y, plo, phi);
    // the consumers' activations of the ROUNDED output (what act_apply computes from the stored tensor: same bits)
    if constexpr (ACTS) {
      float rl[8], rh[8];
      Elem<bf16>::unpack(plo, rl);
      Elem<bf16>::unpack(phi, rh);
      if constexpr (OUTS & 2) {
        float t0[8], t1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { t0[e] = act_apply(ACT_LRELU, rl[e]); t1[e] = act_apply(ACT_LRELU, rh[e]); }
        store_run(a.xa_lrelu, Elem<bf16>::pack(t0), Elem<bf16>::pack(t1));
      }
      if constexpr (OUTS & 4) {
        float t0[8], t1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { t0[e] = act_apply(ACT_RELU, rl[e]); t1[e] = act_apply(ACT_RELU, rh[e]); }
        store_run(a.xa_relu, Elem<bf16>::pack(t0), Elem<bf16>::pack(t1));
      }
    }
  };

  // four fragment sets: three tiles of loads stay in flight behind the tile being finished; trips of four tiles, no branch inside
  // (four sets, three tiles of loads in flight: the stride-2 first layers 0.053 / 0.043 -> 0.045 / 0.038 ms against three sets; conv1_1 +-0)
  uint4 fb0[S], fb1[S], fb2[S], fb3[S];
  int tile = wave_global;
  load_tile(tile, fb0);
  load_tile(tile + nwave, fb1);
  load_tile(tile + 2 * nwave, fb2);
  for (int trip = (my_n + 3) / 4; trip > 0; --trip) {
    load_tile(tile + 3 * nwave, fb3);
    finish_tile(tile, fb0);
    load_tile(tile + 4 * nwave, fb0);
    finish_tile(tile + nwave, fb1);
    load_tile(tile + 5 * nwave, fb1);
    finish_tile(tile + 2 * nwave, fb2);
    load_tile(tile + 6 * nwave, fb2);
    finish_tile(tile + 3 * nwave, fb3);
    tile += 4 * nwave;
  }
}

// ------------------------------------------------------------------------------------------------
// deconv_cout4_kernel: the generator's last layer (decoder_1: 4x4 stride-2 transposed conv, Cin -> 4 channels, f32 output).
// A tiled GEMM wastes 15/16 of its rows on 4 channels and re-launches per parity class.  Here the 4 classes x 4 channels ARE the
// 16 rows of one MFMA tile: K runs over the 3x3 input neighbourhood of a base pixel (the union of the four classes' 2x2 taps;
// a class's unused taps are zero rows of the weight image), columns are 16 consecutive base pixels.  After the K loop lane
// (pixel i, group g) holds the 4 channels of output pixel (2q + g/2, 2r + g%2): one 16-byte f32 store, the four groups of a
// base-pixel run fill two contiguous 512-byte output rows.  Weights: fragment-ordered LDS image built once per block; pixel
// pieces straight from global memory (buffer loads, zeros outside), three (tap) units in flight per wave.
// ------------------------------------------------------------------------------------------------
template <int SPT>   // MFMA steps (32 channels each) per tap = Cin / 32
__global__ __launch_bounds__(256) void deconv_cout4_kernel(const IgemmArgs a, int lgW, int lgH) {
  constexpr int S = 9 * SPT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* wfrag = reinterpret_cast<uint4*>(smem);       // [S][64]
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, g = lane >> 4;
  const int Cin = SPT * 32;
  {
    const bf16* wp = reinterpret_cast<const bf16*>(a.Wp);
    const int nchunk_c = a.Kpad / 32;
    for (int idx = threadIdx.x; idx < S * 64; idx += 256) {
      const int l = idx & 63, s = idx >> 6;
      const int u = s / SPT, c0 = (s % SPT) * 32 + (l >> 4) * 8;
      const int dy = u / 3 - 1, dx = u % 3 - 1;
      const int cls = (l & 15) >> 2, co = l & 3;
      uint4 v = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int tdh = 0, tdw = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c == cls) { tdh = a.taps[c].dh[t]; tdw = a.taps[c].dw[t]; }
        if (tdh == dy && tdw == dx) {
          const int k = t * Cin + c0;
          v = *reinterpret_cast<const uint4*>(wp + (((size_t)cls * nchunk_c + (k >> 5)) * a.wp_rows + co) * 32 + (k & 31));
        }
      }
      wfrag[idx] = v;
    }
    __syncthreads();
  }
  const int P = a.N << (lgW + lgH);                     // base pixels (input grid)
  const int ntile = (P + 15) >> 4;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6), nwave = gridDim.x * 4;
  const int C0 = a.x.C[0], C1 = a.x.C[1];
  __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * C0 * 2));
  __amdgpu_buffer_rsrc_t rs1 = make_rsrc(a.x.ptr[1] ? a.x.ptr[1] : a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * C1 * 2));
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  float bias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bias[e] = a.bias ? a.bias[e] : 0.f;

  // loads of unit (tile, u): SPT pieces of this lane's pixel at tap u
  auto load_unit = [&](int tile, int u, uint4 (&f)[SPT]) {
    const int p = tile * 16 + i;
    const int r = p & ((1 << lgW) - 1), q = (p >> lgW) & ((1 << lgH) - 1), n = p >> (lgW + lgH);
    const int ih = q + u / 3 - 1, iw = r + u % 3 - 1;
    const bool ok = p < P && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
    const int pix = (n * a.Hin + ih) * a.Win + iw;
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
      const int c = k * 32 + g * 8;
      const bool s1 = c >= C0;                          // uniform per k when C0 is a multiple of 32
      const unsigned off = ok ? (unsigned)((pix * (s1 ? C1 : C0) + (s1 ? c - C0 : c)) * 2) : DMA_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(s1 ? rs1 : rs0, (int)off, 0, 0);
      f[k] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };

  uint4 f0[SPT], f1[SPT], f2[SPT];
  int tile = wave_global;
  if (tile < ntile) { load_unit(tile, 0, f0); load_unit(tile, 1, f1); }
  while (tile < ntile) {
    const int nxt = tile + nwave;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    int wl = lane;
    asm volatile("" : "+v"(wl));                        // keep the weight image in LDS (no hoisting into registers)
    auto consume = [&](int u, const uint4 (&f)[SPT]) {
#pragma unroll
      for (int k = 0; k < SPT; ++k) acc = mma16<bf16>(wfrag[(u * SPT + k) * 64 + wl], f[k], acc);
    };
    // units 0..8 rotate through the three fragment sets; unit u+2 is loaded before unit u is consumed
    load_unit(tile, 2, f2); consume(0, f0);
    load_unit(tile, 3, f0); consume(1, f1);
    load_unit(tile, 4, f1); consume(2, f2);
    load_unit(tile, 5, f2); consume(3, f0);
    load_unit(tile, 6, f0); consume(4, f1);
    load_unit(tile, 7, f1); consume(5, f2);
    load_unit(tile, 8, f2); consume(6, f0);
    if (nxt < ntile) load_unit(nxt, 0, f0);
    consume(7, f1);
    if (nxt < ntile) load_unit(nxt, 1, f1);
    consume(8, f2);
    const int p = tile * 16 + i;
    if (p < P) {
      const int r = p & ((1 << lgW) - 1), q = (p >> lgW) & ((1 << lgH) - 1), n = p >> (lgW + lgH);
      float* yp = reinterpret_cast<float*>(a.Y) + ((size_t)(n * a.Hof + 2 * q + (g >> 1)) * a.Wof + 2 * r + (g & 1)) * 4;
      *reinterpret_cast<float4*>(yp) = make_float4(acc[0] + bias[0], acc[1] + bias[1], acc[2] + bias[2], acc[3] + bias[3]);
    }
    tile = nxt;
  }
}

// deconv_cout4_tile_kernel: the same layer with the input staged ONCE per block.  deconv_cout4_kernel fetches every input pixel nine
// times (once per tap of the 3x3 neighbourhood, from whichever lane needs it): 1.2 GB of L1 / L2 traffic for a 134 MB tensor, 0.145 ms
// against 0.02 ms of HBM time (r02 layer table).  Here a block owns 4 rows x 16 columns of base pixels: its 256 threads load the
// 6 x 18 pixel halo tile (1.7x the interior) with 16-byte loads - one tile ahead, in registers, while the current tile computes - and
// store it to LDS at a padded pixel pitch (Cin * 2 + 16 bytes: the 16 lanes of a fragment read hit 64 distinct banks); wave w then
// builds the B fragments of row w for all nine taps from LDS.  Same MFMA tile, weight image and output mapping as above.
template <int SPT, int HALVED>
__global__ __launch_bounds__(256) void deconv_cout4_tile_kernel(const IgemmArgs a, int lgW, int lgH) {
  constexpr int S = 9 * SPT, CIN = SPT * 32, PPP = CIN / 8;       // 16-byte pieces per pixel
  constexpr int PIXB = CIN * 2 + 16, TPX = 6 * 18, NPIECE = TPX * PPP, NJ = (NPIECE + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* wfrag = reinterpret_cast<uint4*>(smem);                  // [S][64]
  char* stage = smem + (size_t)S * 64 * 16;                       // [6][18][PIXB], then 256 dummy 16-byte slots
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  {
    const bf16* wp = reinterpret_cast<const bf16*>(a.Wp);
    const int nchunk_c = a.Kpad / 32;
    for (int idx = threadIdx.x; idx < S * 64; idx += 256) {
      const int l = idx & 63, s = idx >> 6;
      const int u = s / SPT, c0 = (s % SPT) * 32 + (l >> 4) * 8;
      const int dy = u / 3 - 1, dx = u % 3 - 1;
      const int cls = (l & 15) >> 2, co = l & 3;
      uint4 v = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int tdh = 0, tdw = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c == cls) { tdh = a.taps[c].dh[t]; tdw = a.taps[c].dw[t]; }
        if (tdh == dy && tdw == dx) {
          const int k = t * CIN + c0;
          v = *reinterpret_cast<const uint4*>(wp + (((size_t)cls * nchunk_c + (k >> 5)) * a.wp_rows + co) * 32 + (k & 31));
        }
      }
      wfrag[idx] = v;
    }
  }
  const int tw = 1 << (lgW - 4), th = 1 << (lgH - 2);             // tiles per row / per column of one image
  const int ntile = a.N * tw * th;
  const int my_n = (int)blockIdx.x < ntile ? (ntile - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  if (my_n == 0) return;
  const int tile_last = blockIdx.x + (my_n - 1) * gridDim.x;
  // One source per wave: with two equally wide concatenated sources (HALVED: the decoder's skip connection) waves 0-1 fetch the first
  // one's channels and waves 2-3 the second's, so that the buffer descriptor is a scalar select and every piece is ONE load; otherwise
  // (HALVED == 0) there is a single source.
  const int src = HALVED ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7)) : 0;
  const int CS = HALVED ? CIN / 2 : CIN;                          // channels of one source
  constexpr int PPS = HALVED ? PPP / 2 : PPP, NTH = HALVED ? 128 : 256;
  __amdgpu_buffer_rsrc_t rs0 = make_rsrc(src ? a.x.ptr[1] : a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * CS * 2));
  __amdgpu_buffer_rsrc_t rsY = make_rsrc(a.Y, (unsigned)((size_t)a.N * a.Hof * a.Wof * 4 * 4));
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  float bias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bias[e] = a.bias ? a.bias[e] : 0.f;

  // this thread's pieces of a halo tile: (pixel slot, 8-channel group) -> LDS byte offset, channel offset inside the source
  int soff[NJ], spix_r[NJ], spix_c[NJ], sch[NJ];
  bool sok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int idx = (int)(threadIdx.x & (NTH - 1)) + NTH * j;
    const int px = idx / PPS, c = (idx - px * PPS) * 8;
    spix_r[j] = px / 18; spix_c[j] = px - spix_r[j] * 18;
    sch[j] = c;
    sok[j] = idx < TPX * PPS;
    soff[j] = sok[j] ? px * PIXB + (src * CS + c) * 2 : TPX * PIXB + (int)threadIdx.x * 16;   // (beyond the halo: the thread's dummy slot)
  }
  uint4 pre[NJ];
  // The tile loop is branch-free (see conv3x3_cout8_tile_kernel): tiles beyond the block's last one are clamped to it, threads beyond the
  // halo write a dummy LDS slot.
  auto load_tile = [&](int tile_) {
    const int tile = tile_ < tile_last ? tile_ : tile_last;
    const bool live = tile_ <= tile_last;                         // (beyond the block's last tile: every offset out of range, no data moves)
    const int tc = tile & (tw - 1), tr = (tile >> (lgW - 4)) & (th - 1), n = tile >> (lgW - 4 + lgH - 2);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ih = tr * 4 - 1 + spix_r[j], iw = tc * 16 - 1 + spix_c[j];
      const bool ok = live && sok[j] && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      const unsigned off = ok ? (unsigned)((((n * a.Hin + ih) * a.Win + iw) * CS + sch[j]) * 2) : DMA_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)off, 0, 0);
      pre[j] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  int tile = blockIdx.x;
  load_tile(tile);
  __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, rsY, (int)DMA_OOB, 0, 0);   // (see conv3x3_cout8_tile_kernel)
  for (int it = my_n; it > 0; --it) {
    __syncthreads();                                              // previous tile's fragment reads are done (first pass: wfrag is complete)
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<uint4*>(stage + soff[j]) = pre[j];
    __syncthreads();
    load_tile(tile + gridDim.x);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    int wl = lane;
    asm volatile("" : "+v"(wl));                                  // keep the weight image in LDS (no hoisting into registers)
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const char* px = stage + ((wv + u / 3) * 18 + i + u % 3) * PIXB + g * 16;
#pragma unroll
      for (int k = 0; k < SPT; ++k)
        acc = mma16<bf16>(wfrag[(u * SPT + k) * 64 + wl], *reinterpret_cast<const uint4*>(px + k * 64), acc);
    }
    const int tc = tile & (tw - 1), tr = (tile >> (lgW - 4)) & (th - 1), n = tile >> (lgW - 4 + lgH - 2);
    const int q = tr * 4 + wv, r = tc * 16 + i;
    const unsigned yo = (unsigned)(((n * a.Hof + 2 * q + (g >> 1)) * a.Wof + 2 * r + (g & 1)) * 16);
    const float4 o = make_float4(acc[0] + bias[0], acc[1] + bias[1], acc[2] + bias[2], acc[3] + bias[3]);
    __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)}, rsY, (int)yo, 0, 0);
    tile += gridDim.x;
  }
}

// conv3x3_cout8_tile_kernel: 3x3 stride-1 convolution from 64 channels to <= 8 (VGG conv1_1 backward-data: the perceptual gradient
// arriving at the composited image).  As a 16-row GEMM tile on the gather-per-tap kernel every dY pixel is fetched nine times
// (0.193 ms for 0.04 ms of HBM traffic, r02 layer table).  Same plan as deconv_cout4_tile_kernel: a block owns 4 rows x 16 columns,
// stages the 6 x 18 pixel halo once in LDS (one tile ahead in registers), wave w multiplies row w against the weight fragments -
// which are only 18 x 16 bytes per lane and stay in registers.  Rows 0..7 of the MFMA tile are the channels: lanes g = 0 / 1 hold
// channels 0..3 / 4..7 of pixel i, one cross-lane move joins them into the 16-byte output row (epi_store8).
// Round 6: branch-free tile loop (as conv_cin8_kernel, EXPERIMENTS.md 0.7): the block's tile index is clamped to its last tile, the LDS
// staging writes of the threads beyond the halo's 864 pieces go to a dummy slot, the output store is a buffer store whose offset is out of
// range for the lanes that hold no output piece (dropped by the hardware), and the epilogue is the plain one (no bias / activation /
// reference / accumulation: eligibility) - so hipcc's s_waitcnt pass counts exactly and the tile's store stays in flight across the next
// tile's barrier instead of being drained by a vmcnt(0) in front of it.
__global__ __launch_bounds__(256) void conv3x3_cout8_tile_kernel(const IgemmArgs a, int lgW, int lgH) {
  constexpr int PIXB = 64 * 2 + 16, TPX = 6 * 18, NPIECE = TPX * 8, NJ = (NPIECE + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* stage = smem;                                             // [6][18][PIXB], then 256 dummy 16-byte slots
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  uint4 af[18];
  int toff[9];
  {
    const bf16* wp = reinterpret_cast<const bf16*>(a.Wp);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      toff[t] = ((1 + a.taps[0].dh[t]) * 18 + 1 + a.taps[0].dw[t]) * PIXB;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int kk = t * 64 + k * 32 + g * 8;
        af[t * 2 + k] = *reinterpret_cast<const uint4*>(wp + ((size_t)(kk >> 5) * a.wp_rows + i) * 32 + (kk & 31));
      }
    }
  }
  const int tw = 1 << (lgW - 4), th = 1 << (lgH - 2);
  const int ntile = a.N * tw * th;
  const int my_n = (int)blockIdx.x < ntile ? (ntile - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  if (my_n == 0) return;
  const int tile_last = blockIdx.x + (my_n - 1) * gridDim.x;
  __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * 64 * 2));
  __amdgpu_buffer_rsrc_t rsY = make_rsrc(a.Y, (unsigned)((size_t)a.N * a.Hof * a.Wof * a.ldY * 2));
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  int soff[NJ], spix_r[NJ], spix_c[NJ], sch[NJ];
  bool sok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int idx = threadIdx.x + 256 * j;
    const int px = idx >> 3, c = (idx & 7) * 8;
    spix_r[j] = px / 18; spix_c[j] = px - spix_r[j] * 18;
    sch[j] = c;
    sok[j] = idx < NPIECE;
    soff[j] = sok[j] ? px * PIXB + c * 2 : TPX * PIXB + (int)threadIdx.x * 16;      // (beyond the halo: the thread's dummy slot)
  }
  uint4 pre[NJ];
  auto load_tile = [&](int tile_) {
    const int tile = tile_ < tile_last ? tile_ : tile_last;
    const bool live = tile_ <= tile_last;                         // (beyond the block's last tile: every offset out of range, no data moves)
    const int tc = tile & (tw - 1), tr = (tile >> (lgW - 4)) & (th - 1), n = tile >> (lgW - 4 + lgH - 2);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ih = tr * 4 - 1 + spix_r[j], iw = tc * 16 - 1 + spix_c[j];
      const bool ok = live && sok[j] && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      const unsigned off = ok ? (unsigned)((((n * a.Hin + ih) * a.Win + iw) * 64 + sch[j]) * 2) : DMA_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)off, 0, 0);
      pre[j] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  int tile = blockIdx.x;
  load_tile(tile);
  // (a store that goes nowhere - offset out of range - behind the first tile's loads: the loop's first trip then looks like every other one
  // to the s_waitcnt pass, which otherwise merges "no store pending" with "one store pending" into vmcnt(0) at the loop header)
  __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, rsY, (int)DMA_OOB, 0, 0);
  for (int it = my_n; it > 0; --it) {
    __syncthreads();                                              // previous tile's fragment reads are done
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<uint4*>(stage + soff[j]) = pre[j];
    __syncthreads();
    load_tile(tile + gridDim.x);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const char* px = stage + (wv * 18 + i) * PIXB + g * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k) acc = mma16<bf16>(af[t * 2 + k], *reinterpret_cast<const uint4*>(px + toff[t] + k * 64), acc);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = acc[e]; v[4 + e] = __shfl(acc[e], (lane + 16) & 63); }
    const int tl = tile < tile_last ? tile : tile_last;
    const int tc = tl & (tw - 1), tr = (tl >> (lgW - 4)) & (th - 1), n = tl >> (lgW - 4 + lgH - 2);
    const unsigned yo = g == 0 ? (unsigned)((((n * a.Hof + tr * 4 + wv) * a.Wof + tc * 16 + i) * a.ldY) * 2) : DMA_OOB;
    const uint4 pk = Elem<bf16>::pack(v);
    __builtin_amdgcn_raw_buffer_store_b128((u32x4){pk.x, pk.y, pk.z, pk.w}, rsY, (int)yo, 0, 0);
    tile += gridDim.x;
  }
}

// deconv_cout8_tile_kernel: 4x4 stride-2 transposed conv to <= 8 channels (discriminator layer_1 backward-data towards the generator:
// 64 -> 6 (+2 pad) channels at 256x256).  deconv_cout4_tile_kernel with two MFMA tiles: rows 16T .. 16T+15 = parity classes 2T, 2T+1
// x 8 channels; lane (i, g) of tile T holds channels 4 (g & 1) .. +3 of class 2T + (g >> 1) at base pixel i, its neighbour group
// g ^ 1 the other half: one cross-lane move, then the even groups write the 16-byte bf16 rows (epi_store8).
template <int SPT>
__global__ __launch_bounds__(256) void deconv_cout8_tile_kernel(const IgemmArgs a, int lgW, int lgH) {
  constexpr int S = 9 * SPT, CIN = SPT * 32, PPP = CIN / 8;
  constexpr int PIXB = CIN * 2 + 16, TPX = 6 * 18, NPIECE = TPX * PPP, NJ = (NPIECE + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* wfrag = reinterpret_cast<uint4*>(smem);                  // [2][S][64]
  char* stage = smem + (size_t)2 * S * 64 * 16;                   // [6][18][PIXB], then 256 dummy 16-byte slots
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  {
    const bf16* wp = reinterpret_cast<const bf16*>(a.Wp);
    const int nchunk_c = a.Kpad / 32;
    for (int idx = threadIdx.x; idx < 2 * S * 64; idx += 256) {
      const int l = idx & 63, s = (idx >> 6) % S, T = idx / (S * 64);
      const int u = s / SPT, c0 = (s % SPT) * 32 + (l >> 4) * 8;
      const int dy = u / 3 - 1, dx = u % 3 - 1;
      const int cls = 2 * T + ((l & 15) >> 3), co = l & 7;
      uint4 v = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int tdh = 0, tdw = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c == cls) { tdh = a.taps[c].dh[t]; tdw = a.taps[c].dw[t]; }
        if (tdh == dy && tdw == dx) {
          const int k = t * CIN + c0;
          v = *reinterpret_cast<const uint4*>(wp + (((size_t)cls * nchunk_c + (k >> 5)) * a.wp_rows + co) * 32 + (k & 31));
        }
      }
      wfrag[idx] = v;
    }
  }
  const int tw = 1 << (lgW - 4), th = 1 << (lgH - 2);
  const int ntile = a.N * tw * th;
  const int my_n = (int)blockIdx.x < ntile ? (ntile - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  if (my_n == 0) return;
  const int tile_last = blockIdx.x + (my_n - 1) * gridDim.x;
  __amdgpu_buffer_rsrc_t rs0 = make_rsrc(a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * CIN * 2));
  __amdgpu_buffer_rsrc_t rsY = make_rsrc(a.Y, (unsigned)((size_t)a.N * a.Hof * a.Wof * a.ldY * 2));
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  int soff[NJ], spix_r[NJ], spix_c[NJ], sch[NJ];
  bool sok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int idx = threadIdx.x + 256 * j;
    const int px = idx / PPP, c = (idx - px * PPP) * 8;
    spix_r[j] = px / 18; spix_c[j] = px - spix_r[j] * 18;
    sch[j] = c;
    sok[j] = idx < NPIECE;
    soff[j] = sok[j] ? px * PIXB + c * 2 : TPX * PIXB + (int)threadIdx.x * 16;      // (beyond the halo: the thread's dummy slot)
  }
  uint4 pre[NJ];
  // The tile loop is branch-free (see conv3x3_cout8_tile_kernel): tiles beyond the block's last one are clamped to it, threads beyond the
  // halo write a dummy LDS slot, lanes without an output row store to an out-of-range offset.
  auto load_tile = [&](int tile_) {
    const int tile = tile_ < tile_last ? tile_ : tile_last;
    const bool live = tile_ <= tile_last;                         // (beyond the block's last tile: every offset out of range, no data moves)
    const int tc = tile & (tw - 1), tr = (tile >> (lgW - 4)) & (th - 1), n = tile >> (lgW - 4 + lgH - 2);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ih = tr * 4 - 1 + spix_r[j], iw = tc * 16 - 1 + spix_c[j];
      const bool ok = live && sok[j] && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      const unsigned off = ok ? (unsigned)((((n * a.Hin + ih) * a.Win + iw) * CIN + sch[j]) * 2) : DMA_OOB;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)off, 0, 0);
      pre[j] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  int tile = blockIdx.x;
  load_tile(tile);
  __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, rsY, (int)DMA_OOB, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b128((u32x4){0u, 0u, 0u, 0u}, rsY, (int)DMA_OOB, 0, 0);
  for (int it = my_n; it > 0; --it) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<uint4*>(stage + soff[j]) = pre[j];
    __syncthreads();
    load_tile(tile + gridDim.x);
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    int wl = lane;
    asm volatile("" : "+v"(wl));                                  // keep the weight image in LDS
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const char* px = stage + ((wv + u / 3) * 18 + i + u % 3) * PIXB + g * 16;
#pragma unroll
      for (int k = 0; k < SPT; ++k) {
        const uint4 b = *reinterpret_cast<const uint4*>(px + k * 64);
        acc0 = mma16<bf16>(wfrag[(u * SPT + k) * 64 + wl], b, acc0);
        acc1 = mma16<bf16>(wfrag[(S + u * SPT + k) * 64 + wl], b, acc1);
      }
    }
    const int tl = tile < tile_last ? tile : tile_last;
    const int tc = tl & (tw - 1), tr = (tl >> (lgW - 4)) & (th - 1), n = tl >> (lgW - 4 + lgH - 2);
    const int q = tr * 4 + wv, r = tc * 16 + i;
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      const f32x4 acc = T ? acc1 : acc0;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[e]; v[4 + e] = __shfl(acc[e], (lane + 16) & 63); }
      const int cls = 2 * T + (g >> 1);
      const unsigned yo = (g & 1) == 0 ? (unsigned)((((n * a.Hof + 2 * q + (cls >> 1)) * a.Wof + 2 * r + (cls & 1)) * a.ldY) * 2) : DMA_OOB;
      const uint4 pk = Elem<bf16>::pack(v);
      __builtin_amdgcn_raw_buffer_store_b128((u32x4){pk.x, pk.y, pk.z, pk.w}, rsY, (int)yo, 0, 0);
    }
    tile += gridDim.x;
  }
}

// sums the split-K slabs in a fixed order (deterministic) and applies the igemm epilogue
template <typename T, bool DUAL = false>
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(const IgemmArgs a) {
  const int P = a.N * a.Hg * a.Wg;
  const int cq = a.CoutPad >> 2;
  const size_t total = (size_t)a.nclass * P * cq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c0 = (int)(i % cq) * 4;
    const size_t t = i / cq;
    const int pidx = (int)(t % P);
    const int cls = (int)(t / P);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    // the slabs are added in slab order (deterministic, same sums as a one-load-per-trip loop); eight loads are in flight at a
    // time - a thread's slabs lie P * CoutPad floats apart, the pass is pure load latency otherwise
    const float* src = a.partial + (((size_t)cls * a.splitk * P + pidx) * a.CoutPad + c0);
    const size_t sstride = (size_t)P * a.CoutPad;
    for (int s0 = 0; s0 < a.splitk; s0 += 8) {
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = s0 + u < a.splitk ? *reinterpret_cast<const float4*>(src + (size_t)(s0 + u) * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (s0 + u >= a.splitk) break;
        v[0] += x[u].x; v[1] += x[u].y; v[2] += x[u].z; v[3] += x[u].w;
      }
    }
    igemm_epilogue<T, DUAL>(a, cls, pidx, c0, v);
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad_kernel: grid = (M tiles over (tap, g), N tiles over d, splitk over pixels)
//   A operand rows = (tap, g channel) of the gathered tensor, B operand cols = d channel of the
//   dense tensor, K = pixels.  A loader task = E consecutive pixels x E consecutive channels,
//   transposed in registers into E 16-byte pieces "one channel, E consecutive pixels".
// ------------------------------------------------------------------------------------------------
template <typename T> struct Transposer;
template <> struct Transposer<float> {
  __device__ static __forceinline__ void run(const uint4 (&in)[4], uint4 (&out)[4]) {
    out[0] = make_uint4(in[0].x, in[1].x, in[2].x, in[3].x);
    out[1] = make_uint4(in[0].y, in[1].y, in[2].y, in[3].y);
    out[2] = make_uint4(in[0].z, in[1].z, in[2].z, in[3].z);
    out[3] = make_uint4(in[0].w, in[1].w, in[2].w, in[3].w);
  }
};
template <> struct Transposer<bf16> {
  // in[p] = 8 channels of pixel p (dword d = channels 2d, 2d+1); out[c] = 8 pixels of channel c
  __device__ static __forceinline__ uint32_t lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
  __device__ static __forceinline__ uint32_t hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
  __device__ static __forceinline__ void run(const uint4 (&in)[8], uint4 (&out)[8]) {
#define VP_TR(dw, c0)                                                                                 \
    out[c0]     = make_uint4(lo(in[0].dw, in[1].dw), lo(in[2].dw, in[3].dw), lo(in[4].dw, in[5].dw), lo(in[6].dw, in[7].dw)); \
    out[c0 + 1] = make_uint4(hi(in[0].dw, in[1].dw), hi(in[2].dw, in[3].dw), hi(in[4].dw, in[5].dw), hi(in[6].dw, in[7].dw));
    VP_TR(x, 0) VP_TR(y, 2) VP_TR(z, 4) VP_TR(w, 6)
#undef VP_TR
  }
};

// One loader task: E consecutive pixels x E consecutive channels of ONE source tensor, fixed for the whole
// K loop (which tensor, which channels, which tap); only the pixel window moves.
template <typename T> struct WgTask {
  const T* base;      // source pointer + channel offset (plain path)
  int C;              // channels of that source (pixel stride)
  int s, dh, dw;      // source pixel = (q*s + dh, r*s + dw)
  int Hs, Ws;
  int ch;             // channel index inside the PixSrc (prologue path)
  bool ok;
};

template <typename T, int E, int KCH, bool PLAIN>
__device__ __forceinline__ void wg_stage_load(const WgradArgs& a, const WgTask<T>& t, const PixSrc& src, int it, int kq, int P,
                                              uint4 (&rin)[E]) {
  constexpr int KC = 4 * E;
  int pidx = it * (KC * KCH) + kq * E;
  const int hw = a.Hb * a.Wb;
  int n = pidx / hw;
  const int rem = pidx - n * hw;
  int q = rem / a.Wb;
  int r = rem - q * a.Wb;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int ih = q * t.s + t.dh, iw = r * t.s + t.dw;
    const bool ok = t.ok && (pidx + e < P);
    if (PLAIN) {
      const bool inb = ok && (unsigned)ih < (unsigned)t.Hs && (unsigned)iw < (unsigned)t.Ws;
      const T* p = t.base + ((size_t)(n * t.Hs + ih) * t.Ws + iw) * t.C;
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
      const u32x4 v = *(const __attribute__((address_space(1))) u32x4*)(inb ? (const void*)p : a.zeros);
      rin[e] = make_uint4(v.x, v.y, v.z, v.w);    // unconditional global load: all E loads stay in flight
    } else {
      rin[e] = load_pix_piece<T>(src, n, ih, iw, t.Hs, t.Ws, t.ch, ok);
    }
    const bool wrap = (r + 1 == a.Wb);
    r = wrap ? 0 : r + 1;
    const bool wrap2 = wrap && (q + 1 == a.Hb);
    q = wrap ? (wrap2 ? 0 : q + 1) : q;
    n += wrap2 ? 1 : 0;
  }
}

// Padded-grid K walk (a.fastw): K runs over slots of the grid [N][2^lh][2^lw] (Hb, Wb rounded up to powers of two), SL slots per
// iteration, so a slot's (n, q, r) are bit fields of its index - no divisions - and, because 2^lw divides SL, the column r of each of a
// thread's E pixels never changes: column validity and the column part of the address are computed ONCE (WgFix); an iteration only
// adds its row offset.  Slots outside the real grid, and padding taps, load the zero page.
template <int E> struct WgFix {
  int coff[E];        // (r*s + dw) * C for the thread's E pixels (element offset inside a row), valid columns only
  unsigned okmask;    // bit e: column r < Wb and r*s + dw inside [0, Ws)
  int fq, fn;         // fixed (low) part of the row / image index of the thread's pixel group
};

template <typename T, int E, int SL>
__device__ __forceinline__ void wg_fix_init(const WgradArgs& a, const WgTask<T>& t, int kq, WgFix<E>& f) {
  const int mw = (1 << a.lw) - 1, mh = (1 << a.lh) - 1;
  const int l0 = kq * E;                         // slot inside the iteration; 2^lw >= E: the E pixels share one row
  f.fq = (l0 >> a.lw) & mh;
  f.fn = l0 >> (a.lw + a.lh);
  f.okmask = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int r = (l0 + e) & mw;
    const int iw = r * t.s + t.dw;
    const bool ok = r < a.Wb && (unsigned)iw < (unsigned)t.Ws;
    f.coff[e] = ok ? iw * t.C : 0;
    f.okmask |= ok ? (1u << e) : 0u;
  }
}

template <typename T, int E, int SL>
__device__ __forceinline__ void wg_stage_load_fast(const WgradArgs& a, const WgTask<T>& t, const WgFix<E>& f, int it, uint4 (&rin)[E]) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  // uniform part of the slot index: it * SL supplies the high bits of (q, n); the thread's fixed low bits add without carries
  const int hi = it * SL;
  const int q = ((hi >> a.lw) & ((1 << a.lh) - 1)) + f.fq;
  const int n = (hi >> (a.lw + a.lh)) + f.fn;
  const int ih = q * t.s + t.dh;
  const bool rowok = t.ok && n < a.N && q < a.Hb && (unsigned)ih < (unsigned)t.Hs;
  const int rowoff = (n * t.Hs + ih) * t.Ws * t.C;          // elements; every tensor here is far below 2^31 elements
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const bool ok = rowok && ((f.okmask >> e) & 1u);
    const T* p = t.base + (rowoff + f.coff[e]);
    const u32x4 v = *(const __attribute__((address_space(1))) u32x4*)(ok ? (const void*)p : a.zeros);
    rin[e] = make_uint4(v.x, v.y, v.z, v.w);
  }
}

template <typename T, int WC, int WP, int TC, int TP, int KCH, bool PLAIN>
__global__ __launch_bounds__(WC * WP * 64) void wgrad_kernel(const WgradArgs a) {
  constexpr int NT = WC * WP * 64;                    // 4 waves, or 8 for the 256-row tiles
  constexpr int E = Elem<T>::E, KC = 4 * E;           // KC pixels per 64-byte chunk; KCH chunks per iteration
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16;
  constexpr int CH = 4 * (BC + BP);                   // slots of one chunk (A planes then B planes)
  constexpr int BUF = KCH * CH;
  constexpr int TA = KCH * 4 * BC / E, TB = KCH * 4 * BP / E;   // loader tasks per iteration
  static_assert(TA + TB <= NT, "one task per thread");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m_base = blockIdx.x * BC, d_base = blockIdx.y * BP, split = blockIdx.z;
  const int P = a.N * a.Hb * a.Wb;
  const bool fastw = PLAIN && a.fastw;
  const int Pk = fastw ? (a.N << (a.lw + a.lh)) : P;    // K extent: padded slots or real pixels
  const int niter = (Pk + KC * KCH - 1) / (KC * KCH);
  const int per = (niter + a.splitk - 1) / a.splitk;
  const int it0 = split * per, it1 = min(niter, it0 + per);

  // this thread's loader task (A: gathered operand rows (tap, channel); B: dense operand channels)
  const bool isA = tid < TA;
  const bool active = tid < TA + TB;
  const int tt = isA ? tid : tid - TA;
  const int ncg = (isA ? BC : BP) / E;
  const int cg = tt % ncg, kq = tt / ncg;    // channel group, pixel group (kq / 4 = chunk, kq % 4 = plane)
  WgTask<T> task;
  {
    int ch, tap = 0;
    if (isA) { const int m = m_base + cg * E; tap = m >> a.log2Gc; ch = m & a.gc_mask; task.ok = m < a.ntaps * a.Gc; }   // (1 tap: log2Gc = 30, any channel count)
    else { ch = d_base + cg * E; task.ok = ch < a.Dc; }
    task.ok = task.ok && active;
    int tdh = 0, tdw = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) if (t == tap) { tdh = a.taps.dh[t]; tdw = a.taps.dw[t]; }
    task.s = isA ? a.s : 1; task.dh = isA ? tdh : 0; task.dw = isA ? tdw : 0;
    task.Hs = isA ? a.Hgin : a.Hb; task.Ws = isA ? a.Wgin : a.Wb;
    task.ch = ch;
    const int c0 = isA ? a.g.C[0] : a.d.C[0];
    const int c1 = isA ? a.g.C[1] : a.d.C[1];
    const void* p0 = isA ? a.g.ptr[0] : a.d.ptr[0];
    const void* p1 = isA ? a.g.ptr[1] : a.d.ptr[1];
    const bool second = ch >= c0;
    task.C = second ? c1 : c0;
    task.base = reinterpret_cast<const T*>(second ? p1 : p0) + (second ? ch - c0 : ch);
    if (!task.ok) { task.base = reinterpret_cast<const T*>(a.zeros); task.C = 0; }
  }
  const PixSrc& psrc = isA ? a.g : a.d;
  WgFix<E> fix;
  if (fastw) wg_fix_init<T, E, KC * KCH>(a, task, kq, fix);
  auto stage_load = [&](int it, uint4 (&rin)[E]) {
    if (PLAIN && fastw) wg_stage_load_fast<T, E, KC * KCH>(a, task, fix, it, rin);
    else wg_stage_load<T, E, KCH, PLAIN>(a, task, psrc, it, kq, P, rin);
  };

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 rin[E], rout[E];
  const int wc = wave / WP, wpi = wave - wc * WP;
  const int rowA0 = wc * TC * 16, rowB0 = wpi * TP * 16;
  const int store_off = (kq >> 2) * CH + (isA ? 0 : 4 * BC) + (kq & 3) * (isA ? BC : BP);

  if (it0 < it1) {
    stage_load(it0, rin);
    Transposer<T>::run(rin, rout);
    if (active) {
#pragma unroll
      for (int e = 0; e < E; ++e) lds[store_off + lds_slot<1, E>(cg * E + e)] = rout[e];
    }
    __syncthreads();
    for (int it = it0; it < it1; ++it) {
      const int buf = (it - it0) & 1;
      const bool more = it + 1 < it1;
      if (more) stage_load(it + 1, rin);
#pragma unroll
      for (int c = 0; c < KCH; ++c) {
        const uint4* la = lds + buf * BUF + c * CH;
        mma_chunk<T, TC, TP, BC, BP, 1>(la, la + 4 * BC, rowA0, rowB0, lane, acc);
      }
      if (more) {
        Transposer<T>::run(rin, rout);
        if (active) {
          uint4* dst = lds + (buf ^ 1) * BUF + store_off;
#pragma unroll
          for (int e = 0; e < E; ++e) dst[lds_slot<1, E>(cg * E + e)] = rout[e];
        }
      }
      __syncthreads();
    }
  }

  // lane holds d column (lane&15), rows 4*(lane>>4) .. +3
  if (a.splitk == 1) {   // no K split: write the gradient itself (real taps / channels only), no slab, no reduce pass
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) {
        const int m0 = m_base + rowA0 + tc * 16 + 4 * (lane >> 4);
        const int d = d_base + rowB0 + tp * 16 + (lane & 15);
        if (d >= a.Dreal) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + e, tap = m >> a.log2Gc, gc = m & a.gc_mask;
          if (tap < a.ntaps && gc < a.Greal) {
            float* o = a.dW + ((size_t)tap * a.Greal + gc) * a.Dreal + d;
            *o = acc[tc][tp][e] + (a.accumulate ? *o : 0.f);
          }
        }
      }
    return;
  }
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {
      const int m0 = m_base + rowA0 + tc * 16 + 4 * (lane >> 4);
      const int d = d_base + rowB0 + tp * 16 + (lane & 15);
      float* pp = a.partial + ((size_t)split * a.Mpad + m0) * a.Dpad + d;
#pragma unroll
      for (int e = 0; e < 4; ++e) pp[(size_t)e * a.Dpad] = acc[tc][tp][e];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs a) {
  if ((a.Dreal & 3) == 0) {   // 16-byte path
    const int dq = a.Dreal >> 2;
    const size_t total = (size_t)a.ntaps * a.Greal * dq;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
      const int d = (int)(i % dq) * 4;
      const size_t t = i / dq;
      const int gc = (int)(t % a.Greal);
      const int tap = (int)(t / a.Greal);
      const size_t m = (size_t)tap * a.Gc + gc;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      // slabs added in slab order, eight loads in flight (the slabs of one element lie Mpad * Dpad floats apart: load latency)
      const float* src = a.partial + m * a.Dpad + d;
      const size_t sstride = (size_t)a.Mpad * a.Dpad;
      for (int k0 = 0; k0 < a.splitk; k0 += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k0 + u < a.splitk ? *reinterpret_cast<const float4*>(src + (size_t)(k0 + u) * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (k0 + u >= a.splitk) break;
          s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
        }
      }
      float4* o = reinterpret_cast<float4*>(a.dW + (t * a.Dreal + d));
      if (a.accumulate) { const float4 e = *o; s.x += e.x; s.y += e.y; s.z += e.z; s.w += e.w; }
      *o = s;
    }
    return;
  }
  const size_t total = (size_t)a.ntaps * a.Greal * a.Dreal;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int d = (int)(i % a.Dreal);
    const size_t t = i / a.Dreal;
    const int gc = (int)(t % a.Greal);
    const int tap = (int)(t / a.Greal);
    const size_t m = (size_t)tap * a.Gc + gc;
    float s = 0.f;
    for (int k = 0; k < a.splitk; ++k) s += a.partial[((size_t)k * a.Mpad + m) * a.Dpad + d];
    if (a.accumulate) s += a.dW[i];
    a.dW[i] = s;
  }
}

// many splits, few outputs (the 3/6/4/1-channel layers): one block per gradient row (tap, g).  Thread (k group, d quad) sums
// float4s of slabs k = kg, kg + KG, ... - a wave reads whole 16-byte-aligned row segments of consecutive slabs (coalesced; one
// wave per output element read 4 bytes per slab at a 32 KB stride) - and the KG partial sums fold through LDS in a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_wave_kernel(const WgradArgs a) {
  __shared__ float4 sm[256];
  const int row = blockIdx.x;                       // (tap, gc) with gc < Greal
  const int tap = row / a.Greal, gc = row - tap * a.Greal;
  const size_t m = (size_t)tap * a.Gc + gc;
  const int dq = (a.Dreal + 3) >> 2;                // float4 columns (Dpad is a multiple of 16: reads stay inside the slab row)
  int dqp = 1;
  while (dqp < dq) dqp <<= 1;                       // threads per k group (power of two <= 256)
  const int KG = 256 / dqp;
  const int d4 = threadIdx.x % dqp, kg = threadIdx.x / dqp;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d4 < dq) {
    // this thread's slabs kg, kg + KG, ... in that order, four loads in flight
    const float* src = a.partial + m * a.Dpad + d4 * 4;
    const size_t sstride = (size_t)a.Mpad * a.Dpad;
    for (int k0 = kg; k0 < a.splitk; k0 += 4 * KG) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = k0 + u * KG < a.splitk ? *reinterpret_cast<const float4*>(src + (size_t)(k0 + u * KG) * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (k0 + u * KG >= a.splitk) break;
        s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
      }
    }
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (kg == 0 && d4 < dq) {
    for (int j = 1; j < KG; ++j) { const float4 v = sm[j * dqp + d4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    float* o = a.dW + ((size_t)tap * a.Greal + gc) * a.Dreal + d4 * 4;
    const float r[4] = {s.x, s.y, s.z, s.w};
    for (int e = 0; e < 4; ++e)
      if (d4 * 4 + e < a.Dreal) o[e] = r[e] + (a.accumulate ? o[e] : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------
// optional per-launch timing (HIP events on the launch stream) for bench.py's roofline line
// ------------------------------------------------------------------------------------------------
void igemm_tile(int cfg, int* bc, int* bp);
void wgrad_tile(int cfg, int* bm, int* bn);

struct ProfRec { hipEvent_t e0, e1; std::string name; double flops, bytes; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_ev_pool;

static hipEvent_t prof_event() {
  hipEvent_t e;
  if (!g_ev_pool.empty()) { e = g_ev_pool.back(); g_ev_pool.pop_back(); return e; }
  hipEventCreate(&e);
  return e;
}

static std::string g_prof_tag;
static bool g_prof_detail = false;
void profile_enable(int on) { g_prof_on = on != 0; g_prof_detail = on > 1; }
void profile_tag(const char* tag) { if (g_prof_on && g_prof_detail) g_prof_tag = tag ? tag : ""; }

// which kernel family a launch_igemm_cfg call ended up on ("ws" wave-specialised, "dma", "reg" register loader): part of the
// class name of the record, so that a class is ONE kernel template (the names rocprofv3 reports are per kernel too)
static const char* g_prof_family = nullptr;

struct ProfScope {
  bool on;
  ProfRec r;
  hipStream_t st;
  std::string kind;
  int bf, bc, bp;
  ProfScope(const char* kind_, int is_bf16, int bc_, int bp_, double flops, double bytes, hipStream_t s) : on(g_prof_on), st(s), kind(kind_), bf(is_bf16), bc(bc_), bp(bp_) {
    if (!on) return;
    g_prof_family = nullptr;
    r.flops = flops; r.bytes = bytes;
    r.e0 = prof_event(); r.e1 = prof_event();
    hipEventRecord(r.e0, st);
  }
  ~ProfScope() {
    if (!on) return;
    hipEventRecord(r.e1, st);
    char buf[96];
    // classes follow the kernel template instances rocprofv3 lists (family / variant, operand type, tile)
    if ((kind == "igemm" || kind == "wgrad") && g_prof_family) snprintf(buf, sizeof(buf), "%s_%s_%s_%dx%d", kind.c_str(), g_prof_family, bf ? "bf16" : "f32", bc, bp);
    else snprintf(buf, sizeof(buf), "%s_%s_%dx%d", kind.c_str(), bf ? "bf16" : "f32", bc, bp);
    r.name = buf;
    if (g_prof_detail && !g_prof_tag.empty()) r.name = g_prof_tag + " " + r.name;
    g_prof.push_back(r);
  }
};

// JSON array of {name, calls, ms, flops, bytes}; call after the stream has been synchronised
size_t profile_collect(char* out, size_t cap) {
  struct Agg { std::string name; int calls; double ms, flops, bytes; };
  static std::string cache = "[]";
  if (g_prof.empty()) {   // second call of the (size query, copy) pair
    if (out && cap > 0) { strncpy(out, cache.c_str(), cap - 1); out[cap - 1] = 0; }
    return cache.size() + 1;
  }
  std::vector<Agg> agg;
  for (ProfRec& r : g_prof) {
    float ms = 0.f;
    hipEventSynchronize(r.e1);
    hipEventElapsedTime(&ms, r.e0, r.e1);
    g_ev_pool.push_back(r.e0); g_ev_pool.push_back(r.e1);
    Agg* a = nullptr;
    for (Agg& x : agg) if (x.name == r.name) a = &x;
    if (!a) { agg.push_back({r.name, 0, 0, 0, 0}); a = &agg.back(); }
    a->calls++; a->ms += ms; a->flops += r.flops; a->bytes += r.bytes;
  }
  g_prof.clear();
  std::string js = "[";
  for (size_t i = 0; i < agg.size(); ++i) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s{\"name\":\"%s\",\"calls\":%d,\"ms\":%.6f,\"flops\":%.6e,\"bytes\":%.6e}", i ? "," : "",
             agg[i].name.c_str(), agg[i].calls, agg[i].ms, agg[i].flops, agg[i].bytes);
    js += buf;
  }
  js += "]";
  cache = js;
  if (out && cap > 0) { strncpy(out, js.c_str(), cap - 1); out[cap - 1] = 0; }
  return js.size() + 1;
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------
template <typename T, int WC, int WP, int TC, int TP>
static hipError_t launch_igemm_cfg(const IgemmArgs& a, hipStream_t st) {
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16, NW = WC * WP;
  const int P = a.N * a.Hg * a.Wg;
  dim3 grid((P + BP - 1) / BP, a.CoutPad / BC, a.nclass * a.splitk);
  constexpr bool RING = ((BC / 16) % NW == 0) && ((BP / 16) % NW == 0);
  constexpr int RINGB = (RING ? VP_RING : 2) * 4 * (BC + BP) * 16;
  constexpr int NPASS = epi_passes(BC, BP, WP, RINGB);
  size_t smem = RINGB + 64 + BP * 8;      // ring, tap table, output-offset table of the direct epilogue
  const size_t smem_epi = (size_t)(BP / NPASS) * (BC * 4 + 16) + (BP / NPASS) * 8;
  if (smem_epi > smem) smem = smem_epi;
  const bool plain = a.zeros && !a.x.aff_a[0] && !a.x.aff_a[1] && a.x.act == ACT_NONE;
  if (plain) {
    IgemmArgs b = a;
    b.vec_epi = (a.splitk == 1 && a.Cout % 8 == 0 && a.ldY % 8 == 0) ? 1 : 0;
    const bool bst = a.bst_y || a.bst_y2;              // backward sums of a batch-normalised tensor in the epilogue (staged_epilogue, STATS == 2)
    // register-direct epilogue: measured SLOWER than the LDS-staged one (64-byte store segments vs 256-byte rows): opt-in
    // scalar-stepped loader: every 64-byte K chunk inside one tap and one source tensor, sources below the 2 GiB lane-offset range
    constexpr int KCE = 16 * 4 / (int)sizeof(T);
    const size_t xb0 = (size_t)a.N * a.Hin * a.Win * a.x.C[0] * sizeof(T), xb1 = (size_t)a.N * a.Hin * a.Win * a.x.C[1] * sizeof(T);
    b.fastk = (a.Cin % KCE == 0 && a.x.C[0] % KCE == 0 && a.x.C[1] % KCE == 0 && a.x.C[0] + a.x.C[1] == a.Cin &&
               xb0 < 0x70000000ull && xb1 < 0x70000000ull) ? 1 : 0;
    // wave-specialised kernel, per tile shape (bit = launch_igemm cfg index): measured gains for 128x128 (cfg 0), 64x128 (cfg 1), 256x256 (cfg 7); 128x256 is faster without
    constexpr int ws_cfgs = (1 << 0) | (1 << 1) | (1 << 7);
    constexpr int my_cfg = (BC == 128 && BP == 128) ? 0 : (BC == 64 && BP == 128) ? 1 : (BC == 128 && BP == 256) ? 6 : (BC == 256 && BP == 256) ? 7 :
                           (BC == 64 && BP == 256) ? 8 : (BC == 128 && BP == 512) ? 9 : 31;
    if constexpr (((BC + BP) / 16) % 4 == 0 && NW <= 8) {
      if (((ws_cfgs >> my_cfg) & 1) && b.vec_epi && b.fastk && a.splitk == 1) {
        constexpr int NSTW = (NW == 8) ? 4 : 4;
        constexpr int RB = NSTW * 4 * (BC + BP) * 16;
        constexpr int NPE = epi_passes(BC, BP, WP, RB);
        size_t sm = RB + 64 + BP * 8;
        const size_t se = (size_t)(BP / NPE) * (BC * 4 + 16) + (BP / NPE) * 8;
        if (se > sm) sm = se;
        g_prof_family = "ws";
        if (bst && b.split_c) hipLaunchKernelGGL((igemm_ws_kernel<T, WC, WP, TC, TP, NSTW, 2, true>), grid, dim3((NW + 4) * 64), sm, st, b);
        else if (bst) hipLaunchKernelGGL((igemm_ws_kernel<T, WC, WP, TC, TP, NSTW, 2>), grid, dim3((NW + 4) * 64), sm, st, b);
        else if (b.bn_part) hipLaunchKernelGGL((igemm_ws_kernel<T, WC, WP, TC, TP, NSTW, 1>), grid, dim3((NW + 4) * 64), sm, st, b);
        else if (b.split_c) hipLaunchKernelGGL((igemm_ws_kernel<T, WC, WP, TC, TP, NSTW, 0, true>), grid, dim3((NW + 4) * 64), sm, st, b);
        else hipLaunchKernelGGL((igemm_ws_kernel<T, WC, WP, TC, TP, NSTW, 0>), grid, dim3((NW + 4) * 64), sm, st, b);
        return hipGetLastError();
      }
    }
    g_prof_family = "dma";
    if (bst && !b.vec_epi) return hipErrorInvalidValue;          // (the backward sums exist in the staged epilogue only: the host asks for them under its conditions)
    if (bst && b.split_c) hipLaunchKernelGGL((igemm_dma_kernel<T, WC, WP, TC, TP, true, 2, true>), grid, dim3(NW * 64), smem, st, b);
    else if (bst) hipLaunchKernelGGL((igemm_dma_kernel<T, WC, WP, TC, TP, true, 2>), grid, dim3(NW * 64), smem, st, b);
    else if (b.vec_epi && b.bn_part) hipLaunchKernelGGL((igemm_dma_kernel<T, WC, WP, TC, TP, true, 1>), grid, dim3(NW * 64), smem, st, b);
    else if (b.vec_epi && b.split_c) hipLaunchKernelGGL((igemm_dma_kernel<T, WC, WP, TC, TP, true, 0, true>), grid, dim3(NW * 64), smem, st, b);
    else if (b.vec_epi) hipLaunchKernelGGL((igemm_dma_kernel<T, WC, WP, TC, TP, true>), grid, dim3(NW * 64), smem, st, b);
    else if (b.split_c) hipLaunchKernelGGL((igemm_dma_kernel<T, WC, WP, TC, TP, false, 0, true>), grid, dim3(NW * 64), smem, st, b);
    else hipLaunchKernelGGL((igemm_dma_kernel<T, WC, WP, TC, TP, false>), grid, dim3(NW * 64), smem, st, b);
    return hipGetLastError();
  }
  g_prof_family = "reg";
  if (a.bst_y || a.bst_y2) return hipErrorInvalidValue;     // (the staged epilogue of the LDS-DMA kernels only)
  if (a.split_c) return hipErrorInvalidValue;        // the two-output form exists on the LDS-DMA kernels only
  if constexpr (NW == 4) hipLaunchKernelGGL((igemm_kernel<T, WC, WP, TC, TP>), grid, dim3(256), 2 * 4 * (BC + BP) * 16 + 64, st, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}


// 8-channel (padded image) inputs, 64 outputs: what conv_cin8_kernel handles (`zeros` is not looked at: the step executor asks before it
// has filled the run-time pointers in)
bool conv_cin8_eligible(const IgemmArgs& a, int is_bf16) {
  const bool pow2 = (a.Wg & (a.Wg - 1)) == 0 && (a.Hg & (a.Hg - 1)) == 0;
  return is_bf16 && a.Cin == 8 && a.x.C[0] == 8 && a.x.C[1] == 0 && a.Cout == 64 && a.ldY == 64 && a.nclass == 1 && a.splitk == 1 && a.os == 1 &&
         a.Hof == a.Hg && a.Wof == a.Wg && pow2 && !a.ref && !a.accumulate && !a.y_f32 && !a.bn_part && !a.x.aff_a[0] && a.x.act == ACT_NONE &&
         a.ntaps <= 16 && (a.Kpad == 96 || a.Kpad == 128) && (size_t)a.N * a.Hin * a.Win * 16 < 0x70000000ull &&
         (((long long)a.N * a.Hg * a.Wg) & 15) == 0 && (a.out_act == ACT_NONE || a.out_act == ACT_RELU) &&
         (a.Kpad == 128 || !(a.xa_lrelu || a.xa_relu)) && !(a.xa_relu && !a.xa_lrelu);      // (the instantiated output combinations: conv_cin8_kernel)
}

template <typename T> static hipError_t launch_igemm_t(const IgemmArgs& a, int cfg, hipStream_t st) {
  hipError_t e;
  int pbc, pbp;
  igemm_tile(cfg, &pbc, &pbp);
  // algorithmic cost: 2*MACs over real channels; bytes = X + W + Y each touched once
  const double Pn = (double)a.N * a.Hg * a.Wg * a.nclass;
  const double kreal = (double)a.ntaps * a.cin_real;
  const double es = sizeof(T);
  if constexpr (sizeof(T) == 2) {
    // 8-channel (padded image) inputs, 64 outputs: the direct register-resident form (conv_cin8_kernel)
    if (a.zeros && conv_cin8_eligible(a, 1)) {
      ProfScope prof("cin8", true, 64, 16, 2.0 * Pn * a.Cout * kreal,
                     es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.Cout + Pn * a.Cout), st);
      const int ntile = (int)((Pn + 15) / 16);
      int blocks = (ntile + 3) / 4;
      if (blocks > thin_blocks_knob(3)) blocks = thin_blocks_knob(3);
      int lgW = 0, lgH = 0;
      while ((1 << lgW) < a.Wg) ++lgW;
      while ((1 << lgH) < a.Hg) ++lgH;
      // which outputs exist is a template parameter (bit 0 raw output, 1 lrelu copy, 2 relu copy): the tile loop has no branch
      const int outs = (a.Y ? 1 : 0) | (a.xa_lrelu ? 2 : 0) | (a.xa_relu ? 4 : 0);
      void (*kern)(const IgemmArgs, int, int) = nullptr;
      if (a.Kpad == 96) kern = outs == 1 ? conv_cin8_kernel<3, 1> : nullptr;
      else switch (outs) {
        case 1: kern = conv_cin8_kernel<4, 1>; break;
        case 2: kern = conv_cin8_kernel<4, 2>; break;
        case 3: kern = conv_cin8_kernel<4, 3>; break;
        case 6: kern = conv_cin8_kernel<4, 6>; break;
        case 7: kern = conv_cin8_kernel<4, 7>; break;
        default: break;
      }
      if (!kern) return hipErrorInvalidValue;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, st, a, lgW, lgH);
      return hipGetLastError();
    }
  }
  if constexpr (sizeof(T) == 2) {
    // 3x3 stride-1 conv from 64 to <= 8 channels (conv1_1 backward-data): halo tile staged once (conv3x3_cout8_tile_kernel)
    bool near = a.ntaps == 9;
    for (int t = 0; near && t < 9; ++t) near = a.taps[0].dh[t] >= -1 && a.taps[0].dh[t] <= 1 && a.taps[0].dw[t] >= -1 && a.taps[0].dw[t] <= 1;
    if (near && a.zeros && a.nclass == 1 && a.sh == 1 && a.sw == 1 && a.os == 1 && a.Cin == 64 && a.x.C[0] == 64 && a.x.C[1] == 0 &&
        a.CoutPad == 16 && a.Cout <= 8 && a.ldY == 8 && a.splitk == 1 && !a.rowperm && !a.bn_part && !a.x.aff_a[0] && a.x.act == ACT_NONE &&
        (a.Wg & (a.Wg - 1)) == 0 && (a.Hg & (a.Hg - 1)) == 0 && a.Wg >= 16 && a.Hg >= 4 && a.Hof == a.Hg && a.Wof == a.Wg && a.Hin == a.Hg &&
        a.Win == a.Wg && !a.y_f32 && (size_t)a.N * a.Hin * a.Win * 64 * 2 < 0x70000000ull &&
        !a.bias && a.out_act == ACT_NONE && !a.ref && !a.accumulate && !a.split_c) {          // (the kernel's plain epilogue)
      ProfScope prof("cout8", true, 16, 64, 2.0 * Pn * a.Cout * kreal,
                     es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.Cout + Pn * a.Cout), st);
      int lgW = 0, lgH = 0;
      while ((1 << lgW) < a.Wg) ++lgW;
      while ((1 << lgH) < a.Hg) ++lgH;
      int tblocks = a.N << (lgW - 4 + lgH - 2);
      if (tblocks > thin_blocks_knob(0)) tblocks = thin_blocks_knob(0);
      hipLaunchKernelGGL(conv3x3_cout8_tile_kernel, dim3(tblocks), dim3(256), (size_t)6 * 18 * 144 + 256 * 16, st, a, lgW, lgH);
      return hipGetLastError();
    }
  }
  if constexpr (sizeof(T) == 2) {
    // 4x4 stride-2 transposed conv from 64 to <= 8 channels (layer_1 backward-data): deconv_cout8_tile_kernel
    if (a.zeros && a.nclass == 4 && a.os == 2 && a.ntaps == 4 && a.Cin == 64 && a.x.C[0] == 64 && a.x.C[1] == 0 && a.CoutPad == 16 && a.Cout <= 8 &&
        a.ldY == 8 && !a.y_f32 && a.splitk == 1 && !a.rowperm && !a.bn_part && !a.x.aff_a[0] && a.x.act == ACT_NONE && (a.Wg & (a.Wg - 1)) == 0 &&
        (a.Hg & (a.Hg - 1)) == 0 && a.Wg >= 16 && a.Hg >= 4 && a.Hof == 2 * a.Hg && a.Wof == 2 * a.Wg && a.Hin == a.Hg && a.Win == a.Wg &&
        (size_t)a.N * a.Hin * a.Win * 64 * 2 < 0x70000000ull && (size_t)a.N * a.Hof * a.Wof * 8 * 2 < 0x70000000ull &&
        !a.bias && a.out_act == ACT_NONE && !a.ref && !a.accumulate && !a.split_c) {          // (the kernel's plain epilogue)
      ProfScope prof("dcout8", true, 32, 64, 2.0 * Pn * a.Cout * kreal,
                     es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.nclass * a.Cout + Pn * a.Cout), st);
      int lgW = 0, lgH = 0;
      while ((1 << lgW) < a.Wg) ++lgW;
      while ((1 << lgH) < a.Hg) ++lgH;
      int tblocks = a.N << (lgW - 4 + lgH - 2);
      if (tblocks > thin_blocks_knob(1)) tblocks = thin_blocks_knob(1);
      const size_t smt = (size_t)2 * 18 * 64 * 16 + (size_t)6 * 18 * 144 + 256 * 16;
      (void)hipFuncSetAttribute((const void*)deconv_cout8_tile_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smt);
      hipLaunchKernelGGL(deconv_cout8_tile_kernel<2>, dim3(tblocks), dim3(256), smt, st, a, lgW, lgH);
      return hipGetLastError();
    }
  }
  if constexpr (sizeof(T) == 2) {
    // 4-channel f32 transposed conv (decoder_1): the four parity classes x four channels as one MFMA tile (deconv_cout4_kernel)
    const bool pow2 = (a.Wg & (a.Wg - 1)) == 0 && (a.Hg & (a.Hg - 1)) == 0;
    const int spt = a.Cin / 32;
    if (a.zeros && a.nclass == 4 && a.os == 2 && a.ntaps == 4 && a.Cout == 4 && a.y_f32 && a.ldY == 4 && a.splitk == 1 && pow2 &&
        a.Cin % 32 == 0 && (spt == 2 || spt == 4) && a.x.C[0] % 32 == 0 && a.x.C[0] + a.x.C[1] == a.Cin && !a.ref && !a.accumulate &&
        a.out_act == ACT_NONE && !a.x.aff_a[0] && !a.x.aff_a[1] && a.x.act == ACT_NONE && a.Hof == 2 * a.Hg && a.Wof == 2 * a.Wg &&
        (size_t)a.N * a.Hin * a.Win * a.Cin * 2 < 0x70000000ull && (size_t)a.N * a.Hof * a.Wof * 16 < 0x70000000ull) {
      ProfScope prof("cout4", true, 16, 16, 2.0 * Pn * a.Cout * kreal,
                     es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.nclass * a.Cout) + 4.0 * Pn * a.Cout, st);
      const int ntile = (a.N * a.Hg * a.Wg + 15) / 16;
      int blocks = (ntile + 3) / 4;
      if (blocks > 2048) blocks = 2048;
      int lgW = 0, lgH = 0;
      while ((1 << lgW) < a.Wg) ++lgW;
      while ((1 << lgH) < a.Hg) ++lgH;
      const size_t sm = (size_t)9 * spt * 64 * 16;
      const bool halved = a.x.C[1] == a.x.C[0];
      if (lgW >= 4 && lgH >= 2 && (halved || a.x.C[1] == 0)) {    // 4 x 16 base-pixel tiles with the halo staged once in LDS
        const size_t smt = sm + (size_t)6 * 18 * (a.Cin * 2 + 16) + 256 * 16;
        int tblocks = a.N << (lgW - 4 + lgH - 2);
        if (tblocks > thin_blocks_knob(2)) tblocks = thin_blocks_knob(2);
        auto kern = spt == 2 ? (halved ? deconv_cout4_tile_kernel<2, 1> : deconv_cout4_tile_kernel<2, 0>)
                             : (halved ? deconv_cout4_tile_kernel<4, 1> : deconv_cout4_tile_kernel<4, 0>);
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smt);
        hipLaunchKernelGGL(kern, dim3(tblocks), dim3(256), smt, st, a, lgW, lgH);
        return hipGetLastError();
      }
      if (spt == 2) hipLaunchKernelGGL((deconv_cout4_kernel<2>), dim3(blocks), dim3(256), sm, st, a, lgW, lgH);
      else hipLaunchKernelGGL((deconv_cout4_kernel<4>), dim3(blocks), dim3(256), sm, st, a, lgW, lgH);
      return hipGetLastError();
    }
  }
  if (a.patch == 3) {   // few-pixel layers: conv_smallp.hip (plain epilogue; the fused batch-norm forms are launched by the step executor)
    ProfScope prof("smallp", sizeof(T) == 2, pbc, pbp, 2.0 * Pn * a.Cout * kreal,
                   es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.nclass * a.Cout + Pn * a.Cout), st);
    return launch_igemm_smallp(a, sizeof(T) == 2, st);
  }
  if constexpr (sizeof(T) == 2) {
    // 3x3 stride-1 conv from 64 to 64 / 128 channels (VGG conv1_2 forward / backward-data, conv2_1 forward): weights resident in
    // registers, 4 x 16-pixel tiles (conv_c64.hip)
    if (c64_knob() && conv_c64_eligible(a, 1)) {
      ProfScope prof("c64", true, a.Cout, 64, 2.0 * Pn * a.Cout * kreal,
                     es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.nclass * a.Cout + Pn * a.Cout), st);
      return launch_conv_c64(a, st);
    }
  }
  if constexpr (sizeof(T) == 2) {
    // 4x4 stride-2 transposed conv from 128 to 64 channels (backward-data of layer_2 / encoder_2 / encoder_fg_2): weights resident in
    // registers, two parity classes per block (conv_dc64.hip)
    if (conv_dc256_eligible(a, 1)) {
      ProfScope prof("dc256", true, 64, 128, 2.0 * Pn * a.Cout * kreal,
                     es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.nclass * a.Cout + Pn * a.Cout), st);
      return launch_conv_dc256(a, st);
    }
    if (dc64_knob() && conv_dc64_eligible(a, 1)) {
      ProfScope prof("dc64", true, 64, 128, 2.0 * Pn * a.Cout * kreal,
                     es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.nclass * a.Cout + Pn * a.Cout), st);
      return launch_conv_dc64(a, st);
    }
  }
  if constexpr (sizeof(T) == 2) {
    // 4x4 stride-2 conv from 64 to 128 channels in front of a batch-norm (layer_2 / encoder_2 / encoder_fg_2 forward): weights resident
    // in registers, parity-split input patch, batch statistics per block (conv_s2c64.hip)
    if (a.patch == 4) {
      if (!conv_s2c64_eligible(a, 1)) return hipErrorInvalidValue;       // (a plan for this kernel runs on no other: fail loudly)
      ProfScope prof("s2c64", true, 128, 64, 2.0 * Pn * a.Cout * kreal,
                     es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.nclass * a.Cout + Pn * a.Cout), st);
      return launch_conv_s2c64(a, st);
    }
  }
  if (a.patch == 4) return hipErrorInvalidValue;
  if constexpr (sizeof(T) == 2) {
    // 4x4 / stride-1 taps without batch statistics (the discriminator's layer_4 backward-data passes): the unrolled patch kernel with
    // 16 tap steps per chunk, 128-row x 16 x 16-pixel tiles (conv_patch3.hip, KW = 4)
    if (a.patch == 1 && patch4_eligible(a, 1)) {
      ProfScope prof("patch4", true, 128, 256, 2.0 * Pn * a.Cout * kreal,
                     es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.nclass * a.Cout + Pn * a.Cout), st);
      return launch_igemm_patch4(a, st);
    }
  }
  if (a.patch) {   // stride-1 convs with the input patch staged once per channel chunk (conv_patch.hip)
    // class name per kernel template: patch2 (parity classes, conv_patch2.hip), patch3 (unrolled 3x3, conv_patch3.hip), patch (generic)
    const char* pk = a.patch == 2 ? "patch2" : (patch3_knob() && patch3_eligible(a, sizeof(T) == 2) ? "patch3" : "patch");
    ProfScope prof(pk, sizeof(T) == 2, pbc, pbp, 2.0 * Pn * a.Cout * kreal,
                   es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.nclass * a.Cout + Pn * a.Cout), st);
    return launch_igemm_patch(a, sizeof(T) == 2, pbc, pbp, st);
  }
  const bool use_patch = false;
  ProfScope prof(use_patch ? "patch" : "igemm", sizeof(T) == 2, pbc, pbp, 2.0 * Pn * a.Cout * kreal,
                 es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.nclass * a.Cout + Pn * a.Cout), st);
  switch (cfg) {
    case 0: e = launch_igemm_cfg<T, 2, 2, 4, 4>(a, st); break;   // 128 ch x 128 px (the same tile on 8 + 4 waves for small grids: +-0, EXPERIMENTS.md 0.2)
    case 1: e = launch_igemm_cfg<T, 1, 4, 4, 2>(a, st); break;   //  64 ch x 128 px
    case 2: e = launch_igemm_cfg<T, 1, 4, 1, 2>(a, st); break;   //  16 ch x 128 px
    case 3: e = launch_igemm_cfg<T, 4, 1, 2, 2>(a, st); break;   // 128 ch x  32 px
    case 4: e = launch_igemm_cfg<T, 4, 1, 2, 1>(a, st); break;   // 128 ch x  16 px
    case 5: e = launch_igemm_cfg<T, 2, 2, 2, 1>(a, st); break;   //  64 ch x  32 px
    case 6: e = launch_igemm_cfg<T, 2, 4, 4, 4>(a, st); break;   // 128 ch x 256 px, 8 waves
    case 7: e = launch_igemm_cfg<T, 2, 4, 8, 4>(a, st); break;   // 256 ch x 256 px, 8 waves
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  if (a.splitk > 1) {
    const size_t total = (size_t)a.nclass * a.N * a.Hg * a.Wg * (a.CoutPad / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (a.split_c) hipLaunchKernelGGL((igemm_splitk_reduce_kernel<T, true>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((igemm_splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, a);
    e = hipGetLastError();
  }
  return e;
}

// the few-pixel kernel with a fused batch-norm epilogue (SmallPArgs::mode), launched by the step executor; timed like every other conv launch
hipError_t launch_smallp_fused(const SmallPArgs& s, int is_bf16, hipStream_t st) {
  const IgemmArgs& a = s.g;
  const double Pn = (double)a.N * a.Hg * a.Wg * a.nclass, kreal = (double)a.ntaps * a.cin_real, es = is_bf16 ? 2 : 4;
  ProfScope prof("smallp", is_bf16, 32, a.sp_npt * 16, 2.0 * Pn * a.Cout * kreal,
                 es * ((double)a.N * a.Hin * a.Win * a.cin_real + kreal * a.nclass * a.Cout + Pn * a.Cout), st);
  return launch_smallp(s, is_bf16, st);
}

// the one-output-channel backward-data kernel (conv_cout1.hip), launched by the step executor; timed like every other conv launch
hipError_t launch_cout1_bwd_prof(const Cout1Args& a, hipStream_t st) {
  const double px = (double)a.N * a.H * a.W;
  ProfScope prof("cout1bwd", true, 512, 16, 2.0 * px * a.C * a.ks * a.ks,
                 2.0 * ((double)a.N * a.Ho * a.Wo + (double)a.ks * a.ks * a.C + px * a.C), st);
  return launch_conv_cout1_bwd(a, st);
}

hipError_t launch_cout1_wgrad_prof(const Cout1Args& a, hipStream_t st) {
  const double px = (double)a.N * a.H * a.W;
  ProfScope prof("cout1wgrad", true, 16, 512, 2.0 * px * a.C * a.ks * a.ks,
                 2.0 * ((double)a.N * a.Ho * a.Wo + px * a.C) + 4.0 * a.ks * a.ks * a.C, st);
  return launch_conv_cout1_wgrad(a, st);
}

hipError_t launch_igemm(const IgemmArgs& a, int is_bf16, int cfg, hipStream_t st) {
  return is_bf16 ? launch_igemm_t<bf16>(a, cfg, st) : launch_igemm_t<float>(a, cfg, st);
}

void igemm_tile(int cfg, int* bc, int* bp) {
  static const int t[19][2] = {{128, 128}, {64, 128}, {16, 128}, {128, 32}, {128, 16}, {64, 32}, {128, 256}, {256, 256}, {64, 256}, {128, 512},
                               {256, 256}, {128, 512}, {64, 512}, {128, 256}, {64, 256}, {256, 128},     // 10..15: patch kernel tiles (conv_ops.h patch_tile_hw)
                               {32, 16}, {32, 32}, {32, 64}};                                               // 16..18: few-pixel kernel (conv_smallp.hip)
  *bc = t[cfg][0]; *bp = t[cfg][1];
}

template <typename T, int WC, int WP, int TC, int TP>
static hipError_t launch_wgrad_cfg(const WgradArgs& a, hipStream_t st) {
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16;
  dim3 grid(a.Mpad / BC, a.Dpad / BP, a.splitk);
  // bf16 tasks are 8x8 blocks: two chunks per iteration keep all 256 threads loading
  constexpr int KCH = (sizeof(T) == 2) ? 2 : 1;
  const size_t smem = 2 * KCH * 4 * (BC + BP) * 16;
  const bool plain = a.zeros && !a.g.aff_a[0] && !a.g.aff_a[1] && a.g.act == ACT_NONE && !a.d.aff_a[0] && !a.d.aff_a[1] && a.d.act == ACT_NONE;
  if (smem > 64 * 1024) {
    static bool done[2] = {false, false};
    if (!done[plain ? 1 : 0]) {
      if (plain) (void)hipFuncSetAttribute((const void*)wgrad_kernel<T, WC, WP, TC, TP, KCH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      else (void)hipFuncSetAttribute((const void*)wgrad_kernel<T, WC, WP, TC, TP, KCH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      done[plain ? 1 : 0] = true;
    }
  }
  if (plain) hipLaunchKernelGGL((wgrad_kernel<T, WC, WP, TC, TP, KCH, true>), grid, dim3(WC * WP * 64), smem, st, a);
  else hipLaunchKernelGGL((wgrad_kernel<T, WC, WP, TC, TP, KCH, false>), grid, dim3(WC * WP * 64), smem, st, a);
  return hipGetLastError();
}

template <typename T> static hipError_t launch_wgrad_t(const WgradArgs& a, int cfg, hipStream_t st) {
  hipError_t e;
  int pbm, pbn;
  wgrad_tile(cfg, &pbm, &pbn);
  const double Pn = (double)a.N * a.Hb * a.Wb;
  ProfScope prof("wgrad", sizeof(T) == 2, pbm, pbn, 2.0 * Pn * a.ntaps * a.Greal * a.Dreal,
                 sizeof(T) * ((double)a.N * a.Hgin * a.Wgin * a.Greal + Pn * a.Dreal) + 4.0 * a.ntaps * a.Greal * a.Dreal, st);
  if (sizeof(T) == 4 && wgrad_mm_eligible(a, cfg)) {               // one-tap float32 products: transpose-free LDS-DMA kernel (wgrad_mm.hip)
    g_prof_family = "mm";
    e = launch_wgrad_mm(a, cfg, st);
  } else
  switch (cfg) {
    case 0: e = launch_wgrad_cfg<T, 2, 2, 4, 4>(a, st); break;   // 128 rows x 128 cols
    case 1: e = launch_wgrad_cfg<T, 2, 2, 4, 2>(a, st); break;   // 128 rows x  64 cols
    case 2: e = launch_wgrad_cfg<T, 4, 1, 2, 1>(a, st); break;   // 128 rows x  16 cols
    case 5: case 6: {                                               // 256 / 128 rows x 128 cols, LDS-DMA + transpose reads (wgrad_tr.hip; bf16, plain operands)
      const bool plain = a.zeros && !a.g.aff_a[0] && !a.g.aff_a[1] && a.g.act == ACT_NONE && !a.d.aff_a[0] && !a.d.aff_a[1] && a.d.act == ACT_NONE;
      if (sizeof(T) != 2 || !plain) return hipErrorInvalidValue;
      int cols = pbn;
      e = launch_wgrad_tr(a, st, &g_prof_family, &cols);
      prof.bp = cols;                                                // (the 256-column tile where the launcher took it: the class names the template instance)
      break;
    }
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess || a.splitk == 1) return e;
  if (a.splitk >= 32 && a.Dreal <= 1024) {
    hipLaunchKernelGGL(wgrad_reduce_wave_kernel, dim3(a.ntaps * a.Greal), dim3(256), 0, st, a);
    return hipGetLastError();
  }
  const size_t total = (size_t)a.ntaps * a.Greal * a.Dreal / ((a.Dreal & 3) ? 1 : 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_wgrad(const WgradArgs& a, int is_bf16, int cfg, hipStream_t st) {
  return is_bf16 ? launch_wgrad_t<bf16>(a, cfg, st) : launch_wgrad_t<float>(a, cfg, st);
}

void wgrad_tile(int cfg, int* bm, int* bn) {
  static const int t[7][2] = {{128, 128}, {128, 64}, {128, 16}, {256, 256}, {256, 128}, {256, 128}, {128, 128}};   // 5, 6: wgrad_tr.hip
  *bm = t[cfg][0]; *bn = t[cfg][1];
}

}  // namespace vp
