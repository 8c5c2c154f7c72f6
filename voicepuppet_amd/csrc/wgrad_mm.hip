// wgrad_mm_kernel: the float32 weight gradient of a ONE-TAP layer (1x1 convolution / dense layer: BFMNet's matrix products,
// vp_mm_bwd_weight_f32) as an LDS-DMA GEMM with NO transpose anywhere:
//     dW[k][n] = sum over pixels p of x[p][k] * dy[p][n]        x [P, ldx], dy [P, lddy] row-major
// With v_mfma_f32_16x16x4_f32 a lane's operand is ONE float - A: (row = lane % 16, k = lane / 16), B: (k = lane / 16, column = lane % 16)
// - and the contraction index k is the PIXEL.  So the [pixel][channel] rows go global -> LDS by LDS-DMA exactly as they lie in memory
// (16-byte pieces, whole 512 / 256-byte row segments: coalesced), and a fragment is one ds_read_b32 per lane: 4 pixel rows x 16
// consecutive channels.  The general weight-gradient kernel (conv_kernels.hip wgrad_kernel) loads 16-byte pieces into registers,
// transposes 4 x 4 blocks and writes them to LDS; here no vector ALU touches the operands.
// Bank conflicts: a pixel row of the tile is 128 (64) floats = a multiple of the 64 banks, so the 4 rows of a k-step would collide.
// The DMA lanes fetch the pieces of pixel row r rotated by 4 * (r % 4) pieces (which global piece a lane fetches is free): channel c
// of pixel row r lives at float (c + 16 * (r % 4)) % ROW of its LDS row, and the 4 rows of a k-step hit 4 disjoint groups of 16 banks.
// Pixels beyond P and channels beyond the operand's width read zeros through the buffer descriptor (out-of-range offsets).
// Ring of NST stages of 16 pixels, counted vmcnt, one barrier per stage (the loop of igemm_dma_kernel).  Tile 128 x 128 or 128 x 64,
// 4 waves of 64 x 64 / 64 x 32; K split over blocks with the slab layout and the reduce kernels of wgrad_kernel.
#include <stdlib.h>

#include "conv_ops.h"
#include "igemm_device.h"
#include "launch.h"
#include "vp_common.h"

namespace vp {

namespace {

template <int IMM> __device__ __forceinline__ float lds_rd32(int addr) {
  float r;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM));
  return r;
}

// TAPS: the k x k form.  Row m of the gradient is (tap, channel); a 16-byte piece (4 channels) lies in one tap, so a DMA lane has ONE
// tap offset (dh, dw) for the whole launch.  K walks the padded grid [N][2^lh][2^lw] (shifts, no divisions); per stage a lane forms the
// gathered pixel (q * s + dh, r * s + dw) of its row, out-of-image / out-of-grid rows read zeros.  A tile's channels lie in ONE source
// tensor of a virtual concat (checked on the host), chosen per block.
template <int TP, bool TAPS>
__global__ __launch_bounds__(256) void wgrad_mm_kernel(const WgradArgs a) {
  constexpr int BM = 128, BN = 2 * TP * 16;          // rows (channels of x) x columns (channels of dy) of the tile
  constexpr int TC = 4;                              // 2 x 2 waves of (TC x 16) x (TP x 16)
  constexpr int KP = 16, NST = 4;                    // pixels per stage, ring stages
  constexpr int SA = KP * BM * 4, SB = KP * BN * 4;  // bytes of a stage's two operand images
  constexpr int STAGE = SA + SB;
  constexpr int JA = SA / 1024 / 4, JB = SB / 1024 / 4;   // DMA instructions per wave and stage (2; 2 or 1)
  constexpr int NPA = BM / 4, NPB = BN / 4;          // 16-byte pieces per pixel row
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m_base = blockIdx.x * BM, d_base = blockIdx.y * BN, split = blockIdx.z;
  const int P = a.N * a.Hb * a.Wb;
  const int niter = TAPS ? ((a.N << (a.lw + a.lh)) + KP - 1) / KP : (P + KP - 1) / KP;
  const int per = (niter + a.splitk - 1) / a.splitk;
  const int it0 = split * per, it1 = min(niter, it0 + per);
  // the source tensor this tile's channels lie in (a virtual concat never straddles a tile: wgrad_mm_eligible)
  const int tap0 = m_base >> a.log2Gc, chm0 = m_base & a.gc_mask;       // (one tap: log2Gc = 30, tap0 = 0)
  const int gsrc = (!TAPS || chm0 < a.g.C[0]) ? 0 : 1, dsrc = (!TAPS || d_base < a.d.C[0]) ? 0 : 1;
  const int ldx = a.g.C[gsrc], ldd = a.d.C[dsrc];     // pixel pitches in floats
  const int gsub = gsrc ? a.g.C[0] : 0, dsub = dsrc ? a.d.C[0] : 0;
  (void)tap0;

  __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.g.ptr[gsrc], (unsigned)((size_t)a.N * a.Hgin * a.Wgin * ldx * 4));
  __amdgpu_buffer_rsrc_t rsD = make_rsrc(a.d.ptr[dsrc], (unsigned)((size_t)P * ldd * 4));
  // DMA lanes: instruction d of an operand covers 1024 / (4 * ROW) pixel rows; lane -> (pixel row of the stage, piece), rotated.
  // The rotation depends on (row & 3) only and a wave's instructions are 8 (4) rows apart: a lane's piece is the same in all of them.
  unsigned voA[JA], voB[JB];
  int kkA[JA], kkB[JB];
  int chA = 0, chB = 0, dhA = 0, dwA = 0;
  bool okA = false, okB = false;
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int d = wave + 4 * j;
    const int kk = d * (256 / BM) + lane / NPA, i = lane % NPA;
    const int m = m_base + 4 * ((i - 4 * (kk & 3)) & (NPA - 1));
    kkA[j] = kk;
    if (TAPS) {
      const int tap = m >> a.log2Gc, ch = m & a.gc_mask;
      okA = tap < a.ntaps && ch < a.Gc;
      chA = ch - gsub;
      int tdh = 0, tdw = 0;
#pragma unroll
      for (int t = 0; t < 16; ++t) if (t == tap) { tdh = a.taps.dh[t]; tdw = a.taps.dw[t]; }
      dhA = tdh; dwA = tdw;
      voA[j] = 0;
    } else {
      voA[j] = m < a.Gc ? (unsigned)((kk * ldx + m) * 4) : DMA_OOB;
    }
  }
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int d = wave + 4 * j;
    const int kk = d * (256 / BN) + lane / NPB, i = lane % NPB;
    const int ch = d_base + 4 * ((i - 4 * (kk & 3)) & (NPB - 1));
    kkB[j] = kk;
    if (TAPS) { okB = ch < a.Dc; chB = ch - dsub; voB[j] = 0; }
    else voB[j] = ch < a.Dc ? (unsigned)((kk * ldd + ch) * 4) : DMA_OOB;
  }
  // (the stage's pixel offset goes into the LANE offset: a raw buffer's range check covers the lane offset only, not the scalar one,
  // and the rows behind pixel P - 1 of the last stage must read zeros; out-of-range lanes are pinned so that the sum cannot wrap)
  auto issue = [&](int it, int stage) {
    uint4* la = reinterpret_cast<uint4*>(smem + stage * STAGE);
    uint4* lb = reinterpret_cast<uint4*>(smem + stage * STAGE + SA);
    if constexpr (TAPS) {
      const int mw = (1 << a.lw) - 1, mh = (1 << a.lh) - 1;
#pragma unroll
      for (int j = 0; j < JA; ++j) {
        const int sl = it * KP + kkA[j];
        const int r = sl & mw, q = (sl >> a.lw) & mh, n = sl >> (a.lw + a.lh);
        const int ih = q * a.s + dhA, iw = r * a.s + dwA;
        const bool ok = okA && r < a.Wb && q < a.Hb && n < a.N && (unsigned)ih < (unsigned)a.Hgin && (unsigned)iw < (unsigned)a.Wgin;
        dma16_buf(rsX, ok ? (unsigned)((((n * a.Hgin + ih) * a.Wgin + iw) * ldx + chA) * 4) : 0xFFFFFFF0u, 0, la + (wave + 4 * j) * 64);
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const int sl = it * KP + kkB[j];
        const int r = sl & mw, q = (sl >> a.lw) & mh, n = sl >> (a.lw + a.lh);
        const bool ok = okB && r < a.Wb && q < a.Hb && n < a.N;
        dma16_buf(rsD, ok ? (unsigned)((((n * a.Hb + q) * a.Wb + r) * ldd + chB) * 4) : 0xFFFFFFF0u, 0, lb + (wave + 4 * j) * 64);
      }
      return;
    }
    const unsigned sx = (unsigned)it * (unsigned)(KP * 4) * (unsigned)ldx, sd = (unsigned)it * (unsigned)(KP * 4) * (unsigned)ldd;
#pragma unroll
    for (int j = 0; j < JA; ++j) dma16_buf(rsX, voA[j] == DMA_OOB ? 0xFFFFFFF0u : voA[j] + sx, 0, la + (wave + 4 * j) * 64);
#pragma unroll
    for (int j = 0; j < JB; ++j) dma16_buf(rsD, voB[j] == DMA_OOB ? 0xFFFFFFF0u : voB[j] + sd, 0, lb + (wave + 4 * j) * 64);
  };

  // fragment addresses (bytes inside a stage): pixel row kq of a k-step, channel (c0 + lane % 16) rotated by 16 * kq floats
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 15, kq = lane >> 4;
  int adA[TC], adB[TP];
#pragma unroll
  for (int t = 0; t < TC; ++t) adA[t] = (kq * BM + ((wm * 64 + t * 16 + fi + 16 * kq) & (BM - 1))) * 4;
#pragma unroll
  for (int t = 0; t < TP; ++t) adB[t] = SA + (kq * BN + ((wn * (TP * 16) + t * 16 + fi + 16 * kq) & (BN - 1))) * 4;

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (it0 < it1) {
#pragma unroll
    for (int d = 0; d < NST - 1; ++d) if (it0 + d < it1) issue(it0 + d, d);
    int st = 0;
    for (int it = it0; it < it1; ++it) {
      if (it + NST - 2 < it1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (JA + JB)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // tail: fewer batches outstanding than the constant assumes
      __builtin_amdgcn_s_barrier();      // every wave's DMAs of this stage landed; every wave is done reading the previous stage
      asm volatile("" ::: "memory");
      if (it + NST - 1 < it1) issue(it + NST - 1, st == 0 ? NST - 1 : st - 1);
      const int sb = st * STAGE;
      // 4 k-steps of 4 pixels (row offset j * 4 rows as an immediate: A rows are 4 * BM bytes, B rows 4 * BN).  All fragment reads are
      // issued k-step by k-step, then every k-step waits only for ITS reads (LDS returns in order): the reads of the later steps land
      // under the MFMAs of the earlier ones
      float fa[4][TC], fb[4][TP];
      int ada[TC], adb[TP];
#pragma unroll
      for (int t = 0; t < TC; ++t) ada[t] = adA[t] + sb;
#pragma unroll
      for (int t = 0; t < TP; ++t) adb[t] = adB[t] + sb;
#define VP_RD_STEP(J)                                                                     \
      _Pragma("unroll") for (int t = 0; t < TC; ++t) fa[J][t] = lds_rd32<J * 4 * BM * 4>(ada[t]); \
      _Pragma("unroll") for (int t = 0; t < TP; ++t) fb[J][t] = lds_rd32<J * 4 * BN * 4>(adb[t]);
#define VP_MMA_STEP(J, W)                                                                 \
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(W) : "memory");                             \
      _Pragma("unroll") for (int t = 0; t < TC; ++t) asm volatile("" : "+v"(fa[J][t]));      \
      _Pragma("unroll") for (int t = 0; t < TP; ++t) asm volatile("" : "+v"(fb[J][t]));      \
      _Pragma("unroll") for (int tc = 0; tc < TC; ++tc)                                      \
        _Pragma("unroll") for (int tp = 0; tp < TP; ++tp)                                    \
          acc[tc][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[J][tc], fb[J][tp], acc[tc][tp], 0, 0, 0);
      // (the 4-bit lgkmcnt holds 15: the reads run one k-step ahead of the MFMAs, at most two k-steps outstanding)
      VP_RD_STEP(0) VP_RD_STEP(1)
      VP_MMA_STEP(0, TC + TP) VP_RD_STEP(2)
      VP_MMA_STEP(1, TC + TP) VP_RD_STEP(3)
      VP_MMA_STEP(2, TC + TP)
      VP_MMA_STEP(3, 0)
#undef VP_MMA_STEP
#undef VP_RD_STEP
      st = st == NST - 1 ? 0 : st + 1;
    }
  }

  // lane holds column (lane & 15), rows 4 * (lane >> 4) .. + 3 of every 16 x 16 block (the layout of wgrad_kernel: same epilogue)
  const int rowA0 = wm * 64, rowB0 = wn * (TP * 16);
  if (a.splitk == 1) {
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) {
        const int m0 = m_base + rowA0 + tc * 16 + 4 * (lane >> 4);
        const int d = d_base + rowB0 + tp * 16 + (lane & 15);
        if (d >= a.Dreal) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + e, tap = m >> a.log2Gc, gc = m & a.gc_mask;       // (one tap: tap = 0, gc = m)
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

}  // namespace

// float32, prologue-free operands whose channel counts are multiples of 4, tensors below 4 GiB.  One tap, stride 1: the plain form.
// k x k taps (or a strided one-tap layer): the gathering form, if no tile of either operand straddles a virtual concat.
static bool mm_common(const WgradArgs& a, int cfg) {
  if ((cfg != 0 && cfg != 1) || a.g.aff_a[0] || a.g.aff_a[1] || a.d.aff_a[0] || a.d.aff_a[1] || a.g.act != ACT_NONE || a.d.act != ACT_NONE) return false;
  if ((a.g.C[0] & 3) || (a.g.C[1] & 3) || (a.d.C[0] & 3) || (a.d.C[1] & 3) || a.Mpad % 128 || a.Dpad % (cfg == 0 ? 128 : 64)) return false;
  const unsigned long long pg = (unsigned long long)a.N * a.Hgin * a.Wgin, pd = (unsigned long long)a.N * a.Hb * a.Wb;
  const unsigned long long cg = a.g.C[0] > a.g.C[1] ? a.g.C[0] : a.g.C[1], cd = a.d.C[0] > a.d.C[1] ? a.d.C[0] : a.d.C[1];
  return pg * cg * 4 < 0xF0000000ull && pd * cd * 4 < 0xF0000000ull;
}
static bool mm_plain(const WgradArgs& a) {
  return a.ntaps == 1 && a.s == 1 && a.taps.dh[0] == 0 && a.taps.dw[0] == 0 && !a.g.C[1] && !a.d.C[1] && a.Hgin == a.Hb && a.Wgin == a.Wb;
}
bool wgrad_mm_eligible(const WgradArgs& a, int cfg) {
  if (!mm_common(a, cfg)) return false;
  if (mm_plain(a)) return true;
  const int bn = cfg == 0 ? 128 : 64;
  // a 16-byte piece inside one tap, a 128-row tile inside one source: Gc a multiple of 4; with two sources the boundary on a tile edge
  if (a.ntaps > 1 && (a.Gc & (a.Gc - 1))) return false;            // rows decode as (m >> log2Gc, m & gc_mask)
  if (a.g.C[1] && (a.g.C[0] % 128 || a.Gc < 128)) return false;
  if (a.d.C[1] && a.d.C[0] % bn) return false;
  if (a.Gc < 128 && 128 % a.Gc) return false;
  return ((long long)a.N << (a.lw + a.lh)) < (1ll << 30);
}

hipError_t launch_wgrad_mm(const WgradArgs& a, int cfg, hipStream_t st) {
  const int bn = cfg == 0 ? 128 : 64;
  dim3 grid(a.Mpad / 128, a.Dpad / bn, a.splitk);
  const size_t smem = 4 * (size_t)(16 * 128 * 4 + 16 * bn * 4);
  const bool taps = !mm_plain(a);
  auto k = cfg == 0 ? (taps ? wgrad_mm_kernel<4, true> : wgrad_mm_kernel<4, false>) : (taps ? wgrad_mm_kernel<2, true> : wgrad_mm_kernel<2, false>);
  static bool attr_done[4] = {false, false, false, false};
  const int vi = (cfg == 0 ? 0 : 2) + (taps ? 1 : 0);
  if (!attr_done[vi]) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_done[vi] = true; }
  hipLaunchKernelGGL(k, grid, dim3(256), smem, st, a);
  return hipGetLastError();
}

}  // namespace vp
