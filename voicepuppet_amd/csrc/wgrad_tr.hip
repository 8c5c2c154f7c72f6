// wgrad_tr_kernel: backward-weight as an LDS-DMA GEMM with hardware transpose reads (bf16).
//
//   dW[tap][g][d] = sum over pixels k = (n, q, r) of  G[pixel (+) tap, g] * D[pixel, d]
//
// The reduction index is the PIXEL - the slow index of both NHWC operands - while an MFMA lane needs 8 consecutive k of one row.
// wgrad_kernel (conv_kernels.hip) gets there with register loads, 8x8 v_perm transposes and ds_writes in the loop.  Here both tiles
// go global -> LDS by LDS-DMA exactly as they lie in memory ([pixel][channel]: whole 256 / 512-byte rows per pixel, no vector ALU on
// the data, no ds_write), and the fragments come out of LDS with ds_read_b64_tr_b16: in each 16-lane group, lane a supplies the
// address of ONE 8-byte chunk (4 channels of one pixel) and lane t receives element t % 4 of the chunks of lanes 4j + t / 4,
// j = 0..3 - so with lane a addressing (pixel k0 + a / 4, channels 4 (a % 4) ..) lane t ends up with channel t of pixels k0 .. k0+3:
// two reads give the 8 consecutive k of an MFMA 16x16x32 fragment (scripts/probes/tr_wgrad_probe.hip checks this chain on the GPU).
//
// LDS image of a tile: pixel p (0..31 inside the K chunk) owns one row of 16-byte pieces, piece c stored at slot c ^ swz(p),
// swz(p) = ((p & 3) | ((p >> 3) & 1) << 2) << 1: the 32 lanes of a read group touch 8 pixels x 2 pieces x 2 halves = all 64 banks
// (conflict-free for both tile widths); the DMA lanes apply the same permutation to their SOURCE addresses.
//
// K walks the padded grid [N][2^lh][2^lw] (slots outside the image, padding taps and ragged channels read the zero page), split
// over gridDim.z with the same slab + deterministic reduce as wgrad_kernel.  Tile 256 rows (tap, g) x 128 columns d, 8 waves of
// 64 x 64, 64 accumulator registers, 3-deep ring of 24 KB stages, counted vmcnt, one barrier per chunk, two blocks per CU.
#include <stdlib.h>

#include "conv_ops.h"
#include "igemm_device.h"
#include "launch.h"

namespace vp {

__device__ __forceinline__ int tr_swz(int p) { return ((p & 3) | (((p >> 3) & 1) << 2)) << 1; }

typedef short v4s16 __attribute__((ext_vector_type(4)));

// fragment reads behind a __restrict__ parameter (found on the register-double-buffered tile of round 2, EXPERIMENTS.md: keeps the waitcnt pass from draining vmcnt in front of LDS reads
// while LDS-DMAs are in flight); `off`: byte offset of the first 4-pixel group of this lane's 8 consecutive k, the second one is
// four pixel rows further
template <int ROWB>   // bytes of one pixel row of the tile
__device__ __forceinline__ uint4 tr_frag8(const char* __restrict__ stage, int off) {
  typedef __attribute__((address_space(3))) v4s16 lds_v4;
  const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(stage + off));
  const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(stage + off + 4 * ROWB));
  union { v4s16 v[2]; uint4 u; } f;
  f.v[0] = lo; f.v[1] = hi;
  return f.u;
}

// one DMA lane's share of an operand tile, fixed over the K loop: which tensor / channels / tap; only the pixel moves
struct TrTask {
  const bf16* base;     // source pointer + channel offset
  int C;                // channels of that source (pixel stride)
  int s, dh, dw;        // source pixel = (q * s + dh, r * s + dw)
  int Hs, Ws;
  int p;                // pixel slot inside the 32-pixel chunk
  bool ok;
};

__device__ __forceinline__ TrTask tr_task(const PixSrc& src, int ch, bool ok, int s, int dh, int dw, int Hs, int Ws, int p, const void* zeros) {
  TrTask t;
  const bool second = ch >= src.C[0];
  t.C = second ? src.C[1] : src.C[0];
  t.base = reinterpret_cast<const bf16*>(second ? src.ptr[1] : src.ptr[0]) + (second ? ch - src.C[0] : ch);
  t.s = s; t.dh = dh; t.dw = dw; t.Hs = Hs; t.Ws = Ws; t.p = p; t.ok = ok;
  if (!ok) { t.base = reinterpret_cast<const bf16*>(zeros); t.C = 0; }
  return t;
}

__device__ __forceinline__ const void* tr_src(const WgradArgs& a, const TrTask& t, int it) {
  const int slot = it * 32 + t.p;
  const int r = slot & ((1 << a.lw) - 1), q = (slot >> a.lw) & ((1 << a.lh) - 1), n = slot >> (a.lw + a.lh);
  const int ih = q * t.s + t.dh, iw = r * t.s + t.dw;
  const bool ok = t.ok && n < a.N && q < a.Hb && r < a.Wb && (unsigned)ih < (unsigned)t.Hs && (unsigned)iw < (unsigned)t.Ws;
  return ok ? (const void*)(t.base + ((size_t)(n * t.Hs + ih) * t.Ws + iw) * t.C) : a.zeros;
}

// EXACT (fast path only): the padded K grid IS the dY image (Hb = 2^lh, Wb = 2^lw): no slot of a chunk lies outside it, the dY
// lanes need no per-chunk work at all and the gathered lanes only their two tap range checks
template <int WM, int WN, int TC, int TP, int NST, bool FAST, bool EXACT = false>
__global__ __launch_bounds__(WM * WN * 64, (TC * TP > 16 ? 2 : 4)) void wgrad_tr_kernel(const WgradArgs a) {      // (256 x 256 tile: 128 accumulator registers, one block per CU)
  constexpr int NW = WM * WN;
  static_assert(NW == 8, "eight waves");
  constexpr int BM = WM * TC * 16, BN = WN * TP * 16;
  constexpr int PA = BM / 8, PB = BN / 8;                 // 16-byte pieces per pixel row
  constexpr int ASTG = 32 * PA, BSTG = 32 * PB, STG = ASTG + BSTG;   // uint4 slots
  constexpr int JA = ASTG / 64 / NW, JB = BSTG / 64 / NW;            // DMA instructions per wave per chunk
  static_assert(JA * NW * 64 == ASTG && JB * NW * 64 == BSTG, "whole DMA instructions per wave");
  constexpr int J = JA + JB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware block order.  Workgroups go round-robin over the 8 XCDs in linear-id order and every XCD has its own L2; the blocks
  // that share operand bytes are the (M tile, D tile) blocks of ONE K split (same pixels of both tensors).  With a multiple of 8
  // splits, linear id L is re-read as: XCD x = L % 8 owns the splits x, x + 8, ... and walks all tiles of one split before the next
  // - a split's pixels are then fetched into one L2 instead of eight (r02 PMC: 1.2 GB of L2 misses per launch for layer_4's 145 MB).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd_remap && (gridDim.z & 7) == 0) {
    const int ntile = gridDim.x * gridDim.y;
    const int L = bx + by * gridDim.x + bz * ntile;
    const int xcd = L & 7, j = L >> 3;
    bz = xcd + 8 * (j / ntile);
    const int t = j % ntile;
    bx = t % gridDim.x; by = t / gridDim.x;
  }
  const int m_base = bx * BM, d_base = by * BN, split = bz;
  const int niter = (a.N << (a.lw + a.lh)) / 32;          // host guarantees 2^(lw+lh) * N is a multiple of 32
  const int per = (niter + a.splitk - 1) / a.splitk;
  const int it0 = split * per, it1 = min(niter, it0 + per);

  // ---- this lane's DMA tasks ----
  TrTask ta[JA], tb[JB];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int idx = (wave + NW * j) * 64 + lane;          // slot index inside the A stage
    const int p = idx / PA, c = (idx % PA) ^ tr_swz(p);
    const int m = m_base + c * 8;
    const int tap = m >> a.log2Gc, ch = m & a.gc_mask;
    int tdh = 0, tdw = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) if (t == tap) { tdh = a.taps.dh[t]; tdw = a.taps.dw[t]; }
    ta[j] = tr_task(a.g, ch, tap < a.ntaps && ch < a.g.C[0] + a.g.C[1], a.s, tdh, tdw, a.Hgin, a.Wgin, p, a.zeros);
  }
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int idx = (wave + NW * j) * 64 + lane;
    const int p = idx / PB, c = (idx % PB) ^ tr_swz(p);
    const int ch = d_base + c * 8;
    tb[j] = tr_task(a.d, ch, ch < a.Dc, 1, 0, 0, a.Hb, a.Wb, p, a.zeros);
  }
  // ---- fast path: buffer-descriptor DMAs with a SCALAR per-chunk offset ----
  // A chunk is 32 consecutive slots starting at a multiple of 32, so slot = u + p splits into a block-uniform part
  // (nu, qu, ru) and a lane-constant part (pq, pr) without carries; the address splits the same way:
  //   [(nu*Hs + qu*s)*Ws + ru*s]*C  (scalar, per chunk)  +  [(pq*s + dh)*Ws + pr*s + dw]*C + ch  (per lane, once)
  // and only the four range checks remain vector work per chunk (invalid lanes get an out-of-range offset: the DMA writes zeros).
  // Needs ONE source tensor per operand tile (the channel tile does not straddle a virtual concat) and offsets below 2 GiB; the
  // descriptor base sits `guard` bytes in front of the tensor so that lane offsets of negative taps stay non-negative.
  const int W2 = 1 << a.lw;
  const bool gsecond = a.g.C[1] > 0 && (m_base & (a.Gc - 1)) >= a.g.C[0];
  const bool dsecond = a.d.C[1] > 0 && d_base >= a.d.C[0];
  const int gC = gsecond ? a.g.C[1] : a.g.C[0], dC = dsecond ? a.d.C[1] : a.d.C[0];
  constexpr bool fast = FAST;          // host decision (launch_wgrad_tr): one source per operand tile, chunks inside one image, < 2 GiB
  const unsigned gguard = (unsigned)((2 * a.Wgin + 2) * gC * 2);                      // |dh| <= 2 rows, |dw| <= 2 pixels in front
  __amdgpu_buffer_rsrc_t rsG = make_rsrc((const char*)(gsecond ? a.g.ptr[1] : a.g.ptr[0]) - gguard,
                                         (unsigned)((size_t)a.N * a.Hgin * a.Wgin * gC * 2) + gguard);
  __amdgpu_buffer_rsrc_t rsD = make_rsrc(dsecond ? a.d.ptr[1] : a.d.ptr[0], (unsigned)((size_t)a.N * a.Hb * a.Wb * dC * 2));
  int fq[JA], fB[JA], fr[JA], fD[JA];      // lane constants of the gathered tasks: pq, pq*s + dh, pr, pr*s + dw
  unsigned fvo[JA];
  bool fok[JA];
  int bq[JB], br[JB];
  unsigned bvo[JB];
  bool bok[JB];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const TrTask& t = ta[j];
    const int pq = W2 <= 32 ? t.p >> a.lw : 0, pr = W2 <= 32 ? t.p & (W2 - 1) : t.p;
    fq[j] = pq; fB[j] = pq * t.s + t.dh; fr[j] = pr; fD[j] = pr * t.s + t.dw;
    const int chl = (int)(t.base - reinterpret_cast<const bf16*>(gsecond ? a.g.ptr[1] : a.g.ptr[0]));   // channel offset inside the source
    fvo[j] = gguard + (unsigned)(((fB[j] * a.Wgin + fD[j]) * gC + chl) * 2);
    fok[j] = t.ok;
    if (EXACT && !t.ok) fvo[j] = DMA_OOB;
  }
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const TrTask& t = tb[j];
    bq[j] = W2 <= 32 ? t.p >> a.lw : 0; br[j] = W2 <= 32 ? t.p & (W2 - 1) : t.p;
    const int chl = (int)(t.base - reinterpret_cast<const bf16*>(dsecond ? a.d.ptr[1] : a.d.ptr[0]));
    bvo[j] = (unsigned)(((bq[j] * a.Wb + br[j]) * dC + chl) * 2);
    bok[j] = t.ok;
    if (EXACT && !t.ok) bvo[j] = DMA_OOB;
  }
  auto issue_fast = [&](int it, int stage) {
    uint4* la = lds + stage * STG;
    uint4* lb = la + ASTG;
    const int u = it * 32;
    const int ru = u & (W2 - 1), qu = (u >> a.lw) & ((1 << a.lh) - 1), nu = u >> (a.lw + a.lh);
    const bool nok = nu < a.N;
    const unsigned gso = (unsigned)((((nu * a.Hgin + qu * a.s) * a.Wgin + ru * a.s) * gC) * 2);
    const unsigned dso = (unsigned)((((nu * a.Hb + qu) * a.Wb + ru) * dC) * 2);
    const int qs = qu * a.s, rs = ru * a.s;
    if constexpr (EXACT) {
      // (it < niter = N * 2^(lh+lw) / 32: nu < N; fok / bok are folded into fvo / bvo)
#pragma unroll
      for (int j = 0; j < JA; ++j) {
        const bool ok = (unsigned)(qs + fB[j]) < (unsigned)a.Hgin && (unsigned)(rs + fD[j]) < (unsigned)a.Wgin;
        dma16_buf(rsG, ok ? fvo[j] : DMA_OOB, gso, la + (wave + NW * j) * 64);
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) dma16_buf(rsD, bvo[j], dso, lb + (wave + NW * j) * 64);
    } else {
#pragma unroll
      for (int j = 0; j < JA; ++j) {
        const bool ok = fok[j] && nok && qu + fq[j] < a.Hb && ru + fr[j] < a.Wb &&
                        (unsigned)(qs + fB[j]) < (unsigned)a.Hgin && (unsigned)(rs + fD[j]) < (unsigned)a.Wgin;
        dma16_buf(rsG, ok ? fvo[j] : DMA_OOB, gso, la + (wave + NW * j) * 64);
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const bool ok = bok[j] && nok && qu + bq[j] < a.Hb && ru + br[j] < a.Wb;
        dma16_buf(rsD, ok ? bvo[j] : DMA_OOB, dso, lb + (wave + NW * j) * 64);
      }
    }
  };
  auto issue = [&](int it, int stage) {
    if constexpr (fast) { issue_fast(it, stage); return; }
    uint4* la = lds + stage * STG;
    uint4* lb = la + ASTG;
#pragma unroll
    for (int j = 0; j < JA; ++j) dma16(tr_src(a, ta[j], it), la + (wave + NW * j) * 64);
#pragma unroll
    for (int j = 0; j < JB; ++j) dma16(tr_src(a, tb[j], it), lb + (wave + NW * j) * 64);
  };

  // ---- fragment addresses of this lane ----
  const int wm = wave / WN, wn = wave - wm * WN;
  const int g = lane >> 4, t16 = lane & 15;
  const int fp = 8 * g + (t16 >> 2), q2 = t16 & 3, fsw = tr_swz(fp);
  int offA[TC], offB[TP];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc) offA[tc] = fp * (BM * 2) + (((2 * (wm * TC + tc) + (q2 >> 1)) ^ fsw) << 4) + (q2 & 1) * 8;
#pragma unroll
  for (int tp = 0; tp < TP; ++tp) offB[tp] = ASTG * 16 + fp * (BN * 2) + (((2 * (wn * TP + tp) + (q2 >> 1)) ^ fsw) << 4) + (q2 & 1) * 8;

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (it0 < it1) {
#pragma unroll
    for (int dd = 0; dd < NST - 1; ++dd) if (it0 + dd < it1) issue(it0 + dd, dd);
    int st = 0;
    for (int it = it0; it < it1; ++it) {
      if (it + NST - 2 < it1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * J) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int stn = st == 0 ? NST - 1 : st - 1;           // the stage chunk it-1 used
      if (it + NST - 1 < it1) issue(it + NST - 1, stn);
      const char* stage = smem + (size_t)st * STG * 16;
      uint4 fa[TC], fb[TP];
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) fb[tp] = tr_frag8<BN * 2>(stage, offB[tp]);
#pragma unroll
      for (int tc = 0; tc < TC; ++tc) fa[tc] = tr_frag8<BM * 2>(stage, offA[tc]);
#pragma unroll
      for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) acc[tc][tp] = mma16<bf16>(fa[tc], fb[tp], acc[tc][tp]);
      st = st == NST - 1 ? 0 : st + 1;
    }
  }

  // lane holds d column (lane & 15), rows 4 * (lane >> 4) .. + 3 of every 16 x 16 tile
  if (a.splitk == 1) {
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) {
        const int m0 = m_base + (wm * TC + tc) * 16 + 4 * (lane >> 4);
        const int d = d_base + (wn * TP + tp) * 16 + (lane & 15);
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
      const int m0 = m_base + (wm * TC + tc) * 16 + 4 * (lane >> 4);
      const int d = d_base + (wn * TP + tp) * 16 + (lane & 15);
      float* pp = a.partial + ((size_t)split * a.Mpad + m0) * a.Dpad + d;
#pragma unroll
      for (int e = 0; e < 4; ++e) pp[(size_t)e * a.Dpad] = acc[tc][tp][e];
    }
}

template <int WM, int WN, int TC, int TP>
static hipError_t launch_wgrad_tr_t(const WgradArgs& a, hipStream_t st, const char** variant) {
  constexpr int NST = 3;
  constexpr int BM = WM * TC * 16, BN = WN * TP * 16;
  const size_t smem = (size_t)NST * 32 * (BM / 8 + BN / 8) * 16;
  dim3 grid(a.Mpad / BM, a.Dpad / BN, a.splitk);
  WgradArgs b = a;
  const size_t gbytes = (size_t)a.N * a.Hgin * a.Wgin * (a.g.C[0] > a.g.C[1] ? a.g.C[0] : a.g.C[1]) * 2;
  const size_t dbytes = (size_t)a.N * a.Hb * a.Wb * (a.d.C[0] > a.d.C[1] ? a.d.C[0] : a.d.C[1]) * 2;
  bool fast = gbytes < 0x60000000ull && dbytes < 0x60000000ull && a.lw + a.lh >= 5;     // a 32-slot chunk stays inside one image
  if (a.g.C[1] > 0) fast = fast && a.Gc >= BM && a.g.C[0] % BM == 0;                                // an operand tile never straddles a virtual concat
  if (a.d.C[1] > 0) fast = fast && a.d.C[0] % BN == 0;
  b.fast_tr = fast ? 1 : 0;
  b.xcd_remap = 1;
  const bool exact = fast && a.Hb == (1 << a.lh) && a.Wb == (1 << a.lw);
  if (variant) *variant = exact ? "tr_exact" : fast ? "tr_fast" : "tr";      // the template instance, for the profile's class names
  auto kern = exact ? wgrad_tr_kernel<WM, WN, TC, TP, NST, true, true> : fast ? wgrad_tr_kernel<WM, WN, TC, TP, NST, true> : wgrad_tr_kernel<WM, WN, TC, TP, NST, false>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, b);
  return hipGetLastError();
}

// wgrad cfg 5: 256 rows x 128 columns; cfg 6: 128 x 128 (the 6- / 3-channel input layers: 16 taps x 8 padded channels = 128 rows,
// 64 real columns - HBM-bound, the point is the loader: LDS-DMA instead of register loads + 8x8 transposes)
hipError_t launch_wgrad_tr(const WgradArgs& a, hipStream_t st, const char** variant, int* tile_cols) {
  if (tile_cols) *tile_cols = 128;
  // 256 x 256 tile (round 5 experiment): half the operand fill per MFMA of the 256 x 128 tile, one block per CU; where it still fills the CUs
  // (and the dense operand's 256-column tile must not straddle a virtual concat: the fast loader)
  if (wgrad_big_knob() && a.Mpad % 256 == 0 && a.Dpad % 256 == 0 && (long long)(a.Mpad / 256) * (a.Dpad / 256) * a.splitk >= 256 &&
      (a.d.C[1] == 0 || a.d.C[0] % 256 == 0) && a.lw + a.lh >= 5) {       // (few-pixel layers keep the 128-column tile: they run the generic loader)
    if (tile_cols) *tile_cols = 256;
    return launch_wgrad_tr_t<4, 2, 4, 8>(a, st, variant);
  }
  if (a.Mpad % 256 == 0) return launch_wgrad_tr_t<4, 2, 4, 4>(a, st, variant);
  return launch_wgrad_tr_t<4, 2, 2, 4>(a, st, variant);
}

}  // namespace vp
