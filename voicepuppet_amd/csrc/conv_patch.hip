// igemm_patch_kernel: stride-1 convolutions (VGG 3x3, discriminator layer_4 4x4 and their backward-data passes) with the INPUT
// PATCH STAGED ONCE PER CHANNEL CHUNK.  The gather-per-tap kernels (conv_kernels.hip) DMA every input pixel into LDS once per tap
// (9 or 16 times) and are bound by that L2 -> LDS traffic, not by the matrix pipe.  Here a block owns a 2-D
// tile of TH x TW output pixels of one image; per 64-byte channel chunk the (TH + kh - 1) x (TW + kw - 1) input patch is DMA'd into
// LDS ONCE and all taps read their B fragments from shifted positions of it; only the weights (BC x 64 bytes per tap) stream per tap.
// DMA bytes per MAC drop 1.75x (256 ch x 16x16 px), 3.2x (128 ch x 16x32 px) and 5x (64 ch x 16x32 px).
//
// LDS image of the patch: pixel pp = py * PW + px owns 64 consecutive bytes = four 16-byte k-pieces, piece p stored at slot
// p ^ ((pp >> 2) & 3).  A B fragment = 16 consecutive pixels of one patch row starting ANYWHERE (the tap shift), lane (i, g) needs
// piece g of pixel pp0 + i.  It is read as two ds_read_b64 (each serviced in two 32-lane groups over a 256-byte bank window): lanes
// with even g take the low 8 bytes first, lanes with odd g the high 8 bytes first.  Inside a window the four pixels of a class
// (pp & 3) carry (pp >> 2) & 3 = all four values, so the 8 lanes of a class (4 pixels x 2 values of g) hit 8 distinct 8-byte slots:
// conflict-free for every shift.  The price - odd k-groups arrive with their two halves swapped - is paid once by the weight
// packer (PackDesc::kswap stores the same permutation of k in the A operand; a permutation of k common to A and B leaves the
// product unchanged).
//
// Pipeline: step s = (chunk c, tap t); weights of step s in ring stage s % NSTW (issued NSTW-1 steps ahead); patch of chunk c in
// buffer c & 1, the next chunk's patch is issued one DMA instruction per step over the first steps of chunk c (after the barrier
// that retires the last reads of the buffer it overwrites).  One barrier per step, counted vmcnt, every wave issues its share of
// the DMAs; 8 waves (two per SIMD), 128 (or 64) accumulator registers per lane, one block per CU.
#include <stdlib.h>

#include "igemm_device.h"
#include "launch.h"
#include "conv_ops.h"

namespace vp {

// output pixel of tile row `row` (pixel block row / 16, lane row % 16) -> offset into Y, -1 outside the image
template <int TW>
struct PatchTilePix {
  const IgemmArgs& a; int n, y0, x0;
  __device__ __forceinline__ long long operator()(int row) const {
    constexpr int BPR = TW / 16;                         // 16-pixel blocks per tile row
    const int pb = row >> 4, i = row & 15;
    const int y = y0 + pb / BPR, x = x0 + (pb % BPR) * 16 + i;
    if (y >= a.Hg || x >= a.Wg) return -1;
    const long long off = (((long long)n * a.Hof + y) * a.Wof + x) * a.ldY;
    return (off << 8) | (long long)(n / a.ref_group_n);
  }
  // 2x2 max pool of the tile: pooled pixel (row pr, column pc) of this tile -> element offset in the pooled image, -1 outside.
  // Meaningful for 16-pixel-wide tiles (the staged epilogue walks 8 pooled pixels per pooled row) and even image sizes.
  static constexpr bool HAS_POOL = (TW == 16);
  __device__ __forceinline__ long long pool(int pr, int pc) const {
    const int y = (y0 >> 1) + pr, x = (x0 >> 1) + pc;
    if (y >= (a.Hg >> 1) || x >= (a.Wg >> 1)) return -1;
    return (((long long)n * (a.Hg >> 1) + y) * (a.Wg >> 1) + x) * a.ldY;
  }
};

// fragment reads behind __restrict__ parameters: keeps hipcc from draining vmcnt in front of LDS reads that may alias a pending
// LDS-DMA (EXPERIMENTS.md, "double-buffered register fragments", has the story); the counted vmcnt + barrier of the loop is what orders them
template <int TC, int TP>
__device__ __forceinline__ void patch_frag_read(const uint4* __restrict__ pa, const char* __restrict__ pbuf, const int (&boff)[TP],
                                                uint4 (&fa)[TC], uint4 (&fb)[TP]) {
#pragma unroll
  for (int t = 0; t < TP; ++t) {
    const uint2 r1 = *reinterpret_cast<const uint2*>(pbuf + boff[t]);
    const uint2 r2 = *reinterpret_cast<const uint2*>(pbuf + (boff[t] ^ 8));
    fb[t] = make_uint4(r1.x, r1.y, r2.x, r2.y);
  }
#pragma unroll
  for (int t = 0; t < TC; ++t) fa[t] = pa[t * 64];
}

template <typename T, int WC, int WP, int TC, int TP, int TH, int TW, int NSTW, int STATS, int OCC>
__global__ __launch_bounds__(WC * WP * 64, OCC) void igemm_patch_kernel(const IgemmArgs a) {
  constexpr int E = Elem<T>::E, KC = 4 * E;
  constexpr int NW = WC * WP, NT = NW * 64;
  static_assert(NW == 8, "eight waves");
  constexpr int BC = WC * TC * 16, BP = TH * TW;
  static_assert(BP == WP * TP * 16, "pixel blocks of the tile = pixel blocks of the waves");
  constexpr int NBA = BC / 16;
  static_assert(NBA % NW == 0 || NBA == 4, "weight DMAs: whole instructions per wave (64-row tiles: half an instruction per wave)");
  constexpr int JA = (NBA + NW - 1) / NW;
  constexpr int PPAD = ((TH + 3) * (TW + 3) + 127) / 128 * 128;      // patch pixels, padded to whole DMA rounds of the 8 waves
  constexpr int JP = PPAD / 128;                                     // patch DMA instructions per wave
  constexpr int WST = 4 * BC;                                        // uint4 slots of one weight stage
  constexpr int PBUF = 4 * PPAD;                                     // uint4 slots of one patch buffer
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  uint4* lpatch = lds + NSTW * WST;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c_base = blockIdx.y * BC;
  const int tiles_x = (a.Wg + TW - 1) / TW, tiles_y = (a.Hg + TH - 1) / TH;
  const int bt = blockIdx.x;
  const int n = bt / (tiles_x * tiles_y);
  const int trem = bt - n * (tiles_x * tiles_y);
  const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
  // tap t = r * kw + c  ->  (dh, dw) = (p_dhf + r * p_dhs, p_dwf + c * p_dws)
  const int kw = a.p_kw, kh = a.ntaps / kw;
  const int dh0 = a.p_dhs > 0 ? a.p_dhf : a.p_dhf - (kh - 1), dw0 = a.p_dws > 0 ? a.p_dwf : a.p_dwf - (kw - 1);
  const int PW = TW + kw - 1, PH = TH + kh - 1;
  const int npatch = PW * PH;
  const unsigned es = sizeof(T);
  const int C0 = a.x.C[0];
  const int nchunkc = C0 / KC;                  // channel chunks
  const int S = nchunkc * a.ntaps;              // steps

  __amdgpu_buffer_rsrc_t rsW = make_rsrc(reinterpret_cast<const T*>(a.Wp), 0xFFFFFFFFu);
  __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * C0 * es));

  // patch DMA lanes: instruction j of this wave covers patch pixels (wave + 8j) * 16 .. + 15, lane -> (pixel, slot)
  unsigned pvo[JP];
#pragma unroll
  for (int j = 0; j < JP; ++j) {
    const int pp = (wave + NW * j) * 16 + (lane >> 2);
    const int py = pp / PW, px = pp - py * PW;
    const int ih = y0 + dh0 + py, iw = x0 + dw0 + px;
    const bool ok = pp < npatch && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
    const int piece = (lane & 3) ^ ((pp >> 2) & 3);
    pvo[j] = ok ? (unsigned)((((n * a.Hin + ih) * a.Win + iw) * C0 + piece * E) * es) : DMA_OOB;
  }
  // weight DMA lanes (rb_swz image, as in igemm_dma_kernel).  64-row tiles have four 16-row blocks for eight waves: every wave
  // moves HALF a block (lanes 0-31: 8 rows), so that all waves keep issuing the same number of DMAs per step (the counted vmcnt
  // below relies on it)
  constexpr bool HALFW = NBA < NW;
  unsigned wvo[JA];
  {
    const int r = HALFW ? (wave & 1) * 8 + (lane >> 2) : lane >> 2;       // row inside the 16-row block
    const int g = (lane & 3) ^ rb_swz(r & 15);
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const int blk = HALFW ? (wave >> 1) : wave + NW * j;
      wvo[j] = (unsigned)(((c_base + blk * 16 + r) * KC + g * E) * es);
    }
  }
  const unsigned wstep = (unsigned)(a.wp_rows * KC * es);
  // weight issue cursor: step (wc_c, wc_t) -> packed chunk index wc_t * nchunkc + wc_c (tap-major K)
  int wc_c = 0, wc_t = 0, wc_stage = 0;
  auto issue_w = [&]() {
    const unsigned wso = (unsigned)(wc_t * nchunkc + wc_c) * wstep;
    uint4* la = lds + wc_stage * WST;
    if constexpr (HALFW) {
      if (lane < 32) dma16_buf(rsW, wvo[0], wso, la + (wave >> 1) * 64 + (wave & 1) * 32);
    } else {
#pragma unroll
      for (int j = 0; j < JA; ++j) dma16_buf(rsW, wvo[j], wso, la + (wave + NW * j) * 64);
    }
    if (++wc_t == a.ntaps) { wc_t = 0; ++wc_c; }
    wc_stage = wc_stage == NSTW - 1 ? 0 : wc_stage + 1;
  };
  auto issue_p = [&](int chunk, int j) {        // j: compile-time after unrolling at the call sites
    uint4* lb = lpatch + (chunk & 1) * PBUF;
    dma16_buf(rsX, pvo[j], (unsigned)(chunk * KC) * es, lb + (wave + NW * j) * 64);
  };

  const int wc = wave / WP, wpi = wave - wc * WP;
  const int blkA0 = wc * TC, blkB0 = wpi * TP;
  const int fi = lane & 15, fg = lane >> 4;
  const int so = fi * 4 + (fg ^ rb_swz(fi));
  // B fragments: patch pixel of (pixel block t, lane) at tap offset 0
  int bpp[TP];
#pragma unroll
  for (int t = 0; t < TP; ++t) {
    constexpr int BPR = TW / 16;
    const int pb = blkB0 + t;
    bpp[t] = (pb / BPR) * PW + (pb % BPR) * 16 + fi;
  }

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: the whole first patch, then the first NSTW-1 weight stages
#pragma unroll
  for (int j = 0; j < JP; ++j) issue_p(0, j);
#pragma unroll
  for (int d = 0; d < NSTW - 1; ++d) if (d < S) issue_w();

  int c = 0, t = 0, st = 0;
  int tr = 0, tcol = 0;                          // tap row / column of step s
  for (int s = 0; s < S; ++s) {
    if (s + NSTW - 2 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTW - 2) * JA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NSTW - 1 < S) issue_w();
    if (t < JP && c + 1 < nchunkc) {
#pragma unroll
      for (int j = 0; j < JP; ++j) if (j == t) issue_p(c + 1, j);
    }
    // fragments of step s
    const int tapoff = (a.p_dhf + tr * a.p_dhs - dh0) * PW + (a.p_dwf + tcol * a.p_dws - dw0);
    int boff[TP];
#pragma unroll
    for (int q = 0; q < TP; ++q) {
      const int pp = bpp[q] + tapoff;
      boff[q] = (pp << 6) + (((fg ^ (pp >> 2)) & 3) << 4) + ((fg & 1) << 3);
    }
    uint4 fa[TC], fb[TP];
    patch_frag_read<TC, TP>(lds + st * WST + blkA0 * 64 + so, reinterpret_cast<const char*>(lpatch + (c & 1) * PBUF), boff, fa, fb);
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) acc[tc][tp] = mma16<T>(fa[tc], fb[tp], acc[tc][tp]);
    st = st == NSTW - 1 ? 0 : st + 1;
    if (++tcol == kw) { tcol = 0; ++tr; }
    if (++t == a.ntaps) { t = 0; tr = 0; ++c; }
  }

  constexpr int RINGB = (NSTW * WST + 2 * PBUF) * 16;
  constexpr int NPASS = epi_passes(BC, BP, WP, RINGB);
  staged_epilogue<T, TC, TP, BC, BP, NPASS, NT, STATS>(a, PatchTilePix<TW>{a, n, y0, x0}, c_base, blkA0, blkB0, acc, smem, bt, 0);
}

// OCC: waves per SIMD the register allocation must allow (2: one 8-wave block per CU, 4: two)
template <typename T, int WC, int WP, int TC, int TP, int TH, int TW, int NSTW, int OCC>
static hipError_t launch_patch_t(const IgemmArgs& b, hipStream_t st) {
  constexpr int BC = WC * TC * 16, BP = TH * TW;
  constexpr int PPAD = ((TH + 3) * (TW + 3) + 127) / 128 * 128;
  constexpr int RINGB = (NSTW * 4 * BC + 2 * 4 * PPAD) * 16;
  constexpr int NPE = epi_passes(BC, BP, WP, RINGB);
  size_t sm = RINGB;
  const size_t se = (size_t)(BP / NPE) * (BC * 4 + 16) + (BP / NPE) * 8;
  if (se > sm) sm = se;
  const int tiles = b.N * ((b.Hg + TH - 1) / TH) * ((b.Wg + TW - 1) / TW);
  dim3 grid(tiles, b.CoutPad / BC, 1);
  if (b.bst_y || b.bst_y2) return hipErrorInvalidValue;      // (backward sums in the epilogue: the unrolled 4x4 kernel and the 2x2-tap kernel only; the host asks accordingly)
  auto kern = b.bn_part ? igemm_patch_kernel<T, WC, WP, TC, TP, TH, TW, NSTW, 1, OCC> : igemm_patch_kernel<T, WC, WP, TC, TP, TH, TW, NSTW, 0, OCC>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(kern, grid, dim3(512), sm, st, b);
  return hipGetLastError();
}

// bc: channel rows of the tile, bp: pixels of the tile (128: 8 x 16, 256: 16 x 16, 512: 16 x 32).  The 256-pixel tiles of the 128- and 64-row
// variants run TWO blocks per CU (64 / 32 accumulator registers, 72 / 60 KB of LDS): the blocks are not in lockstep with each
// other, so one block's barrier / DMA-issue / ds_read phase overlaps the other's MFMAs.
hipError_t launch_igemm_patch(const IgemmArgs& a, int is_bf16, int bc, int bp, hipStream_t st) {
  if (a.patch == 2) return launch_igemm_patch2(a, is_bf16, bc, bp, st);                                  // 2x2-tap parity classes (conv_patch2.hip)
  if (patch3_knob() && patch3_eligible(a, is_bf16)) return launch_igemm_patch3(a, is_bf16, bc, bp, st);   // 3x3: the unrolled schedule
  IgemmArgs b = a;
  b.vec_epi = 1;
#define VP_PATCH_GO(WC, WP, TC, TP, TH, TW, NS, OCC) \
  (is_bf16 ? launch_patch_t<bf16, WC, WP, TC, TP, TH, TW, NS, OCC>(b, st) : launch_patch_t<float, WC, WP, TC, TP, TH, TW, NS, OCC>(b, st))
  if (bc == 256) return bp == 128 ? VP_PATCH_GO(2, 4, 8, 2, 8, 16, 3, 4) : VP_PATCH_GO(2, 4, 8, 4, 16, 16, 4, 2);
  if (bc == 128) return bp == 512 ? VP_PATCH_GO(1, 8, 8, 4, 16, 32, 4, 2) : VP_PATCH_GO(2, 4, 4, 4, 16, 16, 3, 4);
  return bp == 512 ? VP_PATCH_GO(1, 8, 4, 4, 16, 32, 4, 2) : VP_PATCH_GO(2, 4, 2, 4, 16, 16, 3, 4);
#undef VP_PATCH_GO
}

}  // namespace vp
