// igemm_patch3_kernel: the 3x3 / stride-1 form of the patch kernel (conv_patch.hip) with the whole (chunk pair x tap) schedule
// unrolled.  PMC counters of the generic kernel on VGG conv1_2 (profiles/r02_pmc_sq_patch.txt): 7.6 vector-ALU and 9 scalar
// instructions per MFMA - fragment addresses (a swizzle of pixel index + tap offset), tap / ring / chunk cursors, a branch tree for
// the per-step patch DMA - and a vmcnt that waited for the patch DMA issued one step earlier.  Here
//   * the patch swizzle depends on the COLUMN of a patch pixel only (slot = piece ^ ((px >> 2) & 3)): a tap's row shift is a
//     plain byte offset.  With the taps unrolled the row shift, the buffer of the chunk and the ring stage are instruction
//     immediates; the lane part of a B address is one of 3 (column shifts) x TP precomputed registers, of an A address one
//     register.  A step issues NO vector ALU and a handful of scalar instructions;
//   * the counted vmcnt of every step is exact (a compile-time function of the tap): only the weights of the step, never a patch
//     DMA issued in the last two steps, are waited for;
//   * patches first in LDS (their read offsets fit the 16-bit DS immediate), the weight ring behind them.
// Bank behaviour of the fragment reads is that of conv_patch.hip: of the 16 consecutive pixels of a B fragment, the four that share
// an address class (pp & 3) have four different (px >> 2) & 3, so a 32-lane service group of a ds_read_b64 hits 32 distinct 8-byte
// slots for every column shift.  Two chunks (K = 2 x 64 bytes per tap) per trip: channel counts are multiples of 64 (bf16) / 32 (f32).
#include <stdlib.h>
// outputs of this kernel are the perceptual trunk's full-resolution tensors (34-270 MB per launch): streamed stores (`nt`) in the
// epilogue - batch 32: 8.25 -> 8.19 ms, every class of this file +1 %; for the other kernel families the hint is neutral (section 11)
#ifndef VP_P3_NO_NT
#define VP_NT_STORE 1
#endif
// VP_P3_ABL: build-time ablations of the loop for timing only (results are wrong): 1 no barrier, 2 no vmcnt wait, 4 no DMA in the
// loop, 8 no epilogue, 16 no loop (make ablate3 ABL=n -> ../libvp_p3abl<n>.so, select with VP_LIB)
#ifndef VP_P3_ABL
#define VP_P3_ABL 0
#endif

#include <utility>

#include "conv_ops.h"
#include "igemm_device.h"
#include "launch.h"
#include "patch_device.h"

namespace vp {

template <int TW>
struct Patch3TilePix {
  const IgemmArgs& a; int n, y0, x0;
  __device__ __forceinline__ long long operator()(int row) const {
    constexpr int BPR = TW / 16;
    const int pb = row >> 4, i = row & 15;
    const int y = y0 + pb / BPR, x = x0 + (pb % BPR) * 16 + i;
    if (y >= a.Hg || x >= a.Wg) return -1;
    const long long off = (((long long)n * a.Hof + y) * a.Wof + x) * a.ldY;
    return (off << 8) | (long long)(n / a.ref_group_n);
  }
  static constexpr bool HAS_POOL = (TW == 16);
  __device__ __forceinline__ long long pool(int pr, int pc) const {
    const int y = (y0 >> 1) + pr, x = (x0 >> 1) + pc;
    if (y >= (a.Hg >> 1) || x >= (a.Wg >> 1)) return -1;
    return (((long long)n * (a.Hg >> 1) + y) * (a.Wg >> 1) + x) * a.ldY;
  }
};

// NSTW: stages of the weight ring = NSTW - 1 steps of weights in flight.  A step of a 64-row tile is 8 MFMAs per wave, far shorter than
// the 1-1.5 us an LDS-DMA takes to land under load: with 3 stages the loop ran at the DMA latency (0.7 us per step whatever the tile),
// with 6 the weights of five steps are in flight.  18 steps per trip: NSTW divides 18, stage indices stay compile-time.
// KW: 3 (3x3 taps) or 4 (4x4 taps: the discriminator's layer_4 backward-data passes, round 5 - 32 steps per trip, four ring stages)
template <typename T, int WC, int WP, int TC, int TP, int TH, int TW, int STATS, int OCC, int NSTW = 3, int KW = 3>
__global__ __launch_bounds__(512, OCC) void igemm_patch3_kernel(const IgemmArgs a) {
  constexpr int E = Elem<T>::E, KC = 4 * E;
  constexpr int NW = 8, NT = 512, LA = NSTW - 1;
  constexpr int KS2 = KW * KW, TRIP = 2 * KS2;                       // patch positions of a chunk, steps of a trip (two chunks)
  static_assert(TRIP % NSTW == 0 && NSTW >= 3, "ring stages");
  static_assert(WC * WP == NW, "eight waves");
  constexpr int BC = WC * TC * 16, BP = TH * TW;
  static_assert(BP == WP * TP * 16, "pixel blocks of the tile = pixel blocks of the waves");
  constexpr int NBA = BC / 16;
  static_assert(NBA % NW == 0 || NBA == 4, "weight DMAs: whole instructions per wave (64-row tiles: half an instruction per wave)");
  constexpr int JA = (NBA + NW - 1) / NW;
  constexpr bool HALFW = NBA < NW;
  constexpr int PW = TW + KW - 1, PH = TH + KW - 1, NPATCH = PW * PH;
  constexpr int PPAD = (NPATCH + 127) / 128 * 128;                   // patch pixels, padded to whole DMA rounds of the 8 waves
  constexpr int JP = PPAD / 128;                                     // patch DMA instructions per wave and chunk
  static_assert(JP + LA <= KS2, "one patch DMA per tap step, none in the last LA steps of a chunk");
  constexpr int PBUFB = PPAD * 64;                                   // bytes of one patch buffer
  constexpr int WSTB = 4 * BC * 16;                                  // bytes of one weight stage
  constexpr int WBASE = 2 * PBUFB;
  static_assert(PBUFB + (KW - 1) * PW * 64 + 64 < 65536 && (NSTW - 1) * WSTB + 7 * 1024 + 16 < 65536, "read offsets are DS immediates");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c_base = blockIdx.y * BC;
  const int tiles_x = (a.Wg + TW - 1) / TW, tiles_y = (a.Hg + TH - 1) / TH;
  // XCD-aware tile order (a.xcd_remap): each XCD a contiguous run of tiles, so that halo pixels meet in one L2.  No change for the
  // kernel alone (the shared infinity cache already serves the halos); it takes L2-miss traffic off the fabric the co-running
  // streams share
  int bt = blockIdx.x;
  if (a.xcd_remap && (gridDim.x & 7) == 0) bt = (bt & 7) * (gridDim.x >> 3) + (bt >> 3);
  const int n = bt / (tiles_x * tiles_y);
  const int trem = bt - n * (tiles_x * tiles_y);
  const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
  // tap t = 3r + c reads input pixel (y + p_dhf + r * p_dhs, x + p_dwf + c * p_dws); both steps are +1 (forward) or both -1
  // (backward-data: the flipped kernel).  The loop walks PATCH positions (pr, pc) in a fixed order; the weight chunk that belongs
  // to patch position u is tap u (forward) or tap 8 - u (flipped)
  const bool flip = a.p_dhs < 0;
  const int dh0 = flip ? a.p_dhf - (KW - 1) : a.p_dhf, dw0 = flip ? a.p_dwf - (KW - 1) : a.p_dwf;
  const unsigned es = sizeof(T);
  const int C0 = a.x.C[0];
  const int nchunkc = C0 / KC;                  // channel chunks (even)

  __amdgpu_buffer_rsrc_t rsW = make_rsrc(reinterpret_cast<const T*>(a.Wp), 0xFFFFFFFFu);
  __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * C0 * es));

  // patch DMA lanes: instruction j of this wave covers patch pixels (wave + 8j) * 16 .. + 15, lane -> (pixel, slot)
  unsigned pvo[JP];
#pragma unroll
  for (int j = 0; j < JP; ++j) {
    const int pp = (wave + NW * j) * 16 + (lane >> 2);
    const int py = pp / PW, px = pp - py * PW;
    const int ih = y0 + dh0 + py, iw = x0 + dw0 + px;
    const bool ok = pp < NPATCH && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
    const int piece = (lane & 3) ^ ((px >> 2) & 3);
    pvo[j] = ok ? (unsigned)((((n * a.Hin + ih) * a.Win + iw) * C0 + piece * E) * es) : DMA_OOB;
  }
  // weight DMA lanes (rb_swz image); 64-row tiles: every wave moves half a 16-row block (lanes 0-31)
  unsigned wvo[JA];
  {
    const int r = HALFW ? (wave & 1) * 8 + (lane >> 2) : lane >> 2;
    const int g = (lane & 3) ^ rb_swz(r & 15);
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const int blk = HALFW ? (wave >> 1) : wave + NW * j;
      wvo[j] = (unsigned)(((c_base + blk * 16 + r) * KC + g * E) * es);
    }
  }
  const unsigned wstep = (unsigned)(a.wp_rows * KC * es);
  // weights of patch position u (tap u or 8 - u) and channel chunk c -> ring stage `stage`
  auto issue_w = [&](int u, int chunk, int stage) {
    const int tap = flip ? KS2 - 1 - u : u;
    const unsigned wso = (unsigned)(tap * nchunkc + chunk) * wstep;
    uint4* la = reinterpret_cast<uint4*>(smem + WBASE + stage * WSTB);
    if constexpr (HALFW) {
      if (lane < 32) dma16_buf(rsW, wvo[0], wso, la + (wave >> 1) * 64 + (wave & 1) * 32);
    } else {
#pragma unroll
      for (int j = 0; j < JA; ++j) dma16_buf(rsW, wvo[j], wso, la + (wave + NW * j) * 64);
    }
  };
  auto issue_p = [&](int chunk, int buf, int j) {
    uint4* lb = reinterpret_cast<uint4*>(smem + buf * PBUFB);
    dma16_buf(rsX, pvo[j], (unsigned)(chunk * KC) * es, lb + (wave + NW * j) * 64);
  };

  const int wc = wave / WP, wpi = wave - wc * WP;
  const int blkA0 = wc * TC, blkB0 = wpi * TP;
  const int fi = lane & 15, fg = lane >> 4;
  const int aaddr = WBASE + (blkA0 * 64 + fi * 4 + (fg ^ rb_swz(fi))) * 16;     // (ring base folded in: stage and block offsets fit the immediate)
  // B fragment lane offsets per column shift c: patch pixel (row of the pixel block, its first column + lane + c)
  int tb0[KW][TP];
#pragma unroll
  for (int c = 0; c < KW; ++c)
#pragma unroll
    for (int q = 0; q < TP; ++q) {
      constexpr int BPR = TW / 16;
      const int pb = blkB0 + q;
      const int px = (pb % BPR) * 16 + fi + c;
      const int pp = (pb / BPR) * PW + px;
      tb0[c][q] = (pp << 6) + (((fg ^ (px >> 2)) & 3) << 4) + ((fg & 1) << 3);
    }

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: the whole first patch, then the weights of the first LA steps
#pragma unroll
  for (int j = 0; j < JP; ++j) issue_p(0, 0, j);
#pragma unroll
  for (int d = 0; d < LA; ++d) issue_w(d % KS2, d / KS2, d % NSTW);

  // One trip = 2 chunks x 9 patch positions.  Step U = 9 * cc + u (cc: chunk of the pair = patch buffer, u: patch position);
  // ring stage U % NSTW.  In DMA order behind the weights of step U (issued at step U - LA): the patch piece of step U - LA, then per
  // later step its weights and its patch piece - a piece is issued at the steps with u < JP, always (behind the last chunk it
  // fetches bytes nobody reads, into the idle buffer)
  for (int c = 0; c < ((VP_P3_ABL & 16) ? 0 : nchunkc); c += 2) {
    const bool last_pair = c + 2 >= nchunkc;
    auto step = [&](auto uc) {
      constexpr int U = decltype(uc)::value;
      constexpr int cc = U / KS2, u = U % KS2, pr = u / KW, pc = u % KW, stage = U % NSTW;
      // DMAs issued behind the weights of this step: the patch pieces of the LA previous steps, the weights of the LA - 1 next ones
      constexpr int NPIECE = [] { int n = 0; for (int k = 1; k <= LA; ++k) n += ((U + TRIP - k) % TRIP % KS2) < JP ? 1 : 0; return n; }();
      constexpr int NV = (LA - 1) * JA + NPIECE;
      constexpr int WLAST = (TRIP - 1 - U < LA - 1 ? TRIP - 1 - U : LA - 1);                 // behind the last pair no weights of a next trip follow
#if !(VP_P3_ABL & 2)
      if (U + LA - 1 > TRIP - 1 && last_pair) wait_vm<WLAST * JA + NPIECE>();
      else wait_vm<NV>();
#endif
#if !(VP_P3_ABL & 1)
      __builtin_amdgcn_s_barrier();
#endif
      asm volatile("" ::: "memory");
      // weights LA steps ahead (ring stage (U + LA) % NSTW, last read in step U - 1)
#if !(VP_P3_ABL & 4)
      {
        constexpr int UL = (U + LA) % TRIP;
        if (U + LA < TRIP) issue_w(UL % KS2, c + UL / KS2, (U + LA) % NSTW);
        else if (!last_pair) issue_w(UL % KS2, c + 2 + UL / KS2, (U + LA) % NSTW);
      }
      if constexpr (u < JP) issue_p(c + cc + 1, 1 - cc, u);
#endif
      // eight weight blocks per wave: two read batches of four (16 fragment registers instead of 32); the 64-accumulator tile at
      // two blocks per CU (128 registers per lane): batches of two
      constexpr int NA = TC == 8 ? 4 : ((TC == 4 && OCC == 4) ? 2 : TC);
      patch_step_mma<T, TC, TP, NA, stage * WSTB, cc * PBUFB + pr * PW * 64, true>(aaddr, tb0[pc], acc);
    };
    static_steps(step, std::make_integer_sequence<int, TRIP>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing patch pieces: the epilogue reuses the LDS

  constexpr int RINGB = NSTW * WSTB + 2 * PBUFB;
  constexpr int NPASS = epi_passes(BC, BP, WP, RINGB);
#ifndef VP_P3_NO_FASTEPI
  constexpr int NPASS16 = BC == 64 ? epi_passes16(BC, BP, WP, RINGB) : 0;    // 64-row tiles: +4 %; 128 rows: -8 %, 256 rows: +-0
#else
  constexpr int NPASS16 = 0;
#endif
  if (!(VP_P3_ABL & 8) || acc[0][0][0] == 1.2345f) staged_epilogue<T, TC, TP, BC, BP, NPASS, NT, STATS, NPASS16>(a, Patch3TilePix<TW>{a, n, y0, x0}, c_base, blkA0, blkB0, acc, smem, bt, 0);
}

template <typename T, int WC, int WP, int TC, int TP, int TH, int TW, int OCC, int NSTW = 3, int KW = 3>
static hipError_t launch_patch3_t(const IgemmArgs& b, hipStream_t st) {
  constexpr int BC = WC * TC * 16, BP = TH * TW;
  constexpr int PPAD = ((TH + KW - 1) * (TW + KW - 1) + 127) / 128 * 128;
  constexpr int RINGB = NSTW * 4 * BC * 16 + 2 * PPAD * 64;
  constexpr int NPE = epi_passes(BC, BP, WP, RINGB);
  size_t sm = RINGB;
  const size_t se = (size_t)(BP / NPE) * (BC * 4 + 16) + (BP / NPE) * 8;
  if (se > sm) sm = se;
  const int tiles = b.N * ((b.Hg + TH - 1) / TH) * ((b.Wg + TW - 1) / TW);
  dim3 grid(tiles, b.CoutPad / BC, 1);
  auto kern = b.bn_part ? igemm_patch3_kernel<T, WC, WP, TC, TP, TH, TW, 1, OCC, NSTW, KW> : igemm_patch3_kernel<T, WC, WP, TC, TP, TH, TW, 0, OCC, NSTW, KW>;
  if (b.bst_y) {       // backward sums of a batch-normalised tensor (the discriminator's layer_3 under layer_4's backward-data): the 4x4 form only
    if constexpr (KW == 4) kern = igemm_patch3_kernel<T, WC, WP, TC, TP, TH, TW, 2, OCC, NSTW, KW>;
    else return hipErrorInvalidValue;
  }
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(kern, grid, dim3(512), sm, st, b);
  return hipGetLastError();
}

// 3x3 taps on a +1 / +1 or -1 / -1 grid, an even number of 64-byte channel chunks: what the unrolled kernel handles
bool patch3_eligible(const IgemmArgs& a, int is_bf16) {
  const int kc = is_bf16 ? 32 : 16;
  return a.patch && a.ntaps == 9 && a.p_kw == 3 && a.p_dhs == a.p_dws && (a.p_dhs == 1 || a.p_dhs == -1) && a.x.C[0] % (2 * kc) == 0 &&
         a.x.C[1] == 0;
}

// 4x4 taps (stride 1) on the same kernel (the discriminator's layer_4: forward with its batch statistics, backward-data): bf16, channel
// rows a multiple of 128: the 128-row x 16 x 16-pixel tile (80 KB of LDS with four ring stages: two blocks per CU; the 256-row tile would
// need 96 KB) - plan_make_patch names that tile for 4x4 layers
bool patch4_eligible(const IgemmArgs& a, int is_bf16) {
  return patch4_knob() && is_bf16 && a.patch == 1 && a.ntaps == 16 && a.p_kw == 4 && a.p_dhs == a.p_dws && (a.p_dhs == 1 || a.p_dhs == -1) &&
         a.x.C[0] % 64 == 0 && a.x.C[1] == 0 && a.CoutPad % 128 == 0 && !a.pool_out;
}
hipError_t launch_igemm_patch4(const IgemmArgs& a, hipStream_t st) {
  IgemmArgs b = a;
  b.vec_epi = 1;
  b.xcd_remap = patch_xcd_knob();
  return launch_patch3_t<bf16, 2, 4, 4, 4, 16, 16, 4, 4, 4>(b, st);
}

// same tile menu as launch_igemm_patch (conv_patch.hip)
hipError_t launch_igemm_patch3(const IgemmArgs& a, int is_bf16, int bc, int bp, hipStream_t st) {
  IgemmArgs b = a;
  b.vec_epi = 1;
  b.xcd_remap = patch_xcd_knob();
#define VP_PATCH3_GO(WC, WP, TC, TP, TH, TW, OCC) \
  (is_bf16 ? launch_patch3_t<bf16, WC, WP, TC, TP, TH, TW, OCC>(b, st) : launch_patch3_t<float, WC, WP, TC, TP, TH, TW, OCC>(b, st))
#define VP_PATCH3_GO6(WC, WP, TC, TP, TH, TW, OCC) \
  (is_bf16 ? launch_patch3_t<bf16, WC, WP, TC, TP, TH, TW, OCC, 6>(b, st) : launch_patch3_t<float, WC, WP, TC, TP, TH, TW, OCC, 6>(b, st))
  // 64-row tiles of 16 x 16 pixels: six weight stages (73 KB of LDS, still two blocks per CU; the three-stage form: EXPERIMENTS.md 6 (5))
  if (bc == 256) return bp == 128 ? VP_PATCH3_GO(2, 4, 8, 2, 8, 16, 4) : VP_PATCH3_GO(2, 4, 8, 4, 16, 16, 2);
  if (bc == 128) return bp == 512 ? VP_PATCH3_GO(1, 8, 8, 4, 16, 32, 2) : VP_PATCH3_GO(2, 4, 4, 4, 16, 16, 4);
  if (bp == 512) return VP_PATCH3_GO(1, 8, 4, 4, 16, 32, 2);
  return VP_PATCH3_GO6(2, 4, 2, 4, 16, 16, 4);
#undef VP_PATCH3_GO6
#undef VP_PATCH3_GO
}

}  // namespace vp
