// igemm_patch2_kernel: the four parity classes of a 4x4 / stride-2 TRANSPOSED convolution - the generator's deconv forward
// (pixrefer.py:71-86) and the backward-data of every 4x4 stride-2 conv - in the unrolled patch form of conv_patch3.hip.
//
// A class (ph, pw) is a 2x2-tap stride-1 convolution over the input: output pixel (2y + ph, 2x + pw) sums taps (ta, tb) in {0,1}^2 of
// input pixel (y + ph - ta, x + pw - tb) (conv_ops.h plan_fwd / plan_bwd_data).  The gather-per-tap kernels DMA every input pixel
// four times per class and spend 3-7 vector-ALU instructions per MFMA on it (profiles/r02_pmc_instruction_mix.txt).  Here a block
// owns a TH x TW tile of (y, x) of one image and one class; per 64-byte channel chunk the (TH+1) x (TW+1) input patch is DMA'd into
// LDS once and the four taps read their B fragments from shifted positions (column-only swizzle: row shifts, patch buffer and ring
// stage are DS immediates, see conv_patch3.hip); weights stream per (tap, chunk) through a 4-stage ring (8 steps per trip = 2 chunks
// x 4 patch positions; 8 % 4 == 0 keeps the stage a compile-time constant).  Patch position u = (pr, pc) holds tap 3 - u for both
// callers (dh = ph - ta, dw = pw - tb: the patch origin is tap 3).  Two source tensors (the decoder's virtual concat) are walked
// chunk by chunk: first all chunks of source 0, then source 1; channel counts are multiples of 64 (bf16) / 32 (f32) per source.
#include <stdlib.h>

#include "conv_ops.h"
#include "igemm_device.h"
#include "launch.h"
#include "patch_device.h"

namespace vp {

// tile row -> output element offset: grid pixel (y, x) of class (o0h, o0w) lands on output pixel (y * os + o0h, x * os + o0w)
template <int TW>
struct Patch2TilePix {
  const IgemmArgs& a; int n, y0, x0, oh, ow;
  __device__ __forceinline__ long long operator()(int row) const {
    constexpr int BPR = TW / 16;
    const int pb = row >> 4, i = row & 15;
    const int y = y0 + pb / BPR, x = x0 + (pb % BPR) * 16 + i;
    if (y >= a.Hg || x >= a.Wg) return -1;
    const long long off = (((long long)n * a.Hof + (y * a.os + oh)) * a.Wof + (x * a.os + ow)) * a.ldY;
    return (off << 8) | (long long)(n / a.ref_group_n);
  }
};

// TWOSRC: the pixel operand is a virtual concat of two tensors (decoder layers); single-source launches carry one set of lane offsets
template <typename T, int WC, int WP, int TC, int TP, int TH, int TW, int STATS, int OCC, bool TWOSRC>
__global__ __launch_bounds__(512, OCC) void igemm_patch2_kernel(const IgemmArgs a) {
  constexpr int E = Elem<T>::E, KC = 4 * E;
  constexpr int NW = 8, NT = 512, NSTW = 4, NSTEP = 8;
  static_assert(WC * WP == NW, "eight waves");
  constexpr int BC = WC * TC * 16, BP = TH * TW;
  static_assert(BP == WP * TP * 16, "pixel blocks of the tile = pixel blocks of the waves");
  constexpr int NBA = BC / 16;
  static_assert(NBA % NW == 0 || NBA == 4, "weight DMAs: whole instructions per wave (64-row tiles: half an instruction per wave)");
  constexpr int JA = (NBA + NW - 1) / NW;
  constexpr bool HALFW = NBA < NW;
  constexpr int PW = TW + 1, PH = TH + 1, NPATCH = PW * PH;
  constexpr int PPAD = (NPATCH + 127) / 128 * 128;
  constexpr int JP = PPAD / 128;
  static_assert(JP <= 4, "one patch DMA per tap step");
  constexpr int PBUFB = PPAD * 64;
  constexpr int WSTB = 4 * BC * 16;
  constexpr int WBASE = 2 * PBUFB;
  static_assert(PBUFB + PW * 64 + 64 < 65536 && 3 * WSTB + 7 * 1024 + 16 < 65536, "read offsets are DS immediates");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c_base = blockIdx.y * BC;
  const int cls = blockIdx.z;
  const int tiles_x = (a.Wg + TW - 1) / TW, tiles_y = (a.Hg + TH - 1) / TH;
  const int bt = blockIdx.x;
  const int n = bt / (tiles_x * tiles_y);
  const int trem = bt - n * (tiles_x * tiles_y);
  const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
  const int dh0 = a.taps[cls].dh[3], dw0 = a.taps[cls].dw[3];        // patch origin = tap 3 (ta = tb = 1)
  const unsigned es = sizeof(T);
  const int C0 = a.x.C[0], C1 = TWOSRC ? a.x.C[1] : 0;
  const int n0 = C0 / KC, nchunkc = (C0 + C1) / KC;                  // chunks of source 0 / of both (even each)

  __amdgpu_buffer_rsrc_t rsW = make_rsrc(reinterpret_cast<const T*>(a.Wp) + (size_t)cls * a.wp_rows * a.Kpad, 0xFFFFFFFFu);
  __amdgpu_buffer_rsrc_t rsX0 = make_rsrc(a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * C0 * es));
  __amdgpu_buffer_rsrc_t rsX1 = make_rsrc(a.x.ptr[1] ? a.x.ptr[1] : a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * C1 * es));

  // patch DMA lanes: instruction j of this wave covers patch pixels (wave + 8j) * 16 .. + 15, lane -> (pixel, slot); one lane offset
  // per source (the pixel stride differs)
  unsigned pvo0[JP], pvo1[TWOSRC ? JP : 1];
#pragma unroll
  for (int j = 0; j < JP; ++j) {
    const int pp = (wave + NW * j) * 16 + (lane >> 2);
    const int py = pp / PW, px = pp - py * PW;
    const int ih = y0 + dh0 + py, iw = x0 + dw0 + px;
    const bool ok = pp < NPATCH && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
    const int piece = (lane & 3) ^ ((px >> 2) & 3);
    const int pix = (n * a.Hin + ih) * a.Win + iw;
    pvo0[j] = ok ? (unsigned)((pix * C0 + piece * E) * es) : DMA_OOB;
    if (TWOSRC) pvo1[TWOSRC ? j : 0] = ok ? (unsigned)((pix * C1 + piece * E) * es) : DMA_OOB;
  }
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
  auto issue_w = [&](int u, int chunk, int stage) {
    const unsigned wso = (unsigned)((3 - u) * nchunkc + chunk) * wstep;       // patch position u holds tap 3 - u
    uint4* la = reinterpret_cast<uint4*>(smem + WBASE + stage * WSTB);
    if constexpr (HALFW) {
      if (lane < 32) dma16_buf(rsW, wvo[0], wso, la + (wave >> 1) * 64 + (wave & 1) * 32);
    } else {
#pragma unroll
      for (int j = 0; j < JA; ++j) dma16_buf(rsW, wvo[j], wso, la + (wave + NW * j) * 64);
    }
  };
  auto issue_p = [&](int chunk, int buf, int j) {
    uint4* lb = reinterpret_cast<uint4*>(smem + buf * PBUFB) + (wave + NW * j) * 64;
    if (!TWOSRC || chunk < n0) dma16_buf(rsX0, pvo0[j], (unsigned)(chunk * KC) * es, lb);    // (behind the last chunk: bytes nobody reads)
    else dma16_buf(rsX1, pvo1[TWOSRC ? j : 0], (unsigned)((chunk - n0) * KC) * es, lb);
  };

  const int wc = wave / WP, wpi = wave - wc * WP;
  const int blkA0 = wc * TC, blkB0 = wpi * TP;
  const int fi = lane & 15, fg = lane >> 4;
  const int aaddr = WBASE + (blkA0 * 64 + fi * 4 + (fg ^ rb_swz(fi))) * 16;
  int tb0[2][TP];
#pragma unroll
  for (int c = 0; c < 2; ++c)
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

  // prologue: the whole first patch, then the weights of the first NSTW - 1 steps
#pragma unroll
  for (int j = 0; j < JP; ++j) issue_p(0, 0, j);
#pragma unroll
  for (int d = 0; d < NSTW - 1; ++d) issue_w(d, 0, d);

  // One trip = 2 chunks x 4 patch positions.  Step U = 4 * cc + u; ring stage U % 4.  At step V the weights of step V + 3 are issued,
  // then (u(V) < 2) pieces of the next chunk's patch - always (see conv_patch3.hip).
  for (int c = 0; c < nchunkc; c += 2) {
    const bool last_pair = c + 2 >= nchunkc;
    auto step = [&](auto uc) {
      constexpr int U = decltype(uc)::value;
      constexpr int cc = U / 4, u = U % 4, pr = u / 2, pc = u % 2, stage = U % NSTW;
      // the JP pieces of the next chunk's patch are issued at the chunk's first two steps (PC0 at u = 0, the rest at u = 1), so that
      // none is younger than the weights a later chunk start has to wait for anyway.  In DMA order behind the weights of step U
      // (issued at step U - 3, ahead of that step's pieces): weights of U + 1 and U + 2 and the pieces of the steps U - 3 .. U - 1;
      // the first step of a chunk (u = 0) needs the pieces of U - 3 (its own patch) landed as well
      constexpr int PC0 = JP - JP / 2, PC1 = JP / 2;
      constexpr int NV = (NSTW - 2) * JA + (u == 0 ? 0 : u == 1 ? PC0 : PC0 + PC1);
      constexpr int NLAST = NV - (U + 1 >= NSTEP ? JA : 0) - (U + 2 >= NSTEP ? JA : 0);     // last trip: no weights of steps >= 8
      if (U >= NSTEP - 2 && last_pair) wait_vm<NLAST>();
      else wait_vm<NV>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      {
        constexpr int U3 = (U + NSTW - 1) % NSTEP;
        if (U + NSTW - 1 < NSTEP) issue_w(U3 % 4, c + U3 / 4, (U + NSTW - 1) % NSTW);
        else if (!last_pair) issue_w(U3 % 4, c + 2 + U3 / 4, (U + NSTW - 1) % NSTW);
      }
      if constexpr (u == 0) {
#pragma unroll
        for (int j = 0; j < PC0; ++j) issue_p(c + cc + 1, 1 - cc, j);
      } else if constexpr (u == 1) {
#pragma unroll
        for (int j = PC0; j < JP; ++j) issue_p(c + cc + 1, 1 - cc, j);
      }
      constexpr int NA = TC == 8 ? 4 : ((TC == 4 && OCC == 4) ? 2 : TC);
      patch_step_mma<T, TC, TP, NA, stage * WSTB, cc * PBUFB + pr * PW * 64, false>(aaddr, tb0[pc], acc);
    };
    static_steps(step, std::make_integer_sequence<int, NSTEP>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing patch pieces: the epilogue reuses the LDS

  constexpr int RINGB = NSTW * WSTB + 2 * PBUFB;
  constexpr int NPASS = epi_passes(BC, BP, WP, RINGB);
  staged_epilogue<T, TC, TP, BC, BP, NPASS, NT, STATS>(a, Patch2TilePix<TW>{a, n, y0, x0, a.o0h[cls], a.o0w[cls]}, c_base, blkA0, blkB0, acc,
                                                       smem, bt, cls);
}

template <typename T, int WC, int WP, int TC, int TP, int TH, int TW, int OCC>
static hipError_t launch_patch2_t(const IgemmArgs& b, hipStream_t st) {
  constexpr int BC = WC * TC * 16, BP = TH * TW;
  constexpr int PPAD = ((TH + 1) * (TW + 1) + 127) / 128 * 128;
  constexpr int RINGB = 4 * 4 * BC * 16 + 2 * PPAD * 64;
  constexpr int NPE = epi_passes(BC, BP, WP, RINGB);
  size_t sm = RINGB;
  const size_t se = (size_t)(BP / NPE) * (BC * 4 + 16) + (BP / NPE) * 8;
  if (se > sm) sm = se;
  const int tiles = b.N * ((b.Hg + TH - 1) / TH) * ((b.Wg + TW - 1) / TW);
  dim3 grid(tiles, b.CoutPad / BC, b.nclass);
  const bool two = b.x.C[1] > 0;
  const bool bst = b.bst_y != nullptr;                       // backward sums of the tensor this launch completes the gradient of (single-source launches: backward-data)
  if (bst && two) return hipErrorInvalidValue;
  auto kern = bst ? igemm_patch2_kernel<T, WC, WP, TC, TP, TH, TW, 2, OCC, false>
            : b.bn_part ? (two ? igemm_patch2_kernel<T, WC, WP, TC, TP, TH, TW, 1, OCC, true> : igemm_patch2_kernel<T, WC, WP, TC, TP, TH, TW, 1, OCC, false>)
                        : (two ? igemm_patch2_kernel<T, WC, WP, TC, TP, TH, TW, 0, OCC, true> : igemm_patch2_kernel<T, WC, WP, TC, TP, TH, TW, 0, OCC, false>);
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(kern, grid, dim3(512), sm, st, b);
  return hipGetLastError();
}

// 128- or 64-row tiles of 16 x 16 grid pixels, two blocks per CU
hipError_t launch_igemm_patch2(const IgemmArgs& a, int is_bf16, int bc, int bp, hipStream_t st) {
  IgemmArgs b = a;
  b.vec_epi = 1;
  if (bp != 256) return hipErrorInvalidValue;
#define VP_PATCH2_GO(WC, WP, TC, TP, TH, TW, OCC) \
  (is_bf16 ? launch_patch2_t<bf16, WC, WP, TC, TP, TH, TW, OCC>(b, st) : launch_patch2_t<float, WC, WP, TC, TP, TH, TW, OCC>(b, st))
  if (bc == 128) return VP_PATCH2_GO(2, 4, 4, 4, 16, 16, 4);
  if (bc == 64) return VP_PATCH2_GO(2, 4, 2, 4, 16, 16, 4);
#undef VP_PATCH2_GO
  return hipErrorInvalidValue;
}

}  // namespace vp
