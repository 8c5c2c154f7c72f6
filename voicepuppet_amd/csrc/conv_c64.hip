// conv_c64_kernel: the 3x3 / stride-1 convolution from 64 to 64 channels (VGG conv1_2 forward and its backward-data pass,
// vgg_simple.py:138-141) with the WEIGHTS RESIDENT IN REGISTERS.
//
// Why a kernel of its own (round 5): on the unrolled patch kernel (conv_patch3.hip, 64 x 256 tile) this layer was the largest launch
// of the step at 0.27 of the MFMA peak.  Its weight matrix is tiny (64 x 576 bf16 = 72 KB) but every 256-pixel tile re-fetched all of
// it from L2 into LDS: 72 KB of weights + 41 KB of input patch per 9.4 MMAC, i.e. the kernel ran at the L2 -> LDS fill rate with the
// weights as the larger part of the fill.  Here
//   * a block is TWO waves; wave h owns output channels [32h, 32h + 32) and keeps its 32 x 576 slice of the weights as 36 MFMA A
//     fragments in 144 registers for the life of the block (persistent: a block walks many pixel tiles, the weights are fetched once);
//   * only the input patch of a tile ((4 + 2) x (16 + 2) pixels x 64 channels = 13.5 KB) goes through LDS, by LDS-DMA, DOUBLE-BUFFERED
//     (the next tile's patch is in flight under this tile's MFMAs), and both waves read their B fragments from it (same column-only
//     row permutation as conv_patch3.hip, so the packed weights of the patch plan are read as they are) with one conflict-free ds_read_b128 per fragment;
//   * no LDS for the weights and a two-wave barrier: four blocks per CU (two waves per SIMD, 256 registers each) that drift out of phase,
//     so one block's epilogue sits under another block's MFMAs;
//   * the epilogue leaves the accumulators directly: with the row permutation of the packed weights a lane ends with 8 consecutive
//     channels of one pixel (16 bytes); bias (the accumulators start from it) + relu (forward), relu'(reference) (backward-data) and the
//     fused 2x2 max pool (rows q, q + 1 of the pair a wave has just finished, columns by one DPP exchange) happen in registers.
// bf16 only (the float32 parity path stays on conv_patch3.hip); image sides multiples of 4 x 16.
#include "conv_ops.h"
#include "igemm_device.h"
#include "launch.h"
#include "patch_device.h"

#ifndef C64_LA
#define C64_LA 2
#endif
// C64_ABL: build-time ablations for timing only (results are wrong): 1 = the patch is fetched for a block's first tiles only, 2 = no stores,
// 4 = no MFMAs, 8 = B fragments read for the first steps of a row pair only (make one FILE=conv_c64 VAR=xabl1 DEFS=-DC64_ABL=1, VP_LIB)
#ifndef C64_ABL
#define C64_ABL 0
#endif

namespace vp {

namespace {
constexpr int TH = 4, TW = 16, PW = TW + 2, PH = TH + 2;
constexpr int NPATCH = PW * PH;               // 108 patch pixels
constexpr int PPAD = 128;                     // ... padded to whole DMA rounds of the two waves (2 x 16 pixels)
constexpr int JP = PPAD / 32;                 // patch DMA instructions per wave and chunk
constexpr int PBUFB = PPAD * 64;              // bytes of one channel chunk (32 channels) of a patch
constexpr int BUFB = 2 * PBUFB;               // one patch buffer (both chunks); two buffers per block
}  // namespace

__device__ __forceinline__ float max_dpp_xor1(float v) {    // max(v, v of lane ^ 1): quad_perm [1, 0, 3, 2]
  float r;
  asm("v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }   // one instruction (fmaxf canonicalises first)

template <bool REF>
__global__ __launch_bounds__(128, 2) void conv_c64_kernel(const IgemmArgs a, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fg = lane >> 4;
  // tap t = 3r + c reads input pixel (y + p_dhf + r * p_dhs, x + p_dwf + c * p_dws); both steps +1 (forward) or both -1 (backward-data:
  // the flipped kernel).  The loop walks PATCH positions u = 3 pr + pc; the weights of position u are tap u (forward) or 8 - u (flipped)
  const bool flip = a.p_dhs < 0;
  const int dh0 = flip ? a.p_dhf - 2 : a.p_dhf, dw0 = flip ? a.p_dwf - 2 : a.p_dwf;

  // this wave's weights: MFMA tiles 2 * wave, 2 * wave + 1 of the 64-row block (channels 32 * wave + 8 q + 4 t + e at row 4 q + e of tile
  // t: IgemmArgs::rowperm), one 16-byte fragment per (patch position, chunk, tile): piece fg of packed row fi
  uint4 W[18][2];
  {
    const bf16* wp = reinterpret_cast<const bf16*>(a.Wp);
#pragma unroll
    for (int U = 0; U < 18; ++U) {
      const int cc = U / 9, u = U % 9;
      const int tap = flip ? 8 - u : u;
      const size_t chunk_row0 = (size_t)(tap * 2 + cc) * a.wp_rows;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint4 w = *reinterpret_cast<const uint4*>(wp + (chunk_row0 + (2 * wave + t) * 16 + fi) * 32 + fg * 8);
        // the patch plan packs odd 16-byte k pieces with their halves swapped (PackDesc::kswap, for conv_patch3.hip's 8-byte fragment
        // reads); this kernel reads whole pieces: natural order
        W[U][t] = (fg & 1) ? make_uint4(w.z, w.w, w.x, w.y) : w;
      }
    }
  }
  const int c0 = 32 * wave + 8 * fg;            // the 8 consecutive output channels this lane finishes for pixel fi of a 16-pixel row
  float bia[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bia[e] = a.bias ? a.bias[c0 + e] : 0.f;

  // B fragment lane offsets per column shift pc.  A patch pixel is 64 bytes per chunk; piece p of patch column px lies in slot
  // p ^ ((px >> 1) & 3): with that swizzle the 16 lanes of every ds_read_b128 service group ({0-3, 12-15, 20-27}, ...) hit 16 distinct
  // 16-byte slots modulo 256 bytes for all three column shifts (checked exhaustively; (px >> 2) & 3, the 8-byte-read swizzle of
  // conv_patch3.hip, gives 2-way conflicts here), and a tap's row shift stays a plain byte offset
  int tb0[3];
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) {
    const int px = fi + pc;
    tb0[pc] = (px << 6) + (((fg ^ (px >> 1)) & 3) << 4);
  }
  // patch DMA lanes: instruction j of this wave covers patch pixels (wave + 2j) * 16 .. + 15, lane -> (pixel, slot)
  int ppy[JP], ppx[JP], prel[JP];
#pragma unroll
  for (int j = 0; j < JP; ++j) {
    const int pp = (wave + 2 * j) * 16 + (lane >> 2);
    ppy[j] = pp < NPATCH ? pp / PW : 1 << 20;            // (rows beyond the patch: never inside the image)
    ppx[j] = pp % PW;
    prel[j] = (((lane & 3) ^ ((ppx[j] >> 1) & 3)) * 8) * (int)sizeof(bf16);
  }

  __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * 64 * sizeof(bf16)));
  const int tiles_x = a.Wg / TW, tpi = tiles_x * (a.Hg / TH);
  const bf16* refp = reinterpret_cast<const bf16*>(a.ref);
  bf16* Yp = reinterpret_cast<bf16*>(a.Y);
  bf16* Pp = reinterpret_cast<bf16*>(a.pool_out);
  const bool store_y = !(a.pool_out != nullptr && a.pool_only);
  const bool relu_out = a.out_act == ACT_RELU;

  // XCD-aware tile order: blocks go round-robin over the 8 XCDs, so each XCD takes a contiguous run of every round's tiles and the
  // halo rows / columns of neighbouring tiles meet in one L2
  const int G = gridDim.x;
  int bt = blockIdx.x;
  if ((G & 7) == 0) bt = (bt & 7) * (G >> 3) + (bt >> 3);

  // the input patch of tile t, both channel chunks, into patch buffer `buf`: 8 LDS-DMAs per wave (out-of-image pixels: the descriptor
  // returns zeros)
  auto issue_patch = [&](int t, int buf) {
    const int n = t / tpi, rem = t - n * tpi;
    const int y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TW;
    const int base = ((n * a.Hin + y0 + dh0) * a.Win + x0 + dw0) * 64 * (int)sizeof(bf16);
#pragma unroll
    for (int j = 0; j < JP; ++j) {
      const int ih = y0 + dh0 + ppy[j], iw = x0 + dw0 + ppx[j];
      const bool ok = (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      const unsigned vo = ok ? (unsigned)(base + (ppy[j] * a.Win + ppx[j]) * 128 + prel[j]) : DMA_OOB;
      uint4* l0 = reinterpret_cast<uint4*>(smem + buf * BUFB) + (wave + 2 * j) * 64;
      dma16_buf(rsX, vo, 0u, l0);
      dma16_buf(rsX, vo, 64u, l0 + PBUFB / 16);
    }
  };

  if (bt < ntiles) issue_patch(bt, 0);
  int it = 0;
  for (int t = bt; t < ntiles; t += G, ++it) {
    const int buf = it & 1;
    // this tile's patch has landed (own DMAs; the barrier covers the other wave's) and both waves are done with the other buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + G < ntiles && (!(C64_ABL & 1) || it < 1)) issue_patch(t + G, buf ^ 1);
    const int n = t / tpi, rem = t - n * tpi;
    const int y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TW;
    int tb[3];
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) tb[pc] = tb0[pc] + buf * BUFB;

    // ---- two pairs of tile rows; per pair 18 steps (2 chunks x 9 patch positions) of 4 MFMAs, B fragments read LA steps ahead ----
    static_steps([&](auto rpi) {
      constexpr int q = 2 * decltype(rpi)::value;
      const size_t off0 = ((size_t)(n * a.Hof + y0 + q) * a.Wof + x0 + fi) * 64 + c0;     // elements; row q + 1: + Wof * 64
      uint4 rz[2];
      if constexpr (REF) {
        rz[0] = *reinterpret_cast<const uint4*>(refp + off0);
        rz[1] = *reinterpret_cast<const uint4*>(refp + off0 + (size_t)a.Wof * 64);
      }
      f32x4 acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){bia[4 * i], bia[4 * i + 1], bia[4 * i + 2], bia[4 * i + 3]};
      // (LA + 1 register sets: a step is 4 MFMAs = 64 cycles, an LDS read under load lands later than that)
      constexpr int LA = C64_LA, NS = LA + 1;
      u32x4 rb[NS][2];                                // [set][row of the pair]
      auto rd = [&](auto uc) {
        constexpr int U = decltype(uc)::value, cc = U / 9, u = U % 9, pr = u / 3, pc = u % 3, s = U % NS;
        if constexpr ((C64_ABL & 8) != 0) { if (U >= NS) return; }
        rb[s][0] = lds_rd128<cc * PBUFB + (q + pr) * PW * 64>(tb[pc]);
        rb[s][1] = lds_rd128<cc * PBUFB + (q + 1 + pr) * PW * 64>(tb[pc]);
      };
      static_steps([&](auto uc) { rd(uc); }, std::make_integer_sequence<int, LA>{});
      static_steps([&](auto uc) {
        constexpr int U = decltype(uc)::value, s = U % NS;
        if constexpr (U + LA < 18) rd(std::integral_constant<int, U + LA>{});
        constexpr int AHEAD = (17 - U < LA ? 17 - U : LA);      // steps whose reads were issued behind this step's
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * AHEAD) : "memory");
        uint4 fb[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          asm volatile("" : "+v"(rb[s][r]));
          fb[r] = make_uint4(rb[s][r].x, rb[s][r].y, rb[s][r].z, rb[s][r].w);
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 2; ++r) { if (!(C64_ABL & 4)) acc[tt][r] = mma16<bf16>(W[U][tt], fb[r], acc[tt][r]); else acc[tt][r][0] += __uint_as_float(fb[r].x ^ W[U][tt].x); }
        __builtin_amdgcn_sched_barrier(0);
      }, std::make_integer_sequence<int, 18>{});

      // ---- epilogue of the two rows: activation, act'(reference), one 16-byte store per lane and row; 2x2 max pool ----
      float v[2][8];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = acc[e >> 2][r][e & 3];
          v[r][e] = relu_out ? relu1(x) : x;
        }
        uint4 pk = Elem<bf16>::pack(v[r]);
        if constexpr (REF) {
          // relu'(reference): a bf16 is positive exactly when its 16 bits, read as a signed integer, are
          auto keep = [](unsigned z) {
            const unsigned lo = (int)(short)(z & 0xffffu) > 0 ? 0xffffu : 0u, hi = (int)z >> 16 > 0 ? 0xffff0000u : 0u;
            return lo | hi;
          };
          pk.x &= keep(rz[r].x); pk.y &= keep(rz[r].y); pk.z &= keep(rz[r].z); pk.w &= keep(rz[r].w);
        }
        if (store_y && (!(C64_ABL & 2) || v[r][0] == 1.2345f)) {
          unsigned* yp = reinterpret_cast<unsigned*>(Yp + off0 + (size_t)r * a.Wof * 64);
          __builtin_nontemporal_store(pk.x, yp); __builtin_nontemporal_store(pk.y, yp + 1);
          __builtin_nontemporal_store(pk.z, yp + 2); __builtin_nontemporal_store(pk.w, yp + 3);
        }
      }
      if (Pp) {
        // rounding is monotonic: the maximum of the f32 values, rounded, equals the maximum of the stored (rounded) values
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = max_dpp_xor1(fmaxf(v[0][e], v[1][e]));
        if (!(fi & 1)) {
          const size_t po = ((size_t)(n * (a.Hg >> 1) + ((y0 + q) >> 1)) * (a.Wg >> 1) + ((x0 + fi) >> 1)) * 64 + c0;
          *reinterpret_cast<uint4*>(Pp + po) = Elem<bf16>::pack(m);
        }
      }
    }, std::make_integer_sequence<int, TH / 2>{});
  }
}

// what the kernel handles: a patch-plan 3x3 (conv_ops.h plan_make_patch: permuted rows, kswap) from one 64-channel tensor to 64 channels,
// plain store (no batch statistics, no accumulation, no affine on the reference), image sides multiples of the 4 x 16 tile
bool conv_c64_eligible(const IgemmArgs& a, int is_bf16) {
  if (!is_bf16 || a.patch != 1 || !patch3_eligible(a, 1)) return false;
  if (a.Cout != 64 || a.CoutPad != 64 || a.Cin != 64 || a.x.C[0] != 64 || a.ldY != 64 || !a.rowperm || a.splitk != 1) return false;
  if (a.Hg % TH || a.Wg % TW || a.Hin != a.Hg || a.Win != a.Wg || a.Hof != a.Hg || a.Wof != a.Wg) return false;
  if ((a.out_act != ACT_NONE && a.out_act != ACT_RELU) || (a.ref && (a.ref_act != ACT_RELU || a.out_act != ACT_NONE || a.pool_out))) return false;
  if (a.bn_part || a.accumulate || a.y_f32 || a.ref_a || a.split_c || a.x.aff_a[0] || a.x.act != ACT_NONE) return false;
  if (a.p_dhs != a.p_dws || a.p_dhf != a.p_dwf || a.p_dhf != (a.p_dhs > 0 ? -1 : 1)) return false;          // pad 1
  return true;
}

hipError_t launch_conv_c64(const IgemmArgs& a, hipStream_t st) {
  const int ntiles = a.N * (a.Hg / TH) * (a.Wg / TW);
  const int grid = ntiles < 1024 ? ntiles : 1024;                  // four two-wave blocks on each of the 256 CUs
  if (a.ref) hipLaunchKernelGGL(conv_c64_kernel<true>, dim3(grid), dim3(128), 2 * BUFB, st, a, ntiles);
  else hipLaunchKernelGGL(conv_c64_kernel<false>, dim3(grid), dim3(128), 2 * BUFB, st, a, ntiles);
  return hipGetLastError();
}

}  // namespace vp
