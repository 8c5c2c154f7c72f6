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
//   * the epilogue works on registers: with the row permutation of the packed weights a lane ends with 8 consecutive
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
constexpr int PPAD = 128;                     // ... padded to whole DMA rounds of the block's waves (NW x 16 pixels; NW = 2, 4)
constexpr int PBUFB = PPAD * 64;              // bytes of one channel chunk (32 channels) of a patch; a patch buffer is NCH of them
}  // namespace

// packed-bf16 helpers (two values per register).  A bf16 bit pattern read as a signed 16-bit integer orders like the float for
// non-negative values and is negative exactly for negative floats (and -0), so relu is an integer max with 0 and the max of two relu
// outputs an integer max
__device__ __forceinline__ unsigned pk_max_i16(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ unsigned pk_min_u16(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ unsigned pk_mul_lo_u16(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ unsigned dpp_xor1(unsigned v) {       // value of lane ^ 1: quad_perm [1, 0, 3, 2]
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

// REF: backward-data (output *= relu'(reference)); RELU: forward activation; POOL: also write the 2x2 max pool (RELU outputs only);
// STORE: write the full-resolution output (false with POOL: the real half of the perceptual trunk, nobody reads it)
// NCH: 64-byte channel chunks of the input (2: 64 channels, 4: 128); TPW: 16-channel MFMA tiles per wave - 36 NCH TPW weight registers:
//   NCH 2, TPW 2: 64 -> 64 (NW 2: VGG conv1_2 both passes) and 64 -> 128 (NW 4: conv2_1 forward);
//   NCH 4, TPW 1: 128 -> 128 (NW 8: conv2_2 both passes) and 128 -> 64 (NW 4: conv2_1 backward-data) - a fragment read then feeds half
//   the MFMAs, still 72 reads for 144 MFMAs per wave and tile
// NW: waves of a block; the block's output channels are 16 TPW NW
template <int NCH, int TPW, int NW, bool REF, bool RELU, bool POOL, bool STORE>
__global__ __launch_bounds__(NW * 64, 2) void conv_c64_kernel(const IgemmArgs a, const int ntiles) {
  static_assert(!POOL || (RELU && !REF), "the packed max pool compares relu outputs");
  static_assert((NCH == 2 && TPW == 2) || (NCH == 4 && TPW == 1), "36 NCH TPW <= 144 weight registers");
  constexpr int JP = PPAD / (16 * NW);          // patch DMA instructions per wave and chunk
  constexpr int CW = 16 * TPW;                  // output channels of a wave
  constexpr int COUT = CW * NW, CIN = 32 * NCH;
  constexpr int CB = COUT * 2;                  // bytes of one output pixel
  constexpr int BUFB = NCH * PBUFB;             // one patch buffer; two per block
  constexpr int KS = 9 * NCH;                   // (patch position, chunk) steps of the K loop
  static_assert(PPAD % (16 * NW) == 0, "whole DMA rounds per wave");
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
  uint4 W[KS][TPW];
  {
    const bf16* wp = reinterpret_cast<const bf16*>(a.Wp);
#pragma unroll
    for (int U = 0; U < KS; ++U) {
      const int cc = U / 9, u = U % 9;
      const int tap = flip ? 8 - u : u;
      const size_t chunk_row0 = (size_t)(tap * NCH + cc) * a.wp_rows;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const uint4 w = *reinterpret_cast<const uint4*>(wp + (chunk_row0 + (TPW * wave + t) * 16 + fi) * 32 + fg * 8);
        // the patch plan packs odd 16-byte k pieces with their halves swapped (PackDesc::kswap, for conv_patch3.hip's 8-byte fragment
        // reads); this kernel reads whole pieces: natural order
        W[U][t] = (fg & 1) ? make_uint4(w.z, w.w, w.x, w.y) : w;
      }
    }
  }
  // the 4 TPW consecutive output channels this lane finishes for pixel fi of a 16-pixel row (row permutation of the packed weights:
  // tile T of a 64-row block, accumulator rows 4 fg .. 4 fg + 3 -> channels 32 (T >> 1) + 8 fg + 4 (T & 1) + e)
  const int c0 = TPW == 2 ? 32 * wave + 8 * fg : 64 * (wave >> 2) + 32 * ((wave >> 1) & 1) + 8 * fg + 4 * (wave & 1);
  constexpr int NV = 4 * TPW;                   // values per lane and pixel; NV / 2 packed registers
  float bia[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) bia[e] = a.bias ? a.bias[c0 + e] : 0.f;

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
  // patch DMA lanes: instruction j of this wave covers patch pixels (wave + NW j) * 16 .. + 15, lane -> (pixel, slot)
  int ppy[JP], ppx[JP], prel[JP];
#pragma unroll
  for (int j = 0; j < JP; ++j) {
    const int pp = (wave + NW * j) * 16 + (lane >> 2);
    ppy[j] = pp < NPATCH ? pp / PW : 1 << 20;            // (rows beyond the patch: never inside the image)
    ppx[j] = pp % PW;
    prel[j] = pp < NPATCH ? (ppy[j] * a.Win + ppx[j]) * (CIN * 2) + (((lane & 3) ^ ((ppx[j] >> 1) & 3)) * 8) * (int)sizeof(bf16) : 0;
  }

  __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * CIN * sizeof(bf16)));
  const int tiles_x = a.Wg / TW, tpi = tiles_x * (a.Hg / TH);
  const bf16* refp = reinterpret_cast<const bf16*>(a.ref);
  bf16* Yp = reinterpret_cast<bf16*>(a.Y);
  bf16* Pp = reinterpret_cast<bf16*>(a.pool_out);
  constexpr int NSTY = TH * 16 * CB / 1024 / NW;                     // 1 KB store instructions of the staged tile per wave (2 TPW)
  constexpr int NST = (STORE ? NSTY : 0) + (POOL ? TH / 2 : 0);      // store instructions per tile and wave

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
    const int base = ((n * a.Hin + y0 + dh0) * a.Win + x0 + dw0) * CIN * (int)sizeof(bf16);
#pragma unroll
    for (int j = 0; j < JP; ++j) {
      const int ih = y0 + dh0 + ppy[j], iw = x0 + dw0 + ppx[j];
      const bool ok = (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      const unsigned vo = ok ? (unsigned)(base + prel[j]) : DMA_OOB;
      uint4* l0 = reinterpret_cast<uint4*>(smem + buf * BUFB) + (wave + NW * j) * 64;
#pragma unroll
      for (int c = 0; c < NCH; ++c) dma16_buf(rsX, vo, (unsigned)(c * 64), l0 + c * (PBUFB / 16));
    }
  };

  if (bt < ntiles) issue_patch(bt, 0);
  int it = 0;
  for (int t = bt; t < ntiles; t += G, ++it) {
    const int buf = it & 1;
    // This tile's patch has landed (own DMAs; the barrier covers the other wave's) and both waves are done with the other buffer.
    // The wait is COUNTED: behind this tile's DMAs (issued at the top of the previous trip) the wave has only issued the previous
    // tile's NST stores, which may stay in flight (vector-memory operations retire in issue order on gfx9-family counters - hipcc's own
    // counted waits across loads and stores rely on it); draining them too cost a store round trip per tile
    if (it == 0) wait_vm<0>();
    else wait_vm<NST>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int n = t / tpi, rem = t - n * tpi;
    const int y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TW;
    const size_t off00 = ((size_t)(n * a.Hof + y0) * a.Wof + x0 + fi) * COUT + c0;          // elements; tile row r: + r * Wof * COUT
    // backward-data: the reference rows of the tile, requested BEFORE the next patch so that they return first (loads retire in order)
    unsigned rz[TH][NV / 2];
    if constexpr (REF) {
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        if constexpr (NV == 8) {
          const uint4 z = *reinterpret_cast<const uint4*>(refp + off00 + (size_t)r * a.Wof * COUT);
          rz[r][0] = z.x; rz[r][1] = z.y; rz[r][NV / 2 - 2] = z.z; rz[r][NV / 2 - 1] = z.w;
        } else {
          const uint2 z = *reinterpret_cast<const uint2*>(refp + off00 + (size_t)r * a.Wof * COUT);
          rz[r][0] = z.x; rz[r][1] = z.y;
        }
      }
    }
    if (t + G < ntiles && (!(C64_ABL & 1) || it < 1)) issue_patch(t + G, buf ^ 1);
    int tb[3];
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) tb[pc] = tb0[pc] + buf * BUFB;

    // ---- the MFMAs of the tile: 36 fragment steps.  Step (cc, pc, R) reads ONE B fragment - patch row R, column shift pc, chunk cc - and
    // feeds it to every tile row r it belongs to (patch position pr = R - r in 0..2): 2 .. 6 MFMAs per read, 144 per tile and wave,
    // 36 fragment reads instead of the 72 of a row-by-row schedule.  Fragments are read LA steps ahead (LA + 1 register sets) ----
    f32x4 acc[TH][TPW];
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) acc[r][tt] = (f32x4){bia[4 * tt], bia[4 * tt + 1], bia[4 * tt + 2], bia[4 * tt + 3]};
    constexpr int LA = C64_LA, NS = LA + 1, NSTEP = NCH * 3 * PH;
    u32x4 rb[NS];
    auto rd = [&](auto sc) {
      constexpr int S = decltype(sc)::value, cc = S / (3 * PH), pc = (S / PH) % 3, R = S % PH;
      if constexpr ((C64_ABL & 8) != 0) { if (S >= NS) return; }
      rb[S % NS] = lds_rd128<cc * PBUFB + R * PW * 64>(tb[pc]);
    };
    static_steps([&](auto sc) { rd(sc); }, std::make_integer_sequence<int, LA>{});
    static_steps([&](auto sc) {
      constexpr int S = decltype(sc)::value, cc = S / (3 * PH), pc = (S / PH) % 3, R = S % PH;
      if constexpr (S + LA < NSTEP) rd(std::integral_constant<int, S + LA>{});
      constexpr int AHEAD = (NSTEP - 1 - S < LA ? NSTEP - 1 - S : LA);      // reads issued behind this step's
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD) : "memory");
      asm volatile("" : "+v"(rb[S % NS]));
      const uint4 fb = make_uint4(rb[S % NS].x, rb[S % NS].y, rb[S % NS].z, rb[S % NS].w);
      // rows the fragment belongs to, the row written longest ago first (R - 2 was last touched two reads back)
      static_steps([&](auto ri) {
        constexpr int r = R - 2 + decltype(ri)::value;
        if constexpr (r >= 0 && r < TH) {
          constexpr int U = cc * 9 + (R - r) * 3 + pc;
#pragma unroll
          for (int tt = 0; tt < TPW; ++tt) { if (!(C64_ABL & 4)) acc[r][tt] = mma16<bf16>(W[U][tt], fb, acc[r][tt]); else acc[r][tt][0] += __uint_as_float(fb.x ^ W[U][tt].x); }
        }
      }, std::make_integer_sequence<int, 3>{});
      __builtin_amdgcn_sched_barrier(0);
    }, std::make_integer_sequence<int, NSTEP>{});

    // ---- epilogue: (the bias is in the accumulators) rounding, relu, relu'(reference), one 16-byte store per lane and row, 2x2 max pool -
    // on PACKED bf16 pairs: 4 registers per row ----
    unsigned pk[TH][NV / 2];
#pragma unroll
    for (int r = 0; r < TH; ++r) {
#pragma unroll
      for (int e = 0; e < NV / 2; ++e) {
        unsigned o = Elem<bf16>::pack2(acc[r][(2 * e) >> 2][(2 * e) & 3], acc[r][(2 * e + 1) >> 2][(2 * e + 1) & 3]);
        if constexpr (RELU) o = pk_max_i16(o, 0u);
        // relu'(reference) per 16-bit half: min(max(ref as int16, 0), 1) is 1 exactly for a positive bf16; times the output's bits
        if constexpr (REF) o = pk_mul_lo_u16(o, pk_min_u16(pk_max_i16(rz[r][e], 0u), 0x00010001u));
        pk[r][e] = o;
      }
    }
    if constexpr (STORE) {
      // The waves of a block hold 32-byte (TPW 1: 8-byte ...) slices of every output pixel; written from the registers a store instruction
      // would touch sixteen partial lines.  The tile goes through LDS instead (16-byte slot s of pixel px at slot s ^ (px & 7):
      // conflict-free writes and b128 reads) and leaves as whole pixels, 1 KB contiguous per store instruction
      char* stg = smem + 2 * BUFB;
      constexpr int SPP = CB / 16;                   // 16-byte slots per pixel (8, 16)
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        char* q = stg + (r * 16 + fi) * CB + ((((c0 * 2) >> 4) ^ (fi & 7)) << 4) + ((c0 * 2) & 15);
        if constexpr (NV == 8) *reinterpret_cast<uint4*>(q) = make_uint4(pk[r][0], pk[r][1], pk[r][NV / 2 - 2], pk[r][NV / 2 - 1]);
        else *reinterpret_cast<uint2*>(q) = make_uint2(pk[r][0], pk[r][1]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (a raw s_barrier does not wait for this wave's LDS writes)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!(C64_ABL & 2) || pk[0][0] == 0x12345u) {
#pragma unroll
        for (int j = 0; j < NSTY; ++j) {
          constexpr int PPI = 64 / SPP;              // pixels per store instruction (8, 4)
          const int g = wave * NSTY + j;             // the block's store instructions, in pixel order
          const int pi = g * PPI + lane / SPP, row = pi >> 4, px = pi & 15, sl = lane % SPP;
          const uint4 o = *reinterpret_cast<const uint4*>(stg + (row * 16 + px) * CB + ((sl ^ (px & 7)) << 4));
          unsigned* yp = reinterpret_cast<unsigned*>(Yp + ((size_t)(n * a.Hof + y0 + row) * a.Wof + x0 + px) * COUT + sl * 8);
          __builtin_nontemporal_store(o.x, yp); __builtin_nontemporal_store(o.y, yp + 1);
          __builtin_nontemporal_store(o.z, yp + 2); __builtin_nontemporal_store(o.w, yp + 3);
        }
      }
    }
    if constexpr (POOL) {
#pragma unroll
      for (int q = 0; q < TH; q += 2) {
        unsigned m[NV / 2];
#pragma unroll
        for (int e = 0; e < NV / 2; ++e) {
          m[e] = pk_max_i16(pk[q][e], pk[q + 1][e]);
          m[e] = pk_max_i16(m[e], dpp_xor1(m[e]));
        }
        if (!(fi & 1)) {
          const size_t po = ((size_t)(n * (a.Hg >> 1) + ((y0 + q) >> 1)) * (a.Wg >> 1) + ((x0 + fi) >> 1)) * COUT + c0;
          if constexpr (NV == 8) *reinterpret_cast<uint4*>(Pp + po) = make_uint4(m[0], m[1], m[NV / 2 - 2], m[NV / 2 - 1]);
          else *reinterpret_cast<uint2*>(Pp + po) = make_uint2(m[0], m[1]);
        }
      }
    }
  }
}

// what the kernel handles: a patch-plan 3x3 (conv_ops.h plan_make_patch: permuted rows, kswap) from ONE tensor of 64 channels to 64 / 128 or
// of 128 channels to 64 / 128, plain store (no batch statistics, no accumulation, no affine on the reference), relu / no activation,
// optional fused 2x2 max pool of relu outputs, optional relu'(reference) product, image sides multiples of the 4 x 16 tile
bool conv_c64_eligible(const IgemmArgs& a, int is_bf16) {
  if (!is_bf16 || a.patch != 1 || !patch3_eligible(a, 1)) return false;
  if ((a.Cin != 64 && a.Cin != 128) || a.x.C[0] != a.Cin || (a.Cout != 64 && a.Cout != 128)) return false;
  if (a.CoutPad != a.Cout || a.ldY != a.Cout || !a.rowperm || a.splitk != 1) return false;
  if (a.Hg % TH || a.Wg % TW || a.Hin != a.Hg || a.Win != a.Wg || a.Hof != a.Hg || a.Wof != a.Wg) return false;
  if ((a.out_act != ACT_NONE && a.out_act != ACT_RELU) || (a.ref && (a.ref_act != ACT_RELU || a.out_act != ACT_NONE || a.pool_out))) return false;
  if (a.pool_out && (a.out_act != ACT_RELU || (a.Hg & 1))) return false;
  if (a.pool_only && !a.pool_out) return false;
  if (a.bn_part || a.accumulate || a.y_f32 || a.ref_a || a.split_c || a.x.aff_a[0] || a.x.act != ACT_NONE) return false;
  if (a.p_dhs != a.p_dws || a.p_dhf != a.p_dwf || a.p_dhf != (a.p_dhs > 0 ? -1 : 1)) return false;          // pad 1
  return (size_t)a.N * a.Hin * a.Win * a.Cin * 2 < 0x70000000ull;
}

template <int NCH, int TPW, int NW>
static hipError_t launch_c64_t(const IgemmArgs& a, hipStream_t st) {
  const int ntiles = a.N * (a.Hg / TH) * (a.Wg / TW);
  void (*kern)(const IgemmArgs, const int);
  if (a.ref) kern = conv_c64_kernel<NCH, TPW, NW, true, false, false, true>;
  else if (a.pool_out) kern = a.pool_only ? conv_c64_kernel<NCH, TPW, NW, false, true, true, false> : conv_c64_kernel<NCH, TPW, NW, false, true, true, true>;
  else kern = a.out_act == ACT_RELU ? conv_c64_kernel<NCH, TPW, NW, false, true, false, true> : conv_c64_kernel<NCH, TPW, NW, false, false, false, true>;
  // LDS: two patch buffers + the output staging tile; blocks per CU: 8 / NW (two waves per SIMD) - all of them fit (40 .. 80 KB each)
  const int smem = 2 * NCH * PBUFB + TH * 16 * (16 * TPW * NW * 2);
  const int blocks = 256 * (8 / NW);
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipLaunchKernelGGL(kern, dim3(ntiles < blocks ? ntiles : blocks), dim3(NW * 64), smem, st, a, ntiles);
  return hipGetLastError();
}

hipError_t launch_conv_c64(const IgemmArgs& a, hipStream_t st) {
  if (a.Cin == 64) return a.Cout == 64 ? launch_c64_t<2, 2, 2>(a, st) : launch_c64_t<2, 2, 4>(a, st);
  return a.Cout == 64 ? launch_c64_t<4, 1, 4>(a, st) : launch_c64_t<4, 1, 8>(a, st);
}

}  // namespace vp
