// conv_s2c64_kernel: the 4x4 / stride-2 convolutions from 64 to 128 channels in front of a batch-norm (discriminator layer_2, encoder_2,
// encoder_fg_2: pixrefer.py:61-74, 142-160) with the WEIGHTS RESIDENT IN REGISTERS - the third member of the conv_c64.hip family - and the
// batch statistics formed in the block.
//
// On the generic gather-per-tap kernel (igemm_dma, 128 x 256 tile) these launches ran at 520-630 TF: every 256-pixel tile re-fetched the
// 256 KB weight matrix from L2 and every input pixel four times (once per tap that touches it).  Here
//   * a block is EIGHT waves; wave w keeps the 16 x 1024 weight slice of packed rows 16 w .. 16 w + 15 as 32 MFMA A fragments in 128
//     registers for the life of the block (the generic plan's packing is read unchanged: [K chunk][row][32 k], rows permuted inside
//     64-row blocks - IgemmArgs::rowperm);
//   * a tile is 4 x 16 output pixels; its input patch (10 rows x 34 columns x 64 channels, 43.5 KB, LDS-DMA, double-buffered across
//     tiles) is stored by COLUMN PARITY: [chunk][row][parity][17 columns][64 bytes], so the 16 output pixels of a tile row read 16
//     consecutive slots for every tap (patch column 2 x + kw = parity kw & 1, index x + (kw >> 1)) and the conflict-free slot swizzle
//     of conv_c64.hip applies unchanged;
//   * a fragment read (patch row R, tap column kw, chunk) feeds both output rows it belongs to (kh = R - 2 r): 80 reads for 128 MFMAs
//     per wave and tile;
//   * the tile leaves through LDS as whole output rows (16 pixels x 256 bytes = four 1 KB store instructions per row);
//   * BATCH STATISTICS: a lane keeps running sums of the ROUNDED outputs (and their squares) of its four channels over all tiles of a
//     batch-norm group the block walks; at a group boundary / the end the 16 lanes of a channel quad combine (fixed order) and the block
//     writes ONE partial row per group: bn_part[group][block][2][128] - the chunk count of the finalize is the grid size.
//   * one block per CU (two waves per SIMD), counted vmcnt across the tile loop, XCD-aware block -> tile map: as conv_c64.hip.
//   * TWO-OUTPUT form (PAIR): the backward-data of the 256 -> 64 transposed convolution merged2_decoder_2 is the same convolution with
//     2 x 128 output channels: see the template comment.
// bf16 only; output grids multiples of 4 x 16.
#include "conv_ops.h"
#include "igemm_device.h"
#include "launch.h"
#include "patch_device.h"

namespace vp {

namespace {
constexpr int TH = 4, TW = 16;
constexpr int PRW = 17;                       // slots of one (row, parity) run: patch columns parity, parity + 2, ..
constexpr int PHR = 2 * TH + 2;               // 10 patch rows
constexpr int NPATCH = PHR * 2 * PRW;         // 340 patch pixels
constexpr int NROUND = 22;                    // DMA rounds of 16 pixels (352 >= 340)
constexpr int PBUFB = NROUND * 16 * 64;       // bytes of one channel chunk (32 channels) of a patch
constexpr int NCH = 2;                        // 64 input channels
constexpr int BUFB = NCH * PBUFB;             // one patch buffer; two per block
constexpr int STGB = TH * TW * 256;           // output staging: 4 rows x 16 pixels x 128 channels
constexpr int NW = 8;
constexpr int JP = (NROUND + NW - 1) / NW;    // DMA rounds per wave (the last one only for waves < NROUND - 2 NW)
}  // namespace

__device__ __forceinline__ unsigned s2_pk_max_i16(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ unsigned s2_pk_min_u16(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ unsigned s2_pk_mul_lo_u16(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// STATS: batch statistics of the rounded outputs (the forward layers).  PAIR: the two-output backward-data of a 256 -> 64 transposed
// convolution (merged2_decoder_2: IgemmArgs::split_c = 128): even / odd virtual blocks compute packed rows [0, 128) -> Y / ref /
// accumulate and [128, 256) -> Y2 / ref2 / accumulate2 of the same pixel tiles, output *= relu'(reference), (+= what another consumer
// wrote first)
template <bool STATS, bool PAIR>
__global__ __launch_bounds__(NW * 64, 1) void conv_s2c64_kernel(const IgemmArgs a, const int ntiles) {
  static_assert(!(STATS && PAIR), "statistics belong to the forward layers");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fg = lane >> 4;

  // XCD-aware virtual block index; PAIR: its low bit selects the output half
  const int G = gridDim.x;
  int vb = blockIdx.x;
  if ((G & 7) == 0) vb = (vb & 7) * (G >> 3) + (vb >> 3);
  const int half = PAIR ? (vb & 1) : 0, row0 = 128 * half;
  const int bt = PAIR ? (vb >> 1) : vb, GT = PAIR ? (G >> 1) : G;

  // weights: K chunk (4 kh + kw) * 2 + c, packed row row0 + 16 wave + fi, piece fg
  uint4 W[16][NCH];
  {
    const bf16* wp = reinterpret_cast<const bf16*>(a.Wp);
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        W[t][c] = *reinterpret_cast<const uint4*>(wp + ((size_t)(t * NCH + c) * a.wp_rows + row0 + 16 * wave + fi) * 32 + fg * 8);
  }
  // accumulator rows 4 fg .. 4 fg + 3 of the wave's tile T = wave & 3 of 64-row block wave >> 2: channels 64 b + 32 (T >> 1) + 8 fg + 4 (T & 1) + e
  const int c0 = 64 * (wave >> 2) + 32 * ((wave >> 1) & 1) + 8 * fg + 4 * (wave & 1);
  f32x4 bia = (f32x4){0.f, 0.f, 0.f, 0.f};        // (no bias in front of a batch-norm - it cancels; the plain op has one)
  if (a.bias) bia = (f32x4){a.bias[c0], a.bias[c0 + 1], a.bias[c0 + 2], a.bias[c0 + 3]};

  // B fragment lane offsets per column index shift s = kw >> 1 (conflict-free ds_read_b128: conv_c64.hip)
  int tb0[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int px = fi + s;
    tb0[s] = (px << 6) + (((fg ^ (px >> 1)) & 3) << 4);
  }
  // patch DMA lanes: round wave + NW j covers patch slots 16 (wave + NW j) .. + 15; slot pp = (R * 2 + parity) * 17 + index
  int ppy[JP], ppx[JP], prel[JP];
#pragma unroll
  for (int j = 0; j < JP; ++j) {
    const int pp = (wave + NW * j) * 16 + (lane >> 2);
    const int run = pp / PRW, idx = pp - run * PRW;
    ppy[j] = pp < NPATCH ? (run >> 1) : 1 << 20;             // (slots beyond the patch: never inside the image)
    ppx[j] = 2 * idx + (run & 1);
    prel[j] = pp < NPATCH ? (ppy[j] * a.Win + ppx[j]) * 128 + (((lane & 3) ^ ((idx >> 1) & 3)) * 8) * (int)sizeof(bf16) : 0;
  }
  const bool last_round = wave + NW * (JP - 1) < NROUND;

  __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * 64 * sizeof(bf16)));
  const int tiles_x = a.Wg / TW, tpi = tiles_x * (a.Hg / TH);
  bf16* Yp = reinterpret_cast<bf16*>(half ? a.Y2 : a.Y);
  const bf16* refp = reinterpret_cast<const bf16*>(half ? a.ref2 : a.ref);
  const bool accum = PAIR && (half ? a.accumulate2 : a.accumulate);
  constexpr int NST = TH * TW * 256 / 1024 / NW;             // 1 KB store instructions per tile and wave (2)

  // patch origin: input pixel (2 q0 - 1, 2 r0 - 1)
  auto issue_patch = [&](int t, int buf) {
    const int n = t / tpi, rem = t - n * tpi;
    const int y0 = 2 * (rem / tiles_x) * TH - 1, x0 = 2 * (rem % tiles_x) * TW - 1;
    const int base = ((n * a.Hin + y0) * a.Win + x0) * 128;
#pragma unroll
    for (int j = 0; j < JP; ++j) {
      if (j == JP - 1 && !last_round) break;
      const int ih = y0 + ppy[j], iw = x0 + ppx[j];
      const bool ok = (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      const unsigned vo = ok ? (unsigned)(base + prel[j]) : DMA_OOB;
      uint4* l0 = reinterpret_cast<uint4*>(smem + buf * BUFB) + (wave + NW * j) * 64;
#pragma unroll
      for (int c = 0; c < NCH; ++c) dma16_buf(rsX, vo, (unsigned)(c * 64), l0 + c * (PBUFB / 16));
    }
  };

  // running statistics of the current batch-norm group: this lane's four channels over its pixels
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  int cur_grp = -1;
  unsigned done_mask = 0;                        // groups this block has written a row for
  const int ngroups = STATS ? (ntiles + a.bn_tpg - 1) / a.bn_tpg : 0;
  auto flush = [&](int grp) {
    float s[4], q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[e] = ssum[e]; q[e] = ssq[e];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) { s[e] += __shfl_xor(s[e], m); q[e] += __shfl_xor(q[e], m); }
      ssum[e] = 0.f; ssq[e] = 0.f;
    }
    if (fi == 0) {
      double* row = a.bn_part + ((size_t)(grp * a.bn_nchunk + blockIdx.x) * 2) * a.Cout + c0;
#pragma unroll
      for (int e = 0; e < 4; ++e) { row[e] = (double)s[e]; row[a.Cout + e] = (double)q[e]; }
    }
    done_mask |= 1u << grp;
  };

  if (bt < ntiles) issue_patch(bt, 0);
  // the weights are complete HERE (the loads precede the patch DMAs; the counter retires in order): without a use in front of the loop
  // hipcc waits for them at their first MFMA inside it - every trip, with counts that also drain the next tile's patch
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(W[t][c].x), "+v"(W[t][c].y), "+v"(W[t][c].z), "+v"(W[t][c].w));
  int it = 0;
  for (int t = bt; t < ntiles; t += GT, ++it) {
    const int buf = it & 1;
    // this tile's patch has landed (counted: only the previous tile's NST stores were issued behind its DMAs, conv_c64.hip)
    if (it == 0) wait_vm<0>();
    else wait_vm<NST>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // PAIR: the reference rows (and the rows another consumer wrote) of the tile, requested BEFORE the next patch so that they return first
    uint2 rz[PAIR ? TH : 1], ry[PAIR ? TH : 1];
    if constexpr (PAIR) {
      const int n = t / tpi, rem = t - n * tpi;
      const size_t o00 = ((size_t)(n * a.Hof + (rem / tiles_x) * TH) * a.Wof + (rem % tiles_x) * TW + fi) * 128 + c0;
#pragma unroll
      for (int r = 0; r < TH; ++r) rz[r] = *reinterpret_cast<const uint2*>(refp + o00 + (size_t)r * a.Wof * 128);
      if (accum) {
#pragma unroll
        for (int r = 0; r < TH; ++r) ry[r] = *reinterpret_cast<const uint2*>(Yp + o00 + (size_t)r * a.Wof * 128);
      }
    }
    if (t + GT < ntiles) issue_patch(t + GT, buf ^ 1);
    if constexpr (STATS) {
      const int grp = t / a.bn_tpg;
      if (grp != cur_grp) {
        if (cur_grp >= 0) flush(cur_grp);
        cur_grp = grp;
      }
    }
    int tb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) tb[s] = tb0[s] + buf * BUFB;

    // ---- 80 fragment steps (chunk, kw, patch row R): one ds_read_b128 each, fed to output rows R >> 1 (kh = R & 1) and (R >> 1) - 1
    // (kh = 2 + (R & 1)): 128 MFMAs ----
    f32x4 acc[TH];
#pragma unroll
    for (int r = 0; r < TH; ++r) acc[r] = bia;
    constexpr int LA = 2, NS = LA + 1, NSTEP = NCH * 4 * PHR;
    u32x4 rb[NS];
    auto rd = [&](auto sc) {
      constexpr int S = decltype(sc)::value, c = S / (4 * PHR), kw = (S / PHR) % 4, R = S % PHR;
      rb[S % NS] = lds_rd128<c * PBUFB + (R * 2 + (kw & 1)) * PRW * 64>(tb[kw >> 1]);
    };
    static_steps([&](auto sc) { rd(sc); }, std::make_integer_sequence<int, LA>{});
    static_steps([&](auto sc) {
      constexpr int S = decltype(sc)::value, c = S / (4 * PHR), kw = (S / PHR) % 4, R = S % PHR;
      if constexpr (S + LA < NSTEP) rd(std::integral_constant<int, S + LA>{});
      constexpr int AHEAD = (NSTEP - 1 - S < LA ? NSTEP - 1 - S : LA);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD) : "memory");
      asm volatile("" : "+v"(rb[S % NS]));
      const uint4 fb = make_uint4(rb[S % NS].x, rb[S % NS].y, rb[S % NS].z, rb[S % NS].w);
      static_steps([&](auto ri) {
        constexpr int r = (R >> 1) - decltype(ri)::value;
        if constexpr (r >= 0 && r < TH) {
          constexpr int kh = R - 2 * r;
          acc[r] = mma16<bf16>(W[4 * kh + kw][c], fb, acc[r]);
        }
      }, std::make_integer_sequence<int, 2>{});
      __builtin_amdgcn_sched_barrier(0);
    }, std::make_integer_sequence<int, NSTEP>{});

    // ---- epilogue: rounding, statistics of the rounded values, the tile through LDS, whole output rows out ----
    char* stg = smem + 2 * BUFB;
    // 16-byte slot of this lane's 8 bytes inside its pixel's 256: c0 / 8, at physical slot (c0 / 8) ^ fi (conflict-free ds_write_b64)
    const int wslot = ((((c0 >> 3) ^ fi) & 15) << 4) + ((c0 & 4) << 1);
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      uint2 pk;
      if constexpr (PAIR) {
        if (accum) {      // += the gradient another consumer wrote first (of the masked product: added after the mask below would be wrong)
          const float m0 = __uint_as_float(rz[r].x << 16) > 0.f ? 1.f : 0.f, m1 = __uint_as_float(rz[r].x & 0xffff0000u) > 0.f ? 1.f : 0.f;
          const float m2 = __uint_as_float(rz[r].y << 16) > 0.f ? 1.f : 0.f, m3 = __uint_as_float(rz[r].y & 0xffff0000u) > 0.f ? 1.f : 0.f;
          acc[r][0] = fmaf(acc[r][0], m0, __uint_as_float(ry[r].x << 16)); acc[r][1] = fmaf(acc[r][1], m1, __uint_as_float(ry[r].x & 0xffff0000u));
          acc[r][2] = fmaf(acc[r][2], m2, __uint_as_float(ry[r].y << 16)); acc[r][3] = fmaf(acc[r][3], m3, __uint_as_float(ry[r].y & 0xffff0000u));
        }
      }
      pk.x = Elem<bf16>::pack2(acc[r][0], acc[r][1]);
      pk.y = Elem<bf16>::pack2(acc[r][2], acc[r][3]);
      if constexpr (PAIR) {
        if (!accum) {     // relu'(reference) per 16-bit half: min(max(ref as int16, 0), 1) is 1 exactly for a positive bf16; times the output's bits
          pk.x = s2_pk_mul_lo_u16(pk.x, s2_pk_min_u16(s2_pk_max_i16(rz[r].x, 0u), 0x00010001u));
          pk.y = s2_pk_mul_lo_u16(pk.y, s2_pk_min_u16(s2_pk_max_i16(rz[r].y, 0u), 0x00010001u));
        }
      }
      if constexpr (STATS) {
        const float v0 = __uint_as_float(pk.x << 16), v1 = __uint_as_float(pk.x & 0xffff0000u);
        const float v2 = __uint_as_float(pk.y << 16), v3 = __uint_as_float(pk.y & 0xffff0000u);
        ssum[0] += v0; ssum[1] += v1; ssum[2] += v2; ssum[3] += v3;
        ssq[0] = fmaf(v0, v0, ssq[0]); ssq[1] = fmaf(v1, v1, ssq[1]); ssq[2] = fmaf(v2, v2, ssq[2]); ssq[3] = fmaf(v3, v3, ssq[3]);
      }
      *reinterpret_cast<uint2*>(stg + (r * TW + fi) * 256 + wslot) = pk;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // wave w stores half of output row w >> 1: pixels 8 (w & 1) + 4 j + (lane >> 4); a lane reads PHYSICAL slot lane & 15 of its pixel
    // (= logical slot (lane & 15) ^ pixel: the 64 lanes of an instruction still cover one contiguous 1 KB)
    {
      const int n = t / tpi, rem = t - n * tpi;
      const int q0 = (rem / tiles_x) * TH, r0 = (rem % tiles_x) * TW;
      const int row = wave >> 1;
#pragma unroll
      for (int j = 0; j < NST; ++j) {
        const int px = 8 * (wave & 1) + 4 * j + (lane >> 4), ps = lane & 15;
        const uint4 o = *reinterpret_cast<const uint4*>(stg + (row * TW + px) * 256 + (ps << 4));
        unsigned* yp = reinterpret_cast<unsigned*>(Yp + ((size_t)(n * a.Hof + q0 + row) * a.Wof + r0 + px) * 128 + ((ps ^ px) & 15) * 8);
        __builtin_nontemporal_store(o.x, yp); __builtin_nontemporal_store(o.y, yp + 1);
        __builtin_nontemporal_store(o.z, yp + 2); __builtin_nontemporal_store(o.w, yp + 3);
      }
    }
  }
  if constexpr (STATS) {
    if (cur_grp >= 0) flush(cur_grp);
    // groups this block never walked: zero rows (the finalize sums every block's row of every group)
    for (int g = 0; g < ngroups; ++g) {
      if (done_mask & (1u << g)) continue;
      if (fi == 0) {
        double* row = a.bn_part + ((size_t)(g * a.bn_nchunk + blockIdx.x) * 2) * a.Cout + c0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { row[e] = 0.0; row[a.Cout + e] = 0.0; }
      }
    }
  }
}

// tiles per image / grid of the launch (the step executor sizes the statistics table with them)
int conv_s2c64_grid(const IgemmArgs& a) {
  const int ntiles = a.N * (a.Hg / TH) * (a.Wg / TW);
  return ntiles < 256 ? ntiles : 256;
}
int conv_s2c64_tiles_per_image(const IgemmArgs& a) { return (a.Hg / TH) * (a.Wg / TW); }

// run-time side of plan_s2c64_eligible (conv_ops.h): a plan made by plan_make_s2c64 with a raw bf16 output and a plain input; either the
// forward form (128 output channels, no reference) or the two-output backward-data form (split_c = 128: 2 x 128 channels, relu'(reference))
bool conv_s2c64_eligible(const IgemmArgs& a, int is_bf16) {
  if (!is_bf16 || a.patch != 4 || a.ldY != 128 || a.Cin != 64 || a.x.C[0] != 64 || a.x.C[1] != 0 || !a.rowperm || a.splitk != 1) return false;
  if (a.Hg % TH || a.Wg % TW || a.Hin != 2 * a.Hg || a.Win != 2 * a.Wg) return false;
  if (a.out_act != ACT_NONE || a.y_f32 || a.pool_out || a.x.aff_a[0] || a.x.act != ACT_NONE || a.ref_a) return false;
  if (a.split_c) {
    return a.split_c == 128 && a.Cout == 256 && a.wp_rows == 256 && a.Y2 && a.ref && a.ref2 && a.ref_act == ACT_RELU && !a.y2_f32 && !a.bias && !a.bn_part;
  }
  return a.Cout == 128 && a.wp_rows == 128 && !a.ref && !a.accumulate;
}

hipError_t launch_conv_s2c64(const IgemmArgs& a, hipStream_t st) {
  const int ntiles = a.N * (a.Hg / TH) * (a.Wg / TW);
  const int pair = a.split_c ? 1 : 0;
  int grid = conv_s2c64_grid(a);
  if (pair) grid = 2 * ntiles < 512 ? 2 * ntiles : 512;        // both halves of a tile in neighbouring virtual blocks (one XCD); two rounds of 256
  if (a.bn_part && (pair || a.bn_nchunk != grid || a.bn_tpg <= 0 || (ntiles + a.bn_tpg - 1) / a.bn_tpg > 32)) return hipErrorInvalidValue;
  const int ki = pair ? 2 : (a.bn_part ? 1 : 0);
  void (*kerns[3])(const IgemmArgs, const int) = {conv_s2c64_kernel<false, false>, conv_s2c64_kernel<true, false>, conv_s2c64_kernel<false, true>};
  void (*kern)(const IgemmArgs, const int) = kerns[ki];
  const int smem = 2 * BUFB + STGB;                            // 104 KB
  static bool attr_done[3] = {false, false, false};
  if (!attr_done[ki]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_done[ki] = true; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, st, a, ntiles);
  return hipGetLastError();
}

}  // namespace vp
