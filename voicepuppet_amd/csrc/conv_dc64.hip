// conv_dc64_kernel: the four parity classes of a 4x4 / stride-2 TRANSPOSED convolution from 128 to 64 channels - the backward-data pass of
// the 64 -> 128 stride-2 convolutions (discriminator layer_2 in both gradient passes, encoder_2 / encoder_fg_2: pixrefer.py:61-74) - with
// the WEIGHTS RESIDENT IN REGISTERS, the sibling of conv_c64.hip.
//
// On the unrolled 2x2-tap patch kernel (conv_patch2.hip, 64 x 256 tile) these launches ran at 430-550 TF, the least efficient class of
// the step that carries real time: a class's weight matrix is 64 x 512 (64 KB) and every 256-pixel tile of every class re-fetched it
// from L2 next to a 74 KB input patch that the four classes each fetched again.  Here
//   * a block is FOUR waves and owns one row parity ph: waves 0-1 compute class (ph, 0), waves 2-3 class (ph, 1), each wave 32 of the 64
//     output channels of its class with its 32 x 512 weight slice as 32 MFMA A fragments in 128 registers for the life of the block;
//   * the two classes read ONE input patch ((4 + 1) x (16 + 2) pixels x 128 channels, 24 KB, LDS-DMA, double-buffered across tiles):
//     class pw uses patch columns pw .. pw + 16.  A fragment read (patch row R, column shift, chunk) feeds both tile rows it belongs to;
//   * the outputs of the two classes interleave along an output row: the tile goes through LDS and leaves as whole output rows (32 pixels
//     x 128 bytes = 4 KB contiguous per row) with the act'(reference) product of the chain rule applied in registers before;
//   * two blocks per CU (two waves per SIMD), counted vmcnt across the tile loop, XCD-aware block -> tile map: as conv_c64.hip.
// bf16 only; class grids multiples of 4 x 16; no batch statistics: the launches above.  conv_dc256_kernel (below): the FORWARD form for
// 256 input channels from two tensors with the statistics in the block (merged2_decoder_2).
#include "conv_ops.h"
#include "igemm_device.h"
#include "launch.h"
#include "patch_device.h"

namespace vp {

namespace {
constexpr int TH = 4, TW = 16, PW = TW + 2, PH = TH + 1;
constexpr int NPATCH = PW * PH;               // 90 patch pixels
constexpr int NROUND = 6;                     // DMA rounds of 16 pixels (96 >= 90)
constexpr int PBUFB = NROUND * 16 * 64;       // bytes of one channel chunk (32 channels) of a patch
constexpr int NCH = 4;                        // 128 input channels
constexpr int BUFB = NCH * PBUFB;             // one patch buffer; two per block
constexpr int STGB = TH * 32 * 128;           // output staging: 4 rows x 32 pixels x 64 channels
}  // namespace

__device__ __forceinline__ unsigned dc_pk_max_i16(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ unsigned dc_pk_min_u16(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ unsigned dc_pk_mul_lo_u16(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// RACT: activation whose derivative at the reference multiplies the output (ACT_NONE: no reference, ACT_LRELU, ACT_RELU);
// ACC: the output tensor already holds another consumer's gradient contribution (encoder_1: decoder_1 wrote first): add to it
// CSUM (round 6): the launch completes the gradient of a tensor whose producer has no batch-norm (discriminator layer_1, encoder_1,
// encoder_fg_1): the column sums of the gradient AS STORED - that producer's bias gradient - are formed here as running per-lane sums
// over the tiles the persistent block walks, one partial row per block (IgemmArgs::colsum_part, [grid][2][64] doubles in the layout
// colsum_finalize_kernel reads): the bias gradient costs no pass of its own over the 67-201 MB tensor.
template <int RACT, bool ACC, bool CSUM = false>
__global__ __launch_bounds__(256, 2) void conv_dc64_kernel(const IgemmArgs a, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fg = lane >> 4;
  const int pw = wave >> 1, hh = wave & 1;                  // column parity of this wave's class, its half of the 64 output channels

  // XCD-aware order: consecutive virtual indices (same XCD) are the two row parities of one tile, then the neighbouring tiles
  const int G = gridDim.x;
  int v = blockIdx.x;
  if ((G & 7) == 0) v = (v & 7) * (G >> 3) + (v >> 3);
  const int ph = v & 1, bt = v >> 1, GT = G >> 1;
  const int cls = 2 * ph + pw;

  // weights of class cls: [tap][chunk][row][32 k] (PackDesc of the patch2 plan: permuted rows, odd 16-byte pieces half-swapped - undone
  // here, whole pieces are read); patch position (pr, pc) holds tap 3 - (2 pr + pc)
  uint4 W[4][NCH][2];
  {
    const bf16* wp = reinterpret_cast<const bf16*>(a.Wp) + (size_t)cls * a.wp_rows * a.Kpad;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const uint4 w = *reinterpret_cast<const uint4*>(wp + ((size_t)(t * NCH + c) * a.wp_rows + (2 * hh + tt) * 16 + fi) * 32 + fg * 8);
          W[t][c][tt] = (fg & 1) ? make_uint4(w.z, w.w, w.x, w.y) : w;
        }
  }
  const int c0 = 32 * hh + 8 * fg;              // the 8 consecutive output channels this lane finishes for pixel fi of a tile row

  // B fragment lane offsets: patch column fi + pw + pc, slot of piece fg = fg ^ ((px >> 1) & 3) (conflict-free ds_read_b128, conv_c64.hip)
  int tb0[2];
#pragma unroll
  for (int pc = 0; pc < 2; ++pc) {
    const int px = fi + pw + pc;
    tb0[pc] = (px << 6) + (((fg ^ (px >> 1)) & 3) << 4);
  }
  // patch DMA lanes: rounds wave and wave + 4 (the second only for waves 0, 1) of 16 pixels each
  int ppy[2], ppx[2], prel[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pp = (wave + 4 * j) * 16 + (lane >> 2);
    ppy[j] = pp < NPATCH ? pp / PW : 1 << 20;
    ppx[j] = pp % PW;
    prel[j] = pp < NPATCH ? (ppy[j] * a.Win + ppx[j]) * 256 + (((lane & 3) ^ ((ppx[j] >> 1) & 3)) * 8) * (int)sizeof(bf16) : 0;
  }
  const bool second = wave + 4 < NROUND;

  __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.x.ptr[0], (unsigned)((size_t)a.N * a.Hin * a.Win * 128 * sizeof(bf16)));
  const int tiles_x = a.Wg / TW, tpi = tiles_x * (a.Hg / TH);
  const bf16* refp = reinterpret_cast<const bf16*>(a.ref);
  bf16* Yp = reinterpret_cast<bf16*>(a.Y);
  constexpr int NST = TH;                       // store instructions per tile and wave (the loads of ACC / RACT are issued BEFORE the patch DMAs)

  // patch origin: input pixel (q0 + ph - 1, r0 - 1)
  auto issue_patch = [&](int t, int buf) {
    const int n = t / tpi, rem = t - n * tpi;
    const int y0 = (rem / tiles_x) * TH + ph - 1, x0 = (rem % tiles_x) * TW - 1;
    const int base = ((n * a.Hin + y0) * a.Win + x0) * 128 * (int)sizeof(bf16);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j == 1 && !second) break;
      const int ih = y0 + ppy[j], iw = x0 + ppx[j];
      const bool ok = (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      const unsigned vo = ok ? (unsigned)(base + prel[j]) : DMA_OOB;
      uint4* l0 = reinterpret_cast<uint4*>(smem + buf * BUFB) + (wave + 4 * j) * 64;
#pragma unroll
      for (int c = 0; c < NCH; ++c) dma16_buf(rsX, vo, (unsigned)(c * 64), l0 + c * (PBUFB / 16));
    }
  };

  float cs[CSUM ? 8 : 1];
#pragma unroll
  for (int e = 0; e < (CSUM ? 8 : 1); ++e) cs[e] = 0.f;
  if (bt < ntiles) issue_patch(bt, 0);
  int it = 0;
  for (int t = bt; t < ntiles; t += GT, ++it) {
    const int buf = it & 1;
    // this tile's patch has landed (counted: only the previous tile's NST stores were issued behind its DMAs, conv_c64.hip)
    if (it == 0) wait_vm<0>();
    else wait_vm<NST>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int n = t / tpi, rem = t - n * tpi;
    const int q0 = (rem / tiles_x) * TH, r0 = (rem % tiles_x) * TW;
    // output pixel of (tile row r, column fi) of this wave's class: (2 (q0 + r) + ph, 2 (r0 + fi) + pw)
    const size_t off00 = ((size_t)(n * a.Hof + 2 * q0 + ph) * a.Wof + 2 * (r0 + fi) + pw) * 64 + c0;     // row r: + 2 r Wof 64
    uint4 rz[TH];
    if constexpr (RACT != ACT_NONE) {
#pragma unroll
      for (int r = 0; r < TH; ++r) rz[r] = *reinterpret_cast<const uint4*>(refp + off00 + (size_t)(2 * r) * a.Wof * 64);
    }
    uint4 ry[ACC ? TH : 1];
    if constexpr (ACC) {
#pragma unroll
      for (int r = 0; r < TH; ++r) ry[r] = *reinterpret_cast<const uint4*>(Yp + off00 + (size_t)(2 * r) * a.Wof * 64);
    }
    if (t + GT < ntiles) issue_patch(t + GT, buf ^ 1);
    int tb[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) tb[pc] = tb0[pc] + buf * BUFB;

    // ---- 40 fragment steps (chunk, pc, patch row R): one ds_read_b128 each, fed to tile rows R - 1 (pr = 1) and R (pr = 0): 128 MFMAs ----
    f32x4 acc[TH][2];
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) acc[r][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int LA = 2, NS = LA + 1, NSTEP = NCH * 2 * PH;
    u32x4 rb[NS];
    auto rd = [&](auto sc) {
      constexpr int S = decltype(sc)::value, c = S / (2 * PH), pc = (S / PH) % 2, R = S % PH;
      rb[S % NS] = lds_rd128<c * PBUFB + R * PW * 64>(tb[pc]);
    };
    static_steps([&](auto sc) { rd(sc); }, std::make_integer_sequence<int, LA>{});
    static_steps([&](auto sc) {
      constexpr int S = decltype(sc)::value, c = S / (2 * PH), pc = (S / PH) % 2, R = S % PH;
      if constexpr (S + LA < NSTEP) rd(std::integral_constant<int, S + LA>{});
      constexpr int AHEAD = (NSTEP - 1 - S < LA ? NSTEP - 1 - S : LA);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD) : "memory");
      asm volatile("" : "+v"(rb[S % NS]));
      const uint4 fb = make_uint4(rb[S % NS].x, rb[S % NS].y, rb[S % NS].z, rb[S % NS].w);
      static_steps([&](auto ri) {
        constexpr int r = R - 1 + decltype(ri)::value;
        if constexpr (r >= 0 && r < TH) {
          constexpr int tap = 3 - (2 * (R - r) + pc);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) acc[r][tt] = mma16<bf16>(W[tap][c][tt], fb, acc[r][tt]);
        }
      }, std::make_integer_sequence<int, 2>{});
      __builtin_amdgcn_sched_barrier(0);
    }, std::make_integer_sequence<int, NSTEP>{});

    // ---- epilogue: act'(reference), rounding, the tile through LDS, whole output rows out ----
    char* stg = smem + 2 * BUFB;
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      float vv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vv[e] = acc[r][e >> 2][e & 3];
      uint4 pk;
      if constexpr (RACT == ACT_LRELU || ACC) {
        if constexpr (RACT != ACT_NONE) {
          float z[8];
          Elem<bf16>::unpack(rz[r], z);
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] *= act_grad(RACT, z[e]);
        }
        if constexpr (ACC) {
          float y0[8];
          Elem<bf16>::unpack(ry[r], y0);
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] += y0[e];
        }
        pk = Elem<bf16>::pack(vv);
      } else {
        pk = Elem<bf16>::pack(vv);
        if constexpr (RACT == ACT_RELU) {
          // relu'(reference) per 16-bit half: min(max(ref as int16, 0), 1) is 1 exactly for a positive bf16; times the output's bits
          auto mask = [](unsigned o, unsigned z) { return dc_pk_mul_lo_u16(o, dc_pk_min_u16(dc_pk_max_i16(z, 0u), 0x00010001u)); };
          pk.x = mask(pk.x, rz[r].x); pk.y = mask(pk.y, rz[r].y); pk.z = mask(pk.z, rz[r].z); pk.w = mask(pk.w, rz[r].w);
        }
      }
      if constexpr (CSUM) {
        float f[8];
        Elem<bf16>::unpack(pk, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += f[e];
      }
      // staging pixel = output column 2 fi + pw of row r; 16-byte slot s of column oc at slot s ^ ((oc >> 1) & 7)
      *reinterpret_cast<uint4*>(stg + (r * 32 + 2 * fi + pw) * 128 + (((4 * hh + fg) ^ (fi & 7)) << 4)) = pk;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // wave w stores output row 2 (q0 + w) + ph: 32 pixels x 128 bytes = four 1 KB instructions
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const int oc = j * 8 + (lane >> 3), sl = lane & 7;
      const uint4 o = *reinterpret_cast<const uint4*>(stg + (wave * 32 + oc) * 128 + ((sl ^ ((oc >> 1) & 7)) << 4));
      unsigned* yp = reinterpret_cast<unsigned*>(Yp + ((size_t)(n * a.Hof + 2 * (q0 + wave) + ph) * a.Wof + 2 * r0 + oc) * 64 + sl * 8);
      __builtin_nontemporal_store(o.x, yp); __builtin_nontemporal_store(o.y, yp + 1);
      __builtin_nontemporal_store(o.z, yp + 2); __builtin_nontemporal_store(o.w, yp + 3);
    }
  }
  if constexpr (CSUM) {
    // lanes fi = 0 .. 15 of a 16-lane row hold the same eight channels: fold them, then the two waves (column parities) of a channel half
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = cs[e];
      x += __shfl_xor(x, 1); x += __shfl_xor(x, 2); x += __shfl_xor(x, 4); x += __shfl_xor(x, 8);
      cs[e] = x;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every wave is past its last use of the staging buffer
    float* red = reinterpret_cast<float*>(smem);       // [wave][32]
    if (fi == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave * 32 + fg * 8 + e] = cs[e];
    }
    __syncthreads();
    if (tid < 64) {
      const int h2 = tid >> 5, cc = tid & 31;          // channel 32 h2 + cc: waves h2 (column parity 0) and 2 + h2 (parity 1)
      const float t0 = red[h2 * 32 + cc] + red[(2 + h2) * 32 + cc];
      double* row = a.colsum_part + (size_t)blockIdx.x * 2 * 64;
      row[tid] = (double)t0;
      row[64 + tid] = 0.0;
    }
  }
}

// conv_dc256_kernel: the FORWARD form for 256 input channels from TWO 128-channel tensors (the virtual concat [decoder | encoder skip] of
// merged2_decoder_2, the last wide decoder: pixrefer.py:243-270) with the batch statistics in the block.  K = 1024 per class: a wave keeps
// a 16 x 1024 weight slice (32 fragments, 128 registers), so the block is EIGHT waves - waves 0-3 column parity 0, waves 4-7 parity 1, each
// one 16-channel MFMA tile of the 64 outputs; the patch is (4 + 1) x (16 + 2) pixels x 8 chunks (48 KB, double-buffered: one block per CU);
// wave c fetches chunk c of the patch (chunks 0-3 from the first tensor, 4-7 from the second); 80 fragment reads for 128 MFMAs per wave
// and tile; statistics as conv_s2c64.hip: running per-lane sums of the rounded outputs, one partial row per block AND column parity.
template <bool STATS>
__global__ __launch_bounds__(512, 1) void conv_dc256_kernel(const IgemmArgs a, const int ntiles) {
  constexpr int NC8 = 8;                        // 64-byte chunks of the 256 input channels
  constexpr int BUF8 = NC8 * PBUFB;             // one patch buffer (48 KB)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fg = lane >> 4;
  const int pw = wave >> 2, q = wave & 3;       // column parity of this wave's class, its 16-channel tile of the 64 outputs

  const int G = gridDim.x;
  int v = blockIdx.x;
  if ((G & 7) == 0) v = (v & 7) * (G >> 3) + (v >> 3);
  const int ph = v & 1, bt = v >> 1, GT = G >> 1;
  const int cls = 2 * ph + pw;

  // weights of class cls: [tap][chunk][row][32 k], packed row 16 q + fi (odd 16-byte pieces un-swapped: conv_dc64_kernel)
  uint4 W[4][NC8];
  {
    const bf16* wp = reinterpret_cast<const bf16*>(a.Wp) + (size_t)cls * a.wp_rows * a.Kpad;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < NC8; ++c) {
        const uint4 w = *reinterpret_cast<const uint4*>(wp + ((size_t)(t * NC8 + c) * a.wp_rows + 16 * q + fi) * 32 + fg * 8);
        W[t][c] = (fg & 1) ? make_uint4(w.z, w.w, w.x, w.y) : w;
      }
  }
  // accumulator rows 4 fg .. + 3 of tile q of the 64-row block: channels 32 (q >> 1) + 8 fg + 4 (q & 1) + e
  const int c0 = 32 * (q >> 1) + 8 * fg + 4 * (q & 1);
  f32x4 bia = (f32x4){0.f, 0.f, 0.f, 0.f};        // (no bias in front of a batch-norm - it cancels; the plain op has one)
  if (a.bias) bia = (f32x4){a.bias[c0], a.bias[c0 + 1], a.bias[c0 + 2], a.bias[c0 + 3]};

  int tb0[2];
#pragma unroll
  for (int pc = 0; pc < 2; ++pc) {
    const int px = fi + pw + pc;
    tb0[pc] = (px << 6) + (((fg ^ (px >> 1)) & 3) << 4);
  }
  // patch DMA: wave c fetches chunk c, all six rounds of 16 patch pixels; two 128-channel tensors (the step) or one of 256 (the plain op)
  const bool two = a.x.C[1] != 0;
  const int pixb = (two ? 128 : 256) * (int)sizeof(bf16);        // bytes of a source pixel
  int ppy[NROUND], ppx[NROUND], prel[NROUND];
#pragma unroll
  for (int j = 0; j < NROUND; ++j) {
    const int pp = j * 16 + (lane >> 2);
    ppy[j] = pp < NPATCH ? pp / PW : 1 << 20;
    ppx[j] = pp % PW;
    prel[j] = pp < NPATCH ? (ppy[j] * a.Win + ppx[j]) * pixb + (((lane & 3) ^ ((ppx[j] >> 1) & 3)) * 8) * (int)sizeof(bf16) : 0;
  }
  const unsigned srcbytes = (unsigned)((size_t)a.N * a.Hin * a.Win * pixb);
  __amdgpu_buffer_rsrc_t rsX = make_rsrc((two && wave >= 4) ? a.x.ptr[1] : a.x.ptr[0], srcbytes);
  const unsigned csoff = (unsigned)((two ? (wave & 3) : wave) * 64);
  const int tiles_x = a.Wg / TW, tpi = tiles_x * (a.Hg / TH);
  bf16* Yp = reinterpret_cast<bf16*>(a.Y);
  constexpr int NST = 2;                        // 1 KB store instructions per tile and wave (4 rows x 4 KB / 8 waves)

  auto issue_patch = [&](int t, int buf) {
    const int n = t / tpi, rem = t - n * tpi;
    const int y0 = (rem / tiles_x) * TH + ph - 1, x0 = (rem % tiles_x) * TW - 1;
    const int base = ((n * a.Hin + y0) * a.Win + x0) * pixb;
    uint4* l0 = reinterpret_cast<uint4*>(smem + buf * BUF8 + wave * PBUFB);
#pragma unroll
    for (int j = 0; j < NROUND; ++j) {
      const int ih = y0 + ppy[j], iw = x0 + ppx[j];
      const bool ok = (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      dma16_buf(rsX, ok ? (unsigned)(base + prel[j]) : DMA_OOB, csoff, l0 + j * 64);
    }
  };

  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};

  if (bt < ntiles) issue_patch(bt, 0);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < NC8; ++c) asm volatile("" : "+v"(W[t][c].x), "+v"(W[t][c].y), "+v"(W[t][c].z), "+v"(W[t][c].w));
  int it = 0;
  for (int t = bt; t < ntiles; t += GT, ++it) {
    const int buf = it & 1;
    if (it == 0) wait_vm<0>();
    else wait_vm<NST>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + GT < ntiles) issue_patch(t + GT, buf ^ 1);
    int tb[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) tb[pc] = tb0[pc] + buf * BUF8;

    // ---- 80 fragment steps (chunk, pc, patch row R): one ds_read_b128 each, fed to tile rows R - 1 (pr = 1) and R (pr = 0): 128 MFMAs ----
    f32x4 acc[TH];
#pragma unroll
    for (int r = 0; r < TH; ++r) acc[r] = bia;
    constexpr int LA = 2, NS = LA + 1, NSTEP = NC8 * 2 * PH;
    u32x4 rb[NS];
    auto rd = [&](auto sc) {
      constexpr int S = decltype(sc)::value, c = S / (2 * PH), pc = (S / PH) % 2, R = S % PH;
      rb[S % NS] = lds_rd128<c * PBUFB + R * PW * 64>(tb[pc]);
    };
    static_steps([&](auto sc) { rd(sc); }, std::make_integer_sequence<int, LA>{});
    static_steps([&](auto sc) {
      constexpr int S = decltype(sc)::value, c = S / (2 * PH), pc = (S / PH) % 2, R = S % PH;
      if constexpr (S + LA < NSTEP) rd(std::integral_constant<int, S + LA>{});
      constexpr int AHEAD = (NSTEP - 1 - S < LA ? NSTEP - 1 - S : LA);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD) : "memory");
      asm volatile("" : "+v"(rb[S % NS]));
      const uint4 fb = make_uint4(rb[S % NS].x, rb[S % NS].y, rb[S % NS].z, rb[S % NS].w);
      static_steps([&](auto ri) {
        constexpr int r = R - 1 + decltype(ri)::value;
        if constexpr (r >= 0 && r < TH) {
          constexpr int tap = 3 - (2 * (R - r) + pc);
          acc[r] = mma16<bf16>(W[tap][c], fb, acc[r]);
        }
      }, std::make_integer_sequence<int, 2>{});
      __builtin_amdgcn_sched_barrier(0);
    }, std::make_integer_sequence<int, NSTEP>{});

    // ---- epilogue: rounding, statistics of the rounded values, the tile through LDS, whole output rows out ----
    char* stg = smem + 2 * BUF8;
    // this lane's 8 bytes of output column 2 fi + pw: 16-byte slot (c0 / 8) ^ (fi & 7) of the pixel's 128 bytes
    const int wslot = ((((c0 >> 3) ^ fi) & 7) << 4) + ((c0 & 4) << 1);
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      uint2 pk;
      pk.x = Elem<bf16>::pack2(acc[r][0], acc[r][1]);
      pk.y = Elem<bf16>::pack2(acc[r][2], acc[r][3]);
      if constexpr (STATS) {
        const float v0 = __uint_as_float(pk.x << 16), v1 = __uint_as_float(pk.x & 0xffff0000u);
        const float v2 = __uint_as_float(pk.y << 16), v3 = __uint_as_float(pk.y & 0xffff0000u);
        ssum[0] += v0; ssum[1] += v1; ssum[2] += v2; ssum[3] += v3;
        ssq[0] = fmaf(v0, v0, ssq[0]); ssq[1] = fmaf(v1, v1, ssq[1]); ssq[2] = fmaf(v2, v2, ssq[2]); ssq[3] = fmaf(v3, v3, ssq[3]);
      }
      *reinterpret_cast<uint2*>(stg + (r * 32 + 2 * fi + pw) * 128 + wslot) = pk;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // wave w stores half of output row 2 (q0 + (w >> 1)) + ph: columns 16 (w & 1) + 8 j + (lane >> 3), 16-byte slot lane & 7
    {
      const int n = t / tpi, rem = t - n * tpi;
      const int q0 = (rem / tiles_x) * TH, r0 = (rem % tiles_x) * TW;
      const int row = wave >> 1;
#pragma unroll
      for (int j = 0; j < NST; ++j) {
        const int oc = 16 * (wave & 1) + 8 * j + (lane >> 3), sl = lane & 7;
        // column oc = 2 fi' + pw' was written at slot s ^ (fi' & 7) = s ^ ((oc >> 1) & 7)
        const uint4 o = *reinterpret_cast<const uint4*>(stg + (row * 32 + oc) * 128 + ((sl ^ ((oc >> 1) & 7)) << 4));
        unsigned* yp = reinterpret_cast<unsigned*>(Yp + ((size_t)(n * a.Hof + 2 * (q0 + row) + ph) * a.Wof + 2 * r0 + oc) * 64 + sl * 8);
        __builtin_nontemporal_store(o.x, yp); __builtin_nontemporal_store(o.y, yp + 1);
        __builtin_nontemporal_store(o.z, yp + 2); __builtin_nontemporal_store(o.w, yp + 3);
      }
    }
  }
  if constexpr (STATS) {
    // one partial row per block and column parity (chunk 2 block + pw): the 16 lanes of a channel quad combine in a fixed order
    float s[4], qq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[e] = ssum[e]; qq[e] = ssq[e];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) { s[e] += __shfl_xor(s[e], m); qq[e] += __shfl_xor(qq[e], m); }
    }
    if (fi == 0) {
      double* rowp = a.bn_part + ((size_t)(2 * blockIdx.x + pw) * 2) * a.Cout + c0;
#pragma unroll
      for (int e = 0; e < 4; ++e) { rowp[e] = (double)s[e]; rowp[a.Cout + e] = (double)qq[e]; }
    }
  }
}

// a patch2-plan transposed convolution from TWO 128-channel tensors to 64 channels, raw bf16 output, optional batch statistics of ONE group
bool conv_dc256_eligible(const IgemmArgs& a, int is_bf16) {
  if (!dc64_knob() || !is_bf16 || a.patch != 2 || a.nclass != 4 || a.ntaps != 4 || a.os != 2) return false;
  if (a.Cout != 64 || a.CoutPad != 64 || a.ldY != 64 || a.Cin != 256 || !a.rowperm || a.splitk != 1) return false;
  if (!((a.x.C[0] == 128 && a.x.C[1] == 128) || (a.x.C[0] == 256 && a.x.C[1] == 0))) return false;
  if (a.Hg % TH || a.Wg % TW || a.Hin != a.Hg || a.Win != a.Wg || a.Hof != 2 * a.Hg || a.Wof != 2 * a.Wg) return false;
  if (a.out_act != ACT_NONE || a.y_f32 || a.ref || a.accumulate || a.split_c || a.pool_out || a.x.aff_a[0] || a.x.aff_a[1] || a.x.act != ACT_NONE) return false;
  for (int cls = 0; cls < 4; ++cls) {
    if (a.o0h[cls] != (cls >> 1) || a.o0w[cls] != (cls & 1)) return false;
    for (int t = 0; t < 4; ++t)
      if (a.taps[cls].dh[t] != (cls >> 1) - (t >> 1) || a.taps[cls].dw[t] != (cls & 1) - (t & 1)) return false;
  }
  if (a.N * (a.Hg / TH) * (a.Wg / TW) < 256) return false;       // (two tiles per block and row parity at least: the 128-register weight load)
  // batch statistics: only the one-group form this kernel writes (two partial rows per block); a caller that set up per-tile rows (more
  // than one batch-norm group: per-sample statistics of the inference plans) gets the patch kernel it chunked them for
  // (the caller states that layout itself: bn_tpg == 1 << 30 marks "one group, two partial rows per block" - a per-group table whose chunk
  // count merely happens to equal 2 * grid must not be taken for it: ADVICE r5)
  if (a.bn_part && (a.bn_tpg != (1 << 30) || a.bn_nchunk != 2 * conv_dc256_grid(a))) return false;
  return (size_t)a.N * a.Hin * a.Win * 256 * 2 < 0x70000000ull;
}
int conv_dc256_grid(const IgemmArgs& a) {
  const int ntiles = a.N * (a.Hg / TH) * (a.Wg / TW);          // per row parity
  return 2 * ntiles < 256 ? 2 * ntiles : 256;                  // one eight-wave block on each CU; even: both row parities
}
hipError_t launch_conv_dc256(const IgemmArgs& a, hipStream_t st) {
  const int ntiles = a.N * (a.Hg / TH) * (a.Wg / TW);
  const int grid = conv_dc256_grid(a);
  if (a.bn_part && (a.bn_tpg != (1 << 30) || a.bn_nchunk != 2 * grid)) return hipErrorInvalidValue;
  const int ki = a.bn_part ? 1 : 0;
  void (*kerns[2])(const IgemmArgs, const int) = {conv_dc256_kernel<false>, conv_dc256_kernel<true>};
  void (*kern)(const IgemmArgs, const int) = kerns[ki];
  const int smem = 2 * 8 * PBUFB + STGB;                       // 112 KB
  static bool attr_done[2] = {false, false};
  if (!attr_done[ki]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_done[ki] = true; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, a, ntiles);
  return hipGetLastError();
}

// a patch2-plan transposed convolution (conv_ops.h plan_make_patch2) from one 128-channel tensor to 64 channels, plain stores
bool conv_dc64_eligible(const IgemmArgs& a, int is_bf16) {
  if (!is_bf16 || a.patch != 2 || a.nclass != 4 || a.ntaps != 4 || a.os != 2) return false;
  if (a.Cout != 64 || a.CoutPad != 64 || a.ldY != 64 || a.Cin != 128 || a.x.C[0] != 128 || a.x.C[1] != 0 || !a.rowperm || a.splitk != 1) return false;
  if (a.Hg % TH || a.Wg % TW || a.Hin != a.Hg || a.Win != a.Wg || a.Hof != 2 * a.Hg || a.Wof != 2 * a.Wg) return false;
  if (a.bias || a.out_act != ACT_NONE || a.bn_part || a.y_f32 || a.ref_a || a.split_c || a.pool_out) return false;
  if (a.ref && a.ref_act != ACT_LRELU && a.ref_act != ACT_RELU) return false;
  for (int cls = 0; cls < 4; ++cls) {
    if (a.o0h[cls] != (cls >> 1) || a.o0w[cls] != (cls & 1)) return false;
    for (int t = 0; t < 4; ++t)
      if (a.taps[cls].dh[t] != (cls >> 1) - (t >> 1) || a.taps[cls].dw[t] != (cls & 1) - (t & 1)) return false;
  }
  return (size_t)a.N * a.Hin * a.Win * 128 * 2 < 0x70000000ull;
}

int conv_dc64_grid(const IgemmArgs& a) {
  const int ntiles = a.N * (a.Hg / TH) * (a.Wg / TW);          // per row parity
  return 2 * ntiles < 512 ? 2 * ntiles : 512;                  // two four-wave blocks on each of the 256 CUs; even: both row parities
}
hipError_t launch_conv_dc64(const IgemmArgs& a, hipStream_t st) {
  const int ntiles = a.N * (a.Hg / TH) * (a.Wg / TW);          // per row parity
  const int grid = conv_dc64_grid(a);
  const int ki = (!a.ref ? 0 : (a.ref_act == ACT_LRELU ? 1 : 2)) + (a.accumulate ? 3 : 0) + (a.colsum_part ? 6 : 0);
  void (*kerns[12])(const IgemmArgs, const int) = {conv_dc64_kernel<ACT_NONE, false>, conv_dc64_kernel<ACT_LRELU, false>, conv_dc64_kernel<ACT_RELU, false>,
                                                   conv_dc64_kernel<ACT_NONE, true>, conv_dc64_kernel<ACT_LRELU, true>, conv_dc64_kernel<ACT_RELU, true>,
                                                   conv_dc64_kernel<ACT_NONE, false, true>, conv_dc64_kernel<ACT_LRELU, false, true>, conv_dc64_kernel<ACT_RELU, false, true>,
                                                   conv_dc64_kernel<ACT_NONE, true, true>, conv_dc64_kernel<ACT_LRELU, true, true>, conv_dc64_kernel<ACT_RELU, true, true>};
  void (*kern)(const IgemmArgs, const int) = kerns[ki];
  const int smem = 2 * BUFB + STGB;                            // 64 KB
  static bool attr_done[12] = {};
  if (!attr_done[ki]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_done[ki] = true; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, a, ntiles);
  return hipGetLastError();
}

}  // namespace vp
