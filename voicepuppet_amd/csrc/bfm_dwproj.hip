// dwproj_kernel: the second half of an MfccNet inverted-residual block (tinynet.py:12-142) in ONE kernel - depthwise 7x3 convolution +
// folded batch-norm + ReLU6 of the 6x-expanded tensor, then the 1x1 projection (+ folded batch-norm, + the residual add in place).
//
// Unfused (plan_bfmnet.hip before round 5) the expanded tensor crossed HBM four times per block: written by the expansion GEMM, read and
// written by dwconv7x3_f32_kernel, read by the projection GEMM: 14.7 GB of the 64 x 1 s forward, 1.75 ms in the depthwise kernel alone.
// Here the depthwise result never exists in HBM: it is the projection GEMM's PIXEL OPERAND, made in LDS chunk by chunk.
//   * a block owns TR time rows x the whole mel width W of one clip (NPIX = TR W pixels, a multiple of 16) and ALL output channels
//     (64 RT); four waves, wave w the channels [16 RT w, 16 RT (w + 1)); accumulators RT x NPIX / 16 MFMA tiles per wave;
//   * the K loop runs over chunks of 16 expanded channels (one f32 MFMA K chunk: 4 x v_mfma_f32_16x16x4_f32).  Per chunk, by LDS-DMA
//     (double-buffered, counted with vmcnt): the expanded tensor's halo tile ((TR + 6) x (W + 2) pixels x 64 bytes; rows / columns
//     outside the image are zeros from the buffer descriptor - SAME padding costs no mask), the projection weights' chunk (64 RT rows x
//     64 bytes) and the depthwise taps + bias of the chunk (22 x 64 bytes);
//   * stencil: a thread owns (mel column, channel pair) over SR consecutive rows: (SR + 6) x 3 ds_read_b64, 21 SR packed FMAs, ReLU6,
//     ds_write_b64 into the pixel-operand tile [NPIX][16] (16-byte slot swizzle of conv_c64.hip: conflict-free ds_read_b128 fragments);
//   * MFMA: RT weight fragments + NPIX / 16 pixel fragments per wave and chunk, 4 RT NPIX / 16 MFMAs - the stencil's VALU work is a sixth
//     of the MFMA time and runs under the other resident block's MFMAs (two blocks per CU);
//   * epilogue: + bias, (+ y), 16-byte stores (a lane holds 4 consecutive channels of a pixel).
// No recomputation: the halo costs only extra reads of the expanded tensor (1.2-1.75x, mostly L2 hits between neighbouring tiles).
// float32 only (the parity path; the bf16 trunk keeps the unfused kernels).
#include "audio_args.h"
#include "igemm_device.h"
#include "patch_device.h"
#include "vp_common.h"

#ifndef DWPROJ_ABL
#define DWPROJ_ABL 0     // build-time ablations (make one FILE=bfm_dwproj VAR=abl1 DEFS=-DDWPROJ_ABL=1): 1 no stencil, 2 no MFMAs, 4 no DMA after the first chunk
#endif

namespace vp {

struct DwProjArgs {
  const float* ex;        // [B][H][W][Ce] expanded tensor (after ReLU6)
  const float* wdw;       // [22][Ce]: 21 folded depthwise taps (row 3 kh + kw) + the folded bias row
  const float* Wp;        // packed projection weights [Ce / 16][rows_pad][16] (PackDesc without row permutation)
  const float* bias;      // [cout]
  float* y;               // [B][H][W][cout]
  int B, H, Ce, cout, rows_pad, add;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void dp_lds_wr64(int addr, float2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <int W, int TR, int RT>
__global__ __launch_bounds__(256, 2) void dwproj_kernel(const DwProjArgs a) {
  constexpr int NPIX = TR * W, NPT = NPIX / 16;
  static_assert(NPIX % 16 == 0, "whole MFMA pixel tiles");
  constexpr int HR = TR + 6, HW = W + 2, NSLOT = HR * HW, NRND = (NSLOT + 15) / 16;
  constexpr int HALOB = NRND * 1024;            // halo tile: 64 bytes per pixel slot
  constexpr int COUT = 64 * RT, WPB = COUT * 64;
  constexpr int DWB = 2048;                     // depthwise taps + bias of a chunk: 22 x 64 bytes (two DMA rounds)
  constexpr int SETB = HALOB + WPB + DWB;       // one buffer set; two per block
  constexpr int BPB = NPIX * 64;                // pixel-operand tile (single: rewritten behind the chunk's top barrier)
  constexpr int JH = (NRND + 3) / 4;            // halo DMA rounds per wave
  constexpr int NSEG = 256 / (8 * W) > 0 ? 256 / (8 * W) : 1, SR = (TR + NSEG - 1) / NSEG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fg = lane >> 4;
  const int tiles_h = (a.H + TR - 1) / TR;
  const int b = blockIdx.x / tiles_h, h0 = (blockIdx.x - b * tiles_h) * TR;

  // ---- DMA lanes ----
  __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.ex, (unsigned)((size_t)a.B * a.H * W * a.Ce * sizeof(float)));
  __amdgpu_buffer_rsrc_t rsW = make_rsrc(a.Wp, (unsigned)((size_t)(a.Ce / 16) * a.rows_pad * 64));
  __amdgpu_buffer_rsrc_t rsD = make_rsrc(a.wdw, (unsigned)((size_t)22 * a.Ce * sizeof(float)));
  unsigned hvo[JH];                             // halo: round wave + 4 j, slot 16 round + (lane >> 2), piece lane & 3 (no swizzle: read as float2)
#pragma unroll
  for (int j = 0; j < JH; ++j) {
    const int pp = (wave + 4 * j) * 16 + (lane >> 2);
    const int row = pp / HW, col = pp - row * HW;
    const int ih = h0 - 3 + row, iw = col - 1;
    const bool ok = pp < NSLOT && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)W;
    hvo[j] = ok ? (unsigned)((((size_t)(b * a.H + ih) * W + iw) * a.Ce + (lane & 3) * 4) * sizeof(float)) : DMA_OOB;
  }
  unsigned wvo[RT];                             // weights: round wave + 4 j = rows 16 (wave + 4 j) .., slot lane & 3 holds piece slot ^ ((row >> 1) & 3)
#pragma unroll
  for (int j = 0; j < RT; ++j) {
    const int row = (wave + 4 * j) * 16 + (lane >> 2);
    wvo[j] = (unsigned)((row * 16 + (((lane & 3) ^ ((lane >> 3) & 3)) * 4)) * sizeof(float));
  }
  // depthwise taps + bias: waves 0 and 1, rows 16 wave + (lane >> 2) < 22
  const int drow = wave * 16 + (lane >> 2);
  const unsigned dvo = (wave < 2 && drow < 22) ? (unsigned)(((size_t)drow * a.Ce + (lane & 3) * 4) * sizeof(float)) : DMA_OOB;
  const unsigned wchunk = (unsigned)a.rows_pad * 64u;

  auto issue = [&](int c, int set) {
    char* base = smem + set * SETB;
#pragma unroll
    for (int j = 0; j < JH; ++j)
      if (wave + 4 * j < NRND) dma16_buf(rsX, hvo[j], (unsigned)(c * 64), reinterpret_cast<uint4*>(base) + (wave + 4 * j) * 64);
#pragma unroll
    for (int j = 0; j < RT; ++j) dma16_buf(rsW, wvo[j], (unsigned)c * wchunk, reinterpret_cast<uint4*>(base + HALOB) + (wave + 4 * j) * 64);
    if (wave < 2) dma16_buf(rsD, dvo, (unsigned)(c * 64), reinterpret_cast<uint4*>(base + HALOB + WPB) + wave * 64);
  };

  // ---- stencil items: (row segment, mel column, channel pair) - rows [SR seg, SR seg + SR) of column sw, pair sp; NITEM <= 256 but for W = 40 ----
  constexpr int NITEM = NSEG * 8 * W;
  // ---- fragment lane offsets ----
  const int fsw = ((fg ^ (fi >> 1)) & 3) << 4;
  const int aoff = (wave * 16 * RT + fi) * 64 + fsw;              // + rt * 1024
  const int boff = 2 * SETB + fi * 64 + fsw;                      // + pt * 1024

  f32x4 acc[RT][NPT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) acc[rt][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunk = a.Ce / 16;
  issue(0, 0);
  for (int c = 0; c < nchunk; ++c) {
    const int set = c & 1;
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (c + 1 < nchunk && !(DWPROJ_ABL & 4)) issue(c + 1, set ^ 1);

    // ---- depthwise 7x3 + bias + ReLU6 of this chunk into the pixel-operand tile ----
    for (int item = tid; item < NITEM && !(DWPROJ_ABL & 1); item += 256) {
      const int sseg = item / (8 * W), srem = item - sseg * (8 * W);
      const int sw = srem >> 3, sp = srem & 7, sr0 = sseg * SR;
      const int hb = set * SETB + ((sr0 * HW + sw) * 16 + 2 * sp) * 4;       // halo byte offset of (row sr0, slot sw = image column sw - 1), this pair
      const int db = set * SETB + HALOB + WPB + 8 * sp;
      // all LDS reads of the item first (22 tap / bias pairs, 3 (SR + 6) input pairs), ONE wait, then only packed arithmetic
      u32x2 wr[22], xr[SR + 6][3];
      static_steps([&](auto ki) { constexpr int k = decltype(ki)::value; wr[k] = lds_rd64<k * 64>(db); }, std::make_integer_sequence<int, 22>{});
      static_steps([&](auto ji) {
        constexpr int j = decltype(ji)::value / 3, kw = decltype(ji)::value % 3;         // input row sr0 + j of the halo tile, column sw + kw
        xr[j][kw] = lds_rd64<(j * HW + kw) * 64>(hb);
      }, std::make_integer_sequence<int, 3 * (SR + 6)>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 22; ++k) asm volatile("" : "+v"(wr[k]));
#pragma unroll
      for (int j = 0; j < SR + 6; ++j)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) asm volatile("" : "+v"(xr[j][kw]));
      auto f2 = [](u32x2 u) { return (f32x2){__uint_as_float(u.x), __uint_as_float(u.y)}; };
      f32x2 o[SR];
#pragma unroll
      for (int r = 0; r < SR; ++r) {
        o[r] = f2(wr[21]);
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) o[r] = __builtin_elementwise_fma(f2(wr[3 * kh + kw]), f2(xr[r + kh][kw]), o[r]);      // v_pk_fma_f32
      }
#pragma unroll
      for (int r = 0; r < SR; ++r) {
        if (sr0 + r < TR) {
          const int px = (sr0 + r) * W + sw;
          const float2 v = make_float2(fminf(fmaxf(o[r].x, 0.f), 6.f), fminf(fmaxf(o[r].y, 0.f), 6.f));
          dp_lds_wr64(2 * SETB + px * 64 + ((((sp >> 1) ^ (px >> 1)) & 3) << 4) + (sp & 1) * 8, v);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- projection MFMAs of the chunk ----
    if (!(DWPROJ_ABL & 2)) {
      const int ab = set * SETB + HALOB + aoff;
      u32x4 ra[RT], rb[NPT];
      static_steps([&](auto ti) { constexpr int t = decltype(ti)::value; ra[t] = lds_rd128<t * 1024>(ab); }, std::make_integer_sequence<int, RT>{});
      static_steps([&](auto ti) { constexpr int t = decltype(ti)::value; rb[t] = lds_rd128<t * 1024>(boff); }, std::make_integer_sequence<int, NPT>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < RT; ++t) asm volatile("" : "+v"(ra[t]));
#pragma unroll
      for (int t = 0; t < NPT; ++t) asm volatile("" : "+v"(rb[t]));
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const uint4 fa = make_uint4(ra[rt].x, ra[rt].y, ra[rt].z, ra[rt].w);
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) acc[rt][pt] = mma16<float>(fa, make_uint4(rb[pt].x, rb[pt].y, rb[pt].z, rb[pt].w), acc[rt][pt]);
      }
    }
  }

  // ---- epilogue: a lane holds channels 16 RT wave + 16 rt + 4 fg .. + 3 of pixel 16 pt + fi ----
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int ch = wave * 16 * RT + 16 * rt + 4 * fg;
    const float4 bv = *reinterpret_cast<const float4*>(a.bias + ch);
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
      const int px = 16 * pt + fi, r = px / W, w = px - r * W;
      if (h0 + r >= a.H) continue;
      float* yp = a.y + (((size_t)(b * a.H + h0 + r)) * W + w) * a.cout + ch;
      float4 v = make_float4(acc[rt][pt][0] + bv.x, acc[rt][pt][1] + bv.y, acc[rt][pt][2] + bv.z, acc[rt][pt][3] + bv.w);
      if (a.add) {
        const float4 y0 = *reinterpret_cast<const float4*>(yp);
        v.x += y0.x; v.y += y0.y; v.z += y0.z; v.w += y0.w;
      }
      *reinterpret_cast<float4*>(yp) = v;
    }
  }
}

namespace {
template <int W, int TR, int RT> hipError_t launch_dwproj_t(const DwProjArgs& a, hipStream_t st) {
  constexpr int NPIX = TR * W, HR = TR + 6, HW = W + 2, NRND = (HR * HW + 15) / 16;
  constexpr int smem = 2 * (NRND * 1024 + 64 * RT * 64 + 2048) + NPIX * 64;
  static bool attr_done = false;
  void (*kern)(const DwProjArgs) = dwproj_kernel<W, TR, RT>;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_done = true; }
  const int tiles_h = (a.H + TR - 1) / TR;
  hipLaunchKernelGGL(kern, dim3(a.B * tiles_h), dim3(256), smem, st, a);
  return hipGetLastError();
}
}  // namespace

// mel widths / channel counts of MfccNet's blocks (build_model in plan_bfmnet.hip): W 40, 20, 10, 5, 3; cout 64 .. 256
bool dwproj_eligible(int W, int Ce, int cout) {
  if (Ce % 16 || Ce < 16) return false;
  return (W == 40 && cout == 64) || (W == 20 && (cout == 64 || cout == 128)) || (W == 10 && (cout == 128 || cout == 192)) || (W == 5 && (cout == 192 || cout == 256)) ||
         (W == 3 && cout == 256);
}

hipError_t launch_dwproj(const float* ex, const float* wdw22, const float* Wp, int rows_pad, const float* bias, float* y, int add, int B, int H, int W,
                         int Ce, int cout, hipStream_t st) {
  if (!dwproj_eligible(W, Ce, cout)) return hipErrorInvalidValue;
  if ((size_t)B * H * W * Ce * sizeof(float) >= 0xF0000000ull) return hipErrorInvalidValue;       // lane offsets of the halo DMA
  DwProjArgs a;
  a.ex = ex; a.wdw = wdw22; a.Wp = Wp; a.bias = bias; a.y = y; a.B = B; a.H = H; a.Ce = Ce; a.cout = cout; a.rows_pad = rows_pad; a.add = add;
  if (W == 40) return launch_dwproj_t<40, 2, 1>(a, st);
  if (W == 20) return cout == 64 ? launch_dwproj_t<20, 4, 1>(a, st) : launch_dwproj_t<20, 4, 2>(a, st);
  if (W == 10) return cout == 128 ? launch_dwproj_t<10, 8, 2>(a, st) : launch_dwproj_t<10, 8, 3>(a, st);
  if (W == 5) return cout == 192 ? launch_dwproj_t<5, 16, 3>(a, st) : launch_dwproj_t<5, 16, 4>(a, st);
  return launch_dwproj_t<3, 16, 4>(a, st);
}

}  // namespace vp
