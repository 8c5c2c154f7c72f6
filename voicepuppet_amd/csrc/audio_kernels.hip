// Audio front-end kernels (f32): STFT framing, |DFT| -> mel -> log, and the non-GEMM pieces of MfccNet /
// BFMNet inference (first 9x5 conv, depthwise 7x3, SAME max-pools, batch-norm folding, GRU).
// The GEMM-shaped parts (DFT as a 512x514 matrix product, every 1x1 conv and dense layer) run on the
// f32-MFMA implicit-GEMM kernel of conv_kernels.hip.
// Reference: generator/generator.py:60-80, voicepuppet/bfmnet/tinynet.py:7-212, bfmnet.py:20-122.
#include <stdlib.h>

#include "audio_args.h"
#include "gru_device.h"
#include "vp_common.h"

namespace vp {

// frames[b*F + f][n] = pcm[b][f*hop + n] * hann_periodic(n)      (tf.signal.stft framing, no padding)
__global__ __launch_bounds__(256) void frame_window_kernel(const float* __restrict__ pcm, const float* __restrict__ window,
                                                           float* __restrict__ frames, int B, int L, int F, int win, int hop) {
  const size_t total = (size_t)B * F * win;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int n = (int)(i % win);
    const size_t t = i / win;
    const int f = (int)(t % F), b = (int)(t / F);
    frames[i] = pcm[(size_t)b * L + (size_t)f * hop + n] * window[n];
  }
}

// one block per frame: |re + i*im| for the nb spectrogram bins, dot with the mel matrix, log(. + 1e-6)
__global__ __launch_bounds__(128) void mag_mel_log_kernel(const float* __restrict__ spec, int ld, int nb,
                                                          const float* __restrict__ mel, int nmel, float* __restrict__ out) {
  extern __shared__ float mag[];
  const float* s = spec + (size_t)blockIdx.x * ld;
  for (int k = threadIdx.x; k < nb; k += 128) {
    const float re = s[k], im = s[nb + k];
    mag[k] = sqrtf(re * re + im * im);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < nmel; j += 128) {
    float acc = 0.f;
    for (int k = 0; k < nb; ++k) acc = fmaf(mag[k], mel[(size_t)k * nmel + j], acc);
    out[(size_t)blockIdx.x * nmel + j] = logf(acc + 1e-6f);
  }
}

// ------------------------------------------------------------------------------------------------
// log-mel in ONE launch (DataGenerator.extract_mfcc, generator/generator.py:60-80; round 5) for the reference's frame length 512:
// framing, periodic Hann window, the 512-point real DFT as a 256-point complex FFT (z[n] = x[2n] + i x[2n+1]; Stockham radix 4,
// four stages, one wave per frame: a lane is one radix-4 butterfly per stage, stages exchanged through 2 KB of wave-private LDS) and
// the untangling step X[k] = (Z[k] + conj Z[256-k]) / 2 + W512^k (Z[k] - conj Z[256-k]) / 2i, |X|, the 257 x nmel mel matrix and
// log(. + 1e-6).  24 KFLOP per frame instead of the 526 KFLOP of the dense DFT matrix product it replaces (three launches, two
// round trips of the frames / spectra through HBM).  Twiddles come from tables computed in float64 on the host.
//   block = FPB frames (consecutive over the whole batch), 5 waves; wave w transforms frames w, w + 5, ...; then thread (mel bin m,
//   frame group g) accumulates its mel bin for FPB / 4 frames over the 257 magnitudes (matrix row k read once per block from L2, the
//   magnitudes broadcast from LDS).
// ------------------------------------------------------------------------------------------------
constexpr int LM_FPB = 32, LM_NB = 257, LM_MAGP = 260;
__global__ __launch_bounds__(320) void logmel512_kernel(const float* __restrict__ pcm, const float* __restrict__ window, const float2* __restrict__ w256,
                                                         const float2* __restrict__ w512, const float* __restrict__ mel, float* __restrict__ out,
                                                         int L, int F, int hop, int nmel, int nframes) {
  __shared__ float2 s_w256[256];
  __shared__ float2 s_w512[LM_NB];
  __shared__ float s_win[512];
  __shared__ float2 s_fft[5][256];
  __shared__ float s_mag[LM_FPB][LM_MAGP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 256; i += 320) s_w256[i] = w256[i];
  for (int i = tid; i < LM_NB; i += 320) s_w512[i] = w512[i];
  for (int i = tid; i < 512; i += 320) s_win[i] = window[i];
  __syncthreads();
  const int f0 = blockIdx.x * LM_FPB;
  auto cmul = [](float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); };
  float2* buf = s_fft[wave];
  for (int fl = wave; fl < LM_FPB; fl += 5) {
    const int f = f0 + fl;
    if (f >= nframes) break;                                  // (wave-uniform)
    const float* x = pcm + (size_t)(f / F) * L + (size_t)(f % F) * hop;
    float2 u[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n2 = 2 * (lane + 64 * t);
      u[t] = make_float2(x[n2] * s_win[n2], x[n2 + 1] * s_win[n2 + 1]);
    }
    // four radix-4 Stockham stages: p = 1, 4, 16, 64; lane i reads elements i + 64 t, writes j + p t with j = 4 (i - k) + k, k = i mod p
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int p = 1 << (2 * st);
      const int k = lane & (p - 1), j = ((lane - k) << 2) + k;
      if (st > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) u[t] = buf[lane + 64 * t];
        const int q = k * (64 >> (2 * st));                    // W_{4p}^k = W256^(k * 64 / p)
        u[1] = cmul(u[1], s_w256[q]); u[2] = cmul(u[2], s_w256[2 * q]); u[3] = cmul(u[3], s_w256[3 * q]);
      }
      const float2 v0 = make_float2(u[0].x + u[2].x, u[0].y + u[2].y), v1 = make_float2(u[0].x - u[2].x, u[0].y - u[2].y);
      const float2 v2 = make_float2(u[1].x + u[3].x, u[1].y + u[3].y), d = make_float2(u[1].x - u[3].x, u[1].y - u[3].y);
      const float2 v3 = make_float2(d.y, -d.x);                 // (u1 - u3) * (-i)
      __builtin_amdgcn_wave_barrier();                          // (all lanes have read the previous stage)
      buf[j] = make_float2(v0.x + v2.x, v0.y + v2.y);
      buf[j + p] = make_float2(v1.x + v3.x, v1.y + v3.y);
      buf[j + 2 * p] = make_float2(v0.x - v2.x, v0.y - v2.y);
      buf[j + 3 * p] = make_float2(v1.x - v3.x, v1.y - v3.y);
      __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): the wave's own LDS writes have landed
      __builtin_amdgcn_wave_barrier();
    }
    // untangle: bins k = lane + 64 t (t = 0..3) and bin 256 (lane 0)
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int k = t < 4 ? lane + 64 * t : 256;
      if (t == 4 && lane != 0) break;
      const float2 a = buf[k & 255], b0 = buf[(256 - k) & 255];
      const float2 b = make_float2(b0.x, -b0.y);
      const float2 xe = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
      const float2 dd = make_float2(a.x - b.x, a.y - b.y);
      const float2 xo = make_float2(0.5f * dd.y, -0.5f * dd.x);                    // (a - b) / 2i
      const float2 w = cmul(s_w512[k], xo);
      const float re = xe.x + w.x, im = xe.y + w.y;
      s_mag[fl][k] = sqrtf(re * re + im * im);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // mel + log: thread -> (mel bin m = tid % 80, frame group g = tid / 80), 8 frames each
  const int m = tid % 80, g = tid / 80;
  if (m < nmel) {
    constexpr int FG = LM_FPB / 4;
    float acc[FG];
#pragma unroll
    for (int i = 0; i < FG; ++i) acc[i] = 0.f;
    for (int k = 0; k < LM_NB; ++k) {
      const float w = mel[(size_t)k * nmel + m];
#pragma unroll
      for (int i = 0; i < FG; ++i) acc[i] = fmaf(s_mag[g * FG + i][k], w, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < FG; ++i) {
      const int f = f0 + g * FG + i;
      if (f < nframes) out[(size_t)f * nmel + m] = logf(acc[i] + 1e-6f);
    }
  }
}

// four consecutive channels of an activation tensor stored as float or bf16 (the bf16 trunk of vp_bfmnet: f32 arithmetic, bf16 storage)
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<bf16>(const bf16* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16>(bf16* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(Elem<bf16>::pack2(v.x, v.y), Elem<bf16>::pack2(v.z, v.w));
}
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16>(bf16* p, float v) { *reinterpret_cast<unsigned short*>(p) = (unsigned short)f32_to_bf16_bits(v); }

// y = relu(conv9x5 stride (1,2) SAME (x[B,H,W,1]) * folded_scale + folded_bias)   (tinynet.py:168)
template <typename T>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w /*[45][Cout]*/,
                                                         const float* __restrict__ bias, T* __restrict__ y,
                                                         int B, int H, int W, int Wo, int Cout, int pt, int pl) {
  extern __shared__ float sw[];
  for (int i = threadIdx.x; i < 45 * Cout; i += 256) sw[i] = w[i];
  __syncthreads();
  const size_t total = (size_t)B * H * Wo * Cout;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int co = (int)(i % Cout);
    size_t t = i / Cout;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % H);
    const int b = (int)(t / H);
    float acc = bias[co];
    for (int kh = 0; kh < 9; ++kh) {
      const int ih = oh + kh - pt;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int kw = 0; kw < 5; ++kw) {
        const int iw = ow * 2 + kw - pl;
        if ((unsigned)iw >= (unsigned)W) continue;
        acc = fmaf(x[((size_t)b * H + ih) * W + iw], sw[(kh * 5 + kw) * Cout + co], acc);
      }
    }
    st1<T>(y + i, fmaxf(acc, 0.f));
  }
}

// depthwise 7x3 stride 1 SAME + folded BN + relu6     (tinynet.py:84-103)
// A thread owns 4 channels of one image column and walks a segment of HS output rows DOWN the column: the 21 taps live in registers,
// every input row of the segment (+ 3 halo rows either side) is loaded once (3 pixels: w-1, w, w+1) and scattered into a rotating
// window of 7 output accumulators (output q completes when input row q + 6 of the segment has been folded in, is stored and its
// slot restarts from the bias).  The row loop is unrolled by 7 so that slot and ring indices are compile-time.  Loads are
// branch-free (out-of-image rows / columns read a clamped, valid address and are multiplied by 0) and run PF rows ahead of the
// FMAs through a register ring.  History: round 1 used 8-row strips and tested every pixel with a branch - the compiler drained
// vmcnt to 0 after each load (one 1 KB load in flight per wave, 0.27 of HBM; r02 ISA: 64 loads, 43 s_waitcnt vmcnt(0)) and the 6
// halo rows of every 8-row strip were re-read (1.75x).  Per output the products are accumulated bias first, then kh = 0..6,
// kw = 0..2 (the same sums as a per-pixel loop; padding contributes exact zeros).
template <typename T>
__global__ __launch_bounds__(256) void dwconv7x3_kernel(const T* __restrict__ x, const float* __restrict__ w /*[21][C]*/,
                                                        const float* __restrict__ bias, T* __restrict__ y, int B, int H, int W, int C, int HS,
                                                        int nseg, int rev) {
  constexpr int PF = 3;
  const int cq = C >> 2;
  const size_t total = (size_t)B * nseg * W * cq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % cq) * 4;
    size_t t = i / cq;
    const int ow = (int)(t % W); t /= W;
    const int h0 = (int)(t % nseg) * HS;
    const int b = (int)(t / nseg);
    const int hs = h0 + HS <= H ? HS : H - h0;         // output rows of this segment
    float4 wv[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) wv[k] = *reinterpret_cast<const float4*>(w + (size_t)(rev ? 20 - k : k) * C + c4);   // rev: backward-data = the taps reversed
    // bias == nullptr: the raw convolution (training forward / backward-data, bfm_train.hip): no bias, no relu6
    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) acc[r] = bv;
    const bool okl = ow > 0, okr = ow + 1 < W;
    const T* xb = x + ((size_t)b * H) * W * C + c4;
    T* yb = y + (((size_t)b * H + h0) * W + ow) * C + c4;
    const size_t ol = (size_t)(okl ? ow - 1 : ow) * C, oc = (size_t)ow * C, orr = (size_t)(okr ? ow + 1 : ow) * C;
    const size_t rowstride = (size_t)W * C;
    float4 ring[7][3];
    auto loadrow = [&](int rel, float4 (&r)[3]) {      // input row h0 - 3 + rel
      int ih = h0 - 3 + rel;
      ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
      const T* rp = xb + (size_t)ih * rowstride;
      r[0] = ld4<T>(rp + ol); r[1] = ld4<T>(rp + oc); r[2] = ld4<T>(rp + orr);
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) loadrow(j, ring[j]);
    const int nrows = hs + 6;
    for (int k7 = 0; k7 < nrows; k7 += 7) {
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const int rel = k7 + u;
        loadrow(rel + PF, ring[(u + PF) % 7]);
        const int ih = h0 - 3 + rel;
        const bool okh = (unsigned)ih < (unsigned)H;
        const float ml = (okh && okl) ? 1.f : 0.f, mc = okh ? 1.f : 0.f, mr = (okh && okr) ? 1.f : 0.f;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          float4 xv = ring[u][kw];
          const float m = kw == 0 ? ml : (kw == 1 ? mc : mr);
          xv.x *= m; xv.y *= m; xv.z *= m; xv.w *= m;
#pragma unroll
          for (int kh = 0; kh < 7; ++kh) {             // output q = rel - kh lives in slot (u - kh) mod 7
            const float4 ww = wv[kh * 3 + kw];
            float4& a = acc[(u - kh + 7) % 7];
            a.x = fmaf(xv.x, ww.x, a.x); a.y = fmaf(xv.y, ww.y, a.y); a.z = fmaf(xv.z, ww.z, a.z); a.w = fmaf(xv.w, ww.w, a.w);
          }
        }
        // output q = rel - 6 is complete (slot (u + 1) mod 7): store it and restart the slot
        const int q = rel - 6;
        float4 a = acc[(u + 1) % 7];
        acc[(u + 1) % 7] = bv;
        if (q >= 0 && q < hs) {
          if (bias) {
            a.x = fminf(fmaxf(a.x, 0.f), 6.f); a.y = fminf(fmaxf(a.y, 0.f), 6.f);
            a.z = fminf(fmaxf(a.z, 0.f), 6.f); a.w = fminf(fmaxf(a.w, 0.f), 6.f);
          }
          st4<T>(yb + (size_t)q * rowstride, a);
        }
      }
    }
  }
}

// float32 form: a thread owns TWO adjacent output columns x TWO channels.  The float4 kernel above is latency-bound at 247 registers
// (two waves per SIMD) and loads every input element three times (left / centre / right column of three different threads); here the
// four input columns of a column pair serve two outputs (2x instead of 3x) and 2-channel vectors halve the register file per thread
// (four waves per SIMD): about twice the algorithmic bytes in flight per CU.  Lanes are consecutive channel pairs: 512-byte rows.
__global__ __launch_bounds__(256) void dwconv7x3_f32_kernel(const float* __restrict__ x, const float* __restrict__ w /*[21][C]*/,
                                                            const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W, int C, int HS,
                                                            int nseg, int rev) {
  constexpr int PF = 3;
  const int c2n = C >> 1, WP = (W + 1) >> 1;
  const size_t total = (size_t)B * nseg * WP * c2n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % c2n) * 2;
    size_t t = i / c2n;
    const int ow = (int)(t % WP) * 2; t /= WP;
    const int h0 = (int)(t % nseg) * HS;
    const int b = (int)(t / nseg);
    const int hs = h0 + HS <= H ? HS : H - h0;         // output rows of this segment
    float2 wv[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) wv[k] = *reinterpret_cast<const float2*>(w + (size_t)(rev ? 20 - k : k) * C + c);
    const float2 bv = bias ? *reinterpret_cast<const float2*>(bias + c) : make_float2(0.f, 0.f);
    float2 acc0[7], acc1[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) { acc0[r] = bv; acc1[r] = bv; }
    const bool okl = ow > 0, ok1 = ow + 1 < W, okr = ow + 2 < W;
    const float* xb = x + ((size_t)b * H) * W * C + c;
    float* yb = y + (((size_t)b * H + h0) * W + ow) * C + c;
    const size_t o0 = (size_t)(okl ? ow - 1 : ow) * C, o1 = (size_t)ow * C, o2 = (size_t)(ok1 ? ow + 1 : ow) * C, o3 = (size_t)(okr ? ow + 2 : ow) * C;
    const size_t rowstride = (size_t)W * C;
    float2 ring[7][4];
    auto loadrow = [&](int rel, float2 (&r)[4]) {      // input row h0 - 3 + rel (clamped: the mask below zeroes what lies outside)
      int ih = h0 - 3 + rel;
      ih = ih < 0 ? 0 : (ih >= H ? H - 1 : ih);
      const float* rp = xb + (size_t)ih * rowstride;
      r[0] = *reinterpret_cast<const float2*>(rp + o0); r[1] = *reinterpret_cast<const float2*>(rp + o1);
      r[2] = *reinterpret_cast<const float2*>(rp + o2); r[3] = *reinterpret_cast<const float2*>(rp + o3);
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) loadrow(j, ring[j]);
    const int nrows = hs + 6;
    for (int k7 = 0; k7 < nrows; k7 += 7) {
#pragma unroll
      for (int u = 0; u < 7; ++u) {
        const int rel = k7 + u;
        loadrow(rel + PF, ring[(u + PF) % 7]);
        const int ih = h0 - 3 + rel;
        const bool okh = (unsigned)ih < (unsigned)H;
        const float m[4] = {(okh && okl) ? 1.f : 0.f, okh ? 1.f : 0.f, (okh && ok1) ? 1.f : 0.f, (okh && okr) ? 1.f : 0.f};
        float2 xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { xv[k] = ring[u][k]; xv[k].x *= m[k]; xv[k].y *= m[k]; }
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {               // output q = rel - kh lives in slot (u - kh) mod 7
          float2& a0 = acc0[(u - kh + 7) % 7];
          float2& a1 = acc1[(u - kh + 7) % 7];
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float2 ww = wv[kh * 3 + kw];
            a0.x = fmaf(xv[kw].x, ww.x, a0.x); a0.y = fmaf(xv[kw].y, ww.y, a0.y);
            a1.x = fmaf(xv[kw + 1].x, ww.x, a1.x); a1.y = fmaf(xv[kw + 1].y, ww.y, a1.y);
          }
        }
        // output q = rel - 6 is complete (slot (u + 1) mod 7): store it and restart the slot
        const int q = rel - 6;
        float2 a0 = acc0[(u + 1) % 7], a1 = acc1[(u + 1) % 7];
        acc0[(u + 1) % 7] = bv; acc1[(u + 1) % 7] = bv;
        if (q >= 0 && q < hs) {
          if (bias) {
            a0.x = fminf(fmaxf(a0.x, 0.f), 6.f); a0.y = fminf(fmaxf(a0.y, 0.f), 6.f);
            a1.x = fminf(fmaxf(a1.x, 0.f), 6.f); a1.y = fminf(fmaxf(a1.y, 0.f), 6.f);
          }
          *reinterpret_cast<float2*>(yb + (size_t)q * rowstride) = a0;
          if (ok1) *reinterpret_cast<float2*>(yb + (size_t)q * rowstride + C) = a1;
        }
      }
    }
  }
}

// max-pool kxk, stride s, TF 'SAME' (padding never wins), 4 channels per thread
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void maxpool_same_kernel(const TI* __restrict__ x, TO* __restrict__ y, int B, int H, int W, int C,
                                                           int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo) {
  const int cq = C >> 2;
  const size_t total = (size_t)B * Ho * Wo * cq;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % cq) * 4;
    size_t t = i / cq;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int i2 = 0; i2 < kh; ++i2) {
      const int ih = oh * sh + i2 - pt;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int j2 = 0; j2 < kw; ++j2) {
        const int iw = ow * sw + j2 - pl;
        if ((unsigned)iw >= (unsigned)W) continue;
        const float4 v = ld4<TI>(x + (((size_t)b * H + ih) * W + iw) * C + c4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    st4<TO>(y + i * 4, m);
  }
}

// f32 -> bf16 copy of a narrow trunk tensor (the operand of the next expansion conv; the residual stream itself stays f32)
__global__ __launch_bounds__(256) void cvt_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st4<bf16>(y + i * 4, ld4<float>(x + i * 4));
}

// inference batch-norm folding (contrib batch_norm: no gamma, eps 1e-3):
//   w'[..., c] = w[..., c] / sqrt(var[c] + eps),  b'[c] = beta[c] - mean[c] / sqrt(var[c] + eps)
__global__ __launch_bounds__(256) void fold_bn_kernel(const float* __restrict__ w, const float* __restrict__ beta, const float* __restrict__ mean,
                                                      const float* __restrict__ var, float eps, size_t n, int C, float* __restrict__ wf,
                                                      float* __restrict__ bf) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float s = 1.f / sqrtf(var[c] + eps);
    wf[i] = w[i] * s;
    if (i < (size_t)C) bf[c] = beta[c] - mean[c] * s;
  }
}

// tf.contrib.rnn.GRUCell over T steps (dynamic_rnn semantics): gru_device.h (gru_fwd_kernel<false>), one 1024-thread block per sequence

// x *= m, element-wise (the opt-in decoder dropout masks of BFMNet inference: 0 or 1 / keep_prob)
__global__ void mul_inplace_kernel(float* __restrict__ x, const float* __restrict__ m, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= m[i];
}

// out[b,t,16:20] += ears[b,t] * {-2,-2,-2,-4}     (bfmnet.py:117,209)
__global__ void add_ears_kernel(float* __restrict__ out, const float* __restrict__ ears, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float e = ears[i];
  float* o = out + (size_t)i * 64 + 16;
  o[0] += -2.f * e; o[1] += -2.f * e; o[2] += -2.f * e; o[3] += -4.f * e;
}

static inline int nblk(size_t work, int cap = 4096) {
  size_t b = (work + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b > (size_t)cap ? cap : b);
}

hipError_t launch_frame_window(const float* pcm, const float* window, float* frames, int B, int L, int F, int win, int hop, hipStream_t st) {
  hipLaunchKernelGGL(frame_window_kernel, dim3(nblk((size_t)B * F * win)), dim3(256), 0, st, pcm, window, frames, B, L, F, win, hop);
  return hipGetLastError();
}
hipError_t launch_logmel512(const float* pcm, const float* window, const float* w256, const float* w512, const float* mel, float* out, int B, int L, int F, int hop,
                            int nmel, hipStream_t st) {
  const int nframes = B * F;
  hipLaunchKernelGGL(logmel512_kernel, dim3((nframes + LM_FPB - 1) / LM_FPB), dim3(320), 0, st, pcm, window, reinterpret_cast<const float2*>(w256),
                     reinterpret_cast<const float2*>(w512), mel, out, L, F, hop, nmel, nframes);
  return hipGetLastError();
}
hipError_t launch_mag_mel_log(const float* spec, int ld, int nb, const float* mel, int nmel, float* out, int nframes, hipStream_t st) {
  hipLaunchKernelGGL(mag_mel_log_kernel, dim3(nframes), dim3(128), nb * sizeof(float), st, spec, ld, nb, mel, nmel, out);
  return hipGetLastError();
}
// out_bf16 / in_bf16: storage type of the activation tensors (bf16 trunk of vp_bfmnet); arithmetic is f32 either way
hipError_t launch_conv_first(const float* x, const float* w, const float* bias, void* y, int out_bf16, int B, int H, int W, int Wo, int Cout, int pt, int pl, hipStream_t st) {
  const dim3 grid(nblk((size_t)B * H * Wo * Cout));
  if (out_bf16) hipLaunchKernelGGL((conv_first_kernel<bf16>), grid, dim3(256), 45 * Cout * sizeof(float), st, x, w, bias, (bf16*)y, B, H, W, Wo, Cout, pt, pl);
  else hipLaunchKernelGGL((conv_first_kernel<float>), grid, dim3(256), 45 * Cout * sizeof(float), st, x, w, bias, (float*)y, B, H, W, Wo, Cout, pt, pl);
  return hipGetLastError();
}
hipError_t launch_dwconv7x3(const void* x, const float* w, const float* bias, void* y, int is_bf16, int B, int H, int W, int C, hipStream_t st, int rev) {
  // row segments per column: enough threads to fill the chip (>= ~128k), at least 8 rows each; a segment of HS rows reads HS + 6
  const size_t cols = (size_t)B * W * (C / 4);
  int nseg = (int)((131072 + cols - 1) / cols);
  const int most = H / 8 > 0 ? H / 8 : 1;
  if (nseg > most) nseg = most;
  if (nseg < 1) nseg = 1;
  const int hs = (H + nseg - 1) / nseg;
  nseg = (H + hs - 1) / hs;
  const dim3 grid(nblk(cols * nseg, 8192));
  if (is_bf16) hipLaunchKernelGGL((dwconv7x3_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)x, w, bias, (bf16*)y, B, H, W, C, hs, nseg, rev);
  else {
    // column pairs x channel pairs: as many threads as the float4 form has for even W
    const size_t th = (size_t)B * nseg * ((W + 1) / 2) * (C / 2);
    hipLaunchKernelGGL(dwconv7x3_f32_kernel, dim3(nblk(th, 16384)), dim3(256), 0, st, (const float*)x, w, bias, (float*)y, B, H, W, C, hs, nseg, rev);
  }
  return hipGetLastError();
}
hipError_t launch_maxpool_same(const void* x, void* y, int in_bf16, int out_bf16, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int pt, int pl, int Ho, int Wo, hipStream_t st) {
  const dim3 grid(nblk((size_t)B * Ho * Wo * (C / 4)));
  if (in_bf16 && out_bf16) hipLaunchKernelGGL((maxpool_same_kernel<bf16, bf16>), grid, dim3(256), 0, st, (const bf16*)x, (bf16*)y, B, H, W, C, kh, kw, sh, sw, pt, pl, Ho, Wo);
  else if (in_bf16) hipLaunchKernelGGL((maxpool_same_kernel<bf16, float>), grid, dim3(256), 0, st, (const bf16*)x, (float*)y, B, H, W, C, kh, kw, sh, sw, pt, pl, Ho, Wo);
  else hipLaunchKernelGGL((maxpool_same_kernel<float, float>), grid, dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C, kh, kw, sh, sw, pt, pl, Ho, Wo);
  return hipGetLastError();
}
hipError_t launch_cvt_f32_bf16(const float* x, void* y, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(cvt_f32_bf16_kernel, dim3(nblk(n / 4)), dim3(256), 0, st, x, (bf16*)y, n / 4);
  return hipGetLastError();
}
hipError_t launch_fold_bn(const float* w, const float* beta, const float* mean, const float* var, float eps, size_t n, int C, float* wf, float* bf, hipStream_t st) {
  hipLaunchKernelGGL(fold_bn_kernel, dim3(nblk(n)), dim3(256), 0, st, w, beta, mean, var, eps, n, C, wf, bf);
  return hipGetLastError();
}
hipError_t launch_gru_seq(const float* xg, const float* xc, const float* whg, const float* whc, const int* seq_len, float* out, int B, int T, hipStream_t st) {
  hipLaunchKernelGGL(gru_fwd_kernel<false>, dim3(B), dim3(1024), 0, st, xg, xc, whg, whc, seq_len, out, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                     (float*)nullptr, T);
  return hipGetLastError();
}
hipError_t launch_mul_inplace(float* x, const float* m, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(mul_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, m, n);
  return hipGetLastError();
}
hipError_t launch_add_ears(float* out, const float* ears, int n, hipStream_t st) {
  hipLaunchKernelGGL(add_ears_kernel, dim3((n + 255) / 256), dim3(256), 0, st, out, ears, n);
  return hipGetLastError();
}

}  // namespace vp
