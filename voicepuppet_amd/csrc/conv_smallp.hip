// Few-pixel convolutions (conv_smallp.hip): the generator's 1x1 .. 16x16 bottleneck (pixrefer.py:215-257: merged_encoder_2..5,
// merged_decoder_5..2) in ONE launch per layer and pass.  Such a layer streams 2-16 MB of weights against 4-512 pixels per
// parity class; on the general path it was a 128-row tile kernel at 3-8 % of HBM, a split-K reduce, and one to three
// batch-norm launches, i.e. 3-5 dependent launches of 5-25 us each on the step's critical chain at 4-8 frames per GPU.
//
//   tile     32 output channels x PT = 16 / 32 / 64 pixels of one parity class; grid = (tiles, K splits), 4 waves per block
//   K loop   the four waves of a block take a quarter of the block's K range each; both MFMA operands are loaded straight into
//            registers (16-byte pieces: a weight fragment is 1 KB contiguous in the chunk-major packed layout, a pixel fragment is
//            a gather of 64-byte runs out of L2), up to four 64-byte K chunks (24 loads per lane) in flight per wave.  Taps that
//            fall outside the image for EVERY pixel are never enumerated (2x2 -> 1x1: 4 of 16 taps, 1x1 -> 2x2: 1 of 4 per class).
//   combine  waves fold through LDS in wave order; K splits over blocks write f32 slabs and the last-arriving block of a tile
//            adds them in split order (agent-scope release / acquire around one relaxed ticket, cdna_hip_programming.md 5.x
//            "in-launch split-K reduction") - results do not depend on arrival order.
//   epilogue SP_PLAIN: bias / activation / act'(ref) product / accumulate (epi_store8, as every other igemm kernel);
//            SP_FWD_BN: raw output stored, per-tile column sums of the values as stored; the last-arriving tile of a channel group
//            turns them into mean / biased variance -> scale, shift (pixrefer.py:99-101) and writes act(scale * y + shift) for
//            the consumers (lrelu for encoders, relu for decoders: pixrefer.py:182,243);
//            SP_BWD_BN: dz = [dz +] acc * act'(ref) stored, per-tile sums of dz and dz * zhat; the last-arriving tile of a channel
//            group finishes the batch-norm backward of the receiving tensor: c1, c2, dgamma, dbeta, dy in place.
// Summation orders are fixed (tile sums over pixels in pixel order, tiles in tile order), double accumulation like the
// stand-alone batch-norm kernels (pointwise.hip).
#include <string.h>

#include "igemm_device.h"
#include "launch.h"
#include "smallp_args.h"

namespace vp {

constexpr int SP_CT = 32;                      // output channels per tile (two MFMA row tiles)
constexpr int SP_PITCH = SP_CT + 4;            // floats per staged pixel row (+16 bytes: conflict-free 16-byte accesses)
constexpr int SP_HDR = 1024;                   // bytes: valid-tap list, flags, per-channel coefficients

__device__ __forceinline__ float round_as_stored(float x, bf16*) { return bf16_bits_to_f32(f32_to_bf16_bits(x)); }
__device__ __forceinline__ float round_as_stored(float x, float*) { return x; }

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&f)[8]);
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&f)[8]) { Elem<bf16>::unpack(*reinterpret_cast<const uint4*>(p), f); }
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&f)[8]) {
  Elem<float>::unpack(reinterpret_cast<const uint4*>(p)[0], f);
  Elem<float>::unpack(reinterpret_cast<const uint4*>(p)[1], f + 4);
}
template <typename T> __device__ __forceinline__ void store8t(T* p, const float (&f)[8]);
template <> __device__ __forceinline__ void store8t<bf16>(bf16* p, const float (&f)[8]) { *reinterpret_cast<uint4*>(p) = Elem<bf16>::pack(f); }
template <> __device__ __forceinline__ void store8t<float>(float* p, const float (&f)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}

// 16-byte / 8-byte WRITE-THROUGH stores (sc1) for everything another block of the launch will read: the data leaves this XCD's
// L2 with the store, so publishing needs no release fence (buffer_wbl2 would write back every dirty line of the L2, including
// those of kernels running beside this one on other streams: 1.7-6.5 us per block; cdna_hip_programming.md 6 G16 form R1)
typedef __attribute__((ext_vector_type(4))) unsigned int sp_u32x4;
__device__ __forceinline__ void store16_wt(__amdgpu_buffer_rsrc_t rs, size_t byte_off, const float4& v) {
  const sp_u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)byte_off, 0, 16);
}
__device__ __forceinline__ void store16_wt(__amdgpu_buffer_rsrc_t rs, size_t byte_off, const uint4& v) {
  const sp_u32x4 u = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)byte_off, 0, 16);
}
template <typename T> __device__ __forceinline__ void store8t_wt(__amdgpu_buffer_rsrc_t rs, size_t elem_off, const float (&f)[8]);
template <> __device__ __forceinline__ void store8t_wt<bf16>(__amdgpu_buffer_rsrc_t rs, size_t elem_off, const float (&f)[8]) {
  store16_wt(rs, elem_off * 2, Elem<bf16>::pack(f));
}
template <> __device__ __forceinline__ void store8t_wt<float>(__amdgpu_buffer_rsrc_t rs, size_t elem_off, const float (&f)[8]) {
  store16_wt(rs, elem_off * 4, make_float4(f[0], f[1], f[2], f[3]));
  store16_wt(rs, elem_off * 4 + 16, make_float4(f[4], f[5], f[6], f[7]));
}

// publish this block's write-through stores and draw a ticket on *cnt; true in every thread of the block that drew `last`
// (every wave drains its stores, one relaxed agent-scope ticket, one agent-scope acquire in the winner, then plain loads)
__device__ __forceinline__ bool publish_and_ticket(unsigned* cnt, unsigned last, int* flag_lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int won = t == last;
    if (won) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every ticket is drawn: leave the counter zero for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    *flag_lds = won;
  }
  __syncthreads();
  return *flag_lds != 0;
}

template <typename T, int NPT, int MODE, bool DUAL = false>
__global__ __launch_bounds__(256) void smallp_kernel(const SmallPArgs s) {
  constexpr int E = Elem<T>::E, KC = 4 * E;            // elements per 16-byte piece / per 64-byte K chunk
  constexpr int PT = NPT * 16;
  constexpr int U = 4;                                 // K chunks in flight per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* hdr = reinterpret_cast<int*>(smem);             // [0..15] valid taps, [16] their count, [17] winner flag
  float* coef = reinterpret_cast<float*>(smem + 128);  // [5][32] per-channel coefficients of the final pass
  float* red = reinterpret_cast<float*>(smem + SP_HDR);   // [4 waves][PT][SP_PITCH]
  const IgemmArgs& a = s.g;
  const bool hi = MODE != SP_PLAIN && sizeof(T) == 2 && s.hi != 0;     // float32 storage of the batch-normalised tensor (smallp_args.h)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Pc = a.N * a.Hg * a.Wg;                    // pixels per class
  const int npt = (Pc + PT - 1) / PT;
  const int tile = blockIdx.x;
  const int pt = tile % npt, cls = (tile / npt) % a.nclass, ct = tile / (npt * a.nclass);
  const int ks = blockIdx.y, nks = a.splitk;
  const int ntiles = gridDim.x;
  // two-output launch (IgemmArgs::split_c; both data gradients of a decoder): the channel tiles of the SECOND output take the plain
  // epilogue whatever MODE says - that tensor still has a contribution to come, only the first one's batch-norm backward runs here
  const bool plain2 = DUAL && MODE != SP_PLAIN && ct * SP_CT >= a.split_c;

  if (tid == 0) {
    int n = 0;
    const unsigned m = s.tap_mask[cls];
    for (int t = 0; t < a.ntaps; ++t)
      if ((m >> t) & 1u) hdr[n++] = (t << 24) | (((int)a.taps[cls].dh[t] & 0xff) << 8) | ((int)a.taps[cls].dw[t] & 0xff);
    hdr[16] = n;
  }
  __syncthreads();
  const int nvt = hdr[16];

  // this lane's pixel columns: pixel tp * 16 + (lane & 15) of the tile
  int pn[NPT], pbh[NPT], pbw[NPT];
  bool pok[NPT];
#pragma unroll
  for (int tp = 0; tp < NPT; ++tp) {
    const int pidx = pt * PT + tp * 16 + (lane & 15);
    pok[tp] = pidx < Pc;
    const int hw = a.Hg * a.Wg;
    const int pc = pok[tp] ? pidx : 0;
    const int n = pc / hw, rem = pc - n * hw, q = rem / a.Wg;
    pn[tp] = n * a.Hin; pbh[tp] = q * a.sh; pbw[tp] = (rem - q * a.Wg) * a.sw;
  }
  const T* x0 = reinterpret_cast<const T*>(a.x.ptr[0]);
  const T* x1 = reinterpret_cast<const T*>(a.x.ptr[1]);
  const int C0 = a.x.C[0], C1 = a.x.C[1];
  const int lcpt = a.sp_lcpt;                                      // log2(chunks per tap)
  const int cpt = 1 << lcpt;
  const int total = nvt << lcpt;
  const int parts = nks * 4, id = ks * 4 + wave;
  const int j0 = (int)((long long)total * id / parts), j1 = (int)((long long)total * (id + 1) / parts);
  const T* wbase = reinterpret_cast<const T*>(a.Wp) + (size_t)cls * a.wp_rows * a.Kpad + (size_t)(ct * SP_CT + (lane & 15)) * KC + (lane >> 4) * E;
  const int wstep = a.wp_rows * KC;
  const uint4* zeros = reinterpret_cast<const uint4*>(a.zeros);

  f32x4 acc[2][NPT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NPT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int j = j0; j < j1; j += U) {
    uint4 fa[U][2], fb[U][NPT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = j + u < j1;
      const int jj = live ? j + u : j;
      const int ti = jj >> lcpt, cc = jj & (cpt - 1);
      const int tv = hdr[ti];
      const int tap = tv >> 24, dh = (int)(signed char)((tv >> 8) & 0xff), dw = (int)(signed char)(tv & 0xff);
      const T* wp = wbase + (size_t)(tap * cpt + cc) * wstep;
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) fa[u][tc] = *(live ? reinterpret_cast<const uint4*>(wp + tc * 16 * KC) : zeros);   // (streamed `nt` weight loads: +-0, EXPERIMENTS.md)
      const int c = cc * KC + (lane >> 4) * E;
      const bool s1 = c >= C0;
      const T* xb = s1 ? x1 : x0;
      const int Cs = s1 ? C1 : C0, cl = s1 ? c - C0 : c;
#pragma unroll
      for (int tp = 0; tp < NPT; ++tp) {
        const int ih = pbh[tp] + dh, iw = pbw[tp] + dw;
        const bool ok = live && pok[tp] && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
        fb[u][tp] = *(ok ? reinterpret_cast<const uint4*>(xb + ((pn[tp] + ih) * a.Win + iw) * Cs + cl) : zeros);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int tc = 0; tc < 2; ++tc)
#pragma unroll
        for (int tp = 0; tp < NPT; ++tp) acc[tc][tp] = mma16<T>(fa[u][tc], fb[u][tp], acc[tc][tp]);
  }

  // ---- fold the four waves (wave order), then the K splits (split order) ---------------------------------------------------
#pragma unroll
  for (int tc = 0; tc < 2; ++tc)
#pragma unroll
    for (int tp = 0; tp < NPT; ++tp)
      *reinterpret_cast<f32x4*>(red + ((wave * PT + tp * 16 + (lane & 15)) * SP_PITCH + tc * 16 + 4 * (lane >> 4))) = acc[tc][tp];
  __syncthreads();
  constexpr int NQ = PT * (SP_CT / 4);                 // float4 items of a tile
  const __amdgpu_buffer_rsrc_t rs_slab = make_rsrc(s.slab, 0xFFFFFFFFu);
  const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.Y, 0xFFFFFFFFu);
  for (int idx = tid; idx < NQ; idx += 256) {
    const int p = idx >> 3, c4 = idx & 7;
    float4 v = *reinterpret_cast<const float4*>(red + (p * SP_PITCH + c4 * 4));
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(red + ((w * PT + p) * SP_PITCH + c4 * 4));
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (nks > 1) store16_wt(rs_slab, ((((size_t)ks * ntiles + tile) * PT + p) * SP_CT + c4 * 4) * sizeof(float), v);
    else *reinterpret_cast<float4*>(red + (p * SP_PITCH + c4 * 4)) = v;
  }
  if (nks > 1) {
    if (!publish_and_ticket(s.cnt + tile, (unsigned)(nks - 1), hdr + 17)) return;
    for (int idx = tid; idx < NQ; idx += 256) {
      const int p = idx >> 3, c4 = idx & 7;
      const float* src = s.slab + ((size_t)tile * PT + p) * SP_CT + c4 * 4;
      const size_t sstride = (size_t)ntiles * PT * SP_CT;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k0 = 0; k0 < nks; k0 += 8) {
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = k0 + u < nks ? *reinterpret_cast<const float4*>(src + (size_t)(k0 + u) * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (k0 + u >= nks) break;
          v.x += x[u].x; v.y += x[u].y; v.z += x[u].z; v.w += x[u].w;
        }
      }
      *reinterpret_cast<float4*>(red + (p * SP_PITCH + c4 * 4)) = v;
    }
  }
  __syncthreads();

  // ---- tile epilogue: thread = 8 consecutive channels of one pixel ------------------------------------------------------------
  const LinearPix pix{a, cls, pt * PT, Pc};
  float* til2 = red + PT * SP_PITCH;                   // second staged tile (SP_BWD_BN: dz * zhat)
  const int npa = a.nclass * npt;                      // pixel tiles of a channel group, all classes
  const int pa = cls * npt + pt;
  for (int idx = tid; idx < PT * 4; idx += 256) {
    const int p = idx >> 2, cg = idx & 3;
    const long long ot = pix(p);
    const int c0 = ct * SP_CT + cg * 8;
    float v[8];
    {
      const float4 v0 = *reinterpret_cast<const float4*>(red + (p * SP_PITCH + cg * 8));
      const float4 v1 = *reinterpret_cast<const float4*>(red + (p * SP_PITCH + cg * 8 + 4));
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    }
    float z[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = 0.f;
    if (ot >= 0) {
      const size_t off = (size_t)(ot >> 8) + c0;
      if (MODE == SP_PLAIN || plain2) {
        epi_store8<T, DUAL>(a, ot, c0, off, v);
      } else {
        // SP_FWD_BN: the raw output (a bias in front of a batch-norm cancels; no activation).  SP_BWD_BN: act'(ref) product and the
        // accumulation over the tensor's consumers, as epi_store8.  Stored write-through: the channel group's last tile reads it back.
        if (MODE == SP_BWD_BN) {
          if (a.ref) {
            float zr[8];
            load8<T>(reinterpret_cast<const T*>(a.ref) + off, zr);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= act_grad(a.ref_act, zr[e]);
          }
          if (a.accumulate) {
            float old[8];
            if (hi) load8<float>(reinterpret_cast<const float*>(a.Y) + off, old);
            else load8<T>(reinterpret_cast<const T*>(a.Y) + off, old);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += old[e];
          }
        }
        if (hi) {
          store8t_wt<float>(rs_y, off, v);               // float32 tensor: nothing is rounded, the statistics see what is stored
        } else {
          store8t_wt<T>(rs_y, off, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = round_as_stored(v[e], (T*)nullptr);
        }
      }
      if (MODE == SP_BWD_BN && !plain2) {
        float y[8];
        if (hi) load8<float>(reinterpret_cast<const float*>(s.bn_y) + off, y);
        else load8<T>(reinterpret_cast<const T*>(s.bn_y) + off, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = v[e] * ((y[e] - s.bn_mu[c0 + e]) * s.bn_rstd[c0 + e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    if (MODE != SP_PLAIN && !plain2) {
      *reinterpret_cast<float4*>(red + (p * SP_PITCH + cg * 8)) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(red + (p * SP_PITCH + cg * 8 + 4)) = make_float4(v[4], v[5], v[6], v[7]);
      if (MODE == SP_BWD_BN) {
        *reinterpret_cast<float4*>(til2 + (p * SP_PITCH + cg * 8)) = make_float4(z[0], z[1], z[2], z[3]);
        *reinterpret_cast<float4*>(til2 + (p * SP_PITCH + cg * 8 + 4)) = make_float4(z[4], z[5], z[6], z[7]);
      }
    }
  }
  if (MODE == SP_PLAIN || plain2) return;

  // ---- per-tile column sums (pixel order), then the channel group's last tile finishes the batch-norm ----------------------------
  __syncthreads();
  if (tid < 64) {
    const int c = tid & 31, which = tid >> 5;
    double t = 0;
    if (MODE == SP_FWD_BN) {
      for (int p = 0; p < PT; ++p) { const float x = red[p * SP_PITCH + c]; t += which ? (double)x * x : (double)x; }
    } else {
      const float* src = which ? til2 : red;
      for (int p = 0; p < PT; ++p) t += (double)src[p * SP_PITCH + c];
    }
    __hip_atomic_store(s.part + ((size_t)(ct * npa + pa) * 2 + which) * SP_CT + c, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!publish_and_ticket(s.cnt + ntiles + ct, (unsigned)(npa - 1), hdr + 17)) return;

  const int NP = a.N * a.Hof * a.Wof;                  // pixels of the whole output tensor (one batch-norm group)
  if (tid < 32) {
    const int c = ct * SP_CT + tid;
    double s0 = 0, s1 = 0;
    for (int k = 0; k < npa; ++k) {
      s0 += s.part[((size_t)(ct * npa + k) * 2) * SP_CT + tid];
      s1 += s.part[((size_t)(ct * npa + k) * 2 + 1) * SP_CT + tid];
    }
    if (MODE == SP_FWD_BN) {
      const double mean = s0 / NP;
      double var = s1 / NP - mean * mean;
      if (var < 0) var = 0;
      const float rstd = (float)(1.0 / sqrt(var + (double)s.eps));
      const float g = s.gamma[c], b = s.beta[c];
      const float sc = (var == 0.0) ? 0.f : g * rstd;                                // zero variance: z == beta exactly (bn_finalize_kernel)
      const float sh = (var == 0.0) ? b : (float)((double)b - mean * (double)sc);
      s.aff_a[c] = sc; s.aff_b[c] = sh; s.mu[c] = (float)mean; s.rstd[c] = rstd;
      coef[tid] = sc; coef[32 + tid] = sh;
    } else {
      const float c1 = (float)(s0 / NP), c2 = (float)(s1 / NP);
      s.c1[c] = c1; s.c2[c] = c2;
      if (s.dgamma) { s.dgamma[c] = (float)s1; s.dbeta[c] = (float)s0; }
      if (s.dbias_zero) s.dbias_zero[c] = 0.f;
      const float rs = s.bn_rstd[c];
      coef[tid] = c1; coef[32 + tid] = c2; coef[64 + tid] = s.bn_mu[c]; coef[96 + tid] = rs; coef[128 + tid] = s.bn_gamma[c] * rs;
    }
  }
  __syncthreads();
  // four items per trip, all loads issued before the first use (one block walks the whole channel group: the pass is pure load latency)
  constexpr int FU = 4;
  for (int idx0 = tid; idx0 < NP * 4; idx0 += 256 * FU) {
    float f[FU][8], y[FU][8];
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      const int idx = idx0 + u * 256;
      const int ii = idx < NP * 4 ? idx : idx0;
      const size_t off = (size_t)(ii >> 2) * a.ldY + ct * SP_CT + (ii & 3) * 8;
      if (hi) {
        load8<float>(reinterpret_cast<const float*>(a.Y) + off, f[u]);
        if (MODE == SP_BWD_BN) load8<float>(reinterpret_cast<const float*>(s.bn_y) + off, y[u]);
      } else {
        load8<T>(reinterpret_cast<const T*>(a.Y) + off, f[u]);
        if (MODE == SP_BWD_BN) load8<T>(reinterpret_cast<const T*>(s.bn_y) + off, y[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      const int idx = idx0 + u * 256;
      if (idx >= NP * 4) break;
      const int cg = idx & 3;
      const int c0 = ct * SP_CT + cg * 8;
      const size_t off = (size_t)(idx >> 2) * a.ldY + c0;
      if (MODE == SP_FWD_BN) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[u][e] = fmaf(coef[cg * 8 + e], f[u][e], coef[32 + cg * 8 + e]);
        if (s.out_lrelu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = act_apply(ACT_LRELU, f[u][e]);
          store8t<T>(reinterpret_cast<T*>(s.out_lrelu) + off, o);
        }
        if (s.out_relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = act_apply(ACT_RELU, f[u][e]);
          store8t<T>(reinterpret_cast<T*>(s.out_relu) + off, o);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = cg * 8 + e;
          const float zh = (y[u][e] - coef[64 + k]) * coef[96 + k];
          f[u][e] = coef[128 + k] * (f[u][e] - coef[k] - zh * coef[32 + k]);
        }
        store8t<T>(reinterpret_cast<T*>(hi ? s.dy_out : a.Y) + off, f[u]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
size_t smallp_smem(int npt) { return SP_HDR + (size_t)4 * npt * 16 * SP_PITCH * sizeof(float); }

template <typename T, int NPT>
static hipError_t launch_smallp_t(const SmallPArgs& s, dim3 grid, hipStream_t st) {
  const size_t sm = smallp_smem(NPT);
  switch (s.mode) {
    case SP_PLAIN: hipLaunchKernelGGL((smallp_kernel<T, NPT, SP_PLAIN>), grid, dim3(256), sm, st, s); break;
    case SP_FWD_BN: hipLaunchKernelGGL((smallp_kernel<T, NPT, SP_FWD_BN>), grid, dim3(256), sm, st, s); break;
    case SP_BWD_BN:
      if (s.g.split_c) hipLaunchKernelGGL((smallp_kernel<T, NPT, SP_BWD_BN, true>), grid, dim3(256), sm, st, s);       // paired data gradients
      else hipLaunchKernelGGL((smallp_kernel<T, NPT, SP_BWD_BN>), grid, dim3(256), sm, st, s);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_smallp(const SmallPArgs& s, int is_bf16, hipStream_t st) {
  const IgemmArgs& a = s.g;
  const int npt_t = a.sp_npt, PT = npt_t * 16;
  const int Pc = a.N * a.Hg * a.Wg;
  const int npt = (Pc + PT - 1) / PT;
  if (a.CoutPad % SP_CT || !s.cnt || !a.zeros || (a.splitk > 1 && !s.slab) || (s.mode != SP_PLAIN && !s.part)) return hipErrorInvalidValue;
  if (s.hi && (!is_bf16 || s.mode == SP_PLAIN || (s.mode == SP_BWD_BN && !s.dy_out))) return hipErrorInvalidValue;
  if (a.split_c && s.mode != SP_BWD_BN) return hipErrorInvalidValue;       // the two-output form: first output with its batch-norm backward
  dim3 grid((a.CoutPad / SP_CT) * a.nclass * npt, a.splitk, 1);
  if (is_bf16) {
    if (npt_t == 1) return launch_smallp_t<bf16, 1>(s, grid, st);
    if (npt_t == 2) return launch_smallp_t<bf16, 2>(s, grid, st);
    if (npt_t == 4) return launch_smallp_t<bf16, 4>(s, grid, st);
  } else {
    if (npt_t == 1) return launch_smallp_t<float, 1>(s, grid, st);
    if (npt_t == 2) return launch_smallp_t<float, 2>(s, grid, st);
    if (npt_t == 4) return launch_smallp_t<float, 4>(s, grid, st);
  }
  return hipErrorInvalidValue;
}

// plain-epilogue form behind launch_igemm (IgemmArgs::patch == 3): the stand-alone op entry points and BN-free layers
hipError_t launch_igemm_smallp(const IgemmArgs& a, int is_bf16, hipStream_t st) {
  SmallPArgs s;
  memset(&s, 0, sizeof(s));
  s.g = a;
  for (int c = 0; c < 4; ++c) s.tap_mask[c] = a.sp_mask[c];
  s.slab = a.partial;
  s.cnt = a.sp_cnt;
  s.mode = SP_PLAIN;
  return launch_smallp(s, is_bf16, st);
}

}  // namespace vp
