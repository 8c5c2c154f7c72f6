// Device-side building blocks shared by the LDS-DMA implicit-GEMM kernels (conv_kernels.hip): the swizzled 16-row block
// image, the LDS-DMA wrappers (flat and buffer-descriptor forms) and the staged LDS epilogue with 16-byte row stores.
#pragma once
#include "conv_args.h"
#include "vp_common.h"

namespace vp {

// piece permutation of row i inside a 16-row block: LDS slot = i*4 + (g ^ rb_swz(i)).  The DMA writes
// slot = lane, so lane (i = lane>>2, p = lane&3) fetches global piece p ^ rb_swz(i): four adjacent lanes
// read one contiguous 64-byte segment (coalesced), and with h = (0,2,3,1) every ds_read_b128 service
// group {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... lands on 16 distinct slots mod 16 (conflict-free).
__device__ __forceinline__ int rb_swz(int i) { return (0x78 >> ((i >> 1) & 6)) & 3; }

template <typename T, int TC, int TP, int BC, int BP>
__device__ __forceinline__ void mma_chunk_rb(const uint4* __restrict__ ldsA, const uint4* __restrict__ ldsB,
                                             int blkA0, int blkB0, int lane, f32x4 (&acc)[TC][TP]) {
  const int i = lane & 15, g = lane >> 4;
  const int so = i * 4 + (g ^ rb_swz(i));
  uint4 fa[TC], fb[TP];
#pragma unroll
  for (int t = 0; t < TC; ++t) fa[t] = ldsA[(blkA0 + t) * 64 + so];
#pragma unroll
  for (int t = 0; t < TP; ++t) fb[t] = ldsB[(blkB0 + t) * 64 + so];
#pragma unroll
  for (int tc = 0; tc < TC; ++tc)
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) acc[tc][tp] = mma16<T>(fa[tc], fb[tp], acc[tc][tp]);
}

__device__ __forceinline__ void dma16(const void* src, uint4* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

// Buffer-descriptor form of the same DMA (buffer_load_dwordx4 ... offen lds): address = descriptor base + per-lane
// 32-bit byte offset + SCALAR offset.  Two things the flat form cannot do: an out-of-range offset returns zeros
// (padding pixels need no zero page and no per-lane select), and the per-chunk advance along K is a scalar add,
// so a K chunk costs no vector ALU at all once the per-tap lane offsets exist.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x27000);
}
__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, uint4* lds_dst_wave_uniform) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, (int)voff, (int)soff, 0, 0);
}
constexpr unsigned DMA_OOB = 0xF0000000u;     // lane offset beyond every descriptor's range -> the DMA writes zeros

// ------------------------------------------------------------------------------------------------
// Staged epilogue: the accumulators go through LDS as an f32 [pixel][channel] tile, then every thread
// finishes 8 consecutive channels of one pixel (bias, activation, act'(ref) product, accumulate) and
// issues ONE 16-byte store; 16 adjacent lanes cover 256 contiguous bytes of an NHWC row.  (The direct
// MFMA-layout store writes 8 bytes per lane at a 32-byte granularity and ran at ~0.4 TB/s.)
// ------------------------------------------------------------------------------------------------
// default tile-row -> output-pixel map: rows are consecutive pixels of the flattened (n, q, r) grid
// A tile-row -> output-pixel functor may also know where the 2x2-max-pooled image of its tile goes (2-D tiles only):
// member `static constexpr bool HAS_POOL` + `long long pool(int pooled_row, int pooled_col) const` (offset in elements, -1 outside).
template <typename F, typename = void> struct pix_has_pool { static constexpr bool value = false; };
template <typename F> struct pix_has_pool<F, decltype((void)F::HAS_POOL)> { static constexpr bool value = F::HAS_POOL; };

struct LinearPix {
  const IgemmArgs& a; int cls, p_base, P;
  __device__ __forceinline__ long long operator()(int row) const {
    const int pidx = p_base + row;
    if (pidx >= P) return -1;
    const int hw = a.Hg * a.Wg;
    const int n = pidx / hw, rem = pidx - n * hw, q = rem / a.Wg, r = rem - q * a.Wg;
    long long off = (((long long)n * a.Hof + (q * a.os + a.o0h[cls])) * a.Wof + (r * a.os + a.o0w[cls])) * a.ldY;
    return (off << 8) | (long long)(n / a.ref_group_n);   // BN group of the pixel in the low byte
  }
};

// number of epilogue passes so that the f32 tile of one pass (+ its offset table) fits the ring's LDS; a wave's rows stay in one pass
constexpr int epi_passes(int BC, int BP, int WP, int ring_bytes) {
  for (int np = 1; np <= WP; np *= 2)
    if ((BP / np) * (BC * 4 + 16) + (BP / np) * 8 <= ring_bytes) return np;
  return WP;
}

// the same for the bf16-staged form of the epilogue (staged_epilogue, FASTBF)
constexpr int epi_passes16(int BC, int BP, int WP, int ring_bytes) {
  for (int np = 1; np <= WP; np *= 2)
    if ((BP / np) * (BC * 2 + 16) + (BP / np) * 8 <= ring_bytes) return np;
  return WP;
}

// finish 8 consecutive channels c0..c0+7 of one output pixel (offset `off`, BN group in the low byte of `ot`): bias, activation,
// act'(ref) product, accumulate, one 16-byte store (two for f32)
// DUAL: the two-output form (IgemmArgs::split_c) is compiled only into the kernels that are launched with it - as a run-time test in
// EVERY epilogue it cost the 128-register tiles 4 % of the whole step (batch 32: 8.2 -> 8.6 ms, EXPERIMENTS.md 0.2)
template <typename T, bool DUAL = false>
__device__ __forceinline__ void epi_store8(const IgemmArgs& a, long long ot, int c0, size_t off, float (&v)[8]) {
  void* Yp = a.Y;
  const void* refp = a.ref;
  int accu = a.accumulate, yf32 = a.y_f32;
  if constexpr (DUAL) {
    if (c0 >= a.split_c) {                 // second output (uniform per 8-channel group; per block in practice)
      Yp = a.Y2; refp = a.ref2; accu = a.accumulate2; yf32 = a.y2_f32;
      c0 -= a.split_c; off -= (size_t)a.split_c;
    }
  }
  if (a.bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += a.bias[c0 + e];
  }
  if (a.out_act != ACT_NONE) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act_apply(a.out_act, v[e]);
  }
  if (refp) {
    float z[8];
    const T* rp = reinterpret_cast<const T*>(refp) + off;
    if (sizeof(T) == 2) Elem<bf16>::unpack(*reinterpret_cast<const uint4*>(rp), z);
    else {
      Elem<float>::unpack(reinterpret_cast<const uint4*>(rp)[0], z);
      Elem<float>::unpack(reinterpret_cast<const uint4*>(rp)[1], z + 4);
    }
    if (a.ref_a) {
      const int goff = (int)(ot & 255) * a.Cout + c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = fmaf(a.ref_a[goff + e], z[e], a.ref_b[goff + e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= act_grad(a.ref_act, z[e]);
  }
  if (yf32 || sizeof(T) == 4) {
    float* yp = reinterpret_cast<float*>(Yp) + off;
    if (accu) {
      const float4 e0 = reinterpret_cast<const float4*>(yp)[0], e1 = reinterpret_cast<const float4*>(yp)[1];
      v[0] += e0.x; v[1] += e0.y; v[2] += e0.z; v[3] += e0.w; v[4] += e1.x; v[5] += e1.y; v[6] += e1.z; v[7] += e1.w;
    }
#ifdef VP_NT_STORE
#pragma unroll
    for (int e = 0; e < 8; ++e) __builtin_nontemporal_store(v[e], yp + e);
#else
    reinterpret_cast<float4*>(yp)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(yp)[1] = make_float4(v[4], v[5], v[6], v[7]);
#endif
  } else {
    bf16* yp = reinterpret_cast<bf16*>(Yp) + off;
    if (accu) {
      float e[8];
      Elem<bf16>::unpack(*reinterpret_cast<const uint4*>(yp), e);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += e[k];
    }
#ifdef VP_NT_STORE
    { const uint4 pk = Elem<bf16>::pack(v); __builtin_nontemporal_store(pk.x, reinterpret_cast<unsigned*>(yp)); __builtin_nontemporal_store(pk.y, reinterpret_cast<unsigned*>(yp) + 1);
      __builtin_nontemporal_store(pk.z, reinterpret_cast<unsigned*>(yp) + 2); __builtin_nontemporal_store(pk.w, reinterpret_cast<unsigned*>(yp) + 3); }
#else
    *reinterpret_cast<uint4*>(yp) = Elem<bf16>::pack(v);
#endif
  }
}


// STATS (a batch-normalised layer): the epilogue also produces the layer's batch statistics.  While a pass's f32 tile sits in
// LDS, thread t sums column (channel) t % BC over its share of the rows - of the values AS STORED, i.e. rounded to T - into two
// registers; after the last pass the NT / BC threads of a channel fold through LDS and the block writes one [2][channels]
// partial per pixel tile for bn_finalize_kernel.  The tensor is never re-read.
//
// NPASS16 > 0 (bf16 kernels without statistics: the 3x3 patch kernel's perceptual-trunk launches): when the launch has no reference
// product, no accumulation and a bf16 output, bias + activation + rounding are applied in the lane's own registers - the same
// operations in the same order, so the stored bits are the same - and the tile is staged as bf16: half the LDS bytes, NPASS16
// passes instead of NPASS, the store loop a plain copy; the fused max pool takes the maximum of the rounded values (rounding is monotonic).
// STATS == 2 (the launch that completes the gradient of a batch-normalised tensor, round 6): the epilogue also produces the two sums of
// that tensor's batch-norm BACKWARD pass - as raw moments sum dz and sum dz * y of the gradient AS STORED (after the act'(ref) product, the
// accumulation and the rounding to T) - so bn_reduce_kernel<T, 1> never re-reads y and dz (IgemmArgs::bst_y).
// A thread of the store loop stays on one 8-channel group (NT % CG == 0): sixteen running sums in registers.
template <typename T, int TC, int TP, int BC, int BP, int NPASS, int NT, int STATS = 0, int NPASS16 = 0, bool DUAL = false, typename PixFn>
__device__ __forceinline__ void staged_epilogue(const IgemmArgs& a, const PixFn& pixfn, int c_base, int blkA0, int blkB0,
                                                f32x4 (&acc)[TC][TP], char* smem, int pt = 0, int cls = 0) {
  if constexpr (NPASS16 > 0 && sizeof(T) == 2 && STATS == 0) {
    if (!a.ref && !a.accumulate && !a.y_f32 && (a.out_act == ACT_NONE || a.out_act == ACT_RELU)) {
      constexpr int PITCHB = BC * 2 + 16, CGB = BC / 8, RPB = BP / NPASS16;
      static_assert(RPB % (TP * 16) == 0, "a wave's pixel rows must fall into one pass");
      const int tid = threadIdx.x, lane = tid & 63;
      long long* otab = reinterpret_cast<long long*>(smem + RPB * PITCHB);
#pragma unroll
      for (int ps = 0; ps < NPASS16; ++ps) {
        __syncthreads();
        if (tid < RPB) otab[tid] = pixfn(ps * RPB + tid);
        if (blkB0 * 16 >= ps * RPB && blkB0 * 16 < (ps + 1) * RPB) {
#pragma unroll
          for (int tc = 0; tc < TC; ++tc) {
            const int ch = tile_chan0(a.rowperm, blkA0 + tc, lane >> 4);
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias && c_base + ch < a.Cout) {
#pragma unroll
              for (int e = 0; e < 4; ++e) b4[e] = a.bias[c_base + ch + e];
            }
#pragma unroll
            for (int tp = 0; tp < TP; ++tp) {
              const int row = (blkB0 + tp) * 16 + (lane & 15) - ps * RPB;
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] = acc[tc][tp][e];
                if (a.bias) v[e] += b4[e];
                if (a.out_act != ACT_NONE) v[e] = act_apply(a.out_act, v[e]);
              }
              uint2 pk;
              pk.x = Elem<bf16>::pack2(v[0], v[1]); pk.y = Elem<bf16>::pack2(v[2], v[3]);
              *reinterpret_cast<uint2*>(smem + row * PITCHB + ch * 2) = pk;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __syncthreads();
        bool store_y = true;
        if constexpr (pix_has_pool<PixFn>::value) store_y = !(a.pool_out != nullptr && a.pool_only);
        if (store_y)
          for (int idx = tid; idx < RPB * CGB; idx += NT) {
            const int p = idx / CGB, cgp = idx - p * CGB;
            const long long ot = otab[p];
            const int c0 = c_base + cgp * 8;
            if (ot < 0 || c0 >= a.Cout) continue;
            const uint4 pk = *reinterpret_cast<const uint4*>(smem + p * PITCHB + cgp * 16);
            unsigned* yp = reinterpret_cast<unsigned*>(reinterpret_cast<bf16*>(a.Y) + (size_t)(ot >> 8) + c0);
#ifdef VP_NT_STORE
            __builtin_nontemporal_store(pk.x, yp); __builtin_nontemporal_store(pk.y, yp + 1);
            __builtin_nontemporal_store(pk.z, yp + 2); __builtin_nontemporal_store(pk.w, yp + 3);
#else
            *reinterpret_cast<uint4*>(yp) = pk;
#endif
          }
        if constexpr (pix_has_pool<PixFn>::value) {
          if (a.pool_out != nullptr) {
            static_assert(RPB % 32 == 0, "a pass must hold whole pairs of tile rows");
            constexpr int PR = RPB / 4;
            for (int idx = tid; idx < PR * CGB; idx += NT) {
              const int pp = idx / CGB, cgp = idx - pp * CGB;
              const int pyl = pp >> 3, pxl = pp & 7;
              const int c0 = c_base + cgp * 8;
              const long long po = pixfn.pool(ps * (RPB / 32) + pyl, pxl);
              if (po < 0 || c0 >= a.Cout) continue;
              const int r00 = pyl * 32 + pxl * 2;
              float m[8];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int rr = r00 + (k >> 1) * 16 + (k & 1);
                float v[8];
                Elem<bf16>::unpack(*reinterpret_cast<const uint4*>(smem + rr * PITCHB + cgp * 16), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = k == 0 ? v[e] : fmaxf(m[e], v[e]);
              }
              *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(a.pool_out) + po + c0) = Elem<bf16>::pack(m);
            }
          }
        }
      }
      return;
    }
  }
  constexpr int PITCH = BC * 4 + 16;                 // bytes per pixel row (+16: conflict-free b128 writes)
  constexpr int CG = BC / 8;                         // 8-channel groups per row
  constexpr int RP = BP / NPASS;                     // pixel rows staged per pass (keeps the tile inside the ring's LDS)
  static_assert(RP % (TP * 16) == 0, "a wave's pixel rows must fall into one pass");
  const int tid = threadIdx.x, lane = tid & 63;
  long long* otab = reinterpret_cast<long long*>(smem + RP * PITCH);
  static_assert(STATS != 1 || NT % BC == 0, "column sums: whole thread groups per channel");
  static_assert(STATS != 2 || NT % CG == 0, "backward sums: a thread keeps one channel group");
  float bsum = 0.f, bsq = 0.f;
  // STATS == 2: this thread's channel group and which output it belongs to; RAW moments (sum dz, sum dz * y - the finalize turns the second
  // into sum dz * zhat = rstd * (sum dz * y - mean * sum dz), BnArgs::raw): no per-channel constants in registers next to the accumulators
  float gs0[STATS == 2 ? 8 : 1], gs1[STATS == 2 ? 8 : 1];
  const T* gy = nullptr;
  if constexpr (STATS == 2) {
    int c0 = c_base + (tid % CG) * 8;
    const void* yb = a.bst_y;
    int ct = a.Cout;
    if constexpr (DUAL) {
      ct = a.split_c;
      if (c0 >= a.split_c) { yb = a.bst_y2; c0 -= a.split_c; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { gs0[e] = 0.f; gs1[e] = 0.f; }
    if (yb && c0 < ct) gy = reinterpret_cast<const T*>(yb);
  }
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    __syncthreads();                                 // ring (pass 0) / previous pass's tile no longer needed
    if (tid < RP) otab[tid] = pixfn(ps * RP + tid);
    if (blkB0 * 16 >= ps * RP && blkB0 * 16 < (ps + 1) * RP) {
#pragma unroll
      for (int tp = 0; tp < TP; ++tp)
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
          const int row = (blkB0 + tp) * 16 + (lane & 15) - ps * RP;
          const int ch = tile_chan0(a.rowperm, blkA0 + tc, lane >> 4);
          *reinterpret_cast<f32x4*>(smem + row * PITCH + ch * 4) = acc[tc][tp];
          __builtin_amdgcn_sched_barrier(0);   // keep the AGPR->VGPR copies 4 at a time (register budget of the K loop)
        }
    }
    __syncthreads();
    if constexpr (STATS == 1) {
      const int col = tid % BC;
      for (int rr = tid / BC; rr < RP; rr += NT / BC) {
        if (otab[rr] < 0) continue;
        float x = *reinterpret_cast<const float*>(smem + rr * PITCH + col * 4);
        if (sizeof(T) == 2) x = bf16_bits_to_f32(f32_to_bf16_bits(x));
        bsum += x; bsq = fmaf(x, x, bsq);
      }
    }
    bool store_y = true;
    if constexpr (pix_has_pool<PixFn>::value) store_y = !(a.pool_out != nullptr && a.pool_only);
    if (store_y)
    for (int idx = tid; idx < RP * CG; idx += NT) {
      const int p = idx / CG, cgp = idx - p * CG;
      const long long ot = otab[p];
      const int c0 = c_base + cgp * 8;
      if (ot < 0 || c0 >= a.Cout) continue;
      const size_t off = (size_t)(ot >> 8) + c0;
      float v[8];
      {
        const float4 v0 = *reinterpret_cast<const float4*>(smem + p * PITCH + cgp * 32);
        const float4 v1 = *reinterpret_cast<const float4*>(smem + p * PITCH + cgp * 32 + 16);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
      }
      // STATS == 2: y of the same element, requested BEFORE the store helper's own loads (reference, accumulated gradient) so that all of
      // them wait out one memory latency (behind it the load cost a second round trip per item: +0.17 ms on the batch-32 step)
      uint4 ry0 = make_uint4(0, 0, 0, 0), ry1 = make_uint4(0, 0, 0, 0);
      if constexpr (STATS == 2) {
        if (gy) {
          size_t yo = off;
          if constexpr (DUAL) { if (c0 >= a.split_c) yo -= (size_t)a.split_c; }
          ry0 = reinterpret_cast<const uint4*>(gy + yo)[0];
          if (sizeof(T) == 4) ry1 = reinterpret_cast<const uint4*>(gy + yo)[1];
        }
      }
      epi_store8<T, DUAL>(a, ot, c0, off, v);
      if constexpr (STATS == 2) {
        if (gy) {        // (epi_store8 left the value it stored in v, unrounded)
          float fy[8];
          if (sizeof(T) == 2) Elem<bf16>::unpack(ry0, fy);
          else { Elem<float>::unpack(ry0, fy); Elem<float>::unpack(ry1, fy + 4); }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = sizeof(T) == 2 ? bf16_bits_to_f32(f32_to_bf16_bits(v[e])) : v[e];
            gs0[e] += d;
            gs1[e] = fmaf(d, fy[e], gs1[e]);
          }
        }
      }
    }
    // fused 2x2 / stride-2 max pool of the tile (VGG conv1_2 / conv2_2: slim max_pool2d, vgg_simple.py:141,144): the pass holds
    // whole pairs of 16-pixel tile rows; bias + (monotonic) activation + rounding commute with the max, so this equals pooling the
    // stored tensor bit for bit (relu / identity outputs only: the host does not ask for it otherwise)
    if constexpr (pix_has_pool<PixFn>::value) {
      if (a.pool_out != nullptr) {
        static_assert(RP % 32 == 0, "a pass must hold whole pairs of tile rows");
        constexpr int PR = RP / 4;                     // pooled pixels of the pass: RP / 32 rows of 8
        for (int idx = tid; idx < PR * CG; idx += NT) {
          const int pp = idx / CG, cgp = idx - pp * CG;
          const int pyl = pp >> 3, pxl = pp & 7;
          const int c0 = c_base + cgp * 8;
          const long long po = pixfn.pool(ps * (RP / 32) + pyl, pxl);
          if (po < 0 || c0 >= a.Cout) continue;
          const int r00 = pyl * 32 + pxl * 2;
          float m[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {               // max of the raw accumulators first: bias, activation and rounding are monotonic
            const int rr = r00 + (k >> 1) * 16 + (k & 1);
            const float4 v0 = *reinterpret_cast<const float4*>(smem + rr * PITCH + cgp * 32);
            const float4 v1 = *reinterpret_cast<const float4*>(smem + rr * PITCH + cgp * 32 + 16);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = k == 0 ? v[e] : fmaxf(m[e], v[e]);
          }
          if (a.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.bias + c0), b1 = *reinterpret_cast<const float4*>(a.bias + c0 + 4);
            m[0] += b0.x; m[1] += b0.y; m[2] += b0.z; m[3] += b0.w; m[4] += b1.x; m[5] += b1.y; m[6] += b1.z; m[7] += b1.w;
          }
          if (a.out_act != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = act_apply(a.out_act, m[e]);
          }
          if (sizeof(T) == 2) *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(a.pool_out) + po + c0) = Elem<bf16>::pack(m);
          else {
            float* o = reinterpret_cast<float*>(a.pool_out) + po + c0;
            reinterpret_cast<float4*>(o)[0] = make_float4(m[0], m[1], m[2], m[3]);
            reinterpret_cast<float4*>(o)[1] = make_float4(m[4], m[5], m[6], m[7]);
          }
        }
      }
    }
  }
  if constexpr (STATS == 2) {
    __syncthreads();                                   // the staged tile is dead: reuse its LDS
    float* red = reinterpret_cast<float*>(smem);       // [NT / CG][2][BC]
    {
      float* r0 = red + (tid / CG) * 2 * BC + (tid % CG) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) { r0[e] = gs0[e]; r0[BC + e] = gs1[e]; }
    }
    __syncthreads();
    const int grp = pt / a.bn_tpg, chunk = cls * a.bn_tpg + (pt - grp * a.bn_tpg);
    for (int j = tid; j < 2 * BC; j += NT) {
      float t = 0.f;
      for (int m = 0; m < NT / CG; ++m) t += red[m * 2 * BC + j];
      int c = c_base + (j % BC), ct = a.Cout;
      double* part = a.bn_part;
      if constexpr (DUAL) {
        ct = a.split_c;
        if (c >= a.split_c) { part = a.bn_part2; c -= a.split_c; }
      }
      if (part && c < ct) part[((size_t)(grp * a.bn_nchunk + chunk) * 2 + j / BC) * ct + c] = (double)t;
    }
  }
  if constexpr (STATS == 1) {
    __syncthreads();                                   // the staged tile is dead: reuse its LDS
    float* red = reinterpret_cast<float*>(smem);       // [NT / BC][2][BC]
    red[(tid / BC) * 2 * BC + tid % BC] = bsum;
    red[(tid / BC) * 2 * BC + BC + tid % BC] = bsq;
    __syncthreads();
    if (tid < 2 * BC || (NT < 2 * BC && tid < BC)) {
      const int grp = pt / a.bn_tpg, chunk = cls * a.bn_tpg + (pt - grp * a.bn_tpg);
      for (int j = tid; j < 2 * BC; j += NT) {
        float t = 0.f;
        for (int m = 0; m < NT / BC; ++m) t += red[m * 2 * BC + j];
        const int c = c_base + (j % BC);
        if (c < a.Cout) a.bn_part[((size_t)(grp * a.bn_nchunk + chunk) * 2 + j / BC) * a.Cout + c] = (double)t;
      }
    }
  }
}

}  // namespace vp
