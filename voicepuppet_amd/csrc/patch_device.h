// Device helpers shared by the unrolled patch kernels (conv_patch3.hip: 3x3 stride 1; conv_patch2.hip: the 2x2-tap parity classes
// of the 4x4 stride-2 transposed convolutions): inline-asm LDS fragment reads with immediate offsets, compile-time vmcnt, unroller.
#pragma once
#include <utility>

#include "igemm_device.h"

namespace vp {

// Fragment reads as inline asm.  hipcc drains vmcnt in front of every LDS load it can see while an LDS-DMA is pending (the
// __restrict__ route of conv_patch.hip loses its alias scopes in this fully unrolled form), which would serialise the DMA stream;
// the ring discipline of the loop - counted vmcnt + barrier - is what orders these reads.  The destination registers are only
// valid behind lds_fence() (s_waitcnt lgkmcnt(0) + a dependency on every register, so that no consumer is scheduled above it).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int IMM> __device__ __forceinline__ u32x2 lds_rd64(int addr) {
  u32x2 r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM));
  return r;
}
template <int IMM> __device__ __forceinline__ u32x4 lds_rd128(int addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM));
  return r;
}
// second half of a B piece: address ^ 8, formed inside the asm so that it never occupies a register across steps
template <int IMM> __device__ __forceinline__ u32x2 lds_rd64_x8(int addr) {
  u32x2 r;
  int t;
  asm volatile("v_xor_b32 %1, 8, %2\n\tds_read_b64 %0, %1 offset:%3" : "=v"(r), "=&v"(t) : "v"(addr), "n"(IMM));
  return r;
}
// B fragments of the step + the A fragments [A0, A0 + NA) of the wave, valid on return
template <int NA, int TP, int AIMM, int BIMM, bool WITH_B>
__device__ __forceinline__ void patch3_frag_read(int aaddr, const int (&b0)[TP], uint4 (&fa)[NA], uint4 (&fb)[TP]) {
  u32x4 ra[NA];
  u32x2 rl[TP], rh[TP];
  if constexpr (WITH_B) {
#pragma unroll
    for (int t = 0; t < TP; ++t) { rl[t] = lds_rd64<BIMM>(b0[t]); rh[t] = lds_rd64_x8<BIMM>(b0[t]); }
  }
  static_assert(NA == 2 || NA == 4, "weight blocks per read batch");
  ra[0] = lds_rd128<AIMM>(aaddr); ra[1] = lds_rd128<AIMM + 1024>(aaddr);
  if constexpr (NA >= 4) { ra[2] = lds_rd128<AIMM + 2048>(aaddr); ra[3] = lds_rd128<AIMM + 3072>(aaddr); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (WITH_B) {
#pragma unroll
    for (int t = 0; t < TP; ++t) { asm volatile("" : "+v"(rl[t])); asm volatile("" : "+v"(rh[t])); fb[t] = make_uint4(rl[t].x, rl[t].y, rh[t].x, rh[t].y); }
  }
#pragma unroll
  for (int t = 0; t < NA; ++t) { asm volatile("" : "+v"(ra[t])); fa[t] = make_uint4(ra[t].x, ra[t].y, ra[t].z, ra[t].w); }
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <typename F, int... Is> __device__ __forceinline__ void static_steps(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// The MFMAs of one step (TC weight blocks x TP pixel blocks per wave; stage / buffer / tap offsets are the immediates AIMM, BIMM).
// Weight blocks are read in batches of NA (register budget).  Two batches: rolling form - a block's registers are refilled with
// block + NA as soon as its MFMAs are issued, so the second batch lands under the first batch's MFMAs (VP_P3_NO_ROLL: batch by batch).  ROLL: the 3x3 kernel (128x256 tile
// 940 -> 970 TF); the 2x2-tap kernel measured 1.5 % slower with it
template <typename T, int TC, int TP, int NA, int AIMM, int BIMM, bool ROLL>
__device__ __forceinline__ void patch_step_mma(int aaddr, const int (&b0)[TP], f32x4 (&acc)[TC][TP]) {
  uint4 fb[TP];
#ifndef VP_P3_NO_ROLL
  if constexpr (ROLL && TC / NA == 2) {
    uint4 fa[NA];
    u32x4 rn[NA];
    patch3_frag_read<NA, TP, AIMM, BIMM, true>(aaddr, b0, fa, fb);
    static_steps([&](auto tci) {
      constexpr int tc = decltype(tci)::value;
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) acc[tc][tp] = mma16<T>(fa[tc], fb[tp], acc[tc][tp]);
      __builtin_amdgcn_sched_barrier(0);
      rn[tc] = lds_rd128<AIMM + (NA + tc) * 1024>(aaddr);
      __builtin_amdgcn_sched_barrier(0);
    }, std::make_integer_sequence<int, NA>{});
    static_steps([&](auto tci) {
      constexpr int tc = decltype(tci)::value;
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NA - 1 - tc) : "memory");
      asm volatile("" : "+v"(rn[tc]));
      const uint4 f = make_uint4(rn[tc].x, rn[tc].y, rn[tc].z, rn[tc].w);
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) acc[NA + tc][tp] = mma16<T>(f, fb[tp], acc[NA + tc][tp]);
      __builtin_amdgcn_sched_barrier(0);
    }, std::make_integer_sequence<int, NA>{});
    return;
  }
#endif
  static_steps([&](auto hi) {
    constexpr int h = decltype(hi)::value;
    uint4 fa[NA];
    patch3_frag_read<NA, TP, AIMM + h * NA * 1024, BIMM, h == 0>(aaddr, b0, fa, fb);
#pragma unroll
    for (int tc = 0; tc < NA; ++tc)
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) acc[h * NA + tc][tp] = mma16<T>(fa[tc], fb[tp], acc[h * NA + tc][tp]);
  }, std::make_integer_sequence<int, TC / NA>{});
}


}  // namespace vp
