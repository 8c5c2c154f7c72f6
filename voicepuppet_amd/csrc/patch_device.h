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

}  // namespace vp
