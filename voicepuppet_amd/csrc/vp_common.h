// Shared device helpers for the voicepuppet gfx950 kernels.
// Written for CDNA4 only: 64-wide wavefronts, MFMA 16x16 tiles, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vp {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

enum Act { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4, ACT_RELU6 = 5, ACT_LEAKY = 6 };

__device__ __forceinline__ float act_apply(int act, float x) {
  // lrelu(x, 0.2) = 0.6*x + 0.4*|x|   (pixrefer.py:88-97)
  if (act == ACT_LRELU) return 0.6f * x + 0.4f * fabsf(x);
  if (act == ACT_RELU) return fmaxf(x, 0.f);
  if (act == ACT_TANH) return tanhf(x);
  if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-x));
  if (act == ACT_RELU6) return fminf(fmaxf(x, 0.f), 6.f);          // tinynet.py:9
  if (act == ACT_LEAKY) return fmaxf(x, 0.2f * x);                  // tf.nn.leaky_relu (bfmnet.py:199)
  return x;
}

// derivative of the activation w.r.t. its argument z (tf.abs'(0) = 0, relu'(0) = 0)
__device__ __forceinline__ float act_grad(int act, float z) {
  if (act == ACT_LRELU) return z > 0.f ? 1.0f : (z < 0.f ? 0.2f : 0.6f);
  if (act == ACT_RELU) return z > 0.f ? 1.f : 0.f;
  return 1.f;
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int E = 4;  // elements per 16-byte piece
  __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t hi16) { return __uint_as_float(hi16 << 16); }
#ifdef VP_SW_BF16_CVT   // A/B build only (make variant VAR=swcvt DEFS=-DVP_SW_BF16_CVT): the integer form round 1 used
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
#else
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  // round-to-nearest-even (NaN stays NaN): the hardware convert (v_cvt_pk_bf16_f32 on gfx950) instead of seven integer instructions -
  // the batch-norm statistics epilogue rounds every value it sums this way
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = {f, 0.f};
  const bf2 h = __builtin_convertvector(v, bf2);
  return *reinterpret_cast<const uint32_t*>(&h) & 0xffffu;
}
#endif

template <> struct Elem<bf16> {
  static constexpr int E = 8;
  __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
  // two floats -> packed bf16 pair, round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    const bf2 h = __builtin_convertvector(v, bf2);
    return *reinterpret_cast<const uint32_t*>(&h);
  }
  __device__ static __forceinline__ uint4 pack(const float* f) {
    uint4 v;
    v.x = pack2(f[0], f[1]); v.y = pack2(f[2], f[3]); v.z = pack2(f[4], f[5]); v.w = pack2(f[6], f[7]);
    return v;
  }
  __device__ static __forceinline__ float ld(const bf16* p) { return bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(p)); }
  __device__ static __forceinline__ void st(bf16* p, float v) { *reinterpret_cast<uint16_t*>(p) = (uint16_t)f32_to_bf16_bits(v); }
};

// One 16-byte A piece x one 16-byte B piece -> 16x16 f32 accumulator tile.
//   float: four v_mfma_f32_16x16x4_f32 (lane group g supplies k = 4g+j to the j-th MFMA; the k
//          permutation is the same for A and B, so the sum over k is unchanged)
//   bf16 : one v_mfma_f32_16x16x32_bf16
template <typename T> __device__ __forceinline__ f32x4 mma16(const uint4& a, const uint4& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mma16<float>(const uint4& a, const uint4& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  return c;
}
template <> __device__ __forceinline__ f32x4 mma16<bf16>(const uint4& a, const uint4& b, f32x4 c) {
  union { uint4 u; bf16x8 v; } ua, ub;
  ua.u = a; ub.u = b;
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, c, 0, 0, 0);
}

// Row permutation of the packed weights (IgemmArgs::rowperm).  Inside a 64-row block, channel c = 32*hi + 8*q + 4*lo + e sits in
// MFMA tile t = 2*hi + lo at row 4q + e: after its four tiles a lane (accumulator rows 4q..4q+3) holds channels 8q..8q+7 and
// 32+8q..32+8q+7 of its pixel - two 16-byte runs instead of four 8-byte ones.
__host__ __device__ __forceinline__ int perm_row(int c) {
  return (c & ~63) | ((((c >> 5) & 1) * 2 + ((c >> 2) & 1)) << 4) | (((c >> 3) & 3) << 2) | (c & 3);
}
// first of the 4 consecutive channels lane group q holds for 16-row tile T of a block (T counted from the block's first channel)
__device__ __forceinline__ int tile_chan0(int rowperm, int T, int q) {
  return rowperm ? ((T >> 2) << 6) + (((T >> 1) & 1) << 5) + (q << 3) + ((T & 1) << 2) : T * 16 + 4 * q;
}

// wave64 butterfly sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace vp
