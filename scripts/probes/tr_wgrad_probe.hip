// Probe for the transpose-read weight-gradient kernel: LDS image [32 pixels][128 channels] bf16 (256-byte rows, 16-byte pieces
// XOR-swizzled by the pixel), fragments by ds_read_b64_tr_b16 pairs, one MFMA 16x16x32: D[m][n] = sum_k A[k][m] * B[k][n].
//   hipcc --offload-arch=gfx950 -O3 tr_wgrad_probe.hip -o tr_wgrad_probe && ./tr_wgrad_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __host__ inline int swz(int p) { return ((p & 3) | (((p >> 3) & 1) << 2)) << 1; }
__device__ inline uint2 tr_read(unsigned addr) {
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__global__ void k(const uint16_t* a, const uint16_t* b, float* out, int mblk, int nblk) {
  __shared__ __attribute__((aligned(16))) uint16_t la[32 * 128], lb[32 * 128];
  // fill with the swizzled image: piece c16 of pixel p at slot c16 ^ swz(p)
  for (int i = threadIdx.x; i < 32 * 16; i += 64) {
    const int p = i >> 4, c16 = i & 15;
    for (int e = 0; e < 8; ++e) {
      la[p * 128 + ((c16 ^ swz(p)) << 3) + e] = a[p * 128 + c16 * 8 + e];
      lb[p * 128 + ((c16 ^ swz(p)) << 3) + e] = b[p * 128 + c16 * 8 + e];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, t = lane & 15;
  const int p = 8 * g + (t >> 2), q = t & 3;
  auto frag = [&](const uint16_t* base, int blk) -> bf16x8 {
    const int c16 = 2 * blk + (q >> 1);
    const unsigned addr = (unsigned)(uintptr_t)base + p * 256 + ((c16 ^ swz(p)) << 4) + (q & 1) * 8;
    const uint2 lo = tr_read(addr), hi = tr_read(addr + 4 * 256);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    union { uint4 u; bf16x8 v; } f;
    f.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
    return f.v;
  };
  const bf16x8 fa = frag(la, mblk), fb = frag(lb, nblk);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
  // D layout: col = lane & 15 (B index n), row = 4 * (lane >> 4) + reg (A index m)
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + t] = acc[r];
}
__host__ static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
__host__ static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  static uint16_t ha[32 * 128], hb[32 * 128];
  srand(1);
  for (int i = 0; i < 32 * 128; ++i) { ha[i] = f2bf((rand() % 17 - 8) / 8.f); hb[i] = f2bf((rand() % 13 - 6) / 4.f); }
  uint16_t *da, *db; float* dout;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, 256 * 4);
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  int bad = 0;
  for (int mblk = 0; mblk < 8; mblk += 3) for (int nblk = 1; nblk < 8; nblk += 5) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout, mblk, nblk);
    float h[256];
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
      float ref = 0;
      for (int kk = 0; kk < 32; ++kk) ref += bf2f(ha[kk * 128 + mblk * 16 + m]) * bf2f(hb[kk * 128 + nblk * 16 + n]);
      if (fabsf(ref - h[m * 16 + n]) > 1e-3f) { if (bad < 5) printf("mblk %d nblk %d m %d n %d: got %f want %f\n", mblk, nblk, m, n, h[m * 16 + n], ref); ++bad; }
    }
  }
  printf(bad ? "FAILED: %d mismatches\n" : "tr_wgrad_probe OK (%d)\n", bad);
  return bad != 0;
}
