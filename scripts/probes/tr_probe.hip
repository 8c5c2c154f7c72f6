// Probe of ds_read_b64_tr_b16 semantics on gfx950: hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint16_t* out, int mode) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = lane * 8;                       // lane-linear 8-byte slots
  else addr = (lane & 15) * 32 + (lane >> 4) * 8;        // row = lane&15 (32-byte pitch), 8-byte column group = lane>>4
  addr += (unsigned)(uintptr_t)lds;
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[lane * 4 + 0] = v.x & 0xffff; out[lane * 4 + 1] = v.x >> 16; out[lane * 4 + 2] = v.y & 0xffff; out[lane * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (element index each lane received; lane address = %s)\n", mode, mode ? "(l&15)*32 + (l>>4)*8 bytes" : "l*8 bytes");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 4 == 3) ? "\n" : "   ");
  }
  return 0;
}
