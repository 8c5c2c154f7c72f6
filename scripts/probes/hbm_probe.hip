// What the HBM of this box delivers to plain streaming kernels - the ceiling the thin (8 <-> 64-channel) layers are measured against
// (EXPERIMENTS.md, round 6: VERDICT r5 item 5 asks for >= 4.5 TB/s on layers that mostly WRITE): read-only, write-only (plain and
// non-temporal stores), copy and the 1 : 8 read-to-write mix of VGG conv1_1 (67 MB in, 537 MB out), 512 MB per pass.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/hbm_probe.hip -o scripts/probes/hbm_probe && scripts/probes/hbm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_read(const uint4* __restrict__ a, uint4* sink, size_t n) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = a[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if (acc.x == 0x12345678u) sink[0] = acc;
}
__global__ void k_write(uint4* __restrict__ b, size_t n) {
  const uint4 v = make_uint4(threadIdx.x, 1, 2, 3);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = v;
}
__global__ void k_write_nt(uint4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned* p = reinterpret_cast<unsigned*>(b + i);
    __builtin_nontemporal_store(threadIdx.x, p); __builtin_nontemporal_store(1u, p + 1); __builtin_nontemporal_store(2u, p + 2); __builtin_nontemporal_store(3u, p + 3);
  }
}
__global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// one 16-byte read per eight 16-byte writes (an 8-channel pixel in, a 64-channel pixel out)
__global__ void k_1to8(const uint4* __restrict__ a, uint4* __restrict__ b, size_t npix) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix * 8; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = a[i >> 3];
    b[i] = make_uint4(v.x + (unsigned)(i & 7), v.y, v.z, v.w);
  }
}

int main() {
  const size_t bytes = (size_t)512 << 20, n = bytes / 16;
  uint4 *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes + 64);
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, double moved, auto launch) {
    float best = 1e9f;
    for (int r = 0; r < 10; ++r) {
      hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-34s %8.1f us  %7.2f TB/s\n", name, best * 1e3, moved / (best * 1e-3) / 1e12);
  };
  for (int blocks : {2048, 8192}) {
    printf("grid %d x 256\n", blocks);
    timeit("read 512 MB", (double)bytes, [&] { k_read<<<blocks, 256>>>(a, b + n, n); });
    timeit("write 512 MB", (double)bytes, [&] { k_write<<<blocks, 256>>>(b, n); });
    timeit("write 512 MB, non-temporal", (double)bytes, [&] { k_write_nt<<<blocks, 256>>>(b, n); });
    timeit("copy 512 MB -> 512 MB", 2.0 * bytes, [&] { k_copy<<<blocks, 256>>>(a, b, n); });
    timeit("64 MB in, 512 MB out (1 : 8)", bytes * 1.125, [&] { k_1to8<<<blocks, 256>>>(a, b, n / 8); });
  }
  return 0;
}
