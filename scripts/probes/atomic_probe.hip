// Contention of order-independent (64-bit fixed-point) atomic accumulation of batch-norm partial sums, the alternative to the partial-row +
// finalize-launch scheme (EXPERIMENTS.md, round 6; VERDICT r5 item 2 asked for the measurement): `rows` blocks each add their 2*C partial
// sums to the SAME 2*C totals.  Compared with what it would replace: one bn_finalize launch (5-6 us at any batch size).
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/atomic_probe.hip -o scripts/probes/atomic_probe && scripts/probes/atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void accumulate(unsigned long long* tot, int twoC, int work_iters) {
  // stand-in for the conv tile in front of the epilogue: a little ALU time so that blocks do not arrive in lockstep
  float x = threadIdx.x;
  for (int i = 0; i < work_iters; ++i) x = x * 1.0001f + 0.5f;
  for (int t = threadIdx.x; t < twoC; t += blockDim.x) atomicAdd(&tot[t], (unsigned long long)(1 + (x < 0.f)));
}
__global__ void rows_only(double* part, int twoC, int work_iters) {
  float x = threadIdx.x;
  for (int i = 0; i < work_iters; ++i) x = x * 1.0001f + 0.5f;
  for (int t = threadIdx.x; t < twoC; t += blockDim.x) part[(size_t)blockIdx.x * twoC + t] = (double)(1 + (x < 0.f));
}

int main() {
  unsigned long long* tot; double* part;
  hipMalloc(&tot, 4096 * 8); hipMalloc(&part, (size_t)8192 * 1024 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int cases[][2] = {{64, 128}, {256, 128}, {512, 128}, {3072, 128}, {1536, 256}, {384, 512}, {4096, 64}, {8192, 64}};
  printf("%8s %6s %12s %12s %10s\n", "rows", "C", "atomics us", "rows us", "atomics");
  for (auto& c : cases) {
    const int rows = c[0], twoC = 2 * c[1];
    float best[2] = {1e9f, 1e9f};
    for (int v = 0; v < 2; ++v)
      for (int rep = 0; rep < 20; ++rep) {
        hipMemset(tot, 0, 4096 * 8);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (v == 0) accumulate<<<rows, 256>>>(tot, twoC, 2000);
        else rows_only<<<rows, 256>>>(part, twoC, 2000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best[v]) best[v] = ms;
      }
    unsigned long long h0; hipMemcpy(&h0, tot, 8, hipMemcpyDeviceToHost);
    printf("%8d %6d %12.2f %12.2f %10d   (total[0] = %llu)\n", rows, c[1], best[0] * 1e3, best[1] * 1e3, rows * twoC, h0);
  }
  return 0;
}
