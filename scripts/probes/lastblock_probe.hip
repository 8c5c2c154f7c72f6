// What would folding the batch-norm finalize launch into the epilogue of the launch that writes the partial rows buy on a dependency
// chain?  (the finalize kernels are 5 us each on the few-frame step's chain, 31 per step.)
// A chain of `L` layers, each: a producer with `rows` blocks (every block stores a 32 KB tile and its partial row [2][C] doubles) ->
// finalize (mean / rstd per channel from the rows) -> a consumer that reads the coefficients.  Two forms:
//   separate   producer, bn_finalize-like kernel (one wave per channel, lanes over rows), consumer: three launches per layer
//   lastblock  the producer's blocks publish their row (__threadfence + a device-scope counter); the block that arrives last sums the rows
//              itself (threadFenceReduction pattern: release by every block, acquire by the last one - on this chip the L2s of the eight
//              XCDs are not coherent with each other, so the fences are L2 write-backs / invalidates): two launches per layer
// Printed: microseconds per layer for both, and whether the coefficients agree.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/lastblock_probe.hip -o scripts/probes/lastblock_probe && scripts/probes/lastblock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int LAST>
__global__ __launch_bounds__(256) void producer(float* tile, double* part, int C, float* coef, unsigned* counter, const float* prev_coef) {
  // the tile: 32 KB per block, values depend on the previous layer's coefficients (keeps the chain a chain)
  const float s = prev_coef ? prev_coef[threadIdx.x % C] : 1.f;
  float4 v = make_float4(s + threadIdx.x, s, s * 0.5f, 1.f);
  float4* t = reinterpret_cast<float4*>(tile + (size_t)blockIdx.x * 8192);
  for (int i = threadIdx.x; i < 2048; i += 256) t[i] = v;
  for (int c = threadIdx.x; c < C; c += 256) {
    part[((size_t)blockIdx.x * 2) * C + c] = (double)(c + 1) * (1.0 + 1e-3 * (blockIdx.x & 7));
    part[((size_t)blockIdx.x * 2 + 1) * C + c] = (double)(c + 1) * (c + 1) * 1.5;
  }
  if (LAST == 0) return;
  __shared__ unsigned last;
  __threadfence();                                    // release: this block's row is visible device-wide before its ticket
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();                                    // acquire: the other blocks' rows
  const int rows = gridDim.x;
  if (LAST == 2) {                                    // (fences + ticket only: what the publication itself costs)
    if (threadIdx.x == 0) *counter = 0;
    return;
  }
  // the whole block on the rows: thread = (channel c of a 64-channel slice, row phase r of 256 / 64 = 4), 8 rows x 2 sums in flight per
  // thread (coalesced 512-byte reads per wave), folded over the four waves through LDS
  __shared__ double sm[2][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    double s0 = 0, s1 = 0;
    for (int k = wv; k < rows; k += 4 * 8) {
      double x0[8], x1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = k + 4 * u;
        const bool ok = kk < rows && c < C;
        x0[u] = ok ? __builtin_nontemporal_load(&part[((size_t)kk * 2) * C + c]) : 0.0;
        x1[u] = ok ? __builtin_nontemporal_load(&part[((size_t)kk * 2 + 1) * C + c]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += x0[u]; s1 += x1[u]; }
    }
    __syncthreads();
    sm[0][wv][lane] = s0; sm[1][wv][lane] = s1;
    __syncthreads();
    if (wv == 0 && c < C) {
      const double t0 = (sm[0][0][lane] + sm[0][1][lane]) + (sm[0][2][lane] + sm[0][3][lane]);
      const double t1 = (sm[1][0][lane] + sm[1][1][lane]) + (sm[1][2][lane] + sm[1][3][lane]);
      const double m = t0 / rows;
      coef[c] = (float)m; coef[C + c] = (float)(1.0 / sqrt(t1 / rows - m * m + 1e3));
    }
  }
  if (threadIdx.x == 0) *counter = 0;                 // ready for the next launch
}

// the separate finalize launch: one wave per channel, lanes over rows (bn_finalize_kernel's scheme)
template <bool LAST>
__global__ __launch_bounds__(256) void finalize(const double* part, int rows, int C, float* coef) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double s0 = 0, s1 = 0;
  for (int k = lane; k < rows; k += 64) { s0 += part[((size_t)k * 2) * C + c]; s1 += part[((size_t)k * 2 + 1) * C + c]; }
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  if (lane == 0) { const double m = s0 / rows; coef[c] = (float)m; coef[C + c] = (float)(1.0 / sqrt(s1 / rows - m * m + 1e3)); }
}

__global__ void consumer(const float* coef, int C, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = coef[i % C] * coef[C + i % C];
}

int main() {
  const int L = 40;
  float *tile, *coef[2], *out; double* part; unsigned* counter;
  hipMalloc(&tile, (size_t)4096 * 8192 * 4); hipMalloc(&part, (size_t)4096 * 2 * 512 * 8);
  hipMalloc(&coef[0], 2 * 512 * 4); hipMalloc(&coef[1], 2 * 512 * 4); hipMalloc(&out, 65536 * 4); hipMalloc(&counter, 4);
  hipMemset(counter, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int cases[][2] = {{16, 512}, {64, 512}, {128, 256}, {256, 128}, {512, 128}, {1024, 128}, {2048, 64}, {4096, 64}};
  printf("%8s %6s %18s %18s %22s %8s\n", "rows", "C", "separate us/layer", "lastblock us/layer", "fences+ticket only", "agree");
  for (auto& cs : cases) {
    const int rows = cs[0], C = cs[1];
    float best[3] = {1e9f, 1e9f, 1e9f};
    std::vector<float> h[3];
    for (int v = 0; v < 3; ++v) {
      for (int rep = 0; rep < 8; ++rep) {
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int l = 0; l < L; ++l) {
          float* cf = coef[l & 1];
          const float* prev = l ? coef[(l - 1) & 1] : nullptr;
          if (v == 0) {
            producer<0><<<rows, 256>>>(tile, part, C, cf, counter, prev);
            finalize<false><<<(C + 3) / 4, 256>>>(part, rows, C, cf);
          } else if (v == 1) {
            producer<1><<<rows, 256>>>(tile, part, C, cf, counter, prev);
          } else {
            producer<2><<<rows, 256>>>(tile, part, C, cf, counter, prev);
          }
          consumer<<<64, 256>>>(cf, C, out);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best[v]) best[v] = ms;
      }
      h[v].resize(2 * C);
      hipMemcpy(h[v].data(), coef[(L - 1) & 1], 2 * C * 4, hipMemcpyDeviceToHost);
    }
    bool same = true;
    for (int i = 0; i < 2 * C; ++i) same = same && h[0][i] == h[1][i];
    printf("%8d %6d %18.2f %18.2f %22.2f %8s\n", rows, C, best[0] * 1e3 / L, best[1] * 1e3 / L, best[2] * 1e3 / L, same ? "yes" : "NO");
  }
  return 0;
}
