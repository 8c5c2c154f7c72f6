// GRU recurrence (tf.contrib.rnn.GRUCell under dynamic_rnn, 256 units) shared by the inference executor (audio_kernels.hip) and the
// training step (bfm_train.hip).  One block of 1024 threads per sequence.  The recurrence is serial in time and every step streams the
// 768 KB of recurrent weights through ONE compute unit, so what a step costs is how many loads that block keeps in flight: the
// round-1/2 form (256 threads, thread = unit, 768 dependent fma + loads per thread and step) ran 12 us per step.  Here the k range of
// every matrix-vector product is split over the block - gates: 512 columns x 2 halves of k, candidate: 256 columns x 4 quarters - each
// thread runs four independent accumulators, and the partial sums meet in LDS: 192 loads per thread and step instead of 768.
//   r, u = sigmoid(xg + h . whg)        whg [256][512]  (row = h unit; columns: r then u)
//   c    = tanh(xc + (r * h) . whc)     whc [256][256]
//   h'   = u * h + (1 - u) * c          outputs past seq_len are zero, the state is frozen there
#pragma once
#include <hip/hip_runtime.h>

namespace vp {

// s = sum_{k < n} v[k] * w[k * ld], four independent chains (k ascending inside each), combined in a fixed order
template <int N>
__device__ __forceinline__ float gru_dot(const float* __restrict__ v, const float* __restrict__ w, int ld) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
  for (int k = 0; k < N; k += 4) {
    s0 = fmaf(v[k], w[(size_t)k * ld], s0);
    s1 = fmaf(v[k + 1], w[(size_t)(k + 1) * ld], s1);
    s2 = fmaf(v[k + 2], w[(size_t)(k + 2) * ld], s2);
    s3 = fmaf(v[k + 3], w[(size_t)(k + 3) * ld], s3);
  }
  return (s0 + s1) + (s2 + s3);
}

// TRAIN: also saves r, u, c and h_prev per step (what the backward pass needs)
template <bool TRAIN>
__global__ __launch_bounds__(1024) void gru_fwd_kernel(const float* __restrict__ xg, const float* __restrict__ xc, const float* __restrict__ whg,
                                                       const float* __restrict__ whc, const int* __restrict__ seq_len, float* __restrict__ out,
                                                       float* __restrict__ sr, float* __restrict__ su, float* __restrict__ sc, float* __restrict__ shp,
                                                       int T) {
  __shared__ float h[256], rh[256], part[1024];
  const int b = blockIdx.x, tid = threadIdx.x, n = seq_len[b];
  if (tid < 256) h[tid] = 0.f;
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const size_t o = (size_t)b * T + t;
    if (t >= n) {                                    // (block-uniform) past the end: zero output, frozen state
      if (tid < 256) {
        out[o * 256 + tid] = 0.f;
        if (TRAIN) { sr[o * 256 + tid] = 0.f; su[o * 256 + tid] = 0.f; sc[o * 256 + tid] = 0.f; shp[o * 256 + tid] = h[tid]; }
      }
      continue;
    }
    {   // gates: column c of 512, half kh of the 256 h units
      const int c = tid & 511, kh = tid >> 9;
      part[tid] = gru_dot<128>(h + kh * 128, whg + (size_t)(kh * 128) * 512 + c, 512);
    }
    __syncthreads();
    float r = 0.f, u = 0.f;
    if (tid < 256) {
      const float ar = xg[o * 512 + tid] + (part[tid] + part[512 + tid]);
      const float au = xg[o * 512 + 256 + tid] + (part[256 + tid] + part[768 + tid]);
      r = 1.f / (1.f + expf(-ar)); u = 1.f / (1.f + expf(-au));
      rh[tid] = r * h[tid];
    }
    __syncthreads();
    {   // candidate: column c of 256, quarter kq of the 256 (r * h) values
      const int c = tid & 255, kq = tid >> 8;
      part[tid] = gru_dot<64>(rh + kq * 64, whc + (size_t)(kq * 64) * 256 + c, 256);
    }
    __syncthreads();
    float hn = 0.f;
    if (tid < 256) {
      const float ac = xc[o * 256 + tid] + ((part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid]));
      const float c = tanhf(ac), hp = h[tid];
      hn = u * hp + (1.f - u) * c;
      if (TRAIN) { sr[o * 256 + tid] = r; su[o * 256 + tid] = u; sc[o * 256 + tid] = c; shp[o * 256 + tid] = hp; }
      out[o * 256 + tid] = hn;
    }
    __syncthreads();                                 // every read of h (gate products, r * h, h_prev) is done
    if (tid < 256) h[tid] = hn;
    __syncthreads();
  }
}

// Backward through time to the gate / candidate pre-activations.  whgT [512][256], whcT [256][256]: the recurrent kernels TRANSPOSED
// (row = gate / candidate column, column = h unit), so that thread j's products with row j of the forward kernels read coalesced
// columns - the round-2 form walked row j per thread (64 cache lines per load instruction) and ran 21 us per step.
static __global__ __launch_bounds__(1024) void gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ whgT, const float* __restrict__ whcT,
                                                       const int* __restrict__ seq_len, const float* __restrict__ sr, const float* __restrict__ su,
                                                       const float* __restrict__ sc, const float* __restrict__ shp, float* __restrict__ dag,
                                                       float* __restrict__ dac, int T) {
  __shared__ float s_dag[512], s_dac[256], part[1024];
  const int b = blockIdx.x, tid = threadIdx.x, n = seq_len[b];
  float dh = 0.f;                                    // d loss / d h_t carried backwards (unit tid < 256)
  for (int t = T - 1; t >= 0; --t) {
    const size_t o = (size_t)b * T + t;
    if (t >= n) {                                    // frozen state: dh passes through
      if (tid < 256) { dag[o * 512 + tid] = 0.f; dag[o * 512 + 256 + tid] = 0.f; dac[o * 256 + tid] = 0.f; }
      continue;
    }
    float r = 0.f, u = 0.f, hp = 0.f, d_u = 0.f, dhp = 0.f, d_ac = 0.f;
    if (tid < 256) {
      r = sr[o * 256 + tid]; u = su[o * 256 + tid]; hp = shp[o * 256 + tid];
      const float c = sc[o * 256 + tid];
      const float g = dout[o * 256 + tid] + dh;
      d_u = g * (hp - c);
      const float d_c = g * (1.f - u);
      dhp = g * u;
      d_ac = d_c * (1.f - c * c);
      s_dac[tid] = d_ac;
    }
    __syncthreads();
    {   // d / d (r * h_prev)[j] = sum_m whc[j][m] * d_ac[m] = sum_m whcT[m][j] * d_ac[m]: unit j, quarter q of m
      const int j = tid & 255, q = tid >> 8;
      part[tid] = gru_dot<64>(s_dac + q * 64, whcT + (size_t)(q * 64) * 256 + j, 256);
    }
    __syncthreads();
    float d_ar = 0.f, d_au = 0.f;
    if (tid < 256) {
      const float d_rh = (part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid]);
      const float d_r = d_rh * hp;
      dhp = fmaf(d_rh, r, dhp);
      d_ar = d_r * r * (1.f - r); d_au = d_u * u * (1.f - u);
      s_dag[tid] = d_ar; s_dag[256 + tid] = d_au;
    }
    __syncthreads();
    {   // dh_prev[j] += sum_{m < 512} whg[j][m] * d_ag[m] = sum_m whgT[m][j] * d_ag[m]: unit j, quarter q of m
      const int j = tid & 255, q = tid >> 8;
      part[tid] = gru_dot<128>(s_dag + q * 128, whgT + (size_t)(q * 128) * 256 + j, 256);
    }
    __syncthreads();
    if (tid < 256) {
      dhp += (part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid]);
      dag[o * 512 + tid] = d_ar; dag[o * 512 + 256 + tid] = d_au; dac[o * 256 + tid] = d_ac;
      dh = dhp;
    }
    __syncthreads();                                 // s_dac / s_dag / part are rewritten by the next step
  }
}

}  // namespace vp
