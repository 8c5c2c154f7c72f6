// Register-double-buffered LDS-DMA implicit GEMM for the 128-accumulator tiles (own translation unit: it is the kernel under tuning).
#include <stdlib.h>

#include "igemm_device.h"
#include "launch.h"

namespace vp {

// ------------------------------------------------------------------------------------------------
// igemm_db_kernel: the LDS-DMA GEMM with REGISTER-DOUBLE-BUFFERED fragments, for the big tiles (256 ch x 256 px, 128 ch x 512 px;
// 8 waves = two per SIMD, 128 accumulator registers each, one block per CU).  In igemm_dma_kernel / igemm_ws_kernel every K chunk is
// barrier -> ds_read -> MFMA inside each wave, so the matrix pipe idles through every chunk's LDS latency and all waves read at once.
// Here the fragments of chunk k+1 are read (into the second register set) BEFORE the MFMAs of chunk k issue, so the MFMAs never wait
// for LDS and a wave reaches the next barrier with its reads long complete:
//   iteration k:  vmcnt (own DMAs of chunk k+1 landed) ; lgkmcnt(0) (own reads of chunk k complete) ; barrier
//                 issue DMA of chunk k+NST into the stage chunk k used (every wave finished reading it before the barrier)
//                 ds_read fragments of chunk k+1 -> nxt ; 32 MFMAs on cur ; swap
// Chunk k lives in registers while the ring holds chunks k+1 .. k+NST: NST-1 chunks stay in flight across each barrier.
// fastk operands only (scalar K stepping, hardware zero fill), no K split; epilogue = the staged 16-byte row stores.
// ------------------------------------------------------------------------------------------------
// The fragment reads go through __restrict__ parameters on purpose: hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of any
// LDS read that may alias a pending LDS-DMA (which would drain the whole ring every chunk); reads that carry alias-scope metadata
// are checked against the recorded DMA stores instead, and the counted vmcnt + barrier in the loop is what really orders them.
template <int TC, int TP>
__device__ __forceinline__ void frag_read_rb(const uint4* __restrict__ pa, const uint4* __restrict__ pb, uint4 (&fa)[TC], uint4 (&fb)[TP]) {
#pragma unroll
  for (int t = 0; t < TP; ++t) fb[t] = pb[t * 64];
#pragma unroll
  for (int t = 0; t < TC; ++t) fa[t] = pa[t * 64];
}

template <typename T, int WC, int WP, int TC, int TP, int NST, bool STATS = false>
__global__ __launch_bounds__(WC * WP * 64) void igemm_db_kernel(const IgemmArgs a) {
  constexpr int E = Elem<T>::E, KC = 4 * E;
  constexpr int NW = WC * WP, NT = NW * 64;
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16;
  constexpr int NBA = BC / 16, NBB = BP / 16;
  static_assert(NBA % NW == 0 && NBB % NW == 0, "every wave issues the same number of DMAs per chunk");
  constexpr int JA = NBA / NW, JB = NBB / NW, J = JA + JB;
  constexpr int BUF = 4 * (BC + BP);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  int* ltap = reinterpret_cast<int*>(lds + NST * BUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cls = blockIdx.z;
  const int P = a.N * a.Hg * a.Wg;
  int pt, ct;
  {
    const int nb = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const int q8 = nb >> 3, r8 = nb & 7, xcd = id & 7, slot = id >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    ct = logical % (int)gridDim.y; pt = logical / (int)gridDim.y;
  }
  const int p_base = pt * BP, c_base = ct * BC;
  const int r = lane >> 2, g = (lane & 3) ^ rb_swz(lane >> 2);
  if (tid < 16) ltap[tid] = (tid < a.ntaps) ? (((int)a.taps[cls].dh[tid] << 16) | ((int)a.taps[cls].dw[tid] & 0xffff)) : 0;

  int pn[JB], pbh[JB], pbw[JB];
  bool pok[JB];
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int pidx = p_base + (wave + NW * j) * 16 + r;
    pok[j] = pidx < P;
    const int hw = a.Hg * a.Wg;
    const int pc = pok[j] ? pidx : 0;
    const int n = pc / hw, rem = pc - n * hw, q = rem / a.Wg;
    pn[j] = n * a.Hin; pbh[j] = q * a.sh; pbw[j] = (rem - q * a.Wg) * a.sw;
  }
  const unsigned es = sizeof(T);
  const T* x0 = reinterpret_cast<const T*>(a.x.ptr[0]);
  const T* x1 = reinterpret_cast<const T*>(a.x.ptr[1]);
  const int C0 = a.x.C[0], C1 = a.x.C[1];
  __amdgpu_buffer_rsrc_t rsW = make_rsrc(reinterpret_cast<const T*>(a.Wp) + (size_t)cls * a.wp_rows * a.Kpad, 0xFFFFFFFFu);
  __amdgpu_buffer_rsrc_t rsX0 = make_rsrc(x0, (unsigned)((size_t)a.N * a.Hin * a.Win * C0 * es));
  __amdgpu_buffer_rsrc_t rsX1 = make_rsrc(x1 ? (const void*)x1 : (const void*)x0, (unsigned)((size_t)a.N * a.Hin * a.Win * C1 * es));
  unsigned wvo[JA];
#pragma unroll
  for (int j = 0; j < JA; ++j) wvo[j] = (unsigned)(((c_base + (wave + NW * j) * 16 + r) * KC + g * E) * es);
  const unsigned wstep = (unsigned)(a.wp_rows * KC * es);
  unsigned f_wso = 0, f_xso = 0;
  int f_left = 0, f_tap = 0, f_src = 0;
  bool f_use1 = false;
  unsigned f_xvo[JB];
  auto open_segment = [&]() {
    const bool tok = f_tap < a.ntaps;
    const int tv = ltap[tok ? f_tap : 0];
    const int dh = tv >> 16, dw = (int)(short)(tv & 0xffff);
    f_use1 = f_src != 0;
    const int Cs = f_use1 ? C1 : C0;
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int ih = pbh[j] + dh, iw = pbw[j] + dw;
      const bool ok = pok[j] && tok && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      f_xvo[j] = ok ? (unsigned)((((pn[j] + ih) * a.Win + iw) * Cs + g * E) * es) : DMA_OOB;
    }
    f_left = Cs / KC;
    f_xso = 0;
    if (!f_use1 && C1 > 0) f_src = 1; else { f_src = 0; ++f_tap; }
  };
  auto issue = [&](int stage) {
    uint4* la = lds + stage * BUF;
    uint4* lb = la + 4 * BC;
    if (f_left == 0) open_segment();
#pragma unroll
    for (int j = 0; j < JA; ++j) dma16_buf(rsW, wvo[j], f_wso, la + (wave + NW * j) * 64);
    const __amdgpu_buffer_rsrc_t rx = f_use1 ? rsX1 : rsX0;
#pragma unroll
    for (int j = 0; j < JB; ++j) dma16_buf(rx, f_xvo[j], f_xso, lb + (wave + NW * j) * 64);
    f_wso += wstep;
    f_xso += KC * es;
    --f_left;
  };

  const int nchunk = a.Kpad / KC;
  const int wc = wave / WP, wpi = wave - wc * WP;
  const int blkA0 = wc * TC, blkB0 = wpi * TP;
  const int so = (lane & 15) * 4 + ((lane >> 4) ^ rb_swz(lane & 15));   // slot of this lane's fragment piece inside a 16-row block

  f32x4 acc[TC][TP];
#pragma unroll
  for (int i = 0; i < TC; ++i)
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto frag_read = [&](int stage, uint4 (&fa)[TC], uint4 (&fb)[TP]) {
    const uint4* la = lds + stage * BUF;
    frag_read_rb<TC, TP>(la + blkA0 * 64 + so, la + 4 * BC + blkB0 * 64 + so, fa, fb);
  };
  auto mma_all = [&](const uint4 (&fa)[TC], const uint4 (&fb)[TP]) {
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) acc[tc][tp] = mma16<T>(fa[tc], fb[tp], acc[tc][tp]);
  };
  // one iteration: `cur` holds chunk kc (reads issued in the previous iteration), `nxt` receives chunk kc+1
  auto step = [&](int kc, int stage_cur, uint4 (&fa_cur)[TC], uint4 (&fb_cur)[TP], uint4 (&fa_nxt)[TC], uint4 (&fb_nxt)[TP]) {
    if (kc + NST - 1 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * J) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail: fewer chunks outstanding than the constant assumes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // own reads of chunk kc complete: its stage may be refilled after the barrier
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kc + NST < nchunk) issue(stage_cur);
    const int stage_nxt = stage_cur == NST - 1 ? 0 : stage_cur + 1;
    frag_read(stage_nxt, fa_nxt, fb_nxt);     // unconditional (past the last chunk it reads a stale stage that nobody uses): a
    __builtin_amdgcn_sched_barrier(0);        // branch here makes the compiler wait lgkmcnt(0) before the MFMAs of `cur`; the scheduling
    mma_all(fa_cur, fb_cur);                  // barrier keeps the reads AHEAD of the MFMAs (else they sink below and share one register set)
  };

  __syncthreads();   // tap table visible
  uint4 fa0[TC], fb0[TP], fa1[TC], fb1[TP];
#pragma unroll
  for (int d = 0; d < NST; ++d) if (d < nchunk) issue(d);
  if (nchunk >= NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * J) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  frag_read(0, fa0, fb0);
  int stage = 0;
  for (int kc = 0; kc < nchunk; kc += 2) {
    step(kc, stage, fa0, fb0, fa1, fb1);
    stage = stage == NST - 1 ? 0 : stage + 1;
    if (kc + 1 < nchunk) {
      step(kc + 1, stage, fa1, fb1, fa0, fb0);
      stage = stage == NST - 1 ? 0 : stage + 1;
    }
  }
  constexpr int RINGB = NST * BUF * 16;
  constexpr int NPASS = epi_passes(BC, BP, WP, RINGB);
  staged_epilogue<T, TC, TP, BC, BP, NPASS, NT, STATS>(a, LinearPix{a, cls, p_base, P}, c_base, blkA0, blkB0, acc, smem, pt, cls);
}


template <typename T, int WC, int WP, int NST>
static hipError_t launch_db_t(const IgemmArgs& b, dim3 grid, hipStream_t st) {
  constexpr int TC = 8, TP = 4, NW = WC * WP;
  constexpr int BC = WC * TC * 16, BP = WP * TP * 16;
  constexpr int RB = NST * 4 * (BC + BP) * 16;
  constexpr int NPE = epi_passes(BC, BP, WP, RB);
  size_t sm = (size_t)RB + 64;
  const size_t se = (size_t)(BP / NPE) * (BC * 4 + 16) + (BP / NPE) * 8;
  if (se > sm) sm = se;
  auto kern = b.bn_part ? igemm_db_kernel<T, WC, WP, TC, TP, NST, true> : igemm_db_kernel<T, WC, WP, TC, TP, NST, false>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), sm, st, b);
  return hipGetLastError();
}

// bc: channel rows of the tile (256: 256 ch x 256 px, 128: 128 ch x 512 px)
hipError_t launch_igemm_db(const IgemmArgs& b, int is_bf16, int bc, dim3 grid, hipStream_t st) {
  static const int nst_env = getenv("VP_DB_NST") ? atoi(getenv("VP_DB_NST")) : 0;
  if (bc == 256) {
    if (nst_env == 3) return is_bf16 ? launch_db_t<bf16, 2, 4, 3>(b, grid, st) : launch_db_t<float, 2, 4, 3>(b, grid, st);
    return is_bf16 ? launch_db_t<bf16, 2, 4, 4>(b, grid, st) : launch_db_t<float, 2, 4, 4>(b, grid, st);
  }
  return is_bf16 ? launch_db_t<bf16, 1, 8, 3>(b, grid, st) : launch_db_t<float, 1, 8, 3>(b, grid, st);
}

}  // namespace vp
