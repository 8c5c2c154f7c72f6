// BFM reconstruction for a clip: [T,257] coefficients -> projected vertices + per-vertex colours for the rasteriser.
// Device form of utils/reconstruct_mesh.py `Reconstruction_rotation` (:198-223) + the packing of infer_bfmvid.py:92-99,
// batched over the T frames of a clip.  float64 arithmetic like the reference's numpy (the model bases are promoted to
// float64 once, at model load), so the float32 vertices / integer colours handed to the rasteriser agree with the
// reference's; the work is HBM-bound on the three basis matrices, which are read ONCE per clip instead of once per frame.
//
//   linear  : shape/texture = base[3N,K] . coeff[T,K] + mean   (thread per row over K-MAJOR bases, 32 frames per pass)
//   fnormal : per (frame, triangle) cross product              (:43-46)
//   vertex  : per (frame, vertex) one-ring normal, rotate, project, SH lighting, pack   (:50-52, :208-221, :100-169)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "errors.h"

#pragma clang fp contract(off)

namespace vp {

constexpr int BFM_FT = 32;        // frames per thread (one pass of the bases serves 32 frames)
constexpr int BFM_LIN_THREADS = 128;

struct LinearArgs {
  const double* b1; int k1; int o1;      // K-MAJOR base [k1][rows] (transposed once at model load), coefficient offset into the 257 vector
  const double* b2; int k2; int o2;      // optional second base
  const double* mean;                    // [rows]
  double sub[3];                         // subtracted per coordinate (the mean-shape centre, :27)
  const float* coeff;                    // [T,257]
  double* out;                           // [T,rows]
  int rows, frames;
};

// out[f][row] = sum_k base[k][row] * coeff[f][k] + mean[row] - sub[row % 3].  One thread per row: the k-major base makes
// every load a coalesced 512-byte wave request and the 144 loads of a thread are independent (deep HBM pipelining);
// the clip's coefficients sit in LDS as [k][frame] and are read as wave-uniform broadcasts, two frames per ds_read_b128.
__global__ __launch_bounds__(BFM_LIN_THREADS) void bfm_linear_kernel(LinearArgs a) {
  __shared__ double sc[144 * BFM_FT];
  const int K = a.k1 + a.k2;
  const int f0 = blockIdx.y * BFM_FT;
  for (int i = threadIdx.x; i < K * BFM_FT; i += BFM_LIN_THREADS) {
    const int k = i / BFM_FT, f = f0 + i % BFM_FT;
    sc[i] = (f < a.frames) ? (double)a.coeff[(size_t)f * 257 + (k < a.k1 ? a.o1 + k : a.o2 + k - a.k1)] : 0.0;
  }
  __syncthreads();
  const int row = blockIdx.x * BFM_LIN_THREADS + threadIdx.x;
  if (row >= a.rows) return;
  double acc[BFM_FT];
#pragma unroll
  for (int j = 0; j < BFM_FT; ++j) acc[j] = 0.0;
  const int nf = min(BFM_FT, a.frames - f0);
  auto accumulate = [&](const double* __restrict__ base, int kn, int kofs) {
#pragma unroll 4
    for (int k = 0; k < kn; ++k) {
      const double b = __builtin_nontemporal_load(base + (size_t)k * a.rows + row);
      const double2* c = reinterpret_cast<const double2*>(sc + (kofs + k) * BFM_FT);
      if (nf <= 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const double2 v = c[j]; acc[2 * j] = fma(b, v.x, acc[2 * j]); acc[2 * j + 1] = fma(b, v.y, acc[2 * j + 1]); }
      } else {
#pragma unroll
        for (int j = 0; j < BFM_FT / 2; ++j) { const double2 v = c[j]; acc[2 * j] = fma(b, v.x, acc[2 * j]); acc[2 * j + 1] = fma(b, v.y, acc[2 * j + 1]); }
      }
    }
  };
  accumulate(a.b1, a.k1, 0);
  if (a.k2) accumulate(a.b2, a.k2, a.k1);
  const double add = a.mean[row], sub = a.sub[row % 3];
#pragma unroll
  for (int j = 0; j < BFM_FT; ++j)
    if (j < nf) a.out[(size_t)(f0 + j) * a.rows + row] = (acc[j] + add) - sub;
}

__global__ __launch_bounds__(256) void bfm_fnormal_kernel(const double* __restrict__ shape, const int* __restrict__ tri, double* __restrict__ fn,
                                                          int nver, int ntri) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y;
  if (t > ntri) return;
  double* o = fn + ((size_t)f * (ntri + 1) + t) * 3;
  if (t == ntri) { o[0] = o[1] = o[2] = 0.0; return; }     // the appended zero normal that point_buf pads with (:47-49)
  const double* S = shape + (size_t)f * nver * 3;
  const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
  const double ax = S[3 * i0] - S[3 * i1], ay = S[3 * i0 + 1] - S[3 * i1 + 1], az = S[3 * i0 + 2] - S[3 * i1 + 2];
  const double bx = S[3 * i1] - S[3 * i2], by = S[3 * i1 + 1] - S[3 * i2 + 1], bz = S[3 * i1 + 2] - S[3 * i2 + 2];
  o[0] = ay * bz - az * by;
  o[1] = az * bx - ax * bz;
  o[2] = ax * by - ay * bx;
}

struct VertexArgs {
  const double* shape;       // [T,N,3] unrotated, centred
  const double* tex;         // [T or 1,N,3]
  const double* fn;          // [T,F+1,3]
  const int* point_buf;      // [N,8] 0-based, F = none
  const double* rot;         // [T,9] row-major rotation (Compute_rotation_matrix output)
  const float* coeff;        // [T,257]
  double* face_shape;        // optional outputs of Reconstruction_rotation
  double* face_color;
  double* face_projection;
  double* z_buffer;
  float* vertices;           // [T,N,3]  x, 224-y, z_buffer   (infer_bfmvid.py:92-96)
  float* colors;             // [T,N,3]  clip(0,255) -> int -> float   (:98,102)
  int nver, ntri, frames, tex_frames;
  double focal, center;
  double sh[9];              // a_i*c_i products of Illumination_layer (:138-143), evaluated on the host in double
};

__global__ __launch_bounds__(256) void bfm_vertex_kernel(VertexArgs a) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  const int f = blockIdx.y;
  if (v >= a.nver) return;
  const double* R = a.rot + f * 9;
  const double r00 = R[0], r01 = R[1], r02 = R[2], r10 = R[3], r11 = R[4], r12 = R[5], r20 = R[6], r21 = R[7], r22 = R[8];
  // one-ring vertex normal
  const double* FN = a.fn + (size_t)f * (a.ntri + 1) * 3;
  double nx = 0, ny = 0, nz = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int t = a.point_buf[v * 8 + j];
    nx += FN[3 * t]; ny += FN[3 * t + 1]; nz += FN[3 * t + 2];
  }
  const double len = sqrt((nx * nx + ny * ny) + nz * nz);
  nx /= len; ny /= len; nz /= len;
  const double mx = nx * r00 + ny * r10 + nz * r20, my = nx * r01 + ny * r11 + nz * r21, mz = nx * r02 + ny * r12 + nz * r22;
  // shape: rotated once for the returned face_shape, and once more inside Projection_layer (:211,:214 -> :113)
  const size_t vi = ((size_t)f * a.nver + v) * 3;
  const double sx = a.shape[vi], sy = a.shape[vi + 1], sz = a.shape[vi + 2];
  const double px = sx * r00 + sy * r10 + sz * r20, py = sx * r01 + sy * r11 + sz * r21, pz = sx * r02 + sy * r12 + sz * r22;
  if (a.face_shape) { a.face_shape[vi] = px; a.face_shape[vi + 1] = py; a.face_shape[vi + 2] = pz; }
  const float* C = a.coeff + (size_t)f * 257;
  const double qx = (px * r00 + py * r10 + pz * r20) + (double)C[254];
  const double qy = (px * r01 + py * r11 + pz * r21) + (double)C[255];
  const double qz = -((px * r02 + py * r12 + pz * r22) + (double)C[256]) + 10.0;
  const double ux = a.focal * qx + a.center * qz, uy = a.focal * qy + a.center * qz;
  const double prx = ux / qz, pry = 224.0 - uy / qz, zb = -qz;
  if (a.face_projection) { a.face_projection[((size_t)f * a.nver + v) * 2] = prx; a.face_projection[((size_t)f * a.nver + v) * 2 + 1] = pry; }
  if (a.z_buffer) a.z_buffer[(size_t)f * a.nver + v] = zb;
  a.vertices[vi] = (float)prx; a.vertices[vi + 1] = (float)pry; a.vertices[vi + 2] = (float)zb;
  // SH lighting on the rotated normal
  double Y[9];
  Y[0] = a.sh[0];
  Y[1] = -a.sh[1] * my; Y[2] = a.sh[1] * mz; Y[3] = -a.sh[1] * mx;
  Y[4] = a.sh[2] * mx * my; Y[5] = -a.sh[2] * my * mz;
  Y[6] = a.sh[3] * (3.0 * (mz * mz) - 1.0);
  Y[7] = -a.sh[2] * mx * mz;
  Y[8] = a.sh[4] * (mx * mx - my * my);
  const double* TX = a.tex + ((size_t)(a.tex_frames == 1 ? 0 : f) * a.nver + v) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    double lit = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) lit += Y[k] * ((double)C[227 + c * 9 + k] + (k == 0 ? 0.8 : 0.0));
    const double col = lit * TX[c];
    if (a.face_color) a.face_color[vi + c] = col;
    a.colors[vi + c] = (float)(int)fmin(fmax(col, 0.0), 255.0);
  }
}

}  // namespace vp

extern "C" {

size_t vp_bfm_reconstruct_workspace_bytes(int nver, int ntri, int frames) {
  if (nver < 1 || ntri < 1 || frames < 1) return 0;
  return ((size_t)frames * nver * 3 * 2 + (size_t)frames * (ntri + 1) * 3) * sizeof(double) + 512;
}

int vp_bfm_reconstruct(const vp_bfm_model* m, const float* coeff, const double* rotation, int frames, int shared_texture,
                       double* face_shape, double* face_texture, double* face_color, double* face_projection, double* z_buffer,
                       float* vertices, float* colors, void* workspace, size_t workspace_bytes, void* stream) {
  if (!m || !coeff || !rotation || !vertices || !colors || !workspace || frames < 1 || m->nver < 1 || m->ntri < 1 || !m->idBase ||
      !m->exBase || !m->texBase || !m->meanshape || !m->meantex || !m->tri || !m->point_buf) {
    vp::set_err("vp_bfm_reconstruct: bad argument");
    return VP_ERR_ARG;
  }
  if (workspace_bytes < vp_bfm_reconstruct_workspace_bytes(m->nver, m->ntri, frames)) {
    vp::set_err("vp_bfm_reconstruct: workspace too small");
    return VP_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const int rows = 3 * m->nver;
  const int tex_frames = shared_texture ? 1 : frames;
  double* shape = (double*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  double* tex = face_texture ? face_texture : shape + (size_t)frames * rows;
  double* fn = shape + (size_t)frames * rows * 2;
  vp::LinearArgs la{};
  la.b1 = m->idBase; la.k1 = 80; la.o1 = 0; la.b2 = m->exBase; la.k2 = 64; la.o2 = 80;
  la.mean = m->meanshape; la.sub[0] = m->center[0]; la.sub[1] = m->center[1]; la.sub[2] = m->center[2];
  la.coeff = coeff; la.out = shape; la.rows = rows; la.frames = frames;
  const int nb = (rows + vp::BFM_LIN_THREADS - 1) / vp::BFM_LIN_THREADS;
  hipLaunchKernelGGL(vp::bfm_linear_kernel, dim3(nb, (frames + vp::BFM_FT - 1) / vp::BFM_FT), dim3(vp::BFM_LIN_THREADS), 0, st, la);
  vp::LinearArgs lt{};
  lt.b1 = m->texBase; lt.k1 = 80; lt.o1 = 144; lt.b2 = nullptr; lt.k2 = 0; lt.o2 = 0; lt.mean = m->meantex;
  lt.coeff = coeff; lt.out = tex; lt.rows = rows; lt.frames = tex_frames;
  hipLaunchKernelGGL(vp::bfm_linear_kernel, dim3(nb, (tex_frames + vp::BFM_FT - 1) / vp::BFM_FT), dim3(vp::BFM_LIN_THREADS), 0, st, lt);
  hipLaunchKernelGGL(vp::bfm_fnormal_kernel, dim3((m->ntri + 1 + 255) / 256, frames), dim3(256), 0, st, shape, m->tri, fn, m->nver, m->ntri);
  vp::VertexArgs va{};
  va.shape = shape; va.tex = tex; va.fn = fn; va.point_buf = m->point_buf; va.rot = rotation; va.coeff = coeff;
  va.face_shape = face_shape; va.face_color = face_color; va.face_projection = face_projection; va.z_buffer = z_buffer;
  va.vertices = vertices; va.colors = colors; va.nver = m->nver; va.ntri = m->ntri; va.frames = frames; va.tex_frames = tex_frames;
  va.focal = m->focal; va.center = m->image_center;
  for (int i = 0; i < 5; ++i) va.sh[i] = m->sh[i];
  hipLaunchKernelGGL(vp::bfm_vertex_kernel, dim3((m->nver + 255) / 256, frames), dim3(256), 0, st, va);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

}  // extern "C"
