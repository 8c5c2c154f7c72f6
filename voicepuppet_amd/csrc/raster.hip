// Flat-shaded z-buffer rasteriser, bit-exact with the reference's serial C++ (utils/cython/mesh_core.cpp:169-231,
// inside test :23-50) - the per-frame conditioning-image step between BFMNet and PixReferNet (infer_bfmvid.py:79-108).
//
// The reference visits triangles in index order and overwrites a pixel when the triangle's MEAN depth is strictly
// greater than the buffer, so the survivor of a pixel is the deepest triangle and, among equal depths, the LOWEST index.
// Parallel form: one thread per triangle scans its bounding box and does a 64-bit atomicMax per covered pixel on
// key = (order-preserving bits of the depth) << 32 | (0xFFFFFFFE - triangle index); a second kernel resolves each pixel's
// winner into colour / mask / depth.  All float arithmetic keeps the reference's operation order with contraction off
// (x86-64 g++ -O2 emits no FMA), so the inside test takes the same branch on every pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "errors.h"

#pragma clang fp contract(off)

namespace vp {

__device__ __forceinline__ unsigned int depth_bits(float d) {
  const unsigned int u = __float_as_uint(d + 0.f);           // -0 -> +0: the reference compares floats, where they are equal
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // monotone float -> uint
}

__device__ __forceinline__ bool point_in_tri(float px, float py, float p0x, float p0y, float p1x, float p1y, float p2x, float p2y) {
  const float v0x = p2x - p0x, v0y = p2y - p0y;
  const float v1x = p1x - p0x, v1y = p1y - p0y;
  const float v2x = px - p0x, v2y = py - p0y;
  const float dot00 = ((v0x * v0x) + (v0y * v0y));
  const float dot01 = ((v0x * v1x) + (v0y * v1y));
  const float dot02 = ((v0x * v2x) + (v0y * v2y));
  const float dot11 = ((v1x * v1x) + (v1y * v1y));
  const float dot12 = ((v1x * v2x) + (v1y * v2y));
  const float den = ((dot00 * dot11) - (dot01 * dot01));
  const float inv = (den == 0.f) ? 0.f : (1.f / den);
  const float u = ((dot11 * dot02) - (dot01 * dot12)) * inv;
  const float v = ((dot00 * dot12) - (dot01 * dot02)) * inv;
  return (u >= 0.f) && (v >= 0.f) && ((u + v) < 1.f);
}

__global__ __launch_bounds__(256) void raster_init_kernel(const float* __restrict__ depth, unsigned long long* __restrict__ keys, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) keys[i] = ((unsigned long long)depth_bits(depth[i]) << 32) | 0xFFFFFFFFull;   // beats every triangle of equal depth
}

// bounding boxes up to this many pixels are scanned by the triangle's own lane; larger ones by the whole wave
constexpr int RASTER_LANE_AREA = 32;

__global__ __launch_bounds__(256) void raster_tri_kernel(const float* __restrict__ vertices, const int* __restrict__ triangles,
                                                         unsigned long long* __restrict__ keys, int ntri, int nver, int h, int w, int batch) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const float* V = vertices + (size_t)b * nver * 3;
  unsigned long long* K = keys + (size_t)b * h * w;
  float p0x = 0, p0y = 0, p1x = 0, p1y = 0, p2x = 0, p2y = 0;
  int x_min = 0, x_max = -1, y_min = 0, y_max = -1;
  unsigned int khi = 0, klo = 0;
  if (t < ntri) {
    const int i0 = triangles[3 * t], i1 = triangles[3 * t + 1], i2 = triangles[3 * t + 2];
    p0x = V[3 * i0]; p0y = V[3 * i0 + 1];
    p1x = V[3 * i1]; p1y = V[3 * i1 + 1];
    p2x = V[3 * i2]; p2y = V[3 * i2 + 1];
    const float d0 = V[3 * i0 + 2], d1 = V[3 * i1 + 2], d2 = V[3 * i2 + 2];
    x_min = max((int)ceilf(fminf(p0x, fminf(p1x, p2x))), 0);
    x_max = min((int)floorf(fmaxf(p0x, fmaxf(p1x, p2x))), w - 1);
    y_min = max((int)ceilf(fminf(p0y, fminf(p1y, p2y))), 0);
    y_max = min((int)floorf(fmaxf(p0y, fmaxf(p1y, p2y))), h - 1);
    khi = depth_bits(((d0 + d1) + d2) / 3.f);
    klo = 0xFFFFFFFEu - (unsigned)t;
  }
  const bool covers = x_max >= x_min && y_max >= y_min;
  const int bw = x_max - x_min + 1;
  const int area = covers ? bw * (y_max - y_min + 1) : 0;
  if (area > 0 && area <= RASTER_LANE_AREA) {
    const unsigned long long key = ((unsigned long long)khi << 32) | klo;
    for (int y = y_min; y <= y_max; ++y)
      for (int x = x_min; x <= x_max; ++x)
        if (point_in_tri((float)x, (float)y, p0x, p0y, p1x, p1y, p2x, p2y)) atomicMax(K + (size_t)y * w + x, key);
  }
  // triangles with a large box: one at a time, pixels across the 64 lanes (a BFM frame has none; meshes with
  // slivers or close-ups do, and a single lane walking a 10^4-pixel box would stall its whole wave)
  unsigned long long big = __ballot(area > RASTER_LANE_AREA);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const float q0x = __shfl(p0x, src), q0y = __shfl(p0y, src), q1x = __shfl(p1x, src), q1y = __shfl(p1y, src);
    const float q2x = __shfl(p2x, src), q2y = __shfl(p2y, src);
    const int xm = __shfl(x_min, src), ym = __shfl(y_min, src), qw = __shfl(bw, src), qa = __shfl(area, src);
    const unsigned long long key = ((unsigned long long)__shfl(khi, src) << 32) | __shfl(klo, src);
    for (int i = lane; i < qa; i += 64) {
      const int y = ym + i / qw, x = xm + i % qw;
      if (point_in_tri((float)x, (float)y, q0x, q0y, q1x, q1y, q2x, q2y)) atomicMax(K + (size_t)y * w + x, key);
    }
  }
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const unsigned long long* __restrict__ keys, const float* __restrict__ vertices,
                                                             const int* __restrict__ triangles, const float* __restrict__ colors,
                                                             unsigned char* __restrict__ image, unsigned char* __restrict__ mask,
                                                             float* __restrict__ depth, int nver, int hw, int c, int batch) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)batch * hw) return;
  const unsigned long long k = keys[i];
  const unsigned int low = (unsigned int)k;
  if (low == 0xFFFFFFFFu) return;                          // no triangle beat the initial depth: buffers stay as given
  const int b = (int)(i / hw);
  const int t = (int)(0xFFFFFFFEu - low);
  const int i0 = triangles[3 * t], i1 = triangles[3 * t + 1], i2 = triangles[3 * t + 2];
  const float* V = vertices + (size_t)b * nver * 3;
  const float* C = colors + (size_t)b * nver * c;
  for (int ch = 0; ch < c; ++ch) {
    const float s = ((C[c * i0 + ch] + C[c * i1 + ch]) + C[c * i2 + ch]);
    image[i * c + ch] = (unsigned char)(float)((int)s / 3);  // p_color = (int)(sum)/3 stored through a float (mesh_core.cpp:219)
  }
  mask[i] = 255;
  depth[i] = ((V[3 * i0 + 2] + V[3 * i1 + 2]) + V[3 * i2 + 2]) / 3.f;
}

}  // namespace vp

extern "C" {

size_t vp_render_colors_workspace_bytes(int batch, int h, int w) {
  return (batch < 1 || h < 1 || w < 1) ? 0 : (size_t)batch * h * w * sizeof(unsigned long long) + 256;
}

int vp_render_colors(unsigned char* image, unsigned char* face_mask, const float* vertices, const int* triangles,
                     const float* colors, float* depth_buffer, int ntri, int nver, int h, int w, int c, int batch,
                     void* workspace, void* stream) {
  if (!image || !face_mask || !vertices || (!triangles && ntri > 0) || !colors || !depth_buffer || !workspace || ntri < 0 || nver < 1 || h < 1 ||
      w < 1 || c < 1 || batch < 1) {
    vp::set_err("vp_render_colors: bad argument");
    return VP_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* keys = (unsigned long long*)workspace;
  const size_t n = (size_t)batch * h * w;
  hipLaunchKernelGGL(vp::raster_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, depth_buffer, keys, n);
  if (ntri > 0)
    hipLaunchKernelGGL(vp::raster_tri_kernel, dim3((ntri + 255) / 256, batch), dim3(256), 0, st, vertices, triangles, keys, ntri, nver, h, w, batch);
  hipLaunchKernelGGL(vp::raster_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys, vertices, triangles, colors, image,
                     face_mask, depth_buffer, nver, h * w, c, batch);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

}  // extern "C"
