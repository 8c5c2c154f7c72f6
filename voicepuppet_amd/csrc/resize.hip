// cv2.resize(uint8, dsize) with the default INTER_LINEAR, and the paste of the result into a zero canvas: the last two lines of the
// reference's render_face (voicepuppet/pixrefer/infer_bfmvid.py:110-121).  "Bit-exact for byte buffers" (north star) needs OpenCV's
// FIXED-POINT bilinear, not a float one: OpenCV (modules/imgproc/src/resize.cpp; a dependency the reference does not vendor or pin -
// the algorithm is unchanged across 3.x / 4.x) computes for 8-bit images
//   scale = 1 / (dsize / ssize)                       (double)
//   fx = (float)((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx
//   sx < 0: fx = 0, sx = 0;   sx + 1 >= width: dx beyond xmax reads only S[sx]; sx >= width - 1: fx = 0, sx = width - 1
//   alpha = (short)lrintf((1.f - fx) * 2048), (short)lrintf(fx * 2048)          (INTER_RESIZE_COEF_BITS = 11, round half to even)
//   horizontal pass, int32:  D[dx] = S[sx] * alpha0 + S[sx + 1] * alpha1        (dx >= xmax: S[sx] * 2048)
//   vertical pass:  dst = (((beta0 * (D0 >> 4)) >> 16) + ((beta1 * (D1 >> 4)) >> 16) + 2) >> 2   with source rows clipped to the image
// plus two shortcuts: equal sizes copy, an exact 2x reduction is INTER_AREA's (a + b + c + d + 2) >> 2.
// cv2 cannot be imported here, so this is pinned by hand-derived known answers and a separately written numpy restatement
// (oracle/cv_resize_ref.py, tests/test_cv_resize.py), not by cv2 itself: "unpinned by cv2".
// The coefficient tables are built on the host in the same C float / double arithmetic as OpenCV (this file is compiled with
// -ffp-contract=off); the per-pixel work is integer and runs on the device, all frames of a clip in one launch.
#include <math.h>
#include <string.h>

#include <vector>

#include "errors.h"
#include "vp_common.h"

namespace vp {

struct ResizeTab { int ofs; short a0, a1; };      // source index, the two 11-bit coefficients (a1 = 0 and ofs clipped beyond xmax)

static void build_table(int ssize, int dsize, ResizeTab* tab) {
  const double inv_scale = (double)dsize / ssize;
  const double scale = 1.0 / inv_scale;
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    bool single = false;                             // beyond xmax: only S[s] is read (weight 2048)
    if (s < 0) { f = 0.f; s = 0; }
    if (s + 1 >= ssize) {
      single = true;
      if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    }
    const float c0 = 1.f - f, c1 = f;
    long a0 = lrintf(c0 * 2048.f), a1 = lrintf(c1 * 2048.f);
    a0 = a0 < -32768 ? -32768 : (a0 > 32767 ? 32767 : a0);
    a1 = a1 < -32768 ? -32768 : (a1 > 32767 ? 32767 : a1);
    tab[d].ofs = s;
    tab[d].a0 = single ? (short)2048 : (short)a0;
    tab[d].a1 = single ? (short)0 : (short)a1;
  }
}

// rows: OpenCV keeps the (possibly negative) floor and clips the two source ROWS instead of the coefficient
static void build_row_table(int ssize, int dsize, ResizeTab* tab, int* row1) {
  const double inv_scale = (double)dsize / ssize;
  const double scale = 1.0 / inv_scale;
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    f -= s;
    const float c0 = 1.f - f, c1 = f;
    long b0 = lrintf(c0 * 2048.f), b1 = lrintf(c1 * 2048.f);
    b0 = b0 < -32768 ? -32768 : (b0 > 32767 ? 32767 : b0);
    b1 = b1 < -32768 ? -32768 : (b1 > 32767 ? 32767 : b1);
    const int r0 = s < 0 ? 0 : (s > ssize - 1 ? ssize - 1 : s);
    const int r1 = s + 1 < 0 ? 0 : (s + 1 > ssize - 1 ? ssize - 1 : s + 1);
    tab[d].ofs = r0; tab[d].a0 = (short)b0; tab[d].a1 = (short)b1;
    row1[d] = r1;
  }
}

struct ResizeArgs {
  const unsigned char* src;   // [T][hs][ws][3]
  unsigned char* dst;         // [T][H][W][3] canvas
  const ResizeTab* xt;        // [dw]
  const ResizeTab* yt;        // [dh]
  const int* yrow1;           // [dh]
  int T, hs, ws, dh, dw, H, W, y0, x0;
  int swap_rb;                // cv2.cvtColor(BGR2RGB) in front of the resize (infer_bfmvid.py:110): channel c reads source channel 2 - c
  int mode;                   // 0 bilinear, 1 copy (equal sizes), 2 exact 2x reduction (INTER_AREA)
};

__global__ __launch_bounds__(256) void resize_paste_kernel(const ResizeArgs a) {
  const size_t total = (size_t)a.T * a.dh * a.dw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int dx = (int)(i % a.dw), dy = (int)((i / a.dw) % a.dh), t = (int)(i / ((size_t)a.dw * a.dh));
    const int oy = a.y0 + dy, ox = a.x0 + dx;
    if (oy < 0 || oy >= a.H || ox < 0 || ox >= a.W) continue;
    const unsigned char* s = a.src + (size_t)t * a.hs * a.ws * 3;
    unsigned char* o = a.dst + (((size_t)t * a.H + oy) * a.W + ox) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int sc = a.swap_rb ? 2 - c : c;
      int v;
      if (a.mode == 1) {
        v = s[((size_t)dy * a.ws + dx) * 3 + sc];
      } else if (a.mode == 2) {
        const unsigned char* p = s + ((size_t)(2 * dy) * a.ws + 2 * dx) * 3 + sc;
        v = (p[0] + p[3] + p[(size_t)a.ws * 3] + p[(size_t)a.ws * 3 + 3] + 2) >> 2;
      } else {
        const ResizeTab xt = a.xt[dx], yt = a.yt[dy];
        const int r1 = a.yrow1[dy];
        const int x1 = xt.a1 ? xt.ofs + 1 : xt.ofs;            // (a1 == 0: the second tap is never read by OpenCV either)
        const unsigned char* p0 = s + (size_t)yt.ofs * a.ws * 3;
        const unsigned char* p1 = s + (size_t)r1 * a.ws * 3;
        const int d0 = p0[xt.ofs * 3 + sc] * xt.a0 + p0[x1 * 3 + sc] * xt.a1;
        const int d1 = p1[xt.ofs * 3 + sc] * xt.a0 + p1[x1 * 3 + sc] * xt.a1;
        v = ((((int)yt.a0 * (d0 >> 4)) >> 16) + (((int)yt.a1 * (d1 >> 4)) >> 16) + 2) >> 2;
      }
      o[c] = (unsigned char)v;
    }
  }
}

}  // namespace vp

using namespace vp;

extern "C" {

size_t vp_resize_paste_workspace_bytes(int dst_h, int dst_w) {
  if (dst_h < 1 || dst_w < 1) return 0;
  return (size_t)(dst_h + dst_w) * sizeof(ResizeTab) + (size_t)dst_h * sizeof(int) + 256;
}

// src [frames][src_h][src_w][3] uint8 -> resized to dst_h x dst_w as cv2.resize(src, (dst_w, dst_h)) does (optionally behind
// cv2.cvtColor(BGR2RGB)) and written into canvas [frames][canvas_h][canvas_w][3] at rows y0.., columns x0.. (pixels outside the
// canvas are dropped); the rest of the canvas is NOT touched (the caller zeroes it: np.zeros in infer_bfmvid.py:114).
int vp_resize_paste_u8(const unsigned char* src, int frames, int src_h, int src_w, int dst_h, int dst_w, int swap_rb,
                       unsigned char* canvas, int canvas_h, int canvas_w, int y0, int x0, void* workspace, void* stream) {
  if (!src || !canvas || !workspace || frames < 1 || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1 || canvas_h < 1 || canvas_w < 1) {
    set_err("vp_resize_paste_u8: bad argument");
    return VP_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  ResizeArgs a;
  memset(&a, 0, sizeof(a));
  a.src = src; a.dst = canvas; a.T = frames; a.hs = src_h; a.ws = src_w; a.dh = dst_h; a.dw = dst_w;
  a.H = canvas_h; a.W = canvas_w; a.y0 = y0; a.x0 = x0; a.swap_rb = swap_rb ? 1 : 0;
  a.mode = (dst_h == src_h && dst_w == src_w) ? 1 : ((src_h == 2 * dst_h && src_w == 2 * dst_w) ? 2 : 0);
  if (a.mode == 0) {
    std::vector<ResizeTab> tab((size_t)dst_w + dst_h);
    std::vector<int> r1((size_t)dst_h);
    build_table(src_w, dst_w, tab.data());
    build_row_table(src_h, dst_h, tab.data() + dst_w, r1.data());
    char* ws = (char*)workspace;
    ResizeTab* d_tab = (ResizeTab*)ws;
    int* d_r1 = (int*)(ws + tab.size() * sizeof(ResizeTab));
    // (pageable host memory: the copies are complete when hipMemcpyAsync returns control of the vectors)
    VP_HIP_CHECK(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(ResizeTab), hipMemcpyHostToDevice, st));
    VP_HIP_CHECK(hipMemcpyAsync(d_r1, r1.data(), r1.size() * sizeof(int), hipMemcpyHostToDevice, st));
    VP_HIP_CHECK(hipStreamSynchronize(st));
    a.xt = d_tab; a.yt = d_tab + dst_w; a.yrow1 = d_r1;
  }
  const size_t total = (size_t)frames * dst_h * dst_w;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(resize_paste_kernel, dim3(blocks), dim3(256), 0, st, a);
  VP_HIP_CHECK(hipGetLastError());
  return VP_OK;
}

// the same tables for a host caller / tests: ofs [n], a0 [n], a1 [n] of the column pass (rows = 0) or the row pass (rows = 1: ofs is
// the first source row after clipping, row1 the second)
int vp_resize_linear_table(int src_size, int dst_size, int rows, int* ofs, short* a0, short* a1, int* row1) {
  if (src_size < 1 || dst_size < 1 || !ofs || !a0 || !a1 || (rows && !row1)) { set_err("vp_resize_linear_table: bad argument"); return VP_ERR_ARG; }
  std::vector<ResizeTab> tab((size_t)dst_size);
  if (rows) build_row_table(src_size, dst_size, tab.data(), row1);
  else build_table(src_size, dst_size, tab.data());
  for (int i = 0; i < dst_size; ++i) { ofs[i] = tab[i].ofs; a0[i] = tab[i].a0; a1[i] = tab[i].a1; }
  return VP_OK;
}

}  // extern "C"
