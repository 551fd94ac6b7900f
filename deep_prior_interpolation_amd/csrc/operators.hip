// Linear operators of the anti-aliasing add-on and the POCS regulariser (reference operators/derivative.py, utils/slopes.py,
// utils/processing.py:139-181, utils/pocs.py) as gfx950 kernels.  All of them are HBM-bound one-pass stencils / maps over
// small 2-D sections (datasets/lines is 170 x 100): one thread per output sample, W contiguous, no LDS needed — the
// neighbours of a sample sit in the same or the adjacent cache line.
#include "common.h"

namespace {

inline unsigned op_blocks(size_t n) {
  size_t b = cdivz(n, 256);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---- first / second derivative along the middle axis of [outer][n][inner] ----------------------------------------
// stencil 0 forward : y[i] = (x[i+1] - x[i]) / h   for i <= n-2, 0 at i = n-1        (utils/processing.py:154-155)
// stencil 1 backward: y[i] = (x[i] - x[i-1]) / h   for i >= 1,   0 at i = 0          (156-157)
// stencil 2 centred : y[i] = (x[i+1] - x[i-1]) / 2h for 1 <= i <= n-2, 0 at the ends (152-153)
// stencil 3 second  : y[i] = (x[i+1] - 2 x[i] + x[i-1]) / h^2 for 1 <= i <= n-2      (177)
// adjoint = 1 applies the transpose of that matrix (what autograd needs for a loss term built on the operator).
__global__ __launch_bounds__(256) void diff_axis_kernel(const float* __restrict__ x, size_t total, int n, size_t inner, int stencil,
                                                        float scale, int adjoint, float* __restrict__ y) {
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int i = (int)((idx / inner) % (size_t)n);
    const float c = x[idx];
    const float up = i + 1 < n ? x[idx + inner] : 0.f;     // x[i+1]
    const float dn = i >= 1 ? x[idx - inner] : 0.f;        // x[i-1]
    float r;
    if (!adjoint) {
      if (stencil == 0) r = i <= n - 2 ? (up - c) : 0.f;
      else if (stencil == 1) r = i >= 1 ? (c - dn) : 0.f;
      else if (stencil == 2) r = (i >= 1 && i <= n - 2) ? 0.5f * up - 0.5f * dn : 0.f;
      else r = (i >= 1 && i <= n - 2) ? (up - 2.f * c + dn) : 0.f;
    } else {
      // row j of the forward matrix is active for j in [lo, hi]; column i collects the active rows that touch it
      if (stencil == 0) r = (i >= 1 ? dn : 0.f) - (i <= n - 2 ? c : 0.f);                        // rows j = i-1 (+1) and j = i (-1)
      else if (stencil == 1) r = (i >= 1 ? c : 0.f) - (i + 1 <= n - 1 ? up : 0.f);               // rows j = i (+1) and j = i+1 (-1)
      else if (stencil == 2) r = 0.5f * ((i - 1 >= 1 && i - 1 <= n - 2) ? dn : 0.f) - 0.5f * ((i + 1 >= 1 && i + 1 <= n - 2) ? up : 0.f);
      else r = ((i - 1 >= 1 && i - 1 <= n - 2) ? dn : 0.f) - 2.f * ((i >= 1 && i <= n - 2) ? c : 0.f) + ((i + 1 >= 1 && i + 1 <= n - 2) ? up : 0.f);
    }
    y[idx] = r / scale;        // a division, like the reference (bit parity for spacings that are not powers of two)
  }
}

// ---- Hale2D / directional_laplacian (utils/slopes.py:51-105) on [N][H][W] planes ---------------------------------
// With Dv, Dh the forward differences (last row / column zero) the reference computes
//     p1 = a*Dv x + b*Dh x,  p2 = b*Dv x + c*Dh x,   y = -( Dh p1 + Dv p2 )
// (it re-applies the FORWARD difference instead of the divergence, and crosses the axes; reproduced as is).  One thread per
// sample evaluates p1 at (i,j),(i,j+1) and p2 at (i,j),(i+1,j) from the 3x3 neighbourhood: 4 reads of x and 3 coefficient
// planes per sample, all served by L1/L2 after the first touch.
struct HaleP {
  float p1, p2;
};
__device__ __forceinline__ HaleP hale_p(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                        const float* __restrict__ c, size_t base, int i, int j, int H, int W) {
  const size_t o = base + (size_t)i * W + j;
  const float x0 = x[o];
  const float gv = i <= H - 2 ? x[o + W] - x0 : 0.f;
  const float gh = j <= W - 2 ? x[o + 1] - x0 : 0.f;
  HaleP r;
  r.p1 = a[o] * gv + b[o] * gh;
  r.p2 = b[o] * gv + c[o] * gh;
  return r;
}
__global__ __launch_bounds__(256) void hale2d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ c, size_t total, int H, int W, float* __restrict__ y) {
  const size_t plane = (size_t)H * W;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t base = idx / plane * plane;
    const int i = (int)((idx - base) / W), j = (int)((idx - base) % W);
    const HaleP p = hale_p(x, a, b, c, base, i, j, H, W);
    float ata1 = 0.f, ata2 = 0.f;
    if (j <= W - 2) ata1 = hale_p(x, a, b, c, base, i, j + 1, H, W).p1 - p.p1;      // Dh p1
    if (i <= H - 2) ata2 = hale_p(x, a, b, c, base, i + 1, j, H, W).p2 - p.p2;      // Dv p2
    y[idx] = -(ata1 + ata2);
  }
}
// transpose: with r1 = Dh^T g, r2 = Dv^T g (D^T u)(k) = u(k-1)[k >= 1] - u(k)[k <= n-2]:
//     q1 = a*r1 + b*r2,  q2 = b*r1 + c*r2,   x = -( Dv^T q1 + Dh^T q2 )
__device__ __forceinline__ float dT(float prev, float cur, int k, int n) { return (k >= 1 ? prev : 0.f) - (k <= n - 2 ? cur : 0.f); }
__device__ __forceinline__ HaleP hale_q(const float* __restrict__ g, const float* __restrict__ a, const float* __restrict__ b,
                                        const float* __restrict__ c, size_t base, int i, int j, int H, int W) {
  const size_t o = base + (size_t)i * W + j;
  const float g0 = g[o];
  const float r1 = dT(j >= 1 ? g[o - 1] : 0.f, g0, j, W);
  const float r2 = dT(i >= 1 ? g[o - W] : 0.f, g0, i, H);
  HaleP r;
  r.p1 = a[o] * r1 + b[o] * r2;
  r.p2 = b[o] * r1 + c[o] * r2;
  return r;
}
__global__ __launch_bounds__(256) void hale2d_adj_kernel(const float* __restrict__ g, const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ c, size_t total, int H, int W, float* __restrict__ xo) {
  const size_t plane = (size_t)H * W;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t base = idx / plane * plane;
    const int i = (int)((idx - base) / W), j = (int)((idx - base) % W);
    const HaleP q = hale_q(g, a, b, c, base, i, j, H, W);
    const float q1_up = i >= 1 ? hale_q(g, a, b, c, base, i - 1, j, H, W).p1 : 0.f;
    const float q2_left = j >= 1 ? hale_q(g, a, b, c, base, i, j - 1, H, W).p2 : 0.f;
    xo[idx] = -(dT(q1_up, q.p1, i, H) + dT(q2_left, q.p2, j, W));
  }
}

// ---- structure tensor (utils/slopes.py:6-48) ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void structure_tensor_kernel(const float* __restrict__ x, size_t total, int H, int W, float dv, float dh,
                                                               float* __restrict__ gvv, float* __restrict__ gvh, float* __restrict__ ghh) {
  const size_t plane = (size_t)H * W;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t base = idx / plane * plane;
    const int i = (int)((idx - base) / W), j = (int)((idx - base) % W);
    const float x0 = x[idx];
    const float gv = i <= H - 2 ? (x[idx + W] - x0) / dv : 0.f;
    const float gh = j <= W - 2 ? (x[idx + 1] - x0) / dh : 0.f;
    gvv[idx] = gv * gv;
    gvh[idx] = gv * gh;
    ghh[idx] = gh * gh;
  }
}
// eigen-decomposition of the 2x2 tensor per sample: phi = atan((l1 - gvv) / gvh) with NaN -> 0 (0/0 where the tensor is
// diagonal), anisotropy = 1 - l2 / l1 (left as IEEE gives it, like the reference)
__global__ __launch_bounds__(256) void dips_kernel(const float* __restrict__ gvv, const float* __restrict__ gvh, const float* __restrict__ ghh,
                                                   size_t n, float* __restrict__ phi, float* __restrict__ aniso) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float vv = gvv[i], vh = gvh[i], hh = ghh[i];
    const float t1 = 0.5f * (vv + hh);
    const float d = vv - hh;
    const float t2 = 0.5f * sqrtf(d * d + 4.f * (vh * vh));
    const float l1 = t1 + t2, l2 = t1 - t2;
    float p = atanf((l1 - vv) / vh);
    if (p != p) p = 0.f;
    phi[i] = p;
    aniso[i] = 1.f - l2 / l1;
  }
}

// ---- POCS (utils/pocs.py:5-19, 80-84) -----------------------------------------------------------------------------
// max over a real tensor (the reference thresholds real and imaginary parts of the spectrum as independent reals), two
// stages, deterministic; the threshold th = max * perc / 100 stays on the device.
__global__ __launch_bounds__(256) void max_partial_kernel(const float* __restrict__ x, size_t n, float* __restrict__ part) {
  __shared__ float sh[4];
  float m = -INFINITY;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, x[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__global__ __launch_bounds__(256) void max_final_kernel(const float* __restrict__ part, int nb, float scale, float* __restrict__ out) {
  __shared__ float sh[4];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < nb; i += 256) m = fmaxf(m, part[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) * scale;
}
// y = x * ((x > th) + (x < -th))
__global__ __launch_bounds__(256) void threshold_kernel(const float* __restrict__ x, size_t n, const float* __restrict__ th_ptr, float* __restrict__ y) {
  const float th = *th_ptr;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = x[i];
    y[i] = v * ((v > th ? 1.f : 0.f) + (v < -th ? 1.f : 0.f));
  }
}
// y = wdata + wmask * x      (weighted_data + weighted_mask * adjoint(threshold(forward(x))))
__global__ __launch_bounds__(256) void pocs_project_kernel(const float* __restrict__ x, const float* __restrict__ wdata, const float* __restrict__ wmask,
                                                           size_t n, float* __restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = wdata[i] + wmask[i] * x[i];
}

}  // namespace

extern "C" int dpi_diff_axis(const float* x, size_t outer, int n, size_t inner, int stencil, float spacing, int adjoint, float* y, void* stream) {
  DPI_REQUIRE(x && y && x != y && outer > 0 && n > 0 && inner > 0, "diff_axis: bad argument");
  DPI_REQUIRE(stencil >= 0 && stencil <= 3, "diff_axis: stencil must be 0 (forward), 1 (backward), 2 (centered) or 3 (second)");
  DPI_REQUIRE(spacing != 0.f, "diff_axis: zero spacing");
  const size_t total = outer * (size_t)n * inner;
  const float scale = stencil == 3 ? spacing * spacing : spacing;
  diff_axis_kernel<<<op_blocks(total), 256, 0, (hipStream_t)stream>>>(x, total, n, inner, stencil, scale, adjoint, y);
  return dpi_check_launch("diff_axis");
}

extern "C" int dpi_hale2d(const float* x, const float* a, const float* b, const float* c, size_t N, int H, int W, int adjoint, float* y, void* stream) {
  DPI_REQUIRE(x && a && b && c && y && x != y && N > 0 && H > 0 && W > 0, "hale2d: bad argument");
  const size_t total = N * (size_t)H * W;
  if (adjoint) hale2d_adj_kernel<<<op_blocks(total), 256, 0, (hipStream_t)stream>>>(x, a, b, c, total, H, W, y);
  else hale2d_fwd_kernel<<<op_blocks(total), 256, 0, (hipStream_t)stream>>>(x, a, b, c, total, H, W, y);
  return dpi_check_launch("hale2d");
}

extern "C" int dpi_structure_tensor(const float* x, size_t N, int H, int W, float dv, float dh, float* gvv, float* gvh, float* ghh, void* stream) {
  DPI_REQUIRE(x && gvv && gvh && ghh && N > 0 && H > 0 && W > 0 && dv != 0.f && dh != 0.f, "structure_tensor: bad argument");
  const size_t total = N * (size_t)H * W;
  structure_tensor_kernel<<<op_blocks(total), 256, 0, (hipStream_t)stream>>>(x, total, H, W, dv, dh, gvv, gvh, ghh);
  return dpi_check_launch("structure_tensor");
}

extern "C" int dpi_dips(const float* gvv, const float* gvh, const float* ghh, size_t n, float* phi, float* anisotropy, void* stream) {
  DPI_REQUIRE(gvv && gvh && ghh && phi && anisotropy && n > 0, "dips: bad argument");
  dips_kernel<<<op_blocks(n), 256, 0, (hipStream_t)stream>>>(gvv, gvh, ghh, n, phi, anisotropy);
  return dpi_check_launch("dips");
}

extern "C" size_t dpi_max_ws_floats(size_t n) { return (size_t)op_blocks(n); }

extern "C" int dpi_scaled_max(const float* x, size_t n, float scale, float* ws, float* out, void* stream) {
  DPI_REQUIRE(x && ws && out && n > 0, "scaled_max: bad argument");
  const unsigned nb = op_blocks(n);
  max_partial_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(x, n, ws);
  max_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(ws, (int)nb, scale, out);
  return dpi_check_launch("scaled_max");
}

extern "C" int dpi_threshold(const float* x, size_t n, const float* thresh, float* y, void* stream) {
  DPI_REQUIRE(x && y && thresh && n > 0, "threshold: bad argument");
  threshold_kernel<<<op_blocks(n), 256, 0, (hipStream_t)stream>>>(x, n, thresh, y);
  return dpi_check_launch("threshold");
}

extern "C" int dpi_pocs_project(const float* x, const float* wdata, const float* wmask, size_t n, float* y, void* stream) {
  DPI_REQUIRE(x && wdata && wmask && y && n > 0, "pocs_project: bad argument");
  pocs_project_kernel<<<op_blocks(n), 256, 0, (hipStream_t)stream>>>(x, wdata, wmask, n, y);
  return dpi_check_launch("pocs_project");
}
