// Ops used only by the plain 2-D UNet (reference architectures/unet.py:38-81): MaxPool2d(2,2) and
// ConvTranspose2d(Cin, Cout, kernel 4, stride 2, padding 1) with their backward passes.  The UNet is a side path of
// the hot-path scope, so these are straightforward VALU kernels: correct first, one thread per output.
#include "common.h"

namespace {

// ---- MaxPool 2x2, stride 2, floor mode; ties resolve to the first element in (kh, kw) scan order like aten ------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int H, int W, int Ho, int Wo,
                                                          float* __restrict__ y) {
  const int c = blockIdx.y;
  const size_t Vo = (size_t)Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < Vo; i += (size_t)gridDim.x * 256) {
    const int ow = i % Wo, oh = i / Wo;
    const float* p = x + (size_t)c * H * W + (size_t)(2 * oh) * W + 2 * ow;
    float m = p[0];
    m = p[1] > m ? p[1] : m;
    m = p[W] > m ? p[W] : m;
    m = p[W + 1] > m ? p[W + 1] : m;
    y[(size_t)c * Vo + i] = m;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, int H, int W, int Ho,
                                                          int Wo, float* __restrict__ dx) {
  const int c = blockIdx.y;
  const size_t V = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < V; i += (size_t)gridDim.x * 256) {
    const int w = i % W, h = i / W;
    const int oh = h >> 1, ow = w >> 1;
    float g = 0.f;
    if (oh < Ho && ow < Wo) {
      const float* p = x + (size_t)c * V + (size_t)(2 * oh) * W + 2 * ow;
      int arg = 0;
      float m = p[0];
      if (p[1] > m) { m = p[1]; arg = 1; }
      if (p[W] > m) { m = p[W]; arg = 2; }
      if (p[W + 1] > m) { m = p[W + 1]; arg = 3; }
      if (arg == ((h & 1) * 2 + (w & 1))) g = dy[(size_t)c * Ho * Wo + (size_t)oh * Wo + ow];
    }
    dx[(size_t)c * V + i] = g;
  }
}

// ---- ConvTranspose2d k=4 s=2 p=1:  y[co][oh][ow] = b[co] + sum_ci sum_{kh,kw} x[ci][ih][iw] w[ci][co][kh][kw],
//      oh = 2 ih - 1 + kh  (two valid kh per output row, likewise for columns); output size 2H x 2W -------------------------
__global__ __launch_bounds__(256) void deconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int Cin, int Cout, int H, int W,
                                                         float* __restrict__ y) {
  const int co = blockIdx.y, Ho = 2 * H, Wo = 2 * W;
  const size_t Vo = (size_t)Ho * Wo, V = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < Vo; i += (size_t)gridDim.x * 256) {
    const int ow = i % Wo, oh = i / Wo;
    float acc = bias ? bias[co] : 0.f;
    for (int a = 0; a < 2; ++a) {
      const int kh = ((oh + 1) & 1) + 2 * a, ih = (oh + 1 - kh) >> 1;
      if (oh + 1 - kh < 0 || ih >= H) continue;
      for (int b = 0; b < 2; ++b) {
        const int kw = ((ow + 1) & 1) + 2 * b, iw = (ow + 1 - kw) >> 1;
        if (ow + 1 - kw < 0 || iw >= W) continue;
        const float* xp = x + (size_t)ih * W + iw;
        const float* wp = w + ((size_t)co * 4 + kh) * 4 + kw;
        for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xp[(size_t)ci * V], wp[(size_t)ci * Cout * 16], acc);
      }
    }
    y[(size_t)co * Vo + i] = acc;
  }
}

__global__ __launch_bounds__(256) void deconv_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w, int Cin, int Cout,
                                                              int H, int W, float* __restrict__ dx) {
  const int ci = blockIdx.y, Ho = 2 * H, Wo = 2 * W;
  const size_t Vo = (size_t)Ho * Wo, V = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < V; i += (size_t)gridDim.x * 256) {
    const int iw = i % W, ih = i / W;
    float acc = 0.f;
    for (int kh = 0; kh < 4; ++kh) {
      const int oh = 2 * ih - 1 + kh;
      if (oh < 0 || oh >= Ho) continue;
      for (int kw = 0; kw < 4; ++kw) {
        const int ow = 2 * iw - 1 + kw;
        if (ow < 0 || ow >= Wo) continue;
        const float* gp = dy + (size_t)oh * Wo + ow;
        const float* wp = w + ((size_t)ci * Cout * 4 + kh) * 4 + kw;
        for (int co = 0; co < Cout; ++co) acc = fmaf(gp[(size_t)co * Vo], wp[(size_t)co * 16], acc);
      }
    }
    dx[(size_t)ci * V + i] = acc;
  }
}

// one block per (ci, co): 16 tap accumulators per thread, block reduction
__global__ __launch_bounds__(256) void deconv_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy, int Cin, int Cout,
                                                                int H, int W, float* __restrict__ dw) {
  const int ci = blockIdx.x / Cout, co = blockIdx.x % Cout, Ho = 2 * H, Wo = 2 * W;
  const size_t Vo = (size_t)Ho * Wo, V = (size_t)H * W;
  float acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
  for (size_t i = threadIdx.x; i < V; i += 256) {
    const int iw = i % W, ih = i / W;
    const float xv = x[(size_t)ci * V + i];
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const int oh = 2 * ih - 1 + kh;
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const int ow = 2 * iw - 1 + kw;
        if (oh >= 0 && oh < Ho && ow >= 0 && ow < Wo) acc[kh * 4 + kw] = fmaf(xv, dy[(size_t)co * Vo + (size_t)oh * Wo + ow], acc[kh * 4 + kw]);
      }
    }
  }
  __shared__ float red[4][16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float r = wave_sum(acc[t]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = r;
  }
  __syncthreads();
  if (threadIdx.x < 16) dw[((size_t)ci * Cout + co) * 16 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

inline unsigned nb(size_t n) { size_t b = cdivz(n, 256); if (b > 2048) b = 2048; if (b < 1) b = 1; return (unsigned)b; }

}  // namespace

extern "C" int dpi_maxpool2x2_fwd(const float* x, int C, int H, int W, float* y, void* stream) {
  DPI_REQUIRE(x && y && C > 0 && H >= 2 && W >= 2, "maxpool_fwd: bad argument");
  maxpool_fwd_kernel<<<dim3(nb((size_t)(H / 2) * (W / 2)), C), 256, 0, (hipStream_t)stream>>>(x, H, W, H / 2, W / 2, y);
  return dpi_check_launch("maxpool_fwd");
}
extern "C" int dpi_maxpool2x2_bwd(const float* dy, const float* x, int C, int H, int W, float* dx, void* stream) {
  DPI_REQUIRE(dy && x && dx && C > 0 && H >= 2 && W >= 2, "maxpool_bwd: bad argument");
  maxpool_bwd_kernel<<<dim3(nb((size_t)H * W), C), 256, 0, (hipStream_t)stream>>>(dy, x, H, W, H / 2, W / 2, dx);
  return dpi_check_launch("maxpool_bwd");
}
extern "C" int dpi_deconv4x4s2_fwd(const float* x, const float* w, const float* bias, int Cin, int Cout, int H, int W, float* y,
                                   void* stream) {
  DPI_REQUIRE(x && w && y && Cin > 0 && Cout > 0 && H > 0 && W > 0, "deconv_fwd: bad argument");
  deconv_fwd_kernel<<<dim3(nb((size_t)4 * H * W), Cout), 256, 0, (hipStream_t)stream>>>(x, w, bias, Cin, Cout, H, W, y);
  return dpi_check_launch("deconv_fwd");
}
extern "C" int dpi_deconv4x4s2_bwd_data(const float* dy, const float* w, int Cin, int Cout, int H, int W, float* dx, void* stream) {
  DPI_REQUIRE(dy && w && dx && Cin > 0 && Cout > 0 && H > 0 && W > 0, "deconv_bwd_data: bad argument");
  deconv_bwd_data_kernel<<<dim3(nb((size_t)H * W), Cin), 256, 0, (hipStream_t)stream>>>(dy, w, Cin, Cout, H, W, dx);
  return dpi_check_launch("deconv_bwd_data");
}
extern "C" int dpi_deconv4x4s2_bwd_weight(const float* x, const float* dy, int Cin, int Cout, int H, int W, float* dw, void* stream) {
  DPI_REQUIRE(x && dy && dw && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (long)Cin * Cout < (1l << 30), "deconv_bwd_weight: bad argument");
  deconv_bwd_weight_kernel<<<Cin * Cout, 256, 0, (hipStream_t)stream>>>(x, dy, Cin, Cout, H, W, dw);
  return dpi_check_launch("deconv_bwd_weight");
}
