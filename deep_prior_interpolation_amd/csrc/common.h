// Shared device helpers and host-side error plumbing for libdpi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dpi_hip.h"

#define DPI_WAVE 64

void dpi_set_error(const char* fmt, ...);
int dpi_check_launch(const char* what);
// Validates a caller's descriptor (size field first: a binding built against another struct layout is rejected, not read past).
int dpi_check_conv_desc(const dpi_conv_desc* d);

#define DPI_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      dpi_set_error(__VA_ARGS__);         \
      return DPI_E_ARG;                   \
    }                                     \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }
// optional second input of the MFMA stencil kernel (conv_mfma.hip): y += W2 * x2 through a 1x1(x1) kernel at the output positions
struct MfmaSecond { const float* x2; const float* w2; int C2; long w2_co_stride, w2_c_stride; };

// ---- per-channel load transform ("chain"): T(x) = qs*act(ps*x+pb)+qb -----------------------------
struct Chain {
  float ps, pb, slope, qs, qb;
};
__device__ __forceinline__ Chain load_chain(const float* __restrict__ chain, int c) {
  Chain t;
  if (chain) {
    const float* p = chain + (size_t)c * DPI_CHAIN_STRIDE;
    t.ps = p[0]; t.pb = p[1]; t.slope = p[2]; t.qs = p[3]; t.qb = p[4];
  } else {
    t.ps = 1.f; t.pb = 0.f; t.slope = 1.f; t.qs = 1.f; t.qb = 0.f;
  }
  return t;
}
__device__ __forceinline__ float apply_chain(const Chain& t, float x) {
  float v = fmaf(t.ps, x, t.pb);
  v = v > 0.f ? v : v * t.slope;
  return fmaf(t.qs, v, t.qb);
}

// ---- raw buffer loads ------------------------------------------------------------------------------
// A buffer load whose byte offset is >= num_records returns 0 in hardware (checked per dword; verified on gfx950 by
// tools/ubench/buf_oob.hip).  Zero padding and ragged tile edges therefore need no branch, no select and no 64-bit
// address arithmetic: the per-lane offset is a 32-bit VGPR, "outside" is any negative offset.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dpi_buffer(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes > 0xfffffffcull ? (int)0xfffffffc : (int)bytes, 0x00020000);
}
__device__ __forceinline__ float dpi_buffer_load(__amdgpu_buffer_rsrc_t r, int byte_offset) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_offset, 0, 0));
}

// ---- wave / block reductions (wave = 64 lanes) ------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum `v` over the whole block; result valid in thread 0.  `sh` needs >= blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

// Final reduction over chunks, deterministic (fixed order).  A workgroup owns OB consecutive outputs x (256 / OB) chunk
// groups: thread (o, g) sums chunks g, g + NG, ... of output o with coalesced rows (the earlier 8-lanes-per-output layout
// read 32-byte pieces of 8 different chunk rows per instruction), 4 loads in flight, then the group sums are added in
// order through LDS.  OB = 64 normally, 16 when there are few outputs and many chunks (more groups, shorter chains).
template <int OB>
__global__ __launch_bounds__(256) void dpi_reduce_chunks_kernel(const float* __restrict__ ws, float* __restrict__ out, size_t n, int nchunks) {
  constexpr int NG = 256 / OB;
  __shared__ float part[NG][OB];
  const int o = threadIdx.x % OB, g = threadIdx.x / OB;
  const size_t i = (size_t)blockIdx.x * OB + o;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int c = g;
    for (; c + 3 * NG < nchunks; c += 4 * NG) {
      s0 += ws[(size_t)c * n + i]; s1 += ws[(size_t)(c + NG) * n + i];
      s2 += ws[(size_t)(c + 2 * NG) * n + i]; s3 += ws[(size_t)(c + 3 * NG) * n + i];
    }
    for (; c < nchunks; c += NG) s0 += ws[(size_t)c * n + i];
  }
  part[g][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NG; ++k) s += part[k][o];
    out[i] = s;
  }
}
static inline void dpi_reduce_chunks(const float* ws, float* out, size_t n, int nchunks, hipStream_t st) {
  if (n < 8192 && nchunks > 64) dpi_reduce_chunks_kernel<16><<<(unsigned)cdivz(n, 16), 256, 0, st>>>(ws, out, n, nchunks);
  else dpi_reduce_chunks_kernel<64><<<(unsigned)cdivz(n, 64), 256, 0, st>>>(ws, out, n, nchunks);
}

