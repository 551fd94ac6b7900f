// Shared device helpers and host-side error plumbing for libdpi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dpi_hip.h"
#include "dpi_hip_internal.h"

#define DPI_WAVE 64

void dpi_set_error(const char* fmt, ...);
// packed-weight scratch (conv_bf16_mfma.hip): the slot of one (weight tensor, shape, tag), allocated at its first use; nullptr + error text on failure
void* dpi_pack_slot(const void* w, int kd, int cin, int cout, int tag, size_t nbytes);
size_t dpi_pack_max_slot();
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

// ---- activation storage types (ABI 400) -------------------------------------------------------------
// An activation tensor (or the gradient of one) lives in HBM as fp32 or as bf16 (BASELINE configs[4]: bf16 activations, fp32
// master weights / BatchNorm statistics / Adam).  Arithmetic is fp32 either way: a bf16 element is widened on load (exact) and a
// result is rounded to nearest-even on store.  The `bf` flags are wave-uniform kernel arguments (dpi_conv_desc.io, the `io` masks of
// the *_io entry points); pointers keep their `float*` spelling in the argument structs and are indexed in ELEMENTS through these
// helpers only.  Statistics that describe a stored tensor (BatchNorm partials in a conv epilogue) are taken of the ROUNDED values.
typedef float dpi_f32x2 __attribute__((ext_vector_type(2)));
typedef float dpi_f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned dpi_u32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 dpi_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dpi_bf16_to_f32(unsigned h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ unsigned dpi_pack_bf16(float lo, float hi) {        // one v_cvt_pk_bf16_f32 (round to nearest even)
  const dpi_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dpi_bf16x2));
}
__device__ __forceinline__ unsigned short dpi_f32_to_bf16(float f) { return (unsigned short)(dpi_pack_bf16(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float dpi_round_bf16(float f) { return dpi_bf16_to_f32(dpi_pack_bf16(f, 0.f) & 0xffffu); }
// what a store of `v` into a tensor of that type leaves there
__device__ __forceinline__ float dpi_stored(float v, bool bf) { return bf ? dpi_round_bf16(v) : v; }
__device__ __forceinline__ float dpi_ld(const float* base, size_t i, bool bf) {
  return bf ? dpi_bf16_to_f32(reinterpret_cast<const unsigned short*>(base)[i]) : base[i];
}
__device__ __forceinline__ void dpi_st(float* base, size_t i, float v, bool bf) {
  if (bf) reinterpret_cast<unsigned short*>(base)[i] = dpi_f32_to_bf16(v);
  else base[i] = v;
}
// address of element i (a channel plane, a row) as a `float*` spelled pointer for further dpi_ld / dpi_st / dpi_buffer_t use
__device__ __forceinline__ const float* dpi_at(const float* base, size_t i, bool bf) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + i * (bf ? 2 : 4));
}
__device__ __forceinline__ float* dpi_at(float* base, size_t i, bool bf) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + i * (bf ? 2 : 4));
}
// buffer resource over n ELEMENTS of that type, and a load of element `idx` through it (idx < 0: out of range -> 0, as dpi_buffer_load)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dpi_buffer_t(const float* base, size_t n, bool bf) { return dpi_buffer(base, n * (bf ? 2 : 4)); }
__device__ __forceinline__ float dpi_buffer_load_bf16(__amdgpu_buffer_rsrc_t r, int idx) {
  return dpi_bf16_to_f32((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, idx * 2, 0, 0));
}
// LOAD NOW, WIDEN LATER.  In a software-pipelined kernel the loads of a whole tile are issued back to back and consumed a phase later.
// A load that is widened where it is issued (`bits << 16`) needs its data there: inside the `if (bf16) ... else ...` of a run-time flag the
// wait lands in the branch, every unrolled copy of the branch waits for its own load before the next is issued, and the tile's loads
// serialise (measured: the staging phase of conv_bf16_kernel and conv_bf16_bwd_weight_kernel 1.5-2.7x slower with HALF the bytes).
// So the prefetch keeps the raw 16 bits (zero-extended by the load instruction itself, held in the float register bit for bit) and the
// consumer widens them right before use.
__device__ __forceinline__ float dpi_buffer_load_bf16_raw(__amdgpu_buffer_rsrc_t r, int idx) {
  return __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, idx * 2, 0, 0));
}
__device__ __forceinline__ float dpi_widen_raw(float raw) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, raw) << 16); }
// a float4-shaped holder of four raw bf16 (x, y = the two dwords as loaded) -> four floats
__device__ __forceinline__ float4 dpi_widen_raw4(float4 h) {
  const unsigned u0 = __builtin_bit_cast(unsigned, h.x), u1 = __builtin_bit_cast(unsigned, h.y);
  return make_float4(__builtin_bit_cast(float, u0 << 16), __builtin_bit_cast(float, u0 & 0xffff0000u),
                     __builtin_bit_cast(float, u1 << 16), __builtin_bit_cast(float, u1 & 0xffff0000u));
}
// eight raw bytes (four bf16) at element i of a bf16 tensor into such a holder; no ALU work on the data
__device__ __forceinline__ float4 dpi_ld4_raw_bf16(const float* base, size_t i) {
  const dpi_u32x2v u = *reinterpret_cast<const dpi_u32x2v*>(reinterpret_cast<const unsigned short*>(base) + i);
  // (hipcc: __builtin_bit_cast applied to a vector SUBSCRIPT — bit_cast(float, u[1]) — silently yields element 0: the 8-byte load became a
  //  4-byte load with its dword duplicated.  Copy the elements to scalars first.)
  const unsigned u0 = u.x, u1 = u.y;
  return make_float4(__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1), 0.f, 0.f);
}
// four consecutive elements starting at element i (i % 4 == 0 and an aligned base: 16-byte / 8-byte accesses)
__device__ __forceinline__ float4 dpi_ld4(const float* base, size_t i, bool bf, bool nt) {
  if (bf) {
    const dpi_u32x2v* p = reinterpret_cast<const dpi_u32x2v*>(reinterpret_cast<const unsigned short*>(base) + i);
    const dpi_u32x2v u = nt ? __builtin_nontemporal_load(p) : *p;
    const unsigned u0 = u.x, u1 = u.y;
    return make_float4(__builtin_bit_cast(float, u0 << 16), __builtin_bit_cast(float, u0 & 0xffff0000u),
                       __builtin_bit_cast(float, u1 << 16), __builtin_bit_cast(float, u1 & 0xffff0000u));
  }
  const dpi_f32x4v* p = reinterpret_cast<const dpi_f32x4v*>(base + i);
  const dpi_f32x4v v = nt ? __builtin_nontemporal_load(p) : *p;
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void dpi_st4(float* base, size_t i, float4 v, bool bf, bool nt) {
  if (bf) {
    dpi_u32x2v* p = reinterpret_cast<dpi_u32x2v*>(reinterpret_cast<unsigned short*>(base) + i);
    const dpi_u32x2v u = {dpi_pack_bf16(v.x, v.y), dpi_pack_bf16(v.z, v.w)};
    if (nt) __builtin_nontemporal_store(u, p); else *p = u;
    return;
  }
  dpi_f32x4v* p = reinterpret_cast<dpi_f32x4v*>(base + i);
  const dpi_f32x4v w = {v.x, v.y, v.z, v.w};
  if (nt) __builtin_nontemporal_store(w, p); else *p = w;
}
// MFMA epilogues hold one voxel per lane (16 consecutive voxels of a row per 16-lane group).  Stored as bf16 that is 2 bytes per lane:
// twice the store instructions per byte of a float store.  Where element indices are even-aligned (`pairs`, wave-uniform) the even lanes
// fetch their right neighbour's value (DPP row_shl:1) and store ONE dword for both.  `i` = this lane's element index, `ok` = this lane's
// element lies inside the tensor (for a pair both do: even row length).  Call from converged code (every lane of the 16-group active).
__device__ __forceinline__ void dpi_st_bf16_row(float* base, size_t i, float v, bool ok, bool pairs, int lane_in_row) {
  if (pairs) {
    const float nb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));   // lane j <- lane j + 1
    if (ok && !(lane_in_row & 1)) *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(base) + i) = dpi_pack_bf16(v, nb);
  } else if (ok) {
    reinterpret_cast<unsigned short*>(base)[i] = dpi_f32_to_bf16(v);
  }
}
// storage types of one convolution launch: input / output tensor of THAT launch (forward: x / y; backward-data: dy / dx)
static inline bool dpi_io_in(const dpi_conv_desc* d, bool flip) { return (d->io & (flip ? DPI_IO_DY_BF16 : DPI_IO_X_BF16)) != 0; }
static inline bool dpi_io_out(const dpi_conv_desc* d, bool flip) { return (d->io & (flip ? DPI_IO_DX_BF16 : DPI_IO_Y_BF16)) != 0; }

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

