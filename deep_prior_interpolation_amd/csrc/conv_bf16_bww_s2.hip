// bf16-MFMA backward-weight of the 3x3x3 STRIDE-2 convolutions (the four down-sampling layers of the MultiRes-UNet, reference
// mulresunet.py:185-199 `nn.Conv3d(.., 3, stride=2, padding=1)`) for bf16 tensors in the bf16 arithmetic mode:
//
//     dW[co][ci][kd][kh][kw] = sum_{od,oh,ow} dY[co][od][oh][ow] * X[ci][2 od + kd - 1][2 oh + kh - 1][2 ow + kw - 1]        (X = 0 outside)
//
// as   D[co 16][ci 16] += A[co 16][K 32] * B[K 32][ci 16]   with K = OUTPUT voxels along w.  Eight consecutive ow of one dY row are 16
// contiguous bytes = one lane's share of a v_mfma_f32_16x16x32_bf16 operand (lane = (channel lj, octet lk)), so A goes from global memory
// straight into MFMA registers.  B needs X at 2 ow + kw - 1, ow = ow0 .. ow0 + 7: the 18 elements [2 ow0 - 2, 2 ow0 + 16) of an X row are nine
// dwords d0 .. d8 (two bf16 each: one 4-byte and two 16-byte loads, the latter 32-byte aligned), and the three kw fragments are byte
// permutes of them (v_perm_b32) — kw = 0: the high halves of d0 .. d7, kw = 1: the low halves of d1 .. d8, kw = 2: their high halves.
// No LDS, no conversion.  The fp32-MFMA kernel that served these layers (conv_bwd_weight_mfma_kernel<3, 2, 2, 2>) is matrix-bound:
// 25->25 at 256x128x128 0.45 ms for 0.24 GB of compulsory traffic.
// Work split: a workgroup is three waves, wave = depth tap kd (9 accumulators per 16-channel output tile, MT = 2 tiles: 72 registers), all
// three walk the same octets of dY; grid = (octet chunks, 16-channel tiles of Cin, 32-channel tiles of Cout).  Per-chunk partial sums go to
// the workspace [chunk][Cout][Cin][27] and are summed in fixed order by dpi_reduce_chunks — deterministic, no atomics, as everywhere.
// Applies when x and dy are bf16, precision = 1, no producer chain on x, W a multiple of 16 (an octet of ow never straddles a row, every
// 16-byte load is aligned, W = 2 Wo); everything else keeps the fp32-MFMA kernel (conv_bwd_weight.hip).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct BwS2Args {
  const unsigned short* __restrict__ x;
  const unsigned short* __restrict__ dy;
  float* __restrict__ ws;      // [nchunks][Cout][Cin][27]
  int Cin, Cout;
  int D, H, W, Do, Ho, Wo;
  int noct, oct_per_chunk;     // octets of dY per channel (Do * Ho * Wo / 8); per chunk (a multiple of 4)
};

template <int MT>
__global__ __launch_bounds__(192, 4) void conv_bf16_bww_s2_kernel(BwS2Args a) {
  const int tid = threadIdx.x, lane = tid & 63, kd = tid >> 6;          // wave = depth tap
  const int lk = lane >> 4, lj = lane & 15;
  const int ci0 = blockIdx.y * 16, co0 = blockIdx.z * 16 * MT;
  const size_t V = (size_t)a.D * a.H * a.W, Vo = (size_t)a.Do * a.Ho * a.Wo;
  // rows past Cin / Cout re-read the last real channel (their products land in columns / rows that are never written)
  const unsigned short* __restrict__ const xr = a.x + (size_t)min(ci0 + lj, a.Cin - 1) * V;
  const unsigned short* __restrict__ dyr[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) dyr[m] = a.dy + (size_t)min(co0 + 16 * m + lj, a.Cout - 1) * Vo;
  const int mt_valid = min(MT, (a.Cout - co0 + 15) / 16);
  f32x4 acc[MT][9];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wq = a.Wo >> 3;
  const int q0 = blockIdx.x * a.oct_per_chunk, q1 = min(q0 + a.oct_per_chunk, a.noct);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int qb = q0; qb < q1; qb += 4) {
    const int q = qb + lk;                                               // this lane's octet of dY: (od, oh, ow0 .. ow0 + 7)
    const bool valid = q < q1;
    const int ow0 = (q % wq) * 8, t_ = q / wq, oh = t_ % a.Ho, od = t_ / a.Ho;
    const size_t orow = ((size_t)od * a.Ho + oh) * a.Wo + ow0;
    u32x4 ga[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
      if (m < mt_valid) ga[m] = valid ? *reinterpret_cast<const u32x4*>(dyr[m] + orow) : zero4;
    const int id = 2 * od + kd - 1;
    const bool dok = valid && id >= 0 && id < a.D;
    unsigned dl[3], dm[3][8];                                            // per kh: d0, d1 .. d8
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = 2 * oh + kh - 1;
      const bool ok = dok && ih >= 0 && ih < a.H;
      const unsigned short* __restrict__ row = xr + ((size_t)(ok ? id : 0) * a.H + (ok ? ih : 0)) * a.W + 2 * ow0;      // element 2 ow0 = d1's low half
      const u32x4 lo = ok ? *reinterpret_cast<const u32x4*>(row) : zero4, hi = ok ? *reinterpret_cast<const u32x4*>(row + 8) : zero4;
      dl[kh] = (ok && ow0 > 0) ? *reinterpret_cast<const unsigned*>(row - 2) : 0u;       // left of the row: zero padding
      dm[kh][0] = lo.x; dm[kh][1] = lo.y; dm[kh][2] = lo.z; dm[kh][3] = lo.w;
      dm[kh][4] = hi.x; dm[kh][5] = hi.y; dm[kh][6] = hi.z; dm[kh][7] = hi.w;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      // v_perm_b32(first, second, selector): bytes 0-3 of the selector index the SECOND source, 4-7 the first
      u32x4 b0, b1, b2;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const unsigned e0 = p == 0 ? dl[kh] : dm[kh][2 * p - 1];        // d_{2p}
        const unsigned e1 = dm[kh][2 * p], e2 = dm[kh][2 * p + 1];      // d_{2p+1}, d_{2p+2}
        b0[p] = __builtin_amdgcn_perm(e1, e0, 0x07060302u);              // kw = 0: x[2 ow - 1]: high halves of d_{2p}, d_{2p+1}
        b1[p] = __builtin_amdgcn_perm(e2, e1, 0x05040100u);              // kw = 1: x[2 ow]    : low halves of d_{2p+1}, d_{2p+2}
        b2[p] = __builtin_amdgcn_perm(e2, e1, 0x07060302u);              // kw = 2: x[2 ow + 1]: high halves of d_{2p+1}, d_{2p+2}
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if (m < mt_valid) {
          const bf16x8 af = __builtin_bit_cast(bf16x8, ga[m]);
          acc[m][kh * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, b0), acc[m][kh * 3 + 0], 0, 0, 0);
          acc[m][kh * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, b1), acc[m][kh * 3 + 1], 0, 0, 0);
          acc[m][kh * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, b2), acc[m][kh * 3 + 2], 0, 0, 0);
        }
    }
  }
  // D row = co (4 lk + r), column = ci (lj); every wave owns its depth tap's nine slots of [chunk][Cout][Cin][27]
  const int ci = ci0 + lj;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 16 * m + 4 * lk + r;
      if (co < a.Cout && ci < a.Cin) {
        float* __restrict__ o = a.ws + (((size_t)blockIdx.x * a.Cout + co) * a.Cin + ci) * 27 + kd * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) o[t] = acc[m][t][r];
      }
    }
}

struct BwS2Plan { int noct, oct_per_chunk, nchunks, Do, Ho, Wo; };
BwS2Plan bww_s2_plan(const dpi_conv_desc* d) {
  BwS2Plan p{};
  p.Do = (d->D - 1) / 2 + 1; p.Ho = (d->H - 1) / 2 + 1; p.Wo = d->W / 2;
  p.noct = p.Do * p.Ho * (p.Wo / 8);
  const long other = (long)cdiv(d->Cin, 16) * cdiv(d->Cout, 32);
  const size_t per = (size_t)d->Cout * d->Cin * 27;
  long want = 1280 / other;                                              // ONE round of workgroups: 5 of 3 waves are resident per CU at 126 registers
  const long steps = cdiv(p.noct, 4);
  if (want > steps) want = steps;
  const long max_mem = (long)(((size_t)16 << 20) / per);                 // <= 64 MB of partial sums
  if (want > max_mem) want = max_mem;
  if (want < 1) want = 1;
  p.oct_per_chunk = 4 * cdiv((int)steps, (int)want);
  p.nchunks = cdiv(p.noct, p.oct_per_chunk);
  return p;
}

}  // namespace

bool dpi_conv_bf16_bww_s2_usable(const dpi_conv_desc* d) {
  return d->precision == 1 && d->k == 3 && d->kd == 3 && d->stride == 2 && (d->io & DPI_IO_X_BF16) && (d->io & DPI_IO_DY_BF16) && (d->W & 15) == 0
         && (size_t)d->D * d->H * d->W < ((size_t)1 << 31);
}

size_t dpi_conv_bf16_bww_s2_ws_floats(const dpi_conv_desc* d) { return (size_t)bww_s2_plan(d).nchunks * d->Cout * d->Cin * 27; }

int dpi_conv_bf16_bww_s2_run(const dpi_conv_desc* d, const float* x, const float* dy, float* dw, float* ws, hipStream_t st) {
  const BwS2Plan p = bww_s2_plan(d);
  BwS2Args a{reinterpret_cast<const unsigned short*>(x), reinterpret_cast<const unsigned short*>(dy), ws, d->Cin, d->Cout, d->D, d->H, d->W,
             p.Do, p.Ho, p.Wo, p.noct, p.oct_per_chunk};
  const dim3 grid(p.nchunks, cdiv(d->Cin, 16), cdiv(d->Cout, 32));
  if (d->Cout > 16) conv_bf16_bww_s2_kernel<2><<<grid, 192, 0, st>>>(a);
  else conv_bf16_bww_s2_kernel<1><<<grid, 192, 0, st>>>(a);
  if (int e = dpi_check_launch("conv_bf16_bww_s2")) return e;
  dpi_reduce_chunks(ws, dw, (size_t)d->Cout * d->Cin * 27, p.nchunks, st);
  return dpi_check_launch("reduce_chunks");
}
