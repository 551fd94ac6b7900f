// bf16-MFMA backward-weight of the 3x3x3 stride-1 convolutions (mixed-precision mode, BASELINE configs[4]; reference
// mulresunet.py:75-77 — the weight gradients autograd computes for nn.Conv3d):
//
//     dW[co][ci][kd][kh][kw] = sum_{d,h,w} dY[co][d][h][w] * T(X)[ci][d+kd-1][h+kh-1][w+kw-1]        (T = producer's BN + LeakyReLU chain)
//
// Both operands are activations, rounded to bf16 (round-to-nearest-even) while they are staged into LDS; products are exact in
// fp32 and accumulate in fp32 (v_mfma_f32_16x16x32_bf16).  At 16x the fp32 matrix rate the kernel is bound by streaming X and dY
// once, so the layout serves the loads:
//   * GEMM per tap:  D[co 16][ci 16] += A[co 16][K 32] * B[K 32][ci 16],  K = 32 consecutive w of one (d, h) row.
//     Lane (i = l & 15, g = l >> 4) holds the 8 consecutive w of octet g: one ds_read_b128, and with rows stored as
//     [row][channel 16][32 w] a whole fragment is ONE contiguous 1 KB read — conflict-free by construction.
//   * The kw shift would misalign the 16-byte fragment reads of X, so it is moved to dY: X rows are stored once, aligned, and dY
//     is stored as three copies shifted by +1 / 0 / -1 (dY has no halo in d, h — the copies cost what an X halo in w would).
//     The sum over w is partitioned by the X position, over d, h by the dY position.
//   * A workgroup owns a band (8 rows x 32 columns) and walks a range of depth slices with the X slices in a 4-slot ring: every X
//     slice is loaded once per band (halo only in h: 10 rows per 8) and serves the three kd taps from LDS.
//   * Wave w owns dY rows 2w, 2w+1 of the band and all 27 taps (27 accumulators); an X row fragment is multiplied with the dY
//     fragments of the rows it pairs with (kh = x row - dY row) for the three kw copies: 54 MFMAs per 12 + 6 fragment reads.
//   * Per-chunk partial sums go to the workspace [chunk][Cout][Cin][27] (waves reduced through LDS first) and are summed in
//     fixed order by dpi_reduce_chunks — deterministic, no atomics, as the fp32 kernels.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct BwBArgs {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ dy;
  float* __restrict__ ws;      // [nchunks][Cout][Cin][27]
  int Cin, Cout;
  int D, H, W;
  int nth, ntw, ndc, dlen;     // bands (8 rows x 32 columns), depth chunks per band, slices per chunk
  int xb, dyb;                 // storage type of x / dy in HBM: 1 = bf16 (8-byte pieces of 4 values), 0 = fp32 (16-byte pieces)
};

constexpr int TW = 32;
constexpr int ROWW = 16 * 16;                 // words of one [16 channels][32 w bf16] row block (1 KB)

__device__ __forceinline__ unsigned pack2(float lo, float hi) {      // one v_cvt_pk_bf16_f32
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// exact three-term split of two fp32 values, each term packed (lo | hi << 16): x = h + m + l, every difference exact in fp32
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = pack2(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = pack2(r0, r1);
  const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = pack2(s0, s1);
}
__device__ __forceinline__ float from_prev_lane(float v) {            // lane j <- lane j-1 (inside a row of 16 lanes)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_next_lane(float v) {            // lane j <- lane j+1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
}

// NS = 1: bf16 mode, band of TH = 8 rows, 64 KB LDS (two workgroups per CU).
// NS = 3: split mode — every operand as three bf16 terms (exact), the six partial products >= 2^-16 of the full one accumulated
//         in fp32 (see conv_bf16_mfma.hip): fp32-class accuracy at 6 MFMAs per tap.  Three copies of everything in LDS, so the band
//         is TH = 4 rows (108 KB, one workgroup per CU) with 162 MFMAs per wave and slice.  Measured 25->16: 0.78-0.80 ms against a
//         0.3 ms matrix floor and a 0.23 ms HBM floor: the two phases of a slice (loads in flight / MFMAs) are of equal length and
//         do not overlap well.  Tried without gain: a second register stage of prefetch; an 8-wave workgroup whose wave pairs
//         share a dY row and split the six products (all waves still stage and multiply in the same phases — it needs
//         producer / consumer waves, not a symmetric split).
template <int NS, int TH, bool XB = false, bool DYB = false>        // XB / DYB: x / dy stored as bf16 (template parameters, as in conv_bf16_mfma.hip)
__global__ __launch_bounds__(256, NS == 1 ? 2 : 1) void conv_bf16_bwd_weight_kernel(BwBArgs a) {
  a.xb = XB; a.dyb = DYB;
  constexpr int XR = TH + 2;                    // X rows of a slice (halo in h)
  constexpr int XSLOT = XR * ROWW;              // one X slice of the ring
  constexpr int DCOPY = TH * ROWW;              // one shifted copy of the dY slice
  constexpr int XT = 4 * XSLOT, DT = 3 * DCOPY; // words per term
  constexpr int LDSW = NS * (XT + DT);
  constexpr int EX = XR / 2, EY = TH / 2;       // float4 pieces per thread and slice
  constexpr int RW = TH / 4;                    // dY rows per wave
  constexpr int PF = 1;                         // slices of prefetch held in registers (2 measured no faster in split mode: its
                                                // single workgroup per CU serialises staging, LDS reads and MFMAs, not HBM latency)
  static_assert(LDSW >= 2 * 6912, "the wave reduction reuses the operand buffers");
  __shared__ __attribute__((aligned(16))) unsigned lds[LDSW];
  unsigned* const xl = lds;                     // [term][slot][row][channel][16 words]
  unsigned* const dl = lds + NS * XT;           // [term][copy][row][channel][16 words]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ci0 = blockIdx.y * 16, co0 = blockIdx.z * 16;
  int b = blockIdx.x;
  const int dc = b % a.ndc; b /= a.ndc;
  const int tw_i = b % a.ntw, th_i = b / a.ntw;
  const int d0 = dc * a.dlen, d1 = min(a.D, d0 + a.dlen);
  const int oh0 = th_i * TH, u0 = tw_i * TW;
  const size_t V = (size_t)a.D * a.H * a.W;
  const int HW = a.H * a.W;

  // staging map: thread -> (float4 piece q of a 32-w row, channel ch, rows 2 e + rh)
  const int q = tid & 7, ch = (tid >> 3) & 15, rh = tid >> 7;
  const int gw = u0 + 4 * q;
  const bool xch = ci0 + ch < a.Cin, ych = co0 + ch < a.Cout;
  const float* __restrict__ xc = dpi_at(a.x, (size_t)(xch ? ci0 + ch : 0) * V, a.xb);
  const float* __restrict__ yc = dpi_at(a.dy, (size_t)(ych ? co0 + ch : 0) * V, a.dyb);
  const Chain cx = load_chain(a.chain, xch ? ci0 + ch : 0);
  int xoff[EX], yoff[EY];
#pragma unroll
  for (int e = 0; e < EX; ++e) {
    const int gh = oh0 - 1 + 2 * e + rh;
    xoff[e] = (xch && gh >= 0 && gh < a.H && gw < a.W) ? gh * a.W + gw : -1;
  }
#pragma unroll
  for (int e = 0; e < EY; ++e) {
    const int gh = oh0 + 2 * e + rh;
    yoff[e] = (ych && gh < a.H && gw < a.W) ? gh * a.W + gw : -1;
  }
  // the dY columns next to the run (w = u0 - 1, u0 + 32) feed the shifted copies: fetched by the first / last piece of a row
  const int hdelta = q == 0 ? (u0 > 0 ? -1 : 0) : q == 7 ? (u0 + TW < a.W ? 4 : 0) : 0;

  // PF register stages of prefetch: stage 0 is stored next, stage PF-1 was loaded last
  float4 xr[PF][EX], yr[PF][EY];
  float hr[PF][EY];
  bool x_live[PF];
  auto load_x = [&](int st, int slice) {
    x_live[st] = slice >= 0 && slice < a.D;
    const float* __restrict__ p = dpi_at(xc, (size_t)(x_live[st] ? slice : 0) * HW, a.xb);
    // (bf16 tensors: raw 8-byte pieces held in .x / .y, widened by store_x — "load now, widen later", common.h; ONE uniform branch
    //  around the slice's loads)
    if (a.xb) {
#pragma unroll
      for (int e = 0; e < EX; ++e)
        xr[st][e] = (x_live[st] && xoff[e] >= 0) ? dpi_ld4_raw_bf16(p, xoff[e]) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int e = 0; e < EX; ++e)
        xr[st][e] = (x_live[st] && xoff[e] >= 0) ? *reinterpret_cast<const float4*>(p + xoff[e]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto load_dy = [&](int st, int slice) {
    const bool live = slice < d1;
    const float* __restrict__ p = dpi_at(yc, (size_t)(live ? slice : 0) * HW, a.dyb);
    if (a.dyb) {
#pragma unroll
      for (int e = 0; e < EY; ++e) {
        yr[st][e] = (live && yoff[e] >= 0) ? dpi_ld4_raw_bf16(p, yoff[e]) : make_float4(0.f, 0.f, 0.f, 0.f);
        hr[st][e] = (live && hdelta != 0 && yoff[e] >= 0)
                        ? __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(p)[yoff[e] + hdelta]) : 0.f;     // raw
      }
    } else {
#pragma unroll
      for (int e = 0; e < EY; ++e) {
        yr[st][e] = (live && yoff[e] >= 0) ? *reinterpret_cast<const float4*>(p + yoff[e]) : make_float4(0.f, 0.f, 0.f, 0.f);
        hr[st][e] = (live && hdelta != 0 && yoff[e] >= 0) ? p[yoff[e] + hdelta] : 0.f;
      }
    }
  };
  auto advance = [&]() {
#pragma unroll
    for (int st = 0; st + 1 < PF; ++st) {
      x_live[st] = x_live[st + 1];
#pragma unroll
      for (int e = 0; e < EX; ++e) xr[st][e] = xr[st + 1][e];
#pragma unroll
      for (int e = 0; e < EY; ++e) { yr[st][e] = yr[st + 1][e]; hr[st][e] = hr[st + 1][e]; }
    }
  };
  // two packed pairs -> the NS term planes (stride `ts` words) at `o`
  auto put = [&](unsigned* __restrict__ o, int ts, float p0, float p1, float p2, float p3) {
    if constexpr (NS == 1) {
      *reinterpret_cast<u32x2*>(o) = (u32x2){pack2(p0, p1), pack2(p2, p3)};
    } else {
      unsigned h0, m0, l0, h1, m1, l1;
      split3_pair(p0, p1, h0, m0, l0);
      split3_pair(p2, p3, h1, m1, l1);
      *reinterpret_cast<u32x2*>(o) = (u32x2){h0, h1};
      *reinterpret_cast<u32x2*>(o + ts) = (u32x2){m0, m1};
      *reinterpret_cast<u32x2*>(o + 2 * ts) = (u32x2){l0, l1};
    }
  };
  auto store_x = [&](int slot) {
    unsigned* __restrict__ dst = xl + slot * XSLOT + (rh * 16 + ch) * 16 + 2 * q;
#pragma unroll
    for (int e = 0; e < EX; ++e) {
      float4 v = xr[0][e];
      if (a.xb) {
        if (NS == 1 && !a.chain) {                                 // bf16 in, bf16 operands, no chain: the loaded dwords ARE the packed pairs
          *reinterpret_cast<u32x2*>(dst + 2 * e * ROWW) = (u32x2){__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y)};
          continue;
        }
        v = dpi_widen_raw4(v);
      }
      if (a.chain && x_live[0] && xoff[e] >= 0) {                 // zero padding stays zero
        v.x = apply_chain(cx, v.x); v.y = apply_chain(cx, v.y); v.z = apply_chain(cx, v.z); v.w = apply_chain(cx, v.w);
      }
      put(dst + 2 * e * ROWW, XT, v.x, v.y, v.z, v.w);
    }
  };
  auto store_dy = [&]() {
    unsigned* __restrict__ dst = dl + (rh * 16 + ch) * 16 + 2 * q;
#pragma unroll
    for (int e = 0; e < EY; ++e) {
      const float4 v = a.dyb ? dpi_widen_raw4(yr[0][e]) : yr[0][e];
      const float hv = a.dyb ? dpi_widen_raw(hr[0][e]) : hr[0][e];
      float left = from_prev_lane(v.w), right = from_next_lane(v.x);
      left = q == 0 ? hv : left;
      right = q == 7 ? hv : right;
      unsigned* __restrict__ o = dst + 2 * e * ROWW;
      put(o, DT, v.y, v.z, v.w, right);                        // kw = 0: copy[u] = dY[u + 1]
      put(o + DCOPY, DT, v.x, v.y, v.z, v.w);                  // kw = 1
      put(o + 2 * DCOPY, DT, left, v.x, v.y, v.z);             // kw = 2: copy[u] = dY[u - 1]
    }
  };

  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ring slot of slice s is (s + 1) & 3
  load_x(0, d0 - 1); store_x(d0 & 3);
  load_x(0, d0); store_x((d0 + 1) & 3);
#pragma unroll
  for (int st = 0; st < PF; ++st) { load_x(st, d0 + 1 + st); load_dy(st, d0 + st); }

  const int fo = (lane & 15) * 16 + (lane >> 4) * 4;           // this lane's 16 bytes inside a row block
  auto frag = [&](const unsigned* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p)); };
  for (int d = d0; d < d1; ++d) {
    __syncthreads();                                           // the previous slice's dY copies have been read
    store_x((d + 2) & 3);                                      // slice d + 1
    store_dy();
    __syncthreads();
    advance();
    if (d + PF < d1) { load_x(PF - 1, d + PF + 1); load_dy(PF - 1, d + PF); }   // in flight behind the MFMAs of PF slices
    bf16x8 A[NS][3][RW];
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < RW; ++r) A[n][s][r] = frag(dl + n * DT + s * DCOPY + (RW * wid + r) * ROWW + fo);
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      const unsigned* __restrict__ xs = xl + ((d + kd) & 3) * XSLOT + RW * wid * ROWW + fo;     // slice d - 1 + kd
#pragma unroll
      for (int t = 0; t < RW + 2; ++t) {                       // X row RW wid + t of the slice pairs with dY row r at kh = t - r
        bf16x8 B[NS];
#pragma unroll
        for (int n = 0; n < NS; ++n) B[n] = frag(xs + n * XT + t * ROWW);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          const int kh = t - r;
          if (kh < 0 || kh > 2) continue;
          f32x4* const c = &acc[(kd * 3 + kh) * 3];
          if constexpr (NS == 1) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) c[kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0][kw][r], B[0], c[kw], 0, 0, 0);
          } else {
            // smallest terms first: l*h, h*l, m*m (2^-16), m*h, h*m (2^-8), h*h; the three kw accumulators alternate so that
            // consecutive MFMAs never wait for each other's result
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) c[kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[TA[p]][kw][r], B[TB[p]], c[kw], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- the four waves' partial sums -> one, through LDS (6912 floats per wave), then the chunk's slot of the workspace ----
  float* const red = reinterpret_cast<float*>(lds);
  __syncthreads();
  if (wid >= 2) {
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wid - 2) * 6912 + (t * 4 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (wid < 2) {
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] += red[wid * 6912 + (t * 4 + r) * 64 + lane];
  }
  __syncthreads();
  if (wid == 1) {
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(t * 4 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (wid == 0) {
    const int ci = ci0 + (lane & 15);
    float* __restrict__ w = a.ws + (size_t)blockIdx.x * a.Cout * a.Cin * 27;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 4 * (lane >> 4) + r;
      if (co < a.Cout && ci < a.Cin) {
        float* __restrict__ o = w + ((size_t)co * a.Cin + ci) * 27;
#pragma unroll
        for (int t = 0; t < 27; ++t) o[t] = acc[t][r] + red[(t * 4 + r) * 64 + lane];
      }
    }
  }
}

struct BwBPlan { int nth, ntw, ndc, dlen, nchunks; };

}  // namespace

bool dpi_bf16_force_all();

static BwBPlan bf16_bww_plan(const dpi_conv_desc* d) {
  BwBPlan p{};
  const int TH = d->precision == 2 ? 4 : 8;
  p.nth = cdiv(d->H, TH); p.ntw = cdiv(d->W, TW);
  const int bands = p.nth * p.ntw, blocks = cdiv(d->Cin, 16) * cdiv(d->Cout, 16);
  const size_t per = (size_t)d->Cout * d->Cin * 27;
  // depth chunks per band: ~1536 workgroups (3 rounds of 2 per CU); split mode: ~1024 (4 rounds of 1 per CU)
  size_t want = cdivz(d->precision == 2 ? 1024 : 1536, (size_t)bands * blocks);
  const size_t mem = (((size_t)32 << 20) / per) / bands;       // the workspace stays <= 128 MB
  if (want > mem) want = mem;
  if (want > (size_t)cdiv(d->D, 4)) want = cdiv(d->D, 4);      // >= 4 slices per chunk: two extra X slices are staged per chunk
  if (want < 1) want = 1;
  p.dlen = cdiv(d->D, (int)want);
  p.ndc = cdiv(d->D, p.dlen);
  p.nchunks = bands * p.ndc;
  return p;
}

// Where it applies: precision = 1 (bf16 mode), 3x3x3 stride 1, rows a whole number of float4 (the staging loads are 16 bytes).
// Measured against the fp32 kernels on every level of the default net at 256x128x128 (tools/bench_conv.py --which bwd_weight):
// 25->16 @256x128x128 1.20 -> 0.29 ms, 64->4 0.62 -> 0.34, 51->32 @128x64x64 0.48 -> 0.11, 105->64 @64x32x32 0.28 -> 0.07,
// 212->128 @32x16x16 0.26 -> 0.065, 142->213 @16x8x8 0.090 -> 0.036: faster everywhere, so no size threshold.
bool dpi_conv_bf16_bww_usable(const dpi_conv_desc* d) {
  if (d->precision < 1 || d->k != 3 || d->kd != 3 || d->stride != 1 || (d->W & 3)) return false;
  if (d->precision == 2 && (d->io & (DPI_IO_X_BF16 | DPI_IO_DY_BF16))) return false;       // the split instantiation is compiled for fp32 tensors
  // split mode is matrix-bound (6 MFMAs per tap on 16 x 16 channel blocks): it beats the fp32 kernels where both channel counts
  // fill a block (25->16 1.20 -> 0.77 ms, 51->32 0.48 -> 0.34, 105->64 0.28 -> 0.19, 212->128 0.26 -> 0.19) and loses on the
  // few-channel layers (64->4 0.62 -> 1.29, 8->13 0.33 -> 0.37, 137->8 0.33 -> 0.41), which keep the fp32 kernels
  if (d->precision == 2 && !dpi_bf16_force_all() && (d->Cin < 16 || d->Cout < 16)) return false;
  return (size_t)d->D * d->H * d->W < ((size_t)1 << 29);
}

size_t dpi_conv_bf16_bww_ws_floats(const dpi_conv_desc* d) { return (size_t)bf16_bww_plan(d).nchunks * d->Cout * d->Cin * 27; }

int dpi_conv_bf16_bww_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* dy, float* dw, float* ws, hipStream_t st) {
  const BwBPlan p = bf16_bww_plan(d);
  BwBArgs a{x, chain, dy, ws, d->Cin, d->Cout, d->D, d->H, d->W, p.nth, p.ntw, p.ndc, p.dlen, (d->io & DPI_IO_X_BF16) != 0, (d->io & DPI_IO_DY_BF16) != 0};
  dim3 grid(p.nchunks, cdiv(d->Cin, 16), cdiv(d->Cout, 16));
  if (d->precision == 2) conv_bf16_bwd_weight_kernel<3, 4><<<grid, 256, 0, st>>>(a);
  else if (a.xb && a.dyb) conv_bf16_bwd_weight_kernel<1, 8, true, true><<<grid, 256, 0, st>>>(a);
  else if (a.xb) conv_bf16_bwd_weight_kernel<1, 8, true, false><<<grid, 256, 0, st>>>(a);
  else if (a.dyb) conv_bf16_bwd_weight_kernel<1, 8, false, true><<<grid, 256, 0, st>>>(a);
  else conv_bf16_bwd_weight_kernel<1, 8><<<grid, 256, 0, st>>>(a);
  if (int e = dpi_check_launch("conv_bf16_bwd_weight")) return e;
  dpi_reduce_chunks(ws, dw, (size_t)d->Cout * d->Cin * 27, p.nchunks, st);
  return dpi_check_launch("reduce_chunks");
}
