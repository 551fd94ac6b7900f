// Forward 3x3x3 stride-1 convolution with FEW output channels (Cout <= 4: the 64->4 / 67->4 first convs of the
// full-resolution MultiRes blocks and the 25->1 output conv, reference mulresunet.py:75-77,244) on the fp32 matrix cores.
//
// With the output channels in the 16 MFMA rows only 4 of 16 rows would be used.  Here the rows are (co, kw) pairs:
//     P[(co,kw)][w'] = sum_{ci,kd,kh} W[co][ci][kd][kh][kw] * X[ci][d+kd-1][h+kh-1][w']        (no shift along W)
//     Y[co][w]       = P[(co,0)][w-1] + P[(co,1)][w] + P[(co,2)][w+1]
// so one MFMA covers three taps (9 MFMAs per 4 input channels and 16 columns instead of 27) and the kw shift becomes a
// 3-term add across neighbouring lanes in the epilogue (DPP row shifts inside the 16-lane column groups).
//   A (weights): lane (row i = l&15 -> co = i>>2, kw = i&3 [kw = 3: zero row], k = l>>4 -> ci)    registers, 9 per chunk
//   B (input)  : lane (k = l>>4 -> ci, column j = l&15)                                           LDS halo tile
//   D          : lane (column j, rows 4*(l>>4) + r) -> co = l>>4, kw = r: one lane holds the three kw partials of its co
// Tile: 4 waves = 4 depth slices x 8 rows x 48 input columns (3 column blocks) -> 46 output columns.
#include "common.h"

void dpi_conv_out_dims(const dpi_conv_desc* d, int* Do, int* Ho, int* Wo);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FcArgs {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ w;      // [Cout][Cin][27]
  const float* __restrict__ bias;
  float* __restrict__ y;
  double* __restrict__ partials;    // [ntiles][Cout][2] or NULL
  int Cin, Cout;
  int D, H, W;
  int ntd, nth, ntw;
};

constexpr int TZ = 4, NR = 8, NHQ = 3;
constexpr int TWO = 16 * NHQ - 2;                    // 46 output columns per tile
constexpr int ID = TZ + 2, IH = NR + 2, IW = 16 * NHQ;
constexpr int RS = IW;                               // 48
constexpr int DS = IH * RS;                          // 480
constexpr int CS0 = ID * DS;                         // 2880 = 0 (mod 32)
constexpr int CS = CS0 + 16;                         // = 16 (mod 32): the two channels of a half-wave hit disjoint banks
constexpr int TILE = ID * IH * IW;                   // 2880
constexpr int E = (TILE + 255) / 256;                // 12 (11.25)

__device__ __forceinline__ int xcd_tile(int bid, int ntiles) {
  const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

__device__ __forceinline__ float dpp_prev(float from_prev_block, float v) {   // lane j <- v[j-1]; lane 0 <- from_prev_block[15]
  const int wrap = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, from_prev_block), 0x121, 0xf, 0xf, false);   // row_ror:1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(wrap, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));   // row_shr:1
}
__device__ __forceinline__ float dpp_next(float from_next_block, float v) {   // lane j <- v[j+1]; lane 15 <- from_next_block[0]
  const int wrap = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, from_next_block), 0x12f, 0xf, 0xf, false);   // row_ror:15
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(wrap, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));   // row_shl:1
}

__global__ __launch_bounds__(256, 2) void conv_fewco_mfma_kernel(FcArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[4 * CS];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  const int tile_id = xcd_tile(blockIdx.x, gridDim.x);
  int bt = tile_id;
  const int tw_i = bt % a.ntw; bt /= a.ntw;
  const int th_i = bt % a.nth; bt /= a.nth;
  const int od0 = bt * TZ, oh0 = th_i * NR, ow0 = tw_i * TWO;
  const size_t V = (size_t)a.D * a.H * a.W;

  // halo-tile slots of this thread: input voxel (od0-1+dz, oh0-1+hy, ow0-1+col)
  int goff[E], loff[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int idx = tid + e * 256;
    const int col = idx % IW, row = idx / IW;
    const int hy = row % IH, dz = row / IH;
    const int gd = od0 - 1 + dz, gh = oh0 - 1 + hy, gw = ow0 - 1 + col;
    const bool ok = idx < TILE && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    goff[e] = ok ? (gd * a.H + gh) * a.W + gw : -1;
    loff[e] = idx < TILE ? dz * DS + hy * RS + col : -1;
  }
  auto stage_load = [&](float (&sr)[4][E], int c0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const __amdgpu_buffer_rsrc_t r = dpi_buffer(a.x + (size_t)min(c0 + c, a.Cin - 1) * V, V * sizeof(float));
#pragma unroll
      for (int e = 0; e < E; ++e) sr[c][e] = dpi_buffer_load(r, goff[e] * 4);      // outside the volume -> 0 (zero padding)
    }
  };
  auto stage_store = [&](const float (&sr)[4][E], int c0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const Chain t = load_chain(a.chain, min(c0 + c, a.Cin - 1));
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float v = goff[e] >= 0 ? apply_chain(t, sr[c][e]) : sr[c][e];
        if ((e + 1) * 256 <= TILE || loff[e] >= 0) lds[c * CS + loff[e]] = v;
      }
    }
  };
  // weights of a chunk: lane (row i = lj -> co = lj>>2, kw = lj&3; ci = c0 + lk) keeps its 9 (kd, kh) taps
  const int wco = lj >> 2, wkw = lj & 3;
  auto load_w = [&](float (&wq)[9], int c0) {
    const int ci = c0 + lk;
    const bool ok = wkw < 3 && wco < a.Cout && ci < a.Cin;
    const float* __restrict__ wp = a.w + ((size_t)(ok ? wco : 0) * a.Cin + (ok ? ci : 0)) * 27 + (ok ? wkw : 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) wq[t] = wp[3 * t];
#pragma unroll
    for (int t = 0; t < 9; ++t) wq[t] = ok ? wq[t] : 0.f;
  };

  f32x4 acc[NR][NHQ];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int h = 0; h < NHQ; ++h) acc[r][h] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float sr[4][E], wq[9], wn[9];
  stage_load(sr, 0);
  load_w(wn, 0);
  const int lbase = lk * CS + wid * DS + lj;
  for (int c0 = 0; c0 < a.Cin; c0 += 4) {
    __syncthreads();
    stage_store(sr, c0);
#pragma unroll
    for (int t = 0; t < 9; ++t) wq[t] = wn[t];
    __syncthreads();
    if (c0 + 4 < a.Cin) { stage_load(sr, c0 + 4); load_w(wn, c0 + 4); }
    float bc[NHQ], bn[NHQ];
#pragma unroll
    for (int h = 0; h < NHQ; ++h) bc[h] = lds[lbase + h * 16];
#pragma unroll
    for (int step = 0; step < 3 * IH; ++step) {
      const int kd = step / IH, ir = step % IH;
      if (step + 1 < 3 * IH) {
        const int kd1 = (step + 1) / IH, ir1 = (step + 1) % IH;
#pragma unroll
        for (int h = 0; h < NHQ; ++h) bn[h] = lds[lbase + kd1 * DS + ir1 * RS + h * 16];
      }
      __builtin_amdgcn_sched_barrier(0);              // keep the next step's LDS reads ahead of this step's MFMAs
#pragma unroll
      for (int h = 0; h < NHQ; ++h)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int hr = ir - kh;
          if (hr >= 0 && hr < NR) acc[hr][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[kd * 3 + kh], bc[h], acc[hr][h], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < NHQ; ++h) bc[h] = bn[h];
    }
  }

  // ---- epilogue: lane (column j, co = lk) holds P[(co, kw = r)][16 h + j]; Y[c] = P0[c-1] + P1[c] + P2[c+1], c = 1 .. 46 ----------
  const int co = lk;
  const bool cok = co < a.Cout;
  const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
  const int od = od0 + wid;
  float* __restrict__ yc = a.y + (size_t)(cok ? co : 0) * V + ((size_t)od * a.H + oh0) * a.W + ow0 - 1;
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int hr = 0; hr < NR; ++hr) {
    const int oh = oh0 + hr;
#pragma unroll
    for (int h = 0; h < NHQ; ++h) {
      const float left = dpp_prev(h > 0 ? acc[hr][h - 1][0] : 0.f, acc[hr][h][0]);
      const float right = dpp_next(h + 1 < NHQ ? acc[hr][h + 1][2] : 0.f, acc[hr][h][2]);
      const int c = 16 * h + lj;
      const float v = (left + acc[hr][h][1]) + right + bv;
      if (cok && c >= 1 && c <= TWO && od < a.D && oh < a.H && ow0 - 1 + c < a.W) {
        yc[(size_t)hr * a.W + c] = v;
        s += v;
        q += (double)v * v;
      }
    }
  }
  if (a.partials) {
    __shared__ double red[4][4][2];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (lj == 0) { red[wid][lk][0] = s; red[wid][lk][1] = q; }
    __syncthreads();
    if (tid < 8) {
      const int c = tid >> 1, which = tid & 1;
      const double rsum = red[0][c][which] + red[1][c][which] + red[2][c][which] + red[3][c][which];
      if (c < a.Cout) a.partials[((size_t)tile_id * a.Cout + c) * 2 + which] = rsum;
    }
  }
}

}  // namespace

bool dpi_conv_fewco_usable(const dpi_conv_desc* d) {
  return d->k == 3 && d->kd == 3 && d->stride == 1 && d->Cout <= 4 && d->Cin >= 8 && (size_t)d->D * d->H * d->W >= 32768 &&
         (size_t)d->D * d->H * d->W < ((size_t)1 << 29);
}

int dpi_conv_fewco_tiles(const dpi_conv_desc* d, int* ntd, int* nth, int* ntw) {
  *ntd = cdiv(d->D, TZ); *nth = cdiv(d->H, NR); *ntw = cdiv(d->W, TWO);
  return *ntd * *nth * *ntw;
}

int dpi_conv_fewco_mfma_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias, float* y,
                            double* partials, hipStream_t st) {
  FcArgs a{x, chain, w, bias, y, partials, d->Cin, d->Cout, d->D, d->H, d->W, 0, 0, 0};
  const int ntiles = dpi_conv_fewco_tiles(d, &a.ntd, &a.nth, &a.ntw);
  conv_fewco_mfma_kernel<<<ntiles, 256, 0, st>>>(a);
  return dpi_check_launch("conv_fewco_mfma");
}
