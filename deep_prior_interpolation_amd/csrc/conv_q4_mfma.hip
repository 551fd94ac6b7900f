// 3x3x3 stride-1 convolution with FEW output channels (<= 8) on v_mfma_f32_4x4x1_16B_f32 — forward of the 64->4 / 67->4 / 4->8 /
// 25->1 / 137->8 / 25->8 layers (reference mulresunet.py:70-78,244) and backward-data of the layers with <= 8 INPUT channels
// (4->8, 8->13, 8->17), whose gradient has that few channels.
//
// Why another kernel: v_mfma_f32_16x16x4_f32 (conv_mfma.hip) has 16 output-channel rows; with 4 real channels three quarters of
// every MFMA multiply padding, and the (co, kw)-row packing of conv_fewco_mfma.hip still reaches only 58-61 TFLOP/s.  The 4x4x1
// instruction is 16 independent 4x4 outer products (K = 1) at the same 64 FLOP/clk/SIMD (tools/ubench/mfma4x4: 8.5 clk per
// instruction, 144 TFLOP/s) — rows in units of FOUR, K in units of ONE (no channel padding either):
//
//     D_b[i][j] += A_b[i] * B_b[j]      b = 0..15 blocks, lane = 4 b + i (A) / 4 b + j (B, D); D: lane (b, j), register i
//        A_b[i] = W[co = 4 s + i][ci][tap]        (the same for every block: lane l holds the weight of row l & 3)
//        B_b[j] = X[ci][d + kd - 1][h + kh - 1][w0 + 4 b + j + kw - 1]   -> lane l <-> output column w0 + l: 64 consecutive voxels
//
// so one instruction multiplies one (ci, tap) into 64 voxels x 4 channels, a lane's 4 result registers are its voxel's 4 channels
// (stores are whole 256-byte rows), and the three kw taps of an input row are three conflict-free LDS reads one column apart — no
// VALU instruction at all between the MFMAs of a chunk.  Weights live in registers for a
// whole chunk of CK input channels (CK x 27 x NB values, NB = ceil(Cout / 4)); they reach the lanes through a small LDS table the
// workgroup fills one chunk ahead.
//
// Workgroup = 4 waves = 4 depth slices x R rows x 64 columns.  Halo tile in LDS: [CK][6][R + 2][72] floats — columns w0 - 4 ..
// w0 + 67, i.e. 18 ALIGNED float4 per row: staged with buffer_load_dwordx4 / ds_write_b128 (W % 4 == 0 makes every float4 lie
// wholly inside or wholly outside a row; outside -> offset -16 -> the hardware returns 0 = the zero padding).
// Numerics: fp32 multiply-add chains in (ci, kd, row, kw) order — a plain fp32 direct convolution like the other fp32 kernels.
#include "common.h"

void dpi_conv_out_dims(const dpi_conv_desc* d, int* Do, int* Ho, int* Wo);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct QArgs {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ w;
  const float* __restrict__ bias;
  float* __restrict__ y;
  double* __restrict__ partials;    // [ntiles][Cout][2] or NULL
  int Cin, Cout;
  int D, H, W;
  int ntd, nth, ntw;
  long w_out_stride, w_in_stride;
  int accumulate;
  int dbg;          // timing experiments (dpi_set_q4_debug): bit 0 no x loads after chunk 0, bit 1 no LDS stores after chunk 0, bit 2 no MFMAs, bit 3 no output stores
};

template <int R>
struct QGeo {
  static constexpr int TZ = 4, TY = R, TW = 64;
  static constexpr int ID = TZ + 2, IH = R + 2;
  static constexpr int RS = 72;                 // LDS row: columns w0 - 4 .. w0 + 67
  static constexpr int DS = IH * RS, CS = ID * DS;
  static constexpr int NV4 = ID * IH * 18;      // float4 slots of one channel
  static constexpr int E = (NV4 + 255) / 256;
};

__device__ __forceinline__ int q4_xcd_tile(int bid, int ntiles) {
  const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// NB: 4-row blocks of output channels (Cout <= 4 NB).  CK: input channels per chunk.  FLIP: backward-data (taps mirrored, weight
// strides swapped by the host).  AL: x rows are 16-byte aligned (dwordx4 staging), else four dword loads per slot.
template <int R, int NB, int CK, bool FLIP, bool AL>
__global__ __launch_bounds__(256, 2) void conv_q4_mfma_kernel(QArgs a) {
  using G = QGeo<R>;
  constexpr int TAPS = 27;
  constexpr int NWT = CK * TAPS * NB * 4;           // A-operand table of one chunk: [c][tap][s][i]
  constexpr int WE = (NWT + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[CK * G::CS];
  __shared__ float wl[NWT];
  __shared__ double red[4][4 * NB][2];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ntiles = a.ntd * a.nth * a.ntw;
  const int tile_id = q4_xcd_tile(blockIdx.x, ntiles);
  int bt = tile_id;
  const int tw_i = bt % a.ntw; bt /= a.ntw;
  const int th_i = bt % a.nth; bt /= a.nth;
  const int od0 = bt * G::TZ, oh0 = th_i * G::TY, ow0 = tw_i * G::TW;
  const size_t V = (size_t)a.D * a.H * a.W;

  // this thread's float4 slots of the halo tile: input voxels (od0 - 1 + dz, oh0 - 1 + hy, ow0 - 4 + 4 q .. + 3)
  int goff[G::E], loff[G::E];
#pragma unroll
  for (int e = 0; e < G::E; ++e) {
    const int idx = tid + e * 256;
    const int q = idx % 18, row = idx / 18;
    const int hy = row % G::IH, dz = row / G::IH;
    const int gd = od0 - 1 + dz, gh = oh0 - 1 + hy, gw = ow0 - 4 + 4 * q;
    const bool ok = idx < G::NV4 && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw + 4 <= a.W;
    goff[e] = ok ? (gd * a.H + gh) * a.W + gw : -4;
    loff[e] = idx < G::NV4 ? dz * G::DS + hy * G::RS + 4 * q : -1;
  }
  auto stage_load = [&](f32x4 (&sr)[CK][G::E], int c0) {
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const int ci = min(c0 + c, a.Cin - 1);          // channels past Cin: their weights are zero
      const __amdgpu_buffer_rsrc_t r = dpi_buffer(a.x + (size_t)ci * V, V * sizeof(float));
#pragma unroll
      for (int e = 0; e < G::E; ++e) {
        if constexpr (AL) sr[c][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, goff[e] * 4, 0, 0));
        else {
#pragma unroll
          for (int k = 0; k < 4; ++k) sr[c][e][k] = dpi_buffer_load(r, goff[e] < 0 ? -4 : (goff[e] + k) * 4);
        }
      }
    }
  };
  // Two code paths selected by ONE wave-uniform branch.  (Written as `chain && goff >= 0 ? T(x) : x` hipcc first produced an
  // exec-masked region per element, then — as `chain ? m * T(x) : x` — computed T(x) for every element even without a chain: the
  // store phase of a chunk took 6.5 k clk instead of a few hundred either way.)
  auto stage_store = [&](const f32x4 (&sr)[CK][G::E], int c0) {
    if (a.chain == nullptr) {
#pragma unroll
      for (int c = 0; c < CK; ++c)
#pragma unroll
        for (int e = 0; e < G::E; ++e)
          if ((e + 1) * 256 <= G::NV4 || loff[e] >= 0) *reinterpret_cast<f32x4*>(lds + c * G::CS + loff[e]) = sr[c][e];
      return;
    }
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const Chain t = load_chain(a.chain, min(c0 + c, a.Cin - 1));
#pragma unroll
      for (int e = 0; e < G::E; ++e) {
        const float m = goff[e] >= 0 ? 1.f : 0.f;     // zero padding stays zero through a factor, not a per-lane branch
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = m * apply_chain(t, sr[c][e][k]);
        if ((e + 1) * 256 <= G::NV4 || loff[e] >= 0) *reinterpret_cast<f32x4*>(lds + c * G::CS + loff[e]) = v;
      }
    }
  };
  // A-operand table entry t = ((c * 27 + tap) * NB + s) * 4 + i  <-  W[co = 4 s + i][ci = c0 + c][tap].  The load is RAW (clamped
  // address) and the zero for padded rows / channels is selected when the value is stored to LDS a chunk later: a select right
  // behind the load would put an s_waitcnt vmcnt(0) there, and vmcnt retires in order — it would also wait for the whole halo-tile
  // prefetch issued just before (measured: the first build spent 0.28 of 0.92 ms exactly there).
  auto wt_index = [&](int t, int c0, bool& ok) {
    const int i = t & 3, s = (t >> 2) % NB, ct = (t >> 2) / NB, tap = ct % TAPS, c = ct / TAPS;
    const int co = 4 * s + i, ci = c0 + c;
    ok = t < NWT && co < a.Cout && ci < a.Cin;
    return (ok ? co : 0) * a.w_out_stride + (ok ? ci : 0) * a.w_in_stride + (FLIP ? TAPS - 1 - tap : tap);
  };
  auto load_wt = [&](float (&wreg)[WE], int c0) {
#pragma unroll
    for (int k = 0; k < WE; ++k) {
      bool ok;
      wreg[k] = a.w[wt_index(tid + k * 256, c0, ok)];
    }
  };
  auto store_wt = [&](const float (&wreg)[WE], int c0) {
#pragma unroll
    for (int k = 0; k < WE; ++k) {
      bool ok;
      wt_index(tid + k * 256, c0, ok);
      if (tid + k * 256 < NWT) wl[tid + k * 256] = ok ? wreg[k] : 0.f;
    }
  };

  const int Do = a.D, Ho = a.H, Wo = a.W;
  const int od = od0 + wid, ow = ow0 + lane;
  f32x4 acc[R][NB];
#pragma unroll
  for (int hr = 0; hr < R; ++hr)
#pragma unroll
    for (int s = 0; s < NB; ++s) acc[hr][s] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 sr[CK][G::E];
  float wreg[WE];
  load_wt(wreg, 0);
  stage_load(sr, 0);

  const int xbase = wid * G::DS + 4 + lane;             // centre column of this wave's depth slice (kd = 0), input row 0
  constexpr int NSTEP = CK * 3 * G::IH;                 // (channel, kd, input row) steps of a chunk

  for (int c0 = 0; c0 < a.Cin; c0 += CK) {
    if (!(a.dbg & 32)) __syncthreads();                 // everyone is done with the previous chunk's tile and weight table
    if (!(a.dbg & 2) || c0 == 0) stage_store(sr, c0);
    store_wt(wreg, c0);
    if (!(a.dbg & 32)) __syncthreads();
    if (a.dbg & 4) {
      if (c0 + CK < a.Cin) { load_wt(wreg, c0 + CK); stage_load(sr, c0 + CK); }
      continue;
    }
    float wr[CK][TAPS][NB];
#pragma unroll
    for (int c = 0; c < CK; ++c)
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int s = 0; s < NB; ++s) wr[c][t][s] = wl[((c * TAPS + t) * NB + s) * 4 + (lane & 3)];

    // the three kw taps of an input row are three conflict-free LDS reads one column apart (lane l <-> column w0 + l + kw - 1).
    // Measured alternative (tools/ubench/mfma4x4): ONE read + two DPP lane shifts costs 15 % of the MFMA rate even with three waves
    // per SIMD — DPP moves are VALU instructions and compete with the MFMA issue, LDS reads have their own port.
    auto load_x = [&](float (&xv)[3], int step) {
      const int c = step / (3 * G::IH), kd = (step / G::IH) % 3, ir = step % G::IH;
      const int o = xbase + c * G::CS + kd * G::DS + ir * G::RS;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xv[kw] = lds[o + kw - 1];
    };
    float xv[3], xn[3];
    load_x(xv, 0);
    // next chunk's loads fly behind this chunk's MFMAs.  Issued AFTER the LDS reads above on purpose: placed before them, hipcc's
    // wait-count pass put s_waitcnt vmcnt(1) / vmcnt(0) between those reads (it believes registers they write are still the
    // target of older memory loads) and the whole prefetch completed before the first MFMA.
    __builtin_amdgcn_sched_barrier(0);
    if (c0 + CK < a.Cin) {
      load_wt(wreg, c0 + CK);
      if (!(a.dbg & 1)) stage_load(sr, c0 + CK);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int c = step / (3 * G::IH), kd = (step / G::IH) % 3, ir = step % G::IH;
      if (step + 1 < NSTEP) load_x(xn, step + 1);
      __builtin_amdgcn_sched_barrier(0);                // keep the next step's LDS reads ahead of this step's MFMAs
      // kw outermost, then the (up to three) output rows this input row feeds: consecutive MFMAs hit different accumulators
      // (a dependent 4x4x1 needs 15.5 clk, an independent one issues every 8.5: tools/ubench/mfma4x4)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int hr = ir - kh;
          if (hr >= 0 && hr < R) {
#pragma unroll
            for (int s = 0; s < NB; ++s)
              acc[hr][s] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[c][(kd * 3 + kh) * 3 + kw][s], xv[kw], acc[hr][s], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xv[kw] = xn[kw];
    }
  }

  // ---- epilogue: lane <-> column ow, register i of block s <-> channel 4 s + i --------------------------------------------
  const bool col_ok = od < Do && ow < Wo;
#pragma unroll
  for (int s = 0; s < NB; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = 4 * s + i;
      const bool cok = co < a.Cout;                     // wave-uniform
      const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
      float* __restrict__ yc = a.y + (size_t)(cok ? co : 0) * V + ((size_t)(od < Do ? od : 0) * Ho + oh0) * Wo + ow;
      double sm = 0.0, sq = 0.0;
      // gradient fan-in (accumulate): the destination is read HERE, not into the accumulators before the channel loop — loads
      // pending on accumulator registers made hipcc put an s_waitcnt vmcnt(0) in front of the loop's first MFMA, which on every
      // later chunk waited for the whole halo-tile prefetch issued just before it (measured: 8.7 k instead of 2.8 k clk).  All R
      // loads of a channel are issued together (one wave-uniform branch): read-add-store per element serialised them (2.4 x slower)
      float old[R];
      if (a.accumulate) {
#pragma unroll
        for (int hr = 0; hr < R; ++hr) old[hr] = (cok && col_ok && oh0 + hr < Ho) ? yc[hr * Wo] : 0.f;
      } else {
#pragma unroll
        for (int hr = 0; hr < R; ++hr) old[hr] = 0.f;
      }
#pragma unroll
      for (int hr = 0; hr < R; ++hr) {
        if (cok && col_ok && oh0 + hr < Ho) {
          const float v = acc[hr][s][i] + bv + old[hr];
          if (!(a.dbg & 8)) yc[hr * Wo] = v;
          sm += v;
          sq += (double)v * v;
        }
      }
      if (a.partials && cok) {
        sm = wave_sum(sm);
        sq = wave_sum(sq);
        if (lane == 0) { red[wid][co][0] = sm; red[wid][co][1] = sq; }
      }
    }
  if (a.partials) {
    __syncthreads();
    if (tid < 2 * a.Cout) {
      const int c = tid >> 1, which = tid & 1;
      a.partials[((size_t)tile_id * a.Cout + c) * 2 + which] = red[0][c][which] + red[1][c][which] + red[2][c][which] + red[3][c][which];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Variant with the 4 input channels of a chunk INTERLEAVED in LDS ([depth][row][column][4 channels]): one ds_read_b128 hands a lane
// the four channels of its voxel, so an input row costs 3 LDS reads (the kw taps) for 36 NB MFMAs instead of 3 reads for 9 NB.
// Why it matters (MI355X_MICROARCH.md §LDS, and measured here): a ds_read_b32 reaches its 2-cycle rate only with ~4 waves per SIMD
// issuing long runs of them; with the 2-3 waves this kernel has, every b32 read cost ~10 clk of the wave's issue time and the
// compute loop alone ran at 60 % of the MFMA rate.  ds_read_b128 reaches its rate from one wave per SIMD.
//   staging: a thread loads the same float4 (4 columns) of the 4 channels, transposes the 4x4 block in registers (free) and stores
//   four 16-byte {c0,c1,c2,c3} positions.  Position (row, col) lives at slot 4 g + (j ^ ((g >> 1) & 3)), g = col >> 2, j = col & 3:
//   the XOR spreads the 8 lanes of a ds_write_b128 lane group (64 bytes apart otherwise: two banks quads) over all banks.
//   weights: LDS table [tap][s][i][ci] so that one b128 read gives a lane its row's weights for the 4 channels; held per kd plane.
template <int R>
struct QiGeo {
  static constexpr int TZ = 4, TY = R, TW = 64;
  static constexpr int ID = TZ + 2, IH = R + 2;
  static constexpr int NP = 72;                 // positions per row: columns w0 - 4 .. w0 + 67
  static constexpr int RS = NP * 4, DS = IH * RS, TILE = ID * DS;     // floats
  static constexpr int NSLOT = ID * IH * 18;    // (row, 4-column group) slots
  static constexpr int E = (NSLOT + 255) / 256;
};
__device__ __forceinline__ int qi_pos(int col) {                     // float offset of logical column `col` inside its LDS row
  const int g = col >> 2, j = col & 3;
  return (4 * g + (j ^ ((g >> 1) & 3))) * 4;
}

// KHP (ONE output channel, forward: the 25 -> 1 output layer): the 4 MFMA rows hold the three kh taps of that channel instead of four
// channels (three of which do not exist).  An input row then feeds its three output rows with ONE MFMA per (kw, channel) — accumulator
// of INPUT row q, component kh = the partial sum of output row q - kh — and the epilogue adds the three shifted components:
// 360 MFMAs per chunk and wave instead of 864, rows 75 % instead of 25 % full.
template <int R, int NB, bool FLIP, bool AL, bool KHP = false>
__global__ __launch_bounds__(256, 2) void conv_q4i_mfma_kernel(QArgs a) {
  static_assert(!KHP || (NB == 1 && !FLIP), "kh-packed rows: one output channel, forward only");
  using G = QiGeo<R>;
  constexpr int TAPS = 27;
  constexpr int NWT = KHP ? 9 * 16 : TAPS * NB * 16;   // weight table of one chunk: [tap][s][i][ci]  (KHP: [kd][kw][i = kh][ci])
  constexpr int WE = (NWT + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[G::TILE];
  __shared__ __attribute__((aligned(16))) float wl[NWT];
  __shared__ double red[4][4 * NB][2];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ntiles = a.ntd * a.nth * a.ntw;
  const int tile_id = q4_xcd_tile(blockIdx.x, ntiles);
  int bt = tile_id;
  const int tw_i = bt % a.ntw; bt /= a.ntw;
  const int th_i = bt % a.nth; bt /= a.nth;
  const int od0 = bt * G::TZ, oh0 = th_i * G::TY, ow0 = tw_i * G::TW;
  const size_t V = (size_t)a.D * a.H * a.W;

  int goff[G::E], loff[G::E];
#pragma unroll
  for (int e = 0; e < G::E; ++e) {
    const int idx = tid + e * 256;
    const int q = idx % 18, row = idx / 18;
    const int hy = row % G::IH, dz = row / G::IH;
    const int gd = od0 - 1 + dz, gh = oh0 - 1 + hy, gw = ow0 - 4 + 4 * q;
    const bool ok = idx < G::NSLOT && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw + 4 <= a.W;
    goff[e] = ok ? (gd * a.H + gh) * a.W + gw : -4;
    loff[e] = idx < G::NSLOT ? dz * G::DS + hy * G::RS + 16 * q : -1;      // float offset of the slot's 4-position group
  }
  const int swz = ((tid % 18) >> 1) & 3;           // XOR of this thread's column groups: q = (tid + 256 e) % 18 — 256 % 18 = 4, see below
  auto stage_load = [&](f32x4 (&sr)[G::E][4], int c0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ci = min(c0 + c, a.Cin - 1);          // channels past Cin: their weights are zero
      const __amdgpu_buffer_rsrc_t r = dpi_buffer(a.x + (size_t)ci * V, V * sizeof(float));
#pragma unroll
      for (int e = 0; e < G::E; ++e) {
        if constexpr (AL) sr[e][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, goff[e] * 4, 0, 0));
        else {
#pragma unroll
          for (int k = 0; k < 4; ++k) sr[e][c][k] = dpi_buffer_load(r, goff[e] < 0 ? -4 : (goff[e] + k) * 4);
        }
      }
    }
  };
  auto stage_store = [&](const f32x4 (&sr)[G::E][4], int c0) {
    if (a.chain == nullptr) {                         // ONE wave-uniform branch, see the other variant
#pragma unroll
      for (int e = 0; e < G::E; ++e) {
        const int x = (((tid + e * 256) % 18) >> 1) & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                 // column 4 q + j of the slot: {c0, c1, c2, c3}
          const f32x4 v = (f32x4){sr[e][0][j], sr[e][1][j], sr[e][2][j], sr[e][3][j]};
          if ((e + 1) * 256 <= G::NSLOT || loff[e] >= 0) *reinterpret_cast<f32x4*>(lds + loff[e] + 4 * (j ^ x)) = v;
        }
      }
      return;
    }
    Chain t[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) t[c] = load_chain(a.chain, min(c0 + c, a.Cin - 1));
#pragma unroll
    for (int e = 0; e < G::E; ++e) {
      const int x = (((tid + e * 256) % 18) >> 1) & 3;
      const float m = goff[e] >= 0 ? 1.f : 0.f;       // zero padding stays zero through a factor, not a per-lane branch
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = m * apply_chain(t[c], sr[e][c][j]);
        if ((e + 1) * 256 <= G::NSLOT || loff[e] >= 0) *reinterpret_cast<f32x4*>(lds + loff[e] + 4 * (j ^ x)) = v;
      }
    }
  };
  (void)swz;
  auto wt_index = [&](int t, int c0, bool& ok) {      // t = ((tap * NB + s) * 4 + i) * 4 + c
    if constexpr (KHP) {                              // t = ((kd * 3 + kw) * 4 + kh) * 4 + c, output channel 0
      const int c = t & 3, kh = (t >> 2) & 3, kw = (t >> 4) % 3, kd = min((t >> 4) / 3, 2), ci = c0 + c;
      ok = t < NWT && kh < 3 && ci < a.Cin;
      return (ok ? ci : 0) * a.w_in_stride + kd * 9 + (ok ? kh : 0) * 3 + kw;
    }
    const int c = t & 3, i = (t >> 2) & 3, s = (t >> 4) % NB, tap = (t >> 4) / NB;
    const int co = 4 * s + i, ci = c0 + c;
    ok = t < NWT && co < a.Cout && ci < a.Cin;
    return (ok ? co : 0) * a.w_out_stride + (ok ? ci : 0) * a.w_in_stride + (FLIP ? TAPS - 1 - min(tap, TAPS - 1) : min(tap, TAPS - 1));
  };
  auto load_wt = [&](float (&wreg)[WE], int c0) {
#pragma unroll
    for (int k = 0; k < WE; ++k) {
      bool ok;
      wreg[k] = a.w[wt_index(tid + k * 256, c0, ok)];
    }
  };
  auto store_wt = [&](const float (&wreg)[WE], int c0) {
#pragma unroll
    for (int k = 0; k < WE; ++k) {
      bool ok;
      wt_index(tid + k * 256, c0, ok);
      if (tid + k * 256 < NWT) wl[tid + k * 256] = ok ? wreg[k] : 0.f;
    }
  };

  const int Do = a.D, Ho = a.H, Wo = a.W;
  const int od = od0 + wid, ow = ow0 + lane;
  f32x4 acc[R][NB];
#pragma unroll
  for (int hr = 0; hr < R; ++hr)
#pragma unroll
    for (int s = 0; s < NB; ++s) acc[hr][s] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // KHP: per INPUT row, component kh; two partial accumulators per row (a dependent 4x4x1 MFMA needs ~15 clk, an independent one 8.5)
  f32x4 accp[KHP ? G::IH : 1][2];
#pragma unroll
  for (int q = 0; q < (KHP ? G::IH : 1); ++q) { accp[q][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; accp[q][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  f32x4 sr[G::E][4];
  float wreg[WE];
  load_wt(wreg, 0);
  stage_load(sr, 0);

  int xoff[3];                                        // this lane's three kw positions (logical columns 3 + lane + kw) in a row
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) xoff[kw] = wid * G::DS + qi_pos(3 + lane + kw);
  const f32x4* __restrict__ wl4 = reinterpret_cast<const f32x4*>(wl) + (lane & 3);

  for (int c0 = 0; c0 < a.Cin; c0 += 4) {
    __syncthreads();
    if (!(a.dbg & 2) || c0 == 0) stage_store(sr, c0);
    store_wt(wreg, c0);
    __syncthreads();
    auto load_x = [&](f32x4 (&xv)[3], int kd, int ir) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xv[kw] = *reinterpret_cast<const f32x4*>(lds + xoff[kw] + kd * G::DS + ir * G::RS);
    };
    auto load_w = [&](f32x4 (&wr)[9][NB], int kd) {
      if constexpr (KHP) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) wr[kw][0] = wl4[(kd * 3 + kw) * 4];     // this lane's row kh = lane & 3, four channels
      } else {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int s = 0; s < NB; ++s) wr[t][s] = wl4[((kd * 9 + t) * NB + s) * 4];
      }
    };
    f32x4 wr[9][NB], xv[3], xn[3];
    load_w(wr, 0);
    load_x(xv, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (c0 + 4 < a.Cin) {                               // next chunk's loads fly behind this chunk's MFMAs (after the LDS reads: see above)
      load_wt(wreg, c0 + 4);
      if (!(a.dbg & 1)) stage_load(sr, c0 + 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
      for (int ir = 0; ir < G::IH; ++ir) {
        if (ir + 1 < G::IH) load_x(xn, kd, ir + 1);
        else if (kd + 1 < 3) load_x(xn, kd + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KHP) {
          // 12 dependent MFMAs on one accumulator would wait ~15 clk each: two partial accumulators per row, merged below
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              accp[ir][c & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[kw][0][c], xv[kw][c], accp[ir][c & 1], 0, 0, 0);
        } else {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              const int hr = ir - kh;
              if (hr >= 0 && hr < R) {
#pragma unroll
                for (int s = 0; s < NB; ++s)
                  acc[hr][s] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[kh * 3 + kw][s][c], xv[kw][c], acc[hr][s], 0, 0, 0);
              }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) xv[kw] = xn[kw];
      }
      if (kd + 1 < 3) load_w(wr, kd + 1);
    }
  }

  if constexpr (KHP) {                                // output row hr = input rows hr, hr + 1, hr + 2 through taps kh = 0, 1, 2
#pragma unroll
    for (int hr = 0; hr < R; ++hr)
      acc[hr][0][0] = ((accp[hr][0][0] + accp[hr][1][0]) + (accp[hr + 1][0][1] + accp[hr + 1][1][1])) + (accp[hr + 2][0][2] + accp[hr + 2][1][2]);
  }
  const bool col_ok = od < Do && ow < Wo;
#pragma unroll
  for (int s = 0; s < NB; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = 4 * s + i;
      const bool cok = co < a.Cout;
      const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
      float* __restrict__ yc = a.y + (size_t)(cok ? co : 0) * V + ((size_t)(od < Do ? od : 0) * Ho + oh0) * Wo + ow;
      double sm = 0.0, sq = 0.0;
      // gradient fan-in (accumulate): the destination is read HERE, not into the accumulators before the channel loop — loads
      // pending on accumulator registers made hipcc put an s_waitcnt vmcnt(0) in front of the loop's first MFMA, which on every
      // later chunk waited for the whole halo-tile prefetch issued just before it (measured: 8.7 k instead of 2.8 k clk).  All R
      // loads of a channel are issued together (one wave-uniform branch): read-add-store per element serialised them (2.4 x slower)
      float old[R];
      if (a.accumulate) {
#pragma unroll
        for (int hr = 0; hr < R; ++hr) old[hr] = (cok && col_ok && oh0 + hr < Ho) ? yc[hr * Wo] : 0.f;
      } else {
#pragma unroll
        for (int hr = 0; hr < R; ++hr) old[hr] = 0.f;
      }
#pragma unroll
      for (int hr = 0; hr < R; ++hr) {
        if (cok && col_ok && oh0 + hr < Ho) {
          const float v = acc[hr][s][i] + bv + old[hr];
          if (!(a.dbg & 8)) yc[hr * Wo] = v;
          sm += v;
          sq += (double)v * v;
        }
      }
      if (a.partials && cok) {
        sm = wave_sum(sm);
        sq = wave_sum(sq);
        if (lane == 0) { red[wid][co][0] = sm; red[wid][co][1] = sq; }
      }
    }
  if (a.partials) {
    __syncthreads();
    if (tid < 2 * a.Cout) {
      const int c = tid >> 1, which = tid & 1;
      a.partials[((size_t)tile_id * a.Cout + c) * 2 + which] = red[0][c][which] + red[1][c][which] + red[2][c][which] + red[3][c][which];
    }
  }
}

// (Built, measured and removed in round 3: a "wide" variant for <= 4 INPUT channels and many outputs — backward-data of 67 -> 4 and
//  25 -> 1 — that stages the 4-channel tile once per workgroup, keeps all weights in LDS and walks the output channels in passes of 8.
//  31 parity cases green; 67 -> 4 backward-data 1.10 ms against 0.80 ms on the 16x16x4 kernel (25 -> 1: 0.44 vs 0.30): one workgroup
//  per CU and a per-pass prologue / epilogue every 1728 MFMAs cost more than the five-fold tile staging it saved.)
static int g_q4 = 1;          // dpi_set_q4: 0 off, 1 where it pays (below), 2 every shape it can run (tests)
static int g_q4_ck = 0;        // 0: by shape (q4_launch), 2 / 4: force the planar / the channel-interleaved variant
static int g_q4_dbg = 0;
static int g_q4_khp = getenv("DPI_NO_KHP") ? 0 : 1;   // one output channel: kh taps in the MFMA rows (conv_q4i_mfma_kernel<..., KHP>)

}  // namespace

extern "C" void dpi_set_q4_debug(int flags) { g_q4_dbg = flags; }
extern "C" void dpi_set_q4(int on, int ck) {
  if (on >= 0) g_q4 = on;
  if (ck == 0 || ck == 2 || ck == 4) g_q4_ck = ck;
}

// Where it applies: 3-D 3x3x3 stride 1 with <= 8 output channels (flip: the convolution's INPUT channels are the outputs of the
// backward-data pass), rows a whole number of float4 and at least 48 columns (a 64-lane row tile would otherwise idle), and
// enough tiles to give every CU one.
bool dpi_conv_q4_usable(const dpi_conv_desc* d, bool flip) {
  if (!g_q4 || d->k != 3 || d->kd != 3 || d->stride != 1) return false;
  if (dpi_io_in(d, flip) || dpi_io_out(d, flip)) return false;      // fp32 tensors only: bf16 storage takes the bf16 / 16x16x4 kernels (ABI 400)
  const int cout = flip ? d->Cin : d->Cout;
  if (cout > 8 || (d->W & 3)) return false;
  if (g_q4 == 2) return true;
  if (d->W < 48) return false;
  const long tiles = (long)cdiv(d->D, 4) * cdiv(d->H, 8) * cdiv(d->W, 64);
  return tiles >= 192;
}

int dpi_conv_q4_tiles(const dpi_conv_desc* d, int* ntd, int* nth, int* ntw) {
  *ntd = cdiv(d->D, 4); *nth = cdiv(d->H, 8); *ntw = cdiv(d->W, 64);
  return *ntd * *nth * *ntw;
}

template <int NB, bool FLIP>
static void q4_launch(const QArgs& a, int ntiles, bool aligned, hipStream_t st) {
  const size_t extra = (size_t)((g_q4_dbg >> 8) & 255) * 1024;
  if constexpr (NB == 1) {          // (two row blocks: the interleaved variant spills — 72 weight + 64 accumulator + 80 staging registers)
    // channel-interleaved LDS tile, ds_read_b128 operands: 5 % faster on the long channel loops (64 -> 4: 0.835 -> 0.787 ms, 67 -> 4:
    // 0.781 -> 0.753), equal on the short ones
    if constexpr (!FLIP) {
      if (a.Cout == 1 && g_q4_khp && (g_q4_ck == 4 || g_q4_ck == 0)) {
        if (aligned) conv_q4i_mfma_kernel<8, 1, false, true, true><<<ntiles, 256, extra, st>>>(a);
        else conv_q4i_mfma_kernel<8, 1, false, false, true><<<ntiles, 256, extra, st>>>(a);
        return;
      }
    }
    if (g_q4_ck == 4 || (g_q4_ck == 0 && a.Cin >= 16)) {
      if (aligned) conv_q4i_mfma_kernel<8, NB, FLIP, true><<<ntiles, 256, extra, st>>>(a);
      else conv_q4i_mfma_kernel<8, NB, FLIP, false><<<ntiles, 256, extra, st>>>(a);
      return;
    }
  }
  if (aligned) conv_q4_mfma_kernel<8, NB, 2, FLIP, true><<<ntiles, 256, extra, st>>>(a);
  else conv_q4_mfma_kernel<8, NB, 2, FLIP, false><<<ntiles, 256, extra, st>>>(a);
}

int dpi_conv_q4_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias, float* y,
                    double* partials, bool flip, int accumulate, hipStream_t st) {
  const int cin = flip ? d->Cout : d->Cin, cout = flip ? d->Cin : d->Cout;
  const long w_out = flip ? 27 : (long)d->Cin * 27, w_in = flip ? (long)d->Cin * 27 : 27;
  QArgs a{x, chain, w, bias, y, partials, cin, cout, d->D, d->H, d->W, 0, 0, 0, w_out, w_in, accumulate, g_q4_dbg};
  const int ntiles = dpi_conv_q4_tiles(d, &a.ntd, &a.nth, &a.ntw);
  const bool aligned = ((uintptr_t)x & 15) == 0;
  if (cout <= 4) { if (flip) q4_launch<1, true>(a, ntiles, aligned, st); else q4_launch<1, false>(a, ntiles, aligned, st); }
  else { if (flip) q4_launch<2, true>(a, ntiles, aligned, st); else q4_launch<2, false>(a, ntiles, aligned, st); }
  return dpi_check_launch("conv_q4_mfma");
}
