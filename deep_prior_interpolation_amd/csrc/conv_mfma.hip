// fp32-MFMA stencil convolution (k = 3, stride 1) for gfx950: forward, backward-data (FLIP) and backward-weight.
//
// The arithmetic engine is v_mfma_f32_16x16x4_f32 — f32 in / f32 accumulate, bit-for-bit a k-ordered fmaf chain, at
// the fp32 vector rate — so numerics are those of a plain fp32 direct convolution; what MFMA buys is that one wave
// per SIMD saturates the FMA pipe with ONE LDS read per 2048 FLOP and no scalar-load / SGPR-pair traffic (the limiter
// of the VALU kernel in conv_direct.hip).  This is still a direct (stencil) convolution: the halo
// tile of 4 input channels is staged in LDS exactly as in the VALU kernel (chain applied on load, zero padding
// materialised) and each MFMA consumes one tap of 4 channels for 16 consecutive voxels x 16 output channels:
//
//     D[co 16][vox 16] += A[co 16][ci 4] * B[ci 4][vox 16]          (fixed tap)
//        A = weights (registers, loaded once per 4-channel chunk), lane l: co = l&15, ci = l>>4
//        B = halo tile (LDS), lane l: ci = l>>4, voxel = l&15 (+ tap offset) -> conflict-free ds_read_b32
//          (channel stride = 16 mod 32 banks)
//
// Workgroup = 4 waves = output tile 4x8x32 (3-D; wave w owns depth slice w) or 1x32x32 (2-D; wave w owns 8 rows);
// each wave keeps 16 voxel tiles x MT channel tiles of accumulators.  Every LDS value is read once per chunk and
// feeds up to 3 (kh) x MT MFMAs.
#include "common.h"

void dpi_conv_out_dims(const dpi_conv_desc* d, int* Do, int* Ho, int* Wo);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MArgs {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ w;
  const float* __restrict__ bias;
  float* __restrict__ y;
  double* __restrict__ partials;
  int Cin, Cout;
  int D, H, W;
  int ntd, nth, ntw;
  long w_out_stride, w_in_stride;
  int accumulate;
  // input-channel split ("split-K", coarse levels): split_cps > 0 -> workgroup column blockIdx.z sums input channels
  // [z * split_cps, (z + 1) * split_cps) only and writes its partial result to y + z * Cout * Vo (y = the workspace; no bias,
  // no statistics, no fan-in: splitk_reduce_kernel adds those).  0: blockIdx.z = 0 covers every channel.
  int split_cps;
  // second input through a 1x1(x1) kernel at the OUTPUT positions (x2 != nullptr): y += W2 * x2, x2 [C2][Do][Ho][Wo],
  // W2[co][c] = w2[co * w2_co_stride + c * w2_c_stride].  Backward-data of a 3x3x3 layer and a 1x1x1 layer that read the same
  // tensor (Block3d.conv1 + shortcut, ResPath3d.conv3x3 + conv1x1) as ONE pass over dx instead of write + read-modify-write.
  const float* __restrict__ x2;
  const float* __restrict__ w2;
  int C2;
  long w2_co_stride, w2_c_stride;
  // storage type of the input tensor(s) x / x2 and of the output y in HBM: 1 = bf16, 0 = fp32 (dpi_conv_desc.io; common.h).
  // A split launch (split_cps > 0) writes fp32 partials into the workspace whatever y is; splitk_reduce_kernel stores y.
  int xb, yb;
  // > 0: 1-D grid of 8 * ceil(ntiles / 8) * ny workgroups; XCD x (= workgroup id % 8) walks ITS contiguous tile range with the ny
  // output-channel tiles of a spatial tile back to back — they read the same input tile (and second input), which as the slow grid
  // dimension they fetched from HBM once per channel tile (round 4, as conv_bf16_mfma.hip).  0: the 2-D grid (persistent variants).
  int ny;
};

// Tile geometry.  A wave owns NR output rows x NH 16-voxel column blocks; the 4 waves of a workgroup own
//   3-D, NR = 8: 4 depth slices  (tile 4 x 8 x 16*NH)          3-D, NR = 2: 4 row pairs of ONE depth slice (tile 1 x 8 x 16*NH)
//   2-D        : 4 bands of NR rows (tile 1 x 4*NR x 16*NH)
// The small variants (NR = 2, NH = 1) exist so that the coarse levels of the U-Net (a few thousand voxels, hundreds of
// channels) still produce >= 1000 waves.
template <int KD, int NR, int NH, int S = 1>
struct Geo {
  static constexpr bool SLICES = (KD == 3 && NR >= 4);          // waves split depth; otherwise they split rows
  static constexpr int SD = KD == 3 ? S : 1;
  static constexpr int TZ = SLICES ? 4 : 1;                      // output tile
  static constexpr int TY = SLICES ? NR : 4 * NR;
  static constexpr int TW = 16 * NH;
  static constexpr int ID = (TZ - 1) * SD + KD;                  // input (halo) tile
  static constexpr int IH = (TY - 1) * S + 3;
  static constexpr int IW = (TW - 1) * S + 3;
  static constexpr int RS = (IW + 3) & ~3;                        // row stride (floats)
  // depth-slice stride: for the stride-1 tiles (RS = 4 mod 32) pad to 12 (mod 32) so that the 9 (kd,kh) row offsets
  // kd*12 + kh*4 of one backward-weight tap tile fall on disjoint 4-bank slots (kw + voxel spread = 4 banks)
  static constexpr int DS0 = IH * RS;
  static constexpr int DS = (S == 1 && (RS % 32) == 4) ? DS0 + ((12 - (DS0 % 32)) + 32) % 32 : DS0;
  static constexpr int CS0 = ID * DS;
  // lanes of one ds_read_b32 group are 16 voxels (lane stride S) x 2 channels: channel stride = 16 (mod 32 banks) for
  // unit stride, odd (= 1 mod 32) for stride 2 (even banks for one channel, odd banks for the other)
  static constexpr int CSM = S == 1 ? 16 : 1;
  static constexpr int CS = CS0 + ((CSM - (CS0 % 32)) + 32) % 32;
  static constexpr int TILE = ID * IH * IW;
  static constexpr int E = (TILE + 255) / 256;
  static constexpr int NROW = (NR - 1) * S + 3;                   // input rows a wave's band touches
  static constexpr int NSTEP = KD * NROW;                         // (kd, input row) pairs a wave walks per chunk
  static constexpr int NB = 3 * NH;                               // LDS values per step
};

// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order; used for L2 locality only, never for
// correctness), so give every XCD a CONTIGUOUS range of tiles: halo rows shared by neighbouring tiles then hit the same L2.
__device__ __forceinline__ int xcd_tile(int bid, int ntiles) {
  const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

template <class G>
__device__ __forceinline__ void tile_slots(int tid, int id0, int ih0, int iw0, int D, int H, int W, int (&goff)[G::E], int (&loff)[G::E]) {
#pragma unroll
  for (int e = 0; e < G::E; ++e) {
    const int idx = tid + e * 256;
    const int col = idx % G::IW, row = idx / G::IW;
    const int hy = row % G::IH, dz = row / G::IH;
    const int gd = id0 + dz, gh = ih0 + hy, gw = iw0 + col;
    const bool ok = idx < G::TILE && gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W;
    goff[e] = ok ? (gd * H + gh) * W + gw : -1;
    loff[e] = idx < G::TILE ? dz * G::DS + hy * G::RS + col : -1;
  }
}

// issue the global loads of 4 channels [c0, c0+4) of the halo tile into registers (no wait here).  Buffer loads: slots
// outside the volume carry goff = -1 -> byte offset -4 -> out of range -> 0 from the hardware (the zero padding).
template <class G>
__device__ __forceinline__ void stage_load(float (&sr)[4][G::E], const float* __restrict__ x, int Cin, size_t V, int c0,
                                           const int (&goff)[G::E], bool xb = false) {
  if (xb) {      // bf16 tensor (wave-uniform): 2-byte loads, widened exactly; the same slots, half the bytes
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ci = min(c0 + c, Cin - 1);
      const __amdgpu_buffer_rsrc_t r = dpi_buffer_t(dpi_at(x, (size_t)ci * V, true), V, true);
#pragma unroll
      for (int e = 0; e < G::E; ++e) sr[c][e] = dpi_buffer_load_bf16_raw(r, goff[e]);      // raw bits: stage_store widens (common.h)
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    // channels past Cin re-read the last real channel: their WEIGHTS are zero (load_w), so no uniform branch is needed
    const int ci = min(c0 + c, Cin - 1);
    const __amdgpu_buffer_rsrc_t r = dpi_buffer(x + (size_t)ci * V, V * sizeof(float));
#pragma unroll
    for (int e = 0; e < G::E; ++e) sr[c][e] = dpi_buffer_load(r, goff[e] * 4);
  }
}

// registers -> LDS, applying the per-channel chain to in-volume samples (zero padding stays zero)
template <class G>
__device__ __forceinline__ void stage_store(float* lds, const float (&sr_in)[4][G::E], const float* __restrict__ chain, int Cin, int c0,
                                            const int (&goff)[G::E], const int (&loff)[G::E], bool xb = false) {
  float sr[4][G::E];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < G::E; ++e) sr[c][e] = sr_in[c][e];
  if (xb) {         // the prefetch kept the raw 16 bits of a bf16 tensor (stage_load): widen them here, where the data is consumed anyway
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < G::E; ++e) sr[c][e] = dpi_widen_raw(sr[c][e]);
  }
  // ONE wave-uniform branch: without a chain (every backward-data launch, every layer that reads a materialised tensor) the staging is
  // plain stores.  Written as a per-element `chain && in_volume ? T(x) : x`, hipcc evaluated T(x) for every element and selected
  // (5 VALU instructions x 32 elements per tile that compete with the MFMAs of the other waves for the SIMD; round 3).
  if (chain == nullptr) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < G::E; ++e)
        if ((e + 1) * 256 <= G::TILE || loff[e] >= 0) lds[c * G::CS + loff[e]] = sr[c][e];   // only the last slot can be past the tile
    return;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int ci = min(c0 + c, Cin - 1);
    const Chain t = load_chain(chain, ci);
#pragma unroll
    for (int e = 0; e < G::E; ++e) {
      const float v = goff[e] >= 0 ? apply_chain(t, sr[c][e]) : sr[c][e];
      if ((e + 1) * 256 <= G::TILE || loff[e] >= 0) lds[c * G::CS + loff[e]] = v;
    }
  }
}

// ---- stride-2 forward: vectorised staging into even / odd column planes (round 5) ----------------------------------------------------
// A stride-2 tile reads 13 input elements per output (3 x 17 x 65 for 1 x 8 x 32 outputs) where a stride-1 tile reads 2: with one dword load and
// one LDS store per element the kernel is bound by the NUMBER of L1 requests (52 loads + 52 stores per thread and 4-channel chunk for 108 MFMAs
// per wave), not by the matrix pipe (25->25 at 256x128x128: 0.37 ms for 0.11 ms of MFMAs).  Here a thread loads aligned float4 pieces of an input
// row (columns 2 ow0 - 4 ...: W % 4 == 0 puts every piece wholly inside or outside the row, outside -> offset -16 -> zeros) and stores the even
// columns to one plane and the odd columns to another (two 8-byte LDS stores per piece), so that the 16 lanes of a B operand — input columns
// 2 ow + kw - 1 for consecutive ow — read 16 CONSECUTIVE positions of one plane (the channel stride is then 16 mod 32 as in the stride-1 tiles:
// conflict-free) instead of every second float.  The design of conv_bf16_s2_kernel (round 4) on the fp32 MFMA.
template <int KD, int NR, int NH>
struct GeoV2 {
  static constexpr int TY = 4 * NR, TW = 16 * NH;                 // output tile 1 x TY x TW (the waves split the rows)
  static constexpr int ID = KD == 3 ? 3 : 1, IH = (TY - 1) * 2 + 3;
  static constexpr int NQ = TW / 2 + 1;                           // float4 pieces per input row: columns 2 ow0 - 4 .. 2 ow0 + 2 TW - 1
  static constexpr int PW = 2 * NQ;                               // positions per parity plane and row
  static constexpr int RS = 2 * PW;                               // row = [even plane][odd plane]
  static constexpr int DS = IH * RS;
  static constexpr int CS0 = ID * DS;
  static constexpr int CS = CS0 + ((16 - (CS0 % 32)) + 32) % 32;
  static constexpr int NV4 = ID * IH * NQ;
  static constexpr int E = (NV4 + 255) / 256;
};
template <class GV>
__device__ __forceinline__ void tile_slots_v(int tid, int id0, int ih0, int iw0, int D, int H, int W, int (&goff)[GV::E], int (&loff)[GV::E]) {
#pragma unroll
  for (int e = 0; e < GV::E; ++e) {
    const int idx = tid + e * 256;
    const int q = idx % GV::NQ, row = idx / GV::NQ;
    const int hy = row % GV::IH, dz = row / GV::IH;
    const int gd = id0 + dz, gh = ih0 + hy, gw = iw0 + 4 * q;
    const bool ok = idx < GV::NV4 && gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw + 4 <= W;
    goff[e] = ok ? (gd * H + gh) * W + gw : -4;
    loff[e] = idx < GV::NV4 ? dz * GV::DS + hy * GV::RS + 2 * q : -1;
  }
}
template <class GV>
__device__ __forceinline__ void stage_load_v(f32x4 (&sv)[4][GV::E], const float* __restrict__ x, int Cin, size_t V, int c0, const int (&goff)[GV::E]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int ci = min(c0 + c, Cin - 1);            // channels past Cin: their weights are zero
    const __amdgpu_buffer_rsrc_t r = dpi_buffer(x + (size_t)ci * V, V * sizeof(float));
#pragma unroll
    for (int e = 0; e < GV::E; ++e) sv[c][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, goff[e] * 4, 0, 0));
  }
}
template <class GV>
__device__ __forceinline__ void stage_store_v(float* lds, const f32x4 (&sv)[4][GV::E], const float* __restrict__ chain, int Cin, int c0,
                                              const int (&goff)[GV::E], const int (&loff)[GV::E]) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  if (chain == nullptr) {                     // one wave-uniform branch, as stage_store
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < GV::E; ++e)
        if ((e + 1) * 256 <= GV::NV4 || loff[e] >= 0) {
          float* const p = lds + c * GV::CS + loff[e];
          *reinterpret_cast<f32x2*>(p) = (f32x2){sv[c][e][0], sv[c][e][2]};
          *reinterpret_cast<f32x2*>(p + GV::PW) = (f32x2){sv[c][e][1], sv[c][e][3]};
        }
    return;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const Chain t = load_chain(chain, min(c0 + c, Cin - 1));
#pragma unroll
    for (int e = 0; e < GV::E; ++e) {
      const float m = goff[e] >= 0 ? 1.f : 0.f;           // zero padding stays zero through a factor (as conv_q4_mfma.hip)
      if ((e + 1) * 256 <= GV::NV4 || loff[e] >= 0) {
        float* const p = lds + c * GV::CS + loff[e];
        *reinterpret_cast<f32x2*>(p) = (f32x2){m * apply_chain(t, sv[c][e][0]), m * apply_chain(t, sv[c][e][2])};
        *reinterpret_cast<f32x2*>(p + GV::PW) = (f32x2){m * apply_chain(t, sv[c][e][1]), m * apply_chain(t, sv[c][e][3])};
      }
    }
  }
}

// ---------------------------------------------------------------- forward / backward-data ---------------------------
// WPE = waves per SIMD the register allocation is held to: 3 pays when Cin <= 8 (one or two chunks: the staging of a
// tile is not hidden behind its own MFMAs, only behind other workgroups'), 2 (no cap) is faster for long channel loops.
// PERSIST: the workgroup walks tiles vt = blockIdx.x, blockIdx.x + gridDim.x, ... and requests the first chunk (and, via
// the plane-wise weight reload, the first weights) of its NEXT tile behind the last chunk's MFMAs of the current one, so
// tile addressing, first-load latency and the epilogue's stores are no longer exposed once per tile (they are ~20 % of a
// tile for Cin = 25 and more than half for Cin = 4 / 8).
#ifdef DPI_TRACE
__device__ long long g_blk[8192][4];
__device__ long long g_trace[4][64];
#define TR(i) do { if (trc >= 0 && tid == 0) g_trace[trc][i] = clock64(); } while (0)
#define TRC(i) do { if (trc >= 0 && tid == 0 && (i) < 38) g_trace[trc][i] = clock64(); } while (0)   // per-chunk slots: first 9 chunks
#else
#define TR(i)
#define TRC(i)
#endif
// TAILPACK: a channel count of 4m + 1 (25, 13, 17, 105: the MultiRes widths) leaves ONE channel for the last K = 4 group.
// Instead of 27 MFMAs per output tile with three zero K-slices, that channel's taps are packed four to an MFMA
// (K = tap 4g + lk; the staged group holds the channel in all four slots, so lane group lk reads its own slot at its own
// tap offset): 7 MFMAs per tile, -10.6 % of all MFMAs for Cin = 25.
// IOB: the launch reads or writes a bf16 tensor (MArgs::xb / yb).  A template parameter, not just a run-time flag: the second load / store
// path costs the register-capped fp32 variants 12-70 bytes of scratch per lane (measured on the ISA), so the fp32 instantiations are
// compiled without it.
// YLOOP (round 5): a layer whose staged input is ONE 4-channel chunk (backward-data of 67->4: the gradient has four channels) and whose
// output has several 16-channel tiles (67 = 5 tiles).  One workgroup per spatial tile stages the chunk ONCE and walks the MArgs::ny channel
// tiles itself — weights of the next tile requested behind the current tile's last MFMAs, accumulators / second input / epilogue per tile —
// instead of ny workgroups each addressing, fetching and staging the same four channels.
template <int KD, int NR, int NH, bool FLIP, int S = 1, int WPE = 2, bool PERSIST = false, bool TAILPACK = false, bool IOB = false, bool YLOOP = false,
          bool S2V = false>
__global__ __launch_bounds__(256, WPE) void conv_mfma_kernel(MArgs a) {
  static_assert(!YLOOP || (!PERSIST && !TAILPACK), "the channel-tile loop is built for the one-tile-per-workgroup variants without a packed tail");
  static_assert(!S2V || (S == 2 && !FLIP && !PERSIST && !TAILPACK && !IOB && NR < 4), "vectorised stride-2 staging: forward, fp32 tensors, row-split tiles");
  using GV = GeoV2<KD, NR < 4 ? NR : 2, NH>;
  if constexpr (!IOB) { a.xb = 0; a.yb = 0; }
  using G = Geo<KD, NR, NH, S>;
  constexpr int TAPS = KD * 9;
  constexpr int PD = (KD - 1) / 2;
  constexpr int NT = NR * NH;                       // voxel tiles per wave
  __shared__ __attribute__((aligned(16))) float lds[4 * (S2V ? GV::CS : G::CS)];
  __shared__ double red[4][16][2];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  const int ntiles = a.ntd * a.nth * a.ntw;
  int vt0 = blockIdx.x, ytile = blockIdx.y;
  if constexpr (!PERSIST) {
    if (a.ny > 0) {
      const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
      const int i = YLOOP ? j : j / a.ny;
      ytile = YLOOP ? 0 : j - i * a.ny;
      if (i >= q8 + (xcd < r8 ? 1 : 0)) return;               // past this XCD's tile range (whole workgroup, before any barrier)
      vt0 = i * 8 + xcd;                                       // xcd_tile() maps it to tile i of XCD xcd's range
    }
  }
  int n0 = ytile * 16;                      // (YLOOP: advanced per channel tile)
  const size_t V = (size_t)a.D * a.H * a.W;
  const int Do = (a.D + 2 * PD - KD) / G::SD + 1, Ho = (a.H - 1) / S + 1, Wo = (a.W - 1) / S + 1;
  const size_t Vo = (size_t)Do * Ho * Wo;
#ifdef DPI_TRACE
  const int trc = blockIdx.y != 0 ? -1 : blockIdx.x == 0 ? 0 : blockIdx.x == 1 ? 1 : blockIdx.x == 30 ? 2 : blockIdx.x == 50 ? 3 : -1;
  TR(0);
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 8192) {
    g_blk[blockIdx.x][0] = wall_clock64();
    g_blk[blockIdx.x][2] = __builtin_amdgcn_s_getreg(63492);
    g_blk[blockIdx.x][3] = __builtin_amdgcn_s_getreg(63508);
  }
#endif
  auto tile_origin = [&](int vt, int& tile_id, int& od0, int& oh0, int& ow0) {
    tile_id = xcd_tile(vt, ntiles);
    int bt = tile_id;
    const int tw_i = bt % a.ntw; bt /= a.ntw;
    const int th_i = bt % a.nth; bt /= a.nth;
    od0 = bt * G::TZ; oh0 = th_i * G::TY; ow0 = tw_i * G::TW;
  };

  const int wz = G::SLICES ? wid : 0, wh = G::SLICES ? 0 : wid * NR;
  const int lbase = S2V ? lk * GV::CS + wh * 2 * GV::RS + lj : lk * G::CS + wz * G::SD * G::DS + wh * S * G::RS + lj * S;

  // weights of a 4-channel chunk: lane (co = lj, ci = lk) keeps its TAPS filter taps.
  // The taps of depth plane kd are dead once that plane's steps are done, so the NEXT chunk's weights replace them plane
  // by plane: requested (raw, from a clamped address) into 9 staging registers when the plane starts, selected into
  // place when it ends — a whole plane of MFMAs hides the load, and no second 27-register set is needed.
  int co_w = n0 + lj;
  auto w_ptr = [&](int c0, bool& ok) {
    const int ci = c0 + lk;
    ok = co_w < a.Cout && ci < a.Cin;
    return a.w + (ok ? co_w : 0) * a.w_out_stride + (ok ? ci : 0) * a.w_in_stride;
  };
  auto load_w_raw = [&](float (&wn)[9], int c0, int kd) {
    bool ok;
    const float* __restrict__ wp = w_ptr(c0, ok);
#pragma unroll
    for (int t = 0; t < 9; ++t) wn[t] = wp[FLIP ? (TAPS - 1 - (kd * 9 + t)) : kd * 9 + t];
  };
  auto commit_w = [&](float (&wr)[TAPS], const float (&wn)[9], int c0, int kd) {
    const bool ok = co_w < a.Cout && c0 + lk < a.Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[kd * 9 + t] = ok ? wn[t] : 0.f;
  };

  const int c_lo = a.split_cps ? (int)blockIdx.z * a.split_cps : 0;
  float* __restrict__ const ybase = a.y + (size_t)blockIdx.z * a.Cout * Vo;      // (blockIdx.z > 0 only in split launches: fp32 workspace)
  constexpr int EE = S2V ? GV::E : G::E;
  int goff[EE], loff[EE];
  float wr[TAPS], wn[9], sr[S2V ? 1 : 4][S2V ? 1 : G::E];
  f32x4 sv[S2V ? 4 : 1][S2V ? GV::E : 1];
  // staging of one 4-channel chunk: scalar slots of the halo tile, or (S2V) aligned float4 pieces into even / odd column planes
  auto slots = [&](int od, int oh, int ow) {
    if constexpr (S2V) tile_slots_v<GV>(tid, od * G::SD - PD, oh * 2 - 1, ow * 2 - 4, a.D, a.H, a.W, goff, loff);
    else tile_slots<G>(tid, od * G::SD - PD, oh * S - 1, ow * S - 1, a.D, a.H, a.W, goff, loff);
  };
  auto sload = [&](int c0) {
    if constexpr (S2V) stage_load_v<GV>(sv, a.x, a.Cin, V, c0, goff);
    else stage_load<G>(sr, a.x, a.Cin, V, c0, goff, a.xb);
  };
  auto sstore = [&](int c0) {
    if constexpr (S2V) stage_store_v<GV>(lds, sv, a.chain, a.Cin, c0, goff, loff);
    else stage_store<G>(lds, sr, a.chain, a.Cin, c0, goff, loff, a.xb);
  };
  int vt = vt0, tile_id, od0, oh0, ow0;
  tile_origin(vt, tile_id, od0, oh0, ow0);
  slots(od0, oh0, ow0);
  sload(c_lo);
  {
    bool ok;
    const float* __restrict__ wp = w_ptr(c_lo, ok);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) wr[t] = wp[FLIP ? (TAPS - 1 - t) : t];     // all in flight together
#pragma unroll
    for (int t = 0; t < TAPS; ++t) wr[t] = ok ? wr[t] : 0.f;
  }

  constexpr int NTG = (TAPS + 3) / 4;                // tap groups of the packed tail
  const bool tail = TAILPACK && S == 1 && (a.Cin & 3) == 1 && (a.Cin > 4 || a.Cin == 1);   // Cin = 1: the packed tail is the whole layer
  const int cin_main = a.split_cps ? min(c_lo + a.split_cps, a.Cin) : (tail ? a.Cin - 1 : a.Cin);   // (the split never runs the tap-packed variants)
  int ttoff[TAILPACK ? NTG : 1];
  float wt[TAILPACK ? NTG : 1];
  if constexpr (TAILPACK) {
#pragma unroll
    for (int g = 0; g < NTG; ++g) {
      const int t = 4 * g + lk;
      const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
      ttoff[g] = t < TAPS ? kd * G::DS + kh * G::RS + kw : 0;
      wt[g] = 0.f;
    }
  }
  auto load_tail_w = [&]() {                           // raw loads (clamped address), selected in commit_tail_w
    if constexpr (TAILPACK) {
      const bool ok = co_w < a.Cout;
      const float* __restrict__ wp = a.w + (ok ? co_w : 0) * a.w_out_stride + (a.Cin - 1) * a.w_in_stride;
#pragma unroll
      for (int g = 0; g < NTG; ++g) {
        const int t = 4 * g + lk < TAPS ? 4 * g + lk : TAPS - 1;
        wt[g] = wp[FLIP ? (TAPS - 1 - t) : t];
      }
    }
  };
  auto commit_tail_w = [&]() {
    if constexpr (TAILPACK) {
#pragma unroll
      for (int g = 0; g < NTG; ++g) wt[g] = (co_w < a.Cout && 4 * g + lk < TAPS) ? wt[g] : 0.f;
    }
  };

  for (;;) {
    const int vt_next = vt + (int)gridDim.x;
    const bool has_next = PERSIST && vt_next < ntiles;
    int tile_n = 0, od_n = 0, oh_n = 0, ow_n = 0;
    if (has_next) tile_origin(vt_next, tile_n, od_n, oh_n, ow_n);

    const int nyt = YLOOP ? a.ny : 1;
    for (int yt = 0; yt < nyt; ++yt) {
    if constexpr (YLOOP) {
      if (yt > 0) {       // next channel tile of the same staged chunk: its weights were requested (raw) behind the previous tile's MFMAs
        n0 = yt * 16; co_w = n0 + lj;
        const bool ok = co_w < a.Cout && c_lo + lk < a.Cin;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) wr[t] = ok ? wr[t] : 0.f;
      }
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.accumulate) {
      // gradient fan-in: start the accumulators from the destination (loads overlap the first chunk's staging)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = n0 + 4 * lk + r;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int od = od0 + wz, oh = oh0 + wh + t / NH, ow = ow0 + (t % NH) * 16 + lj;
          if (co < a.Cout && od < Do && oh < Ho && ow < Wo) acc[t][r] = dpi_ld(ybase, (size_t)co * Vo + ((size_t)od * Ho + oh) * Wo + ow, a.yb);
        }
      }
    }

    TR(1);
    for (int c0 = c_lo; c0 < cin_main; c0 += 4) {
      if (!YLOOP || yt == 0) {                           // (YLOOP: the one chunk stays staged for every channel tile)
      __syncthreads();                                   // everyone is done reading the previous chunk
      TRC(2 + (c0 / 4) * 4);
      sstore(c0);
      TRC(3 + (c0 / 4) * 4);
      __syncthreads();
      TRC(4 + (c0 / 4) * 4);
      }
      const bool more = c0 + 4 < cin_main;
      const bool tail_next = tail && !more;
      if (more || tail_next) sload(c0 + 4);                        // prefetch the next group behind this one's MFMAs
      else if (has_next) {                                        // ... or the first chunk of the next tile
        slots(od_n, oh_n, ow_n);
        sload(c_lo);
      }
      // software-pipelined walk over (kd, input row): LDS values of step s+1 are requested before the MFMAs of step s
      float bc[G::NB], bn[G::NB];
      auto load_b = [&](float (&b)[G::NB], int step) {
        const int kd = step / G::NROW, ir = step % G::NROW;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            if constexpr (S2V)      // input column 2 (ow0 + 16 h + lj) + kw - 1 = piece-relative 2 m + kw + 3: kw 0 -> odd plane m + 1, 1 -> even m + 2, 2 -> odd m + 2
              b[h * 3 + kw] = lds[lbase + kd * GV::DS + ir * GV::RS + h * 16 + (kw == 1 ? 2 : GV::PW + (kw == 0 ? 1 : 2))];
            else b[h * 3 + kw] = lds[lbase + kd * G::DS + ir * G::RS + h * 16 * S + kw];
          }
      };
      load_b(bc, 0);
      const int cn = more ? c0 + 4 : c_lo;                  // chunk whose weights are fetched next (chunk 0: next tile / harmless)
      if (tail_next) load_tail_w();
#pragma unroll
      for (int step = 0; step < G::NSTEP; ++step) {
        const int kd = step / G::NROW, ir = step % G::NROW;
        if (step + 1 < G::NSTEP) load_b(bn, step + 1);
        if constexpr (!YLOOP) { if (ir == 0) load_w_raw(wn, cn, kd); }
        // keep the requests above AHEAD of this step's MFMAs (the scheduler otherwise sinks every ds_read to just before
        // its first use and the wave then waits out the full LDS latency ~27 times per chunk)
        if (WPE <= 2) __builtin_amdgcn_sched_barrier(0);
        // kw outermost: consecutive MFMAs go to DIFFERENT accumulators (a dependent 16x16x4 f32 MFMA needs 40 cycles,
        // an independent one issues every 32)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int hr = (ir - kh) / S;                // output row fed by this input row through tap kh
            if (ir - kh >= 0 && (ir - kh) % S == 0 && hr < NR) {
#pragma unroll
              for (int h = 0; h < NH; ++h)
                acc[hr * NH + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[(kd * 3 + kh) * 3 + kw], bc[h * 3 + kw], acc[hr * NH + h], 0, 0, 0);
            }
          }
        if (WPE <= 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G::NB; ++i) bc[i] = bn[i];
        if constexpr (!YLOOP) { if (ir == G::NROW - 1) commit_w(wr, wn, cn, kd); }
      }
      if (tail_next) commit_tail_w();
      TRC(5 + (c0 / 4) * 4);
    }
    if constexpr (TAILPACK) {
      if (tail && cin_main == 0) { load_tail_w(); commit_tail_w(); }     // a single input channel: no main chunk has fetched the packed weights
      if (tail) {
        __syncthreads();
        sstore(cin_main);     // all four slots: channel Cin-1
        __syncthreads();
        if (has_next) {
          slots(od_n, oh_n, ow_n);
          sload(0);
        }
#pragma unroll
        for (int hr = 0; hr < NR; ++hr) {
          float tb[NH][NTG];
#pragma unroll
          for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int g = 0; g < NTG; ++g) tb[h][g] = lds[lbase + ttoff[g] + hr * G::RS + h * 16];
#pragma unroll
          for (int g = 0; g < NTG; ++g)
#pragma unroll
            for (int h = 0; h < NH; ++h)
              acc[hr * NH + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[g], tb[h][g], acc[hr * NH + h], 0, 0, 0);
        }
      }
    }
    if constexpr (YLOOP) {
      if (yt + 1 < nyt) {       // wr is dead until the next channel tile: its raw weights (clamped address) travel behind the second input / epilogue
        const int con = (yt + 1) * 16 + lj, ci = c_lo + lk;
        const bool okn = con < a.Cout && ci < a.Cin;
        const float* __restrict__ wp = a.w + (okn ? con : 0) * a.w_out_stride + (okn ? ci : 0) * a.w_in_stride;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) wr[t] = wp[FLIP ? (TAPS - 1 - t) : t];
      }
    }
    if constexpr (FLIP && !PERSIST) if (a.x2 != nullptr) {     // (only backward-data launches of the non-persistent variants carry a second input)
      // 1x1x1 contribution: D[co][vox] += A[co][c 4] * B[c 4][vox] per 4-channel group of x2, B straight from global memory in MFMA
      // layout (lane = (c = lk, vox = lj): 64 B per channel and row segment; each value is used by this wave only, so LDS would
      // add nothing).  ONE buffer over all of x2: channels >= C2 and out-of-volume voxels are out of range -> 0.
      const __amdgpu_buffer_rsrc_t r2 = dpi_buffer_t(a.x2, (size_t)a.C2 * Vo, a.xb);
      int voff[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int od = od0 + wz, oh = oh0 + wh + t / NH, ow = ow0 + (t % NH) * 16 + lj;
        voff[t] = (od < Do && oh < Ho && ow < Wo) ? (od * Ho + oh) * Wo + ow : -1;
      }
      const int nk2 = (a.C2 + 3) >> 2;
      auto load2 = [&](float (&b)[NT], float& wv, int kc) {
        const int c = kc * 4 + lk;
        const bool ok = c < a.C2 && co_w < a.Cout;
        const float wraw = a.w2[(ok ? co_w : 0) * a.w2_co_stride + (ok ? c : 0) * a.w2_c_stride];
        wv = ok ? wraw : 0.f;
        const int cbase = c * (int)Vo;                      // host guarantees C2 * Vo * 4 < 2^31 (+ one channel group of slack)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          b[t] = a.xb ? dpi_buffer_load_bf16(r2, voff[t] >= 0 ? cbase + voff[t] : -1) : dpi_buffer_load(r2, voff[t] >= 0 ? (cbase + voff[t]) * 4 : -4);
      };
      float b0[NT], b1[NT], w0v, w1v;
      load2(b0, w0v, 0);
      for (int kc = 0; kc < nk2; kc += 2) {                 // two groups per trip: the next group's loads are in flight behind this one's MFMAs
        if (kc + 1 < nk2) load2(b1, w1v, kc + 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0v, b0[t], acc[t], 0, 0, 0);
        if (kc + 1 < nk2) {
          if (kc + 2 < nk2) load2(b0, w0v, kc + 2);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1v, b1[t], acc[t], 0, 0, 0);
        }
      }
    }
    TR(40);

    // ---- epilogue: D row = co (4*lk + r), D col = voxel lj -----------------------------------------------------------
    // interior tile with a full channel tile (the common case): no per-element bounds tests, one base pointer per row r
    const bool interior = od0 + G::TZ <= Do && oh0 + G::TY <= Ho && ow0 + G::TW <= Wo && n0 + 16 <= a.Cout;
    const int vbase = ((od0 + wz) * Ho + oh0 + wh) * Wo + ow0 + lj;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = n0 + 4 * lk + r;
      const bool cok = co < a.Cout;
      const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
      float* __restrict__ yc = dpi_at(ybase, (size_t)(cok ? co : 0) * Vo + vbase, a.yb);
      double s = 0.0, q = 0.0;
      if (a.yb) {
        // bf16 destination: two voxels per dword store where the rows are even-aligned (dpi_st_bf16_row); statistics describe what is stored
        const bool pairs = !(Wo & 1) && !(Vo & 1) && !((uintptr_t)ybase & 3);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int od = od0 + wz, oh = oh0 + wh + t / NH, ow = ow0 + (t % NH) * 16 + lj;
          const bool ok = interior || (cok && od < Do && oh < Ho && ow < Wo);
          const float v = dpi_round_bf16(acc[t][r] + bv);
          dpi_st_bf16_row(yc, (t / NH) * Wo + (t % NH) * 16, v, ok, pairs, lj);
          if (ok) { s += v; q += (double)v * v; }
        }
      } else if (interior) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float v = acc[t][r] + bv;
          yc[(t / NH) * Wo + (t % NH) * 16] = v;
          if (a.partials) { s += v; q += (double)v * v; }
        }
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int od = od0 + wz, oh = oh0 + wh + t / NH, ow = ow0 + (t % NH) * 16 + lj;
          if (cok && od < Do && oh < Ho && ow < Wo) {
            const float v = acc[t][r] + bv;
            yc[(t / NH) * Wo + (t % NH) * 16] = v;
            s += v;
            q += (double)v * v;
          }
        }
      }
      if (a.partials) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (lj == 0) { red[wid][4 * lk + r][0] = s; red[wid][4 * lk + r][1] = q; }
      }
    }
    if (a.partials) {
      __syncthreads();
      if (tid < 32) {
        const int c = tid >> 1, which = tid & 1;
        const double rsum = red[0][c][which] + red[1][c][which] + red[2][c][which] + red[3][c][which];
        if (n0 + c < a.Cout) a.partials[((size_t)tile_id * a.Cout + n0 + c) * 2 + which] = rsum;
      }
    }
    }       // channel tiles (YLOOP)
    TR(41);
    if (!has_next) break;
    vt = vt_next; tile_id = tile_n; od0 = od_n; oh0 = oh_n; ow0 = ow_n;
  }
#ifdef DPI_TRACE
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 8192) g_blk[blockIdx.x][1] = wall_clock64();
#endif
}

// ---------------------------------------------------------------- backward-weight ----------------------------------
//   dW[co][ci][tap] = sum_v dY[co][v] * X'[ci][v + tap]
//   D[co 16][tap 16] += A[co 16][vox 4] * B[vox 4][tap 16]     per input channel, two tap tiles (27 of 32 columns used)
//      A = dY straight from global memory (lane: co = l&15, voxel = 4s + (l>>4)); one load feeds 8 MFMAs
//      B = X' halo tile in LDS, lane: voxel = 4s + (l>>4), tap = 16*tt + (l&15) -> per-lane tap offset
struct BwMArgs {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ dy;
  float* __restrict__ ws;   // [nchunks][Cout][Cin][TAPS]
  int Cin, Cout;
  int D, H, W;
  int ntd, nth, ntw, ntiles, tiles_per_chunk;
  // swap = 1: the roles are exchanged — `x` (staged in LDS, 4 channels per block, tap-shifted) is dY and `dy` (the
  // 16-row A operand) is X:  D[ci][(co, t')] = sum_v X[ci][v] dY[co][v + t'] = dW[co][ci][TAPS-1-t'].  With few output
  // channels (64 -> 4: M = 4 of 16 rows the usual way round) this fills the tile: M = 16 input channels, N = 4 co x 27.
  int swap;
  int y0;                   // first 4-channel group of this launch (blockIdx.y is relative to it)
  // XCD-aware order (ngroups > 0): the launch is 1-D over L = xcd + 8 * (group + ngroups * (chunk / 8)) with chunk % 8 = xcd.
  // Workgroup L runs on XCD L % 8 (dispatch is round-robin over the 8 XCDs), so the ngroups workgroups that share a chunk's dY
  // rows — and the chunk's X halo rows — run back to back on ONE XCD and find them in its L2 instead of re-reading HBM
  // (measured before: 3.4 GB fetched per launch for 0.69 GB algorithmic on 25 -> 16).
  int ngroups, nchunks;
  // storage type of the tensor behind `x` (staged) and behind `dy` (the 16-row operand), roles as AFTER a swap: 0 = fp32, 1 = bf16;
  // dyb = 2: bf16 whose row pieces are not 4-byte aligned (odd row length or channel size): four 2-byte loads instead of one 8-byte load
  int xb, dyb;
};

// TC != 0: the launch covers a final group that holds only TC real channels (Cin = 4m + 1: TC = 1) and computes just their
// (channel, tap) column tiles: 2 instead of 7 — the padded channels' 80 columns are not multiplied at all.
template <int KD, int S, int NR, int NH, int NG = 1>
struct BwLds {
  using G = Geo<KD, NR, NH, S>;
  static constexpr int NTQ0 = (4 * KD * 9 + 15) / 16;
  static constexpr int LDSF = NG * (4 * G::CS > 4 * NTQ0 * 256 ? 4 * G::CS : 4 * NTQ0 * 256);   // halo tile of 4 NG channels / the cross-wave reduction
  static constexpr int DYRS = 4 * (4 * NH) + 4;            // row stride of the dY transpose buffer: = 4 (mod 32) -> 2-pass reads
};

// NG = 2: the workgroup stages TWO 4-channel groups per tile and multiplies each dY operand into 14 column tiles instead of 7 — the
// dY rows (fetch, transpose through LDS, 64 reads per wave and tile), the tile addressing and the two barriers are paid once per 896
// MFMAs instead of once per 448, and a layer's dY is streamed half as often (70 KB of LDS: two workgroups per CU, which costs this
// kernel nothing).  first_group: the first of the NG groups this workgroup owns (-1: blockIdx.y, or the XCD-aware 1-D order).
template <int KD, int S, int NR, int NH, int TC, int NG = 1, bool IOB = false>
__device__ __forceinline__ void conv_bwd_weight_mfma_body(const BwMArgs& a_in, float* __restrict__ lds, float* __restrict__ dyl_all, int first_group = -1) {
  BwMArgs a = a_in;
  if constexpr (!IOB) { a.xb = 0; a.dyb = 0; }       // (a template parameter for the same reason as in conv_mfma_kernel)
  static_assert(NG == 1 || TC == 0, "the column-trimmed tail is a single group");
#ifdef DPI_TRACE
  const int trc = (blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 4) ? (int)blockIdx.x : -1;
  int trn = 0;
#define TRW() do { if (trc >= 0 && threadIdx.x == 0 && trn < 64) g_trace[trc][trn++] = clock64(); } while (0)
#else
#define TRW()
#endif
  TRW();
  using G = Geo<KD, NR, NH, S>;
  constexpr int TAPS = KD * 9;
  constexpr int PD = (KD - 1) / 2;
  constexpr int NQ = (TC ? TC : 4) * TAPS;    // (channel, tap) columns of this block: 108 (3-D) / 36 (2-D); tail: TC x TAPS
  constexpr int NTQ = (NQ + 15) / 16;         // MFMA column tiles: 7 (96 % full) / 3 (75 %)
  constexpr int KS = 4 * NH;                  // k-steps (4 voxels each) per output row
  constexpr int JP = KS / 4;                  // float4 pieces of a dY row each lane fetches (16 channels x KS pieces / 64 lanes)
  constexpr int DYRS = BwLds<KD, S, NR, NH>::DYRS;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  const int wch = lane >> 2, wp = lane & 3;   // dY fetch mapping: channel, float4 piece
  float* __restrict__ dyw = dyl_all + wid * 16 * DYRS;
  int chunk = blockIdx.x, group = first_group >= 0 ? first_group : (int)blockIdx.y;
  if (first_group < 0 && a.ngroups > 0) {
    const int L = blockIdx.x, r = L >> 3;
    group = r % a.ngroups;
    chunk = (r / a.ngroups) * 8 + (L & 7);
    if (chunk >= a.nchunks) return;
  }
  const int c0 = (group + a.y0) * 4, n0 = blockIdx.z * 16;
  const size_t V = (size_t)a.D * a.H * a.W;
  const int Do = (a.D + 2 * PD - KD) / G::SD + 1, Ho = (a.H - 1) / S + 1, Wo = (a.W - 1) / S + 1;
  const size_t Vo = (size_t)Do * Ho * Wo;
  const int wz = G::SLICES ? wid : 0, wh = G::SLICES ? 0 : wid * NR;

  // per-lane (channel, tap) offsets inside the halo tile: column q = 16 t + lj -> channel q / TAPS, tap q % TAPS
  int toff[NTQ];
#pragma unroll
  for (int t = 0; t < NTQ; ++t) {
    int q = t * 16 + lj;
    if (q >= NQ) q = NQ - 1;                   // unused columns: any valid address
    const int c = q / TAPS, tap = q % TAPS;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    toff[t] = c * G::CS + kd * G::DS + kh * G::RS + kw;
  }
  const int lbase = wz * G::SD * G::DS + wh * S * G::RS + lk * S;
  const __amdgpu_buffer_rsrc_t dyb = dpi_buffer_t(dpi_at(a.dy, (size_t)n0 * Vo, a.dyb), (size_t)min(16, a.Cout - n0) * Vo, a.dyb);

  constexpr int NA = NG * NTQ;                 // accumulators / B operands per k-step
  f32x4 acc[NA];
#pragma unroll
  for (int t = 0; t < NA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int t_begin = chunk * a.tiles_per_chunk;
  const int t_end = min(t_begin + a.tiles_per_chunk, a.ntiles);
  int goff[G::E], loff[G::E];
  float sr[4][G::E], sr2[NG == 2 ? 4 : 1][NG == 2 ? G::E : 1];
  auto tile_origin = [&](int tile, int& od0, int& oh0, int& ow0) {
    int bt = tile;
    const int tw_i = bt % a.ntw; bt /= a.ntw;
    const int th_i = bt % a.nth; bt /= a.nth;
    od0 = bt * G::TZ; oh0 = th_i * G::TY; ow0 = tw_i * G::TW;
  };
  int od0, oh0, ow0;
  if (t_begin < t_end) {
    tile_origin(t_begin, od0, oh0, ow0);
    tile_slots<G>(tid, od0 * G::SD - PD, oh0 * S - 1, ow0 * S - 1, a.D, a.H, a.W, goff, loff);
    stage_load<G>(sr, a.x, a.Cin, V, c0, goff, a.xb);
    if constexpr (NG == 2) stage_load<G>(sr2, a.x, a.Cin, V, c0 + 4, goff, a.xb);
  }
  for (int tile = t_begin; tile < t_end; ++tile) {
    TRW();
    __syncthreads();
    TRW();
    stage_store<G>(lds, sr, a.chain, a.Cin, c0, goff, loff, a.xb);
    if constexpr (NG == 2) stage_store<G>(lds + 4 * G::CS, sr2, a.chain, a.Cin, c0 + 4, goff, loff, a.xb);
    __syncthreads();
    TRW();
    const int cod = od0 + wz, coh0 = oh0 + wh, cow0 = ow0;
    // dY rows reach the MFMA A layout (lane = (co, voxel 4s + lk)) through a wave-private LDS transpose: read straight
    // from global memory that layout makes every load touch 16 channels x 16 B (the L1 / address path, not the matrix
    // pipe, then bounds the kernel: measured 60 -> 86 TFLOP/s with the loads collapsed onto one channel).  Here each lane
    // fetches whole float4 pieces (lane -> channel l>>2, piece l&3 [+4]: 64 B contiguous per channel and instruction).
    auto load_raw = [&](f32x4 (&raw)[JP], int hr) {
      const int oh = coh0 + hr;
      const bool row_ok = n0 + wch < a.Cout && cod < Do && oh < Ho;
      const int base = wch * (int)Vo + (cod * Ho + oh) * Wo + cow0;       // host guarantees 16 * Vo * 4 < 2^31
#pragma unroll
      for (int j = 0; j < JP; ++j) {
        const int p4 = 4 * (wp + 4 * j);
        if (a.dyb == 1) {        // four bf16 in one 8-byte piece, kept raw in [0], [1] (put_row widens)
          const dpi_u32x2v u = __builtin_bit_cast(dpi_u32x2v, __builtin_amdgcn_raw_buffer_load_b64(dyb, (row_ok && cow0 + p4 < Wo) ? (base + p4) * 2 : -8, 0, 0));
          const unsigned u0 = u.x, u1 = u.y;     // (hipcc: __builtin_bit_cast of a vector SUBSCRIPT u[1] yields element 0 — copy to scalars first)
          raw[j] = (f32x4){__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1), 0.f, 0.f};
        } else if (a.dyb) {      // pieces at odd element offsets: a multi-dword buffer load needs dword alignment; raw 16 bits each
#pragma unroll
          for (int e = 0; e < 4; ++e) raw[j][e] = dpi_buffer_load_bf16_raw(dyb, (row_ok && cow0 + p4 < Wo) ? base + p4 + e : -1);
        } else
          raw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyb, (row_ok && cow0 + p4 < Wo) ? (base + p4) * 4 : -16, 0, 0));
      }
    };
    auto put_row = [&](const f32x4 (&raw)[JP]) {
#pragma unroll
      for (int j = 0; j < JP; ++j) {
        const int p4 = 4 * (wp + 4 * j);
        f32x4 v = raw[j];
        if (a.dyb == 1) {
          const float4 wv = dpi_widen_raw4(make_float4(v[0], v[1], 0.f, 0.f));
          v = (f32x4){wv.x, wv.y, wv.z, wv.w};
        } else if (a.dyb) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = dpi_widen_raw(v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = cow0 + p4 + e < Wo ? v[e] : 0.f;   // columns past the row end belong to the next row
        *reinterpret_cast<f32x4*>(dyw + wch * DYRS + p4) = v;
      }
    };
    auto get_row = [&](float (&g)[KS]) {
#pragma unroll
      for (int s = 0; s < KS; ++s) g[s] = dyw[lj * DYRS + 4 * s + lk];
    };
    // vmcnt retires in order: rows requested AFTER the next tile's prefetch would make their first use wait for the
    // whole prefetch.  So the first rows are requested before it, later rows two rows ahead of their use.
    constexpr int NPRE = NR < 4 ? NR : 4;
    f32x4 raw[NR][JP];
#pragma unroll
    for (int hr = 0; hr < NPRE; ++hr) load_raw(raw[hr], hr);
    if (tile + 1 < t_end) {                                  // prefetch the next tile behind this tile's MFMAs
      tile_origin(tile + 1, od0, oh0, ow0);
      tile_slots<G>(tid, od0 * G::SD - PD, oh0 * S - 1, ow0 * S - 1, a.D, a.H, a.W, goff, loff);
      stage_load<G>(sr, a.x, a.Cin, V, c0, goff, a.xb);
      if constexpr (NG == 2) stage_load<G>(sr2, a.x, a.Cin, V, c0 + 4, goff, a.xb);
    }
    TRW();
    // Software pipeline (as in the forward kernel): the B operands of step (hr, s + 1) and the A row of hr + 1 are requested
    // BEFORE the 7 MFMAs of step (hr, s) and pinned there with sched_barrier — hipcc otherwise sinks every ds_read to just
    // before its first use and the wave waits out the LDS latency once per step (MFMA pipe 58 % busy, round-2 SQ counters).
    // steps are (row hr, group g, k-step s): the A operands of a row serve both groups, the B operands are 7 per step either way
    auto load_b = [&](float (&b)[NTQ], int hr, int g, int s_) {
#pragma unroll
      for (int t = 0; t < NTQ; ++t) b[t] = lds[g * 4 * G::CS + lbase + hr * S * G::RS + 4 * s_ * S + toff[t]];
    };
    float gc[KS], gn[KS], bc[NTQ], bn[NTQ];
    put_row(raw[0]);
    get_row(gc);
    if (1 < NR) put_row(raw[1]);                             // LDS ops of a wave execute in order: safe after the reads above
    load_b(bc, 0, 0, 0);
#pragma unroll
    for (int hr = 0; hr < NR; ++hr) {
      if (hr + 3 < NR && hr + 3 >= NPRE) load_raw(raw[hr + 3], hr + 3);   // enters the row buffer 1.5 rows from now
#pragma unroll
      for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if (g == NG - 1 && s == KS / 2 && hr + 1 < NR) {     // next row's A operands; then its successor may enter the row buffer
            get_row(gn);
            if (hr + 2 < NR) put_row(raw[hr + 2]);
          }
          if (s + 1 < KS) load_b(bn, hr, g, s + 1);
          else if (g + 1 < NG) load_b(bn, hr, g + 1, 0);
          else if (hr + 1 < NR) load_b(bn, hr + 1, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < NTQ; ++t) acc[g * NTQ + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(gc[s], bc[t], acc[g * NTQ + t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < NTQ; ++t) bc[t] = bn[t];
        }
      }
      if (hr + 1 < NR) {
#pragma unroll
        for (int s = 0; s < KS; ++s) gc[s] = gn[s];
      }
    }
  }
  // ---- cross-wave reduction through LDS, then one partial per (chunk, co, ci, tap) ------------------------------------
  __syncthreads();
  float* red = lds;   // [4 waves][NA][4 r][64 lanes]
#pragma unroll
  for (int t = 0; t < NA; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wid * NA + t) * 4 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  for (int e = tid; e < NA * 256; e += 256) {
    const int l = e & 63, r = (e >> 6) & 3, t = (e >> 8) % NTQ, g = (e >> 8) / NTQ;
    const float sum = red[e] + red[e + NA * 256] + red[e + 2 * NA * 256] + red[e + 3 * NA * 256];
    const int co = n0 + 4 * (l >> 4) + r, q = t * 16 + (l & 15), ci = c0 + 4 * g + q / TAPS, tap = q % TAPS;
    if (co < a.Cout && q < NQ && ci < a.Cin) {
      if (a.swap) a.ws[(((size_t)chunk * a.Cin + ci) * a.Cout + co) * TAPS + (TAPS - 1 - tap)] = sum;   // (staged, A) = (co, ci)
      else a.ws[(((size_t)chunk * a.Cout + co) * a.Cin + ci) * TAPS + tap] = sum;
    }
  }
}

template <int KD, int S, int NR, int NH, int TC = 0, bool IOB = false>
__global__ __launch_bounds__(256) void conv_bwd_weight_mfma_kernel(BwMArgs a) {
  using L = BwLds<KD, S, NR, NH>;
  __shared__ __attribute__((aligned(16))) float lds[L::LDSF];
  __shared__ __attribute__((aligned(16))) float dyl[4 * 16 * L::DYRS];
  conv_bwd_weight_mfma_body<KD, S, NR, NH, TC, 1, IOB>(a, lds, dyl);
}

// Staged channel count 4m + 1 in ONE launch: the workgroups of the last group (blockIdx.y = gridDim.y - 1) run the column-trimmed
// body (one real channel: 2 MFMA column tiles instead of 7) next to the full groups, instead of a second launch behind them whose
// 1/25 of the work took 11 % of the time (25 -> 16 at 256x128x128: 0.94 + 0.11 ms).
template <int KD, int S, int NR, int NH>
__global__ __launch_bounds__(256) void conv_bwd_weight_mfma_merged_kernel(BwMArgs a) {
  using L = BwLds<KD, S, NR, NH>;
  __shared__ __attribute__((aligned(16))) float lds[L::LDSF];
  __shared__ __attribute__((aligned(16))) float dyl[4 * 16 * L::DYRS];
  if (blockIdx.y + 1 == gridDim.y) conv_bwd_weight_mfma_body<KD, S, NR, NH, 1>(a, lds, dyl);
  else conv_bwd_weight_mfma_body<KD, S, NR, NH, 0>(a, lds, dyl);
}

// Two groups per workgroup (see NG above): blockIdx.y < npairs owns groups 2y, 2y + 1; then, if the number of full groups is odd, one
// workgroup column with the last full group alone; then the column-trimmed one-channel tail group if the staged count is 4m + 1.
template <int KD, int S, int NR, int NH>
__global__ __launch_bounds__(256, 2) void conv_bwd_weight_mfma_pair_kernel(BwMArgs a, int npairs, int nfull) {
  using L = BwLds<KD, S, NR, NH, 2>;
  __shared__ __attribute__((aligned(16))) float lds[L::LDSF];
  __shared__ __attribute__((aligned(16))) float dyl[4 * 16 * L::DYRS];
  const int y = blockIdx.y;
  if (y < npairs) conv_bwd_weight_mfma_body<KD, S, NR, NH, 0, 2>(a, lds, dyl, 2 * y);
  else if (2 * npairs < nfull && y == npairs) conv_bwd_weight_mfma_body<KD, S, NR, NH, 0, 1>(a, lds, dyl, 2 * npairs);
  else conv_bwd_weight_mfma_body<KD, S, NR, NH, 1, 1>(a, lds, dyl, nfull);
}

// ---------------------------------------------------------------- backward-weight, few output channels ---------------
// With Cout <= 5 the [co x tap] MFMA tile above is three-quarters empty.  Move the kw shift to the dy side instead:
//     dW[co][ci][kd][kh][kw] = sum_{d,h,u} X'[ci][d+kd-1][h+kh-1][u] * dy[co][d][h][u - kw]      (u = input column)
//     D[(ci,kd,kh) 16][(co,kw) 16] += A[(ci,kd,kh)][u 4] * B[u 4][(co,kw)]
// A = halo tile in LDS (6 input channels x 9 (kd,kh) = 54 of 64 rows over 4 MFMA tiles; per-lane row offsets),
// B = dy read straight from global memory (lane: u = 4s + (l>>4), co = (l&15)/3, kw = (l&15)%3; 3*Cout of 16 columns).
// One B value feeds 4 MFMAs.  3-D, stride 1 only (the three full-resolution layers this exists for).
struct BwSArgs {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ dy;
  float* __restrict__ ws;   // [nchunks][Cout][Cin][27]
  int Cin, Cout;
  int D, H, W;
  int ntd, nth, ntw, ntiles, tiles_per_chunk;
  int xb, dyb;              // storage type of x / dy: 1 = bf16
};

__global__ __launch_bounds__(256) void conv_bwd_weight_smallco_kernel(BwSArgs a) {
  constexpr int CB = 6;                                  // input channels per block
  constexpr int TZ = 4, TY = 8, ID = 6, IH = 10, IW = 34, RS = 36;
  constexpr int CS0 = ID * IH * RS;                      // 2160
  constexpr int CS = CS0 + 2;                            // = 18 (mod 32): rows of one lane group land on distinct banks mostly
  constexpr int TILE = ID * IH * RS;                     // the 2 pad columns of every row are staged too (as zeros): the
  constexpr int E = (TILE + 255) / 256;                  // 9th k-step reads them, and 0 * stale-LDS-NaN would poison the sum
  __shared__ __attribute__((aligned(16))) float lds[CB * CS];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  const int c0 = blockIdx.y * CB;
  const size_t V = (size_t)a.D * a.H * a.W;

  // A-operand row of this lane in each of the 4 M tiles: q = 16 t + lj -> (channel q/9, kd, kh)
  int aoff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int q = 16 * t + lj;
    if (q >= CB * 9) q = CB * 9 - 1;                     // unused rows: any valid address
    const int cl = q / 9, kd = (q % 9) / 3, kh = q % 3;
    aoff[t] = cl * CS + (kd * IH + kh) * RS;
  }
  const int bco = lj / 3, bkw = lj % 3;
  const bool bok = bco < a.Cout && lj < 15;
  // dY rows go through a wave-private LDS row buffer (coalesced float4 fetch, see conv_bwd_weight_mfma_kernel):
  // [5 channels][4 zeros | 32 voxels | 8 zeros]; the shifted / out-of-tile columns of the B operand read the zero margins.
  constexpr int DYRS = 44;
  __shared__ __attribute__((aligned(16))) float dyl[4][5 * DYRS];
  float* __restrict__ dyw = dyl[wid];
  for (int i = lane; i < 5 * DYRS; i += 64) dyw[i] = 0.f;
  const __amdgpu_buffer_rsrc_t dyb = dpi_buffer_t(a.dy, (size_t)a.Cout * V, a.dyb);   // Cout <= 5: host keeps 5*V*4 < 2^31
  const int wch = lane >> 3, wp4 = 4 * (lane & 7);         // fetch mapping: channel, first voxel of the float4 piece
  const int boff = bok ? bco * DYRS + 4 + lk - bkw : 0;    // B operand: voxel 4s + lk - kw of channel co (index 0 is a zero)

  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int t_begin = blockIdx.x * a.tiles_per_chunk;
  const int t_end = min(t_begin + a.tiles_per_chunk, a.ntiles);
  int goff[E], loff[E];
  float sr[CB][E];
  auto slots = [&](int tile, int& od0, int& oh0, int& ow0) {
    int bt = tile;
    const int tw_i = bt % a.ntw; bt /= a.ntw;
    const int th_i = bt % a.nth; bt /= a.nth;
    od0 = bt * TZ; oh0 = th_i * TY; ow0 = tw_i * 32;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int idx = tid + e * 256;
      const int col = idx % RS, row = idx / RS;
      const int hy = row % IH, dz = row / IH;
      const int gd = od0 - 1 + dz, gh = oh0 - 1 + hy, gw = ow0 - 1 + col;
      const bool ok = idx < TILE && col < IW && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
      goff[e] = ok ? (gd * a.H + gh) * a.W + gw : -1;
      loff[e] = idx < TILE ? (dz * IH + hy) * RS + col : -1;
    }
  };
  auto gload = [&]() {                                   // buffer loads: goff = -1 (outside the volume) -> 0; channels past
#pragma unroll                                           // Cin re-read the last one (their rows of dW are never written)
    for (int c = 0; c < CB; ++c) {
      const __amdgpu_buffer_rsrc_t r = dpi_buffer_t(dpi_at(a.x, (size_t)min(c0 + c, a.Cin - 1) * V, a.xb), V, a.xb);
      if (a.xb) {
#pragma unroll
        for (int e = 0; e < E; ++e) sr[c][e] = dpi_buffer_load_bf16_raw(r, goff[e]);          // raw bits, widened at the LDS store below
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e) sr[c][e] = dpi_buffer_load(r, goff[e] * 4);
      }
    }
  };
  int od0 = 0, oh0 = 0, ow0 = 0;
  if (t_begin < t_end) { slots(t_begin, od0, oh0, ow0); gload(); }
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const Chain t = load_chain(a.chain, min(c0 + c, a.Cin - 1));
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float xv = a.xb ? dpi_widen_raw(sr[c][e]) : sr[c][e];
        const float v = goff[e] >= 0 ? apply_chain(t, xv) : xv;
        if ((e + 1) * 256 <= TILE || loff[e] >= 0) lds[c * CS + loff[e]] = v;
      }
    }
    __syncthreads();
    const int cod = od0 + wid, coh0 = oh0, cow0 = ow0;
    auto load_raw = [&](int hr) {
      const int oh = coh0 + hr;
      const bool ok = wch < a.Cout && cod < a.D && oh < a.H && cow0 + wp4 < a.W;
      const int el = wch * (int)V + (cod * a.H + oh) * a.W + cow0 + wp4;
      if (a.dyb == 1) {          // raw: put_row widens
        const dpi_u32x2v u = __builtin_bit_cast(dpi_u32x2v, __builtin_amdgcn_raw_buffer_load_b64(dyb, ok ? el * 2 : -8, 0, 0));
        const unsigned u0 = u.x, u1 = u.y;
        return (f32x4){__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1), 0.f, 0.f};
      }
      if (a.dyb) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dpi_buffer_load_bf16_raw(dyb, ok ? el + e : -1);
        return v;
      }
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dyb, ok ? el * 4 : -16, 0, 0));
    };
    auto put_row = [&](f32x4 v) {
      if (a.dyb == 1) {
        const float4 wv = dpi_widen_raw4(make_float4(v[0], v[1], 0.f, 0.f));
        v = (f32x4){wv.x, wv.y, wv.z, wv.w};
      } else if (a.dyb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dpi_widen_raw(v[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = cow0 + wp4 + e < a.W ? v[e] : 0.f;
      if (wch < 5) *reinterpret_cast<f32x4*>(dyw + wch * DYRS + 4 + wp4) = v;
    };
    auto get_row = [&](float (&g)[9]) {
#pragma unroll
      for (int s = 0; s < 9; ++s) g[s] = bok ? dyw[boff + 4 * s] : 0.f;
    };
    // (rows requested after the next tile's prefetch wait for it — vmcnt retires in order — so all TY rows of this
    //  tile are requested before it: one float4 per lane and row)
    f32x4 raw[TY];
#pragma unroll
    for (int hr = 0; hr < TY; ++hr) raw[hr] = load_raw(hr);
    if (tile + 1 < t_end) { slots(tile + 1, od0, oh0, ow0); gload(); }
    const int lrow = (wid * IH) * RS + lk;               // this wave's depth slice, column 4s + lk added below
    put_row(raw[0]);
#pragma unroll
    for (int hr = 0; hr < TY; ++hr) {
      float g[9];
      get_row(g);
      if (hr + 1 < TY) put_row(raw[hr + 1]);             // a wave's LDS operations execute in order: safe after the reads
#pragma unroll
      for (int s = 0; s < 9; ++s) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float av = lds[aoff[t] + lrow + hr * RS + 4 * s];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, g[s], acc[t], 0, 0, 0);
        }
      }
    }
  }
  // ---- cross-wave reduction, one partial per (chunk, co, ci, tap): D row q = 16t + 4*lk + r, col = (co, kw) ------------
  __syncthreads();
  float* red = lds;   // [4 waves][4 t][4 r][64 lanes]
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wid * 4 + t) * 4 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  for (int e = tid; e < 4 * 4 * 64; e += 256) {
    const int l = e & 63, r = (e >> 6) & 3, t = e >> 8;
    const float sum = red[e] + red[e + 1024] + red[e + 2048] + red[e + 3072];
    const int q = 16 * t + 4 * (l >> 4) + r, j = l & 15;
    const int cl = q / 9, kd = (q % 9) / 3, kh = q % 3, co = j / 3, kw = j % 3, ci = c0 + cl;
    if (q < CB * 9 && j < 15 && co < a.Cout && ci < a.Cin)
      a.ws[(((size_t)blockIdx.x * a.Cout + co) * a.Cin + ci) * 27 + (kd * 3 + kh) * 3 + kw] = sum;
  }
}

// ---------------------------------------------------------------- backward-data, stride 2 ---------------------------
// dx[ci][i] = sum_co sum_k dy[co][(i + 1 - k) / 2] * w[co][ci][k]   (per axis; only integer quotients contribute).
// Writing i = 2m + p: p = 0 uses tap k = 1 at o = m; p = 1 uses k = 2 at o = m and k = 0 at o = m + 1.  Each of the
// 2^nd parity classes is therefore a small stride-1 convolution of dy over the m grid (1/2/4/8 taps, 27 in total):
//     D[ci 16][m 16] += A[ci 16][co 4] * B[co 4][m 16 (+shift)]
// A = weights in registers (lane: ci = l&15, co = l>>4), B = dy halo tile (one extra sample per axis) in LDS.
// Workgroup = m tile 1 x 8 x 16 (3-D) / 1 x 8 x 16 (2-D), wave = 2 rows, all parity classes of those rows.
struct BdS2Args {
  const float* __restrict__ dy;
  const float* __restrict__ w;     // [Cout][Cin][TAPS]
  float* __restrict__ dx;
  int Cin, Cout;
  int D, H, W, Do, Ho, Wo;
  int ntd, nth, ntw;
  int accumulate;
  int dyb, dxb;            // storage type of dy / dx: 1 = bf16
};

template <int KD>
__global__ __launch_bounds__(256) void conv_bwd_data_s2_mfma_kernel(BdS2Args a) {
  constexpr int TAPS = KD * 9;
  constexpr int NCLS = KD == 3 ? 8 : 4;
  constexpr int ID = KD == 3 ? 2 : 1, IH = 9, IW = 17, RS = 20;
  constexpr int CS0 = ID * IH * RS;
  constexpr int CS = CS0 + ((16 - (CS0 % 32)) + 32) % 32;
  constexpr int TILE = ID * IH * IW;
  constexpr int E = (TILE + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[4 * CS];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  int bt = blockIdx.x;
  const int tw_i = bt % a.ntw; bt /= a.ntw;
  const int th_i = bt % a.nth; bt /= a.nth;
  const int md0 = bt, mh0 = th_i * 8, mw0 = tw_i * 16;
  const int n0 = blockIdx.y * 16;
  const size_t Vo = (size_t)a.Do * a.Ho * a.Wo, V = (size_t)a.D * a.H * a.W;

  int goff[E], loff[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int idx = tid + e * 256;
    const int col = idx % IW, row = idx / IW;
    const int hy = row % IH, dz = row / IH;
    const int gd = md0 + dz, gh = mh0 + hy, gw = mw0 + col;
    const bool ok = idx < TILE && gd < a.Do && gh < a.Ho && gw < a.Wo;
    goff[e] = ok ? (gd * a.Ho + gh) * a.Wo + gw : -1;
    loff[e] = idx < TILE ? (dz * IH + hy) * RS + col : -1;
  }
  const int lbase = lk * CS + (wid * 2) * RS + lj;

  f32x4 acc[NCLS][2];
#pragma unroll
  for (int c = 0; c < NCLS; ++c) { acc[c][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[c][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  const int ci_w = n0 + lj;
  // register prefetch of the next 4-channel chunk (weights + dy halo tile) behind this chunk's MFMAs: with 54 MFMAs per
  // chunk the two global round trips per chunk otherwise dominate the kernel
  float wn[TAPS], sr[4][E];
  auto fetch = [&](int c0) {
    const int co = c0 + lk;
    const bool ok = ci_w < a.Cin && co < a.Cout;
    const float* __restrict__ wp = a.w + ((size_t)(ok ? co : 0) * a.Cin + (ok ? ci_w : 0)) * TAPS;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) wn[t] = wp[t];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // channels past Cout re-read the last one: their weights are zero
      const __amdgpu_buffer_rsrc_t r = dpi_buffer_t(dpi_at(a.dy, (size_t)min(c0 + c, a.Cout - 1) * Vo, a.dyb), Vo, a.dyb);
      if (a.dyb) {
#pragma unroll
        for (int e = 0; e < E; ++e) sr[c][e] = dpi_buffer_load_bf16_raw(r, goff[e]);       // raw bits, widened at the LDS store
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e) sr[c][e] = dpi_buffer_load(r, goff[e] * 4);          // outside the tile / volume -> 0
      }
    }
  };
  fetch(0);
  for (int c0 = 0; c0 < a.Cout; c0 += 4) {
    float wr[TAPS];
    {
      const bool ok = ci_w < a.Cin && c0 + lk < a.Cout;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) wr[t] = ok ? wn[t] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < E; ++e)
        if ((e + 1) * 256 <= TILE || loff[e] >= 0) lds[c * CS + loff[e]] = a.dyb ? dpi_widen_raw(sr[c][e]) : sr[c][e];
    __syncthreads();
    if (c0 + 4 < a.Cout) fetch(c0 + 4);
    // every (shift_d, shift_h, shift_w, row) sample of the tile feeds the classes whose parity allows that shift
#pragma unroll
    for (int sd = 0; sd < ID; ++sd)
#pragma unroll
      for (int ir = 0; ir < 3; ++ir)                       // dy row offset inside this wave's 2-row band (+1 halo)
#pragma unroll
        for (int sw = 0; sw < 2; ++sw) {
          const float b = lds[lbase + (sd * IH + ir) * RS + sw];
#pragma unroll
          for (int sh = 0; sh < 2; ++sh) {
            const int row = ir - sh;                         // m-row (0/1) that reads dy row `ir` with shift sh
            if (row < 0 || row > 1) continue;
#pragma unroll
            for (int cls = 0; cls < NCLS; ++cls) {
              const int pw = cls & 1, ph = (cls >> 1) & 1, pd = KD == 3 ? (cls >> 2) & 1 : 0;
              if ((sw && !pw) || (sh && !ph) || (sd && !pd)) continue;   // even parity has no shifted tap
              const int kw = pw ? (sw ? 0 : 2) : 1, kh = ph ? (sh ? 0 : 2) : 1, kd = KD == 3 ? (pd ? (sd ? 0 : 2) : 1) : 0;
              acc[cls][row] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[(kd * 3 + kh) * 3 + kw], b, acc[cls][row], 0, 0, 0);
            }
          }
        }
  }
  // ---- store: D row = ci (4*lk + r), D col = m column lj ---------------------------------------------------------------
  // Round 6: the two column-parity classes of a position are NEIGHBOURS in dx (iw = 2 m and 2 m + 1): stored (and, with `accumulate`, read) as
  // one float2 per lane — 16 lanes cover a whole 128-byte line instead of every second float of it twice — with all reads of a row issued
  // before the first store.  The gradient fan-in of an encoder output (ops.FanIn) runs through this read-modify-write.  Rows of odd length (or
  // bf16 storage) keep the scalar form.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const bool vec2 = !a.dxb && ((a.W | (int)(V & 1)) & 1) == 0 && (((uintptr_t)a.dx) & 7) == 0;      // wave-uniform
  if (vec2) {
#pragma unroll
    for (int cp = 0; cp < NCLS / 2; ++cp) {
      const int ph = cp & 1, pd = KD == 3 ? (cp >> 1) & 1 : 0;
#pragma unroll
      for (int row = 0; row < 2; ++row) {
        const int id = KD == 3 ? 2 * md0 + pd : md0, ih = 2 * (mh0 + wid * 2 + row) + ph, iw = 2 * (mw0 + lj);
        if (id < a.D && ih < a.H && iw < a.W) {
          f32x2 old[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ci = n0 + 4 * lk + r;
            const size_t o = (size_t)min(ci, a.Cin - 1) * V + ((size_t)id * a.H + ih) * a.W + iw;
            old[r] = a.accumulate ? *reinterpret_cast<const f32x2*>(a.dx + o) : (f32x2){0.f, 0.f};
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ci = n0 + 4 * lk + r;
            if (ci < a.Cin) {
              const size_t o = (size_t)ci * V + ((size_t)id * a.H + ih) * a.W + iw;
              const f32x2 v = a.accumulate ? (f32x2){old[r][0] + acc[2 * cp][row][r], old[r][1] + acc[2 * cp + 1][row][r]}
                                           : (f32x2){acc[2 * cp][row][r], acc[2 * cp + 1][row][r]};
              *reinterpret_cast<f32x2*>(a.dx + o) = v;
            }
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int cls = 0; cls < NCLS; ++cls) {
    const int pw = cls & 1, ph = (cls >> 1) & 1, pd = KD == 3 ? (cls >> 2) & 1 : 0;
#pragma unroll
    for (int row = 0; row < 2; ++row) {
      const int id = KD == 3 ? 2 * md0 + pd : md0, ih = 2 * (mh0 + wid * 2 + row) + ph, iw = 2 * (mw0 + lj) + pw;
      if (id < a.D && ih < a.H && iw < a.W) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = n0 + 4 * lk + r;
          if (ci < a.Cin) {
            const size_t o = (size_t)ci * V + ((size_t)id * a.H + ih) * a.W + iw;
            dpi_st(a.dx, o, a.accumulate ? dpi_ld(a.dx, o, a.dxb) + acc[cls][row][r] : acc[cls][row][r], a.dxb);
          }
        }
      }
    }
  }
}

// ---- split-K reduction: y[co][v] = (y[co][v] +) bias[co] + sum_s ws[s][co][v], fixed order; BatchNorm statistics of y ------------
// grid (nblk, ceil(Cout / 4)), one wave per channel; block b owns voxels [b * per, (b + 1) * per) and writes the stat partial
// (b, co) — any partition into nblk blocks serves dpi_bn_finalize, which only sums over blocks.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int nsplit, const float* __restrict__ bias,
                                                            float* __restrict__ y, int accumulate, double* __restrict__ partials,
                                                            int Cout, size_t Vo, size_t per, bool yb) {
  const int lane = threadIdx.x & 63, co = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (co >= Cout) return;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < Vo ? lo + per : Vo;
  const float bv = bias ? bias[co] : 0.f;
  double sum = 0.0, sq = 0.0;
  for (size_t v = lo + lane; v < hi; v += 64) {
    float t = ws[(size_t)co * Vo + v];
    for (int sp = 1; sp < nsplit; ++sp) t += ws[((size_t)sp * Cout + co) * Vo + v];
    t += bv;
    if (accumulate) t += dpi_ld(y, (size_t)co * Vo + v, yb);
    t = dpi_stored(t, yb);
    dpi_st(y, (size_t)co * Vo + v, t, yb);
    sum += t; sq += (double)t * t;
  }
  if (partials) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); sq += __shfl_xor(sq, o, 64); }
    if (lane == 0) { partials[((size_t)blockIdx.x * Cout + co) * 2] = sum; partials[((size_t)blockIdx.x * Cout + co) * 2 + 1] = sq; }
  }
}

}  // namespace

// ---- host-side entry points used by the dispatchers in conv_direct.hip / conv_bwd_weight.hip ---------------------------
// variant selection: big tiles while they still give >= 512 workgroups, else the small-tile kernels; stride 2 always
// uses the small tiles (its halo tile is 4x larger)
void dpi_mfma_variant(const dpi_conv_desc* d, int cout, int* nr, int* nh) {
  int Do, Ho, Wo;
  dpi_conv_out_dims(d, &Do, &Ho, &Wo);
  const int tz = d->kd == 3 ? 4 : 1, ty = d->kd == 3 ? 8 : 32;
  const long nb = (long)cdiv(Do, tz) * cdiv(Ho, ty) * cdiv(Wo, 32) * cdiv(cout, 16);
  if (nb >= 512 && d->stride == 1) { *nr = 8; *nh = 2; }
  else { *nr = 2; *nh = Wo > 16 ? 2 : 1; }
}

// Half-height tiles (4 x 4 x 32, ~115-140 VGPRs, 3-4 waves / SIMD) instead of the persistent 4 x 8 x 32 variant: measured
// 3-8 % faster for backward-data of long channel loops and, with the tap-packed tail, for forward layers with 4m + 1 input
// channels (25 -> 16: 108.9 -> 111.9 TFLOP/s); the big tile stays ahead for the other forward layers (51 -> 32).
bool dpi_mfma_half_tile(const dpi_conv_desc* d, bool flip) {
  const int cin = flip ? d->Cout : d->Cin, cout = flip ? d->Cin : d->Cout;
  int nr, nh;
  dpi_mfma_variant(d, cout, &nr, &nh);
  if (nr != 8 || d->kd != 3) return false;
  // ONE input channel (backward-data of the 25 -> 1 output layer): the tap-packed tail is the whole layer, 7 MFMAs per 16 x 16
  // output block instead of 27 with three of four K slices empty (0.31 -> ms measured below)
  if (cin == 1) return true;
  if (cin <= 8) return false;
  return flip || ((cin & 3) == 1);
}

int dpi_mfma_tiles(const dpi_conv_desc* d, int nr, int nh, int* ntd, int* nth, int* ntw) {
  int Do, Ho, Wo;
  dpi_conv_out_dims(d, &Do, &Ho, &Wo);
  const bool slices = d->kd == 3 && nr >= 4;
  const int tz = slices ? 4 : 1, ty = slices ? nr : 4 * nr;
  *ntd = cdiv(Do, tz); *nth = cdiv(Ho, ty); *ntw = cdiv(Wo, 16 * nh);
  return *ntd * *nth * *ntw;
}

template <int KD, bool FLIP, bool IOB = false>
static void launch_variant(const MArgs& a_in, int nr, int nh, int stride, dim3 grid_in, hipStream_t st) {
  if constexpr (!IOB) {
    if (a_in.xb || a_in.yb) { launch_variant<KD, FLIP, true>(a_in, nr, nh, stride, grid_in, st); return; }
  }
  // one-tile-per-workgroup variants: channel tiles of a spatial tile back to back (MArgs::ny); the persistent variants keep the 2-D grid
  const bool persistent = stride == 1 && nr == 8 && a_in.Cin > 8;
  MArgs a = a_in;
  dim3 grid = grid_in;
  if (!persistent && grid_in.y > 1) {
    a.ny = (int)grid_in.y;
    grid = dim3(8 * ((grid_in.x + 7) / 8) * grid_in.y, 1, grid_in.z);
  }
  if (stride == 2) {
    if constexpr (!FLIP) {
      if constexpr (!IOB) {
        // vectorised staging into even / odd column planes: rows of whole aligned float4 pieces (W % 4 == 0, 16-byte aligned channel planes)
        static const bool s2v = getenv("DPI_NO_S2V") == nullptr;
        if (s2v && a.W % 4 == 0 && (((size_t)a.D * a.H * a.W) & 3) == 0 && ((uintptr_t)a.x & 15) == 0) {
          if (nh == 2) conv_mfma_kernel<KD, 2, 2, false, 2, 2, false, false, false, false, true><<<grid, 256, 0, st>>>(a);
          else conv_mfma_kernel<KD, 2, 1, false, 2, 2, false, false, false, false, true><<<grid, 256, 0, st>>>(a);
          return;
        }
      }
      if (nh == 2) conv_mfma_kernel<KD, 2, 2, false, 2, 2, false, false, IOB><<<grid, 256, 0, st>>>(a);
      else conv_mfma_kernel<KD, 2, 1, false, 2, 2, false, false, IOB><<<grid, 256, 0, st>>>(a);
    }
    return;
  }
  const bool tailpack = (a.Cin & 3) == 1 && (a.Cin > 4 || a.Cin == 1);     // instantiated separately: the packed tail costs ~15 VGPRs
  if (nr == 4) {
    if constexpr (KD == 3) {
      if (tailpack) conv_mfma_kernel<KD, 4, 2, FLIP, 1, 3, false, true, IOB><<<grid, 256, 0, st>>>(a);
      else conv_mfma_kernel<KD, 4, 2, FLIP, 1, 3, false, false, IOB><<<grid, 256, 0, st>>>(a);
    }
    return;
  }
  if (nr == 8) {
    // persistent workgroups: 2 per CU, a multiple of 8 so that a workgroup's
    // tiles stay on its XCD; fewer tiles than that -> one tile per workgroup as before
    static const bool persist = getenv("DPI_NO_PERSIST") == nullptr;
    dim3 pg = grid;
    // OPT-IN (DPI_YLOOP=1).  Measured (round 5, tools/bench_conv.py --reps 200, 67 -> 4 backward-data at 256x128x128): within the 168 registers of
    // the occupancy-3 variants the tile loop spills (0.756 -> 2.39 ms); at 256 registers / two workgroups per CU it is 0.755 -> 0.710 ms
    // alone (0.962 -> 0.929 with gradient fan-in), and inside the iteration — with the fused 1x1x1 term, next to the weight-gradient
    // streams — the family gets SLOWER (6.22 -> 6.31 ms, iteration 29.5-30.1 either way: profiles/r05/ab_yloop.txt): the third workgroup per
    // CU hides more of the one-chunk tile's prologue and epilogue than sharing the staged chunk saves.
    static const bool yloop = getenv("DPI_YLOOP") != nullptr && getenv("DPI_YLOOP")[0] == '1';
    if constexpr (KD == 3 && FLIP && !IOB) {
      // one staged chunk, several output-channel tiles (backward-data of 67 -> 4 with its 1x1x1 sibling): the workgroup walks the tiles
      if (yloop && a.Cin <= 4 && a.ny > 1 && a.split_cps == 0 && grid_in.z == 1) {
        conv_mfma_kernel<KD, 8, 2, FLIP, 1, 2, false, false, IOB, true><<<dim3(8 * ((grid_in.x + 7) / 8), 1, 1), 256, 0, st>>>(a);      // (256 registers: at the 168 of the occupancy-3 variants the tile loop spills — 3 x slower)
        return;
      }
    }
    if (a.Cin <= 8) conv_mfma_kernel<KD, 8, 2, FLIP, 1, 3, false, false, IOB><<<grid, 256, 0, st>>>(a);   // occupancy 3 beats persistence here (measured)
    else {
      if (persist && grid.x > 512) pg.x = 512;        // 2 workgroups per CU (256 VGPRs)
      if (tailpack) conv_mfma_kernel<KD, 8, 2, FLIP, 1, 2, true, true, IOB><<<pg, 256, 0, st>>>(a);
      else conv_mfma_kernel<KD, 8, 2, FLIP, 1, 2, true, false, IOB><<<pg, 256, 0, st>>>(a);
    }
  }
  else if (nh == 2) conv_mfma_kernel<KD, 2, 2, FLIP, 1, 2, false, false, IOB><<<grid, 256, 0, st>>>(a);
  else conv_mfma_kernel<KD, 2, 1, FLIP, 1, 2, false, false, IOB><<<grid, 256, 0, st>>>(a);
}

#ifdef DPI_TRACE
extern "C" int dpi_debug_read_blocks(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk), sizeof(long long) * 8192 * 4); }
extern "C" int dpi_debug_read_trace(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(long long) * 4 * 64); }
#endif
// Input-channel split of the small-tile variants (the coarse levels of the U-Net): a launch of 64 tiles x 3 channel tiles whose
// workgroups each walk 139 four-channel chunks (554 -> 35 at 32x16x16) leaves a quarter of the CUs idle for 290 us.  Split the
// channel loop over blockIdx.z into partial outputs (workspace) and sum them in fixed order: nsplit x as many workgroups, each
// with 1 / nsplit of the serial chain.  Returns the number of splits (1: none).
static int g_splitk = getenv("DPI_SPLITK") ? atoi(getenv("DPI_SPLITK")) : 1;
extern "C" void dpi_set_splitk(int on) { g_splitk = on; }
int dpi_mfma_splitk(const dpi_conv_desc* d, bool flip) {
  if (!g_splitk || d->k != 3) return 1;
  const int cin = flip ? d->Cout : d->Cin, cout = flip ? d->Cin : d->Cout;
  int nr, nh, a, b, c;
  dpi_mfma_variant(d, cout, &nr, &nh);
  if (nr != 2 || dpi_mfma_half_tile(d, flip)) return 1;           // the persistent / tap-packed big-tile variants fill the chip
  const long wgs = (long)dpi_mfma_tiles(d, nr, nh, &a, &b, &c) * cdiv(cout, 16);
  const int chunks = cdiv(cin, 4);
  // measured on the coarse-level shapes of the default net (tools/bench_conv.py, DPI_SPLITK=0 / 1): the split pays when the launch
  // is about one workgroup per CU or less and every split keeps >= 8 chunks; 2-3 splits of a 13-18 chunk loop lose 5-15 % to the
  // extra reduction launch
  if (wgs > 512 || (wgs > 256 && chunks < 24)) return 1;
  long ns = 1536 / wgs;
  if (ns > chunks / 8) ns = chunks / 8;
  if (ns > 8) ns = 8;
  return ns < 2 ? 1 : (int)ns;
}
size_t dpi_conv_mfma_ws_floats(const dpi_conv_desc* d, bool flip) {
  const int ns = dpi_mfma_splitk(d, flip);
  if (ns < 2) return 0;
  int Do, Ho, Wo;
  dpi_conv_out_dims(d, &Do, &Ho, &Wo);
  return (size_t)ns * (flip ? d->Cin : d->Cout) * Do * Ho * Wo;
}

// whether dpi_conv_mfma_run can add a 1x1x1 second input of C2 channels in the same pass (MfmaSecond)
bool dpi_conv_mfma_second_ok(const dpi_conv_desc* d, bool flip, int C2, bool have_ws) {
  int Do, Ho, Wo;
  dpi_conv_out_dims(d, &Do, &Ho, &Wo);
  if (!flip || d->stride != 1 || C2 < 1 || ((size_t)C2 + 4) * Do * Ho * Wo * sizeof(float) >= ((size_t)1 << 31)) return false;
  int nr, nh;
  dpi_mfma_variant(d, d->Cin, &nr, &nh);
  if (nr == 8 && !dpi_mfma_half_tile(d, flip) && d->Cout > 8) return false;   // the persistent variant is compiled without it
  return !(have_ws && dpi_mfma_splitk(d, flip) > 1);        // the split launches write partial outputs: the pair stays two launches there
}

int dpi_conv_mfma_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias, float* y,
                      double* partials, bool flip, int accumulate, float* ws, size_t ws_floats, hipStream_t st, const MfmaSecond* sec) {
  const int taps = d->kd * 9;
  const int cin = flip ? d->Cout : d->Cin, cout = flip ? d->Cin : d->Cout;
  const long w_out = flip ? taps : (long)d->Cin * taps, w_in = flip ? (long)d->Cin * taps : taps;
  MArgs a{x, chain, w, bias, y, partials, cin, cout, d->D, d->H, d->W, 0, 0, 0, w_out, w_in, accumulate, 0, nullptr, nullptr, 0, 0, 0,
          dpi_io_in(d, flip), dpi_io_out(d, flip)};
  int nr, nh;
  dpi_mfma_variant(d, cout, &nr, &nh);
  if (dpi_mfma_half_tile(d, flip)) nr = 4;
  const int ntiles = dpi_mfma_tiles(d, nr, nh, &a.ntd, &a.nth, &a.ntw);
  dim3 grid(ntiles, cdiv(cout, 16));
  if (sec) { a.x2 = sec->x2; a.w2 = sec->w2; a.C2 = sec->C2; a.w2_co_stride = sec->w2_co_stride; a.w2_c_stride = sec->w2_c_stride; }
  const int nsplit = (ws && !sec) ? dpi_mfma_splitk(d, flip) : 1;
  if (nsplit > 1) {
    int Do, Ho, Wo;
    dpi_conv_out_dims(d, &Do, &Ho, &Wo);
    const size_t Vo = (size_t)Do * Ho * Wo;
    DPI_REQUIRE(ws_floats >= (size_t)nsplit * cout * Vo, "conv (split): workspace of %zu floats, need %zu", ws_floats, (size_t)nsplit * cout * Vo);
    a.split_cps = 4 * cdiv(cdiv(cin, 4), nsplit);
    a.y = ws; a.bias = nullptr; a.partials = nullptr; a.accumulate = 0; a.yb = 0;
    grid.z = cdiv(cin, a.split_cps);                                // <= nsplit; every column owns at least one chunk
    const int nz = (int)grid.z;
    if (d->kd == 3) { if (flip) launch_variant<3, true>(a, nr, nh, d->stride, grid, st); else launch_variant<3, false>(a, nr, nh, d->stride, grid, st); }
    else { if (flip) launch_variant<1, true>(a, nr, nh, d->stride, grid, st); else launch_variant<1, false>(a, nr, nh, d->stride, grid, st); }
    if (int e = dpi_check_launch("conv_mfma (split)")) return e;
    splitk_reduce_kernel<<<dim3(ntiles, cdiv(cout, 4)), 256, 0, st>>>(ws, nz, bias, y, accumulate, partials, cout, Vo, cdivz(Vo, (size_t)ntiles),
                                                                      dpi_io_out(d, flip));
    return dpi_check_launch("splitk_reduce");
  }
  if (d->kd == 3) { if (flip) launch_variant<3, true>(a, nr, nh, d->stride, grid, st); else launch_variant<3, false>(a, nr, nh, d->stride, grid, st); }
  else { if (flip) launch_variant<1, true>(a, nr, nh, d->stride, grid, st); else launch_variant<1, false>(a, nr, nh, d->stride, grid, st); }
  return dpi_check_launch("conv_mfma");
}

// tile utilisation of the two orientations (rows padded to 16, staged channels to 4)
static bool mfma_bw_swap_better(const dpi_conv_desc* d) {
  if (d->stride != 1) return false;
  auto util = [](int rows, int staged) { return (double)rows / (16.0 * cdiv(rows, 16)) * staged / (4.0 * cdiv(staged, 4)); };
  return util(d->Cin, d->Cout) > 1.15 * util(d->Cout, d->Cin);
}

static int g_bw_want = getenv("DPI_BW_WANT") ? atoi(getenv("DPI_BW_WANT")) : 2304;        // tuning knobs (dpi_set_bw_tuning): workgroups aimed at, XCD-aware workgroup order on/off
// XCD-aware (chunk, group) order: OFF by default.  Isolated launches gain 4-6 % (25 -> 16: 1.187 -> 1.136 ms), but inside the
// iteration (weight gradients on the side stream next to the backward-data / BatchNorm chain) it LOSES: 37.2-37.4 vs 36.6-36.8 ms
// per iteration in an A/B of four bench runs (round 2).  DPI_BW_XCD_ORDER=1 / dpi_set_bw_tuning switch it on for experiments.
static int g_bw_xcd_order = getenv("DPI_BW_XCD_ORDER") ? atoi(getenv("DPI_BW_XCD_ORDER")) : 0;
extern "C" void dpi_set_bw_tuning(int want_workgroups, int xcd_order) {
  if (want_workgroups > 0) g_bw_want = want_workgroups;
  if (xcd_order >= 0) g_bw_xcd_order = xcd_order;
}
// occupancy experiments: DPI_BW_EXTRA_LDS=<KiB> of unused dynamic LDS per workgroup (44 KB static: 3 workgroups per CU by default)
static size_t bw_extra_lds() { static const size_t v = getenv("DPI_BW_EXTRA_LDS") ? (size_t)atoi(getenv("DPI_BW_EXTRA_LDS")) * 1024 : 0; return v; }
struct MfmaBwPlan { int nchunks, tiles_per_chunk, ntiles, ntd, nth, ntw, nr, nh, npairs, nfull, ny; };
// two 4-channel groups per workgroup (conv_bwd_weight_mfma_pair_kernel): 0 off, 1 every layer with >= 2 full groups, 2 (default) only
// group loops of >= 4 full groups at >= 1024 tiles (25->16, 17->26, 51->32 of the default net).  Every layer is faster in isolation
// with it, but inside the iteration the weight gradients share the chip with the backward-data chain on another stream and a 79 KB
// workgroup leaves that chain less room: six alternating bench runs each gave 32.95 (off) / 33.20 (every layer) / 32.84 ms (restricted).
static int g_bw_pair = getenv("DPI_BW_PAIR") ? atoi(getenv("DPI_BW_PAIR")) : 2;
extern "C" void dpi_set_bw_pair(int mode) { if (mode >= 0 && mode <= 2) g_bw_pair = mode; }
static MfmaBwPlan mfma_bw_plan(const dpi_conv_desc* d, bool swap = false) {
  MfmaBwPlan p{};
  {
    const int staged = swap ? d->Cout : d->Cin, groups = cdiv(staged, 4);
    const bool s1k3 = d->stride == 1 && d->kd == 3;
    const int tailg = (s1k3 && (staged & 3) == 1) ? 1 : 0;
    p.nfull = groups - tailg;
    p.npairs = (g_bw_pair && s1k3 && !g_bw_xcd_order && p.nfull >= 2 && d->io == 0) ? p.nfull / 2 : 0;     // (bf16 tensors: the plain kernels only)
    if (g_bw_pair == 2) {                                      // restricted: long group loops at the two finest levels only
      int a_, b_, c_;
      if (p.nfull < 4 || dpi_mfma_tiles(d, 8, 2, &a_, &b_, &c_) < 1024) p.npairs = 0;
    }
    p.ny = p.npairs ? p.npairs + (p.nfull & 1) + tailg : groups;
  }
  if (d->stride == 1) { p.nr = 8; p.nh = 2; }
  else { int Do, Ho, Wo; dpi_conv_out_dims(d, &Do, &Ho, &Wo); p.nr = 2; p.nh = Wo > 16 ? 2 : 1; }
  p.ntiles = dpi_mfma_tiles(d, p.nr, p.nh, &p.ntd, &p.nth, &p.ntw);
  const size_t per = (size_t)d->Cout * d->Cin * d->kd * 9;
  const size_t max_chunks_mem = per ? ((size_t)32 << 20) / per : 1;
  const size_t blocks_other = (size_t)p.ny * (swap ? cdiv(d->Cin, 16) : cdiv(d->Cout, 16));
  // ~2304 workgroups (3 per CU x 256 CUs x 3) measured best on the full-resolution layers.  (A model that picks the chunk
  // length by whole occupancy rounds of the main launch was measured 14-17 % SLOWER on every layer, round 2: the kernel is
  // not round-quantised — its limiter is the dY / X re-read traffic, see the XCD-aware workgroup order in the kernel.)
  size_t want = (size_t)g_bw_want / blocks_other;
  if (want < 1) want = 1;
  if (want > (size_t)p.ntiles) want = p.ntiles;
  if (want > max_chunks_mem) want = max_chunks_mem;
  if (want < 1) want = 1;
  if (g_bw_xcd_order && want > 8) want = want / 8 * 8;        // whole groups of 8 chunks (one per XCD)
  p.tiles_per_chunk = (int)cdivz(p.ntiles, want);
  p.nchunks = cdiv(p.ntiles, p.tiles_per_chunk);
  return p;
}

size_t dpi_conv_bwd_weight_mfma_ws_floats(const dpi_conv_desc* d) {
  const int n0 = mfma_bw_plan(d, false).nchunks, n1 = mfma_bw_swap_better(d) ? mfma_bw_plan(d, true).nchunks : 0;
  return (size_t)(n0 > n1 ? n0 : n1) * d->Cout * d->Cin * d->kd * 9;     // either orientation (the chain decides at run time)
}
bool dpi_conv_bwd_weight_mfma_swapped(const dpi_conv_desc* d, const float* chain) { return chain == nullptr && mfma_bw_swap_better(d); }

// dyb class of a bf16 row operand: 1 when every 4-element piece starts 4-byte aligned (even row length and channel size, aligned base)
static int bf16_row_class(const void* base, int row_len, size_t channel_elems) {
  return ((row_len & 1) == 0 && (channel_elems & 1) == 0 && ((uintptr_t)base & 3) == 0) ? 1 : 2;
}

int dpi_conv_bwd_weight_mfma_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* dy, float* dw, float* ws,
                                 hipStream_t st) {
  const bool swap = dpi_conv_bwd_weight_mfma_swapped(d, chain);     // the chain can only be applied to the staged tensor
  const MfmaBwPlan p = mfma_bw_plan(d, swap);
  BwMArgs a{x, chain, dy, ws, d->Cin, d->Cout, d->D, d->H, d->W, p.ntd, p.nth, p.ntw, p.ntiles, p.tiles_per_chunk, 0, 0, 0, p.nchunks,
            (d->io & DPI_IO_X_BF16) != 0, (d->io & DPI_IO_DY_BF16) != 0};
  dim3 grid(p.nchunks, cdiv(d->Cin, 4), cdiv(d->Cout, 16));
  if (swap) {
    a.x = dy; a.chain = nullptr; a.dy = x; a.Cin = d->Cout; a.Cout = d->Cin; a.swap = 1;
    a.xb = (d->io & DPI_IO_DY_BF16) != 0; a.dyb = (d->io & DPI_IO_X_BF16) != 0;
    grid = dim3(p.nchunks, cdiv(d->Cout, 4), cdiv(d->Cin, 16));
  }
  if (a.dyb) {        // the 16-row operand is read in 4-element pieces of its rows (output-sized without a swap, input-sized — stride 1 — with it)
    int Do_, Ho_, Wo_;
    dpi_conv_out_dims(d, &Do_, &Ho_, &Wo_);
    a.dyb = swap ? bf16_row_class(a.dy, d->W, (size_t)d->D * d->H * d->W) : bf16_row_class(a.dy, Wo_, (size_t)Do_ * Ho_ * Wo_);
  }
  // (chunk, group) grid -> 1-D XCD-aware order (see BwMArgs::ngroups)
  auto xcd_grid = [&](dim3 g) {
    if (!g_bw_xcd_order) { a.ngroups = 0; return g; }
    a.ngroups = (int)g.y;
    return dim3(8u * g.y * (unsigned)cdiv((int)g.x, 8), 1, g.z);
  };
  if (a.xb || a.dyb) {
    // bf16 tensors: the IOB instantiations of the plain kernel (the full groups and, for a staged count of 4m + 1, the column-trimmed tail
    // group as a second launch).  In bf16 precision mode this path serves the stride-2 layers and rows that are not whole float4
    // (conv_bf16_bww.hip takes the rest).
    a.ngroups = 0;
    if (d->stride == 1 && d->kd == 3 && (a.Cin & 3) == 1) {
      if (grid.y > 1) conv_bwd_weight_mfma_kernel<3, 1, 8, 2, 0, true><<<dim3(grid.x, grid.y - 1, grid.z), 256, 0, st>>>(a);
      a.y0 = (int)grid.y - 1;
      conv_bwd_weight_mfma_kernel<3, 1, 8, 2, 1, true><<<dim3(grid.x, 1, grid.z), 256, 0, st>>>(a);
    } else if (d->stride == 1) {
      if (d->kd == 3) conv_bwd_weight_mfma_kernel<3, 1, 8, 2, 0, true><<<grid, 256, 0, st>>>(a);
      else conv_bwd_weight_mfma_kernel<1, 1, 8, 2, 0, true><<<grid, 256, 0, st>>>(a);
    } else if (p.nh == 2) {
      if (d->kd == 3) conv_bwd_weight_mfma_kernel<3, 2, 2, 2, 0, true><<<grid, 256, 0, st>>>(a);
      else conv_bwd_weight_mfma_kernel<1, 2, 2, 2, 0, true><<<grid, 256, 0, st>>>(a);
    } else {
      if (d->kd == 3) conv_bwd_weight_mfma_kernel<3, 2, 2, 1, 0, true><<<grid, 256, 0, st>>>(a);
      else conv_bwd_weight_mfma_kernel<1, 2, 2, 1, 0, true><<<grid, 256, 0, st>>>(a);
    }
  } else if (p.npairs > 0) {
    a.ngroups = 0;
    conv_bwd_weight_mfma_pair_kernel<3, 1, 8, 2><<<dim3(grid.x, p.ny, grid.z), 256, 0, st>>>(a, p.npairs, p.nfull);
  } else if (d->stride == 1 && d->kd == 3 && (a.Cin & 3) == 1) {
    // staged channel count 4m + 1: full groups in one launch, the one-channel group in a second, column-trimmed one (a single staged
    // channel — the swapped 25 -> 1 output layer — is that second launch alone: 2 column tiles instead of 7)
    static const bool merged = getenv("DPI_BW_TWO_LAUNCHES") == nullptr;
    if (merged && grid.y > 1 && !g_bw_xcd_order) {
      a.ngroups = 0;
      conv_bwd_weight_mfma_merged_kernel<3, 1, 8, 2><<<grid, 256, bw_extra_lds(), st>>>(a);
    } else {
      dim3 gmain(grid.x, grid.y - 1, grid.z), gtail(grid.x, 1, grid.z);
      if (grid.y > 1) {
        dim3 gm = xcd_grid(gmain);
        conv_bwd_weight_mfma_kernel<3, 1, 8, 2><<<gm, 256, bw_extra_lds(), st>>>(a);
      }
      a.y0 = (int)grid.y - 1;
      dim3 gt = xcd_grid(gtail);
      conv_bwd_weight_mfma_kernel<3, 1, 8, 2, 1><<<gt, 256, 0, st>>>(a);
    }
  } else if (d->stride == 1) {
    dim3 g = xcd_grid(grid);
    if (d->kd == 3) conv_bwd_weight_mfma_kernel<3, 1, 8, 2><<<g, 256, bw_extra_lds(), st>>>(a);
    else conv_bwd_weight_mfma_kernel<1, 1, 8, 2><<<g, 256, 0, st>>>(a);
  } else if (p.nh == 2) {
    dim3 g = xcd_grid(grid);
    if (d->kd == 3) conv_bwd_weight_mfma_kernel<3, 2, 2, 2><<<g, 256, 0, st>>>(a);
    else conv_bwd_weight_mfma_kernel<1, 2, 2, 2><<<g, 256, 0, st>>>(a);
  } else {
    dim3 g = xcd_grid(grid);
    if (d->kd == 3) conv_bwd_weight_mfma_kernel<3, 2, 2, 1><<<g, 256, 0, st>>>(a);
    else conv_bwd_weight_mfma_kernel<1, 2, 2, 1><<<g, 256, 0, st>>>(a);
  }
  if (int e = dpi_check_launch("conv_bwd_weight_mfma")) return e;
  const size_t per = (size_t)d->Cout * d->Cin * d->kd * 9;
  dpi_reduce_chunks(ws, dw, per, p.nchunks, st);
  return dpi_check_launch("reduce_chunks");
}

int dpi_conv_bwd_data_s2_mfma_run(const dpi_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate, hipStream_t st) {
  int Do, Ho, Wo;
  dpi_conv_out_dims(d, &Do, &Ho, &Wo);
  BdS2Args a{dy, w, dx, d->Cin, d->Cout, d->D, d->H, d->W, Do, Ho, Wo, 0, 0, 0, accumulate, (d->io & DPI_IO_DY_BF16) != 0, (d->io & DPI_IO_DX_BF16) != 0};
  const int md = d->kd == 3 ? cdiv(d->D, 2) : d->D, mh = cdiv(d->H, 2), mw = cdiv(d->W, 2);
  a.ntd = md; a.nth = cdiv(mh, 8); a.ntw = cdiv(mw, 16);
  dim3 grid(a.ntd * a.nth * a.ntw, cdiv(d->Cin, 16));
  if (d->kd == 3) conv_bwd_data_s2_mfma_kernel<3><<<grid, 256, 0, st>>>(a);
  else conv_bwd_data_s2_mfma_kernel<1><<<grid, 256, 0, st>>>(a);
  return dpi_check_launch("conv_bwd_data_s2_mfma");
}

// few-output-channel backward-weight (3-D, stride 1, Cout <= 5)
struct SmallBwPlan { int nchunks, tiles_per_chunk, ntiles, ntd, nth, ntw; };
static SmallBwPlan small_bw_plan(const dpi_conv_desc* d) {
  SmallBwPlan p{};
  p.ntd = cdiv(d->D, 4); p.nth = cdiv(d->H, 8); p.ntw = cdiv(d->W, 32);
  p.ntiles = p.ntd * p.nth * p.ntw;
  const size_t per = (size_t)d->Cout * d->Cin * 27;
  const size_t max_chunks_mem = per ? ((size_t)32 << 20) / per : 1;
  size_t want = cdivz(1024, (size_t)cdiv(d->Cin, 6));
  if (want > (size_t)p.ntiles) want = p.ntiles;
  if (want > max_chunks_mem) want = max_chunks_mem;
  if (want < 1) want = 1;
  p.tiles_per_chunk = (int)cdivz(p.ntiles, want);
  p.nchunks = cdiv(p.ntiles, p.tiles_per_chunk);
  return p;
}
size_t dpi_conv_bwd_weight_smallco_ws_floats(const dpi_conv_desc* d) {
  return (size_t)small_bw_plan(d).nchunks * d->Cout * d->Cin * 27;
}
int dpi_conv_bwd_weight_smallco_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* dy, float* dw, float* ws,
                                    hipStream_t st) {
  const SmallBwPlan p = small_bw_plan(d);
  BwSArgs a{x, chain, dy, ws, d->Cin, d->Cout, d->D, d->H, d->W, p.ntd, p.nth, p.ntw, p.ntiles, p.tiles_per_chunk,
            (d->io & DPI_IO_X_BF16) != 0, (d->io & DPI_IO_DY_BF16) ? bf16_row_class(dy, d->W, (size_t)d->D * d->H * d->W) : 0};
  conv_bwd_weight_smallco_kernel<<<dim3(p.nchunks, cdiv(d->Cin, 6)), 256, 0, st>>>(a);
  if (int e = dpi_check_launch("conv_bwd_weight_smallco")) return e;
  const size_t per = (size_t)d->Cout * d->Cin * 27;
  dpi_reduce_chunks(ws, dw, per, p.nchunks, st);
  return dpi_check_launch("reduce_chunks");
}
