// bf16-MFMA stencil convolution (k = 3, stride 1; forward and backward-data) for gfx950 — the mixed-precision mode of
// BASELINE configs[4] ("bf16 activations + fp32 master weights"): operands are rounded to bf16 (round-to-nearest-even) while
// they are staged into LDS, products are exact in fp32 and accumulate in fp32 (v_mfma_f32_16x16x32_bf16); the activation tensors
// are fp32 or bf16 in HBM (dpi_conv_desc.io, template parameters XB / YB), the master weights, BatchNorm statistics and Adam stay
// fp32.  16x the matrix rate of the fp32 MFMA path (conv_mfma.hip): the MFMA phase is bound by its LDS operand reads, the kernel
// as a whole by the latency of the next channel group's tile at 4 workgroups per CU.
//
// Implicit GEMM per output tile:   D[co 16][vox 16] += A[co 16][K 32] * B[K 32][vox 16]
//     K block = 8 input channels x 4 taps (tap = 4 g + lane>>4, g = 0 .. ceil(TAPS / 4) - 1): a small channel count still fills K,
//     and one staged group of 8 channels serves all 27 taps.
//     B = halo tile in LDS as [position][8 channels] bf16 (16 B per position): lane (vox = l & 15, tap slot = l >> 4) reads ONE
//         ds_read_b128 at position(vox) + offset(tap) — 16 consecutive positions per 16-lane group, conflict-free.
//     A = weights of the 8-channel group as ready-made fragments [g][lane][8] in LDS, copied 16 bytes per thread from the layer's
//         PACKED bf16 copy (conv_bf16_pack_kernel, one small launch in front of every launch of this kernel; the torch layout
//         [Cout][Cin][27] gathered by every tile cost 64 cache-line lookups per load instruction and bound the kernel).
// NS = 3 (precision = 2, "split" mode): every fp32 operand is split EXACTLY into three bf16 terms x = h + m + l (8 + 8 + 8
// significant bits; h = rne(x), m = rne(x - h), l = rne(x - h - m), each difference exact in fp32) and six of the nine partial
// products — hh, hm, mh, mm, hl, lh, i.e. all those >= 2^-16 of the full product — are accumulated in fp32.  The three dropped
// ones are <= 2^-24 relative each, the size of one fp32 rounding, so the result carries fp32-class accuracy (verified against
// the fp64 oracle at the fp32 path's own tolerance) at 16 / 6 = 2.7 x the fp32 matrix rate.
// Workgroup = 4 waves = output tile 4x4x32 (or the small 1x8x16/32 variants for coarse levels), as in conv_mfma.hip; D layout,
// epilogue (bias, BatchNorm {sum, sum^2} partials, gradient fan-in) are those of the fp32 kernel.
#include "common.h"

#include <climits>
#include <iterator>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

void dpi_conv_out_dims(const dpi_conv_desc* d, int* Do, int* Ho, int* Wo);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct BArgs {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ w;
  const float* __restrict__ bias;
  float* __restrict__ y;
  double* __restrict__ partials;
  int Cin, Cout;
  int D, H, W;
  int ntd, nth, ntw;
  int ny;          // output-channel tiles of the launch (workgroups per spatial tile)
  long w_out_stride, w_in_stride;
  const unsigned short* __restrict__ wpk;      // the weights as bf16 A fragments (conv_bf16_pack_kernel), written before this launch
  int accumulate;
  int debug;     // tuning experiments only (dpi_set_bf16_debug): 1 skip the MFMA phase, 2 skip the global loads, 4 skip the LDS stores, 8 cache-hot loads
  int xb, yb;    // storage type of x / y in HBM: 1 = bf16 (dpi_conv_desc.io), 0 = fp32
  // second input through a 1x1x1 kernel at the OUTPUT positions (x2 != nullptr; backward-data launches only): y += W2 * x2 with
  // x2 [C2][D][H][W] of x's storage type and W2[co][c] = w2[co * w2_co_stride + c * w2_c_stride] — the input gradient of a 3x3x3 layer and the
  // 1x1x1 layer beside it (Block3d.conv1 + shortcut, ResPath3d.conv3x3 + conv1x1, mulresunet.py:72-113) in ONE pass over dx, as MfmaSecond
  // does for the fp32 kernel.  Without it the pair is two launches: the 1x1x1 one writes dx, this one reads it back and adds (67->4 + 67->25
  // at 256x128x128 with bf16 tensors: 0.51 + 1.14 ms, of which 0.73 ms are that read-back; fused: one launch).
  const float* __restrict__ x2;
  const float* __restrict__ w2;
  int C2;
  long w2_co_stride, w2_c_stride;
};

// WIDE (bf16 input tensors): the halo tile starts 4 columns left of the output tile and is 8 columns wider than it, so that every row is
// a whole number of ALIGNED 4-element pieces (8 bytes of a bf16 tensor): one load instruction per piece instead of one per element —
// 16 instead of 40 load instructions per thread and 8-channel group, 32 instead of 48 prefetch registers.  (Neutral on time by itself:
// what the phase-skipping runs had attributed to the x loads were the weight gathers, DESIGN §3 / profiles/README.md round 4.)
template <int KD, int NR, int NH, bool WIDE = false>
struct GeoB {
  static constexpr bool SLICES = (KD == 3 && NR >= 4);          // waves split depth; otherwise they split rows
  static constexpr int TZ = SLICES ? 4 : 1;                      // output tile
  static constexpr int TY = SLICES ? NR : 4 * NR;
  static constexpr int TW = 16 * NH;
  static constexpr int ID = TZ - 1 + KD;                         // input (halo) tile, positions stored densely [ID][IH][IW]
  static constexpr int IH = TY + 2;
  static constexpr int IW = WIDE ? TW + 8 : TW + 2;
  static constexpr int COL0 = WIDE ? 4 : 1;                      // tile column of global column ow0
  static constexpr int QW = IW / 4;                              // WIDE: 4-element pieces per row, pieces per tile, per thread
  static constexpr int NQ = ID * IH * QW;
  static constexpr int EQ = (NQ + 255) / 256;
  static constexpr int TILE = ID * IH * IW;
  static constexpr int E = (TILE + 255) / 256;
  static constexpr int TAPS = KD * 9;
  static constexpr int NTG = (TAPS + 3) / 4;                     // tap groups of 4 (7 for 3-D: 27 of 28 K slots used)
  static constexpr int WE = (NTG * 64 * 8 + 255) / 256;          // weight elements each thread converts per channel group
};

// round-to-nearest-even fp32 -> bf16 (the rounding torch.Tensor.bfloat16() applies), two values packed into one dword by ONE
// v_cvt_pk_bf16_f32 (the integer formulation u + 0x7fff + lsb costs 4-5 VALU operations per value, and the staging path of this
// kernel is VALU-bound in split mode)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned bf16_bits(float f) { return pack_bf16(f, 0.f) & 0xffffu; }

// Workgroup -> (spatial tile, output-channel tile).  Workgroups go to the 8 XCDs round-robin by their linear id; XCD x walks a contiguous
// range of tiles (neighbours share halo lines in ITS L2, as in conv_mfma.hip) and runs the ny channel tiles of a spatial tile BACK TO
// BACK: they read the same input tile (and the same second input); with the channel tile as the slow grid dimension they ran a whole
// launch apart and every pass over the input came from HBM again.  The grid is 8 * ceil(ntiles / 8) * ny workgroups; the few past an
// XCD's range leave at once (false).
__device__ __forceinline__ bool xcd_tile_b(int bid, int ntiles, int ny, int& tile, int& ytile) {
  const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, j = bid >> 3;
  const int i = j / ny;
  ytile = j - i * ny;
  tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  return i < q + (xcd < r ? 1 : 0);
}

// exact three-term split of an fp32 value into bf16 bit patterns
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = bf16_bits(x);
  const float r1 = x - __builtin_bit_cast(float, h << 16);
  m = bf16_bits(r1);
  const float r2 = r1 - __builtin_bit_cast(float, m << 16);
  l = bf16_bits(r2);
}
// ... of two values at once, each term already packed (lo | hi << 16): 3 conversions + 4 subtractions + 4 mask / shift operations
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = pack_bf16(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = pack_bf16(r0, r1);
  const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = pack_bf16(s0, s1);
}

// XB / YB: storage type of x / y (true = bf16) as TEMPLATE parameters: as run-time flags they cost the fp32-tensor instantiation 10 % (and the
// storage mode as much), measured in the iteration (profiles/README.md, round 4).
// MT: 16-channel output tiles per workgroup (2 when the layer has more than 16 output channels): every staged tile and every B fragment
// read from LDS — one per MFMA, the busiest unit of the MFMA phase — then feeds MT MFMAs.
template <int KD, int NR, int NH, bool FLIP, int NS = 1, bool XB = false, bool YB = false, int MT = 1>
__global__ __launch_bounds__(256, NS != 1 ? 2 : MT == 2 ? 3 : 4) void conv_bf16_kernel(BArgs a) {
  static_assert(MT == 1 || NS == 1, "two output tiles: bf16 arithmetic mode only");
  a.xb = XB; a.yb = YB;
  using G = GeoB<KD, NR, NH, XB>;
  constexpr int TAPS = G::TAPS, NTG = G::NTG, PD = (KD - 1) / 2, NT = NR * NH;
  constexpr int WT = 128;                                                    // halfwords per tap of the A-fragment buffer: 16 co x 8 ci
  constexpr int XW = G::TILE * 4, WW = NTG * 4 * WT;                        // words / halfwords per operand copy
  __shared__ __attribute__((aligned(16))) unsigned xl[NS * XW];             // [term][position][8 bf16]
  __shared__ __attribute__((aligned(16))) unsigned short wl[MT * NS * WW];  // [tile | term][tap = 4 g + lane / 16][co = lane % 16][8 bf16]: A fragments
  __shared__ double red[4][16 * MT][2];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  int tile_id, ytile;
  if (!xcd_tile_b(blockIdx.x, a.ntd * a.nth * a.ntw, a.ny, tile_id, ytile)) return;
  const int n0 = ytile * 16 * MT;
  const size_t V = (size_t)a.D * a.H * a.W;
  const int Do = a.D, Ho = a.H, Wo = a.W;                      // stride 1, 'same' padding
  const size_t Vo = V;

  int od0, oh0, ow0;
  {
    int bt = tile_id;
    const int tw_i = bt % a.ntw; bt /= a.ntw;
    const int th_i = bt % a.nth; bt /= a.nth;
    od0 = bt * G::TZ; oh0 = th_i * G::TY; ow0 = tw_i * G::TW;
  }
  const int wz = G::SLICES ? wid : 0, wh = G::SLICES ? 0 : wid * NR;

  // halo-tile slots of this thread: position idx = tid + 256 e  <->  (dz, hy, col), and the global voxel offset (or -1 = padding)
  static_assert(!XB || NS == 1, "bf16 tensors: bf16 arithmetic mode only");
  int goff[XB ? 1 : G::E];
  if constexpr (!XB)
#pragma unroll
  for (int e = 0; e < G::E; ++e) {
    const int idx = tid + e * 256;
    const int col = idx % G::IW, row = idx / G::IW;
    const int hy = row % G::IH, dz = row / G::IH;
    const int gd = od0 - PD + dz, gh = oh0 - 1 + hy, gw = ow0 - G::COL0 + col;
    const bool ok = idx < G::TILE && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    goff[e] = ok ? (gd * a.H + gh) * a.W + gw : -1;
  }
  // WIDE staging: this thread's 4-element pieces (piece qi = tid + 256 e <-> (dz, hy, q)): element offset of the piece or -1, and its first
  // tile position.  Needs rows that are whole pieces (W % 4 == 0) and an 8-byte aligned tensor: dpi_conv_bf16_usable.
  int qoff[XB ? G::EQ : 1], qpos[XB ? G::EQ : 1];
  if constexpr (XB) {
#pragma unroll
    for (int e = 0; e < G::EQ; ++e) {
      const int qi = tid + e * 256;
      const int q = qi % G::QW, row = qi / G::QW;
      const int hy = row % G::IH, dz = row / G::IH;
      const int gd = od0 - PD + dz, gh = oh0 - 1 + hy, gw = ow0 - G::COL0 + 4 * q;
      const bool ok = qi < G::NQ && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
      qoff[e] = ok ? (gd * a.H + gh) * a.W + gw : -1;
      if (a.debug & 8) qoff[e] = (qi * 4) & 0x3fff;              // experiment: every tile reads the same 32 KB of each channel (cache hits)
      qpos[e] = (dz * G::IH + hy) * G::IW + 4 * q;
    }
  }
  // per-lane tap offsets (in positions) of the NTG tap groups; K slot lk of group g is tap 4 g + lk (clamped: its weights are 0)
  int toff[NTG];
#pragma unroll
  for (int g = 0; g < NTG; ++g) {
    const int t = min(4 * g + lk, TAPS - 1);
    const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    toff[g] = (kd * G::IH + kh) * G::IW + kw;
  }
  const int pbase = (wz * G::IH + wh) * G::IW + lj + (G::COL0 - 1);   // position of output voxel (row 0, column block 0) at tap (0,0,0)

  float sr[XB ? 1 : 8][XB ? 1 : G::E];                         // next channel group, raw fp32, in flight
  unsigned sq[XB ? 8 : 1][XB ? G::EQ : 1][2];                  // WIDE: the group's pieces, two raw dwords (four bf16) each
  auto load_x = [&](int c0) {
    if constexpr (XB) {   // bf16 tensor: the raw 16-bit halves are kept and transposed when staged
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int ci = min(c0 + c, a.Cin - 1);
        const __amdgpu_buffer_rsrc_t r = dpi_buffer_t(dpi_at(a.x, (size_t)ci * V, true), V, true);
#pragma unroll
        for (int e = 0; e < G::EQ; ++e) {
          const dpi_u32x2v u = __builtin_bit_cast(dpi_u32x2v, __builtin_amdgcn_raw_buffer_load_b64(r, qoff[e] >= 0 ? qoff[e] * 2 : -8, 0, 0));
          sq[c][e][0] = u.x; sq[c][e][1] = u.y;
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int ci = min(c0 + c, a.Cin - 1);                   // channels past Cin: their weights are zero
        const __amdgpu_buffer_rsrc_t r = dpi_buffer(a.x + (size_t)ci * V, V * sizeof(float));
#pragma unroll
        for (int e = 0; e < G::E; ++e) sr[c][e] = dpi_buffer_load(r, goff[e] * 4);
      }
    }
  };
  // Weights of a channel group: ready-made A fragments [term][tap = 4 g + lane / 16][co = lane % 16][8 ci] (conv_bf16_pack_kernel, one
  // tiny launch in front of this one), 16 bytes per thread and load, coalesced, stored as they come.  Gathered from the fp32 weights in
  // fragment order HERE, every lane of an instruction read its own cache line — 64 tag lookups for 256 bytes, 14 times per thread and
  // group — and the L1 was the busiest unit of the CU: 25->16 forward 0.295 -> 0.208 ms with the lookups taken away; reading in memory
  // order instead trades them for bank conflicts of the 2-byte staging writes (0.27 / 0.30 ms) (profiles/README.md, round 4).
  constexpr int WV = MT * NS * WW / 8, WPE = (WV + 255) / 256; // 16-byte pieces of a group's fragments; per thread
  u32x4 wq[WPE];                                               // next channel group's pieces of this thread
  const u32x4* __restrict__ const wsrc = reinterpret_cast<const u32x4*>(a.wpk) + (size_t)ytile * ((a.Cin + 7) >> 3) * WV;
  auto load_w = [&](int c0) {
#pragma unroll
    for (int j = 0; j < WPE; ++j) {
      const int i = tid + j * 256;
      wq[j] = wsrc[(c0 >> 3) * WV + ((j + 1) * 256 <= WV || i < WV ? i : 0)];
    }
  };
  if (!(a.debug & 2)) { load_x(0); load_w(0); }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // (gradient fan-in — accumulate — reads the destination in the EPILOGUE: loads left pending on the accumulator registers here put an
  //  s_waitcnt vmcnt(0) in front of the first MFMA, i.e. behind the whole first prefetch; measured on the Block3d fan-in launches)

  for (int c0 = 0; c0 < a.Cin; c0 += 8) {
    __syncthreads();                                           // everyone is done reading the previous group
    // registers -> LDS: chain (BN + LeakyReLU of the producer) on in-volume samples, round to bf16, 8 channels per position
    if constexpr (XB) {
      if (!(a.debug & 4)) {
        // 4 positions x 8 channels per piece: position j of the piece takes half (j & 1) of dword (j >> 1) of every channel; its 16 bytes
        // in the operand buffer are the 8 channels' halves, two per dword — a 4 x 8 transposition of 16-bit values in registers
#pragma unroll
        for (int e = 0; e < G::EQ; ++e) {
          const int qi = tid + e * 256;
          if ((e + 1) * 256 <= G::NQ || qi < G::NQ) {
            if (a.chain == nullptr) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                unsigned o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  // v_perm_b32: bytes 0-3 of the selector index the second source, 4-7 the first
                  o[k] = __builtin_amdgcn_perm(sq[2 * k + 1][e][j >> 1], sq[2 * k][e][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
                }
                *reinterpret_cast<u32x4*>(xl + (qpos[e] + j) * 4) = (u32x4){o[0], o[1], o[2], o[3]};

              }
            } else {
              unsigned o[4][4];                                                   // [position][channel pair]
              const bool in = qoff[e] >= 0;                                       // zero padding stays zero
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const Chain ch0 = load_chain(a.chain, min(c0 + 2 * k, a.Cin - 1)), ch1 = load_chain(a.chain, min(c0 + 2 * k + 1, a.Cin - 1));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const unsigned u0 = sq[2 * k][e][j >> 1], u1 = sq[2 * k + 1][e][j >> 1];
                  const float x0 = __builtin_bit_cast(float, (j & 1) ? (u0 & 0xffff0000u) : (u0 << 16));
                  const float x1 = __builtin_bit_cast(float, (j & 1) ? (u1 & 0xffff0000u) : (u1 << 16));
                  o[j][k] = pack_bf16(in ? apply_chain(ch0, x0) : x0, in ? apply_chain(ch1, x1) : x1);
                }
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(xl + (qpos[e] + j) * 4) = (u32x4){o[j][0], o[j][1], o[j][2], o[j][3]};

            }
          }
        }
      }
    }
    if (!(a.debug & 4)) {
      if constexpr (!XB) {
      if (a.chain) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const Chain ch = load_chain(a.chain, min(c0 + c, a.Cin - 1));
#pragma unroll
          for (int e = 0; e < G::E; ++e) sr[c][e] = goff[e] >= 0 ? apply_chain(ch, sr[c][e]) : sr[c][e];   // zero padding stays zero
        }
      }
#pragma unroll
      for (int e = 0; e < G::E; ++e) {
        const int idx = tid + e * 256;
        if ((e + 1) * 256 <= G::TILE || idx < G::TILE) {
          if constexpr (NS == 1) {
            *reinterpret_cast<u32x4*>(xl + idx * 4) = (u32x4){pack_bf16(sr[0][e], sr[1][e]), pack_bf16(sr[2][e], sr[3][e]),
                                                              pack_bf16(sr[4][e], sr[5][e]), pack_bf16(sr[6][e], sr[7][e])};
          } else {
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) split3_pair(sr[2 * c][e], sr[2 * c + 1][e], h[c], m[c], l[c]);
            *reinterpret_cast<u32x4*>(xl + idx * 4) = (u32x4){h[0], h[1], h[2], h[3]};
            *reinterpret_cast<u32x4*>(xl + XW + idx * 4) = (u32x4){m[0], m[1], m[2], m[3]};
            *reinterpret_cast<u32x4*>(xl + 2 * XW + idx * 4) = (u32x4){l[0], l[1], l[2], l[3]};
          }
        }
      }
      }
#pragma unroll
      for (int j = 0; j < WPE; ++j) {
        const int i = tid + j * 256;
        if ((j + 1) * 256 <= WV || i < WV) *reinterpret_cast<u32x4*>(wl + i * 8) = wq[j];
      }
    }
    __syncthreads();
    if (c0 + 8 < a.Cin && !(a.debug & 2)) { load_x(c0 + 8); load_w(c0 + 8); }   // next group behind this group's MFMAs
    if (a.debug & 1) continue;

#pragma unroll
    for (int g = 0; g < NTG; ++g) {
      // the A fragment of this tap group only lives across its NT MFMAs (7 resident fragments cost 28 registers = the
      // third wave per SIMD)
      auto frag_w = [&](int term) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wl + term * WW + (4 * g + lk) * WT + lj * 8)); };
      auto frag_x = [&](int term, int p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xl + term * XW + p * 4)); };
      // the tap group's base position is made opaque HERE: left to itself the compiler hoists all NTG * NT loop-invariant fragment addresses
      // out of the channel loop (56 registers, spilled) instead of one base per group + the instruction's immediate offset
      int pg = pbase + toff[g];
      asm volatile("" : "+v"(pg));
      if constexpr (NS == 1) {
        bf16x8 af[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = frag_w(m);
#pragma unroll
        for (int t = 0; t < NT; ++t) {                         // consecutive MFMAs go to different accumulators
          const int p = pg + (t / NH) * G::IW + (t % NH) * 16;
          const bf16x8 xf = frag_x(0, p);
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], xf, acc[m][t], 0, 0, 0);
        }
      } else {
        const bf16x8 wh = frag_w(0), wm = frag_w(1), wlo = frag_w(2);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int p = pg + (t / NH) * G::IW + (t % NH) * 16;
          const bf16x8 xh = frag_x(0, p), xm = frag_x(1, p), xlo = frag_x(2, p);
          // smallest terms first: l*h, h*l (2^-16), m*m (2^-16), h*m, m*h (2^-8), h*h
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh, acc[0][t], 0, 0, 0);
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xlo, acc[0][t], 0, 0, 0);
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, acc[0][t], 0, 0, 0);
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, acc[0][t], 0, 0, 0);
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, acc[0][t], 0, 0, 0);
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[0][t], 0, 0, 0);
        }
      }
    }
  }

  if constexpr (FLIP && NS == 1 && KD == 3) if (a.x2 != nullptr) {
    // 16 channels of x2 per step: two [position][8 channels] planes of the tile's OUTPUT positions (no halo) in the operand buffer; lane slot
    // lk < 2 multiplies plane lk (K = 32 with 16 real channels: the A fragments of slots 2, 3 are zero, their B reads repeat planes 0, 1)
    constexpr int NPOS = G::TZ * G::TY * G::TW, E2 = (NPOS + 255) / 256;
    static_assert(2 * NPOS <= G::TILE, "the second input reuses the halo-tile buffer");
    // bf16 tensors: 4-element pieces as above — the pieces of plane (qi / NQ2) are (qi % NQ2) <-> (dz, hy, q); 8 loads of 8 bytes per piece
    constexpr int NQ2 = NPOS / 4, EQ2 = (2 * NQ2 + 255) / 256;
    int g2[XB ? EQ2 : E2];
    if constexpr (XB) {
#pragma unroll
      for (int e = 0; e < EQ2; ++e) {
        const int qi = (tid + e * 256) % NQ2;
        const int q = qi % (G::TW / 4), row = qi / (G::TW / 4);
        const int hy = row % G::TY, dz = row / G::TY;
        const int od = od0 + dz, oh = oh0 + hy, ow = ow0 + 4 * q;
        g2[e] = (tid + e * 256 < 2 * NQ2 && od < Do && oh < Ho && ow < Wo) ? (od * Ho + oh) * Wo + ow : -1;
      }
    } else {
#pragma unroll
      for (int e = 0; e < E2; ++e) {
        const int idx = tid + e * 256;
        const int col = idx % G::TW, row = idx / G::TW;
        const int hy = row % G::TY, dz = row / G::TY;
        const int od = od0 + dz, oh = oh0 + hy, ow = ow0 + col;
        g2[e] = (idx < NPOS && od < Do && oh < Ho && ow < Wo) ? (od * Ho + oh) * Wo + ow : -1;
      }
    }
    for (int c0 = 0; c0 < a.C2; c0 += 16) {
      float s2[XB ? 1 : 16][XB ? 1 : E2], w2q[2 * MT];
      unsigned q2[XB ? 8 : 1][XB ? EQ2 : 1][2];
      if constexpr (XB) {
#pragma unroll
        for (int e = 0; e < EQ2; ++e) {
          const int plane = (tid + e * 256) / NQ2;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const __amdgpu_buffer_rsrc_t r = dpi_buffer_t(dpi_at(a.x2, (size_t)min(c0 + 8 * plane + c, a.C2 - 1) * Vo, true), Vo, true);
            const dpi_u32x2v u = __builtin_bit_cast(dpi_u32x2v, __builtin_amdgcn_raw_buffer_load_b64(r, g2[e] >= 0 ? g2[e] * 2 : -8, 0, 0));
            q2[c][e][0] = u.x; q2[c][e][1] = u.y;
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const __amdgpu_buffer_rsrc_t r = dpi_buffer(a.x2 + (size_t)min(c0 + c, a.C2 - 1) * Vo, Vo * sizeof(float));
#pragma unroll
          for (int e = 0; e < E2; ++e) s2[c][e] = dpi_buffer_load(r, g2[e] * 4);
        }
      }
#pragma unroll
      for (int j = 0; j < 2 * MT; ++j) {
        const int q = tid + (j & 1) * 256;                       // A fragment element (lane64 = q >> 3, i = q & 7) of output tile j / 2
        const int i = q & 7, l64 = q >> 3;
        const int co = n0 + 16 * (j >> 1) + (l64 & 15), c = c0 + 8 * (l64 >> 4) + i;
        const bool ok = (l64 >> 4) < 2 && co < a.Cout && c < a.C2;
        const float v = a.w2[(ok ? co : 0) * a.w2_co_stride + (ok ? c : 0) * a.w2_c_stride];
        w2q[j] = ok ? v : 0.f;
      }
      __syncthreads();                                           // the main loop's last group (or the previous step) has been read
      if constexpr (XB) {
#pragma unroll
        for (int e = 0; e < EQ2; ++e) {
          const int qi = tid + e * 256;
          if ((e + 1) * 256 <= 2 * NQ2 || qi < 2 * NQ2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              unsigned o[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_perm(q2[2 * k + 1][e][j >> 1], q2[2 * k][e][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
              *reinterpret_cast<u32x4*>(xl + (4 * qi + j) * 4) = (u32x4){o[0], o[1], o[2], o[3]};     // plane * NPOS + 4 * piece + j
            }
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < E2; ++e) {
          const int idx = tid + e * 256;
          if ((e + 1) * 256 <= NPOS || idx < NPOS) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
              *reinterpret_cast<u32x4*>(xl + (g * NPOS + idx) * 4) = (u32x4){pack_bf16(s2[8 * g][e], s2[8 * g + 1][e]), pack_bf16(s2[8 * g + 2][e], s2[8 * g + 3][e]),
                                                                             pack_bf16(s2[8 * g + 4][e], s2[8 * g + 5][e]), pack_bf16(s2[8 * g + 6][e], s2[8 * g + 7][e])};
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 2 * MT; ++j) wl[tid + j * 256] = (unsigned short)bf16_bits(w2q[j]);
      __syncthreads();
      bf16x8 af[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wl + m * 512 + lane * 8));
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int p2 = ((wz * G::TY + wh + t / NH) * G::TW + (t % NH) * 16 + lj) + (lk & 1) * NPOS;
        const bf16x8 xf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xl + p2 * 4));
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], xf, acc[m][t], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: D row = co (4*lk + r), D col = voxel lj (same layout as the fp32 16x16x4 MFMA) ---------------------------
  const bool interior = od0 + G::TZ <= Do && oh0 + G::TY <= Ho && ow0 + G::TW <= Wo && n0 + 16 * MT <= a.Cout;
  const int vbase = ((od0 + wz) * Ho + oh0 + wh) * Wo + ow0 + lj;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (a.accumulate) {                                          // all of the tile's destination values requested together, then added
      float old[4][NT];
  #pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = n0 + 16 * m + 4 * lk + r;
        const float* __restrict__ yo = dpi_at(a.y, (size_t)(co < a.Cout ? co : 0) * Vo + vbase, a.yb);
  #pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int od = od0 + wz, oh = oh0 + wh + t / NH, ow = ow0 + (t % NH) * 16 + lj;
          const bool ok = interior || (co < a.Cout && od < Do && oh < Ho && ow < Wo);
          const size_t o = (t / NH) * Wo + (t % NH) * 16;
          old[r][t] = ok ? (YB ? __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(yo)[o]) : yo[o]) : 0.f;     // bf16: raw, widened below
        }
      }
  #pragma unroll
      for (int r = 0; r < 4; ++r)
  #pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t][r] += YB ? dpi_widen_raw(old[r][t]) : old[r][t];
    }
  #pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = n0 + 16 * m + 4 * lk + r;
      const bool cok = co < a.Cout;
      const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
      float* __restrict__ yc = dpi_at(a.y, (size_t)(cok ? co : 0) * Vo + vbase, a.yb);
      double s = 0.0, q = 0.0;
      if (a.yb) {
        // bf16 destination: two voxels per dword store where the rows are even-aligned (dpi_st_bf16_row); statistics describe what is stored
        const bool pairs = !(Wo & 1) && !(Vo & 1) && !((uintptr_t)a.y & 3);
  #pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int od = od0 + wz, oh = oh0 + wh + t / NH, ow = ow0 + (t % NH) * 16 + lj;
          const bool ok = interior || (cok && od < Do && oh < Ho && ow < Wo);
          const float v = dpi_round_bf16(acc[m][t][r] + bv);
          dpi_st_bf16_row(yc, (t / NH) * Wo + (t % NH) * 16, v, ok, pairs, lj);
          if (ok) { s += v; q += (double)v * v; }
        }
      } else if (interior) {
  #pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float v = acc[m][t][r] + bv;
          yc[(t / NH) * Wo + (t % NH) * 16] = v;
          if (a.partials) { s += v; q += (double)v * v; }
        }
      } else {
  #pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int od = od0 + wz, oh = oh0 + wh + t / NH, ow = ow0 + (t % NH) * 16 + lj;
          if (cok && od < Do && oh < Ho && ow < Wo) {
            const float v = acc[m][t][r] + bv;
            yc[(t / NH) * Wo + (t % NH) * 16] = v;
            s += v;
            q += (double)v * v;
          }
        }
      }
      if (a.partials) {
  #pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (lj == 0) { red[wid][16 * m + 4 * lk + r][0] = s; red[wid][16 * m + 4 * lk + r][1] = q; }
      }
    }
  }
  if (a.partials) {
    __syncthreads();
    if (tid < 32 * MT) {
      const int c = tid >> 1, which = tid & 1;
      const double rsum = red[0][c][which] + red[1][c][which] + red[2][c][which] + red[3][c][which];
      if (n0 + c < a.Cout) a.partials[((size_t)tile_id * a.Cout + n0 + c) * 2 + which] = rsum;
    }
  }
}

// The layer's weights as A fragments: block (group, co block) writes [term][tap 4 NTG][co 16][ci 8] bf16 — rounded (NS = 1) or split into
// three bf16 terms (NS = 3), taps flipped for backward-data, zeros for channels / taps past the end.  One launch in front of every
// conv_bf16_kernel launch, on its stream: a few microseconds over <= 1 MB, instead of the same gather by every tile of the main kernel.
template <int KD, bool FLIP, int NS>
__global__ __launch_bounds__(256) void conv_bf16_pack_kernel(const float* __restrict__ w, long w_out_stride, long w_in_stride, int Cin, int Cout,
                                                             unsigned short* __restrict__ out, int mt) {
  constexpr int TAPS = KD * 9, NTG = (TAPS + 3) / 4, WW = NTG * 512;
  // the mt 16-channel tiles of one workgroup of the main kernel lie side by side: [co block / mt][group][co block % mt]
  unsigned short* __restrict__ const o = out + (((size_t)(blockIdx.y / mt) * gridDim.x + blockIdx.x) * mt + blockIdx.y % mt) * NS * WW;
  for (int q = threadIdx.x; q < WW; q += 256) {
    const int ci = blockIdx.x * 8 + (q & 7), co = blockIdx.y * 16 + ((q >> 3) & 15), tap = q >> 7;
    const bool ok = co < Cout && ci < Cin && tap < TAPS;
    const float v = ok ? w[co * w_out_stride + ci * w_in_stride + (FLIP ? TAPS - 1 - tap : tap)] : 0.f;
    if constexpr (NS == 1) o[q] = (unsigned short)bf16_bits(v);
    else {
      unsigned h, m, l;
      split3(v, h, m, l);
      o[q] = (unsigned short)h; o[WW + q] = (unsigned short)m; o[2 * WW + q] = (unsigned short)l;
    }
  }
}


// ---- stride 2 (forward, bf16 tensors): the four down-sampling layers (mulresunet.py:185-199, nn.Conv3d(.., 3, stride=2, padding=1)) ----------
// Same implicit GEMM, tile and fragment layouts as above with two changes.  (1) A workgroup owns a 2 x 4 x 16 OUTPUT tile (wave = (slice,
// row pair), two 16-voxel blocks per wave) and stages the 5 x 9 x 40 input tile that starts 4 columns left of column 2 ow0 (whole aligned
// 4-element pieces, as GeoB WIDE).  (2) The tile is stored with EVEN and ODD input columns in separate planes —
// position(dz, hy, col) = ((dz * 9 + hy) * 2 + (col & 1)) * 20 + (col >> 1) — so that the 16 lanes of a B-fragment read, which want input
// columns 2 ow + kw - 1 for 16 consecutive ow, read 16 consecutive positions (conflict-free) instead of every second one.
// MT output-channel tiles per workgroup (the tile is small: 8 accumulator registers per channel tile).
struct S2Args {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ bias;
  float* __restrict__ y;
  double* __restrict__ partials;
  const unsigned short* __restrict__ wpk;
  int Cin, Cout;
  int D, H, W, Do, Ho, Wo;
  int ntd, nth, ntw, ny;
};

template <int MT>
__global__ __launch_bounds__(256, 4) void conv_bf16_s2_kernel(S2Args a) {
  constexpr int TAPS = 27, NTG = 7, WW = NTG * 512;
  constexpr int TZ = 2, TY = 4, TW = 16, ID = 5, IH = 9, IW = 40, HP = IW / 2;            // HP: positions per parity plane of a row
  constexpr int TILE = ID * IH * IW, NQ = ID * IH * (IW / 4), EQ = (NQ + 255) / 256;
  constexpr int WV = MT * WW / 8, WPE = (WV + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned xl[TILE * 4];
  __shared__ __attribute__((aligned(16))) unsigned short wl[MT * WW];
  __shared__ double red[4][16 * MT][2];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  int tile_id, ytile;
  if (!xcd_tile_b(blockIdx.x, a.ntd * a.nth * a.ntw, a.ny, tile_id, ytile)) return;
  const int n0 = ytile * 16 * MT;
  const size_t V = (size_t)a.D * a.H * a.W, Vo = (size_t)a.Do * a.Ho * a.Wo;
  int od0, oh0, ow0;
  {
    int bt = tile_id;
    const int tw_i = bt % a.ntw; bt /= a.ntw;
    const int th_i = bt % a.nth; bt /= a.nth;
    od0 = bt * TZ; oh0 = th_i * TY; ow0 = tw_i * TW;
  }
  const int wz = wid >> 1, wh = (wid & 1) * 2;                 // this wave: output slice wz, rows wh, wh + 1
  // this thread's 4-element pieces of the input tile: global element offset (or -1) and first position of the piece's row
  int qoff[EQ], qrow[EQ];
#pragma unroll
  for (int e = 0; e < EQ; ++e) {
    const int qi = tid + e * 256;
    const int q = qi % (IW / 4), row = qi / (IW / 4);
    const int hy = row % IH, dz = row / IH;
    const int gd = 2 * od0 - 1 + dz, gh = 2 * oh0 - 1 + hy, gw = 2 * ow0 - 4 + 4 * q;
    const bool ok = qi < NQ && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    qoff[e] = ok ? (gd * a.H + gh) * a.W + gw : -1;
    qrow[e] = (dz * IH + hy) * IW + 2 * q;                     // columns 4q .. 4q+3 -> plane (j & 1), index 2q + (j >> 1)
  }
  int toff[NTG];
#pragma unroll
  for (int g = 0; g < NTG; ++g) {
    const int t = min(4 * g + lk, TAPS - 1);
    const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    // input column 2 ow + kw - 1 = tile column 2 ow_l + kw + 3: parity of kw + 3, index ow_l + (kw + 3) / 2
    toff[g] = (kd * IH + kh) * IW + ((kw + 3) & 1) * HP + ((kw + 3) >> 1);
  }
  const int pbase = (2 * wz * IH + 2 * wh) * IW + lj;         // output (wz, wh, lj) at tap (0, 0, 0): tile row (2 wz, 2 wh)

  unsigned sq[8][EQ][2];
  auto load_x = [&](int c0) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int ci = min(c0 + c, a.Cin - 1);
      const __amdgpu_buffer_rsrc_t r = dpi_buffer_t(dpi_at(a.x, (size_t)ci * V, true), V, true);
#pragma unroll
      for (int e = 0; e < EQ; ++e) {
        const dpi_u32x2v u = __builtin_bit_cast(dpi_u32x2v, __builtin_amdgcn_raw_buffer_load_b64(r, qoff[e] >= 0 ? qoff[e] * 2 : -8, 0, 0));
        sq[c][e][0] = u.x; sq[c][e][1] = u.y;
      }
    }
  };
  u32x4 wq[WPE];
  const u32x4* __restrict__ const wsrc = reinterpret_cast<const u32x4*>(a.wpk) + (size_t)ytile * ((a.Cin + 7) >> 3) * WV;
  auto load_w = [&](int c0) {
#pragma unroll
    for (int j = 0; j < WPE; ++j) {
      const int i = tid + j * 256;
      wq[j] = wsrc[(c0 >> 3) * WV + ((j + 1) * 256 <= WV || i < WV ? i : 0)];
    }
  };
  load_x(0); load_w(0);
  f32x4 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) { acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = acc[m][0]; }

  for (int c0 = 0; c0 < a.Cin; c0 += 8) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EQ; ++e) {
      const int qi = tid + e * 256;
      if ((e + 1) * 256 <= NQ || qi < NQ) {
        unsigned o[4][4];
        if (a.chain == nullptr) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) o[j][k] = __builtin_amdgcn_perm(sq[2 * k + 1][e][j >> 1], sq[2 * k][e][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
        } else {
          const bool in = qoff[e] >= 0;                       // zero padding stays zero
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const Chain ch0 = load_chain(a.chain, min(c0 + 2 * k, a.Cin - 1)), ch1 = load_chain(a.chain, min(c0 + 2 * k + 1, a.Cin - 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned u0 = sq[2 * k][e][j >> 1], u1 = sq[2 * k + 1][e][j >> 1];
              const float x0 = __builtin_bit_cast(float, (j & 1) ? (u0 & 0xffff0000u) : (u0 << 16));
              const float x1 = __builtin_bit_cast(float, (j & 1) ? (u1 & 0xffff0000u) : (u1 << 16));
              o[j][k] = pack_bf16(in ? apply_chain(ch0, x0) : x0, in ? apply_chain(ch1, x1) : x1);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<u32x4*>(xl + (qrow[e] + (j & 1) * HP + (j >> 1)) * 4) = (u32x4){o[j][0], o[j][1], o[j][2], o[j][3]};
      }
    }
#pragma unroll
    for (int j = 0; j < WPE; ++j) {
      const int i = tid + j * 256;
      if ((j + 1) * 256 <= WV || i < WV) *reinterpret_cast<u32x4*>(wl + i * 8) = wq[j];
    }
    __syncthreads();
    if (c0 + 8 < a.Cin) { load_x(c0 + 8); load_w(c0 + 8); }
#pragma unroll
    for (int g = 0; g < NTG; ++g) {
      int pg = pbase + toff[g];
      asm volatile("" : "+v"(pg));
      bf16x8 af[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wl + m * WW + (g * 64 + lane) * 8));
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 xf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xl + (pg + t * 2 * IW) * 4));
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], xf, acc[m][t], 0, 0, 0);
      }
    }
  }

  // epilogue: D row = co (4 lk + r), D col = output voxel lj of rows wh, wh + 1 of slice wz
  const int od = od0 + wz;
  const bool pairs = !(a.Wo & 1) && !(Vo & 1) && !((uintptr_t)a.y & 3);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = n0 + 16 * m + 4 * lk + r;
      const bool cok = co < a.Cout;
      const float bv = (a.bias && cok) ? a.bias[co] : 0.f;
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int oh = oh0 + wh + t, ow = ow0 + lj;
        const bool ok = cok && od < a.Do && oh < a.Ho && ow < a.Wo;
        const float v = dpi_round_bf16(acc[m][t][r] + bv);
        float* __restrict__ yc = dpi_at(a.y, (size_t)(cok ? co : 0) * Vo + ((size_t)min(od, a.Do - 1) * a.Ho + min(oh, a.Ho - 1)) * a.Wo + ow0, true);
        dpi_st_bf16_row(yc, lj, v, ok, pairs, lj);
        if (ok) { s += v; q += (double)v * v; }
      }
      if (a.partials) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if (lj == 0) { red[wid][16 * m + 4 * lk + r][0] = s; red[wid][16 * m + 4 * lk + r][1] = q; }
      }
    }
  }
  if (a.partials) {
    __syncthreads();
    if (tid < 32 * MT) {
      const int c = tid >> 1, which = tid & 1;
      const double rsum = red[0][c][which] + red[1][c][which] + red[2][c][which] + red[3][c][which];
      if (n0 + c < a.Cout) a.partials[((size_t)tile_id * a.Cout + n0 + c) * 2 + which] = rsum;
    }
  }
}


// ---- stride 2, backward-data (bf16 tensors): dx[ci][i] = sum_co sum_k W[co][ci][k] dY[co][o] over the taps with i = 2 o + k - 1 per axis -----
// An even fine index has ONE such tap per axis (k = 1, o = i / 2), an odd one two (k = 0, o = (i + 1) / 2; k = 2, o = (i - 1) / 2): the eight
// parity classes (pd, ph, pw) of the output have 1 / 2 / 2 / 4 / 2 / 4 / 4 / 8 of the 27 taps.  GEMM per class:
//     D[ci 16][16 fine voxels of one parity along w] += A[ci 16][K 32 = 4 taps x 8 co] * B[K][voxels]
// with B from the small coarse dY tile in LDS ([position][8 co], 2 x 3 x 20 positions for a 2 x 4 x 32 fine tile: voxel j of a class reads
// coarse column j or j + 1 — consecutive positions) and A from a class-wise packing of the weights (conv_bf16_s2_bwd_pack_kernel: nine K
// blocks per (16 ci, 8 co): classes 0 .. 6 one each, class 7 two; unused tap slots zero).  9 MFMAs per 8 dY channels and fine tile instead of
// the 7 x 8 of a zero-inserted stride-1 launch.  Wave = (depth parity, row pair); the two w-parities of a row sit in the same lane, so the
// epilogue stores (and, for a gradient fan-in, first reads) one packed dword per lane.
struct S2BArgs {
  const float* __restrict__ dy;
  float* __restrict__ dx;
  const unsigned short* __restrict__ wpk;
  int Cin, Cout;               // of the layer: dx has Cin channels, dy Cout
  int D, H, W, Do, Ho, Wo;
  int ntd, nth, ntw, ny;
  int accumulate;
};

// tap and coarse offset of K-slot q of parity class (pd, ph, pw): a parity-1 axis takes one bit of q (w first, then h, then d):
// bit 0 -> k = 0, coarse offset +1;  bit 1 -> k = 2, offset 0;  a parity-0 axis: k = 1, offset 0.  false: the class has fewer slots.
__host__ __device__ __forceinline__ bool s2_slot(int pd, int ph, int pw, int q, int& kd, int& kh, int& kw, int& od, int& oh, int& ow) {
  const int n = pd + ph + pw;
  const bool valid = q < (1 << n);
  const int vw = pw ? (q & 1) : 0; q = pw ? q >> 1 : q;
  const int vh = ph ? (q & 1) : 0; q = ph ? q >> 1 : q;
  const int vd = pd ? (q & 1) : 0;
  kw = pw ? (vw ? 2 : 0) : 1; ow = pw ? (vw ? 0 : 1) : 0;
  kh = ph ? (vh ? 2 : 0) : 1; oh = ph ? (vh ? 0 : 1) : 0;
  kd = pd ? (vd ? 2 : 0) : 1; od = pd ? (vd ? 0 : 1) : 0;
  return valid;
}

__global__ __launch_bounds__(256) void conv_bf16_s2_bwd_pack_kernel(const float* __restrict__ w, int Cin, int Cout, unsigned short* __restrict__ out, int mt) {
  // block (co group, ci tile) -> [9 K blocks][lane = (slot lk, ci lj)][8 co]; tiles of one workgroup side by side: [tile / mt][group][tile % mt]
  unsigned short* __restrict__ const o = out + (((size_t)(blockIdx.y / mt) * gridDim.x + blockIdx.x) * mt + blockIdx.y % mt) * (9 * 512);
  for (int e = threadIdx.x; e < 9 * 512; e += 256) {
    const int blk = e >> 9, lane = (e >> 3) & 63, j = e & 7;
    const int c8 = blk < 8 ? blk : 7, s_ = blk == 8 ? 1 : 0;
    int kd, kh, kw, od, oh, ow;
    const bool valid = s2_slot(c8 >> 2, (c8 >> 1) & 1, c8 & 1, 4 * s_ + (lane >> 4), kd, kh, kw, od, oh, ow);
    const int co = blockIdx.x * 8 + j, ci = blockIdx.y * 16 + (lane & 15);
    const float v = (valid && co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 27 + (kd * 3 + kh) * 3 + kw] : 0.f;
    o[e] = (unsigned short)bf16_bits(v);
  }
}

template <int MT>
__global__ __launch_bounds__(256, 4) void conv_bf16_s2_bwd_kernel(S2BArgs a) {
  constexpr int CH = 3, CW = 20, TILE = 2 * CH * CW, NQ = 2 * CH * (CW / 4);
  constexpr int WB = 9 * 512, WV = MT * WB / 8, WPE = (WV + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned xl[TILE * 4];
  __shared__ __attribute__((aligned(16))) unsigned short wl[MT * WB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  int tile_id, ytile;
  if (!xcd_tile_b(blockIdx.x, a.ntd * a.nth * a.ntw, a.ny, tile_id, ytile)) return;
  const int m0 = ytile * 16 * MT;
  const size_t V = (size_t)a.D * a.H * a.W, Vo = (size_t)a.Do * a.Ho * a.Wo;
  int cd0, ch0, cw0;                                           // coarse origin of the tile; the fine origin is twice that
  {
    int bt = tile_id;
    const int tw_i = bt % a.ntw; bt /= a.ntw;
    const int th_i = bt % a.nth; bt /= a.nth;
    cd0 = bt; ch0 = 2 * th_i; cw0 = 16 * tw_i;
  }
  const int pd = wid & 1, bh = wid >> 1;                       // this wave: fine slice 2 cd0 + pd, fine rows 2 ch0 + 2 bh + {0, 1}
  // staging: threads 0 .. 29 own one 4-element piece of the coarse tile each (8 channels of it per group)
  int qoff, qpos;
  {
    const int q = tid % (CW / 4), row = tid / (CW / 4);
    const int cy = row % CH, cz = row / CH;
    const int gd = cd0 + cz, gh = ch0 + cy, gw = cw0 + 4 * q;
    const bool ok = tid < NQ && gd < a.Do && gh < a.Ho && gw < a.Wo;
    qoff = ok ? (gd * a.Ho + gh) * a.Wo + gw : -1;
    qpos = (cz * CH + cy) * CW + 4 * q;
  }
  // this wave's K blocks: classes (pd, ph, pw) for (ph, pw) = 00, 01, 10, 11 and, for pd = 1, the second block of class 111
  const int nblk = pd ? 5 : 4;
  int poff[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int ph = i < 4 ? (i >> 1) : 1, pw = i < 4 ? (i & 1) : 1;
    int kd, kh, kw, od, oh, ow;
    s2_slot(pd, ph, pw, (i == 4 ? 4 : 0) + lk, kd, kh, kw, od, oh, ow);
    poff[i] = (od * CH + bh + oh) * CW + ow + lj;
  }
  unsigned sq[8][2];
  auto load_x = [&](int c0) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int co = min(c0 + c, a.Cout - 1);                  // channels past Cout: their weights are zero
      const __amdgpu_buffer_rsrc_t r = dpi_buffer_t(dpi_at(a.dy, (size_t)co * Vo, true), Vo, true);
      const dpi_u32x2v u = __builtin_bit_cast(dpi_u32x2v, __builtin_amdgcn_raw_buffer_load_b64(r, qoff >= 0 ? qoff * 2 : -8, 0, 0));
      sq[c][0] = u.x; sq[c][1] = u.y;
    }
  };
  u32x4 wq[WPE];
  const u32x4* __restrict__ const wsrc = reinterpret_cast<const u32x4*>(a.wpk) + (size_t)ytile * ((a.Cout + 7) >> 3) * WV;
  auto load_w = [&](int c0) {
#pragma unroll
    for (int j = 0; j < WPE; ++j) {
      const int i = tid + j * 256;
      wq[j] = wsrc[(c0 >> 3) * WV + ((j + 1) * 256 <= WV || i < WV ? i : 0)];
    }
  };
  load_x(0); load_w(0);
  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[m][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < a.Cout; c0 += 8) {
    __syncthreads();
    if (tid < NQ) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_perm(sq[2 * k + 1][j >> 1], sq[2 * k][j >> 1], (j & 1) ? 0x07060302u : 0x05040100u);
        *reinterpret_cast<u32x4*>(xl + (qpos + j) * 4) = (u32x4){o[0], o[1], o[2], o[3]};
      }
    }
#pragma unroll
    for (int j = 0; j < WPE; ++j) {
      const int i = tid + j * 256;
      if ((j + 1) * 256 <= WV || i < WV) *reinterpret_cast<u32x4*>(wl + i * 8) = wq[j];
    }
    __syncthreads();
    if (c0 + 8 < a.Cout) { load_x(c0 + 8); load_w(c0 + 8); }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i < nblk) {                                          // wave-uniform
        const int blk = pd ? (i < 4 ? 4 + i : 8) : i, cls = i < 4 ? i : 3;
        const bf16x8 xf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xl + poff[i] * 4));
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][cls] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wl + m * WB + (blk * 64 + lane) * 8)), xf,
                                                                acc[m][cls], 0, 0, 0);
      }
    }
  }
  // epilogue: D row = ci (4 lk + r), D col = j (lj): fine columns 2 (cw0 + j) + {0, 1} of rows 2 (ch0 + bh) + {0, 1}, slice 2 cd0 + pd
  const int id = 2 * cd0 + pd, iw = 2 * (cw0 + lj);
  unsigned short* __restrict__ const dxh = reinterpret_cast<unsigned short*>(a.dx);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = m0 + 16 * m + 4 * lk + r;
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const int ih = 2 * (ch0 + bh) + ph;
        if (ci < a.Cin && id < a.D && ih < a.H && iw < a.W) {
          unsigned* __restrict__ const p = reinterpret_cast<unsigned*>(dxh + (size_t)ci * V + ((size_t)id * a.H + ih) * a.W + iw);
          float v0 = acc[m][ph * 2 + 0][r], v1 = acc[m][ph * 2 + 1][r];
          if (a.accumulate) {
            const unsigned old = *p;
            v0 += __builtin_bit_cast(float, old << 16); v1 += __builtin_bit_cast(float, old & 0xffff0000u);
          }
          *p = pack_bf16(v0, v1);
        }
      }
    }
}

}  // namespace

// Packed-weight scratch: one slot per (weight tensor, direction, shape, kernel family), carved from 64 MB chunks and kept for the life of
// the process.  Every launch re-packs into its layer's slot, on the launch's stream, so the slot always holds what the main kernel behind
// it reads; two launches of the SAME layer on different streams write identical bytes.  Keyed by stream instead, a graph capture (which
// runs on a stream of its own) would need its region allocated during the capture, and two graphs captured on one stream would share one.
// A layer first seen during a capture takes its slot from a chunk that already exists (hipMalloc is not capturable): run one eager
// iteration first, as for every captured workload here.
static constexpr size_t kPackChunk = 64u << 20, kPackMaxSlot = 16u << 20, kPackMaxTotal = (size_t)4 << 30;
// Round 5: (a) per DEVICE — a chunk lives on the device that was current when it was allocated, so the device is part of
// the key and every device has its own chunks; (b) recyclable — dpi_pack_forget(w) hands the slots of a weight tensor that is going away
// (the host side calls it when a patch's network is replaced: every patch builds a new net, 343 of them per configs[2] volume) to a free
// list by size, from which the next layer of that size takes its slot: a long multi-patch job holds one net's worth of slots per
// concurrently optimised patch instead of growing until the 4 GB cap fails every launch.
struct PackDev {
  std::vector<void*> chunks;
  char* chunk = nullptr;
  size_t used = kPackChunk;
  std::multimap<size_t, void*> free_slots;      // bytes -> slot, from dpi_pack_forget
};
struct PackSlot { void* p; size_t bytes; };
static std::mutex g_pack_mutex;
static std::map<int, PackDev> g_pack_dev;
static std::map<std::tuple<int, const void*, int, int, int, int>, PackSlot> g_pack_slots;
size_t dpi_pack_max_slot() { return kPackMaxSlot; }
void* dpi_pack_slot(const void* w, int kd, int cin, int cout, int tag, size_t nbytes) {
  std::lock_guard<std::mutex> lock(g_pack_mutex);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dpi_set_error("packed-weight scratch: hipGetDevice failed"); return nullptr; }
  const auto key = std::make_tuple(dev, w, kd, cin, cout, tag);
  const auto it = g_pack_slots.find(key);
  if (it != g_pack_slots.end()) return it->second.p;
  const size_t bytes = (nbytes + 255) & ~(size_t)255;
  if (bytes > kPackChunk) { dpi_set_error("packed-weight scratch: %zu bytes in one slot", bytes); return nullptr; }
  PackDev& D = g_pack_dev[dev];
  const auto fr = D.free_slots.find(bytes);
  if (fr != D.free_slots.end()) {               // a forgotten layer's slot of exactly this size (same shape on another patch's net)
    void* const slot = fr->second;
    D.free_slots.erase(fr);
    g_pack_slots.emplace(key, PackSlot{slot, bytes});
    return slot;
  }
  if (D.used + bytes > kPackChunk) {
    void* p = nullptr;
    if ((D.chunks.size() + 1) * kPackChunk > kPackMaxTotal) {
      dpi_set_error("packed-weight scratch: %zu MB held on device %d for %zu (weight tensor, shape) pairs; call dpi_pack_forget() for tensors that are gone, or dpi_pack_release()",
                    D.chunks.size() * (kPackChunk >> 20), dev, g_pack_slots.size());
      return nullptr;
    }
    if (hipMalloc(&p, kPackChunk) != hipSuccess) {
      (void)hipGetLastError();
      dpi_set_error("cannot allocate packed-weight scratch (a layer's first launch inside a graph capture? run one eager iteration first)");
      return nullptr;
    }
    D.chunks.push_back(p);
    D.chunk = static_cast<char*>(p);
    D.used = 0;
  }
  void* const slot = D.chunk + D.used;
  D.used += bytes;
  g_pack_slots.emplace(key, PackSlot{slot, bytes});
  return slot;
}
extern "C" size_t dpi_pack_scratch_bytes(void) {
  std::lock_guard<std::mutex> lock(g_pack_mutex);
  size_t n = 0;
  for (const auto& kv : g_pack_dev) n += kv.second.chunks.size() * kPackChunk;
  return n;
}
extern "C" size_t dpi_pack_slot_count(void) {
  std::lock_guard<std::mutex> lock(g_pack_mutex);
  return g_pack_slots.size();
}
extern "C" int dpi_pack_forget(const void* w) {
  // The caller guarantees that no launch reading these slots is in flight or captured in a graph that will still be replayed (the host side
  // calls it after a patch's optimisation has been synchronised, when the patch's network is replaced).
  std::lock_guard<std::mutex> lock(g_pack_mutex);
  int n = 0;
  // the slot map is ordered by (device, weight pointer, ...): one range per device instead of a scan over every slot (a network has ~70 weight tensors
  // and the per-patch driver forgets them all: the scan was slots x tensors under the global mutex)
  for (auto& dv : g_pack_dev) {
    auto it = g_pack_slots.lower_bound(std::make_tuple(dv.first, w, INT_MIN, INT_MIN, INT_MIN, INT_MIN));
    while (it != g_pack_slots.end() && std::get<0>(it->first) == dv.first && std::get<1>(it->first) == w) {
      dv.second.free_slots.emplace(it->second.bytes, it->second.p);
      it = g_pack_slots.erase(it);
      ++n;
    }
  }
  return n;
}
extern "C" int dpi_pack_release(void) {
  std::lock_guard<std::mutex> lock(g_pack_mutex);
  int cur = 0;
  (void)hipGetDevice(&cur);
  int rc = DPI_OK;
  for (auto it = g_pack_dev.begin(); it != g_pack_dev.end();) {
    const int dev = it->first;
    // every OWNING device is synchronised before its chunks go (a launch on any of its streams may still read a slot).  A device that cannot be
    // synchronised keeps its chunks AND its slots (nothing of it is freed: later launches there still find valid memory); every other device's
    // chunks, slots and free list go together, so no slot ever points into freed memory.
    if (!it->second.chunks.empty() && (hipSetDevice(dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) {
      (void)hipGetLastError(); dpi_set_error("dpi_pack_release: synchronising device %d failed (its scratch is kept)", dev); rc = DPI_E_LAUNCH; ++it; continue;
    }
    for (void* p : it->second.chunks) (void)hipFree(p);
    for (auto sl = g_pack_slots.begin(); sl != g_pack_slots.end();) sl = std::get<0>(sl->first) == dev ? g_pack_slots.erase(sl) : std::next(sl);
    it = g_pack_dev.erase(it);
  }
  (void)hipSetDevice(cur);
  return rc;
}
static size_t bf16_pack_bytes(int kd, int cin, int cout, int ns) { return (size_t)2 * cdiv(cout, 32) * cdiv(cin, 8) * ns * ((kd * 9 + 3) / 4) * 512 * sizeof(unsigned short); }

// tile variant: the 4x4x32 tile (2-D: 1x16x32) while it still gives >= 512 workgroups, else row-band tiles of one depth slice.
// (The 4x8x32 tile of the fp32 kernel needs 64 prefetch + 64 accumulator registers here and spills.)
static int g_split_nh = 2;      // column blocks of the split-mode tile (tuning: dpi_set_bf16_debug bit 4 selects 1); measured: 4x4x32 at one
                                // workgroup per CU (81 KB LDS) beats 4x4x16 at three on every layer but 25 -> 1
static int g_bf16_nh = 2;       // ... of the bf16-mode tile (bit 5 selects 1)
static void bf16_variant(const dpi_conv_desc* d, int cout, int* nr, int* nh) {
  const int tz = d->kd == 3 ? 4 : 1, ty = d->kd == 3 ? 4 : 16;
  const long nb = (long)cdiv(d->D, tz) * cdiv(d->H, ty) * cdiv(d->W, 32) * cdiv(cout, 16);
  if (nb >= 512) { *nr = 4; *nh = d->precision == 2 ? g_split_nh : g_bf16_nh; }    // split mode: three operand copies in LDS -> 4x4x16 tile
  else { *nr = 2; *nh = d->W > 16 ? 2 : 1; }
}
static int bf16_tiles(const dpi_conv_desc* d, int nr, int nh, int* ntd, int* nth, int* ntw) {
  const bool slices = d->kd == 3 && nr >= 4;
  const int tz = slices ? 4 : 1, ty = slices ? nr : 4 * nr;
  *ntd = cdiv(d->D, tz); *nth = cdiv(d->H, ty); *ntw = cdiv(d->W, 16 * nh);
  return *ntd * *nth * *ntw;
}

static int g_bf16_debug = 0;
static int g_bf16_all = 0;      // 1: every 3x3(x3) stride-1 convolution (tests); 0: only where the kernel beats the fp32 one
bool dpi_bf16_force_all() { return g_bf16_all != 0; }
extern "C" void dpi_set_bf16_debug(int flags) { g_bf16_debug = (flags & 7) | ((flags & 64) ? 8 : 0); g_bf16_all = (flags >> 3) & 1; g_split_nh = (flags & 16) ? 1 : 2; g_bf16_nh = (flags & 32) ? 1 : 2; }

// Where the mode applies (measured in the iteration, profiles/r02_bf16_kernel_stats_layers.txt): the big-tile variant — full
// resolution and the first coarse level — is 1.2-2.1x faster than the fp32 kernels except for 4 input channels (one half-empty K
// block per tile).  The row-band variants of the coarse levels (hundreds of channels = tens of 8-channel groups, few tiles) were slower
// than the fp32 kernels while every tile gathered its own weights; with the packed weights they win (round 4, below).
static bool bf16_pays(const dpi_conv_desc* d, bool flip) {
  if (g_bf16_all) return true;
  const int cin = flip ? d->Cout : d->Cin, cout = flip ? d->Cin : d->Cout;
  int nr, nh;
  bf16_variant(d, cout, &nr, &nh);
  // bf16 STORAGE (a tensor of this launch is bf16): every big-tile shape — the 4x4x1 kernel that serves the few-channel layers in fp32
  // does not take bf16 tensors, and half the staged bytes move the break-even of this (load-bound) kernel
  if (d->precision == 1 && (dpi_io_in(d, flip) || dpi_io_out(d, flip))) return true;
  // split mode (six MFMAs and three LDS fragments per position): wins 12-48 % over the fp32 kernels when both channel counts are
  // >= 8 (25<->16, 51<->32, 137<->8, 8<->13), loses with <= 4 channels on either side (64->4: 1.28 vs 0.96 ms)
  if (d->precision == 2) return nr == 4 && cin >= 8 && cout >= 8;
  // the row-band variants of the coarse levels: 2-3.5x faster than the fp32 kernels since the weights come pre-packed (51->17 at 64x32x32
  // 0.066 -> 0.026 ms, 212->128 at 32x16x16 0.130 -> 0.042, 276->17 0.279 -> 0.076; level with them only at 16x8x8)
  if (nr != 4) return true;
  return cin > 4 || cout > 16;      // 4 -> 8 forward: 217 vs 196 us (fp32); 4 -> 67 (backward-data of 67 -> 4): 888 vs 1056 us
}

bool dpi_conv_bf16_usable(const dpi_conv_desc* d, bool flip) {
  if (d->precision == 2 && (dpi_io_in(d, flip) || dpi_io_out(d, flip))) return false;      // the split instantiations are compiled for fp32 tensors
  if (bf16_pack_bytes(d->kd, flip ? d->Cout : d->Cin, flip ? d->Cin : d->Cout, d->precision == 2 ? 3 : 1) > kPackMaxSlot) return false;   // the launch's roles (dpi_conv_bf16_run swaps them for flip)
  if (dpi_io_in(d, flip) && (d->W & 3)) return false;      // bf16 input: staged in aligned 4-element pieces of a row (GeoB WIDE); else conv_mfma
  return d->precision >= 1 && d->k == 3 && d->stride == 1 && bf16_pays(d, flip);
}

// whether dpi_conv_bf16_run adds a 1x1x1 second input in the same pass: bf16 arithmetic mode, 3-D, backward-data
bool dpi_conv_bf16_second_ok(const dpi_conv_desc* d, bool flip) { return flip && d->precision == 1 && d->kd == 3 && dpi_conv_bf16_usable(d, flip); }

int dpi_conv_bf16_stat_blocks(const dpi_conv_desc* d) {
  int nr, nh, a, b, c;
  bf16_variant(d, d->Cout, &nr, &nh);
  return bf16_tiles(d, nr, nh, &a, &b, &c);
}

template <int KD, bool FLIP, int NS, bool XB = false, bool YB = false, int MT = 1>
static void launch_bf16_m(const BArgs& a, int nr, int nh, dim3 grid, hipStream_t st) {
  if (nr == 4 && nh == 2) conv_bf16_kernel<KD, 4, 2, FLIP, NS, XB, YB, MT><<<grid, 256, 0, st>>>(a);
  else if (nr == 4) conv_bf16_kernel<KD, 4, 1, FLIP, NS, XB, YB, MT><<<grid, 256, 0, st>>>(a);
  else if (nh == 2) conv_bf16_kernel<KD, 2, 2, FLIP, NS, XB, YB, MT><<<grid, 256, 0, st>>>(a);
  else conv_bf16_kernel<KD, 2, 1, FLIP, NS, XB, YB, MT><<<grid, 256, 0, st>>>(a);
}
// two output tiles per workgroup: only where the launch is short of workgroups anyway (the coarsest levels: 212->128 at 32x16x16 0.042 ->
// 0.039 / 0.033 ms forward / backward, 71->106 0.022 -> 0.018); at full resolution the third and fourth workgroup per CU are worth more
// than the shared staging (16->25 backward at 256x128x128: 0.245 -> 0.257 ms with 3 per CU, 0.286 with 2)
static int bf16_mt(int kd, int cout, int ns, int nr, int ntiles) { return (kd == 3 && ns == 1 && cout > 16 && nr != 4 && ntiles * cdiv(cout, 16) <= 256) ? 2 : 1; }
template <int KD, bool FLIP, int NS, bool XB = false, bool YB = false>
static void launch_bf16_t(const BArgs& a, int nr, int nh, dim3 grid, hipStream_t st) {
  if constexpr (KD == 3 && NS == 1) {
    if (bf16_mt(KD, a.Cout, NS, nr, a.ntd * a.nth * a.ntw) == 2) { launch_bf16_m<KD, FLIP, NS, XB, YB, 2>(a, nr, nh, grid, st); return; }
  }
  launch_bf16_m<KD, FLIP, NS, XB, YB, 1>(a, nr, nh, grid, st);
}
template <int KD, bool FLIP, int NS>
static void launch_bf16(const BArgs& a, int nr, int nh, dim3 grid, hipStream_t st) {
  if constexpr (NS == 1) {          // (split mode takes fp32 tensors only: dpi_conv_bf16_usable)
    if (a.xb && a.yb) { launch_bf16_t<KD, FLIP, NS, true, true>(a, nr, nh, grid, st); return; }
    if (a.xb) { launch_bf16_t<KD, FLIP, NS, true, false>(a, nr, nh, grid, st); return; }
    if (a.yb) { launch_bf16_t<KD, FLIP, NS, false, true>(a, nr, nh, grid, st); return; }
  }
  launch_bf16_t<KD, FLIP, NS, false, false>(a, nr, nh, grid, st);
}

int dpi_conv_bf16_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias, float* y,
                      double* partials, bool flip, int accumulate, hipStream_t st, const MfmaSecond* sec) {
  const int taps = d->kd * 9;
  const int cin = flip ? d->Cout : d->Cin, cout = flip ? d->Cin : d->Cout;
  const long w_out = flip ? taps : (long)d->Cin * taps, w_in = flip ? (long)d->Cin * taps : taps;
  int nr, nh;
  bf16_variant(d, cout, &nr, &nh);
  const int ns = d->precision == 2 ? 3 : 1;
  int t0, t1, t2;
  const int mt = bf16_mt(d->kd, cout, ns, nr, bf16_tiles(d, nr, nh, &t0, &t1, &t2));
  unsigned short* const wpk = static_cast<unsigned short*>(dpi_pack_slot(w, d->kd, cin, cout, (flip ? 1 : 0) | (ns << 1) | (mt << 3), bf16_pack_bytes(d->kd, cin, cout, ns)));
  if (!wpk) return DPI_E_LAUNCH;
  {
    const dim3 pg(cdiv(cin, 8), cdiv(cout, 16 * mt) * mt);
    if (d->precision == 2) {
      if (d->kd == 3) { if (flip) conv_bf16_pack_kernel<3, true, 3><<<pg, 256, 0, st>>>(w, w_out, w_in, cin, cout, wpk, mt); else conv_bf16_pack_kernel<3, false, 3><<<pg, 256, 0, st>>>(w, w_out, w_in, cin, cout, wpk, mt); }
      else { if (flip) conv_bf16_pack_kernel<1, true, 3><<<pg, 256, 0, st>>>(w, w_out, w_in, cin, cout, wpk, mt); else conv_bf16_pack_kernel<1, false, 3><<<pg, 256, 0, st>>>(w, w_out, w_in, cin, cout, wpk, mt); }
    } else {
      if (d->kd == 3) { if (flip) conv_bf16_pack_kernel<3, true, 1><<<pg, 256, 0, st>>>(w, w_out, w_in, cin, cout, wpk, mt); else conv_bf16_pack_kernel<3, false, 1><<<pg, 256, 0, st>>>(w, w_out, w_in, cin, cout, wpk, mt); }
      else { if (flip) conv_bf16_pack_kernel<1, true, 1><<<pg, 256, 0, st>>>(w, w_out, w_in, cin, cout, wpk, mt); else conv_bf16_pack_kernel<1, false, 1><<<pg, 256, 0, st>>>(w, w_out, w_in, cin, cout, wpk, mt); }
    }
  }
  BArgs a{x, chain, w, bias, y, partials, cin, cout, d->D, d->H, d->W, 0, 0, 0, 0, w_out, w_in, wpk, accumulate, g_bf16_debug,
          dpi_io_in(d, flip), dpi_io_out(d, flip), nullptr, nullptr, 0, 0, 0};
  if (a.xb && (((uintptr_t)x & 7) || (sec && ((uintptr_t)sec->x2 & 7)))) { dpi_set_error("conv_bf16_mfma: a bf16 input tensor must be 8-byte aligned"); return DPI_E_ARG; }
  if (sec) { a.x2 = sec->x2; a.w2 = sec->w2; a.C2 = sec->C2; a.w2_co_stride = sec->w2_co_stride; a.w2_c_stride = sec->w2_c_stride; }
  const int ntiles = bf16_tiles(d, nr, nh, &a.ntd, &a.nth, &a.ntw);
  a.ny = cdiv(cout, 16 * mt);
  dim3 grid(8 * cdiv(ntiles, 8) * a.ny);
  if (d->precision == 2) {
    if (d->kd == 3) { if (flip) launch_bf16<3, true, 3>(a, nr, nh, grid, st); else launch_bf16<3, false, 3>(a, nr, nh, grid, st); }
    else { if (flip) launch_bf16<1, true, 3>(a, nr, nh, grid, st); else launch_bf16<1, false, 3>(a, nr, nh, grid, st); }
  } else {
    if (d->kd == 3) { if (flip) launch_bf16<3, true, 1>(a, nr, nh, grid, st); else launch_bf16<3, false, 1>(a, nr, nh, grid, st); }
    else { if (flip) launch_bf16<1, true, 1>(a, nr, nh, grid, st); else launch_bf16<1, false, 1>(a, nr, nh, grid, st); }
  }
  return dpi_check_launch("conv_bf16_mfma");
}

// ---- stride-2 forward on the bf16 MFMA (conv_bf16_s2_kernel): bf16 x and y, bf16 arithmetic, rows of whole 4-element pieces ------------------
bool dpi_conv_bf16_s2_usable(const dpi_conv_desc* d) {
  return d->precision == 1 && d->k == 3 && d->kd == 3 && d->stride == 2 && (d->io & DPI_IO_X_BF16) && (d->io & DPI_IO_Y_BF16) && (d->W & 3) == 0
         && (size_t)d->D * d->H * d->W < ((size_t)1 << 30) && bf16_pack_bytes(3, d->Cin, d->Cout, 1) * 2 <= kPackMaxSlot;
}
static int bf16_s2_tiles(const dpi_conv_desc* d, int* ntd, int* nth, int* ntw) {
  int Do, Ho, Wo;
  dpi_conv_out_dims(d, &Do, &Ho, &Wo);
  *ntd = cdiv(Do, 2); *nth = cdiv(Ho, 4); *ntw = cdiv(Wo, 16);
  return *ntd * *nth * *ntw;
}
int dpi_conv_bf16_s2_stat_blocks(const dpi_conv_desc* d) { int a, b, c; return bf16_s2_tiles(d, &a, &b, &c); }
int dpi_conv_bf16_s2_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias, float* y, double* partials,
                         hipStream_t st) {
  if ((uintptr_t)x & 7) { dpi_set_error("conv_bf16_s2: a bf16 input tensor must be 8-byte aligned"); return DPI_E_ARG; }
  const int mt = d->Cout > 32 ? 4 : (d->Cout > 16 ? 2 : 1);
  const int cout_pad = cdiv(d->Cout, 16 * mt) * mt;            // 16-channel tiles written by the pack kernel
  unsigned short* const wpk = static_cast<unsigned short*>(dpi_pack_slot(w, 3, d->Cin, d->Cout, 32 | (mt << 6),
                                                                         (size_t)cout_pad * cdiv(d->Cin, 8) * 7 * 512 * sizeof(unsigned short)));
  if (!wpk) return DPI_E_LAUNCH;
  conv_bf16_pack_kernel<3, false, 1><<<dim3(cdiv(d->Cin, 8), cout_pad), 256, 0, st>>>(w, (long)d->Cin * 27, 27, d->Cin, d->Cout, wpk, mt);
  S2Args a{x, chain, bias, y, partials, wpk, d->Cin, d->Cout, d->D, d->H, d->W, 0, 0, 0, 0, 0, 0, cdiv(d->Cout, 16 * mt)};
  dpi_conv_out_dims(d, &a.Do, &a.Ho, &a.Wo);
  const int ntiles = bf16_s2_tiles(d, &a.ntd, &a.nth, &a.ntw);
  const dim3 grid(8 * cdiv(ntiles, 8) * a.ny);
  if (mt == 4) conv_bf16_s2_kernel<4><<<grid, 256, 0, st>>>(a);
  else if (mt == 2) conv_bf16_s2_kernel<2><<<grid, 256, 0, st>>>(a);
  else conv_bf16_s2_kernel<1><<<grid, 256, 0, st>>>(a);
  return dpi_check_launch("conv_bf16_s2");
}

// ---- stride-2 backward-data on the bf16 MFMA (conv_bf16_s2_bwd_kernel): bf16 dy and dx, bf16 arithmetic, W a multiple of 8 ----------------
bool dpi_conv_bf16_s2_bwd_usable(const dpi_conv_desc* d) {
  return d->precision == 1 && d->k == 3 && d->kd == 3 && d->stride == 2 && (d->io & DPI_IO_DY_BF16) && (d->io & DPI_IO_DX_BF16) && (d->W & 7) == 0
         && (size_t)d->D * d->H * d->W < ((size_t)1 << 30) && (size_t)cdiv(d->Cin, 16) * cdiv(d->Cout, 8) * 9 * 512 * 2 * 4 <= kPackMaxSlot;
}
int dpi_conv_bf16_s2_bwd_run(const dpi_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate, hipStream_t st) {
  const int mt = d->Cin > 16 ? 2 : 1;
  const int tiles_pad = cdiv(d->Cin, 16 * mt) * mt;
  unsigned short* const wpk = static_cast<unsigned short*>(dpi_pack_slot(w, 3, d->Cin, d->Cout, 33 | (mt << 6),
                                                                         (size_t)tiles_pad * cdiv(d->Cout, 8) * 9 * 512 * sizeof(unsigned short)));
  if (!wpk) return DPI_E_LAUNCH;
  conv_bf16_s2_bwd_pack_kernel<<<dim3(cdiv(d->Cout, 8), tiles_pad), 256, 0, st>>>(w, d->Cin, d->Cout, wpk, mt);
  S2BArgs a{dy, dx, wpk, d->Cin, d->Cout, d->D, d->H, d->W, 0, 0, 0, cdiv(d->D, 2), cdiv(d->H, 4), cdiv(d->W, 32), cdiv(d->Cin, 16 * mt), accumulate};
  dpi_conv_out_dims(d, &a.Do, &a.Ho, &a.Wo);
  const int ntiles = a.ntd * a.nth * a.ntw;
  const dim3 grid(8 * cdiv(ntiles, 8) * a.ny);
  if (mt == 2) conv_bf16_s2_bwd_kernel<2><<<grid, 256, 0, st>>>(a);
  else conv_bf16_s2_bwd_kernel<1><<<grid, 256, 0, st>>>(a);
  return dpi_check_launch("conv_bf16_s2_bwd");
}
