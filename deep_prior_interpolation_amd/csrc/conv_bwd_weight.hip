// Backward-weight of the k=3 (kd = 3 or 1) and k=1 convolutions, fp32.
//
//   dw[co][ci][tap] = sum_p dy[co][p] * T(x)[ci][p*S + tap - pad]
//
// Workgroup = 4 waves.  The block stages the halo tile of 4 input channels in LDS (chain applied while
// staging, exactly as the forward does); wave w owns input channel ci0+w and sweeps the whole spatial
// tile, keeping CO_B x TAPS accumulators per lane.  dy is read straight from global memory (float4 per
// lane, shared by the 4 waves through L1).  Each block walks a contiguous chunk of tiles, then reduces
// its accumulators across the 64 lanes and writes one partial per (chunk, co, ci, tap); a second kernel
// sums the chunk partials in fixed order (deterministic, no atomics).
#include "common.h"
#include <algorithm>

void dpi_conv_out_dims(const dpi_conv_desc* d, int* Do, int* Ho, int* Wo);
size_t dpi_conv_bwd_weight_mfma_ws_floats(const dpi_conv_desc* d);
bool dpi_conv_bwd_weight_mfma_swapped(const dpi_conv_desc* d, const float* chain);
int dpi_conv_bwd_weight_mfma_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* dy, float* dw, float* ws,
                                 hipStream_t st);
size_t dpi_conv_pw_bwd_weight_mfma_ws_floats(const dpi_conv_desc* d);
int dpi_conv_pw_bwd_weight_mfma_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* dy, float* dw, float* ws,
                                    hipStream_t st);
size_t dpi_conv_bwd_weight_smallco_ws_floats(const dpi_conv_desc* d);
int dpi_conv_bwd_weight_smallco_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* dy, float* dw, float* ws,
                                    hipStream_t st);
bool dpi_conv_bf16_bww_usable(const dpi_conv_desc* d);
size_t dpi_conv_bf16_bww_ws_floats(const dpi_conv_desc* d);
int dpi_conv_bf16_bww_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* dy, float* dw, float* ws, hipStream_t st);
// conv_bf16_bww_s2.hip: 3x3x3 stride 2, x and dy bf16, bf16 arithmetic, no chain
bool dpi_conv_bf16_bww_s2_usable(const dpi_conv_desc* d);
size_t dpi_conv_bf16_bww_s2_ws_floats(const dpi_conv_desc* d);
int dpi_conv_bf16_bww_s2_run(const dpi_conv_desc* d, const float* x, const float* dy, float* dw, float* ws, hipStream_t st);
static bool bw_use_smallco(const dpi_conv_desc* d) {
  return d->k == 3 && d->kd == 3 && d->stride == 1 && d->Cout <= 5 && (size_t)d->D * d->H * d->W >= 32768 &&
         (size_t)d->D * d->H * d->W < ((size_t)1 << 26);   // one 32-bit buffer offset spans the (<= 5) dY channels
}
// few output channels, input not chained: the MFMA kernel in its swapped orientation (X rows x (co, tap) columns)
static bool bw_use_mfma_swapped(const dpi_conv_desc* d, const float* x_chain) {
  return d->k == 3 && d->stride == 1 && d->Cin >= 8 && (size_t)d->D * d->H * d->W < ((size_t)1 << 25) &&
         dpi_conv_bwd_weight_mfma_swapped(d, x_chain);
}
static int g_bw_mfma_min_cout = 8;
extern "C" void dpi_set_bwd_weight_mfma_min_cout(int n) { g_bw_mfma_min_cout = n; }
// the MFMA kernel addresses 16 dY channels through one 32-bit buffer offset: 16 * Vo * 4 bytes must stay below 2^31
static bool bw_use_mfma(const dpi_conv_desc* d) {
  return d->k == 3 && d->Cout >= g_bw_mfma_min_cout && (size_t)d->D * d->H * d->W < ((size_t)1 << 25);
}

namespace {

struct BwArgs {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ dy;
  float* __restrict__ ws;   // [nchunks][Cout][Cin][TAPS]
  int Cin, Cout;
  int D, H, W, Do, Ho, Wo;
  int ntd, nth, ntw, ntiles, tiles_per_chunk;
  int xb, dyb;              // storage type of x / dy: 1 = bf16
};

template <int KD, int S, int CO_B, int OW, int TZ, int TY>
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(BwArgs a) {
  constexpr int KS = 3, TX = 8, CI_B = 4;
  constexpr int SD = (KD > 1) ? S : 1;
  constexpr int ID = (TZ - 1) * SD + KD;
  constexpr int IH = (TY - 1) * S + KS;
  constexpr int IW = (TX * OW - 1) * S + KS;
  constexpr int IWP = (IW + 3) & ~3;
  constexpr int TILE = ID * IH * IW;
  constexpr int NT = 256;
  constexpr int E = (TILE + NT - 1) / NT;
  constexpr int CH_LDS = ID * IH * IWP;
  constexpr int WIN = (OW - 1) * S + KS;
  constexpr int TAPS = KD * KS * KS;
  constexpr int PD = (KD - 1) / 2;
  constexpr int NSUB = TZ * TY / 8;   // 8x8 lane patches per tile

  __shared__ __attribute__((aligned(16))) float lds[CI_B * CH_LDS];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ltx = lane & 7, lty = lane >> 3;
  const int ci0 = blockIdx.y * CI_B, co_base = blockIdx.z * CO_B;
  const int my_ci = ci0 + wid;
  const size_t V = (size_t)a.D * a.H * a.W, Vo = (size_t)a.Do * a.Ho * a.Wo;
  const bool wo_vec = (OW == 4) && ((a.Wo & 3) == 0);

  float acc[CO_B][TAPS];
#pragma unroll
  for (int c = 0; c < CO_B; ++c)
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[c][t] = 0.f;

  const int t_begin = blockIdx.x * a.tiles_per_chunk;
  const int t_end = min(t_begin + a.tiles_per_chunk, a.ntiles);
  for (int tile = t_begin; tile < t_end; ++tile) {
    int bt = tile;
    const int tw_i = bt % a.ntw; bt /= a.ntw;
    const int th_i = bt % a.nth; bt /= a.nth;
    const int td_i = bt;
    const int od0 = td_i * TZ, oh0 = th_i * TY, ow0 = tw_i * TX * OW;
    const int id0 = od0 * SD - PD, ih0 = oh0 * S - 1, iw0 = ow0 * S - 1;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int idx = tid + e * NT;
      if (idx < TILE) {
        const int col = idx % IW, row = idx / IW;
        const int hy = row % IH, dz = row / IH;
        const int gd = id0 + dz, gh = ih0 + hy, gw = iw0 + col;
        const bool ok = gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        const int go = (gd * a.H + gh) * a.W + gw;
        const int lo = (dz * IH + hy) * IWP + col;
#pragma unroll
        for (int c = 0; c < CI_B; ++c) {
          const int ci = ci0 + c;
          float v = 0.f;
          if (ok && ci < a.Cin) v = apply_chain(load_chain(a.chain, ci), dpi_ld(a.x, (size_t)ci * V + go, a.xb));
          lds[c * CH_LDS + lo] = v;
        }
      }
    }
    __syncthreads();
    if (my_ci < a.Cin) {
#pragma unroll 1
      for (int sub = 0; sub < NSUB; ++sub) {
        const int tz = sub / (TY / 8), ty = (sub % (TY / 8)) * 8 + lty;
        const int od = od0 + tz, oh = oh0 + ty, ow = ow0 + ltx * OW;
        float g[CO_B][OW];
        const bool row_ok = od < a.Do && oh < a.Ho;
        const size_t obase = ((size_t)od * a.Ho + oh) * a.Wo + ow;
#pragma unroll
        for (int co = 0; co < CO_B; ++co) {
          const int cog = co_base + co;
          const float* gp = dpi_at(a.dy, (size_t)min(cog, a.Cout - 1) * Vo + obase, a.dyb);
          if (row_ok && cog < a.Cout && wo_vec && ow + 3 < a.Wo) {
            const float4 f = dpi_ld4(gp, 0, a.dyb, false);
            g[co][0] = f.x; g[co][1] = f.y; if (OW > 2) { g[co][2] = f.z; g[co][3] = f.w; }
          } else {
#pragma unroll
            for (int o = 0; o < OW; ++o) g[co][o] = (row_ok && cog < a.Cout && ow + o < a.Wo) ? dpi_ld(gp, o, a.dyb) : 0.f;
          }
        }
        const int lbase = wid * CH_LDS + ((tz * SD) * IH + ty * S) * IWP + ltx * OW * S;
#pragma unroll
        for (int kd = 0; kd < KD; ++kd)
#pragma unroll
          for (int kh = 0; kh < KS; ++kh) {
            float win[WIN];
#pragma unroll
            for (int i = 0; i < WIN; ++i) win[i] = lds[lbase + (kd * IH + kh) * IWP + i];
#pragma unroll
            for (int kw = 0; kw < KS; ++kw)
#pragma unroll
              for (int co = 0; co < CO_B; ++co)
#pragma unroll
                for (int o = 0; o < OW; ++o)
                  acc[co][(kd * KS + kh) * KS + kw] = fmaf(g[co][o], win[o * S + kw], acc[co][(kd * KS + kh) * KS + kw]);
          }
      }
    }
  }
  // cross-lane reduction, one partial per (chunk, co, ci, tap)
#pragma unroll
  for (int co = 0; co < CO_B; ++co)
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const float r = wave_sum(acc[co][t]);
      if (lane == 0 && my_ci < a.Cin && co_base + co < a.Cout)
        a.ws[(((size_t)blockIdx.x * a.Cout + co_base + co) * a.Cin + my_ci) * TAPS + t] = r;
    }
}

// 1x1: dw[co][ci] = sum_v dy[co][v] * T(x)[ci][v].  Block: 256 threads, 4 voxels each; CI_B x CO_B tile.
struct BwPwArgs {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ dy;
  float* __restrict__ ws;   // [nchunks][Cout][Cin]
  int Cin, Cout;
  size_t V;
  size_t vox_per_chunk;     // multiple of 1024
  int xb, dyb;
};

template <int CI_B, int CO_B>
__global__ __launch_bounds__(256) void conv_bwd_weight_pw_kernel(BwPwArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ci0 = blockIdx.y * CI_B, co0 = blockIdx.z * CO_B;
  float acc[CI_B][CO_B];
#pragma unroll
  for (int i = 0; i < CI_B; ++i)
#pragma unroll
    for (int o = 0; o < CO_B; ++o) acc[i][o] = 0.f;
  const size_t vbeg = (size_t)blockIdx.x * a.vox_per_chunk;
  const size_t vend = vbeg + a.vox_per_chunk < a.V ? vbeg + a.vox_per_chunk : a.V;
  const bool vec = (a.V & 3) == 0;
  for (size_t v0 = vbeg + (size_t)tid * 4; v0 < vend; v0 += 1024) {
    float xi[CI_B][4], g[CO_B][4];
#pragma unroll
    for (int i = 0; i < CI_B; ++i) {
      const int ci = min(ci0 + i, a.Cin - 1);
      const float* p = dpi_at(a.x, (size_t)ci * a.V + v0, a.xb);
      const Chain t = load_chain(a.chain, ci);
      if (vec) {
        const float4 f = dpi_ld4(p, 0, a.xb, false);
        xi[i][0] = f.x; xi[i][1] = f.y; xi[i][2] = f.z; xi[i][3] = f.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) xi[i][k] = v0 + k < a.V ? dpi_ld(p, k, a.xb) : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) xi[i][k] = (v0 + k < a.V) ? apply_chain(t, xi[i][k]) : 0.f;
    }
#pragma unroll
    for (int o = 0; o < CO_B; ++o) {
      const int co = min(co0 + o, a.Cout - 1);
      const float* p = dpi_at(a.dy, (size_t)co * a.V + v0, a.dyb);
      if (vec) {
        const float4 f = dpi_ld4(p, 0, a.dyb, false);
        g[o][0] = f.x; g[o][1] = f.y; g[o][2] = f.z; g[o][3] = f.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) g[o][k] = v0 + k < a.V ? dpi_ld(p, k, a.dyb) : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < CI_B; ++i)
#pragma unroll
      for (int o = 0; o < CO_B; ++o)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][o] = fmaf(xi[i][k], g[o][k], acc[i][o]);
  }
  __shared__ float red[4][CI_B * CO_B];
#pragma unroll
  for (int i = 0; i < CI_B; ++i)
#pragma unroll
    for (int o = 0; o < CO_B; ++o) {
      const float r = wave_sum(acc[i][o]);
      if (lane == 0) red[wid][i * CO_B + o] = r;
    }
  __syncthreads();
  if (tid < CI_B * CO_B) {
    const int i = tid / CO_B, o = tid % CO_B;
    if (ci0 + i < a.Cin && co0 + o < a.Cout)
      a.ws[((size_t)blockIdx.x * a.Cout + co0 + o) * a.Cin + ci0 + i] =
          red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
  }
}

// 8 lanes per output element: lane p sums chunks p, p+8, ... in order, then a fixed xor tree (deterministic)
__global__ void reduce_chunks_kernel(const float* __restrict__ ws, float* __restrict__ out, size_t n, int nchunks) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = gid >> 3;
  const int part = gid & 7;
  float s = 0.f;
  if (i < n)
    for (int c = part; c < nchunks; c += 8) s += ws[(size_t)c * n + i];
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
  if (i < n && part == 0) out[i] = s;
}

struct BwPlan { int nchunks, tiles_per_chunk, ntiles, ntd, nth, ntw; size_t vox_per_chunk; };

BwPlan plan(const dpi_conv_desc* d) {
  BwPlan p{};
  int Do, Ho, Wo;
  dpi_conv_out_dims(d, &Do, &Ho, &Wo);
  const size_t per = (size_t)d->Cout * d->Cin * d->kd * d->k * d->k;
  const size_t max_chunks_mem = per ? ((size_t)32 << 20) / per : 1;   // <= 128 MiB of partials
  if (d->k == 1) {
    const size_t V = (size_t)Do * Ho * Wo;
    const size_t units = cdivz(V, 1024);
    const size_t blocks_other = (size_t)cdiv(d->Cin, 4) * cdiv(d->Cout, 8);
    size_t want = cdivz(1024, blocks_other);
    if (want > units) want = units;
    if (want > max_chunks_mem) want = max_chunks_mem;
    if (want < 1) want = 1;
    const size_t upc = cdivz(units, want);
    p.vox_per_chunk = upc * 1024;
    p.nchunks = (int)cdivz(units, upc);
    return p;
  }
  const int tz = d->kd == 3 ? 4 : 1, ty = d->kd == 3 ? 8 : 32, tw = d->stride == 1 ? 32 : 16;
  p.ntd = cdiv(Do, tz); p.nth = cdiv(Ho, ty); p.ntw = cdiv(Wo, tw);
  p.ntiles = p.ntd * p.nth * p.ntw;
  const size_t blocks_other = (size_t)cdiv(d->Cin, 4) * cdiv(d->Cout, 4);
  size_t want = cdivz(1024, blocks_other);
  if (want > (size_t)p.ntiles) want = p.ntiles;
  if (want > max_chunks_mem) want = max_chunks_mem;
  if (want < 1) want = 1;
  p.tiles_per_chunk = (int)cdivz(p.ntiles, want);
  p.nchunks = cdiv(p.ntiles, p.tiles_per_chunk);
  return p;
}

}  // namespace

extern "C" size_t dpi_conv_bwd_weight_ws_floats(const dpi_conv_desc* d) {
  if (dpi_check_conv_desc(d) != DPI_OK) return 0;
  if (dpi_conv_bf16_bww_s2_usable(d)) {       // whether it runs depends on the chain and the alignment given at launch: size for both
    dpi_conv_desc f = *d;
    f.precision = 0;
    return std::max(dpi_conv_bf16_bww_s2_ws_floats(d), dpi_conv_bwd_weight_ws_floats(&f));
  }
  if (dpi_conv_bf16_bww_usable(d)) {          // the fp32 kernels stay the fallback for unaligned views: size for both
    dpi_conv_desc f = *d;
    f.precision = 0;
    return std::max(dpi_conv_bf16_bww_ws_floats(d), dpi_conv_bwd_weight_ws_floats(&f));
  }
  if (bw_use_mfma(d)) return dpi_conv_bwd_weight_mfma_ws_floats(d);
  // whether the swapped MFMA path runs depends on the chain given at launch: size for either
  const size_t sw = bw_use_mfma_swapped(d, nullptr) ? dpi_conv_bwd_weight_mfma_ws_floats(d) : 0;
  if (bw_use_smallco(d)) return std::max(sw, dpi_conv_bwd_weight_smallco_ws_floats(d));
  if (d->k == 1 && d->Cout >= g_bw_mfma_min_cout) return dpi_conv_pw_bwd_weight_mfma_ws_floats(d);
  const BwPlan p = plan(d);
  return std::max(sw, (size_t)p.nchunks * d->Cout * d->Cin * d->kd * d->k * d->k);
}

extern "C" int dpi_conv_bwd_weight(const dpi_conv_desc* d, const float* x, const float* x_chain, const float* dy,
                                   float* dw, float* ws, size_t ws_floats, void* stream) {
  if (int e = dpi_check_conv_desc(d)) return e;
  DPI_REQUIRE(x && dy && dw && ws, "conv_bwd_weight: null argument");
  DPI_REQUIRE((d->k == 1 || d->k == 3) && (d->kd == d->k || d->kd == 1) && (d->stride == 1 || d->stride == 2),
              "conv_bwd_weight: unsupported k=%d kd=%d stride=%d", d->k, d->kd, d->stride);
  hipStream_t st = (hipStream_t)stream;
  // staging loads of 4 values: 16 bytes from an fp32 tensor, 8 from a bf16 one
  const uintptr_t misal = ((uintptr_t)x & ((d->io & DPI_IO_X_BF16) ? 7 : 15)) | ((uintptr_t)dy & ((d->io & DPI_IO_DY_BF16) ? 7 : 15));
  if (dpi_conv_bf16_bww_s2_usable(d) && x_chain == nullptr && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0) {
    if (ws_floats < dpi_conv_bf16_bww_s2_ws_floats(d)) {
      dpi_set_error("conv_bwd_weight: workspace %zu < %zu floats", ws_floats, dpi_conv_bf16_bww_s2_ws_floats(d));
      return DPI_E_WORKSPACE;
    }
    return dpi_conv_bf16_bww_s2_run(d, x, dy, dw, ws, st);
  }
  if (dpi_conv_bf16_bww_usable(d) && misal == 0) {
    if (ws_floats < dpi_conv_bf16_bww_ws_floats(d)) {
      dpi_set_error("conv_bwd_weight: workspace %zu < %zu floats", ws_floats, dpi_conv_bf16_bww_ws_floats(d));
      return DPI_E_WORKSPACE;
    }
    return dpi_conv_bf16_bww_run(d, x, x_chain, dy, dw, ws, st);
  }
  if (bw_use_mfma(d) || bw_use_mfma_swapped(d, x_chain)) {
    if (ws_floats < dpi_conv_bwd_weight_mfma_ws_floats(d)) {
      dpi_set_error("conv_bwd_weight: workspace %zu < %zu floats", ws_floats, dpi_conv_bwd_weight_mfma_ws_floats(d));
      return DPI_E_WORKSPACE;
    }
    return dpi_conv_bwd_weight_mfma_run(d, x, x_chain, dy, dw, ws, st);
  }
  if (bw_use_smallco(d)) {
    if (ws_floats < dpi_conv_bwd_weight_smallco_ws_floats(d)) {
      dpi_set_error("conv_bwd_weight: workspace %zu < %zu floats", ws_floats, dpi_conv_bwd_weight_smallco_ws_floats(d));
      return DPI_E_WORKSPACE;
    }
    return dpi_conv_bwd_weight_smallco_run(d, x, x_chain, dy, dw, ws, st);
  }
  if (d->k == 1 && d->Cout >= g_bw_mfma_min_cout) {
    if (ws_floats < dpi_conv_pw_bwd_weight_mfma_ws_floats(d)) {
      dpi_set_error("conv_bwd_weight: workspace %zu < %zu floats", ws_floats, dpi_conv_pw_bwd_weight_mfma_ws_floats(d));
      return DPI_E_WORKSPACE;
    }
    return dpi_conv_pw_bwd_weight_mfma_run(d, x, x_chain, dy, dw, ws, st);
  }
  const BwPlan p = plan(d);
  const size_t per = (size_t)d->Cout * d->Cin * d->kd * d->k * d->k;
  if (ws_floats < per * p.nchunks) {
    dpi_set_error("conv_bwd_weight: workspace %zu < %zu floats", ws_floats, per * p.nchunks);
    return DPI_E_WORKSPACE;
  }
  int Do, Ho, Wo;
  dpi_conv_out_dims(d, &Do, &Ho, &Wo);
  if (d->k == 1) {
    BwPwArgs a{x, x_chain, dy, ws, d->Cin, d->Cout, (size_t)Do * Ho * Wo, p.vox_per_chunk, (d->io & DPI_IO_X_BF16) != 0, (d->io & DPI_IO_DY_BF16) != 0};
    dim3 grid(p.nchunks, cdiv(d->Cin, 4), cdiv(d->Cout, 8));
    conv_bwd_weight_pw_kernel<4, 8><<<grid, 256, 0, st>>>(a);
  } else {
    BwArgs a{x, x_chain, dy, ws, d->Cin, d->Cout, d->D, d->H, d->W, Do, Ho, Wo,
             p.ntd, p.nth, p.ntw, p.ntiles, p.tiles_per_chunk, (d->io & DPI_IO_X_BF16) != 0, (d->io & DPI_IO_DY_BF16) != 0};
    dim3 grid(p.nchunks, cdiv(d->Cin, 4), cdiv(d->Cout, 4));
    if (d->kd == 3) {
      if (d->stride == 1) conv_bwd_weight_kernel<3, 1, 4, 4, 4, 8><<<grid, 256, 0, st>>>(a);
      else conv_bwd_weight_kernel<3, 2, 4, 2, 4, 8><<<grid, 256, 0, st>>>(a);
    } else {
      if (d->stride == 1) conv_bwd_weight_kernel<1, 1, 4, 4, 1, 32><<<grid, 256, 0, st>>>(a);
      else conv_bwd_weight_kernel<1, 2, 4, 2, 1, 32><<<grid, 256, 0, st>>>(a);
    }
  }
  if (int e = dpi_check_launch("conv_bwd_weight")) return e;
  dpi_reduce_chunks(ws, dw, per, p.nchunks, st);
  return dpi_check_launch("reduce_chunks");
}
