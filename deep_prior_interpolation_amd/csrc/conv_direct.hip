// Direct (stencil) convolution for k=3 in H/W (kd = 3 or 1), stride 1 or 2, zero padding 1, fp32.
//
// One workgroup = 256 threads computes an output tile TZ x TY x (TX*OW) for CO_B output channels.
// Input channels are streamed through LDS in chunks of CI_B halo tiles; the optional per-channel
// "chain" (BatchNorm-apply + LeakyReLU of the producer) is applied while staging, so the activation
// tensor is read exactly once in its raw form.  Weights are wave-uniform and are fetched with scalar
// loads (SGPR operands of v_fmac_f32); each thread keeps CO_B x OW accumulators and a
// kd x 3 x ((OW-1)*S+3) register window per input channel.  The epilogue adds the bias, optionally
// accumulates into the destination (backward-data), stores, and reduces per-channel
// {sum, sum^2} in double precision for the BatchNorm that follows (two-stage, deterministic).
//
// The same kernel serves backward-data of stride-1 convolutions (FLIP: taps reversed, channel roles
// swapped through the weight strides).
#include "common.h"

namespace {

struct ConvArgs {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ w;
  const float* __restrict__ bias;
  float* __restrict__ y;
  double* __restrict__ partials;
  int Cin, Cout;
  int D, H, W;        // input
  int Do, Ho, Wo;     // output
  int ntd, nth, ntw;  // tiles
  long w_out_stride, w_in_stride;
  int accumulate;
  int xb, yb;         // storage type of x / y in HBM: 1 = bf16 (dpi_conv_desc.io), 0 = fp32
};

template <int KD, int S, int CO_B, int OW, int TZ, int TY, int TX, int CI_B, bool FLIP>
__global__ __launch_bounds__(TZ* TY* TX) void conv_direct_kernel(ConvArgs a) {
  constexpr int KS = 3;
  constexpr int NT = TZ * TY * TX;
  constexpr int SD = (KD > 1) ? S : 1;
  constexpr int ID = (TZ - 1) * SD + KD;
  constexpr int IH = (TY - 1) * S + KS;
  constexpr int IW = (TX * OW - 1) * S + KS;
  constexpr int IWP = (IW + 3) & ~3;
  constexpr int TILE = ID * IH * IW;        // staged elements per channel
  constexpr int E = (TILE + NT - 1) / NT;   // per thread
  constexpr int CH_LDS = ID * IH * IWP;
  constexpr int WIN = (OW - 1) * S + KS;
  constexpr int TAPS = KD * KS * KS;
  constexpr int PD = (KD - 1) / 2;

  __shared__ __attribute__((aligned(16))) float lds[CI_B * CH_LDS];

  const int tid = threadIdx.x;
  const int tx = tid % TX, ty = (tid / TX) % TY, tz = tid / (TX * TY);
  int bt = blockIdx.x;
  const int tw_i = bt % a.ntw; bt /= a.ntw;
  const int th_i = bt % a.nth; bt /= a.nth;
  const int td_i = bt;
  const int co_base = blockIdx.y * CO_B;

  const int od0 = td_i * TZ, oh0 = th_i * TY, ow0 = tw_i * TX * OW;
  const int id0 = od0 * SD - PD, ih0 = oh0 * S - 1, iw0 = ow0 * S - 1;
  const size_t V = (size_t)a.D * a.H * a.W;

  // per-thread staging slots (identical for every channel chunk)
  int goff[E], loff[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int idx = tid + e * NT;
    const int col = idx % IW, row = idx / IW;
    const int hy = row % IH, dz = row / IH;
    const int gd = id0 + dz, gh = ih0 + hy, gw = iw0 + col;
    const bool ok = idx < TILE && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
    goff[e] = ok ? (gd * a.H + gh) * a.W + gw : -1;
    loff[e] = idx < TILE ? (dz * IH + hy) * IWP + col : -1;
  }

  float acc[CO_B][OW];
#pragma unroll
  for (int c = 0; c < CO_B; ++c)
#pragma unroll
    for (int o = 0; o < OW; ++o) acc[c][o] = 0.f;

  const int lbase = ((tz * SD) * IH + ty * S) * IWP + tx * OW * S;

  for (int ci0 = 0; ci0 < a.Cin; ci0 += CI_B) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CI_B; ++c) {
      const int ci = ci0 + c;
      if (ci < a.Cin) {
        const float* __restrict__ xc = dpi_at(a.x, (size_t)ci * V, a.xb);
        const Chain t = load_chain(a.chain, ci);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if (loff[e] >= 0) {
            float v = 0.f;
            if (goff[e] >= 0) v = apply_chain(t, dpi_ld(xc, goff[e], a.xb));
            lds[c * CH_LDS + loff[e]] = v;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < CI_B; ++c) {
      const int ci = ci0 + c;
      if (ci >= a.Cin) break;
      float win[KD][KS][WIN];
#pragma unroll
      for (int kd = 0; kd < KD; ++kd)
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
          for (int i = 0; i < WIN; ++i) win[kd][kh][i] = lds[c * CH_LDS + lbase + (kd * IH + kh) * IWP + i];
#pragma unroll
      for (int co = 0; co < CO_B; ++co) {
        const int cog = min(co_base + co, a.Cout - 1);
        const float* __restrict__ wp = a.w + cog * a.w_out_stride + ci * a.w_in_stride;
#pragma unroll
        for (int kd = 0; kd < KD; ++kd)
#pragma unroll
          for (int kh = 0; kh < KS; ++kh)
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
              const int tap = (kd * KS + kh) * KS + kw;
              const float wv = wp[FLIP ? (TAPS - 1 - tap) : tap];
#pragma unroll
              for (int o = 0; o < OW; ++o) acc[co][o] = fmaf(wv, win[kd][kh][o * S + kw], acc[co][o]);
            }
      }
    }
  }

  // ---------------- epilogue ----------------
  const int od = od0 + tz, oh = oh0 + ty, ow = ow0 + tx * OW;
  const bool row_ok = od < a.Do && oh < a.Ho;
  const size_t Vo = (size_t)a.Do * a.Ho * a.Wo;
  const size_t obase = ((size_t)od * a.Ho + oh) * a.Wo + ow;
  const bool vec_ok = (OW == 4) && row_ok && (ow + 3 < a.Wo) && ((a.Wo & 3) == 0);
  __shared__ double red[(NT / 64) * CO_B * 2];
#pragma unroll
  for (int co = 0; co < CO_B; ++co) {
    const int cog = co_base + co;
    const bool cok = cog < a.Cout;
    const float b = (a.bias && cok) ? a.bias[cog] : 0.f;
    double s = 0.0, q = 0.0;
    float* yp = dpi_at(a.y, (size_t)min(cog, a.Cout - 1) * Vo + obase, a.yb);
    float v[OW];
#pragma unroll
    for (int o = 0; o < OW; ++o) v[o] = acc[co][o] + b;
    if (cok && row_ok) {
      if (vec_ok) {
        if (a.accumulate) {
          const float4 old = dpi_ld4(yp, 0, a.yb, false);
          v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
        }
#pragma unroll
        for (int o = 0; o < OW; ++o) v[o] = dpi_stored(v[o], a.yb);       // statistics describe what is stored
        dpi_st4(yp, 0, make_float4(v[0], v[1], v[2], v[3]), a.yb, false);
#pragma unroll
        for (int o = 0; o < OW; ++o) { s += v[o]; q += (double)v[o] * v[o]; }
      } else {
#pragma unroll
        for (int o = 0; o < OW; ++o)
          if (ow + o < a.Wo) {
            if (a.accumulate) v[o] += dpi_ld(yp, o, a.yb);
            v[o] = dpi_stored(v[o], a.yb);
            dpi_st(yp, o, v[o], a.yb);
            s += v[o]; q += (double)v[o] * v[o];
          }
      }
    }
    if (a.partials) {
      s = wave_sum(s);
      q = wave_sum(q);
      if ((tid & 63) == 0) {
        red[((tid >> 6) * CO_B + co) * 2 + 0] = s;
        red[((tid >> 6) * CO_B + co) * 2 + 1] = q;
      }
    }
  }
  if (a.partials) {
    __syncthreads();
    if (tid < CO_B * 2) {
      const int co = tid >> 1, which = tid & 1;
      double r = 0.0;
#pragma unroll
      for (int wv = 0; wv < NT / 64; ++wv) r += red[(wv * CO_B + co) * 2 + which];
      if (co_base + co < a.Cout)
        a.partials[((size_t)blockIdx.x * a.Cout + co_base + co) * 2 + which] = r;
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// 1x1x1 convolution (channel GEMM), VALU variant: each thread owns 4 consecutive voxels x CO_B channels,
// streams input channels straight from global memory (no reuse across threads, so no LDS), weights in
// SGPRs.  (The MFMA variant lives in conv1x1_mfma.hip.)
struct PwArgs {
  const float* __restrict__ x;
  const float* __restrict__ chain;
  const float* __restrict__ w;
  const float* __restrict__ bias;
  float* __restrict__ y;
  double* __restrict__ partials;
  int Cin, Cout;
  size_t V;
  long w_out_stride, w_in_stride;
  int accumulate;
  int xb, yb;
};

template <int CO_B>
__global__ __launch_bounds__(256) void conv_pw_kernel(PwArgs a) {
  const int tid = threadIdx.x;
  const size_t v0 = ((size_t)blockIdx.x * 256 + tid) * 4;
  const int co_base = blockIdx.y * CO_B;
  const bool full = v0 + 3 < a.V && (a.V & 3) == 0;
  float acc[CO_B][4];
#pragma unroll
  for (int c = 0; c < CO_B; ++c)
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[c][o] = 0.f;
#pragma unroll 2
  for (int ci = 0; ci < a.Cin; ++ci) {
    const float* __restrict__ xc = dpi_at(a.x, (size_t)ci * a.V, a.xb);
    const Chain t = load_chain(a.chain, ci);
    float in[4];
    if (full) {
      const float4 f = dpi_ld4(xc, v0, a.xb, false);
      in[0] = f.x; in[1] = f.y; in[2] = f.z; in[3] = f.w;
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o) in[o] = (v0 + o < a.V) ? dpi_ld(xc, v0 + o, a.xb) : 0.f;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) in[o] = apply_chain(t, in[o]);
#pragma unroll
    for (int co = 0; co < CO_B; ++co) {
      const int cog = min(co_base + co, a.Cout - 1);
      const float wv = a.w[cog * a.w_out_stride + ci * a.w_in_stride];
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[co][o] = fmaf(wv, in[o], acc[co][o]);
    }
  }
  __shared__ double red[4 * CO_B * 2];
#pragma unroll
  for (int co = 0; co < CO_B; ++co) {
    const int cog = co_base + co;
    const bool cok = cog < a.Cout;
    const float b = (a.bias && cok) ? a.bias[cog] : 0.f;
    float* yp = dpi_at(a.y, (size_t)min(cog, a.Cout - 1) * a.V + v0, a.yb);
    double s = 0.0, q = 0.0;
    float v[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) v[o] = acc[co][o] + b;
    if (cok) {
      if (full) {
        if (a.accumulate) {
          const float4 old = dpi_ld4(yp, 0, a.yb, false);
          v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) v[o] = dpi_stored(v[o], a.yb);
        dpi_st4(yp, 0, make_float4(v[0], v[1], v[2], v[3]), a.yb, false);
#pragma unroll
        for (int o = 0; o < 4; ++o) { s += v[o]; q += (double)v[o] * v[o]; }
      } else {
#pragma unroll
        for (int o = 0; o < 4; ++o)
          if (v0 + o < a.V) {
            if (a.accumulate) v[o] += dpi_ld(yp, o, a.yb);
            v[o] = dpi_stored(v[o], a.yb);
            dpi_st(yp, o, v[o], a.yb);
            s += v[o]; q += (double)v[o] * v[o];
          }
      }
    }
    if (a.partials) {
      s = wave_sum(s);
      q = wave_sum(q);
      if ((tid & 63) == 0) {
        red[((tid >> 6) * CO_B + co) * 2 + 0] = s;
        red[((tid >> 6) * CO_B + co) * 2 + 1] = q;
      }
    }
  }
  if (a.partials) {
    __syncthreads();
    if (tid < CO_B * 2) {
      const int co = tid >> 1, which = tid & 1;
      double r = 0.0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) r += red[(wv * CO_B + co) * 2 + which];
      if (co_base + co < a.Cout)
        a.partials[((size_t)blockIdx.x * a.Cout + co_base + co) * 2 + which] = r;
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// Backward-data of the stride-2 convolution (transposed conv, gather form).  Each thread produces 4
// consecutive input voxels for CI_B input channels; taps are selected by parity.  Small share of the
// work (4 layers), kept simple.
struct BwdS2Args {
  const float* __restrict__ dy;
  const float* __restrict__ w;   // [Cout][Cin][kd][3][3]
  float* __restrict__ dx;
  int Cin, Cout;
  int D, H, W, Do, Ho, Wo;
  int kd;
  int accumulate;
  int dyb, dxb;       // storage type of dy / dx: 1 = bf16
};

template <int CI_B>
__global__ __launch_bounds__(256) void conv_bwd_data_s2_kernel(BwdS2Args a) {
  const size_t V = (size_t)a.D * a.H * a.W, Vo = (size_t)a.Do * a.Ho * a.Wo;
  const size_t vox = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int ci_base = blockIdx.y * CI_B;
  if (vox >= V) return;
  const int iw = vox % a.W, ih = (vox / a.W) % a.H, id = vox / ((size_t)a.W * a.H);
  const int KD = a.kd, PD = (KD - 1) / 2, SD = KD > 1 ? 2 : 1;
  const int taps = KD * 9;
  float acc[CI_B];
#pragma unroll
  for (int c = 0; c < CI_B; ++c) acc[c] = 0.f;
  for (int kd = 0; kd < KD; ++kd) {
    const int nd = id + PD - kd;
    if (nd < 0 || (nd % SD) != 0) continue;
    const int od = nd / SD;
    if (od >= a.Do) continue;
    for (int kh = 0; kh < 3; ++kh) {
      const int nh = ih + 1 - kh;
      if (nh < 0 || (nh & 1)) continue;
      const int oh = nh >> 1;
      if (oh >= a.Ho) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int nw = iw + 1 - kw;
        if (nw < 0 || (nw & 1)) continue;
        const int ow = nw >> 1;
        if (ow >= a.Wo) continue;
        const int tap = (kd * 3 + kh) * 3 + kw;
        const size_t o = ((size_t)od * a.Ho + oh) * a.Wo + ow;
        for (int co = 0; co < a.Cout; ++co) {
          const float g = dpi_ld(a.dy, (size_t)co * Vo + o, a.dyb);
#pragma unroll
          for (int c = 0; c < CI_B; ++c) {
            const int ci = min(ci_base + c, a.Cin - 1);
            acc[c] = fmaf(g, a.w[((size_t)co * a.Cin + ci) * taps + tap], acc[c]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CI_B; ++c) {
    const int ci = ci_base + c;
    if (ci < a.Cin) {
      const size_t o = (size_t)ci * V + vox;
      dpi_st(a.dx, o, a.accumulate ? dpi_ld(a.dx, o, a.dxb) + acc[c] : acc[c], a.dxb);
    }
  }
}

// choose the per-thread output-channel block: the fewest blocks of <=16 channels, then the least padding
const int kCoChoices[] = {1, 2, 4, 6, 8, 9, 12, 13, 14, 15, 16};
int pick_co_b(int Cout) {
  int best = 16, best_pad = 1 << 30;
  const int nblk_min = cdiv(Cout, 16);
  for (int c : kCoChoices) {
    const int nb = cdiv(Cout, c);
    if (nb != nblk_min) continue;
    const int pad = nb * c - Cout;
    if (pad < best_pad) { best_pad = pad; best = c; }
  }
  return best;
}

struct Geo { int tz, ty, txow; };
Geo conv_geo(int kd, int stride) {
  if (kd == 3) return stride == 1 ? Geo{4, 8, 32} : Geo{4, 8, 16};
  return stride == 1 ? Geo{1, 32, 32} : Geo{1, 32, 16};
}

template <int KD, int S, bool FLIP, int CO_B>
void launch_geo(const ConvArgs& a, dim3 grid, hipStream_t st) {
  if constexpr (KD == 3 && S == 1) conv_direct_kernel<3, 1, CO_B, 4, 4, 8, 8, 4, FLIP><<<grid, 256, 0, st>>>(a);
  else if constexpr (KD == 3 && S == 2) conv_direct_kernel<3, 2, CO_B, 2, 4, 8, 8, 2, FLIP><<<grid, 256, 0, st>>>(a);
  else if constexpr (KD == 1 && S == 1) conv_direct_kernel<1, 1, CO_B, 4, 1, 32, 8, 8, FLIP><<<grid, 256, 0, st>>>(a);
  else conv_direct_kernel<1, 2, CO_B, 2, 1, 32, 8, 4, FLIP><<<grid, 256, 0, st>>>(a);
}

template <int KD, int S, bool FLIP>
void launch_co(const ConvArgs& a, int co_b, dim3 grid, hipStream_t st) {
  switch (co_b) {
    case 1: launch_geo<KD, S, FLIP, 1>(a, grid, st); break;
    case 2: launch_geo<KD, S, FLIP, 2>(a, grid, st); break;
    case 4: launch_geo<KD, S, FLIP, 4>(a, grid, st); break;
    case 6: launch_geo<KD, S, FLIP, 6>(a, grid, st); break;
    case 8: launch_geo<KD, S, FLIP, 8>(a, grid, st); break;
    case 9: launch_geo<KD, S, FLIP, 9>(a, grid, st); break;
    case 12: launch_geo<KD, S, FLIP, 12>(a, grid, st); break;
    case 13: launch_geo<KD, S, FLIP, 13>(a, grid, st); break;
    case 14: launch_geo<KD, S, FLIP, 14>(a, grid, st); break;
    case 15: launch_geo<KD, S, FLIP, 15>(a, grid, st); break;
    default: launch_geo<KD, S, FLIP, 16>(a, grid, st); break;
  }
}

template <int CO_B>
void launch_pw(const PwArgs& a, dim3 grid, hipStream_t st) {
  conv_pw_kernel<CO_B><<<grid, 256, 0, st>>>(a);
}
void launch_pw_co(const PwArgs& a, int co_b, dim3 grid, hipStream_t st) {
  switch (co_b) {
    case 1: launch_pw<1>(a, grid, st); break;
    case 2: launch_pw<2>(a, grid, st); break;
    case 4: launch_pw<4>(a, grid, st); break;
    case 6: launch_pw<6>(a, grid, st); break;
    case 8: launch_pw<8>(a, grid, st); break;
    case 9: launch_pw<9>(a, grid, st); break;
    case 12: launch_pw<12>(a, grid, st); break;
    case 13: launch_pw<13>(a, grid, st); break;
    case 14: launch_pw<14>(a, grid, st); break;
    case 15: launch_pw<15>(a, grid, st); break;
    default: launch_pw<16>(a, grid, st); break;
  }
}

int check_desc(const dpi_conv_desc* d) { return dpi_check_conv_desc(d); }

}  // namespace

int dpi_check_conv_desc(const dpi_conv_desc* d) {
  DPI_REQUIRE(d, "conv: null descriptor");
  DPI_REQUIRE(d->size == (int)sizeof(dpi_conv_desc), "conv: descriptor size field is %d, this library's dpi_conv_desc has %d bytes (stale binding? "
              "set desc.size = sizeof(dpi_conv_desc) of the header you compiled against; dpi_conv_desc_size() gives the library's)", d->size, (int)sizeof(dpi_conv_desc));
  DPI_REQUIRE(d->Cin > 0 && d->Cout > 0 && d->D > 0 && d->H > 0 && d->W > 0, "conv: non-positive dims");
  DPI_REQUIRE(d->k == 1 || d->k == 3, "conv: k must be 1 or 3 (got %d)", d->k);
  DPI_REQUIRE(d->kd == d->k || d->kd == 1, "conv: kd must be k (3-D) or 1 (2-D), got %d", d->kd);
  DPI_REQUIRE(d->kd == d->k || d->D == 1, "conv: 2-D kernels need D == 1");
  DPI_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride must be 1 or 2 (got %d)", d->stride);
  DPI_REQUIRE(d->k == 3 || d->stride == 1, "conv: 1x1 convolution supports stride 1 only");
  // the stencil kernels address a channel through 32-bit BYTE offsets (buffer loads): 4 * D*H*W must stay below 2^31
  DPI_REQUIRE((size_t)d->D * d->H * d->W < (1ull << 29), "conv: spatial volume %d x %d x %d exceeds the 32-bit byte offsets of the kernels (2^29 voxels per patch)",
              d->D, d->H, d->W);
  DPI_REQUIRE(d->precision >= 0 && d->precision <= 2, "conv: precision must be 0 (fp32), 1 (bf16 operands) or 2 (three-term bf16 split), got %d", d->precision);
  DPI_REQUIRE((d->io & ~(DPI_IO_X_BF16 | DPI_IO_Y_BF16 | DPI_IO_DY_BF16 | DPI_IO_DX_BF16)) == 0, "conv: unknown storage-type bits in io = %d", d->io);
  // 8-byte pieces of bf16 rows go through 32-bit ELEMENT offsets of up to 16 channels (backward-weight): the same 2^29-voxel bound covers them
  return DPI_OK;
}

// MFMA stencil path (conv_mfma.hip): k = 3, stride 1, enough output channels to fill a 16-row MFMA tile
int dpi_conv_mfma_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias, float* y,
                      double* partials, bool flip, int accumulate, float* ws, size_t ws_floats, hipStream_t st, const MfmaSecond* sec = nullptr);
size_t dpi_conv_mfma_ws_floats(const dpi_conv_desc* d, bool flip);
bool dpi_conv_mfma_second_ok(const dpi_conv_desc* d, bool flip, int C2, bool have_ws);
void dpi_conv_pw_mfma_plan(size_t V, int cout, int* vox_per_block, int* mt);
bool dpi_conv_fewco_usable(const dpi_conv_desc* d);
bool dpi_conv_q4_usable(const dpi_conv_desc* d, bool flip);
int dpi_conv_q4_tiles(const dpi_conv_desc* d, int* ntd, int* nth, int* ntw);
int dpi_conv_q4_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias, float* y,
                    double* partials, bool flip, int accumulate, hipStream_t st);
int dpi_conv_fewco_tiles(const dpi_conv_desc* d, int* ntd, int* nth, int* ntw);
int dpi_conv_fewco_mfma_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias, float* y,
                            double* partials, hipStream_t st);
static int g_fewco_mfma = 1;
extern "C" void dpi_set_fewco_mfma(int on) { g_fewco_mfma = on; }
int dpi_conv_pw_mfma_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias, float* y,
                         double* partials, bool flip, int accumulate, hipStream_t st);
int dpi_conv_bwd_data_s2_mfma_run(const dpi_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate, hipStream_t st);
void dpi_mfma_variant(const dpi_conv_desc* d, int cout, int* nr, int* nh);
int dpi_mfma_tiles(const dpi_conv_desc* d, int nr, int nh, int* ntd, int* nth, int* ntw);
bool dpi_mfma_half_tile(const dpi_conv_desc* d, bool flip);
bool dpi_conv_bf16_usable(const dpi_conv_desc* d, bool flip);
// conv_bf16_mfma.hip: 3x3x3 stride-2 forward, bf16 x and y, bf16 arithmetic
bool dpi_conv_bf16_s2_usable(const dpi_conv_desc* d);
int dpi_conv_bf16_s2_stat_blocks(const dpi_conv_desc* d);
bool dpi_conv_bf16_s2_bwd_usable(const dpi_conv_desc* d);
int dpi_conv_bf16_s2_bwd_run(const dpi_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate, hipStream_t st);
int dpi_conv_bf16_s2_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias, float* y, double* partials,
                         hipStream_t st);
int dpi_conv_bf16_stat_blocks(const dpi_conv_desc* d);
int dpi_conv_bf16_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias, float* y,
                      double* partials, bool flip, int accumulate, hipStream_t st, const MfmaSecond* sec = nullptr);
bool dpi_conv_bf16_second_ok(const dpi_conv_desc* d, bool flip);
static int g_mfma_min_cout = 8;
extern "C" void dpi_set_mfma_min_cout(int n) { g_mfma_min_cout = n; }

void dpi_conv_out_dims(const dpi_conv_desc* d, int* Do, int* Ho, int* Wo) {
  const int p = (d->k - 1) / 2, pd = (d->kd - 1) / 2, sd = d->kd > 1 ? d->stride : 1;
  *Do = (d->D + 2 * pd - d->kd) / sd + 1;
  *Ho = (d->H + 2 * p - d->k) / d->stride + 1;
  *Wo = (d->W + 2 * p - d->k) / d->stride + 1;
}

extern "C" int dpi_conv_fwd_stat_blocks(const dpi_conv_desc* d) {
  if (check_desc(d) != DPI_OK) return 0;
  int Do, Ho, Wo;
  dpi_conv_out_dims(d, &Do, &Ho, &Wo);
  if (dpi_conv_bf16_s2_usable(d)) return dpi_conv_bf16_s2_stat_blocks(d);
  if (dpi_conv_bf16_usable(d, false)) return dpi_conv_bf16_stat_blocks(d);
  if (dpi_conv_q4_usable(d, false)) { int a, b, c; return dpi_conv_q4_tiles(d, &a, &b, &c); }
  if (d->k == 1 && d->Cout >= g_mfma_min_cout) {
    int vpb, mt;
    dpi_conv_pw_mfma_plan((size_t)Do * Ho * Wo, d->Cout, &vpb, &mt);
    return (int)cdivz((size_t)Do * Ho * Wo, vpb);
  }
  if (d->k == 1) return (int)cdivz((size_t)Do * Ho * Wo, 1024);
  if (d->k == 3 && d->Cout >= g_mfma_min_cout) {
    int nr, nh, a, b, c;
    dpi_mfma_variant(d, d->Cout, &nr, &nh);
    if (dpi_mfma_half_tile(d, false)) nr = 4;
    return dpi_mfma_tiles(d, nr, nh, &a, &b, &c);
  }
  if (g_fewco_mfma && !(d->io & (DPI_IO_X_BF16 | DPI_IO_Y_BF16)) && dpi_conv_fewco_usable(d)) { int a, b, c; return dpi_conv_fewco_tiles(d, &a, &b, &c); }
  const Geo g = conv_geo(d->kd, d->stride);
  return cdiv(Do, g.tz) * cdiv(Ho, g.ty) * cdiv(Wo, g.txow);
}

// which launches take the fp32-MFMA stencil path of conv_mfma.hip (the order of the tests in conv_run)
static bool takes_mfma_path(const dpi_conv_desc* d, bool flip) {
  const int cout = flip ? d->Cin : d->Cout;
  if (dpi_conv_bf16_usable(d, flip) || dpi_conv_q4_usable(d, flip) || (!flip && dpi_conv_bf16_s2_usable(d))) return false;
  return d->k == 3 && cout >= g_mfma_min_cout && (d->stride == 1 || !flip);
}

static int conv_run(const dpi_conv_desc* d, const float* x, const float* chain, const float* w, const float* bias,
                    float* y, double* partials, bool flip, int accumulate, float* ws, size_t ws_floats, hipStream_t st) {
  // For flip (backward-data of a stride-1 conv) the caller passes dy as x and swaps channel roles:
  // "Cin" of this launch = d->Cout, "Cout" = d->Cin, spatial dims unchanged.
  int Do, Ho, Wo;
  dpi_conv_out_dims(d, &Do, &Ho, &Wo);
  const int taps = d->kd * d->k * d->k;
  const int cin = flip ? d->Cout : d->Cin, cout = flip ? d->Cin : d->Cout;
  const long w_out = flip ? taps : (long)d->Cin * taps, w_in = flip ? (long)d->Cin * taps : taps;
  if (!flip && !accumulate && dpi_conv_bf16_s2_usable(d)) return dpi_conv_bf16_s2_run(d, x, chain, w, bias, y, partials, st);
  if (dpi_conv_bf16_usable(d, flip)) return dpi_conv_bf16_run(d, x, chain, w, bias, y, partials, flip, accumulate, st);
  if (dpi_conv_q4_usable(d, flip)) return dpi_conv_q4_run(d, x, chain, w, bias, y, partials, flip, accumulate, st);
  const int xb = dpi_io_in(d, flip), yb = dpi_io_out(d, flip);
  if (d->k == 3 && cout >= g_mfma_min_cout && (d->stride == 1 || !flip))
    return dpi_conv_mfma_run(d, x, chain, w, bias, y, partials, flip, accumulate, ws, ws_floats, st);
  if (d->k == 1 && cout >= g_mfma_min_cout) return dpi_conv_pw_mfma_run(d, x, chain, w, bias, y, partials, flip, accumulate, st);
  if (!flip && !accumulate && g_fewco_mfma && !xb && !yb && dpi_conv_fewco_usable(d)) return dpi_conv_fewco_mfma_run(d, x, chain, w, bias, y, partials, st);
  const int co_b = pick_co_b(cout);
  if (d->k == 1) {
    PwArgs a{x, chain, w, bias, y, partials, cin, cout, (size_t)Do * Ho * Wo, w_out, w_in, accumulate, xb, yb};
    dim3 grid((unsigned)cdivz(a.V, 1024), cdiv(cout, co_b));
    launch_pw_co(a, co_b, grid, st);
    return dpi_check_launch("conv_pw");
  }
  const Geo g = conv_geo(d->kd, d->stride);
  ConvArgs a{x, chain, w, bias, y, partials, cin, cout, d->D, d->H, d->W, Do, Ho, Wo,
             cdiv(Do, g.tz), cdiv(Ho, g.ty), cdiv(Wo, g.txow), w_out, w_in, accumulate, xb, yb};
  dim3 grid(a.ntd * a.nth * a.ntw, cdiv(cout, co_b));
  if (d->kd == 3) {
    if (d->stride == 1) { if (flip) launch_co<3, 1, true>(a, co_b, grid, st); else launch_co<3, 1, false>(a, co_b, grid, st); }
    else launch_co<3, 2, false>(a, co_b, grid, st);
  } else {
    if (d->stride == 1) { if (flip) launch_co<1, 1, true>(a, co_b, grid, st); else launch_co<1, 1, false>(a, co_b, grid, st); }
    else launch_co<1, 2, false>(a, co_b, grid, st);
  }
  return dpi_check_launch("conv_direct");
}

extern "C" size_t dpi_conv_fwd_ws_floats(const dpi_conv_desc* d) {
  if (check_desc(d) != DPI_OK) return 0;
  return takes_mfma_path(d, false) ? dpi_conv_mfma_ws_floats(d, false) : 0;
}

extern "C" size_t dpi_conv_bwd_data_ws_floats(const dpi_conv_desc* d) {
  if (check_desc(d) != DPI_OK || d->stride != 1) return 0;
  return takes_mfma_path(d, true) ? dpi_conv_mfma_ws_floats(d, true) : 0;
}

extern "C" int dpi_conv_fwd_ws(const dpi_conv_desc* d, const float* x, const float* x_chain, const float* w, const float* bias, float* y,
                               double* stat_partials, float* ws, size_t ws_floats, void* stream) {
  if (int e = check_desc(d)) return e;
  DPI_REQUIRE(x && w && y, "conv_fwd: null tensor");
  DPI_REQUIRE(ws || ws_floats == 0, "conv_fwd: workspace size without a workspace");
  return conv_run(d, x, x_chain, w, bias, y, stat_partials, false, 0, ws, ws_floats, (hipStream_t)stream);
}

extern "C" int dpi_conv_fwd(const dpi_conv_desc* d, const float* x, const float* x_chain, const float* w,
                            const float* bias, float* y, double* stat_partials, void* stream) {
  return dpi_conv_fwd_ws(d, x, x_chain, w, bias, y, stat_partials, nullptr, 0, stream);
}

extern "C" int dpi_conv_bwd_data(const dpi_conv_desc* d, const float* dy, const float* w, float* dx,
                                 int accumulate, void* stream) {
  return dpi_conv_bwd_data_ws(d, dy, w, dx, accumulate, nullptr, 0, stream);
}

// dx (+)= conv_transpose(dy3, w3) + conv_transpose(dy1, w1) for a 3x3(x3) layer d3 and a 1x1(x1) layer d1 reading the same tensor
static int g_dual = getenv("DPI_NO_DUAL") ? 0 : 1;
extern "C" void dpi_set_dual_bwd_data(int on) { g_dual = on; }
extern "C" int dpi_conv_bwd_data_dual(const dpi_conv_desc* d3, const float* dy3, const float* w3, const dpi_conv_desc* d1, const float* dy1,
                                      const float* w1, float* dx, int accumulate, float* ws, size_t ws_floats, void* stream) {
  if (int e = check_desc(d3)) return e;
  if (int e = check_desc(d1)) return e;
  DPI_REQUIRE(dy3 && w3 && dy1 && w1 && dx, "conv_bwd_data_dual: null tensor");
  DPI_REQUIRE(ws || ws_floats == 0, "conv_bwd_data_dual: workspace size without a workspace");
  DPI_REQUIRE(d3->k == 3 && d1->k == 1 && d3->stride == 1 && d1->stride == 1, "conv_bwd_data_dual: needs a 3x3(x3) and a 1x1(x1) stride-1 layer");
  DPI_REQUIRE(d3->Cin == d1->Cin && d3->D == d1->D && d3->H == d1->H && d3->W == d1->W, "conv_bwd_data_dual: the two layers read different tensors");
  hipStream_t st = (hipStream_t)stream;
  DPI_REQUIRE((d3->io & (DPI_IO_DX_BF16 | DPI_IO_X_BF16)) == (d1->io & (DPI_IO_DX_BF16 | DPI_IO_X_BF16)), "conv_bwd_data_dual: the two layers disagree on the storage type of their input");
  if (g_dual && dpi_conv_bf16_second_ok(d3, true) && (d3->io & DPI_IO_DY_BF16) == (d1->io & DPI_IO_DY_BF16)) {
    // bf16 arithmetic mode: the 1x1x1 term as extra K blocks of the bf16-MFMA kernel (conv_bf16_mfma.hip)
    const MfmaSecond sec{dy1, w1, d1->Cout, 1, (long)d1->Cin};
    return dpi_conv_bf16_run(d3, dy3, nullptr, w3, nullptr, dx, nullptr, true, accumulate, st, &sec);
  }
  if (g_dual && takes_mfma_path(d3, true) && dpi_conv_mfma_second_ok(d3, true, d1->Cout, ws != nullptr)) {
    // W2[ci][co1] = w1[co1][ci]: rows of this launch are the layers' INPUT channels
    const MfmaSecond sec{dy1, w1, d1->Cout, 1, (long)d1->Cin};
    return dpi_conv_mfma_run(d3, dy3, nullptr, w3, nullptr, dx, nullptr, true, accumulate, nullptr, 0, st, &sec);
  }
  if (int e = conv_run(d1, dy1, nullptr, w1, nullptr, dx, nullptr, true, accumulate, nullptr, 0, st)) return e;
  return conv_run(d3, dy3, nullptr, w3, nullptr, dx, nullptr, true, 1, ws, ws_floats, st);
}

extern "C" int dpi_conv_bwd_data_ws(const dpi_conv_desc* d, const float* dy, const float* w, float* dx,
                                    int accumulate, float* ws, size_t ws_floats, void* stream) {
  if (int e = check_desc(d)) return e;
  DPI_REQUIRE(dy && w && dx, "conv_bwd_data: null tensor");
  DPI_REQUIRE(ws || ws_floats == 0, "conv_bwd_data: workspace size without a workspace");
  hipStream_t st = (hipStream_t)stream;
  if (d->stride == 1) return conv_run(d, dy, nullptr, w, nullptr, dx, nullptr, true, accumulate, ws, ws_floats, st);
  // bf16 tensors in the bf16 arithmetic mode: parity-class GEMMs on the bf16 MFMA (8-byte pieces of dy, dword stores of dx)
  if (dpi_conv_bf16_s2_bwd_usable(d) && ((uintptr_t)dy & 7) == 0 && ((uintptr_t)dx & 3) == 0) return dpi_conv_bf16_s2_bwd_run(d, dy, w, dx, accumulate, st);
  if (d->Cin >= g_mfma_min_cout) return dpi_conv_bwd_data_s2_mfma_run(d, dy, w, dx, accumulate, st);
  int Do, Ho, Wo;
  dpi_conv_out_dims(d, &Do, &Ho, &Wo);
  BwdS2Args a{dy, w, dx, d->Cin, d->Cout, d->D, d->H, d->W, Do, Ho, Wo, d->kd, accumulate, (d->io & DPI_IO_DY_BF16) != 0, (d->io & DPI_IO_DX_BF16) != 0};
  const size_t V = (size_t)d->D * d->H * d->W;
  dim3 grid((unsigned)cdivz(V, 256), cdiv(d->Cin, 8));
  conv_bwd_data_s2_kernel<8><<<grid, 256, 0, st>>>(a);
  return dpi_check_launch("conv_bwd_data_s2");
}
