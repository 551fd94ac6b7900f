// Masked L1/MSE loss with fused gradient and SNR/PCORR sums, multi-tensor Adam, Philox input noise,
// overlap-add patch reassembly.
#include "common.h"

namespace {

constexpr int kLossBlocks = 1024;

// partial layout per block: {sum|d| or d^2, sum t^2, sum (t-o)^2, sum o, sum t, sum o^2, sum o*t, unused}
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ out, const float* __restrict__ img,
                                                           const float* __restrict__ mask, size_t n, int kind, float gscale,
                                                           float* __restrict__ dout, double* __restrict__ ws) {
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  const float inv_n = gscale / (float)n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float o = out[i], t = img[i], m = mask[i];
    const float d = o * m - t * m;
    float g;
    if (kind == 1) { acc[0] += (double)d * d; g = 2.f * d * m * inv_n; }
    else { acc[0] += fabsf(d); g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * m * inv_n; }
    if (dout) dout[i] = g;
    const float e = t - o;
    acc[1] += (double)t * t; acc[2] += (double)e * e; acc[3] += o; acc[4] += t;
    acc[5] += (double)o * o; acc[6] += (double)o * t;
  }
  __shared__ double sh[4];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const double r = block_sum(acc[k], sh);
    if (threadIdx.x == 0) ws[(size_t)blockIdx.x * 8 + k] = r;
  }
}

__global__ __launch_bounds__(64) void loss_final_kernel(const double* __restrict__ ws, int nblk, double n, double* __restrict__ res) {
  double acc[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += ws[(size_t)b * 8 + k];
    acc[k] = wave_sum(s);
  }
  if (threadIdx.x == 0) {
    const double loss = acc[0] / n;
    const double snr = 10.0 * log10(acc[1] / acc[2]);
    const double mo = acc[3] / n, mt = acc[4] / n;
    const double cov = acc[6] - n * mo * mt;
    const double vt = acc[1] - n * mt * mt, vo = acc[5] - n * mo * mo;
    res[0] = loss; res[1] = snr; res[2] = cov / (sqrt(vt) * sqrt(vo));
    res[3] = acc[1]; res[4] = acc[2]; res[5] = acc[3]; res[6] = acc[4]; res[7] = acc[5];
  }
}

// ---- Adam -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(const dpi_adam_tensor* __restrict__ tensors, const int64_t* __restrict__ sizes,
                                                   const float* __restrict__ step_lr, double beta1d, double beta2d, double epsd,
                                                   const int* __restrict__ active) {
  if (active && *active == 0) return;
  const dpi_adam_tensor t = tensors[blockIdx.y];
  const size_t n = (size_t)sizes[blockIdx.y];
  const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  // scalars exactly as torch.optim.Adam derives them: in double on the "host side", then rounded to fp32
  const double step = (double)step_lr[0], lr = (double)step_lr[1];
  const float beta2 = (float)beta2d, eps = (float)epsd;
  const float omb1 = (float)(1.0 - beta1d), omb2 = (float)(1.0 - beta2d);
  const float step_size = (float)(lr / (1.0 - pow(beta1d, step)));
  const float bc2s = (float)sqrt(1.0 - pow(beta2d, step));
  for (size_t i = i0; i < n; i += (size_t)gridDim.x * 1024) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k < n) {
        const float g = t.g[i + k];
        const float m0 = t.m[i + k];
        const float m = m0 + omb1 * (g - m0);                   // exp_avg.lerp_(grad, 1-beta1)
        const float v = beta2 * t.v[i + k] + (omb2 * g) * g;    // mul_(beta2).addcmul_(g, g, value=1-beta2)
        t.m[i + k] = m;
        t.v[i + k] = v;
        const float denom = sqrtf(v) / bc2s + eps;
        t.p[i + k] -= step_size * (m / denom);
      }
  }
}

// ---- Philox4x32-10 ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4(uint64_t ctr, uint64_t stream_id, uint64_t seed, float (&z)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  // Box-Muller on two pairs of uniforms in (0,1]
  const float u0 = ((c[0] >> 8) + 1) * (1.f / 16777216.f), u1 = (c[1] >> 8) * (1.f / 16777216.f);
  const float u2 = ((c[2] >> 8) + 1) * (1.f / 16777216.f), u3 = (c[3] >> 8) * (1.f / 16777216.f);
  const float r0 = sqrtf(-2.f * __logf(u0)), r1 = sqrtf(-2.f * __logf(u2));
  float s0, c0, s1, c1;
  __sincosf(6.283185307179586f * u1, &s0, &c0);
  __sincosf(6.283185307179586f * u3, &s1, &c1);
  z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}

// zregen: instead of READING z (zin) the kernel re-draws it — z = zstd * N(0,1) of the Philox stream (zseed, zstream), exactly the values
// dpi_fill_normal(mean 0) wrote — so the fixed network input never has to be streamed from HBM again (1.07 GB per iteration at the bench patch).
struct ZRegen { int on; float zstd; uint64_t zseed, zstream; };
__global__ __launch_bounds__(256) void noise_kernel(const float* __restrict__ zin, size_t n, float mean, float std, uint64_t seed,
                                                    const uint64_t* __restrict__ step_ptr, uint64_t stream_id,
                                                    float* __restrict__ out, bool ob = false, ZRegen zr = ZRegen{0, 0.f, 0, 0}) {        // ob: `out` is stored as bf16
  typedef float nz_f32x4 __attribute__((ext_vector_type(4)));
  const uint64_t sid = step_ptr ? *step_ptr : stream_id;
  const bool vec = (n & 3) == 0;
  const bool nt = n >= ((size_t)32 << 20);          // streaming tensors beyond the Infinity Cache: non-temporal accesses (elementwise.hip)
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q * 4 < n; q += (size_t)gridDim.x * 256) {
    float z[4];
    philox4(q, sid, seed, z);
    const size_t i = q * 4;
    if (vec) {
      float4 b = make_float4(mean, mean, mean, mean);
      if (zr.on) {
        float zz[4];
        philox4(q, zr.zstream, zr.zseed, zz);
        b = make_float4(fmaf(zr.zstd, zz[0], 0.f), fmaf(zr.zstd, zz[1], 0.f), fmaf(zr.zstd, zz[2], 0.f), fmaf(zr.zstd, zz[3], 0.f));   // bit for bit what fill_normal stored
      } else if (zin) {
        if (nt) { const nz_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nz_f32x4*>(zin + i)); b = make_float4(v[0], v[1], v[2], v[3]); }
        else b = *reinterpret_cast<const float4*>(zin + i);
      }
      b.x = fmaf(std, z[0], b.x); b.y = fmaf(std, z[1], b.y); b.z = fmaf(std, z[2], b.z); b.w = fmaf(std, z[3], b.w);
      dpi_st4(out, i, b, ob, nt);
    } else {
      float zz[4] = {0.f, 0.f, 0.f, 0.f};
      if (zr.on) philox4(q, zr.zstream, zr.zseed, zz);
      for (int k = 0; k < 4 && i + k < n; ++k) dpi_st(out, i + k, fmaf(std, z[k], zr.on ? fmaf(zr.zstd, zz[k], 0.f) : (zin ? zin[i + k] : mean)), ob);
    }
  }
}

// ---- overlap-add -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ patch, int pd, int ph, int pw, int od, int oh, int ow,
                                                          float* __restrict__ acc, int D, int H, int W) {
  const size_t n = (size_t)pd * ph * pw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int w = i % pw, h = (i / pw) % ph, d = i / ((size_t)pw * ph);
    acc[((size_t)(od + d) * H + oh + h) * W + ow + w] += patch[i];
  }
}

// number of windows {k*s : 0 <= k*s <= N-p} covering coordinate x
__device__ __forceinline__ int hits(int x, int N, int p, int s) {
  const int kmax = (N - p) / s;
  int lo = x - p + 1; lo = lo <= 0 ? 0 : (lo + s - 1) / s;
  int hi = x / s; if (hi > kmax) hi = kmax;
  return hi >= lo ? hi - lo + 1 : 0;
}

__global__ __launch_bounds__(256) void overlap_norm_kernel(float* __restrict__ acc, int D, int H, int W, int pd, int ph, int pw, int sd,
                                                           int sh, int sw, float gain) {
  const size_t n = (size_t)D * H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int w = i % W, h = (i / W) % H, d = i / ((size_t)W * H);
    const int c = hits(d, D, pd, sd) * hits(h, H, ph, sh) * hits(w, W, pw, sw);
    acc[i] = acc[i] / ((float)c * gain);
  }
}

// ---- device-resident loop control (one thread): history row, best-output flag, ReduceLROnPlateau, EarlyStopping ---------
// state (double[8]): {iter, loss_min, plateau_best, plateau_bad, es_best, es_bad, es_has_best, reserved}
__global__ void loop_control_kernel(const double* __restrict__ metrics, double* __restrict__ state, double* __restrict__ hist,
                                    int max_iters, float* __restrict__ step_lr, int* __restrict__ active, int* __restrict__ improved,
                                    int use_plateau, double factor, double threshold, int patience, double min_lr, double lr_eps,
                                    int es_patience, double es_min_delta) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  *improved = 0;
  if (!*active) return;
  const int it = (int)state[0];
  const double loss = metrics[0];
  const double lr = (double)step_lr[1];
  if (it < max_iters) { hist[4 * it + 0] = loss; hist[4 * it + 1] = metrics[1]; hist[4 * it + 2] = metrics[2]; hist[4 * it + 3] = lr; }
  if (it == 0 || loss <= state[1]) { state[1] = loss; *improved = 1; }                     // main.py:173-180
  if (use_plateau) {                                                                        // torch ReduceLROnPlateau, mode min / rel
    if (loss < state[2] * (1.0 - threshold)) { state[2] = loss; state[3] = 0.0; }
    else state[3] += 1.0;
    if (state[3] > (double)patience) {
      const double nl = fmax(lr * factor, min_lr);
      if (lr - nl > lr_eps) step_lr[1] = (float)nl;
      state[3] = 0.0;
    }
  }
  if (es_patience != 0) {                                                                   // utils/torch.py:216-275, percentage mode
    if (state[6] == 0.0) { state[4] = loss; state[6] = 1.0; }
    else if (loss != loss) *active = 0;                                                     // NaN loss stops immediately
    else {
      if (loss < state[4] - state[4] * es_min_delta / 100.0) { state[5] = 0.0; state[4] = loss; }
      else state[5] += 1.0;
      if (state[5] >= (double)es_patience) *active = 0;
    }
  }
  state[0] = (double)(it + 1);
}

__global__ __launch_bounds__(256) void copy_if_kernel(const int* __restrict__ flag, const float* __restrict__ src, float* __restrict__ dst,
                                                      size_t n) {
  if (*flag == 0) return;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

inline unsigned nblocks(size_t n) {
  size_t b = cdivz(n, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" size_t dpi_loss_ws_doubles(size_t n) { (void)n; return (size_t)kLossBlocks * 8; }

extern "C" int dpi_masked_loss(const float* out, const float* img, const float* mask, size_t n, int kind, float grad_scale,
                               float* dout, double* ws, double* result, void* stream) {
  DPI_REQUIRE(out && img && mask && ws && result && n > 0, "masked_loss: bad argument");
  DPI_REQUIRE(kind == 0 || kind == 1, "masked_loss: kind must be 0 (L1) or 1 (MSE)");
  size_t nb = cdivz(n, 256 * 8);
  if (nb > kLossBlocks) nb = kLossBlocks;
  if (nb < 1) nb = 1;
  loss_partial_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(out, img, mask, n, kind, grad_scale, dout, ws);
  if (int e = dpi_check_launch("loss_partial")) return e;
  loss_final_kernel<<<1, 64, 0, (hipStream_t)stream>>>(ws, (int)nb, (double)n, result);
  return dpi_check_launch("loss_final");
}

extern "C" int dpi_adam_multi(const dpi_adam_tensor* tensors, const int64_t* sizes, int ntensors, const float* step_lr,
                              double beta1, double beta2, double eps, const int* active, void* stream) {
  DPI_REQUIRE(tensors && sizes && step_lr && ntensors > 0 && ntensors <= 65535, "adam_multi: bad argument");
  // grid.x is sized for the largest tensor by the caller-independent cap below; blocks past a tensor's end exit.
  adam_kernel<<<dim3(256, ntensors), 256, 0, (hipStream_t)stream>>>(tensors, sizes, step_lr, beta1, beta2, eps, active);
  return dpi_check_launch("adam_multi");
}

extern "C" int dpi_noise_add_io(const float* z, size_t n, float std, uint64_t seed, const uint64_t* step_ptr, float* out, unsigned io,
                                void* stream) {
  DPI_REQUIRE(z && out && n > 0, "noise_add: bad argument");
  DPI_REQUIRE((io & ~3u) == 0, "noise_add: unknown storage-type bits in io = %u", io);
  noise_kernel<<<nblocks(cdivz(n, 4)), 256, 0, (hipStream_t)stream>>>(z, n, 0.f, std, seed, step_ptr, 0, out, (io & DPI_STORE_FWD_BF16) != 0);
  return dpi_check_launch("noise_add");
}
extern "C" int dpi_noise_add_regen_io(size_t n, float z_std, uint64_t z_seed, uint64_t z_stream_id, float std, uint64_t seed,
                                      const uint64_t* step_ptr, float* out, unsigned io, void* stream) {
  DPI_REQUIRE(out && n > 0, "noise_add_regen: bad argument");
  DPI_REQUIRE((io & ~3u) == 0, "noise_add_regen: unknown storage-type bits in io = %u", io);
  noise_kernel<<<nblocks(cdivz(n, 4)), 256, 0, (hipStream_t)stream>>>(nullptr, n, 0.f, std, seed, step_ptr, 0, out, (io & DPI_STORE_FWD_BF16) != 0,
                                                                      ZRegen{1, z_std, z_seed, z_stream_id});
  return dpi_check_launch("noise_add_regen");
}
extern "C" int dpi_noise_add(const float* z, size_t n, float std, uint64_t seed, const uint64_t* step_ptr, float* out,
                             void* stream) {
  return dpi_noise_add_io(z, n, std, seed, step_ptr, out, 0, stream);
}

extern "C" int dpi_fill_normal(float* out, size_t n, float mean, float std, uint64_t seed, uint64_t stream_id, void* stream) {
  DPI_REQUIRE(out && n > 0, "fill_normal: bad argument");
  noise_kernel<<<nblocks(cdivz(n, 4)), 256, 0, (hipStream_t)stream>>>(nullptr, n, mean, std, seed, nullptr, stream_id, out);
  return dpi_check_launch("fill_normal");
}

extern "C" int dpi_overlap_add(const float* patch, int pd, int ph, int pw, int od, int oh, int ow, float* acc, int D, int H, int W,
                               void* stream) {
  DPI_REQUIRE(patch && acc && od >= 0 && oh >= 0 && ow >= 0 && od + pd <= D && oh + ph <= H && ow + pw <= W,
              "overlap_add: patch (%d,%d,%d)@(%d,%d,%d) outside volume (%d,%d,%d)", pd, ph, pw, od, oh, ow, D, H, W);
  overlap_add_kernel<<<nblocks((size_t)pd * ph * pw), 256, 0, (hipStream_t)stream>>>(patch, pd, ph, pw, od, oh, ow, acc, D, H, W);
  return dpi_check_launch("overlap_add");
}

extern "C" int dpi_overlap_normalize(float* acc, int D, int H, int W, int pd, int ph, int pw, int sd, int sh, int sw, float gain,
                                     void* stream) {
  DPI_REQUIRE(acc && pd <= D && ph <= H && pw <= W && sd > 0 && sh > 0 && sw > 0 && gain != 0.f, "overlap_normalize: bad argument");
  overlap_norm_kernel<<<nblocks((size_t)D * H * W), 256, 0, (hipStream_t)stream>>>(acc, D, H, W, pd, ph, pw, sd, sh, sw, gain);
  return dpi_check_launch("overlap_normalize");
}

extern "C" int dpi_loop_control(const double* metrics, double* state, double* hist, int max_iters, float* step_lr, int* active,
                                int* improved, int use_plateau, double factor, double threshold, int patience, double min_lr,
                                double lr_eps, int es_patience, double es_min_delta, void* stream) {
  DPI_REQUIRE(metrics && state && hist && step_lr && active && improved && max_iters > 0, "loop_control: bad argument");
  loop_control_kernel<<<1, 64, 0, (hipStream_t)stream>>>(metrics, state, hist, max_iters, step_lr, active, improved, use_plateau, factor,
                                                        threshold, patience, min_lr, lr_eps, es_patience, es_min_delta);
  return dpi_check_launch("loop_control");
}

extern "C" int dpi_copy_if(const int* flag, const float* src, float* dst, size_t n, void* stream) {
  DPI_REQUIRE(flag && src && dst && n > 0, "copy_if: bad argument");
  copy_if_kernel<<<nblocks(n), 256, 0, (hipStream_t)stream>>>(flag, src, dst, n);
  return dpi_check_launch("copy_if");
}
