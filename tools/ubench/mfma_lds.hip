// Micro-benchmark: does an LDS read stream (values consumed as MFMA B operands one step later) slow the MFMA issue rate?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: no LDS reads, 1: 6 ds_read_b32 per 18 MFMAs (prefetched one step ahead), 2: same + 2 global loads
__global__ __launch_bounds__(256, 2) void k(float* out, const float* in, long long* cyc, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63, lk = lane >> 4, lj = lane & 15;
  const int base = lk * 2064 + lj;     // channel stride = 16 mod 32 banks
  float w[9];
  for (int i = 0; i < 9; ++i) w[i] = 0.01f * (i + lane);
  float bc[6], bn[6], g0 = 0.f, g1 = 0.f;
  for (int i = 0; i < 6; ++i) bc[i] = lds[base + i];
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int step = 0; step < 10; ++step) {
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 6; ++i) bn[i] = lds[base + ((it + step) & 31) * 36 + (i / 3) * 16 + (i % 3)];
      }
      if (MODE == 2 && (step & 3) == 0) { g0 = in[(size_t)(blockIdx.x * 256 + threadIdx.x) + (size_t)((it * 10 + step) & 1023) * 65536]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            acc[((kh + step) & 7) * 2 + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[kh * 3 + kw], bc[h * 3 + kw], acc[((kh + step) & 7) * 2 + h], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 6; ++i) bc[i] = bn[i];
      }
      if (MODE == 2) w[step % 9] += g0 * 1e-9f;
    }
  }
  long long t1 = clock64();
  float s = g1;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(int blocks, int iters, float* out, float* in, long long* cyc) {
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, in, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 180;
    printf("mode %d blocks %d: %.3f ms  %.1f TFLOP/s  cycles per mfma per wave %.2f\n", MODE, blocks, ms, n * 2048.0 * blocks * 4 / ms / 1e9, (double)c / n);
  }
}
int main() {
  float *out, *in; long long* cyc;
  (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&in, (size_t)1024 * 65536 * 4 + 1024 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
  (void)hipMemset(in, 0, (size_t)1024 * 65536 * 4 + 1024 * 256 * 4);
  run<0>(256, 2000, out, in, cyc); run<0>(512, 2000, out, in, cyc);
  run<1>(256, 2000, out, in, cyc); run<1>(512, 2000, out, in, cyc);
  run<2>(256, 2000, out, in, cyc); run<2>(512, 2000, out, in, cyc);
  return 0;
}
