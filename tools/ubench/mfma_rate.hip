// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 (independent accumulators) and the shader clock under that load.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, long long* cyc, long long* wall, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}
template <int NACC>
void run(int blocks, int iters, const char* tag) {
  float* out; long long *cyc, *wall;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8); hipMalloc(&wall, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, 256>>>(out, cyc, wall, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c, w; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
    const double n_per_wave = (double)iters * 4 * NACC;
    const double flops = n_per_wave * 2048.0 * blocks * 4;
    printf("%s blocks=%d nacc=%d: %.3f ms  %.1f TFLOP/s  cycles/mfma(wave)=%.2f  shader MHz=%.0f (wall 100MHz ticks %lld)\n", tag, blocks, NACC, ms,
           flops / ms / 1e9, (double)c / n_per_wave, (double)c / ((double)w / 100.0), w);
  }
}
int main() {
  run<8>(256, 20000, "1 wave/SIMD ");
  run<8>(512, 20000, "2 waves/SIMD");
  run<8>(1024, 10000, "4 waves/SIMD");
  run<2>(512, 40000, "2w nacc=2   ");
  run<1>(256, 80000, "1w dependent");
  return 0;
}
