// Micro-benchmark: a streaming elementwise pass (y = a x + b over 25 x 256x128x128 floats, 419 MB in, 419 MB out — larger than the
// 256 MB Infinity Cache) with plain / non-temporal loads and stores and different launch sizes: what HBM rate can the elementwise
// kernels of the iteration expect, and does `nt` (streaming, no reuse) change it?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void axpb(const float4* __restrict__ x, float4* __restrict__ y, size_t n4, float a, float b) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 f;
    if (MODE & 1) { f.x = __builtin_nontemporal_load(&x[i].x); f.y = __builtin_nontemporal_load(&x[i].y); f.z = __builtin_nontemporal_load(&x[i].z); f.w = __builtin_nontemporal_load(&x[i].w); }
    else f = x[i];
    f.x = a * f.x + b; f.y = a * f.y + b; f.z = a * f.z + b; f.w = a * f.w + b;
    if (MODE & 2) { __builtin_nontemporal_store(f.x, &y[i].x); __builtin_nontemporal_store(f.y, &y[i].y); __builtin_nontemporal_store(f.z, &y[i].z); __builtin_nontemporal_store(f.w, &y[i].w); }
    else y[i] = f;
  }
}
template <int MODE>
void run(const char* tag, const float4* x, float4* y, size_t n4, int blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  axpb<MODE><<<blocks, 256>>>(x, y, n4, 1.5f, 0.25f);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) axpb<MODE><<<blocks, 256>>>(x, y, n4, 1.5f, 0.25f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s blocks %6d: %.3f ms per pass  %.0f GB/s (read + write)\n", tag, blocks, ms / 10, 2.0 * n4 * 16 / (ms / 10) / 1e6);
}
int main() {
  const size_t n4 = (size_t)25 * 256 * 128 * 128 / 4;
  float4 *x, *y; (void)hipMalloc(&x, n4 * 16); (void)hipMalloc(&y, n4 * 16);
  (void)hipMemset(x, 0, n4 * 16);
  for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
    run<0>("plain", x, y, n4, blocks);
    run<2>("nt store", x, y, n4, blocks);
    run<3>("nt load + nt store", x, y, n4, blocks);
  }
  return 0;
}
