// Micro-benchmark: LDS read instructions at low occupancy (1-3 waves per SIMD) and with addresses that are only 4-byte aligned —
// what a stencil kernel needs to fetch several consecutive taps / voxels with one instruction: ds_read_b32 x4, ds_read2_b32 x2,
// ds_read_b64 x2, ds_read_b128 (aligned and misaligned by 4 / 8 / 12 bytes).  Prints correctness and clocks per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void lds_loop(float* out, long long* cyc, int iters, int mis) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // every lane reads 4 consecutive floats starting at float index 4 * lane * 5 % 2048 * ... + mis (mis = 0..3 floats of misalignment)
  unsigned addr = (unsigned)((lane * 36 + mis) * 4 + (threadIdx.x >> 6) * 9216);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float a, b, c, d;
      const unsigned ad = addr + r * 64;
      if (MODE == 0) {
        asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %4 offset:8\n ds_read_b32 %3, %4 offset:12\n s_waitcnt lgkmcnt(0)"
                     : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(ad));
      } else if (MODE == 1) {
        f32x2 p, q;
        asm volatile("ds_read2_b32 %0, %2 offset1:1\n ds_read2_b32 %1, %2 offset0:2 offset1:3\n s_waitcnt lgkmcnt(0)" : "=v"(p), "=v"(q) : "v"(ad));
        a = p[0]; b = p[1]; c = q[0]; d = q[1];
      } else if (MODE == 2) {
        f32x2 p, q;
        asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n s_waitcnt lgkmcnt(0)" : "=v"(p), "=v"(q) : "v"(ad));
        a = p[0]; b = p[1]; c = q[0]; d = q[1];
      } else {
        f32x4 p;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(p) : "v"(ad));
        a = p[0]; b = p[1]; c = p[2]; d = p[3];
      }
      s0 += a; s1 += b; s2 += c; s3 += d;
    }
  }
  long long t1 = clock64();
  out[(blockIdx.x * 256 + threadIdx.x) * 4 + 0] = s0; out[(blockIdx.x * 256 + threadIdx.x) * 4 + 1] = s1;
  out[(blockIdx.x * 256 + threadIdx.x) * 4 + 2] = s2; out[(blockIdx.x * 256 + threadIdx.x) * 4 + 3] = s3;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* tag, int blocks, int mis) {
  const int iters = 2000;
  float* out; long long* cyc;
  hipMalloc(&out, (size_t)blocks * 256 * 16); hipMalloc(&cyc, blocks * 8);
  lds_loop<MODE><<<blocks, 256>>>(out, cyc, iters, mis);
  hipDeviceSynchronize();
  lds_loop<MODE><<<blocks, 256>>>(out, cyc, iters, mis);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  float h[4]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
  // lane 0 of wave 0 reads floats mis + r*16 + {0,1,2,3}, r = 0..7, iters times
  double e0 = 0; for (int r = 0; r < 8; ++r) e0 += mis + r * 16;
  const bool ok = h[0] == (float)(e0 * iters) || fabs(h[0] - e0 * iters) < 1e-3 * e0 * iters + 1;
  printf("%-22s blocks/CU %d  misaligned by %2d B: %6.1f clk per group of 4 floats (wave)  %s (lane 0 sum %.0f, expected %.0f)\n", tag, blocks / 256, mis * 4,
         (double)c / (iters * 8.0), ok ? "values ok" : "VALUES WRONG", h[0], e0 * iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int wg : {256, 512, 768}) {
    run<0>("4 x ds_read_b32", wg, 1);
    run<1>("2 x ds_read2_b32", wg, 1);
    run<2>("2 x ds_read_b64", wg, 0);
    run<2>("2 x ds_read_b64", wg, 1);
    run<3>("ds_read_b128", wg, 0);
    run<3>("ds_read_b128", wg, 1);
    run<3>("ds_read_b128", wg, 2);
  }
  return 0;
}
