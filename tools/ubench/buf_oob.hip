// Check: raw buffer loads return 0 for offsets past num_records (used for branch-free zero padding).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int n) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 4, 0x00020000);
  int off = threadIdx.x * 4;
  if (threadIdx.x & 1) off = 0x7ffffff0;
  y[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
  int off4 = threadIdx.x * 16;
  if (threadIdx.x & 2) off4 = -16;                       // negative (as unsigned: huge) offset
  f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off4, 0, 0));
  y[64 + threadIdx.x] = v[0] + v[1] + v[2] + v[3];
  // partially out of range dwordx4 at the end of the buffer
  f4 t = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (n - 2) * 4, 0, 0));
  if (threadIdx.x == 0) { y[128] = t[0]; y[129] = t[1]; y[130] = t[2]; y[131] = t[3]; }
}
int main() {
  const int n = 1024;
  float *x, *y, hx[1024], hy[132];
  for (int i = 0; i < n; ++i) hx[i] = i + 1;
  (void)hipMalloc(&x, n * 4 + 4096); (void)hipMalloc(&y, 132 * 4);
  (void)hipMemset(x, 0x7f, n * 4 + 4096);
  (void)hipMemcpy(x, hx, n * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(x, y, n);
  (void)hipMemcpy(hy, y, 132 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) { float e = (i & 1) ? 0.f : i + 1; if (hy[i] != e) { ++bad; printf("b32 lane %d got %g want %g\n", i, hy[i], e); } }
  for (int i = 0; i < 64; ++i) { float e = (i & 2) ? 0.f : 16.f * i + 10; if (hy[64 + i] != e) { ++bad; printf("b128 lane %d got %g want %g\n", i, hy[64 + i], e); } }
  printf("tail dwordx4: %g %g %g %g (want 1023 1024 0 0)\n", hy[128], hy[129], hy[130], hy[131]);
  printf("bad=%d\n", bad);
  return bad;
}
