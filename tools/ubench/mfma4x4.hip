// Micro-benchmark / layout probe: v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 outer products, K = 1) on gfx950 — issue rate with
// independent and dependent accumulators, operand / result layout, and the DPP wave shifts used to derive the kw taps of a
// 64-voxel row from one LDS read (csrc/conv_c4_mfma.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void loop4(float* out, long long* cyc, long long* wall, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}
// MFMAs with VALU work (2 DPP moves per 3 MFMAs) and one LDS read per 3 MFMAs in the same wave: the instruction mix of the kernel
template <int NACC>
__global__ __launch_bounds__(256) void loop4_mix(float* out, long long* cyc, long long* wall, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-4f;
  __syncthreads();
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f;
  const int lane = threadIdx.x & 63;
  long long t0 = clock64(), w0 = wall_clock64();
  int off = lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        const float xc = lds[(off + (r * NACC + i) * 66) & 4095];
        const float xl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, xc), 0x138, 0xf, 0xf, false));
        const float xr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, xc), 0x130, 0xf, 0xf, false));
        acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, xl, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, xc, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, xr, acc[i], 0, 0, 0);
      }
    }
    off += 7;
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}
// the conv_q4 step: [2 DPP (or none: MODE 1; or 2 plain v_mov: MODE 2), 9 MFMAs on 3 rotating accumulators], A operands in 9 registers
template <int MODE>
__global__ __launch_bounds__(256) void loop4_step(float* out, long long* cyc, long long* wall, int iters) {
  f32x4 acc[10];
  for (int i = 0; i < 10; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float w[9];
  for (int i = 0; i < 9; ++i) w[i] = threadIdx.x * 1e-3f + i;
  float xc = threadIdx.x * 2e-3f, xh = 1.f;
  long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float xl, xr;
      if (MODE == 0) {
        xl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, xh), __builtin_bit_cast(int, xc), 0x138, 0xf, 0xf, false));
        xr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, xh), __builtin_bit_cast(int, xc), 0x130, 0xf, 0xf, false));
      } else if (MODE == 2) {
        xl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, xh), __builtin_bit_cast(int, xc), 0x111, 0xf, 0xf, false));   // row_shr:1
        xr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, xh), __builtin_bit_cast(int, xc), 0x101, 0xf, 0xf, false));   // row_shl:1
      } else { xl = xc; xr = xh; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float xv = kw == 0 ? xl : kw == 1 ? xc : xr;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) acc[r + kh] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[kh * 3 + kw], xv, acc[r + kh], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      xc += 1.f; xh = xc * 0.5f;
    }
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}
__global__ void layout(float* d, int* sh) {
  const int l = threadIdx.x;
  // A: lane (block b = l>>2, row i = l&3) ; B: lane (block b, column j = l&3)
  const float a = 1.f + (l & 3);                 // A_b[i] = 1 + i
  const float b = 100.f * (l >> 2) + 10.f * (l & 3) + 1.f;   // B_b[j]
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
  sh[l] = __builtin_amdgcn_update_dpp(-1, l, 0x138, 0xf, 0xf, false);         // wave_shr:1, lanes without a source keep `old`
  sh[64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x130, 0xf, 0xf, false);    // wave_shl:1
}
template <class K>
void run(K kern, int nacc, int per_it, int blocks, int iters, const char* tag) {
  float* out; long long *cyc, *wall;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8); hipMalloc(&wall, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, cyc, wall, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c, w; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
    const double n_per_wave = (double)iters * per_it * nacc;
    const double flops = n_per_wave * 512.0 * blocks * 4;
    printf("%s blocks=%d nacc=%d: %.3f ms  %.1f TFLOP/s  cycles/mfma(wave)=%.2f  shader MHz=%.0f\n", tag, blocks, nacc, ms,
           flops / ms / 1e9, (double)c / n_per_wave, (double)c / ((double)w / 100.0));
  }
}
int main() {
  run(loop4<8>, 8, 8, 256, 40000, "4x4x1 1 wave/SIMD  ");
  run(loop4<8>, 8, 8, 512, 40000, "4x4x1 2 waves/SIMD ");
  run(loop4<2>, 2, 8, 512, 160000, "4x4x1 2w nacc=2    ");
  run(loop4<1>, 1, 8, 256, 320000, "4x4x1 1w dependent ");
  run(loop4_step<0>, 1, 72, 256, 20000, "step wave_shr dpp 1w");
  run(loop4_step<0>, 1, 72, 512, 20000, "step wave_shr dpp 2w");
  run(loop4_step<0>, 1, 72, 768, 20000, "step wave_shr dpp 3w");
  run(loop4_step<2>, 1, 72, 256, 20000, "step row_shr dpp 1w");
  run(loop4_step<2>, 1, 72, 768, 20000, "step row_shr dpp 3w");
  run(loop4_step<1>, 1, 72, 256, 20000, "step no dpp 1w");
  run(loop4_step<1>, 1, 72, 768, 20000, "step no dpp 3w");
  float* d; int* sh; hipMalloc(&d, 256 * 4); hipMalloc(&sh, 128 * 4);
  layout<<<1, 64>>>(d, sh);
  float hd[256]; int hs[128];
  hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost); hipMemcpy(hs, sh, sizeof(hs), hipMemcpyDeviceToHost);
  printf("layout: lane l reg r = D value (expect A_b[i]*B_b[j]: which of (i,j) is reg / lane?)\n");
  for (int l : {0, 1, 2, 3, 4, 5, 63}) printf("  lane %2d: %8.0f %8.0f %8.0f %8.0f\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
  printf("wave_shr:1 lanes 0,1,15,16,17,31,32,63: %d %d %d %d %d %d %d %d\n", hs[0], hs[1], hs[15], hs[16], hs[17], hs[31], hs[32], hs[63]);
  printf("wave_shl:1 lanes 0,1,15,16,17,31,32,62,63: %d %d %d %d %d %d %d %d %d\n", hs[64], hs[65], hs[79], hs[80], hs[81], hs[95], hs[96], hs[126], hs[127]);
  return 0;
}
