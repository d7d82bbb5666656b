// Microbenchmark: the SUSTAINED dense bf16 MFMA rate of an MI355X under its package power cap, with operands held in
// registers (no LDS, no memory traffic at all), for three operand classes: zeros, small integers, random bf16 values.
// 512-thread blocks (2 waves per SIMD), one per CU, 8 independent accumulator tiles per wave; runs ~2 s per class so that the
// clock settles.  Build: hipcc --offload-arch=gfx950 -O3 mfma_power_probe.hip -o mfma_power_probe; sample rocm-smi beside it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

__global__ __launch_bounds__(512, 1) void probe(int iters, int mode, float* out) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[4], b[2];
  unsigned s = 12345u + threadIdx.x * 7919u + blockIdx.x * 104729u;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) {
    const float r = ((lcg(s) >> 8) & 0xffff) / 65536.f * 2.f - 1.f;  // uniform (-1, 1): every mantissa / exponent bit toggles
    a[i][e] = (__bf16)(mode == 0 ? 0.f : mode == 1 ? (float)(threadIdx.x & 3) : r);
  }
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) {
    const float r = ((lcg(s) >> 8) & 0xffff) / 65536.f * 2.f - 1.f;
    b[i][e] = (__bf16)(mode == 0 ? 0.f : mode == 1 ? 1.f : r * 0.05f);
  }
  for (int q = 0; q < iters; ++q) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
  if (t == 12345.678f) out[0] = t;  // keep the chains alive
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"zero operands", "small integers", "random bf16"};
  for (int mode = 0; mode < 3; ++mode) {
    int iters = 20000;
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, iters, mode, out);  // warm-up
    hipDeviceSynchronize();
    // calibrate to ~2 s
    hipEventRecord(e0); hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, iters, mode, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * 2000.f / ms);
    hipEventRecord(e0); hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, iters, mode, out); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 8 /*waves*/ * iters * 16 /*mfma*/ * 32768.0;
    printf("%-16s %8.1f ms  %8.1f TFLOP/s  (= %.2f GHz at 1024 FLOP/clk/SIMD over %d CUs)\n", names[mode], ms, flop / ms / 1e9,
           flop / ms / 1e6 / (blocks * 4 * 1024.0), blocks);
    fflush(stdout);
  }
  return 0;
}
