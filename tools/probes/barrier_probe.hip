// Microbenchmark: MFMA issue rate and s_barrier ping-pong cost on gfx950 (512-thread blocks, 1 per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE>  // 0: MFMA only; 1: ping-pong halves with 2 barriers per phase; 2: all waves same phase, 2 barriers; 3: barriers only
__global__ __launch_bounds__(512, 1) void probe(int phases, float* out) {
  const int wave = threadIdx.x >> 6, wm = wave >> 2;
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  auto mma = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  if (MODE == 0) {
    for (int q = 0; q < phases; ++q) mma();
  } else if (MODE == 1) {
    if (wm == 1) __builtin_amdgcn_s_barrier();
    for (int q = 0; q < phases; ++q) {
      __builtin_amdgcn_s_barrier();
      mma();
      __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
  } else if (MODE == 2) {
    for (int q = 0; q < phases; ++q) {
      __builtin_amdgcn_s_barrier();
      mma();
      __builtin_amdgcn_s_barrier();
    }
  } else {
    for (int q = 0; q < phases; ++q) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 12345.f) out[0] = s;
}

template <int MODE> static void run(const char* name, float* out) {
  const int phases = 4000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, 10, out);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, phases, out);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double cyc = ms * 1e-3 * 2.4e9 / phases;
  const double tf = 256.0 * 8 * phases * 16 * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%-44s %8.1f cycles/phase(@2.4GHz)  %7.1f TFLOP/s\n", name, cyc, tf);
}

int main() {
  float* out; (void)hipMalloc(&out, 64);
  run<0>("16 MFMA 32x32x16 per phase, 8 waves, no sync", out);
  run<1>("ping-pong halves, 2 barriers per phase", out);
  run<2>("all waves in phase, 2 barriers per phase", out);
  run<3>("2 barriers per phase only", out);
  return 0;
}
