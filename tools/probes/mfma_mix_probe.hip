// Microbenchmark: what the INSTRUCTION MIX of conv_hpipe_kernel sustains under the MI355X package power cap -- the dense bf16 MFMA
// stream of tools/probes/mfma_power_probe.hip (random bf16 operands, 512-thread blocks, one per CU, 2 waves per SIMD) with the
// operand traffic of the tower kernel switched on step by step:
//   mode 0  operands in registers (the r3 probe's "random bf16" row: the ceiling without any data movement)
//   mode 1  + operands re-read from LDS every k-step: 12 ds_read_b128 per 16 MFMAs per wave (wave tile 128 x 64: 4 A + 2 B fragments
//             per 8 MFMAs), conflict-free addresses, data = random bf16
//   mode 2  + the L2 -> LDS stream: 3 global_load_lds_dwordx4 per wave and 16-MFMA phase (24 KiB per CU and phase = one 16-KiB weight
//             stage + 8 KiB of halo, the per-phase DMA volume of conv_hpipe), from a 4-MiB L2-resident buffer, retired with counted vmcnt
// 256 blocks, ~2 s per mode so that clock and power settle; sample `rocm-smi --showpower --showclocks` beside it (tools/power_trace_probe.sh).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_mix_probe.hip -o mfma_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(int iters, const char* __restrict__ gsrc, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [0, 64K): fragment source; [64K, 64K + 4 x 24K): DMA ring
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned s = 12345u + tid * 7919u + blockIdx.x * 104729u;
  for (int i = tid; i < 65536 / 2; i += 512) {
    const float r = ((lcg(s) >> 8) & 0xffff) / 65536.f * 2.f - 1.f;
    reinterpret_cast<__bf16*>(smem)[i] = (__bf16)(((i >> 3) & 1) ? r * 0.05f : r);
  }
  __syncthreads();
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(smem + (i * 64 + lane) * 16);
  for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const bf16x8*>(smem + 8192 + (i * 64 + lane) * 16);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // conflict-free fragment addresses: lane l reads 16 bytes at row l (64-byte pitch, swizzled chunk), as the conv kernels do
  const unsigned fbase = lds0 + (lane & 31) * 64 + ((((lane >> 5)) ^ ((lane >> 2) & 3)) << 4) + wave * 256;
  const char* gp = gsrc + tid * 16;  // + a block- and phase-dependent offset below, wrapped inside the 4-MiB buffer
  for (int q = 0; q < iters; ++q) {
    if (MODE >= 2) {  // 3 x 1 KiB per wave into ring stage q & 3; the stage issued two phases ago must have landed
      char* dst = smem + 65536 + (q & 3) * 24576 + wave * 3072;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const unsigned off = ((unsigned)blockIdx.x * 32768u + (unsigned)q * 24576u + (unsigned)j * 8192u) & ((4u << 20) - 1);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gp + off), (lds_ptr_t)(dst + j * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (MODE >= 1) {
        const unsigned ad = fbase + ((q * 2 + k) & 7) * 2048;  // walks 16 KiB of the fragment source
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(a[i]) : "v"(ad + i * 4096u));
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(b[i]) : "v"(ad + 32768u + i * 4096u));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    }
  }
  float t = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
  if (t == 12345.678f) out[0] = t;
}

template <int MODE>
static void run(int blocks, const char* gsrc, float* out, const char* name) {
  const int lds = 65536 + 4 * 24576;
  (void)hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int iters = 20000;
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), lds, 0, iters, gsrc, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), lds, 0, iters, gsrc, out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  iters = (int)(iters * 2500.f / ms);
  (void)hipEventRecord(e0); hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), lds, 0, iters, gsrc, out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 8 * iters * 16 * 32768.0;
  printf("%-58s %8.1f ms  %8.1f TFLOP/s  (= %.2f GHz-equivalent at 1024 FLOP/clk/SIMD)\n", name, ms, flop / ms / 1e9, flop / ms / 1e6 / (blocks * 4 * 1024.0));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  float* out;
  char* g;
  if (hipMalloc(&out, 4) != hipSuccess || hipMalloc(&g, (4u << 20) + 65536) != hipSuccess) return 1;
  (void)hipMemset(g, 0x3c, (4u << 20) + 65536);
  run<0>(blocks, g, out, "mode 0: random bf16 operands in registers");
  run<1>(blocks, g, out, "mode 1: + 12 ds_read_b128 per 16 MFMAs per wave");
  run<2>(blocks, g, out, "mode 2: + 3 global_load_lds per wave and phase (24 KiB / CU)");
  return 0;
}
