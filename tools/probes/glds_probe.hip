// Microbenchmark: L2 -> CU load paths on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 glds_probe.hip -o glds_probe
//   mode 0: global_load_lds dwordx4 (LDS-DMA), mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128, mode 2: global_load_dwordx4 -> VGPR only
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int MODE, int ROWB>  // ROWB: bytes per row piece (128: 8 lanes/row, 64: 4 lanes/row)
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, size_t span, int iters, int row_pitch, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int LPR = ROWB / 16;  // lanes per row
  const int r = tid / LPR, c = tid % LPR;
  uint4 accv = make_uint4(0, 0, 0, 0);
  size_t base = ((size_t)blockIdx.x * 7919u * 4096u) % span;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      size_t off = (base + (size_t)(k * (256 / LPR) + r) * row_pitch + c * 16) % span;
      const char* p = src + off;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)p, (lds_ptr_t)(smem + k * 4096 + wave * 1024), 16, 0, 0);
      } else {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        if (MODE == 1) *reinterpret_cast<uint4*>(smem + k * 4096 + tid * 16) = v;
        else { accv.x ^= v.x; accv.y ^= v.y; accv.z ^= v.z; accv.w ^= v.w; }
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    base = (base + 8 * (256 / LPR) * (size_t)row_pitch) % span;
  }
  if (MODE != 0) {
    uint4 v = *reinterpret_cast<uint4*>(smem + tid * 16);
    accv.x ^= v.x;
  }
  if (accv.x == 0x12345678u) sink[0] = accv.y ^ accv.z ^ accv.w;
}

template <int MODE, int ROWB>
static void run(const char* name, const char* d, size_t span, int blocks_per_cu, int row_pitch, unsigned* sink) {
  const int iters = 2000;
  const int grid = 256 * blocks_per_cu;
  const size_t lds = 160 * 1024 / blocks_per_cu > 65536 ? 65536 : 160 * 1024 / blocks_per_cu;  // force the occupancy
  hipFuncSetAttribute((const void*)probe<MODE, ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const size_t lds_use = blocks_per_cu == 1 ? 65536 : (160 * 1024 / blocks_per_cu) & ~1023;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<MODE, ROWB>), dim3(grid), dim3(256), lds_use > 65536 ? 65536 : lds_use, 0, d, span, 10, row_pitch, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<MODE, ROWB>), dim3(grid), dim3(256), lds_use > 65536 ? 65536 : lds_use, 0, d, span, iters, row_pitch, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)grid * iters * 8 * 4096;
  printf("%-28s rowB=%3d blocks/CU=%d pitch=%5d : %7.2f TB/s  (%5.1f B/clk/CU @2.4GHz)\n", name, ROWB, blocks_per_cu, row_pitch,
         bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
  (void)lds;
}

int main() {
  const size_t span = 2u << 20;  // 2 MiB: L2 resident
  char* d; hipMalloc(&d, span + (1 << 20)); hipMemset(d, 1, span + (1 << 20));
  unsigned* sink; hipMalloc(&sink, 64);
  for (int bpc : {1, 2, 4}) {
    run<0, 128>("glds dwordx4", d, span, bpc, 512, sink);
    run<0, 64>("glds dwordx4", d, span, bpc, 512, sink);
    run<0, 128>("glds dwordx4 (contig rows)", d, span, bpc, 128, sink);
    run<1, 128>("gload->vgpr->ds_write_b128", d, span, bpc, 512, sink);
    run<2, 128>("gload->vgpr", d, span, bpc, 512, sink);
  }
  return 0;
}
