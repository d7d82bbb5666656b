// GroupNorm + ReLU of the cls tower's last layer fused into the class-conditional 1x1 conv (N <= 32 classes), bf16.
//
//   logits[row][n] = sum_c bf16(relu(a[seg][c] * x[row][c] + b[seg][c])) * W[n][c] + bias[n]
//
// Reference ops: the last `GroupNorm(32, 256)` + `ReLU` of `MetaFCOSHead.cls_tower` followed by `CondConvBasic`
// (sylph/modeling/meta_fcos/fcos.py:582-667, head_utils.py:60-81).  Unfused, the normalised tensor is written by
// gn_apply_partials_kernel and read back by the 128x32 conv_igemm launch (2 x 734 MB at B = 64); nobody else reads it.  Here
// every wave streams 32-row groups straight from HBM into MFMA A-fragment registers (lane = (row, k half): 16 bytes per
// k-step, a row's 512 bytes over 16 k-steps), applies the same fused multiply-add + ReLU + bf16 rounding as the apply kernel
// (bit-identical operand values), and multiplies by the 32 x 256 code matrix held in registers.  No LDS for data, no barriers:
// the pass is HBM-bound (one read of the tower output, one write of the fp32 logits).
#include "common.h"

namespace sylph {

__global__ __launch_bounds__(256) void gn_logits_kernel(const bf16_t* __restrict__ x, int ld, const float2* __restrict__ coef,
                                                        const bf16_t* __restrict__ w, const float* __restrict__ bias, int N,
                                                        float* __restrict__ out, int out_ld, const SegDesc* __restrict__ segs,
                                                        const int2* __restrict__ tiles, int n_tiles) {
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  typedef short s16x2v __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  __shared__ __attribute__((aligned(16))) float cf[4][512];  // per wave: (a0, a1, b0, b1) per channel pair of its current segment
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;

  bf16x8 Wf[16];  // B operand: lane (n = l31, k half lh)
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) Wf[ks] = *reinterpret_cast<const bf16x8*>(w + l31 * 256 + ks * 16 + lh * 8);
  float bs[16];   // D^T: register 4q + e of a lane is class 8q + 4lh + e of row l31
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = 8 * q + 4 * lh + e;
      bs[4 * q + e] = (bias && n < N) ? bias[n] : 0.f;
    }

  int cur_seg = -1;
  const int n_groups = n_tiles * 4, stride = gridDim.x * 4;
  for (int g = blockIdx.x * 4 + wave; g < n_groups; g += stride) {
    const int2 tl = tiles[g >> 2];
    const int seg = tl.x, r0 = tl.y + (g & 3) * 32;
    const SegDesc& sd = segs[seg];
    const int nrows = sd.out_H * sd.out_W;
    if (r0 >= nrows) continue;  // wave-uniform
    if (seg != cur_seg) {       // wave-private coefficient table (same-wave LDS traffic is ordered: no barrier)
      cur_seg = seg;
      const float2* cp = coef + (size_t)seg * 256;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pr = lane + 64 * i;  // channel pair
        const float2 c0 = cp[2 * pr], c1 = cp[2 * pr + 1];
        *reinterpret_cast<float4*>(&cf[wave][4 * pr]) = make_float4(c0.x, c1.x, c0.y, c1.y);
      }
    }
    const int row = r0 + l31;
    const bool valid = row < nrows;
    const size_t grow = (size_t)(sd.out_row0 + (valid ? row : nrows - 1));
    const bf16_t* xp = x + grow * ld + lh * 8;
    u32x4 xv[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) xv[ks] = *reinterpret_cast<const u32x4*>(xp + ks * 16);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float* cq = &cf[wave][(ks * 16 + lh * 8) * 2];  // 4 channel pairs x (a0, a1, b0, b1)
      u32x4 yv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 c4 = *reinterpret_cast<const float4*>(cq + 4 * e);
        const f32x2v xf = {__uint_as_float(xv[ks][e] << 16), __uint_as_float(xv[ks][e] & 0xffff0000u)};
        const f32x2v av = {c4.x, c4.y}, bv = {c4.z, c4.w};
        const f32x2v r = __builtin_elementwise_fma(xf, av, bv);
        bf16x2 pk;
        pk[0] = (bf16_t)r[0];
        pk[1] = (bf16_t)r[1];
        const s16x2v z = {0, 0};
        yv[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2v, pk), z));  // ReLU on the bf16 pair
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[ks], __builtin_bit_cast(bf16x8, yv), acc, 0, 0, 0);
    }
    if (valid) {
      float* op = out + grow * out_ld + 4 * lh;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(op + 8 * q) =
            make_float4(acc[4 * q] + bs[4 * q], acc[4 * q + 1] + bs[4 * q + 1], acc[4 * q + 2] + bs[4 * q + 2], acc[4 * q + 3] + bs[4 * q + 3]);
    }
  }
}

// x: raw (un-normalised) tower output [rows][ld] bf16; coef: [segments][256] (a, b); w: [32][256] bf16 (rows >= N zero);
// out: fp32 [rows][out_ld >= 32]; segs / tiles: the 128-row pointwise tile table of the head
int launch_gn_logits(const void* x, int ld, const float2* coef, const void* w, const float* bias, int N, float* out, int out_ld,
                     const SegDesc* segs, const int2* tiles, int n_tiles, hipStream_t s) {
  if (N > 32 || out_ld < 32 || n_tiles <= 0) return -1;
  const int want = n_tiles;  // one block = 4 row groups = one 128-row tile per sweep
  const int grid = want < 2048 ? want : 2048;
  hipLaunchKernelGGL(gn_logits_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ld, coef, (const bf16_t*)w, bias, N, out, out_ld,
                     segs, tiles, n_tiles);
  return (int)hipGetLastError();
}

}  // namespace sylph
