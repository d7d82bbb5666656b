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
        if (8 * q + 4 * lh < out_ld)  // out_ld: any multiple of 4 >= N (8 for <= 8 classes)
          *reinterpret_cast<float4*>(op + 8 * q) =
              make_float4(acc[4 * q] + bs[4 * q], acc[4 * q + 1] + bs[4 * q + 1], acc[4 * q + 2] + bs[4 * q + 2], acc[4 * q + 3] + bs[4 * q + 3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Last GroupNorm + ReLU of the bbox tower fused into the 3x3 prediction convs (bbox_pred 4 + ctrness 1 [+ iou 1] channels).
// A 3x3 conv with a handful of output channels is linear in its taps:
//     pred[r][n] = sum_tap ( xn[r + shift(tap)] . W[tap][n] ),     xn = relu(GN(x)),  zero outside the map,
// so pass 1 streams xn ONCE (same HBM -> register scheme as gn_logits_kernel) against the 9 * Cp <= 64 stacked tap rows and
// writes the per-position tap responses (fp32, <= 64 per position); pass 2 adds the nine shifted responses, bias, per-level
// Scale and ReLU (`F.relu(scale_l(bbox_pred(t)))`, fcos.py:640-660).  Unfused: apply (0.73 GB read + 0.73 GB write) + a halo-mode
// 128x32 conv (0.73 GB read); fused: 0.73 GB read + 0.28 GB write + 0.28 GB read.
template <int NT>
__global__ __launch_bounds__(256) void gn_taps_kernel(const bf16_t* __restrict__ x, int ld, const float2* __restrict__ coef,
                                                      const bf16_t* __restrict__ w, float* __restrict__ out, int sw, size_t plane_rows,
                                                      const SegDesc* __restrict__ segs, const int2* __restrict__ tiles, int n_tiles) {
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  typedef short s16x2v __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  __shared__ __attribute__((aligned(16))) float cf[4][512];
  constexpr int TR_PITCH = NT * 32 + 4;  // floats per row of the transpose tile (+4: the 16-byte row pieces of 8 lanes fall on different banks)
  __shared__ __attribute__((aligned(16))) float tr[4][32 * TR_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  bf16x8 Wf[NT][16];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) Wf[t][ks] = *reinterpret_cast<const bf16x8*>(w + (t * 32 + l31) * 256 + ks * 16 + lh * 8);
  int cur_seg = -1;
  const int n_groups = n_tiles * 4, stride = gridDim.x * 4;
  for (int g = blockIdx.x * 4 + wave; g < n_groups; g += stride) {
    const int2 tl = tiles[g >> 2];
    const int seg = tl.x, r0 = tl.y + (g & 3) * 32;
    const SegDesc& sd = segs[seg];
    const int nrows = sd.out_H * sd.out_W;
    if (r0 >= nrows) continue;
    if (seg != cur_seg) {
      cur_seg = seg;
      const float2* cp = coef + (size_t)seg * 256;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pr = lane + 64 * i;
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
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float* cq = &cf[wave][(ks * 16 + lh * 8) * 2];
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
        yv[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2v, pk), z));
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[t][ks], __builtin_bit_cast(bf16x8, yv), acc[t], 0, 0, 0);
    }
    // column j = kh * sw + kw * cp + n (sw = slice width, a multiple of 4): columns [kh sw, (kh + 1) sw) of a row are its record in
    // plane kh, so that pass 2 reads near-contiguous sw-float records per kernel row.  The 32 rows of the group are contiguous in
    // every plane (32 * sw floats): the accumulator tile is transposed through wave-private LDS and written as whole 16-byte
    // pieces of that run, 1 KiB per store instruction (direct from the accumulator layout a store touched 32 records for 32 bytes each).
    float* st = &tr[wave][0];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(st + l31 * TR_PITCH + t * 32 + 8 * q + 4 * lh) =
            make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
    const int vrows = min(32, nrows - r0), q4 = sw >> 2;
    const size_t grow0 = (size_t)sd.out_row0 + r0;
    for (int kh = 0; kh < 3; ++kh) {
      float* dst = out + ((size_t)kh * plane_rows + grow0) * sw;
      for (int i = lane; i < vrows * q4; i += 64) {
        const int row = i / q4, c4 = i - row * q4;
        *reinterpret_cast<float4*>(dst + (size_t)i * 4) = *reinterpret_cast<const float4*>(st + row * TR_PITCH + kh * sw + c4 * 4);
      }
    }
  }
}

// pass 2: block = one 128-row tile, one thread per output position.  For kernel row kh the tile needs the 130 CONTIGUOUS records
// [r0 + (kh - 1) W - 1, r0 + (kh - 1) W + 128] of plane kh: staged through LDS with coalesced float4 loads (pitch sw + 1 floats:
// conflict-free column reads); positions outside the map are masked at use (their records belong to other rows / segments).
__global__ __launch_bounds__(128) void tap_gather_kernel(const float* __restrict__ planes, int sw, size_t plane_rows, int cp,
                                                         const float* __restrict__ bias, int relu_nch, int mul_nch, float* __restrict__ out,
                                                         int out_ld, const SegDesc* __restrict__ segs, const int2* __restrict__ tiles) {
  __shared__ float sm[3][130 * 25];  // sw <= 24
  const int2 tl = tiles[blockIdx.x];
  const SegDesc& sd = segs[tl.x];
  const int H = sd.out_H, W = sd.out_W, HWn = H * W, r0 = tl.y, tid = threadIdx.x, pitch = sw + 1;
  const int q4 = sw >> 2;  // float4s per record
  // all of a thread's loads (<= 3 x 7 float4: sw <= 24 -> q4 <= 6, ceil(130 * 6 / 128) = 7) are issued before the first LDS store:
  // the staging is a pure HBM stream, and one load per loop trip was latency-bound (2.1 TB/s)
  constexpr int MAXIT = 7;
  float4 v[3][MAXIT];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int first = r0 + (kh - 1) * W - 1;  // map-relative row of LDS record 0
    const float* pl = planes + ((size_t)kh * plane_rows + sd.out_row0) * sw;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int i = tid + it * 128;
      const int rec = i / q4, c4 = i - rec * q4;
      const int rr = first + rec;
      v[kh][it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 130 * q4 && (unsigned)rr < (unsigned)HWn) v[kh][it] = *reinterpret_cast<const float4*>(pl + (size_t)rr * sw + c4 * 4);
    }
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int i = tid + it * 128;
      if (i < 130 * q4) {
        const int rec = i / q4, c4 = i - rec * q4;
        float* d = &sm[kh][rec * pitch + c4 * 4];
        d[0] = v[kh][it].x; d[1] = v[kh][it].y; d[2] = v[kh][it].z; d[3] = v[kh][it].w;
      }
    }
  }
  __syncthreads();
  const int r = r0 + tid;
  if (r >= HWn) return;
  const int y = r / W, xx = r - y * W;
  float acc[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) acc[n] = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {  // fixed order: kh outer, kw inner
    if ((unsigned)(y + kh - 1) >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      if ((unsigned)(xx + kw - 1) >= (unsigned)W) continue;
      const float* tp = &sm[kh][(tid + kw) * pitch + kw * cp];  // record of row r + (kh - 1) W + (kw - 1)
#pragma unroll
      for (int n = 0; n < 8; ++n)
        if (n < cp) acc[n] += tp[n];
    }
  }
  float* op = out + (size_t)(sd.out_row0 + r) * out_ld;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    if (n >= cp) break;
    float t = acc[n] + bias[n];
    if (n < mul_nch) t *= sd.mul;
    if (n < relu_nch) t = t > 0.f ? t : 0.f;
    op[n] = t;
  }
}

// w_taps: [64][256] bf16, row kh * sw + kw * cp + n = W[n][kh][kw][:] with sw = roundup4(3 * cp) (other rows zero);
// planes_ws: fp32 [3][plane_rows][sw]
int launch_gn_pred_taps(const void* x, int ld, const float2* coef, const void* w_taps, int cp, const float* bias, int relu_nch, int mul_nch,
                        float* planes_ws, size_t plane_rows, float* out, int out_ld, const SegDesc* segs, const int2* tiles, int n_tiles,
                        hipStream_t s) {
  const int sw = (3 * cp + 3) & ~3;
  if (cp < 1 || 3 * sw > 64 || n_tiles <= 0) return -1;
  const int grid = n_tiles < 2048 ? n_tiles : 2048;
  hipLaunchKernelGGL(gn_taps_kernel<2>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ld, coef, (const bf16_t*)w_taps, planes_ws, sw, plane_rows,
                     segs, tiles, n_tiles);
  hipLaunchKernelGGL(tap_gather_kernel, dim3(n_tiles), dim3(128), 0, s, planes_ws, sw, plane_rows, cp, bias, relu_nch, mul_nch, out, out_ld, segs,
                     tiles);
  return (int)hipGetLastError();
}

// x: raw (un-normalised) tower output [rows][ld] bf16; coef: [segments][256] (a, b); w: [32][256] bf16 (rows >= N zero);
// out: fp32 [rows][out_ld], out_ld a multiple of 4 >= N; segs / tiles: the 128-row pointwise tile table of the head
int launch_gn_logits(const void* x, int ld, const float2* coef, const void* w, const float* bias, int N, float* out, int out_ld,
                     const SegDesc* segs, const int2* tiles, int n_tiles, hipStream_t s) {
  if (N > 32 || out_ld < N || (out_ld & 3) != 0 || n_tiles <= 0) return -1;
  const int want = n_tiles;  // one block = 4 row groups = one 128-row tile per sweep
  const int grid = want < 2048 ? want : 2048;
  hipLaunchKernelGGL(gn_logits_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ld, coef, (const bf16_t*)w, bias, N, out, out_ld,
                     segs, tiles, n_tiles);
  return (int)hipGetLastError();
}

}  // namespace sylph
