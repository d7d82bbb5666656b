// ResNet stem: 7x7 stride-2 pad-3 conv (3 -> 64) + FrozenBN + ReLU, bf16, as a dedicated MFMA kernel.
//
// Reference op: detectron2 BasicStem.conv1 (+ norm + relu) as used by build_resnet_fpn_backbone
// (sylph/modeling/backbone/fpn.py:21-45).  The generic implicit-GEMM kernel runs this layer LDS-DMA bound: with
// only 3 input channels every 128-byte K-slice is a fresh gather, 96 KB of loads per 128x64 tile.  Here the input
// patch of a tile (21 x 38 pixels x 4 channels = 6.4 KB) is staged once in LDS and the im2col happens in the
// fragment reads; the weights (64 x 224 bf16) live in REGISTERS for the whole persistent block:
//
//   * tile = 8 x 16 output positions x 64 channels, 4 waves, wave w = patch rows 2w, 2w+1 (32 positions);
//   * K = 7 kernel rows x (8 pixels x 4 channels) = 224 = 14 MFMA k-steps (pixel 7 and channel 3 carry zero
//     weights); k-step (kh, h): lane (position, lh) reads the 2 pixels kw = 4h + 2lh, +1 of input row 2py + kh:
//     one aligned ds_read_b128 at ((2py + kh) * 48 + 2px + 4h + 2lh) * 8 bytes.  The 48-pixel patch pitch makes two
//     consecutive patch rows 768 B = 0 mod 256 apart, so every ds_read_b128 lane group covers all 64 banks once;
//   * D^T MFMA (weights as the A operand): a lane ends up with 4 consecutive channels of one position; the
//     epilogue goes through an fp32 LDS tile like conv_igemm.hip (scale/shift, ReLU, 16-byte stores);
//   * persistent blocks (grid-stride over tiles); the patches of the next two tiles are in flight in registers.
#include "common.h"
#include "kernels.h"

namespace sylph {

namespace {
constexpr int PP = 48;                 // patch pitch in pixels
constexpr int PROWS = 21, PCOLS = 38;  // patch extent actually read
constexpr int PATCH_BYTES = 8192;      // 21 * 48 * 8 = 8064, rounded
constexpr int SCP = 68;                // fp32 pitch of the epilogue tile (64 + 4)
constexpr int NLOAD = (PROWS * PCOLS + 255) / 256;  // 8-byte pixel loads per lane per tile (4)
}  // namespace

__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           bf16_t* __restrict__ out, int H, int W, int H2, int W2,
                                                           int tiles_y, int tiles_x, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  float* sC = reinterpret_cast<float*>(smem + PATCH_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;

  // weights -> registers: k-step ks = (kh, h), fragment j = channels 32j .. 32j+31
  bf16x8 wb[14][2];
#pragma unroll
  for (int ks = 0; ks < 14; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) wb[ks][j] = *reinterpret_cast<const bf16x8*>(wp + (j * 32 + l31) * 224 + ks * 16 + lh * 8);

  const int m = wave * 32 + l31, py = m >> 4, px = m & 15;
  const int aoff = ((2 * py) * PP + 2 * px + 2 * lh) * 8;

  const int c8 = tid & 7, rr = tid >> 3;
  float sc[8], sh[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 s4 = reinterpret_cast<const float4*>(scale + c8 * 8)[h], b4 = reinterpret_cast<const float4*>(shift + c8 * 8)[h];
    sc[4 * h] = s4.x; sc[4 * h + 1] = s4.y; sc[4 * h + 2] = s4.z; sc[4 * h + 3] = s4.w;
    sh[4 * h] = b4.x; sh[4 * h + 1] = b4.y; sh[4 * h + 2] = b4.z; sh[4 * h + 3] = b4.w;
  }

  // Loads are UNCONDITIONAL (clamped address, value zeroed afterwards) and fetch() runs every iteration (clamped
  // tile): with a static number of loads per tile the compiler can retire a patch with a counted vmcnt instead of
  // vmcnt(0), which would also wait for the previous tile's output stores and for the other patch in flight.
  // tile-independent per-lane constants of the patch loader and of the epilogue (only the tile origin changes)
  int f_pr[NLOAD], f_pc[NLOAD], f_lds[NLOAD];
  uint32_t f_in = 0;
#pragma unroll
  for (int r = 0; r < NLOAD; ++r) {
    const int idx = tid + 256 * r;
    f_pr[r] = idx / PCOLS; f_pc[r] = idx - f_pr[r] * PCOLS;
    f_lds[r] = (f_pr[r] * PP + f_pc[r]) * 8;
    f_in |= (idx < PROWS * PCOLS ? 1u : 0u) << r;
  }

  auto fetch = [&](int tile_in, uint2 (&v)[NLOAD], uint32_t& okmask) {
    const int tile = tile_in < ntiles ? tile_in : ntiles - 1;
    const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, b = t2 / tiles_y;
    const int iy0 = ty * 16 - 3, ix0 = tx * 32 - 3;
    okmask = 0;
    const bf16_t* xb = x + (size_t)b * H * W * 4;
#pragma unroll
    for (int r = 0; r < NLOAD; ++r) {
      const int iy = iy0 + f_pr[r], ix = ix0 + f_pc[r];
      const bool ok = ((f_in >> r) & 1u) && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
      // issued through inline asm so that the compiler does not schedule a vmcnt(0) of its own before park():
      // the wait is the counted one in do_tile() (loads retire in order, so <= NLOAD outstanding ops means this
      // patch has landed while the other patch may still be in flight)
      const bf16_t* src = xb + (cy * W + cx) * 4;
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v[r]) : "v"(src) : "memory");
      okmask |= (ok ? 1u : 0u) << r;
    }
  };
  auto park = [&](const uint2 (&v)[NLOAD], uint32_t okmask) {
#pragma unroll
    for (int r = 0; r < NLOAD; ++r) {
      const uint2 t = ((okmask >> r) & 1u) ? v[r] : make_uint2(0u, 0u);
      if ((f_in >> r) & 1u) *reinterpret_cast<uint2*>(patch + f_lds[r]) = t;
    }
  };

  // the weight / scale / shift loads retire here, once: inside the loop only patch loads and output stores are in flight
#pragma unroll
  for (int ks = 0; ks < 14; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(wb[ks][j]));
#pragma unroll
  for (int e = 0; e < 8; ++e) { asm volatile("" : "+v"(sc[e])); asm volatile("" : "+v"(sh[e])); }

  // One tile: park its patch (fetched two tiles ago), immediately re-use the same registers to fetch the patch of
  // the tile two steps ahead, MFMAs, epilogue.  Two register sets alternate, so no load is ever waited for (or
  // moved) before its own park: the ~2 us HBM round trip is longer than one tile's work.
  const int g = gridDim.x;
  auto do_tile = [&](int tile, uint2 (&q)[NLOAD], uint32_t& qm) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    static_assert(NLOAD == 4, "the vmcnt immediate above is NLOAD");
    park(q, qm);
    lds_barrier();
    fetch(tile + 2 * g, q, qm);

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(patch + aoff + (kh * PP + 4 * h) * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[kh * 2 + h][j], fa, acc[j], 0, 0, 0);
      }

    const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, b = t2 / tiles_y;
    const int oy0 = ty * 8, ox0 = tx * 16;
#pragma unroll
    for (int p = 0; p < 2; ++p) {  // 64 positions per pass: waves 2p, 2p+1
      if (p > 0) lds_barrier();
      if ((wave >> 1) == p) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4)
            *reinterpret_cast<float4*>(sC + ((wave & 1) * 32 + l31) * SCP + j * 32 + 8 * q4 + 4 * lh) =
                make_float4(acc[j][4 * q4], acc[j][4 * q4 + 1], acc[j][4 * q4 + 2], acc[j][4 * q4 + 3]);
      }
      lds_barrier();
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int rl = rr + 32 * it, mm = p * 64 + rl;
        const int oy = oy0 + (mm >> 4), ox = ox0 + (mm & 15);
        if (oy < H2 && ox < W2) {
          const float4 lo = *reinterpret_cast<const float4*>(sC + rl * SCP + c8 * 8);
          const float4 hi = *reinterpret_cast<const float4*>(sC + rl * SCP + c8 * 8 + 4);
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) { v[e] = v[e] * sc[e] + sh[e]; v[e] = v[e] > 0.f ? v[e] : 0.f; }
          store8<bf16_t>(out + (((size_t)b * H2 + oy) * W2 + ox) * 64 + c8 * 8, v);
        }
      }
    }
    lds_barrier();  // patch and sC are rewritten by the next tile
  };

  uint2 qa[NLOAD], qb[NLOAD];
  uint32_t ma, mb;
  int tile = blockIdx.x;
  fetch(tile, qa, ma);
  fetch(tile + g, qb, mb);
  while (tile < ntiles) {
    do_tile(tile, qa, ma);
    tile += g;
    if (tile >= ntiles) break;
    do_tile(tile, qb, mb);
    tile += g;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem + 3x3 stride-2 max-pool in ONE kernel (BasicStem.forward: conv1 -> norm -> relu -> max_pool2d(3, 2, 1)).
// Unfused, the 64-channel stem output (2.2 GB at B = 64, 800 x 1344) is written by stem_conv_kernel and read back by
// maxpool_kernel: both launches run at the HBM rate (0.69 + 0.60 ms).  Here a tile is 17 x 15 stem positions (two MFMA passes
// of 128; 14 % recompute at the overlapping row / column) = 8 x 7 pooled positions: the stem tile goes to LDS as bf16
// (the same rounding the unfused path applies before pooling, so the result is bit-identical) and only the pooled
// tensor (0.55 GB) is written.  Stem positions outside the image are stored as 0: every value is a ReLU output (>= 0) and each
// window holds at least one real position, so 0 is as good as the -inf padding of max_pool2d.
namespace {
constexpr int FR = 17, FC = 15;                  // stem tile
constexpr int FPR = 8, FPC = 7;                  // pooled tile
constexpr int FPROWS = 2 * FR + 5, FPCOLS = 36;  // input patch: 39 rows x (2 * 15 + 5 = 35, + the zero-weight 8th column)
constexpr int FPATCH_BYTES = 15104;              // 39 * 48 * 8 = 14 976, rounded to 128
constexpr int FTP = 144;                         // stem tile row pitch: 64 ch bf16 + 16 B pad
constexpr int FTILE_BYTES = FR * FC * FTP;       // 36 720
constexpr int FNLOAD = (FPROWS * FPCOLS + 255) / 256;  // 6
constexpr int FLDS = FPATCH_BYTES + ((FTILE_BYTES + 127) / 128) * 128 + 128 * 4;
}  // namespace

// RAW = true: the network-input normalisation of `preprocess_kernel` (elementwise.hip) is applied HERE, on the way from the
// caller's fp32 (3, h, w) planes into the LDS patch -- (p - mean) * (1 / std) rounded to bf16, zero outside the image's own
// h x w, the same arithmetic, so the result is bit-identical -- and the [B][H][W][4] bf16 copy of the batch (0.55 GB written and
// read back at B = 64) is never made.  Three 4-byte loads per patch pixel instead of one 8-byte load.
struct StemRawArgs { const ImageDesc* imgs; int B; float m0, m1, m2, is0, is1, is2; };

template <bool RAW>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const bf16_t* __restrict__ x, const StemRawArgs raw, const bf16_t* __restrict__ wp,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           bf16_t* __restrict__ out, char* __restrict__ trash, int H, int W, int H2,
                                                           int W2, int H4, int W4, int tiles_y, int tiles_x, int ntiles) {
  constexpr int NV = RAW ? 3 * FNLOAD : FNLOAD;  // VMEM loads per thread and tile
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch = smem;
  char* st = smem + FPATCH_BYTES;
  float* ss = reinterpret_cast<float*>(smem + FPATCH_BYTES + ((FTILE_BYTES + 127) / 128) * 128);  // scale[64], shift[64]
  ImageDesc* descs = reinterpret_cast<ImageDesc*>(smem + FLDS);  // RAW: the batch's image table (a global read per tile would join the counted vmcnt queue)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  if (RAW)
    for (int i = tid; i < raw.B; i += 256) descs[i] = raw.imgs[i];

  bf16x8 wb[14][2];
#pragma unroll
  for (int ks = 0; ks < 14; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) wb[ks][j] = *reinterpret_cast<const bf16x8*>(wp + (j * 32 + l31) * 224 + ks * 16 + lh * 8);
  if (tid < 128) ss[tid] = tid < 64 ? scale[tid] : shift[tid - 64];

  // per pass p: this lane's stem position m = p * 128 + wave * 32 + l31 (m = 255 does not exist: it recomputes 254 and is dropped)
  int aoff[2], sy[2], sx[2], toff[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int m = min(p * 128 + wave * 32 + l31, FR * FC - 1);
    sy[p] = m / FC;
    sx[p] = m - sy[p] * FC;
    aoff[p] = ((2 * sy[p]) * PP + 2 * sx[p] + 2 * lh) * 8;
    toff[p] = (p * 128 + wave * 32 + l31 < FR * FC) ? m * FTP + 8 * lh : -1;
  }
  int f_pr[FNLOAD], f_pc[FNLOAD], f_lds[FNLOAD];
  uint32_t f_in = 0;
#pragma unroll
  for (int r = 0; r < FNLOAD; ++r) {
    const int idx = tid + 256 * r;
    f_pr[r] = idx / FPCOLS; f_pc[r] = idx - f_pr[r] * FPCOLS;
    f_lds[r] = (f_pr[r] * PP + f_pc[r]) * 8;
    f_in |= (idx < FPROWS * FPCOLS ? 1u : 0u) << r;
  }
  // pooled outputs of this lane: item tid + 256 r -> (pooled position, 8-channel group)
  int p_off[2], p_oy[2], p_ox[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int idx = tid + 256 * r, pp = idx >> 3;
    p_oy[r] = pp / FPC; p_ox[r] = pp - p_oy[r] * FPC;
    p_off[r] = pp < FPR * FPC ? ((2 * p_oy[r]) * FC + 2 * p_ox[r]) * FTP + (idx & 7) * 16 : -1;
  }
  char* const my_trash = trash + ((size_t)blockIdx.x * 256 + tid) * 16;

  auto tile_origin = [&](int tile, int& b, int& oy0, int& ox0) {
    const int tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y;
    b = t2 / tiles_y; oy0 = ty * FPR; ox0 = tx * FPC;
  };
  // v: RAW ? three fp32 planes per pixel (v[3 r + c]) : the packed bf16 pixel in (v[2 r], v[2 r + 1])
  auto fetch = [&](int tile_in, unsigned (&v)[3 * FNLOAD], uint32_t& okmask) {
    int b, oy0, ox0;
    tile_origin(tile_in < ntiles ? tile_in : ntiles - 1, b, oy0, ox0);
    const int iy0 = 4 * oy0 - 5, ix0 = 4 * ox0 - 5;  // stem row 2 oy0 - 1 reads input rows 2 (2 oy0 - 1) - 3 ..
    okmask = 0;
    if (RAW) {
      const ImageDesc d = descs[b];
      const size_t plane = (size_t)d.h * d.w;
#pragma unroll
      for (int r = 0; r < FNLOAD; ++r) {
        const int iy = iy0 + f_pr[r], ix = ix0 + f_pc[r];
        const bool ok = ((f_in >> r) & 1u) && (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;
        const int cy = min(max(iy, 0), d.h - 1), cx = min(max(ix, 0), d.w - 1);
        const float* src = d.ptr + (size_t)cy * d.w + cx;
        asm volatile("global_load_dword %0, %1, off" : "=v"(v[3 * r]) : "v"(src) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(v[3 * r + 1]) : "v"(src + plane) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(v[3 * r + 2]) : "v"(src + 2 * plane) : "memory");
        okmask |= (ok ? 1u : 0u) << r;
      }
    } else {
      const bf16_t* xb = x + (size_t)b * H * W * 4;
#pragma unroll
      for (int r = 0; r < FNLOAD; ++r) {
        const int iy = iy0 + f_pr[r], ix = ix0 + f_pc[r];
        const bool ok = ((f_in >> r) & 1u) && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
        const bf16_t* src = xb + (cy * W + cx) * 4;
        uint2 t;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(t) : "v"(src) : "memory");
        v[2 * r] = t.x; v[2 * r + 1] = t.y;
        okmask |= (ok ? 1u : 0u) << r;
      }
    }
  };
  auto park = [&](const unsigned (&v)[3 * FNLOAD], uint32_t okmask) {
#pragma unroll
    for (int r = 0; r < FNLOAD; ++r) {
      uint2 t = make_uint2(0u, 0u);
      if ((okmask >> r) & 1u) {
        if (RAW) {
          typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
          bf16x2 lo2;
          lo2[0] = (bf16_t)((__uint_as_float(v[3 * r]) - raw.m0) * raw.is0);
          lo2[1] = (bf16_t)((__uint_as_float(v[3 * r + 1]) - raw.m1) * raw.is1);
          bf16x2 hi2;
          hi2[0] = (bf16_t)((__uint_as_float(v[3 * r + 2]) - raw.m2) * raw.is2);
          hi2[1] = (bf16_t)0.f;
          t = make_uint2(__builtin_bit_cast(unsigned, lo2), __builtin_bit_cast(unsigned, hi2));
        } else {
          t = make_uint2(v[2 * r], v[2 * r + 1]);
        }
      }
      if ((f_in >> r) & 1u) *reinterpret_cast<uint2*>(patch + f_lds[r]) = t;
    }
  };
#pragma unroll
  for (int ks = 0; ks < 14; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(wb[ks][j]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (RAW) __syncthreads();  // the image table is complete before the first fetch reads it

  const int g = gridDim.x;
  int nth = 0;
  auto do_tile = [&](int tile, unsigned (&q)[3 * FNLOAD], uint32_t& qm) {
    // issued after this tile's patch loads (two tiles ago): 2 stores, the next tile's NV loads, 2 stores -- all may stay in flight
    // (the first two tiles have fewer operations behind their patch)
    static_assert(FNLOAD == 6, "the vmcnt immediates below are NV = 6 / 18 (+ 2 stores per finished tile)");
    if (nth == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV) : "memory");
    else if (nth == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV + 2) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV + 4) : "memory");
    ++nth;
    park(q, qm);
    lds_barrier();
    fetch(tile + 2 * g, q, qm);
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
    const int sy0 = 2 * oy0 - 1, sx0 = 2 * ox0 - 1;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f32x16 acc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bf16x8 fa = *reinterpret_cast<const bf16x8*>(patch + aoff[p] + (kh * PP + 4 * h) * 8);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[kh * 2 + h][j], fa, acc[j], 0, 0, 0);
        }
      // D^T layout: this lane holds channels 32 j + 8 q4 + 4 lh .. + 3 of its position
      const bool inimg = (unsigned)(sy0 + sy[p]) < (unsigned)H2 && (unsigned)(sx0 + sx[p]) < (unsigned)W2;
      const unsigned keep = inimg ? 0xffffffffu : 0u;
      if (toff[p] >= 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int n0 = 32 * j + 8 * q4 + 4 * lh;
            const f32x4 sv = *reinterpret_cast<const f32x4*>(ss + n0), bv = *reinterpret_cast<const f32x4*>(ss + 64 + n0);
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            u32x2 o;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              bf16x2 v2;
              v2[0] = (bf16_t)(acc[j][4 * q4 + 2 * hh] * sv[2 * hh] + bv[2 * hh]);
              v2[1] = (bf16_t)(acc[j][4 * q4 + 2 * hh + 1] * sv[2 * hh + 1] + bv[2 * hh + 1]);
              const s16x2 z = {0, 0};
              o[hh] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v2), z)) & keep;  // ReLU on the bf16 pair
            }
            *reinterpret_cast<u32x2*>(st + toff[p] + 64 * j + 16 * q4) = o;
          }
      }
    }
    lds_barrier();
    // pool: 9 taps of 16 B; non-negative bf16 order like their bit patterns, so the max is a packed unsigned 16-bit max
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
      u16x8 mx = {0, 0, 0, 0, 0, 0, 0, 0};
      if (p_off[r] >= 0) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            mx = __builtin_elementwise_max(mx, *reinterpret_cast<const u16x8*>(st + p_off[r] + (dy * FC + dx) * FTP));
      }
      const int oy = oy0 + p_oy[r], ox = ox0 + p_ox[r];
      const bool okp = p_off[r] >= 0 && oy < H4 && ox < W4;
      char* dst = okp ? reinterpret_cast<char*>(out) + ((((size_t)b * H4 + oy) * W4 + ox) * 64 + (tid & 7) * 8) * 2 : my_trash;
      *reinterpret_cast<u16x8*>(dst) = mx;  // unconditional: a fixed number of stores per tile keeps the vmcnt arithmetic exact
    }
    lds_barrier();  // patch and stem tile are rewritten by the next tile
  };

  unsigned qa[3 * FNLOAD], qb[3 * FNLOAD];
  uint32_t ma, mb;
  int tile = blockIdx.x;
  fetch(tile, qa, ma);
  fetch(tile + g, qb, mb);
  while (tile < ntiles) {
    do_tile(tile, qa, ma);
    tile += g;
    if (tile >= ntiles) break;
    do_tile(tile, qb, mb);
    tile += g;
  }
}

// out: [B][H4][W4][64] (the max-pooled stem); trash: >= grid x 256 x 16 B
int launch_stem_pool(const void* x, const void* wp, const float* scale, const float* shift, void* out, void* trash, int B, int H, int W,
                     int H2, int W2, int H4, int W4, hipStream_t s) {
  const int tiles_y = (H4 + FPR - 1) / FPR, tiles_x = (W4 + FPC - 1) / FPC, ntiles = B * tiles_y * tiles_x;
  const int grid = ntiles < 512 ? ntiles : 512;  // 2 persistent blocks per CU
  hipLaunchKernelGGL(stem_pool_kernel<false>, dim3(grid), dim3(256), FLDS, s, (const bf16_t*)x, StemRawArgs{}, (const bf16_t*)wp, scale, shift,
                     (bf16_t*)out, (char*)trash, H, W, H2, W2, H4, W4, tiles_y, tiles_x, ntiles);
  return (int)hipGetLastError();
}

// the same from the caller's raw images: imgs_dev[B] = (fp32 (3, h, w) planes, h, w), normalised with (p - mean) * (1 / std) on the way
int launch_stem_pool_raw(const ImageDesc* imgs_dev, const float* mean, const float* stdv, const void* wp, const float* scale,
                         const float* shift, void* out, void* trash, int B, int H, int W, int H2, int W2, int H4, int W4, hipStream_t s) {
  if (B > STEM_RAW_MAX_BATCH) return -1;
  const int tiles_y = (H4 + FPR - 1) / FPR, tiles_x = (W4 + FPC - 1) / FPC, ntiles = B * tiles_y * tiles_x;
  const int grid = ntiles < 512 ? ntiles : 512;  // 2 persistent blocks per CU
  const StemRawArgs raw{imgs_dev, B, mean[0], mean[1], mean[2], 1.f / stdv[0], 1.f / stdv[1], 1.f / stdv[2]};
  hipLaunchKernelGGL(stem_pool_kernel<true>, dim3(grid), dim3(256), FLDS + (size_t)B * sizeof(ImageDesc), s, (const bf16_t*)nullptr, raw,
                     (const bf16_t*)wp, scale, shift, (bf16_t*)out, (char*)trash, H, W, H2, W2, H4, W4, tiles_y, tiles_x, ntiles);
  return (int)hipGetLastError();
}

// wp: [64][7][8][4] bf16 (kernel column 7 and channel 3 zero); x: [B][H][W][4]; out: [B][H2][W2][64]
int launch_stem_conv(const void* x, const void* wp, const float* scale, const float* shift, void* out, int B, int H, int W, int H2,
                     int W2, hipStream_t s) {
  const int tiles_y = (H2 + 7) / 8, tiles_x = (W2 + 15) / 16, ntiles = B * tiles_y * tiles_x;
  const int grid = ntiles < 512 ? ntiles : 512;  // 2 persistent blocks per CU
  const size_t lds = PATCH_BYTES + (size_t)64 * SCP * 4;
  hipLaunchKernelGGL(stem_conv_kernel, dim3(grid), dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)wp, scale, shift, (bf16_t*)out,
                     H, W, H2, W2, tiles_y, tiles_x, ntiles);
  return (int)hipGetLastError();
}

}  // namespace sylph
