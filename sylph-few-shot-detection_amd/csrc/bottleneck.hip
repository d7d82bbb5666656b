// Fused ResNet identity bottleneck for the HBM-bound stage res2 (C = 256, mid = 64), bf16:
//
//     y = relu(x + bn3(conv1x1_{64->256}( relu(bn2(conv3x3_{64->64}( relu(bn1(conv1x1_{256->64}(x))) ))) )))
//
// (detectron2 BottleneckBlock with FrozenBN, stride 1, no projection: blocks 1.. of a stage; call site
// sylph/modeling/meta_arch/meta_one_stage_detector.py:75,181,273.)  Unfused, the block moves 2 048 B per position through
// HBM (x is read twice -- conv1 input and residual --, the two 64-channel intermediates are written and read back) and its
// three launches sit at the 5.3-5.5 TB/s ceiling.  Here one block owns a ph x pw patch (<= 128 positions) and keeps
// everything between x and y on chip: 1 024 B per position (x once + the L2-hot re-read of the residual tile, y once).
//
//   P1  t1 = relu(bn1(x_halo . W1^T)) on the (ph+2) x (pw+2) halo of the patch (conv1 is RECOMPUTED on the halo ring,
//       +40 % of its flops, instead of exchanging t1 through HBM); GEMM 192 x 64 x 256, K in 8 stages of 32 channels;
//       halo positions outside the image are forced to 0 (they are conv2's zero padding).  t1 -> LDS, bf16.
//   P2  t2 = relu(bn2(conv3x3(t1))): the nine taps read shifted rows of the t1 halo in LDS (conv_igemm.hip HALO addressing);
//       GEMM 128 x 64 x 576, weights in 5 stages of two taps.  t2 -> LDS (over t1), bf16.
//   P3  y = relu(bn3(t2 . W3^T) + x): GEMM 128 x 256 x 64 in two 128-channel stages, epilogue through an fp32 LDS tile with
//       16-byte residual loads and stores.
//
// One software pipeline runs through all 15 stages: every stage is 4 global_load_lds per lane into a ring of three 16.5-KiB
// LDS slots, issued two stages ahead and retired with a COUNTED vmcnt(4); one barrier per stage.  256 threads (4 waves),
// 75 KiB of LDS -> two blocks per CU (the second block computes while this one waits or runs an epilogue).
// MFMA operands are swapped (D^T) as in conv_igemm.hip: a lane owns 4 consecutive channels of one position.
#include "common.h"

namespace sylph {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

namespace {
constexpr int MID = 64, C = 256;
constexpr int T1_ROWS = 192;                 // halo rows reserved (>= (ph+2)*(pw+2), <= 184 used)
constexpr int T1_BYTES = T1_ROWS * 128;      // 24 576: t1 halo [192][64 ch]; t2 [128][64 ch] aliases it
constexpr int SLOT = 32 * (128 + 4) * 4;     // 16 896: one ring slot (>= 16 KiB stage; also one fp32 epilogue pass 32 x 132)
constexpr int RING_OFF = T1_BYTES;
constexpr int BN_OFF = RING_OFF + 3 * SLOT;  // s1 b1 s2 b2 (64 each) s3 b3 (256 each), fp32
constexpr int LDS_BYTES = BN_OFF + (4 * MID + 2 * C) * 4;  // 78 336
constexpr int NST1 = C / 32, NST2 = 5, NST3 = 2;  // 8 + 5 + 2 stages
constexpr int W1S_OFF = T1_ROWS * 64;        // inside a P1 slot: x halo slice [192][64 B], then W1 slice [64][64 B]

#define BK_WAITV(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define BK_BAR()                                      \
  do {                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                     \
    asm volatile("" ::: "memory");                    \
  } while (0)
}  // namespace

__global__ __launch_bounds__(256, 2) void bottleneck64_kernel(const BottleneckArgs a) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware block -> tile map: XCD x owns a contiguous chunk of patches (neighbouring halos share that XCD's L2)
  const int L = blockIdx.x;
  const int xcd = L & 7, q0 = L >> 3;
  const int chunk = (a.n_tiles + 7) >> 3;
  const int mt = xcd * chunk + q0;
  if (q0 >= chunk || mt >= a.n_tiles) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;

  const int2 tile = a.tiles[mt];
  const SegDesc sd = a.segs[tile.x];
  const int oy0 = tile.y >> 16, ox0 = tile.y & 0xffff;
  const int PW = sd.pw, HW2 = sd.hpitch, HR = (sd.ph + 2) * HW2, NPOS = sd.ph * sd.pw;

  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ zero = reinterpret_cast<const T*>(a.zeros);
  char* const t1 = smem;
  float* const bn = reinterpret_cast<float*>(smem + BN_OFF);

  // ---- FrozenBN scale / shift tables -> LDS (plain loads, before any LDS-DMA is in flight) -------------------------
  for (int i = tid; i < 4 * MID + 2 * C; i += 256) {
    const float* src = i < MID ? a.s1 + i : i < 2 * MID ? a.b1 + (i - MID) : i < 3 * MID ? a.s2 + (i - 2 * MID)
                     : i < 4 * MID ? a.b2 + (i - 3 * MID) : i < 4 * MID + C ? a.s3 + (i - 4 * MID) : a.b3 + (i - 4 * MID - C);
    bn[i] = *src;
  }
  const float *s1 = bn, *b1 = bn + MID, *s2 = bn + 2 * MID, *b2 = bn + 3 * MID, *s3 = bn + 4 * MID, *b3 = bn + 4 * MID + C;

  // ---- loader state ------------------------------------------------------------------------------------------------
  // 64-byte rows (P1): lane (r4, s4) fills 16-byte slot s4 of row (round * 64 + r4); slot s holds chunk s ^ ((row >> 2) & 3)
  const int r4 = tid >> 2, s4 = tid & 3;
  int xsrc[3];
  unsigned xmask = 0;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int h = t * 64 + r4;
    const int hy = (int)(((unsigned)h * sd.inv_hw2) >> 16), hx = h - hy * HW2;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    const bool ok = h < HR && (unsigned)iy < (unsigned)sd.in_H && (unsigned)ix < (unsigned)sd.in_W;
    xsrc[t] = (sd.in_row0 + iy * sd.in_W + ix) * C + (s4 ^ ((h >> 2) & 3)) * 8;
    xmask |= (ok ? 1u : 0u) << t;
  }
  const int w1src = r4 * C + (s4 ^ ((r4 >> 2) & 3)) * 8;  // W1 [64][256]: row r4, one round
  // 128-byte rows (P2, P3): lane (r0, c16) fills slot c16 of row (round * 32 + r0); slot s holds chunk s ^ ((row >> 1) & 7)
  const int r0 = tid >> 3, c16 = tid & 7;
  const int cl = c16 ^ ((r0 >> 1) & 7);  // (row >> 1) & 7 == (r0 >> 1) & 7 for row = round * 32 + r0

  auto issue = [&](int s) {  // the 4 loads of stage s into ring slot s % 3
    char* d = smem + RING_OFF + (s % 3) * SLOT + wave * 1024;  // wave-uniform; lane l lands at +16 l
    if (s < NST1) {
      const int k0 = s * 32;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const T* src = ((xmask >> t) & 1u) ? x + (xsrc[t] + k0) : zero + s4 * 8;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(d + t * 4096), 16, 0, 0);
      }
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a.w1 + (w1src + k0)), (lds_ptr_t)(d + W1S_OFF), 16, 0, 0);
    } else if (s < NST1 + NST2) {
      const int j = s - NST1;
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // taps 2j, 2j+1 (the ninth stage half re-reads tap 8: constant load count)
        const int tap = 2 * j + (u >> 1) > 8 ? 8 : 2 * j + (u >> 1);
        const int n = (u & 1) * 32 + r0;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a.w2 + ((n * 9 + tap) * MID + cl * 8)), (lds_ptr_t)(d + u * 4096), 16, 0, 0);
      }
    } else {
      const int q = s - NST1 - NST2;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = q * 128 + u * 32 + r0;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a.w3 + (n * MID + cl * 8)), (lds_ptr_t)(d + u * 4096), 16, 0, 0);
      }
    }
  };

  // ---- per-lane geometry of the rows this lane produces / reads -------------------------------------------------------
  // P1 output rows: halo rows h = rt * 32 + l31 for rt = wave and rt = 4 + (wave >> 1)
  int t1row[2], t1swz[2];
  bool t1in[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int h = (i == 0 ? wave : 4 + (wave >> 1)) * 32 + l31;
    const int hy = (int)(((unsigned)h * sd.inv_hw2) >> 16), hx = h - hy * HW2;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    t1row[i] = h;
    t1swz[i] = ((hy * PW + hx) >> 1) & 7;
    t1in[i] = h < HR && hx < PW + 2 && (unsigned)iy < (unsigned)sd.in_H && (unsigned)ix < (unsigned)sd.in_W;
  }
  // P2 / P3 rows: patch position m = wave * 32 + l31
  const int m = wave * 32 + l31;
  const int my = (int)(((unsigned)m * sd.inv_pw) >> 16), mx = m - my * PW;
  const int hb = my * HW2 + mx;

  const int swz4 = (l31 >> 2) & 3, swz8 = (l31 >> 1) & 7;

  // ---- pipeline ------------------------------------------------------------------------------------------------------
  issue(0);
  issue(1);

  // ===== P1: t1 = relu(bn1(x_halo . W1^T)) ==============================================================================
  f32x16 acc1[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
  const int rtB = 4 + (wave >> 1), ctB = wave & 1;  // the wave's third tile
#pragma unroll 1
  for (int s = 0; s < NST1; ++s) {
    BK_WAITV(4);
    BK_BAR();
    issue(s + 2);
    const char* st = smem + RING_OFF + (s % 3) * SLOT;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = ((ks * 2 + lh) ^ swz4) << 4;
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(st + (wave * 32 + l31) * 64 + so);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(st + (rtB * 32 + l31) * 64 + so);
      const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(st + W1S_OFF + l31 * 64 + so);
      const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(st + W1S_OFF + (32 + l31) * 64 + so);
      acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a0, acc1[0], 0, 0, 0);
      acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a0, acc1[1], 0, 0, 0);
      acc1[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ctB ? w1 : w0, a1, acc1[2], 0, 0, 0);
    }
  }
  // t1 -> LDS (bf16, 128-byte rows, halo swizzle keyed on k = hy * pw + hx); positions outside the image are conv2's padding
  {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ri = i < 2 ? 0 : 1, ct = i < 2 ? i : ctB;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = ct * 32 + 8 * g + 4 * lh;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc1[i][4 * g + e] * s1[n0 + e] + b1[n0 + e];
          v = (v > 0.f && t1in[ri]) ? v : 0.f;
          o[e] = (bf16_t)v;
        }
        *reinterpret_cast<bf16x4*>(t1 + t1row[ri] * 128 + (((ct * 4 + g) ^ t1swz[ri]) << 4) + 8 * lh) = o;
      }
    }
  }

  // ===== P2: t2 = relu(bn2(conv3x3(t1))) ================================================================================
  f32x16 acc2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
#pragma unroll 1
  for (int j = 0; j < NST2; ++j) {
    const int s = NST1 + j;
    BK_WAITV(4);
    BK_BAR();  // (also publishes t1 before the first tap)
    issue(s + 2);
    const char* st = smem + RING_OFF + (s % 3) * SLOT;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      const int tap = 2 * j + tl;
      if (tap > 8) break;
      const int kh = tap / 3, kw = tap - 3 * kh;
      const char* arow = t1 + (hb + kh * HW2 + kw) * 128;
      const int asw = ((l31 + kh * PW + kw) >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ch = ks * 2 + lh;
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + ((ch ^ asw) << 4));
        const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(st + tl * 8192 + l31 * 128 + ((ch ^ swz8) << 4));
        const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(st + tl * 8192 + (32 + l31) * 128 + ((ch ^ swz8) << 4));
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, av, acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, av, acc2[1], 0, 0, 0);
      }
    }
  }
  BK_BAR();  // every wave has finished reading t1: t2 may overwrite it
  {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    const int msw = (m >> 1) & 7;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = ct * 32 + 8 * g + 4 * lh;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc2[ct][4 * g + e] * s2[n0 + e] + b2[n0 + e];
          o[e] = (bf16_t)(v > 0.f ? v : 0.f);
        }
        *reinterpret_cast<bf16x4*>(t1 + m * 128 + (((ct * 4 + g) ^ msw) << 4) + 8 * lh) = o;
      }
  }

  // ===== P3: y = relu(bn3(t2 . W3^T) + x) ===============================================================================
  T* __restrict__ y = reinterpret_cast<T*>(a.y);
  const int c8 = tid & 15, rr = tid >> 4;  // epilogue: lane = 8 channels of row rr (+16) of a 32-row pass
#pragma unroll 1
  for (int q = 0; q < NST3; ++q) {
    const int s = NST1 + NST2 + q;
    if (q + 1 < NST3) BK_WAITV(4); else BK_WAITV(0);
    BK_BAR();  // (also publishes t2 before the first stage)
    const char* st = smem + RING_OFF + (s % 3) * SLOT;
    f32x16 acc3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[i][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ch = ks * 2 + lh;
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(t1 + m * 128 + ((ch ^ swz8) << 4));
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(st + (ct * 32 + l31) * 128 + ((ch ^ swz8) << 4));
        acc3[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, av, acc3[ct], 0, 0, 0);
      }
    }
    // epilogue of this 128-channel half: one 32-row pass per wave through the free ring slot ((s + 1) % 3 for q = 0, s - 1 for q = 1)
    float* const sC = reinterpret_cast<float*>(smem + RING_OFF + ((q == 0 ? s + 2 : s + 1) % 3) * SLOT);
    const int nb = q * 128 + c8 * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = s3[nb + e]; sh[e] = b3[nb + e]; }
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
      lds_barrier();
      if (wave == p) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(sC + l31 * 132 + ct * 32 + 8 * g + 4 * lh) =
                make_float4(acc3[ct][4 * g], acc3[ct][4 * g + 1], acc3[ct][4 * g + 2], acc3[ct][4 * g + 3]);
      }
      lds_barrier();
      int posv[2];
      bool pv[2];
      uint4 rraw[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {  // all residual loads of the pass first, then the math and the stores
        const int mm = p * 32 + rr + 16 * it;
        const int yy = (int)(((unsigned)mm * sd.inv_pw) >> 16), xx = mm - yy * PW;
        const int oy = oy0 + yy, ox = ox0 + xx;
        pv[it] = mm < NPOS && oy < sd.out_H && ox < sd.out_W;
        posv[it] = (sd.out_row0 + oy * sd.out_W + ox) * C + nb;
        rraw[it] = pv[it] ? *reinterpret_cast<const uint4*>(x + posv[it]) : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        if (!pv[it]) continue;
        const int rl = rr + 16 * it;
        const float4 lo = *reinterpret_cast<const float4*>(sC + rl * 132 + c8 * 8);
        const float4 hi = *reinterpret_cast<const float4*>(sC + rl * 132 + c8 * 8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const uint4 r4v = rraw[it];
        const unsigned rw[4] = {r4v.x, r4v.y, r4v.z, r4v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float res = __uint_as_float((e & 1) ? (rw[e >> 1] & 0xffff0000u) : (rw[e >> 1] << 16));
          const float t = v[e] * sc[e] + sh[e] + res;
          v[e] = t > 0.f ? t : 0.f;
        }
        store8<T>(y + posv[it], v);
      }
    }
    if (q + 1 < NST3) lds_barrier();  // the pass buffer is the slot stage s + ... may not be reused before everyone left it
  }
}

int launch_bottleneck64(const BottleneckArgs& a, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)bottleneck64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -7;
    attr_set = true;
  }
  const int chunk = (a.n_tiles + 7) / 8;
  hipLaunchKernelGGL(bottleneck64_kernel, dim3(8 * chunk), dim3(256), LDS_BYTES, s, a);
  return (int)hipGetLastError();
}


// =====================================================================================================================
// v2: persistent, one 512-thread block per CU.  v1 above keeps <= 48 KiB of x loads in flight per CU and is HBM-LATENCY
// bound (measured 1.73 ms per block at B = 64 vs 1.83 ms for the three separate launches).  Here the WHOLE x halo of the
// NEXT tile (192 rows x 512 B = 96 KiB) is in flight while the current tile runs conv2 / conv3:
//
//   * LDS: x halo [192][256 ch] 96 KiB | t1 / t2 24 KiB | four 8.5-KiB weight slots | 2 x 2 KiB hand-off tile | BN tables.
//   * Load roles (vmcnt is per wave and retires in order, so mixing streams in one wave would serialise them):
//       wave 0      streams the weights: 17 stages of 8 KiB per tile (4 x W1, 9 x W2 taps, 4 x W3), three stages ahead,
//                   counted vmcnt(16); it never issues a store or any other VMEM instruction;
//       waves 4-7   fetch the next tile's x halo (24 loads per lane) right after conv1 has consumed the current one, and
//                   wait for it (vmcnt(0)) only at the next tile's first stage;
//       all waves   compute in every phase.
//   * The residual is NOT re-read from memory: after conv1 every lane copies the 64 x-values its conv3 outputs will need from
//     the LDS halo into 32 registers (then the halo buffer is free for the next tile).  HBM traffic: x once, y once.
//   * conv3 epilogue in registers (D^T layout: a lane owns 4 consecutive channels of a position): bn3 + residual + ReLU, 8-byte
//     stores.  Wave 0 may not store (it would break its load count): its 32 x 32 tile goes through a 2-KiB LDS hand-off to
//     wave 4, which owns the same positions.
namespace {
constexpr int P_XROWS = 192;
constexpr int P_XBUF = P_XROWS * 512;          // 98 304
constexpr int P_T1 = P_XBUF;                   // 24 576
constexpr int P_WSLOT = 8448;                  // 8 KiB stage (+256)
constexpr int P_WR = P_T1 + T1_BYTES;          // 4 slots
constexpr int P_STG = P_WR + 4 * P_WSLOT;      // 2 x 2 KiB
constexpr int P_BN = P_STG + 4096;
constexpr int P_LDS = P_BN + (4 * MID + 2 * C) * 4;  // 163 840 = all of the CU
static_assert(P_LDS <= 163840, "bottleneck v2 LDS budget");
constexpr int NSTG = 17;
}  // namespace

__global__ __launch_bounds__(512, 1) void bottleneck64p_kernel(const BottleneckArgs a) {
  typedef bf16_t T;
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));  // ext_vector LDS accesses: hipcc adds no vmcnt(0) for them beside LDS-DMA; HIP's uint2 struct gets one
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ zero = reinterpret_cast<const T*>(a.zeros);
  T* __restrict__ y = reinterpret_cast<T*>(a.y);
  char* const xb = smem;
  char* const t1 = smem + P_T1;
  float* const bn = reinterpret_cast<float*>(smem + P_BN);

  for (int i = tid; i < 4 * MID + 2 * C; i += 512) {
    const float* src = i < MID ? a.s1 + i : i < 2 * MID ? a.b1 + (i - MID) : i < 3 * MID ? a.s2 + (i - 2 * MID)
                     : i < 4 * MID ? a.b2 + (i - 3 * MID) : i < 4 * MID + C ? a.s3 + (i - 4 * MID) : a.b3 + (i - 4 * MID - C);
    bn[i] = *src;
  }
  const float *s1 = bn, *b1 = bn + MID, *s2 = bn + 2 * MID, *b2 = bn + 3 * MID, *s3 = bn + 4 * MID, *b3 = bn + 4 * MID + C;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the plain loads above are the only non-DMA loads of the kernel

  // persistent tile walk: blocks of one XCD (blockIdx & 7) take neighbouring patches at the same time
  const int G = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, gx = (G + 7) >> 3;
  const int chunk = (a.n_tiles + 7) >> 3;
  auto tile_of = [&](int it) { const int q = it * gx + jb; return __builtin_amdgcn_readfirstlane(q < chunk ? xcd * chunk + q : a.n_tiles); };

  // ---- weight stream (wave 0): stage g of the endless sequence -> slot g & 3 ------------------------------------------
  const int wr_row = lane >> 3, wr_c = lane & 7;
  auto issue_w = [&](int g) {
    const int s = g % NSTG;
    char* d = smem + P_WR + (g & 3) * P_WSLOT;  // lane l lands at +16 l of each 1-KiB piece
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = u * 8 + wr_row;
      const int cl = wr_c ^ ((n >> 1) & 7);
      const T* src = s < 4 ? a.w1 + (n * C + 64 * s + cl * 8)
                   : s < 13 ? a.w2 + ((n * 9 + (s - 4)) * MID + cl * 8)
                            : a.w3 + ((64 * (s - 13) + n) * MID + cl * 8);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(d + u * 1024), 16, 0, 0);
    }
  };
  // ---- x halo (waves 4-7): 24 rounds of 8 rows x 512 B; slot s of row r holds chunk s ^ (r & 31) ----------------------
  auto issue_x = [&](const i32x8 d) {
    const int row0 = d[0], H = d[1], W = d[2], oy0 = d[3] >> 16, ox0 = d[3] & 0xffff, HW2 = d[5] + 2, HR = (d[4] + 2) * HW2;
    const unsigned inv_hw2 = (unsigned)d[7];
    const int tt = tid - 256, rr8 = tt >> 5, sl = tt & 31;
#pragma unroll 4
    for (int r = 0; r < 24; ++r) {
      const int h = r * 8 + rr8;
      const int hy = (int)(((unsigned)h * inv_hw2) >> 16), hx = h - hy * HW2;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      const bool ok = h < HR && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const T* src = ok ? x + ((size_t)(row0 + iy * W + ix) * C + (sl ^ (h & 31)) * 8) : zero + (sl & 3) * 8;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(xb + r * 4096 + (wave - 4) * 1024), 16, 0, 0);
    }
  };

  // tile descriptor through the SCALAR cache (a vector load here would put a vmcnt(0) into every wave's stream: hipcc
  // will not use s_load for memory it cannot prove read-only)
  auto load_tile = [&](int t) {
    i32x8 v;
    const BkTile* p = a.bk + t;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
    return v;
  };

  int g = 0;  // weight stage counter (never reset)
  if (wave == 0) { issue_w(0); issue_w(1); issue_w(2); }
  int t = tile_of(0);
  i32x8 td = load_tile(t < a.n_tiles ? t : 0);
  if (wave >= 4 && t < a.n_tiles) issue_x(td);

  const int swz8 = (l31 >> 1) & 7;
  const int rt = wave & 3, ctw = wave >> 2;  // P2 / P3 tile of this wave; P1: waves 0-3 row tile `wave` (both column tiles), waves 4-7 one tile
  const int rt1 = wave < 4 ? wave : 4 + ((wave - 4) >> 1), ct1 = wave < 4 ? 0 : (wave - 4) & 1;

  for (int it = 0; t < a.n_tiles; ++it) {
    const int row0 = td[0], IH = td[1], IW = td[2], oy0 = td[3] >> 16, ox0 = td[3] & 0xffff;
    const int PW = td[5], HW2 = PW + 2, HR = (td[4] + 2) * HW2, NPOS = td[4] * PW;
    const unsigned inv_pw = (unsigned)td[6], inv_hw2 = (unsigned)td[7];
    const int t_next = tile_of(it + 1);
    const i32x8 td_next = load_tile(t_next < a.n_tiles ? t_next : 0);

    // per-lane geometry
    const int h1 = rt1 * 32 + l31;  // P1 output halo row
    const int h1y = (int)(((unsigned)h1 * inv_hw2) >> 16), h1x = h1 - h1y * HW2;
    const bool in1 = h1 < HR && (unsigned)(oy0 - 1 + h1y) < (unsigned)IH && (unsigned)(ox0 - 1 + h1x) < (unsigned)IW;
    const int sw1 = ((h1y * PW + h1x) >> 1) & 7;
    const int m = rt * 32 + l31;  // P2 / P3 position
    const int my = (int)(((unsigned)m * inv_pw) >> 16), mx = m - my * PW;
    const int hb = my * HW2 + mx;
    const bool pv = m < NPOS && oy0 + my < IH && ox0 + mx < IW;
    const size_t ypos = (size_t)(row0 + (oy0 + my) * IW + ox0 + mx) * C;

    // ===== P1 ==========================================================================================================
    f32x16 acc1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
#pragma unroll 1
    for (int s = 0; s < 4; ++s, ++g) {
      if (wave == 0) BK_WAITV(16);
      else if (wave >= 4 && s == 0) BK_WAITV(0);  // this tile's x halo (and the previous tile's stores)
      BK_BAR();
      if (wave == 0) issue_w(g + 3);
      const char* st = smem + P_WR + (g & 3) * P_WSLOT;
      const int row = rt1 * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cx = s * 8 + ks * 2 + lh, cw = ks * 2 + lh;
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(xb + row * 512 + ((cx ^ (row & 31)) << 4));
        const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(st + (ct1 * 32 + l31) * 128 + ((cw ^ swz8) << 4));
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, av, acc1[0], 0, 0, 0);
        if (wave < 4) {
          const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(st + (32 + l31) * 128 + ((cw ^ swz8) << 4));
          acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, av, acc1[1], 0, 0, 0);
        }
      }
    }
    {  // t1 -> LDS; residual values of this lane's conv3 outputs -> registers
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i == 1 && wave >= 4) break;
        const int ct = wave < 4 ? i : ct1;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int n0 = ct * 32 + 8 * gq + 4 * lh;
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc1[i][4 * gq + e] * s1[n0 + e] + b1[n0 + e];
            v = (v > 0.f && in1) ? v : 0.f;
            o[e] = (bf16_t)v;
          }
          *reinterpret_cast<bf16x4*>(t1 + h1 * 128 + (((ct * 4 + gq) ^ sw1) << 4) + 8 * lh) = o;
        }
      }
    }
    u32x2 res[16];
    {
      const int hc = (my + 1) * HW2 + mx + 1;  // halo row of the centre position m
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int cch = q * 8 + ctw * 4 + gq;  // 16-byte chunk of channels q*64 + ctw*32 + 8 gq ..
          res[q * 4 + gq] = *reinterpret_cast<const u32x2*>(xb + hc * 512 + ((cch ^ (hc & 31)) << 4) + 8 * lh);
        }
      // Re-define the 16 values through an asm: hipcc otherwise ties their first USE (conv3 epilogue, two phases later) to
      // vmcnt(0) -- they were read from the LDS-DMA target buffer -- which would drain the weight stream / the x prefetch.
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]), "+v"(res[5]), "+v"(res[6]), "+v"(res[7]),
                     "+v"(res[8]), "+v"(res[9]), "+v"(res[10]), "+v"(res[11]), "+v"(res[12]), "+v"(res[13]), "+v"(res[14]), "+v"(res[15]));
    }

    // ===== P2 ==========================================================================================================
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap, ++g) {
      if (wave == 0) BK_WAITV(16);
      BK_BAR();  // (first tap: t1 published, everyone is done with the x halo)
      if (wave == 0) issue_w(g + 3);
      if (tap == 0 && wave >= 4 && t_next < a.n_tiles) issue_x(td_next);
      const char* st = smem + P_WR + (g & 3) * P_WSLOT;
      const int kh = tap / 3, kw = tap - 3 * kh;
      const char* arow = t1 + (hb + kh * HW2 + kw) * 128;
      const int asw = ((l31 + kh * PW + kw) >> 1) & 7;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ch = ks * 2 + lh;
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + ((ch ^ asw) << 4));
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(st + (ctw * 32 + l31) * 128 + ((ch ^ swz8) << 4));
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, av, acc2, 0, 0, 0);
      }
    }
    BK_BAR();  // every wave has finished reading t1: t2 may overwrite it
    {
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
      const int msw = (m >> 1) & 7;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n0 = ctw * 32 + 8 * gq + 4 * lh;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc2[4 * gq + e] * s2[n0 + e] + b2[n0 + e];
          o[e] = (bf16_t)(v > 0.f ? v : 0.f);
        }
        *reinterpret_cast<bf16x4*>(t1 + m * 128 + (((ctw * 4 + gq) ^ msw) << 4) + 8 * lh) = o;
      }
    }

    // ===== P3 ==========================================================================================================
#pragma unroll 1
    for (int q = 0; q < 4; ++q, ++g) {
      if (wave == 0) BK_WAITV(16);
      BK_BAR();  // (q = 0: t2 published; q > 0: hand-off tile q-1 published)
      if (wave == 0) issue_w(g + 3);
      if (q > 0 && wave == 4 && pv) {  // wave 0's tile of the previous stage (same positions, channels 32 lower)
        const char* sg = smem + P_STG + ((q - 1) & 1) * 2048;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          *reinterpret_cast<u32x2*>(y + ypos + (q - 1) * 64 + 8 * gq + 4 * lh) = *reinterpret_cast<const u32x2*>(sg + l31 * 64 + gq * 16 + 8 * lh);
      }
      const char* st = smem + P_WR + (g & 3) * P_WSLOT;
      f32x16 acc3;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int ch = ks * 2 + lh;
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(t1 + m * 128 + ((ch ^ swz8) << 4));
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(st + (ctw * 32 + l31) * 128 + ((ch ^ swz8) << 4));
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, av, acc3, 0, 0, 0);
      }
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n0 = q * 64 + ctw * 32 + 8 * gq + 4 * lh;
        const u32x2 rv = res[q * 4 + gq];
        const float r4[4] = {__uint_as_float(rv[0] << 16), __uint_as_float(rv[0] & 0xffff0000u), __uint_as_float(rv[1] << 16),
                             __uint_as_float(rv[1] & 0xffff0000u)};
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc3[4 * gq + e] * s3[n0 + e] + b3[n0 + e] + r4[e];
          o[e] = (bf16_t)(v > 0.f ? v : 0.f);
        }
        if (wave == 0) *reinterpret_cast<bf16x4*>(smem + P_STG + (q & 1) * 2048 + l31 * 64 + gq * 16 + 8 * lh) = o;
        else if (pv) *reinterpret_cast<bf16x4*>(y + ypos + n0) = o;
      }
    }
    // the last hand-off tile (q = 3) is stored after the next barrier: the first stage of the next tile, or here at the end
    BK_BAR();
    if (wave == 4 && pv) {
      const char* sg = smem + P_STG + 1 * 2048;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<u32x2*>(y + ypos + 3 * 64 + 8 * gq + 4 * lh) = *reinterpret_cast<const u32x2*>(sg + l31 * 64 + gq * 16 + 8 * lh);
    }
    t = t_next;
    td = td_next;
  }
}

int launch_bottleneck64p(const BottleneckArgs& a, hipStream_t s) {
  static bool attr_set = false;
  static int ncu = 256;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)bottleneck64p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS) != hipSuccess) return -7;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      ncu = prop.multiProcessorCount;
    attr_set = true;
  }
  const int grid = a.n_tiles < ncu ? a.n_tiles : ncu;
  hipLaunchKernelGGL(bottleneck64p_kernel, dim3(grid), dim3(512), P_LDS, s, a);
  return (int)hipGetLastError();
}

}  // namespace sylph
