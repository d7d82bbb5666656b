// 3x3 stride-1 convolution 128 -> 128 channels + FrozenBN + ReLU with the WEIGHTS IN REGISTERS (round 5): conv2 of the res3
// bottleneck blocks (detectron2 BottleneckBlock.conv2 at the call site sylph/modeling/meta_arch/meta_one_stage_detector.py:181,273).
//
// On conv_igemm's halo mode these four launches ran at 0.41 of the MFMA peak: every 128-position tile streams the layer's 295 KB of
// weights from L2 into LDS again (2.6 GB of LDS-DMA per launch at B = 64 for 0.4 GB of activations).  The weights of 32 output
// channels are 72 MFMA fragments = 288 registers: one wave per SIMD can hold them (the design of bottleneck64_kernel's conv2):
//
//   * ONE persistent 256-thread block per CU, one wave per SIMD; wave w owns output channels 32 w .. 32 w + 31 and keeps their
//     fragments for the whole launch: taps 0..7 in 256 AGPRs (inline-asm MFMA with an AGPR operand), tap 8 in 32 VGPRs.
//   * LDS holds only activations: the (ph + 2) x (pw + 2) input halo of a <= 128-position patch, rows padded to 272 bytes (written by
//     ds_write, so the pad costs nothing and every fragment read is conflict-free with an immediate k offset), double-buffered: the
//     16-byte pieces of the NEXT patch's halo are loaded into registers at the top of a patch (out-of-image pieces become zeros: the
//     conv's padding) and written to the other buffer at its end.
//   * K loop: 9 taps x 8 k-steps, every wave walks all four 32-row tiles of the patch: 288 MFMAs on four independent accumulators,
//     288 ds_read_b128 through a 3-deep fragment ring (inline asm, counted lgkmcnt).
//   * epilogue: fma(acc, scale, shift) -> ReLU -> bf16 -> a block-wide LDS tile [128 rows][256 B] (piece p of row r at slot p ^ (r & 15))
//     -> 16-byte stores, 16 lanes per 256-byte row.  Two barriers per patch.
//
// Numerics: bf16 operands, fp32 accumulation over the taps in tap order, v = acc * scale + shift, ReLU, bf16: the rounding points of
// conv_igemm's halo mode (oracle/bf16.py conv_epilogue).
// Ablation switches (RW_NOSTORE, RW_NOHALO, RW_NOMFMA, RW_NOREAD: measurement aids) exist only in -DSYLPH_ABLATE builds (tools/build_variant.sh)
#ifndef SYLPH_ABLATE
#undef RW_NOSTORE
#undef RW_NOHALO
#undef RW_NOMFMA
#undef RW_NOREAD
#endif
#include <type_traits>
#include <utility>

#include "common.h"

namespace sylph {

namespace {
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
constexpr int RW_CH = 128;
constexpr int RW_HROWS = 192;                 // halo rows of a buffer
constexpr int RW_TP = 272;                    // halo row pitch: 256 B of channels + 16 B pad
constexpr int RW_HB = RW_HROWS * RW_TP;       // 52 224
constexpr int RW_STG = 2 * RW_HB;             // store staging [128][256 B]
constexpr int RW_TAB = RW_STG + 128 * 256;    // y byte offset of every patch position (512 B)
constexpr int RW_BN = RW_TAB + 512;           // scale[128] | shift[128] fp32
constexpr int RW_LDS = RW_BN + 2 * RW_CH * 4;  // 138 752
static_assert(RW_LDS <= 160 * 1024, "LDS budget");
constexpr int RW_NPC = 11;  // 16-byte halo pieces per thread: halos of up to 176 rows (conv_rw3_patch_ok); reads of pad positions stay inside RW_HROWS

#define RW_MFMA_A(acc, w, av) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(av))
#define RW_MFMA_A0(acc, w, av) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(av))
#define RW_MFMA_V(acc, w, av) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(av))
#define RW_BAR()                                       \
  do {                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                      \
    asm volatile("" ::: "memory");                     \
  } while (0)
}  // namespace

__global__ __launch_bounds__(256, 1) void conv_rw3_kernel(const BottleneckArgs a) {
  typedef bf16_t T;
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const char* __restrict__ x = reinterpret_cast<const char*>(a.x);
  char* __restrict__ y = reinterpret_cast<char*>(a.y);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;

  // ---- this wave's weights -> registers: output channels 32 wave + l31; k-step k = tap * 8 + ks covers input channels 16 ks + 8 lh .. ----
  bf16x8 Wa[64], Wv[8];
  {
    const T* wp = a.w2 + ((size_t)(wave * 32 + l31) * 9 * RW_CH + lh * 8);  // [Cout][3][3][Cin]: tap * 128 + ks * 16 == k * 16
#pragma unroll
    for (int k = 0; k < 64; ++k) Wa[k] = *reinterpret_cast<const bf16x8*>(wp + k * 16);
#pragma unroll
    for (int k = 0; k < 8; ++k) Wv[k] = *reinterpret_cast<const bf16x8*>(wp + (64 + k) * 16);
  }
  {
    float* bn = reinterpret_cast<float*>(smem + RW_BN);
    bn[tid] = tid < RW_CH ? a.s2[tid] : a.b2[tid - RW_CH];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // persistent tile walk: blocks of one XCD (blockIdx & 7) take neighbouring patches at the same time
  const int G = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, gx = (G + 7) >> 3;
  const int chunk = (a.n_tiles + 7) >> 3;
  auto tile_of = [&](int it) { const int q = it * gx + jb; return __builtin_amdgcn_readfirstlane(q < chunk ? xcd * chunk + q : a.n_tiles); };
  auto load_tile = [&](int t) {
    i32x8 v;
    const BkTile* p = a.bk + t;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
    return v;
  };
  // the 16-byte pieces of a patch's halo this thread carries: piece p = tid + 256 j = (halo row p >> 4, chunk p & 15); pieces outside
  // the image are zeros (the conv's padding), pieces past the halo are not written
  u32x4 hreg[RW_NPC];
  auto halo_load = [&](const i32x8 d) {
    const int row0 = d[0], H = d[1], W = d[2], oy0 = d[3] >> 16, ox0 = d[3] & 0xffff, HW2 = d[5] + 2, HR = (d[4] + 2) * HW2;
    const unsigned inv_hw2 = (unsigned)d[7];
#pragma unroll
    for (int j = 0; j < RW_NPC; ++j) {
      const int p = tid + 256 * j, h = p >> 4, c = p & 15;
      const int hy = (int)(((unsigned)h * inv_hw2) >> 16), hx = h - hy * HW2;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      const bool in = h < HR && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      hreg[j] = u32x4{0u, 0u, 0u, 0u};
      if (in) hreg[j] = *reinterpret_cast<const u32x4*>(x + ((size_t)(unsigned)(row0 + iy * W + ix) * (RW_CH * 2) + c * 16));
    }
  };
  auto halo_write = [&](int buf) {
#pragma unroll
    for (int j = 0; j < RW_NPC; ++j) {
      const int p = tid + 256 * j, h = p >> 4, c = p & 15;
      *reinterpret_cast<u32x4*>(smem + buf * RW_HB + h * RW_TP + c * 16) = hreg[j];
    }
  };
  auto relu_pk = [](unsigned u) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, u), z));
  };
  auto pack2 = [](float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v;
    v[0] = (bf16_t)lo;
    v[1] = (bf16_t)hi;
    return __builtin_bit_cast(unsigned, v);
  };

  int t = tile_of(0);
  if (t >= a.n_tiles) return;
  i32x8 td = load_tile(t);
  halo_load(td);
  halo_write(0);

  for (int it = 0; t < a.n_tiles; ++it) {
    const int row0 = td[0], IH = td[1], IW = td[2], oy0 = td[3] >> 16, ox0 = td[3] & 0xffff;
    const int PW = td[5], HW2 = PW + 2, NPOS = td[4] * PW;
    const unsigned inv_pw = (unsigned)td[6];
    const int t_next = tile_of(it + 1);
    const i32x8 td_next = load_tile(t_next < a.n_tiles ? t_next : t);
    RW_BAR();  // this patch's halo is in buffer it & 1 (written at the end of the previous iteration); the staging tile and the table are free
    // y byte offset of patch position m (0xffffffff: no such pixel), read after the staging barrier
    if (tid < 128) {
      const int m = tid;
      const int my = (int)(((unsigned)m * inv_pw) >> 16), mx = m - my * PW;
      const bool pv = m < NPOS && oy0 + my < IH && ox0 + mx < IW;
      *reinterpret_cast<unsigned*>(smem + RW_TAB + m * 4) = pv ? (unsigned)(row0 + (oy0 + my) * IW + ox0 + mx) * (unsigned)(RW_CH * 2) : 0xffffffffu;
    }
#ifndef RW_NOHALO
    if (t_next < a.n_tiles) halo_load(td_next);  // in flight under this patch's K loop
#endif

    // ===== K loop: acc[i] = sum over taps and channels of halo(position 32 i + l31 shifted by the tap) x W ================================
    f32x16 acc[4];
    {
      int lz;
      asm volatile("v_mov_b32 %0, 0" : "=v"(lz));
      unsigned hrow[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = i * 32 + l31 + lz;
        const int my = (int)(((unsigned)m * inv_pw) >> 16);
        hrow[i] = lds0 + (it & 1) * RW_HB + (my * HW2 + (m - my * PW)) * RW_TP + 16 * lh;
      }
      constexpr int D = 3;
      bf16x8 af[D][4];
      auto rd = [&](auto kc) {  // the four fragments of k-step k = tap * 8 + ks
        constexpr int k = decltype(kc)::value, tap = k >> 3, ks = k & 7, kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned ad = hrow[i] + (kh * HW2 + kw) * RW_TP;
          bf16x8& dst = af[k % D][i];
#ifdef RW_NOREAD
          asm volatile("" : "=v"(dst) : "v"(ad));
#else
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(ks * 32));
#endif
        }
      };
      auto step = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k + D - 1 < 72) rd(std::integral_constant<int, k + D - 1>{});
        constexpr int ahead = (k + D - 1 < 72 ? D - 1 : 71 - k) * 4;  // fragment reads issued after those of k-step k
        bf16x8* f = af[k % D];
        const bf16x8 *wa = Wa, *wv = Wv;  // (asm operands naming an enclosing local directly do not capture it in a generic lambda)
        f32x16* ac = acc;
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(ahead));
#ifdef RW_NOMFMA
        if constexpr (k == 0 || k == 71)
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (k == 0) RW_MFMA_A0(ac[i], wa[0], f[i]);
          else if constexpr (k < 64) RW_MFMA_A(ac[i], wa[k], f[i]);
          else RW_MFMA_V(ac[i], wv[k - 64], f[i]);
        }
      };
      rd(std::integral_constant<int, 0>{});
      rd(std::integral_constant<int, 1>{});
      static_for<0, 72>(step);
      // inline-asm MFMAs are invisible to the hazard recogniser: the wait states it would insert before the first VALU read of an accumulator
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])::"memory");
    }

    // ===== epilogue: FrozenBN + ReLU -> bf16 -> the staging tile (row m, piece 4 wave + g, half lh) ====================================
    {
      const float* sp = reinterpret_cast<const float*>(smem + RW_BN) + wave * 32 + 4 * lh;
      f32x4 sv[4], bv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        sv[g] = *reinterpret_cast<const f32x4*>(sp + 8 * g);
        bv[g] = *reinterpret_cast<const f32x4*>(sp + RW_CH + 8 * g);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = i * 32 + l31;
        char* wp = smem + RW_STG + m * 256 + lh * 8;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 o;
          o[0] = relu_pk(pack2(acc[i][4 * g] * sv[g][0] + bv[g][0], acc[i][4 * g + 1] * sv[g][1] + bv[g][1]));
          o[1] = relu_pk(pack2(acc[i][4 * g + 2] * sv[g][2] + bv[g][2], acc[i][4 * g + 3] * sv[g][3] + bv[g][3]));
          *reinterpret_cast<u32x2*>(wp + (((4 * wave + g) ^ (m & 15)) << 4)) = o;
        }
      }
    }
    RW_BAR();  // the staging tile is complete; every wave is done reading this patch's halo
    // 16-byte stores: thread -> (row q >> 4, slot q & 15) of the staging tile, 16 lanes per 256-byte row
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = tid + 256 * j, row = q >> 4, slot = q & 15;
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + RW_STG + q * 16);
      const unsigned yo = *reinterpret_cast<const unsigned*>(smem + RW_TAB + row * 4);
#ifdef RW_NOSTORE
      if (yo == 0xfffffff0u)
#else
      if (yo != 0xffffffffu)
#endif
        *reinterpret_cast<u32x4*>(y + ((size_t)yo + ((slot ^ (row & 15)) << 4))) = v;
    }
#ifndef RW_NOHALO
    if (t_next < a.n_tiles) halo_write((it + 1) & 1);  // the other buffer: last read in the previous patch's K loop
#endif
    t = t_next;
    td = td_next;
  }
}

bool conv_rw3_patch_ok(int ph, int pw) { return ph * pw <= 128 && (ph + 2) * (pw + 2) <= RW_NPC * 16; }

int launch_conv_rw3(const BottleneckArgs& a, hipStream_t s) {
  static bool attr_set = false;
  static int ncu = 256;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_rw3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RW_LDS) != hipSuccess) return -7;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      ncu = prop.multiProcessorCount;
    attr_set = true;
  }
  const int want = (a.n_tiles + 7) & ~7;
  const int grid = want < ncu ? want : (ncu & ~7);
  hipLaunchKernelGGL(conv_rw3_kernel, dim3(grid), dim3(256), RW_LDS, s, a);
  return (int)hipGetLastError();
}

}  // namespace sylph
