// 3x3 stride-1 convolution 128 -> 128 channels + FrozenBN + ReLU with the WEIGHTS IN REGISTERS (round 5): conv2 of the res3
// bottleneck blocks (detectron2 BottleneckBlock.conv2 at the call site sylph/modeling/meta_arch/meta_one_stage_detector.py:181,273).
//
// On conv_igemm's halo mode these four launches ran at 0.41 of the MFMA peak (323 us at B = 64): every 128-position tile streams the
// layer's 295 KB of weights from L2 into LDS again (2.6 GB of LDS-DMA per launch for 0.4 GB of activations).  The weights of 32 output
// channels are 72 MFMA fragments = 288 registers: one wave per SIMD can hold them (the design of bottleneck64_kernel's conv2):
//
//   * ONE persistent 256-thread block per CU, one wave per SIMD; wave w owns output channels 32 w .. 32 w + 31 and keeps their
//     fragments for the whole launch: taps 0..7 in 256 AGPRs (inline-asm MFMA with an AGPR operand), tap 8 in 32 VGPRs.
//   * LDS holds only activations: the (ph + 2) x (pw + 2) input halo of a <= 128-position patch, pixels padded to 272 bytes and rows
//     of pixels to a pitch that makes every fragment read bank-conflict-free (rw_row_pitch), double-buffered.  The halo travels through
//     registers (buffer loads: padding and ragged edges are out-of-range offsets that read as zeros) and is written with ds_write_b128.
//   * K loop: 9 taps x 8 k-steps, every wave walks all four 32-row tiles of the patch: 288 MFMAs on four independent accumulators,
//     288 ds_read_b128 through a 3-deep fragment ring (inline asm, counted lgkmcnt), k offsets as instruction immediates.
//   * epilogue: fma(acc, scale, shift) -> ReLU -> bf16 -> a block-wide LDS tile [128 rows][256 B] (piece p of row r at slot p ^ (r & 15))
//     -> 16-byte buffer stores, 16 lanes per 256-byte row.  Two barriers per patch.
//   * With one wave per SIMD nothing hides what is not an MFMA, and an in-order wave that waits for the matrix pipe at every MFMA has
//     only the ~32 cycles of the MFMA just issued to hide anything in.  So the next patch's halo loads and LDS writes and the previous
//     patch's stores are cut into pieces of a few instructions, one piece behind one MFMA of the K loop (hook() below, pinned with
//     sched_barriers).  s_memtime stamps (-DRW_TIMING, SYLPH_ABLATE builds) priced every step of that at B = 64:
//         serial phases (load, K loop, epilogue, store)                                       304 us / launch
//         + halo row pitch == pw (mod 16) pixels-of-16-bytes (2-way bank conflicts on every read before)     the reads alone 124 -> 70 us
//         + the same work as three blocks of code inside the K loop                           300 us (a block extends its k-step by its length)
//         + one piece per k-step, buffer addressing (no VALU, no exec masks, no vmcnt(0) before a zero fill)  295 us
//         + one piece per MFMA slot                                                           267 us
//         + loads / stores spread over 40 k-steps (back to back they hit the CU's limit of requests in flight)   262 us
//     = 12 100 cycles per patch, 9 950 of them the K loop at 138 cycles per k-step (4 MFMAs = 128), 1 000 the epilogue.
//
// Numerics: bf16 operands, fp32 accumulation over the taps in tap order, v = acc * scale + shift, ReLU, bf16: the rounding points of
// conv_igemm's halo mode (oracle/bf16.py conv_epilogue).
// Ablation switches (RW_NOSTORE, RW_NOHALO, RW_NOMFMA, RW_NOREAD: measurement aids) exist only in -DSYLPH_ABLATE builds (tools/build_variant.sh)
#ifndef SYLPH_ABLATE
#undef RW_NOSTORE
#undef RW_NOHALO
#undef RW_NOMFMA
#undef RW_NOREAD
#undef RW_TIMING
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
constexpr int RW_NPC = 12;  // 16-byte halo pieces per thread = rows of pixels of a halo (ph + 2 <= 12; its pw + 2 <= 16 pixels are the thread's tid >> 4)
constexpr unsigned RW_OOB = 0xffffff00u;  // a byte offset no tensor reaches (+ < 256): buffer loads return zeros there, buffer stores are dropped

// LDS pitch of a halo ROW OF PIXELS (pw + 2 pixels of RW_TP bytes): a ds_read_b128 serves its 64 lanes in four groups of 16, lanes
// {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of each half, one 16-byte bank group (address / 16 mod 16) per lane and cycle.  A lane reads
// patch position m = (m / pw, m % pw): with pitch / 16 == pw (mod 16) the bank group of position m is m + const (mod 16), distinct over
// every lane group; the natural pitch (pw + 2) * RW_TP gives 2-way conflicts on every read (measured: the reads alone took as long
// as the MFMAs).
__host__ __device__ inline int rw_row_pitch(int pw) {
  const int k0 = (pw + 2) * (RW_TP / 16);
  return (k0 + ((pw - k0) & 15)) * 16;
}
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
  // x and y through buffer descriptors (raw, 2 GiB): an offset past num_records reads as zeros / is not written -- the conv's padding
  // and the non-pixels of ragged patches need neither a branch nor an exec mask nor a VALU instruction
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y, 0, 0x80000000u, 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;

  // ---- this wave's weights -> registers: output channels 32 wave + l31; k-step k = tap * 8 + ks covers input channels 16 ks + 8 lh .. ----
  // taps 0..7 in the 256 AGPRs, tap 8 in 32 VGPRs, for the whole launch
  bf16x8 Wa[64], Wv[8];
  const T* wp = a.w2 + ((size_t)(wave * 32 + l31) * 9 * RW_CH + lh * 8);  // [Cout][3][3][Cin]: tap * 128 + ks * 16 == k * 16
#pragma unroll
  for (int k = 0; k < 64; ++k) Wa[k] = *reinterpret_cast<const bf16x8*>(wp + k * 16);
#pragma unroll
  for (int k = 0; k < 8; ++k) Wv[k] = *reinterpret_cast<const bf16x8*>(wp + (64 + k) * 16);
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
  // The halo of a patch through registers: thread (hx = tid >> 4, chunk c = tid & 15) carries the 16-byte piece c of pixel hx of EVERY halo
  // row hy = j (piece j).  Per patch one VGPR (halo_cols: byte offset of the pixel column, RW_OOB outside the image or past the halo);
  // per piece everything else is scalar: the row's byte offset goes into the load's soffset, a row outside the image gets a
  // descriptor without records.
  u32x4 hreg[RW_NPC];
  unsigned hvoff = RW_OOB, hlds = 0;
  auto halo_cols = [&](int buf, const i32x8 d) {
    const int W = d[2], ox0 = d[3] & 0xffff, HW2 = d[5] + 2;
    const int hx = tid >> 4, c = tid & 15, ix = ox0 - 1 + hx;
    const bool in = hx < HW2 && (unsigned)ix < (unsigned)W;
    hvoff = in ? (unsigned)(ix * (RW_CH * 2) + c * 16) : RW_OOB;
    hlds = buf * RW_HB + (hx < HW2 ? hx * RW_TP + c * 16 : 256);  // idle columns write the pad bytes of pixel 0
  };
  auto halo_load_piece = [&](int j, const i32x8 d) {
    const int row0 = d[0], H = d[1], W = d[2], oy0 = d[3] >> 16, iy = oy0 - 1 + j;
    const bool in = j < d[4] + 2 && (unsigned)iy < (unsigned)H;
    u32x4* hr = hreg;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x), 0, in ? 0x80000000u : 0u, 0x00020000);  // no records: zeros
    hr[j] = __builtin_amdgcn_raw_buffer_load_b128(r, hvoff, in ? (row0 + iy * W) * (RW_CH * 2) : 0, 0);
  };
  auto halo_write_piece = [&](int j, const i32x8 d) {  // rows past the halo hold zeros and land where only pad positions read
    const u32x4* hr = hreg;
    *reinterpret_cast<u32x4*>(smem + hlds + j * rw_row_pitch(d[5])) = hr[j];
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
  halo_cols(0, td);
#pragma unroll
  for (int j = 0; j < RW_NPC; ++j) halo_load_piece(j, td);
#pragma unroll
  for (int j = 0; j < RW_NPC; ++j) halo_write_piece(j, td);
  if (tid < 128) *reinterpret_cast<unsigned*>(smem + RW_TAB + tid * 4) = RW_OOB;  // no previous patch: the first K loop stores nothing

  // Everything that is not the K loop runs INSIDE it (one wave per SIMD: nothing else would hide it), see hook() below.
  const unsigned stg_ad = lds0 + RW_STG + tid * 16, tab_ad = lds0 + RW_TAB + (tid >> 4) * 4;  // piece j: + 4096 j / + 64 j
  const unsigned st_sw = ((tid & 15) ^ ((tid >> 4) & 15)) << 4;                               // byte offset of this thread's piece in its row
  auto store_prev_all = [&]() {  // the plain version: after the last patch
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + RW_STG + tid * 16 + j * 4096);
      const unsigned yo = *reinterpret_cast<const unsigned*>(smem + RW_TAB + (tid >> 4) * 4 + j * 64);
#ifdef RW_NOSTORE
      if (yo == 0xfffffff0u)
#endif
      __builtin_amdgcn_raw_buffer_store_b128(v, yr, yo + st_sw, 0, 0);
    }
  };

#ifdef RW_TIMING
  unsigned long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long rt0 = 0, rt1 = 0;
#define RW_STAMP(i) do { if (it == 10) ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define RW_STAMP(i) do { } while (0)
#endif
  for (int it = 0; t < a.n_tiles; ++it) {
#ifdef RW_TIMING
    if (it == 10) rt0 = wall_clock64();
    if (it == 11) rt1 = wall_clock64();
    if (it == 11) ts[8] = __builtin_readcyclecounter();
#endif
    RW_STAMP(0);
    const int row0 = td[0], IH = td[1], IW = td[2], oy0 = td[3] >> 16, ox0 = td[3] & 0xffff;
    const int PW = td[5], NPOS = td[4] * PW, PY = rw_row_pitch(PW);
    const unsigned inv_pw = (unsigned)td[6];
    int t_next = a.n_tiles;
    i32x8 td_next = td;
    RW_BAR();  // this patch's halo is in buffer it & 1, the previous patch's staging tile and table are complete
    RW_STAMP(1);

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
        hrow[i] = lds0 + (it & 1) * RW_HB + my * PY + (m - my * PW) * RW_TP + 16 * lh;
      }
      constexpr int D = 3;
      bf16x8 af[D][4];
      u32x4 sv;
      unsigned syo;
      auto rd = [&](auto kc) {  // the four fragments of k-step k = tap * 8 + ks
        constexpr int k = decltype(kc)::value, tap = k >> 3, ks = k & 7, kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned ad = hrow[i] + kh * PY + kw * RW_TP;
          bf16x8& dst = af[k % D][i];
#ifdef RW_NOREAD
          asm volatile("" : "=v"(dst) : "v"(ad));
#else
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(ks * 32));
#endif
        }
      };
      // hook(k, i): the piece of non-K-loop work issued right after MFMA i of k-step k.  A wave issues in order and an MFMA waits for
      // the pipe, so only the ~32 cycles of the MFMA just issued hide anything: pieces are a handful of instructions each, pinned in place
      // by sched_barriers (measured with the same work as one block per k-step: every instruction of it extended the k-step).
      //   k 0..1    next tile's descriptor, its per-thread column offset / LDS address
      //   k 2 + 3 j halo row j of the next patch: one buffer load (twelve, spread out: back to back they ran into the CU's limit of
      //             requests in flight and every one of them held the wave for ~70 cycles)
      //   k 3 + 5 j / 4 + 5 j  the previous patch's staging tile: piece j read (inline asm, in the ring's order) / stored, j < 8
      //   k 46..57  halo row j = k - 46 from its registers to the other LDS buffer
      int l_soff = 0, l_step = 0;
      unsigned l_rows = 0;
      auto hook = [&](auto kc, auto ic) {
        constexpr int k = decltype(kc)::value, i = decltype(ic)::value;
        u32x4& svr = sv;
        unsigned& syor = syo;
        if constexpr (k == 0 && i == 0) t_next = tile_of(it + 1);
        if constexpr (k == 0 && i == 1) {  // scalar load, waited for one k-step later (a C++ load becomes a VMEM load + readfirstlanes + vmcnt(0) here)
          const BkTile* p = a.bk + (t_next < a.n_tiles ? t_next : t);
          asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(td_next) : "s"(p));
        }
        if constexpr (k == 1 && i == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(td_next));
#ifndef RW_NOHALO
        if constexpr (k == 1 && i == 1) halo_cols((it + 1) & 1, td_next);
        if constexpr (k == 1 && i == 2) {  // bit j: halo row j of the next patch lies inside the image (none when there is no next patch)
          const int oy0n = td_next[3] >> 16, lo = max(0, 1 - oy0n), hi = min(td_next[4] + 2, td_next[1] - oy0n + 1);
          l_rows = (t_next < a.n_tiles && hi > lo) ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
        }
        if constexpr (k == 1 && i == 3) {  // byte offset of halo row 0, of one image row
          l_step = td_next[2] * (RW_CH * 2);
          l_soff = (td_next[0] + ((td_next[3] >> 16) - 1) * td_next[2]) * (RW_CH * 2);
        }
        if constexpr (k >= 2 && k < 2 + 3 * RW_NPC && (k - 2) % 3 == 0 && i == 1) {
          constexpr int j = (k - 2) / 3;
          u32x4* hr = hreg;
          const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x), 0, (l_rows >> j) & 1u ? 0x80000000u : 0u, 0x00020000);
          hr[j] = __builtin_amdgcn_raw_buffer_load_b128(r, hvoff, l_soff + j * l_step, 0);
        }
        if constexpr (k >= 46 && k < 46 + RW_NPC && i == 0) {  // the other buffer: last read in the previous patch's K loop
          const u32x4* hr = hreg;
          *reinterpret_cast<u32x4*>(smem + hlds + (k - 46) * PY) = hr[k - 46];  // (every patch of a launch has the same ph x pw)
        }
#endif
        if constexpr (k >= 4 && k <= 39 && (k - 4) % 5 == 0 && i == 0) {
          asm volatile("" : "+v"(svr), "+v"(syor));
#ifdef RW_NOSTORE
          if (syor == 0xfffffff0u)
#endif
          __builtin_amdgcn_raw_buffer_store_b128(svr, yr, syor + st_sw, 0, 0);
        }
      };
      auto step = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int sj = (k - 3) / 5;  // store piece read (k == 3 + 5 sj) / stored (k == 4 + 5 sj) at this k-step
        u32x4& svr = sv;
        unsigned& syor = syo;
        if constexpr (k >= 3 && k <= 38 && (k - 3) % 5 == 0) {  // older than this step's fragment reads: complete at the NEXT step's wait
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(svr) : "v"(stg_ad), "n"(sj * 4096));
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(syor) : "v"(tab_ad), "n"(sj * 64));
        }
        if constexpr (k + D - 1 < 72) rd(std::integral_constant<int, k + D - 1>{});
        constexpr int ahead = (k + D - 1 < 72 ? D - 1 : 71 - k) * 4;  // fragment reads issued after those of k-step k
        bf16x8* f = af[k % D];
        const bf16x8 *wa = Wa, *wv = Wv;  // (asm operands naming an enclosing local directly do not capture it in a generic lambda)
        f32x16* ac = acc;
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(ahead));
        static_for<0, 4>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
#ifdef RW_NOMFMA
          if constexpr (k == 0 || k == 71)
#endif
          {
            if constexpr (k == 0) RW_MFMA_A0(ac[i], wa[0], f[i]);
            else if constexpr (k < 64) RW_MFMA_A(ac[i], wa[k], f[i]);
            else RW_MFMA_V(ac[i], wv[k - 64], f[i]);
          }
          __builtin_amdgcn_sched_barrier(0);
          hook(kc, ic);
          __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (k == 16) RW_STAMP(2);
        if constexpr (k == 28) RW_STAMP(3);
        if constexpr (k == 57) RW_STAMP(4);
      };
      rd(std::integral_constant<int, 0>{});
      rd(std::integral_constant<int, 1>{});
      static_for<0, 72>(step);
      // inline-asm MFMAs are invisible to the hazard recogniser: the wait states it would insert before the first VALU read of an accumulator
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])::"memory");
    }
    RW_STAMP(5);
    RW_BAR();  // every wave has stored the previous patch's staging tile (and is done reading this patch's halo)
    RW_STAMP(6);

    // ===== epilogue: FrozenBN + ReLU -> bf16 -> the staging tile (row m, piece 4 wave + g, half lh); y byte offsets of the positions ===
    if (tid < 128) {  // RW_OOB: no such pixel
      const int m = tid;
      const int my = (int)(((unsigned)m * inv_pw) >> 16), mx = m - my * PW;
      const bool pv = m < NPOS && oy0 + my < IH && ox0 + mx < IW;
      *reinterpret_cast<unsigned*>(smem + RW_TAB + m * 4) = pv ? (unsigned)(row0 + (oy0 + my) * IW + ox0 + mx) * (unsigned)(RW_CH * 2) : RW_OOB;
    }
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
    RW_STAMP(7);
    t = t_next;
    td = td_next;
  }
  RW_BAR();
  store_prev_all();
#ifdef RW_TIMING
  if (blockIdx.x == 8 && lane == 0)
    printf("wave %d: top->barA %llu  ->k16 %llu  ->k28 %llu  ->k57 %llu  ->K end %llu  ->barB %llu  ->epi %llu  | whole iteration %llu cycles = %llu ns\n", wave,
           ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5], ts[7] - ts[6], ts[8] - ts[0], (rt1 - rt0) * 10);
#endif
}

bool conv_rw3_patch_ok(int ph, int pw) {
  const int PY = rw_row_pitch(pw);
  // every halo piece has a register, and the fragment reads of the pad positions (m up to 127, bottom-right tap) stay inside a buffer
  return ph * pw <= 128 && ph + 2 <= RW_NPC && pw + 2 <= 16 && (127 / pw + 2) * PY + (127 % pw + 2) * RW_TP + 256 <= RW_HB &&
         (RW_NPC - 1) * PY + (pw + 2) * RW_TP <= RW_HB;
}

int launch_conv_rw3(const BottleneckArgs& a, hipStream_t s) {
  static PerDeviceOnce once;
  const int dev = current_device(), ncu = device_cu_count(dev);
  if (!once.run(dev, [] { return hipFuncSetAttribute((const void*)conv_rw3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RW_LDS) == hipSuccess; })) return -7;
  const int want = (a.n_tiles + 7) & ~7;
  const int grid = want < ncu ? want : (ncu & ~7);
  hipLaunchKernelGGL(conv_rw3_kernel, dim3(grid), dim3(256), RW_LDS, s, a);
  return (int)hipGetLastError();
}

}  // namespace sylph
