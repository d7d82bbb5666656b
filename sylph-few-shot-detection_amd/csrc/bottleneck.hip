// Fused ResNet identity bottleneck for the HBM-bound stage res2 (C = 256, mid = 64), bf16, WEIGHTS IN REGISTERS:
//
//     y = relu(x + bn3(conv1x1_{64->256}( relu(bn2(conv3x3_{64->64}( relu(bn1(conv1x1_{256->64}(x))) ))) )))
//
// (detectron2 BottleneckBlock with FrozenBN, stride 1, no projection: blocks 1.. of a stage; call site
// sylph/modeling/meta_arch/meta_one_stage_detector.py:75,181,273.)  Unfused, the block moves 2 048 B per position through
// HBM (x is read twice -- conv1 input and residual --, the two 64-channel intermediates are written and read back) and its
// three launches sit at the 5.3-5.5 TB/s ceiling: 1.83 ms at B = 64.  Fused: x once, y once = 1 024 B per position.
//
// Two earlier fused designs (git history: "experiment: fused res2 identity bottleneck kernels") streamed the 136 KB of
// weights through LDS for every 128-position tile: 17 small barrier-separated stages per tile made them sync / latency
// bound (1.73 ms with two 4-wave blocks per CU, 2.27 ms persistent with one loader wave).  This kernel removes the weight
// stream altogether:
//
//   * ONE persistent 256-thread block per CU, one wave per SIMD, so every lane may use up to 512 VGPRs.  Each wave loads, once,
//     the MFMA weight fragments of the output channels it owns: W1 (its 32 of 64 mid channels, K 256: 64 VGPRs), W2 (the same
//     32 channels, 9 taps x K 64: 144 VGPRs), W3 (its 64 of 256 output channels, K 64: 32 VGPRs).
//   * LDS holds only activations: the x halo of the tile [192 rows][256 ch] (96 KiB, filled by global_load_lds; chunk swizzle
//     c ^ (row & 31)), t1 / t2 (24 KiB, bf16) and the FrozenBN tables.  Four barriers per tile.
//   * Right after conv1 has consumed the halo (and every lane has copied the 128 residual values of its conv3 outputs from it
//     into registers), all waves issue the NEXT tile's halo: 96 KiB per CU in flight during conv2 / conv3 -- enough to cover
//     the HBM latency at the CU's bandwidth share, which the two-block design could not.
//   * conv3 epilogue in registers (D^T MFMA layout: a lane owns 4 consecutive channels of a position): bn3 + residual + ReLU ->
//     bf16 -> a wave-private 32 x 64-channel LDS tile -> read back row-major -> 16-byte stores, 8 lanes per 128-byte line (round 5).
//     Rounds 2-4 stored the fragments directly (8 bytes per lane: every instruction is 32 write requests of 16 bytes, and a CU
//     retires about one request per cycle): 1.29 -> 1.09 ms per launch at B = 64, profiles/r5_bottleneck_ablation.txt.
//   * Round-5 experiment (git history: "experiment: x halo as a ring of 64-channel chunks"): conv1 walking K in four 24-KiB chunks
//     through five ring slots, so that the next patch's halo is in flight during conv1 too.  Parity-clean; it hides 86 us more of
//     the loads per launch and costs 140 us of compute (seven barriers per patch instead of four, four pipeline fills in conv1).
//
// P1 recomputes conv1 on the (ph+2) x (pw+2) halo (+40 % of its flops); halo positions outside the image are forced to 0
// (they are conv2's zero padding).  P2 reads shifted rows of the t1 halo exactly like conv_igemm.hip's HALO mode.
#include "common.h"

#include <stdlib.h>

namespace sylph {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

namespace {
constexpr int MID = 64, C = 256;
constexpr int XROWS = 192;
constexpr int XB_BYTES = XROWS * 512;          // 98 304: x halo
constexpr int TP = 144;                        // t1 / t2 row pitch: 128 B of channels + 16 B pad (bank-conflict-free, no swizzle)
constexpr int T1_OFF = XB_BYTES;               // 27 648: t1 halo [192][64 ch]; t2 [128][64 ch] aliases it
constexpr int BN_OFF = T1_OFF + XROWS * TP;    // s1 b1 s2 b2 (64 each) s3 b3 (256 each), fp32
constexpr int LDS_BYTES = BN_OFF + (4 * MID + 2 * C) * 4;  // 129 024

// MFMA with the weight fragment read straight from an AGPR and the accumulator in arch VGPRs.  Through the builtin hipcc keeps
// weights and accumulators in AGPRs only as spill space and pays a v_accvgpr_read per use (~450 per tile, all on the one wave
// that also has to issue the MFMAs).  Inline asm is invisible to the hazard recogniser: BK_MFMA_DRAIN* before the first VALU
// read of an accumulator supplies the wait states (16-pass MFMA: 18) it would have inserted.
#define BK_MFMA(acc, w, av) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(av))
// first k-step of a chain: srcC = 0 (a VALU zero-fill followed by an MFMA reading it is a 2-wait-state hazard nobody would pad)
#define BK_MFMA0(acc, w, av) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(av))
// (the accumulators are operands of the drain: their VALU reads must not be scheduled above it)
#define BK_MFMA_DRAIN2(a0, a1) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a0), "+v"(a1)::"memory")
#define BK_MFMA_DRAIN3(a0, a1, a2) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a0), "+v"(a1), "+v"(a2)::"memory")

// Ablation switches (measurement aids: BK_ONLY = run only phase 1 / 2 / 3 of every tile, BK_NOBAR, BK_NOP1, BK_NOP2, BK_NOX, BK_NOE3,
// BK_NOSTORE) exist ONLY in builds made with -DSYLPH_ABLATE (tools/build_variant.sh -> lib/variants/): the product library is
// compiled without it and every switch is forced off here.
#ifndef SYLPH_ABLATE
#undef BK_ONLY
#undef BK_NOBAR
#undef BK_NOP1
#undef BK_NOP2
#undef BK_NOX
#undef BK_NOE3
#undef BK_NOSTORE
#undef BK_TIMING
#endif
// BK_TIMING (SYLPH_ABLATE builds): s_memtime stamps of the phases of patch 10, printed by block 8 (tools/bench_bottleneck.py shows them)
#ifdef BK_TIMING
#define BK_STAMP(i) do { if (it == 10) ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define BK_STAMP(i) do { } while (0)
#endif
#ifndef BK_ONLY
#define BK_ONLY 0
#endif
#ifdef BK_NOBAR
#define BK_BAR() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define BK_BAR()                                       \
  do {                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                      \
    asm volatile("" ::: "memory");                     \
  } while (0)
#endif
}  // namespace

// NR1 / NR2 / NR3: 32-row MFMA tiles a wave processes in P1 (halo rows) / P2 / P3 (patch positions); XR = NR1 * 64 halo rows.
//   <3, 2, 4, false>: patches of <= 128 positions, ONE 96-KiB halo buffer, the next halo issued after conv1 (round 2);
//   <2, 1, 2, true>:  (round-3 experiment, no longer instantiated: patches of <= 64 positions, TWO 64-KiB halo buffers, the next
//                     patch's halo issued a whole tile ahead.  The halo round trip was hidden, but 2.2 x as many tiles paid the
//                     per-tile fixed costs: 1.56 ms vs 1.31 ms per launch at B = 64.  Its LDS no longer fits beside the store staging.)
template <int NR1, int NR2, int NR3, bool DB>
__global__ __launch_bounds__(256, 1) void bottleneck64_kernel(const BottleneckArgs a) {
  constexpr int XR = NR1 * 64;                     // halo rows of a buffer
  constexpr int XBB = XR * 512;                    // one halo buffer
  constexpr int T1O = (DB ? 2 : 1) * XBB;          // t1 halo [XR][64 ch] (pitch TP); t2 aliases it
  constexpr int BNO = T1O + XR * TP;
  constexpr int STG = BNO + (4 * MID + 2 * C) * 4;  // store staging: 4 KiB per wave (32 rows x 128 B, piece p of row r at slot p ^ (r & 7))
  constexpr int TAB = STG + 4 * 4096;               // y byte offset of every patch position, [row tile][row & 7][(row >> 3) & 3] (512 B)
  constexpr int NST = NR3 * 4;                     // stores per lane and tile (16 bytes each: whole 128-byte lines per 8 lanes)
  typedef bf16_t T;
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));  // ext_vector LDS accesses: hipcc adds no vmcnt(0) for them beside LDS-DMA
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  T* __restrict__ y = reinterpret_cast<T*>(a.y);
  char* const trash = reinterpret_cast<char*>(a.trash) + ((size_t)blockIdx.x * 256 + tid) * 128;  // 128 B per thread: the 8 stores of a row tile
  char* xb = smem;  // the current tile's halo buffer
  char* const t1 = smem + T1O;
  float* const bn = reinterpret_cast<float*>(smem + BNO);
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
  const unsigned stg = smem_lds + STG + wave * 4096;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  // ---- weights -> registers (plain loads, before any LDS-DMA exists) -------------------------------------------------------
  const int ct1 = wave >> 1;  // P1 / P2: this wave's 32 mid channels
  bf16x8 W1f[16], W2f[36], W3f[2][4];
  {
    const T* w1p = a.w1 + ((ct1 * 32 + l31) * C + lh * 8);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) W1f[ks] = *reinterpret_cast<const bf16x8*>(w1p + ks * 16);
    const T* w2p = a.w2 + ((ct1 * 32 + l31) * 9 * MID + lh * 8);
#pragma unroll
    for (int k = 0; k < 36; ++k) W2f[k] = *reinterpret_cast<const bf16x8*>(w2p + k * 16);  // tap * 64 + ks * 16 == k * 16
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const T* w3p = a.w3 + (((2 * wave + j) * 32 + l31) * MID + lh * 8);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) W3f[j][ks] = *reinterpret_cast<const bf16x8*>(w3p + ks * 16);
    }
  }
  for (int i = tid; i < 4 * MID + 2 * C; i += 256) {
    const float* src = i < MID ? a.s1 + i : i < 2 * MID ? a.b1 + (i - MID) : i < 3 * MID ? a.s2 + (i - 2 * MID)
                     : i < 4 * MID ? a.b2 + (i - 3 * MID) : i < 4 * MID + C ? a.s3 + (i - 4 * MID) : a.b3 + (i - 4 * MID - C);
    bn[i] = *src;
  }
  const float *s1 = bn, *b1 = bn + MID, *s2 = bn + 2 * MID, *b2 = bn + 3 * MID, *s3 = bn + 4 * MID, *b3 = bn + 4 * MID + C;
  (void)b1; (void)b2; (void)s3; (void)b3;  // read through s1 / s2 offsets or kept in registers (below)
  // conv3's FrozenBN constants of this lane's 64 output channels stay in registers (P3 is VALU bound: 64 fewer LDS reads per tile)
  f32x4 s3r[2][4], b3r[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      s3r[j][gq] = *reinterpret_cast<const f32x4*>(a.s3 + 64 * wave + 32 * j + 8 * gq + 4 * lh);
      b3r[j][gq] = *reinterpret_cast<const f32x4*>(a.b3 + 64 * wave + 32 * j + 8 * gq + 4 * lh);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // persistent tile walk: blocks of one XCD (blockIdx & 7) take neighbouring patches at the same time
  const int G = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, gx = (G + 7) >> 3;
  const int chunk = (a.n_tiles + 7) >> 3;
  auto tile_of = [&](int it) { const int q = it * gx + jb; return __builtin_amdgcn_readfirstlane(q < chunk ? xcd * chunk + q : a.n_tiles); };
  // tile descriptor through the SCALAR cache (a vector load would put a vmcnt(0) into the stream; hipcc will not use s_load
  // for memory it cannot prove read-only)
  auto load_tile = [&](int t) {
    i32x8 v;
    const BkTile* p = a.bk + t;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
    return v;
  };
  // x halo -> LDS: rounds of 8 rows x 512 B; slot s of row r holds 16-byte chunk s ^ (r & 31).  Halo rows outside the image
  // are loaded from a clamped (valid) address instead of a zero page: whatever lands there is never used -- P1 masks t1 to 0
  // for those rows (they are conv2's zero padding) and the residual is only read at stored positions.
  const int xr = tid >> 5, xs = tid & 31;
  auto issue_x = [&](const i32x8 d, char* dstb) {
    const int row0 = d[0], H = d[1], W = d[2], oy0 = d[3] >> 16, ox0 = d[3] & 0xffff, HW2 = d[5] + 2, HR = (d[4] + 2) * HW2;
    const int nr = (HR + 7) >> 3;  // rows >= HR are never read by P2
    int hy = 0, hx = xr;
    for (int r = 0; r < nr; ++r) {
      while (hx >= HW2) { hx -= HW2; ++hy; }  // (a single step for patches at least 6 wide; narrow maps may wrap twice)
      const int h = r * 8 + xr;
      const int iy = min(max(oy0 - 1 + hy, 0), H - 1), ix = min(max(ox0 - 1 + hx, 0), W - 1);
      // per-image 64-bit base (wave-uniform, from the tile descriptor) + a 32-bit offset INSIDE the image: no limit on the batch (rounds
      // 2-5 offset the whole tensor with 32 bits: the fused kernels fell away above 124 images of 800 x 1333)
      const unsigned off = ((unsigned)(iy * W + ix) << 9) + (unsigned)((xs ^ (h & 31)) << 4);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(reinterpret_cast<const char*>(x) + ((size_t)(unsigned)row0 << 9) + off), (lds_ptr_t)(dstb + r * 4096 + wave * 1024), 16, 0, 0);
      hx += 8;
    }
  };

  int t = tile_of(0);
  i32x8 td = load_tile(t < a.n_tiles ? t : 0);
  if (t < a.n_tiles) issue_x(td, xb);
  const int rb1 = (wave & 1) * NR1, rb2 = (wave & 1) * NR2;  // first row tile of this wave in P1 (3 tiles) / P2 (2 tiles)
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  // bf16 pair: ReLU as a packed signed-16-bit max with 0 (sign bit set <=> negative), then AND with a keep mask
  auto relu_pk = [](unsigned u, unsigned keep) {
    const s16x2 z = {0, 0};
    const s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, u), z);
    return __builtin_bit_cast(unsigned, r) & keep;
  };
  auto pack2 = [](float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v;
    v[0] = (bf16_t)lo;
    v[1] = (bf16_t)hi;
    return __builtin_bit_cast(unsigned, v);
  };

#ifdef BK_TIMING
  unsigned long long ts[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int it = 0; t < a.n_tiles; ++it) {
#ifdef BK_TIMING
    if (it == 11) ts[11] = __builtin_readcyclecounter();
#endif
    BK_STAMP(0);
    const int row0 = td[0], IH = td[1], IW = td[2], oy0 = td[3] >> 16, ox0 = td[3] & 0xffff;
    char* const yimg = reinterpret_cast<char*>(y) + (size_t)(unsigned)row0 * (size_t)(C * 2);  // this image's first output row (wave-uniform)
    const int PW = td[5], HW2 = PW + 2, HR = (td[4] + 2) * HW2, NPOS = td[4] * PW;
    const unsigned inv_pw = (unsigned)td[6], inv_hw2 = (unsigned)td[7];
    const int t_next = tile_of(it + 1);
    const i32x8 td_next = load_tile(t_next < a.n_tiles ? t_next : 0);

    // this tile's halo has landed.  The counter retires in issue order and every lane issues exactly 32 stores per tile AFTER
    // the halo loads of the next one (lanes without a valid position write to a private trash slot instead of being masked
    // off), so vmcnt(32) leaves the previous tile's stores -- and their ~2 us of write acknowledgement -- in flight
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    BK_BAR();  // ... for every wave; t1 / t2 of the previous tile are free
    BK_STAMP(1);
    // y byte offset of patch position m (0xffffffff: no such pixel -> the store goes to the trash slot), for the transposed stores of
    // P3: entry [m >> 5][m & 7][(m >> 3) & 3], so that a lane reads the four rows it stores of a row tile with one ds_read_b128.
    // (The previous patch's P3 is over for every wave -- barrier above; the next reader is this patch's P3, three barriers on.)
    if (tid < NR3 * 32) {
      const int m = tid;
      const int my = (int)(((unsigned)m * inv_pw) >> 16), mx = m - my * PW;
      const bool pv = m < NPOS && oy0 + my < IH && ox0 + mx < IW;
      const unsigned off = pv ? (unsigned)((oy0 + my) * IW + ox0 + mx) * (unsigned)(C * 2) : 0xffffffffu;  // inside the image
      *reinterpret_cast<unsigned*>(smem + TAB + ((m >> 5) * 32 + (m & 7) * 4 + ((m >> 3) & 3)) * 4) = off;
    }
    if (DB) {
      // the other buffer was last read in P1 / the residual copy of the previous tile: free.  Its refill is in flight during this
      // whole tile; the vmcnt wait above (next iteration) leaves exactly this tile's NST stores, issued after it, outstanding.
      char* const other = xb == smem ? smem + XBB : smem;
      if (t_next < a.n_tiles) issue_x(td_next, other);
    }

    // ===== P1: t1 = relu(bn1(x_halo . W1^T)), row tiles rb1 .. rb1+2, channels ct1 ===========================================
    // (lz*: zero, opaque to the compiler and re-made per tile and phase: LDS addresses would otherwise be hoisted out of the tile
    //  loop as ~150 loop-invariant VGPRs, leaving no registers to pipeline the fragment reads)
    if (BK_ONLY == 0 || BK_ONLY == 1) {
      int lz1;
      asm volatile("v_mov_b32 %0, 0" : "=v"(lz1));
      const int l31a = l31 + lz1;
      f32x16 acc1[NR1];
      // the three row tiles are 32 rows = 16 KiB apart and share the swizzle key (row & 31 == l31): one address per k-step,
      // the tiles are immediate offsets
      const char* abase = xb + (rb1 * 32 + l31a) * 512;
      const int akey = (l31a ^ lh) << 4;
      constexpr int D1 = 3;  // fragment ring: reads run D1 - 1 k-steps ahead of the MFMAs (one wave per SIMD: nothing else hides LDS latency)
      bf16x8 af[D1][NR1];
#pragma unroll
      for (int ks = 0; ks < D1 - 1; ++ks) {
        const char* ap = abase + ((ks * 32) ^ akey);
#pragma unroll
        for (int i = 0; i < NR1; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(ap + i * 16384);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, NR1 * (D1 - 1), 0);
#ifdef BK_NOP1
      constexpr int K1 = 4;
#else
      constexpr int K1 = 16;
#endif
#pragma unroll
      for (int ks = 0; ks < K1; ++ks) {
        if (ks + D1 - 1 < 16) {
          const char* ap = abase + (((ks + D1 - 1) * 32) ^ akey);
#pragma unroll
          for (int i = 0; i < NR1; ++i) af[(ks + D1 - 1) % D1][i] = *reinterpret_cast<const bf16x8*>(ap + i * 16384);
        }
#pragma unroll
        for (int i = 0; i < NR1; ++i) {
          if (ks == 0) BK_MFMA0(acc1[i], W1f[ks], af[ks % D1][i]);
          else BK_MFMA(acc1[i], W1f[ks], af[ks % D1][i]);
        }
      }
      if constexpr (NR1 == 3) BK_MFMA_DRAIN3(acc1[0], acc1[1], acc1[2]);
      else BK_MFMA_DRAIN2(acc1[0], acc1[1]);
      BK_STAMP(2);
      const float* sp = s1 + ct1 * 32 + 4 * lh;  // b1 = s1 + 64
#pragma unroll
      for (int i = 0; i < NR1; ++i) {
        const int h = (rb1 + i) * 32 + l31a;
        const int hy = (int)(((unsigned)h * inv_hw2) >> 16), hx = h - hy * HW2;
        const bool in1 = h < HR && (unsigned)(oy0 - 1 + hy) < (unsigned)IH && (unsigned)(ox0 - 1 + hx) < (unsigned)IW;
        const unsigned keep = in1 ? 0xffffffffu : 0u;
        char* wp = t1 + h * TP + ct1 * 64 + 8 * lh;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(sp + 8 * gq), bv = *reinterpret_cast<const f32x4*>(sp + MID + 8 * gq);
          u32x2 o;
          o[0] = relu_pk(pack2(acc1[i][4 * gq] * sv[0] + bv[0], acc1[i][4 * gq + 1] * sv[1] + bv[1]), keep);
          o[1] = relu_pk(pack2(acc1[i][4 * gq + 2] * sv[2] + bv[2], acc1[i][4 * gq + 3] * sv[3] + bv[3]), keep);
          *reinterpret_cast<u32x2*>(wp + gq * 16) = o;
        }
      }
    }
    BK_STAMP(3);
    // residual values of this lane's conv3 outputs (positions rt * 32 + l31, channels 64 wave + 32 j + 8 gq + 4 lh ..): halo -> registers
    u32x2 res[NR3 * 8];
    int lzr;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lzr));
#pragma unroll
    for (int rt = 0; rt < NR3; ++rt) {
      const int m = rt * 32 + l31 + lzr;
      const int my = (int)(((unsigned)m * inv_pw) >> 16), mx = m - my * PW;
      const int hc = (my + 1) * HW2 + mx + 1;
      const int rowoff = hc * 512 + ((hc & 31) << 4) + 8 * lh;  // chunk c of this row sits at rowoff ^ (c << 4)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          res[rt * 8 + j * 4 + gq] = *reinterpret_cast<const u32x2*>(xb + (rowoff ^ ((8 * wave + 4 * j + gq) << 4)));
    }
    // (re-defined through asm: hipcc would otherwise tie their first use to vmcnt(0), they come from the LDS-DMA target)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(res[4]), "+v"(res[5]), "+v"(res[6]), "+v"(res[7]),
                   "+v"(res[8]), "+v"(res[9]), "+v"(res[10]), "+v"(res[11]), "+v"(res[12]), "+v"(res[13]), "+v"(res[14]), "+v"(res[15]));
    if constexpr (NR3 == 4)
      asm volatile(""
                   : "+v"(res[16]), "+v"(res[17]), "+v"(res[18]), "+v"(res[19]), "+v"(res[20]), "+v"(res[21]), "+v"(res[22]), "+v"(res[23]),
                     "+v"(res[24]), "+v"(res[25]), "+v"(res[26]), "+v"(res[27]), "+v"(res[28]), "+v"(res[29]), "+v"(res[30]), "+v"(res[31]));
    BK_STAMP(4);
    BK_BAR();  // t1 complete; every wave is done with the x halo
    BK_STAMP(5);
#ifndef BK_NOX
    if (!DB && t_next < a.n_tiles) issue_x(td_next, xb);
#endif
    BK_STAMP(6);

    // ===== P2: t2 = relu(bn2(conv3x3(t1))), row tiles rb2, rb2+1, channels ct1 ===============================================
    // t1 / t2 rows are PADDED to 144 B (they are written by ds_write, not by DMA): conflict-free without a swizzle, so the
    // k-step and tile offsets of every read are instruction immediates (one VALU add per tap and tile instead of per read)
    if (BK_ONLY == 0 || BK_ONLY == 2) {
      int lz2;
      asm volatile("v_mov_b32 %0, 0" : "=v"(lz2));
      const int l31b = l31 + lz2;
      f32x16 acc2[NR2];
      const char* hrow[NR2];
#pragma unroll
      for (int i = 0; i < NR2; ++i) {
        const int m = (rb2 + i) * 32 + l31b;
        const int my = (int)(((unsigned)m * inv_pw) >> 16);
        hrow[i] = t1 + (my * HW2 + (m - my * PW)) * TP + 16 * lh;
      }
      // Fragment ring, D2 - 1 k-steps ahead of the MFMAs.  The reads are inline asm with COUNTED waits: the next patch's halo is
      // in flight (LDS-DMA) during this phase, and while one is pending hipcc treats the LGKM counter as out of order and turns
      // every wait for a compiler-visible ds_read into lgkmcnt(0) -- i.e. it would wait for the reads just issued as well.
      constexpr int D2 = 4;
      bf16x8 af[D2][NR2];
      auto p2_read = [&](int k) {  // k-step k = tap * 4 + ks: both row tiles
        const int tap = k >> 2, ks = k & 3, kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
        for (int i = 0; i < NR2; ++i) {
          const unsigned ad = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(hrow[i] + (kh * HW2 + kw) * TP);
          if (ks == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(af[k % D2][i]) : "v"(ad));
          else if (ks == 1) asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(af[k % D2][i]) : "v"(ad));
          else if (ks == 2) asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(af[k % D2][i]) : "v"(ad));
          else asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(af[k % D2][i]) : "v"(ad));
        }
      };
#pragma unroll
      for (int k = 0; k < D2 - 1; ++k) p2_read(k);
#ifdef BK_NOP2
      constexpr int K2 = 8;
#else
      constexpr int K2 = 36;
#endif
#pragma unroll
      for (int k = 0; k < K2; ++k) {
        if (k + D2 - 1 < 36) p2_read(k + D2 - 1);
        // reads issued after those of k-step k: NR2 per k-step still ahead
        const int ahead = (k + D2 - 1 < 36 ? D2 - 1 : 35 - k) * NR2;
        if constexpr (NR2 == 2) {
          if (ahead == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af[k % D2][0]), "+v"(af[k % D2][1]));
          else if (ahead == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[k % D2][0]), "+v"(af[k % D2][1]));
          else if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[k % D2][0]), "+v"(af[k % D2][1]));
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[k % D2][0]), "+v"(af[k % D2][1]));
        } else {
          if (ahead == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(af[k % D2][0]));
          else if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[k % D2][0]));
          else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(af[k % D2][0]));
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[k % D2][0]));
        }
#pragma unroll
        for (int i = 0; i < NR2; ++i) {
          if (k == 0) BK_MFMA0(acc2[i], W2f[k], af[k % D2][i]);
          else BK_MFMA(acc2[i], W2f[k], af[k % D2][i]);
        }
      }
      if constexpr (NR2 == 2) BK_MFMA_DRAIN2(acc2[0], acc2[1]);
      else asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc2[0])::"memory");
      BK_STAMP(7);
      BK_BAR();  // every wave has finished reading t1: t2 may overwrite it
      BK_STAMP(8);
      const float* sp = s2 + ct1 * 32 + 4 * lh;  // b2 = s2 + 64
#pragma unroll
      for (int i = 0; i < NR2; ++i) {
        char* wp = t1 + ((rb2 + i) * 32 + l31b) * TP + ct1 * 64 + 8 * lh;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(sp + 8 * gq), bv = *reinterpret_cast<const f32x4*>(sp + MID + 8 * gq);
          u32x2 o;
          o[0] = relu_pk(pack2(acc2[i][4 * gq] * sv[0] + bv[0], acc2[i][4 * gq + 1] * sv[1] + bv[1]), 0xffffffffu);
          o[1] = relu_pk(pack2(acc2[i][4 * gq + 2] * sv[2] + bv[2], acc2[i][4 * gq + 3] * sv[3] + bv[3]), 0xffffffffu);
          *reinterpret_cast<u32x2*>(wp + gq * 16) = o;
        }
      }
    }
    BK_STAMP(9);
    BK_BAR();  // t2 complete
    BK_STAMP(10);

    // ===== P3: y = relu(bn3(t2 . W3^T) + x), all four row tiles, channels 64 wave .. 64 wave + 63 ==========================
    // Software pipelined over the row tiles: the 8 MFMAs of tile rt + 1 are issued one per epilogue chunk of tile rt (k-step
    // outer, channel tile inner, so the two dependent chains alternate).  Back to back they would stall the wave's in-order
    // issue for ~8 x 40 cycles per tile and then leave the matrix pipe idle during ~600 cycles of epilogue VALU work.
    if (BK_ONLY == 0 || BK_ONLY == 3) {
      int lz3;
      asm volatile("v_mov_b32 %0, 0" : "=v"(lz3));
      const char* pbase = t1 + (l31 + lz3) * TP + 16 * lh;
      bf16x8 av[2][4];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) av[b][ks] = *reinterpret_cast<const bf16x8*>(pbase + b * 32 * TP + ks * 32);
      f32x16 acc3[2][2];  // [buffer][channel tile]
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (ks == 0) BK_MFMA0(acc3[0][j], W3f[j][0], av[0][0]);
          else BK_MFMA(acc3[0][j], W3f[j][ks], av[0][ks]);
        }
#pragma unroll
      for (int rt = 0; rt < NR3; ++rt) {
        const int cb = rt & 1, nb = cb ^ 1;
        BK_MFMA_DRAIN2(acc3[cb][0], acc3[cb][1]);
        if (rt + 2 < NR3) {  // fragments of tile rt + 2 into the buffer tile rt has just finished with
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) av[cb][ks] = *reinterpret_cast<const bf16x8*>(pbase + (rt + 2) * 32 * TP + ks * 32);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int j = c >> 2, gq = c & 3;
          if (rt < NR3 - 1) {
            const int ks2 = c >> 1, j2 = c & 1;
            if (ks2 == 0) BK_MFMA0(acc3[nb][j2], W3f[j2][0], av[nb][0]);
            else BK_MFMA(acc3[nb][j2], W3f[j2][ks2], av[nb][ks2]);
          }
          const f32x4 sv = s3r[j][gq], bv = b3r[j][gq];
          const u32x2 rv = res[rt * 8 + j * 4 + gq];
          u32x2 o;
#ifdef BK_NOE3
          o[0] = pack2(acc3[cb][j][4 * gq] + sv[0] + bv[0] + __uint_as_float(rv[0]), acc3[cb][j][4 * gq + 1]);
          o[1] = pack2(acc3[cb][j][4 * gq + 2], acc3[cb][j][4 * gq + 3] + __uint_as_float(rv[1]));
#else
          o[0] = relu_pk(pack2(acc3[cb][j][4 * gq] * sv[0] + bv[0] + __uint_as_float(rv[0] << 16),
                               acc3[cb][j][4 * gq + 1] * sv[1] + bv[1] + __uint_as_float(rv[0] & 0xffff0000u)), 0xffffffffu);
          o[1] = relu_pk(pack2(acc3[cb][j][4 * gq + 2] * sv[2] + bv[2] + __uint_as_float(rv[1] << 16),
                               acc3[cb][j][4 * gq + 3] * sv[3] + bv[3] + __uint_as_float(rv[1] & 0xffff0000u)), 0xffffffffu);
#endif
          // into the wave's staging tile: row l31, 16-byte piece p = 4 j + gq (this lane holds half lh of it)
          asm volatile("ds_write_b64 %0, %1" ::"v"(stg + l31 * 128 + (((c ^ (l31 & 7))) << 4) + lh * 8), "v"(o) : "memory");
        }
        // read the 32 x 64-channel tile back row-major -- 8 lanes cover the 128 bytes this wave owns of a row -- and store whole
        // lines: 4 stores of 16 bytes per lane instead of 8 of 8 bytes, 8 write requests per instruction instead of 32 (the
        // 8-byte fragment stores cost ~250 us per launch at B = 64: one 16-byte request per row and instruction)
        u32x4 ov[4], yo;
        asm volatile("ds_read_b128 %0, %1" : "=v"(yo) : "v"(smem_lds + TAB + rt * 128 + (lane >> 3) * 16));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (lane >> 3) + 8 * q;
          asm volatile("ds_read_b128 %0, %1" : "=v"(ov[q]) : "v"(stg + row * 128 + (((lane & 7) ^ (row & 7)) << 4)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ov[0]), "+v"(ov[1]), "+v"(ov[2]), "+v"(ov[3]), "+v"(yo));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          char* dst = yo[q] != 0xffffffffu ? yimg + ((size_t)yo[q] + 128 * wave + (lane & 7) * 16) : trash;
#ifdef BK_NOSTORE
          if (ov[q][0] == 0x12345678u) *reinterpret_cast<u32x4*>(dst) = ov[q];
#else
          *reinterpret_cast<u32x4*>(dst) = ov[q];
#endif
        }
      }
    }
    t = t_next;
    td = td_next;
    if (DB) xb = xb == smem ? smem + XBB : smem;
  }
#ifdef BK_TIMING
  if (blockIdx.x == 8 && lane == 0)
    printf("wave %d: wait+bar %llu | P1 K %llu  epi %llu  res copy %llu  bar %llu | issue_x %llu  P2 K %llu  bar %llu  epi %llu  bar %llu | P3 %llu | total %llu\n", wave,
           ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[6] - ts[5], ts[7] - ts[6], ts[8] - ts[7], ts[9] - ts[8],
           ts[10] - ts[9], ts[11] - ts[10], ts[11] - ts[0]);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// First block of res2 (64 -> 64 -> 64 -> 256, projection shortcut), same design.  What differs from the identity block:
//   * x has 64 channels: a halo is 192 rows x 128 B = 24 KiB, so TWO halo buffers fit and the next patch's halo is issued at
//     the top of a tile, a whole tile ahead (the identity block can only issue it after conv1);
//   * no residual registers: conv3 and the projection are ONE GEMM over K = [t2 | x] (128) against the packed c3sc weights
//     ([256][128], both FrozenBN scales folded in fp32 before the bf16 rounding, shifts summed -- api_weights.hip), the x half of
//     the A operand is read straight from the halo's centre rows;  y = relu(acc + shift).
namespace {
constexpr int PX_BYTES = XROWS * 128;         // one halo buffer
constexpr int PT1_OFF = 2 * PX_BYTES;         // t1 / t2, pitch TP
constexpr int PBN_OFF = PT1_OFF + XROWS * TP; // s1 b1 s2 b2
constexpr int PSTG_OFF = PBN_OFF + 4 * MID * 4;  // store staging: 4 KiB per wave
constexpr int PTAB_OFF = PSTG_OFF + 4 * 4096;   // y byte offset of every patch position (512 B)
constexpr int PLDS_BYTES = PTAB_OFF + 512;
}  // namespace

__global__ __launch_bounds__(256, 1) void bottleneck64p_kernel(const BottleneckArgs a) {
  typedef bf16_t T;
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  T* __restrict__ y = reinterpret_cast<T*>(a.y);
  char* const trash = reinterpret_cast<char*>(a.trash) + ((size_t)blockIdx.x * 256 + tid) * 128;
  char* const t1 = smem + PT1_OFF;
  float* const bn = reinterpret_cast<float*>(smem + PBN_OFF);
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
  const unsigned stg = smem_lds + PSTG_OFF + wave * 4096;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  const int ct1 = wave >> 1;
  bf16x8 W1f[4], W2f[36], W3f[2][8];
  {
    const T* w1p = a.w1 + ((ct1 * 32 + l31) * MID + lh * 8);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) W1f[ks] = *reinterpret_cast<const bf16x8*>(w1p + ks * 16);
    const T* w2p = a.w2 + ((ct1 * 32 + l31) * 9 * MID + lh * 8);
#pragma unroll
    for (int k = 0; k < 36; ++k) W2f[k] = *reinterpret_cast<const bf16x8*>(w2p + k * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const T* w3p = a.w3 + (((2 * wave + j) * 32 + l31) * (2 * MID) + lh * 8);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) W3f[j][ks] = *reinterpret_cast<const bf16x8*>(w3p + ks * 16);
    }
  }
  for (int i = tid; i < 4 * MID; i += 256) {
    const float* src = i < MID ? a.s1 + i : i < 2 * MID ? a.b1 + (i - MID) : i < 3 * MID ? a.s2 + (i - 2 * MID) : a.b2 + (i - 3 * MID);
    bn[i] = *src;
  }
  const float *s1 = bn, *s2 = bn + 2 * MID;
  f32x4 b3r[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) b3r[j][gq] = *reinterpret_cast<const f32x4*>(a.b3 + 64 * wave + 32 * j + 8 * gq + 4 * lh);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const int G = gridDim.x, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, gx = (G + 7) >> 3;
  const int chunk = (a.n_tiles + 7) >> 3;
  auto tile_of = [&](int it) { const int q = it * gx + jb; return __builtin_amdgcn_readfirstlane(q < chunk ? xcd * chunk + q : a.n_tiles); };
  auto load_tile = [&](int t) {
    i32x8 v;
    const BkTile* p = a.bk + t;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
    return v;
  };
  // x halo -> LDS buffer `buf`: 6 rounds of 32 rows x 128 B; slot s of row r holds 16-byte chunk s ^ ((r >> 1) & 7); rows outside
  // the image come from a clamped address (never used: P1 masks t1 there, P3 only stores valid positions)
  const int xr = tid >> 3, xs = tid & 7;
  auto issue_x = [&](const i32x8 d, int buf) {
    const int row0 = d[0], H = d[1], W = d[2], oy0 = d[3] >> 16, ox0 = d[3] & 0xffff, HW2 = d[5] + 2, HR = (d[4] + 2) * HW2;
    const unsigned inv_hw2 = (unsigned)d[7];
    const int nr = (HR + 31) >> 5;
    for (int r = 0; r < nr; ++r) {
      const int h = r * 32 + xr;
      const int hy = (int)(((unsigned)h * inv_hw2) >> 16), hx = h - hy * HW2;
      const int iy = min(max(oy0 - 1 + hy, 0), H - 1), ix = min(max(ox0 - 1 + hx, 0), W - 1);
      const unsigned off = ((unsigned)(iy * W + ix) << 7) + (unsigned)((xs ^ ((h >> 1) & 7)) << 4);  // inside the image; 64-bit image base below
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(reinterpret_cast<const char*>(x) + ((size_t)(unsigned)row0 << 7) + off), (lds_ptr_t)(smem + buf * PX_BYTES + r * 4096 + wave * 1024), 16, 0, 0);
    }
  };
  auto relu_pk = [](unsigned u, unsigned keep) {
    const s16x2 z = {0, 0};
    const s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, u), z);
    return __builtin_bit_cast(unsigned, r) & keep;
  };
  auto pack2 = [](float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v;
    v[0] = (bf16_t)lo;
    v[1] = (bf16_t)hi;
    return __builtin_bit_cast(unsigned, v);
  };

  int t = tile_of(0);
  i32x8 td = load_tile(t < a.n_tiles ? t : 0);
  if (t < a.n_tiles) issue_x(td, 0);
  const int rb1 = (wave & 1) * 3, rb2 = (wave & 1) * 2;

  for (int it = 0; t < a.n_tiles; ++it) {
    const int row0 = td[0], IH = td[1], IW = td[2], oy0 = td[3] >> 16, ox0 = td[3] & 0xffff;
    char* const yimg = reinterpret_cast<char*>(y) + (size_t)(unsigned)row0 * (size_t)(C * 2);  // this image's first output row (wave-uniform)
    const int PW = td[5], HW2 = PW + 2, HR = (td[4] + 2) * HW2, NPOS = td[4] * PW;
    const unsigned inv_pw = (unsigned)td[6], inv_hw2 = (unsigned)td[7];
    const int t_next = tile_of(it + 1);
    const i32x8 td_next = load_tile(t_next < a.n_tiles ? t_next : 0);
    char* const xb = smem + (it & 1) * PX_BYTES;

    // this tile's halo (issued a whole tile ago, before the previous tile's 32 stores) has landed
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // (16 whole-line stores per lane and tile)
    BK_BAR();  // ... for every wave; the other halo buffer and t1 / t2 of the previous tile are free
    if (t_next < a.n_tiles) issue_x(td_next, (it & 1) ^ 1);
    // y byte offset of patch position m (0xffffffff: no such pixel), entry [m >> 5][m & 7][(m >> 3) & 3] (see bottleneck64_kernel)
    if (tid < 128) {
      const int m = tid;
      const int my = (int)(((unsigned)m * inv_pw) >> 16), mx = m - my * PW;
      const bool pv = m < NPOS && oy0 + my < IH && ox0 + mx < IW;
      const unsigned off = pv ? (unsigned)((oy0 + my) * IW + ox0 + mx) * (unsigned)(C * 2) : 0xffffffffu;  // inside the image
      *reinterpret_cast<unsigned*>(smem + PTAB_OFF + ((m >> 5) * 32 + (m & 7) * 4 + ((m >> 3) & 3)) * 4) = off;
    }

    // ===== P1: t1 = relu(bn1(x_halo . W1^T)) =================================================================================
    {
      int lz1;
      asm volatile("v_mov_b32 %0, 0" : "=v"(lz1));
      const int l31a = l31 + lz1;
      f32x16 acc1[3];
      const char* abase = xb + (rb1 * 32 + l31a) * 128;
      const int akey = ((((l31a >> 1) & 7)) ^ lh) << 4;
      bf16x8 af[4][3];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const char* ap = abase + ((ks * 32) ^ akey);
#pragma unroll
        for (int i = 0; i < 3; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(ap + i * 4096);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (ks == 0) BK_MFMA0(acc1[i], W1f[ks], af[ks][i]);
          else BK_MFMA(acc1[i], W1f[ks], af[ks][i]);
        }
      BK_MFMA_DRAIN3(acc1[0], acc1[1], acc1[2]);
      const float* sp = s1 + ct1 * 32 + 4 * lh;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int h = (rb1 + i) * 32 + l31a;
        const int hy = (int)(((unsigned)h * inv_hw2) >> 16), hx = h - hy * HW2;
        const bool in1 = h < HR && (unsigned)(oy0 - 1 + hy) < (unsigned)IH && (unsigned)(ox0 - 1 + hx) < (unsigned)IW;
        const unsigned keep = in1 ? 0xffffffffu : 0u;
        char* wp = t1 + h * TP + ct1 * 64 + 8 * lh;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(sp + 8 * gq), bv = *reinterpret_cast<const f32x4*>(sp + MID + 8 * gq);
          u32x2 o;
          o[0] = relu_pk(pack2(acc1[i][4 * gq] * sv[0] + bv[0], acc1[i][4 * gq + 1] * sv[1] + bv[1]), keep);
          o[1] = relu_pk(pack2(acc1[i][4 * gq + 2] * sv[2] + bv[2], acc1[i][4 * gq + 3] * sv[3] + bv[3]), keep);
          *reinterpret_cast<u32x2*>(wp + gq * 16) = o;
        }
      }
    }
    // the halo-centre rows (the x half of conv3's A operand) of this lane's four positions
    int xoff[4];
    int lzr;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lzr));
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int m = rt * 32 + l31 + lzr;
      const int my = (int)(((unsigned)m * inv_pw) >> 16), mx = m - my * PW;
      const int hc = min((my + 1) * HW2 + mx + 1, XROWS - 1);
      xoff[rt] = hc * 128 + (((((hc >> 1) & 7)) ^ lh) << 4);  // chunk pair ks' of this row sits at xoff ^ (ks' * 32)
    }
    BK_BAR();  // t1 complete

    // ===== P2: t2 = relu(bn2(conv3x3(t1))) ==================================================================================
    {
      int lz2;
      asm volatile("v_mov_b32 %0, 0" : "=v"(lz2));
      const int l31b = l31 + lz2;
      f32x16 acc2[2];
      const char* hrow[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = (rb2 + i) * 32 + l31b;
        const int my = (int)(((unsigned)m * inv_pw) >> 16);
        hrow[i] = t1 + (my * HW2 + (m - my * PW)) * TP + 16 * lh;
      }
      // Fragment ring, D2 - 1 k-steps ahead of the MFMAs.  The reads are inline asm with COUNTED waits: the next patch's halo is
      // in flight (LDS-DMA) during this phase, and while one is pending hipcc treats the LGKM counter as out of order and turns
      // every wait for a compiler-visible ds_read into lgkmcnt(0) -- i.e. it would wait for the reads just issued as well.
      constexpr int D2 = 4;
      bf16x8 af[D2][2];
      auto p2_read = [&](int k) {  // k-step k = tap * 4 + ks: both row tiles
        const int tap = k >> 2, ks = k & 3, kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned ad = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(hrow[i] + (kh * HW2 + kw) * TP);
          if (ks == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(af[k % D2][i]) : "v"(ad));
          else if (ks == 1) asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(af[k % D2][i]) : "v"(ad));
          else if (ks == 2) asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(af[k % D2][i]) : "v"(ad));
          else asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(af[k % D2][i]) : "v"(ad));
        }
      };
#pragma unroll
      for (int k = 0; k < D2 - 1; ++k) p2_read(k);
#ifdef BK_NOP2
      constexpr int K2 = 8;
#else
      constexpr int K2 = 36;
#endif
#pragma unroll
      for (int k = 0; k < K2; ++k) {
        if (k + D2 - 1 < 36) p2_read(k + D2 - 1);
        // reads issued after those of k-step k: two per k-step still ahead
        const int ahead = (k + D2 - 1 < 36 ? D2 - 1 : 35 - k) * 2;
        if (ahead == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af[k % D2][0]), "+v"(af[k % D2][1]));
        else if (ahead == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[k % D2][0]), "+v"(af[k % D2][1]));
        else if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[k % D2][0]), "+v"(af[k % D2][1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[k % D2][0]), "+v"(af[k % D2][1]));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (k == 0) BK_MFMA0(acc2[i], W2f[k], af[k % D2][i]);
          else BK_MFMA(acc2[i], W2f[k], af[k % D2][i]);
        }
      }
      BK_MFMA_DRAIN2(acc2[0], acc2[1]);
      BK_BAR();  // every wave has finished reading t1: t2 may overwrite it
      const float* sp = s2 + ct1 * 32 + 4 * lh;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        char* wp = t1 + ((rb2 + i) * 32 + l31b) * TP + ct1 * 64 + 8 * lh;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const f32x4 sv = *reinterpret_cast<const f32x4*>(sp + 8 * gq), bv = *reinterpret_cast<const f32x4*>(sp + MID + 8 * gq);
          u32x2 o;
          o[0] = relu_pk(pack2(acc2[i][4 * gq] * sv[0] + bv[0], acc2[i][4 * gq + 1] * sv[1] + bv[1]), 0xffffffffu);
          o[1] = relu_pk(pack2(acc2[i][4 * gq + 2] * sv[2] + bv[2], acc2[i][4 * gq + 3] * sv[3] + bv[3]), 0xffffffffu);
          *reinterpret_cast<u32x2*>(wp + gq * 16) = o;
        }
      }
    }
    BK_BAR();  // t2 complete

    // ===== P3: y = relu([t2 | x] . [W3 | Wsc]^T + shift), software pipelined over the row tiles like the identity block ======
    {
      int lz3;
      asm volatile("v_mov_b32 %0, 0" : "=v"(lz3));
      const char* pbase = t1 + (l31 + lz3) * TP + 16 * lh;
      const char* xc = xb + lz3;
      bf16x8 av[2][8];
      auto load_av = [&](int b, int rt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) av[b][ks] = *reinterpret_cast<const bf16x8*>(pbase + rt * 32 * TP + ks * 32);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) av[b][4 + ks] = *reinterpret_cast<const bf16x8*>(xc + (xoff[rt] ^ (ks * 32)));
      };
      load_av(0, 0);
      load_av(1, 1);
      f32x16 acc3[2][2];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (ks == 0) BK_MFMA0(acc3[0][j], W3f[j][0], av[0][0]);
          else BK_MFMA(acc3[0][j], W3f[j][ks], av[0][ks]);
        }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int cb = rt & 1, nb = cb ^ 1;
        BK_MFMA_DRAIN2(acc3[cb][0], acc3[cb][1]);
        if (rt + 2 < 4) load_av(cb, rt + 2);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int j = c >> 2, gq = c & 3;
          if (rt < 3) {  // two of tile rt + 1's sixteen MFMAs per epilogue chunk: k-step c, both channel tiles
            if (c == 0) { BK_MFMA0(acc3[nb][0], W3f[0][0], av[nb][0]); BK_MFMA0(acc3[nb][1], W3f[1][0], av[nb][0]); }
            else { BK_MFMA(acc3[nb][0], W3f[0][c], av[nb][c]); BK_MFMA(acc3[nb][1], W3f[1][c], av[nb][c]); }
          }
          const f32x4 bv = b3r[j][gq];
          u32x2 o;
          o[0] = relu_pk(pack2(acc3[cb][j][4 * gq] + bv[0], acc3[cb][j][4 * gq + 1] + bv[1]), 0xffffffffu);
          o[1] = relu_pk(pack2(acc3[cb][j][4 * gq + 2] + bv[2], acc3[cb][j][4 * gq + 3] + bv[3]), 0xffffffffu);
          asm volatile("ds_write_b64 %0, %1" ::"v"(stg + l31 * 128 + (((c ^ (l31 & 7))) << 4) + lh * 8), "v"(o) : "memory");
        }
        // whole-line stores through the wave's staging tile (see bottleneck64_kernel)
        u32x4 ov[4], yo;
        asm volatile("ds_read_b128 %0, %1" : "=v"(yo) : "v"(smem_lds + PTAB_OFF + rt * 128 + (lane >> 3) * 16));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (lane >> 3) + 8 * q;
          asm volatile("ds_read_b128 %0, %1" : "=v"(ov[q]) : "v"(stg + row * 128 + (((lane & 7) ^ (row & 7)) << 4)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ov[0]), "+v"(ov[1]), "+v"(ov[2]), "+v"(ov[3]), "+v"(yo));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          char* dst = yo[q] != 0xffffffffu ? yimg + ((size_t)yo[q] + 128 * wave + (lane & 7) * 16) : trash;
          *reinterpret_cast<u32x4*>(dst) = ov[q];
        }
      }
    }
    t = t_next;
    td = td_next;
  }
}

// small != 0: the double-buffered 64-position variant (the tile table must hold patches of <= 64 positions with <= 128 halo rows)
int launch_bottleneck64(const BottleneckArgs& a, int small, hipStream_t s) {
  constexpr int lds_stage = 4 * 4096 + 512;  // store staging + position table
  constexpr int lds_big = 192 * 512 + 192 * TP + (4 * MID + 2 * C) * 4 + lds_stage;
  static_assert(lds_big == LDS_BYTES + lds_stage && lds_big <= 160 * 1024, "LDS budget");
  static PerDeviceOnce once;
  const int dev = current_device(), ncu = device_cu_count(dev);
  if (!once.run(dev, [] { return hipFuncSetAttribute((const void*)bottleneck64_kernel<3, 2, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big) == hipSuccess; })) return -7;
  // the tile walk pairs blockIdx & 7 (XCD) with blockIdx >> 3: the grid must be a whole number of 8-block rounds
  const int want = (a.n_tiles + 7) & ~7;
  const int grid = want < ncu ? want : (ncu & ~7);
  if (small) return -1;  // the 64-position double-buffered variant (round 3, measured 1.56 vs 1.31 ms) no longer fits beside the store staging
  hipLaunchKernelGGL((bottleneck64_kernel<3, 2, 4, false>), dim3(grid), dim3(256), lds_big, s, a);
  return (int)hipGetLastError();
}

int launch_bottleneck64p(const BottleneckArgs& a, hipStream_t s) {
  static PerDeviceOnce once;
  const int dev = current_device(), ncu = device_cu_count(dev);
  if (!once.run(dev, [] { return hipFuncSetAttribute((const void*)bottleneck64p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PLDS_BYTES) == hipSuccess; })) return -7;
  const int want = (a.n_tiles + 7) & ~7;
  const int grid = want < ncu ? want : (ncu & ~7);
  hipLaunchKernelGGL(bottleneck64p_kernel, dim3(grid), dim3(256), PLDS_BYTES, s, a);
  return (int)hipGetLastError();
}

}  // namespace sylph
