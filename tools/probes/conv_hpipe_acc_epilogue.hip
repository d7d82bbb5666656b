// PROBE (not built into the product): conv_hpipe.hip with the round-6 accumulator-layout epilogue (one bf16 staging pass, one barrier).
// Parity-clean, measured equal on the layer micro-benchmark and 0.3-1 % slower in the headline step: profiles/r6_hpipe_epilogue.txt.
// Build: cp over csrc/conv_hpipe.hip (or tools/build_variant.sh with this file) -- same interface.
// Deep-pipelined 3x3 (stride 1, pad 1) implicit-GEMM convolution with a HALO A operand, for the MFMA-bound layers
// (FCOS towers fcos.py:72-122, FPN output convs, bottleneck conv2 of res4/res5).
//
// One 512-thread block per CU owns a 256 (positions) x 256 (channels) output tile.  The 256 positions are TWO
// independent ph x pw patches (<= 128 positions each, shapes picked per pyramid level on the host: 10 x 12 on the
// 100 x 168 / 50 x 84 maps), one per wave row:
//
//   * 8 waves = 2 (patch) x 4 (64-channel column); wave tile 128 x 64 = 4 x 2 MFMA 32x32x16 tiles (128 accumulator
//     VGPRs): 12 ds_read_b128 per 16 MFMAs.
//   * K order: 32-channel half-slice outer, the 9 taps inner.  A operand: the (ph+2) x (pw+2) input halo of each patch
//     for one half-slice (64-byte rows) is fetched ONCE and all nine taps read shifted rows of it, so the only per-tap
//     traffic is the weight tile (256 rows x 64 B = 16 KiB): L2->LDS bytes per flop are 0.58x those of a plain
//     256x256 GEMM tile (the round-1 conv_pipe kernel starved on exactly that stream) and 0.36x those of the
//     128x128 halo kernel.
//   * LDS: four 16-KiB weight stages (ring) + two 32-KiB halo buffers (double buffer) = 128 KiB.  The weight tile of
//     phase q+3 and (during taps 0..3) one quarter of the next half-slice's halo are issued by global_load_lds right after
//     the fragment reads of phase q and waited for two phases later with a COUNTED s_waitcnt vmcnt(N): never a drain in
//     the steady state.  The 9 taps are unrolled, so every N and every tap offset is an immediate.  The loads sit in the
//     L (fragment read) segment, whose instruction stream has the slack; the M segment is 16 bare MFMAs.  Load addresses
//     cost no VALU work: weights use a wave-uniform base + a constant per-lane offset, the halo running per-lane pointers.
//   * The two wave rows (the two waves that share a SIMD) run staggered by one barrier: while one issues its 16
//     MFMAs (s_setprio 1) the other does its fragment reads and address arithmetic.  Raw s_barrier + explicit
//     waitcnts only (a __syncthreads would drain the LDS-DMA queue).
//
// Hazards (B_n = n-th barrier, seg n = between B_n and B_n+1; row 0 does L(q) in seg 2q and M(q) in seg 2q+1, row 1 one
// segment later):
//   RAW  weight stage of phase q+1 (issued in L(q-2)) is read from seg 2q+2 on; both rows execute their vmcnt wait for
//        their own phase-(q+1) loads in seg 2q+1 (row 0 at the end of M(q), row 1 at the end of L(q)), i.e. before B_{2q+2}.
//        The halo of half-slice c+1 is issued in L(c,0..3); the in-order vmcnt waits of phases (c,4..6) retire it long
//        before L(c+1,0).
//   WAR  stage (q-1)&3 (and, at tap 0, the halo buffer last read in L(q-1)) is refilled by row 0 in seg 2q and by row 1 in
//        seg 2q+1; the last reader (row 1, L(q-1), seg 2q-1) retires its ds_reads with lgkmcnt(0) before B_{2q}.
//
// Weights are RE-PACKED for this kernel (hpipe_pack_weights_kernel, once per layer): [n tile][half-slice][tap] -> one
// contiguous 16-KiB block that already is the LDS image of a stage.  With the generic [n][kh][kw][c] layout the 256 rows of
// a phase sit 4 608 B apart: every phase touched the same 4 of the 16 L2 channels, in half cache lines, from all CUs at
// once, and the kernel ran 32 % slower than with the loads removed.  Now a stage is a linear 16-KiB copy (128 full lines
// over all channels).  Blocks also start the K loop at different half-slices (rotation by tile index), which spreads the
// 64-byte-per-512-byte halo rows of concurrently running blocks over the channels.
//
// LDS images (lane-linear global_load_lds, swizzle on the SOURCE side, same XOR on the fragment reads):
//   weights [256 rows][64 B]: slot s of row r holds 16-byte chunk s ^ ((r >> 2) & 3) (applied by the re-pack);
//   halo    [2 patches][256 rows][64 B], row h = hy * (pw + 4) + hx: slot s holds chunk s ^ ((k >> 2) & 3), k = hy * pw + hx.
//           The reader of tap (kh, kw) at patch position m sits on k = m + kh * pw + kw; the 16 lanes of a ds_read_b128
//           group ({0-3, 12-15, 20-27} + ...) hold 16 distinct k mod 16.  The 16-byte bank slot of a read is
//           (h & 3) * 4 + slot, and with the pitch pw + 4 (two unused entries per halo row) h = k + 4 hy, so it depends on
//           k mod 16 only: conflict-free for every patch shape.  (Pitch pw + 2 made it depend on the parity of hy: measured
//           35 % of all LDS cycles were bank conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.)
//
// Scope (launch_conv checks): bf16 in/out, 3x3 s1 p1, no residual, no per-segment Scale, ReLU on all channels or none,
// Cout % 256 == 0, Cin % 32 == 0, padded scale/shift; optional fused GroupNorm partial statistics (one per patch).
#include <stdlib.h>

#include "common.h"

// Ablation switches (measurement aids: HP_NOWAITV, HP_NOLDS, HP_NOLOAD, HP_NOEPI) exist ONLY in builds made with -DSYLPH_ABLATE
// (tools/build_variant.sh -> lib/variants/): the product library is compiled without it and every switch is forced off here.
#ifndef SYLPH_ABLATE
#undef HP_NOWAITV
#undef HP_NOLDS
#undef HP_NOLOAD
#undef HP_NOEPI
#undef HP_TIMING
#endif

namespace sylph {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

namespace {
constexpr int PNT = 512;
constexpr int BSTAGE = 256 * 64;               // one weight stage: 256 output channels x 32 input channels
constexpr int NSTAGE = 4;
constexpr int HPROWS = 256;                    // halo rows reserved per patch
constexpr int HBUF = 2 * HPROWS * 64;          // one halo buffer (both patches)
constexpr int HALO_OFF = NSTAGE * BSTAGE;
constexpr int COEF_OFF = HALO_OFF + 2 * HBUF;   // fused input GroupNorm: (a, b) per patch and input channel, [2][Cin] float2
constexpr int COEF_MAX_CIN = 512;
constexpr int SS_OFF = COEF_OFF + 2 * COEF_MAX_CIN * 8;     // epilogue scale [256] | shift [256] of this block's channels (fp32)
constexpr int LDS_BYTES = SS_OFF + 2 * 256 * 4;             // 141 312
static_assert(256 * 512 <= COEF_OFF, "the bf16 epilogue tile (256 rows x 512 B) aliases the weight ring + halo buffers only");

// loads a wave issues in the L segment of tap t: the two weight halves of phase q+3, plus one halo piece on taps 0..3
constexpr int NPIECE = 2 * HPROWS / 128;  // block-wide halo loads per half-slice (taps 0..3 carry one each)
constexpr int nload(int t) { return ((t % 9 + 9) % 9) < NPIECE ? 3 : 2; }

#define HP_SCHED_FENCE __builtin_amdgcn_sched_barrier(0)
#define HP_BAR()                                   \
  do {                                             \
    asm volatile("" ::: "memory");                 \
    HP_SCHED_FENCE;                                \
    __builtin_amdgcn_s_barrier();                  \
    HP_SCHED_FENCE;                                \
    asm volatile("" ::: "memory");                 \
  } while (0)
#ifdef HP_NOWAITV
#define HP_WAITV(N) asm volatile("" ::: "memory")
#else
#define HP_WAITV(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#endif
#define HP_WAITL() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
}  // namespace

template <bool GNIN>
__global__ __launch_bounds__(PNT, 1) void conv_hpipe_kernel(const ConvArgs a) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware block -> tile map (as conv_igemm.hip); an M tile is a PAIR of patches
  const int L = blockIdx.x;
  const int xcd = L & 7, q0 = L >> 3;
  const int chunk = (a.n_mtiles + 7) >> 3;
  const int m_local = q0 / a.n_ntiles;
  const int nt = q0 - m_local * a.n_ntiles;
  const int mt = xcd * chunk + m_local;
  if (m_local >= chunk || mt >= a.n_mtiles) return;
#ifdef HP_TIMING  // (SYLPH_ABLATE builds) s_memtime stamps of a block's prologue / K loop / epilogue, printed by 64 blocks of a launch
  const unsigned long long hp_t0 = __builtin_readcyclecounter();
  unsigned long long hp_t1 = 0, hp_t2 = 0;
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  // epilogue scale / shift of the block's 256 channels: fetched FIRST (ahead of the LDS-DMA queue, so that the wait for it is not a wait
  // for the operand loads), parked in LDS behind B_0, read back by the epilogue with 100-cycle LDS reads instead of global loads
  f32x4 ssv = {1.f, 1.f, 1.f, 1.f};
  if (tid < 64) { if (a.scale) ssv = *reinterpret_cast<const f32x4*>(a.scale + nt * 256 + tid * 4); }
  else if (tid < 128) { ssv = (f32x4){0.f, 0.f, 0.f, 0.f}; if (a.shift) ssv = *reinterpret_cast<const f32x4*>(a.shift + nt * 256 + (tid - 64) * 4); }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;

  int2 tl0 = a.tiles[2 * mt], tl1 = a.tiles[2 * mt + 1];
  const int rot_key = (int)((unsigned)tl0.x >> 20);  // index of this pair inside its image (api_conv.hip make_geom_patch)
  tl0.x &= 0xfffff;
  const SegDesc sd0 = a.segs[tl0.x], sd1 = a.segs[tl1.x];

  const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
  const T* __restrict__ wt = reinterpret_cast<const T*>(a.wt);
  const T* __restrict__ zero = reinterpret_cast<const T*>(a.zeros);
  const int Cin = a.Cin;
  const int ncc = Cin >> 5;  // 32-channel half-slices
  const int goff = a.group_cout > 0 ? ((nt * 256) / a.group_cout) * a.group_in_off : 0;
  // K-loop rotation: this block walks the half-slices c0, c0+1, ... (mod ncc); keyed on the pair's place inside its own image, so the
  // summation order of an image's outputs does not depend on the batch around it
  const int c0 = (rot_key + nt) % ncc;

  // ---- loader state --------------------------------------------------------------------------------------------
  // lane (r4, s4) of a block-wide global_load_lds fetches 16-byte slot s4 of LDS row (round * 128 + r4)
  const int r4 = tid >> 2, s4 = tid & 3;
  // halo: LDS rows [0, 256) patch 0, [256, 512) patch 1; four rounds of 128 rows, halo row (hy, hx) at hy * hpitch + hx.
  // Each lane keeps a running 64-bit source pointer per halo piece (advanced by 64 B per half-slice; lanes outside the
  // image / past the halo stay on the zero page), so issuing a piece costs no VALU work.
  const char* hptr[NPIECE];
  unsigned hmask = 0, hcs = 0;  // per piece: inside the image?  / logical 16-byte chunk (8 channels) this lane fetches
#pragma unroll
  for (int g = 0; g < NPIECE; ++g) {
    const bool p1 = g >= NPIECE / 2;
    const SegDesc& sd = p1 ? sd1 : sd0;
    const int ty = p1 ? tl1.y : tl0.y;
    const int h = g * 128 + r4 - (p1 ? HPROWS : 0);
    const int PW = sd.pw, HP = sd.hpitch, HR = (sd.ph + 2) * HP;
    const int hy = (int)(((unsigned)h * sd.inv_hw2) >> 16), hx = h - hy * HP;
    const int iy = (ty >> 16) - 1 + hy, ix = (ty & 0xffff) - 1 + hx;
    const bool ok = h < HR && hx < PW + 2 && (unsigned)iy < (unsigned)sd.in_H && (unsigned)ix < (unsigned)sd.in_W;
    const int cs = s4 ^ (((hy * PW + hx) >> 2) & 3);
    hcs |= (unsigned)cs << (2 * g);
    hptr[g] = ok ? reinterpret_cast<const char*>(in + ((size_t)(sd.in_row0 + iy * sd.in_W + ix) * a.in_ld + cs * 8 + goff + c0 * 32))
                 : reinterpret_cast<const char*>(zero + s4 * 8);
    hmask |= (ok ? 1u : 0u) << g;
  }
  // weights: stage image of (n tile, half-slice c, tap t) = 16 KiB at ((nt * ncc + c) * 9 + t) * 16 KiB; lane copies
  // bytes [tid * 16, +16) of each 8-KiB half: wave-uniform base (SALU) + a constant per-lane offset
  const char* const wtile = reinterpret_cast<const char*>(wt) + (size_t)nt * ncc * 9 * BSTAGE;
  const unsigned wvo = (unsigned)tid * 16u;
  bool loads_on = true;  // ablation builds (HP_NOLOAD) switch the main-loop loads off after a prologue that fills every stage
  auto issue_halo = [&](int g, int buf) {  // piece g of the half-slice the running pointers stand on -> halo buffer buf
    if (!loads_on) return;
    {
    char* d = smem + HALO_OFF + buf * HBUF + g * 8192 + wave * 1024;  // wave-uniform; lane l lands at +16 l
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)hptr[g], (lds_ptr_t)d, 16, 0, 0);
    }
  };
  // step the halo pointers to the next half-slice of the rotated walk (wrap: back by Cin - 32 channels)
  auto advance_halo = [&](int cc_next) {
    const int step = (c0 + cc_next == ncc) ? (32 - Cin) * 2 : 64;  // wave-uniform
#pragma unroll
    for (int g = 0; g < NPIECE; ++g) hptr[g] += ((hmask >> g) & 1u) ? step : 0;
  };
  auto issue_w = [&](int stage, int j, int blk) {  // 8-KiB half j of weight block blk (= rotated half-slice * 9 + tap)
    if (!loads_on) return;
    {
    char* d = smem + stage * BSTAGE + j * 8192 + wave * 1024;
    const char* src = wtile + (size_t)blk * BSTAGE + j * 8192 + wvo;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)d, 16, 0, 0);
    }
  };
  auto rot = [&](int cc) { const int c = c0 + cc; return c >= ncc ? c - ncc : c; };

  // Fused GroupNorm(+ReLU) of the input (a.gn_coef): every lane rewrites, in LDS, exactly the 16 bytes (8 channels of one
  // halo position) it fetched itself -- so the only ordering it needs is its own vmcnt wait for that load -- as
  // bf16(relu(a * x + b)).  Lanes on the zero page (conv padding, halo pad entries) are skipped: the padding of the
  // NORMALISED tensor is zero.  Readers see the result after the next lgkmcnt(0) + barrier, phases before its first use.
  constexpr bool gn_in = GNIN;  // the plain instantiation carries none of this
  const bool gn_relu = a.gn_relu != 0;
  // The LDS accesses are inline asm: for a compiler-visible ds_read hipcc inserts s_waitcnt vmcnt(0) (it must assume the
  // LDS-DMA still in flight aliases the read), which would drain the whole load pipeline in 4 of 9 phases.  The data read
  // here was fetched by THIS lane and retired by this wave's counted vmcnt two phases ago; the lgkmcnt wait is tied to the
  // loaded registers ("+v") so that no consumer can be scheduled above it.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  u32x4 gx;
  f32x4v gc0, gc1, gc2, gc3;  // (a, b) pairs of channels (0,1) (2,3) (4,5) (6,7)
  auto gn_addr = [&](int g, int cc_of_piece) { return lds0 + HALO_OFF + (cc_of_piece & 1) * HBUF + g * 8192 + tid * 16; };
  auto gn_read = [&](int g, int cc_of_piece) {  // issue the five LDS reads of piece g (no wait)
    const unsigned d = gn_addr(g, cc_of_piece);
    const int ch = rot(cc_of_piece) * 32 + (int)((hcs >> (2 * g)) & 3u) * 8;
    const unsigned cf = lds0 + COEF_OFF + ((g >= NPIECE / 2 ? Cin : 0) + ch) * 8;
    asm volatile("ds_read_b128 %0, %1" : "=v"(gx) : "v"(d));
    asm volatile("ds_read_b128 %0, %1" : "=v"(gc0) : "v"(cf));
    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(gc1) : "v"(cf));
    asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(gc2) : "v"(cf));
    asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(gc3) : "v"(cf));
  };
  auto gn_finish = [&](int g, int cc_of_piece) {  // wait for them, transform, write back (branch-free: no cut in the MFMA stream)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gx), "+v"(gc0), "+v"(gc1), "+v"(gc2), "+v"(gc3));
    const unsigned live = ((hmask >> g) & 1u) ? 0xffffffffu : 0u;  // zero-page lanes (conv padding) keep their zeros
    const f32x4v cs[4] = {gc0, gc1, gc2, gc3};                      // per channel pair: (a_lo, a_hi, b_lo, b_hi)
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef short s16x2v __attribute__((ext_vector_type(2)));
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    u32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x2v xv = {__uint_as_float(gx[e] << 16), __uint_as_float(gx[e] & 0xffff0000u)};
      const f32x2v av = {cs[e][0], cs[e][1]}, bv = {cs[e][2], cs[e][3]};
      const f32x2v r = __builtin_elementwise_fma(xv, av, bv);
      bf16x2 pk;
      pk[0] = (bf16_t)r[0]; pk[1] = (bf16_t)r[1];
      unsigned u = __builtin_bit_cast(unsigned, pk);
      if (gn_relu) {  // ReLU on the bf16 pair: packed signed max with 0
        const s16x2v z = {0, 0};
        u = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2v, u), z));
      }
      y[e] = u & live;
    }
    asm volatile("ds_write_b128 %0, %1" ::"v"(gn_addr(g, cc_of_piece)), "v"(y) : "memory");
  };
  auto gn_piece = [&](int g, int cc_of_piece) { gn_read(g, cc_of_piece); gn_finish(g, cc_of_piece); };

  // ---- fragment addressing -----------------------------------------------------------------------------------------
  const SegDesc& sdm = wm ? sd1 : sd0;  // this wave row's patch
  const int PWm = sdm.pw, HW2m = sdm.hpitch;
  int a0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = i * 32 + l31;
    const int my = (int)(((unsigned)m * sdm.inv_pw) >> 16);
    a0[i] = (wm * HPROWS + my * HW2m + (m - my * PWm)) * 64;
  }
  int offB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) offB[ks] = (wn * 64 + l31) * 64 + (((ks * 2 + lh) ^ ((l31 >> 2) & 3)) << 4);

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 fa[2][4], fb[2][2];

  // L(cc, t): the 12 fragment reads of tap t of half-slice cc
  auto ldfrag = [&](int cc, int t) {
    const int kh = t / 3, kw = t - 3 * kh;
    const char* bs = smem + ((cc + t) & 3) * BSTAGE;  // phase q = 9 cc + t; q & 3 == (cc + t) & 3
    const char* hs = smem + HALO_OFF + (cc & 1) * HBUF + (kh * HW2m + kw) * 64;
    const int f = ((l31 + kh * PWm + kw) >> 2) & 3;
#ifdef HP_NOLDS
    if (cc + t > 0) return;
#endif
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = ((ks * 2 + lh) ^ f) << 4;
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(bs + offB[ks] + j * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(hs + a0[i] + so);
    }
  };

  // The loads of phase q+3 = tap t+3 (weights into stage (q+3) & 3) and, on taps 0..2, piece t of the next half-slice's
  // halo.  Issued in the L segment, after the fragment reads: the other wave row is in its MFMA segment meanwhile.
  // Past the last phase the weight loads re-read K offset 0 into a stage nobody reads any more (constant load count).
  auto issue_next = [&](int cc, int t) {
    const int t3 = (t + 3) % 9, cc3 = cc + (t + 3) / 9;
    const int blk = cc3 < ncc ? rot(cc3) * 9 + t3 : 0;
    const int st3 = (cc + t + 3) & 3;
    if (t == 0 && cc + 1 < ncc) advance_halo(cc + 1);  // the pointers now stand on (rotated) half-slice cc + 1; the last one is re-read at the end: harmless
    if (t < NPIECE) issue_halo(t, (cc + 1) & 1);
    issue_w(st3, 0, blk);
    issue_w(st3, 1, blk);
  };
  // M(cc, t): 16 back-to-back MFMAs.  With a fused input GroupNorm, taps 3..6 also transform halo piece t - 3 of the next
  // half-slice (landed: its load was issued in L(cc, t - 3) and retired by this wave's counted wait two phases later); the
  // ~40 VALU / LDS instructions are spread between the MFMAs (sched_group_barrier), where the wave has free issue slots.
  auto mma = [&](int cc, int t) {
    __builtin_amdgcn_s_setprio(1);
    // (unconditional on the last half-slice too: it then rewrites the re-read copy in the buffer nobody reads any more)
    const bool xf = gn_in && t >= 3 && t < 3 + NPIECE;
    if (xf) {  // the five LDS reads go out first; their latency hides behind the first four MFMAs
      gn_read(t - 3, cc + 1);
      HP_SCHED_FENCE;
    }
    int n = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);  // D^T
          if (xf && n == 3) {
            HP_SCHED_FENCE;
            gn_finish(t - 3, cc + 1);  // ~40 VALU + one LDS write, spread between the remaining MFMAs below
          }
          ++n;
        }
    if (xf) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // up to four VALU
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // Tell the compiler's waitcnt pass that the fragments are complete HERE (it is called right after the explicit lgkmcnt(0)
  // of the L segment): otherwise it inserts its own s_waitcnt lgkmcnt(0) at the first MFMA of the M segment, behind the
  // GroupNorm transform's LDS reads issued there, and their latency is exposed again.
  auto frags_ready = [&]() {
    asm volatile("" ::"v"(fa[0][0]), "v"(fa[0][1]), "v"(fa[0][2]), "v"(fa[0][3]), "v"(fa[1][0]), "v"(fa[1][1]), "v"(fa[1][2]),
                 "v"(fa[1][3]), "v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[1][0]), "v"(fb[1][1]));
  };

  // ---- prologue: halo of half-slice 0 and the weights of phases 0..2 ---------------------------------------------
#pragma unroll
  for (int g = 0; g < NPIECE; ++g) issue_halo(g, 0);
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    issue_w(t, 0, c0 * 9 + t);
    issue_w(t, 1, c0 * 9 + t);
  }
#ifdef HP_NOLOAD
  issue_w(3, 0, c0 * 9 + 3); issue_w(3, 1, c0 * 9 + 3);
  for (int g = 0; g < NPIECE; ++g) issue_halo(g, 1);
  HP_WAITV(0);
  loads_on = false;
#endif
  if (gn_in) {  // (a, b) of both patches' segments -> LDS (plain loads: issued after the DMA queue, waited below)
    // stored per channel PAIR as (a0, a1, b0, b1): the transform is then one packed FMA per bf16 pair
    for (int idx = tid; idx < Cin; idx += PNT) {
      const int pch = idx >= Cin / 2 ? 1 : 0, ch = (idx - pch * (Cin / 2)) * 2;
      const int seg = pch ? tl1.x : tl0.x;
      const float2 c0v = a.gn_coef[(size_t)seg * a.in_ld + goff + ch], c1v = a.gn_coef[(size_t)seg * a.in_ld + goff + ch + 1];
      *reinterpret_cast<float4*>(smem + COEF_OFF + (pch * Cin + ch) * 8) = make_float4(c0v.x, c1v.x, c0v.y, c1v.y);
    }
  }
  HP_WAITV(4);  // halo + phase 0 landed (phases 1, 2 in flight)
  if (tid < 128) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + SS_OFF + tid * 16), "v"(ssv) : "memory");
  if (gn_in) {
    HP_WAITL();
    HP_BAR();   // the coefficient table is complete for every wave
#pragma unroll
    for (int g = 0; g < NPIECE; ++g) gn_piece(g, 0);
    HP_WAITL();
  }
  HP_BAR();     // B_0
#ifdef HP_TIMING
  hp_t1 = __builtin_readcyclecounter();
#endif

  // vmcnt immediates: a wave needs its loads of phase q+1 (issued in L(q-2)) landed before B_{2q+2}; the loads issued after
  // them are the groups of L(q-1) and L(q).  Row 0 waits at the end of M(q) (seg 2q+1), row 1 at the end of L(q) (seg 2q+1).
#define HP_PHASE0(t)                      \
  ldfrag(cc, t);                          \
  HP_SCHED_FENCE;                         \
  issue_next(cc, t);                      \
  HP_WAITL();                             \
  if (GNIN) frags_ready();                \
  HP_BAR();                               \
  mma(cc, t);                             \
  HP_WAITV(nload((t) - 1) + nload(t));    \
  HP_BAR();
#define HP_PHASE1(t)                      \
  ldfrag(cc, t);                          \
  HP_SCHED_FENCE;                         \
  issue_next(cc, t);                      \
  HP_WAITV(nload((t) - 1) + nload(t));    \
  HP_WAITL();                             \
  if (GNIN) frags_ready();                \
  HP_BAR();                               \
  mma(cc, t);                             \
  HP_BAR();

  if (wm == 0) {
    for (int cc = 0; cc < ncc; ++cc) {
      HP_PHASE0(0) HP_PHASE0(1) HP_PHASE0(2) HP_PHASE0(3) HP_PHASE0(4) HP_PHASE0(5) HP_PHASE0(6) HP_PHASE0(7) HP_PHASE0(8)
    }
    HP_BAR();
  } else {
    HP_BAR();  // the stagger
    for (int cc = 0; cc < ncc; ++cc) {
      HP_PHASE1(0) HP_PHASE1(1) HP_PHASE1(2) HP_PHASE1(3) HP_PHASE1(4) HP_PHASE1(5) HP_PHASE1(6) HP_PHASE1(7) HP_PHASE1(8)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zero-page tail loads must not land in the epilogue tile
  __syncthreads();
#ifdef HP_TIMING
  hp_t2 = __builtin_readcyclecounter();
#endif

  // ---- fused epilogue (round 6): transform in the accumulator layout, ONE bf16 staging pass, one barrier, whole-line stores --------
  // D^T layout: lane (l31, lh) of wave (wm, wn) holds, in acc[i][j][4g .. 4g+3], channels wn*64 + j*32 + 8g + 4lh .. +3 of patch wm's
  // position i*32 + l31.  scale / shift / ReLU / the GroupNorm partial sums are applied right there (packed fp32 math), the four values
  // become one 8-byte bf16 write into a [256 rows][256 ch] tile of pitch 520 B (130 dwords: the 32 rows of a half-wave's ds_write_b64
  // land on 32 distinct bank pairs), and after ONE barrier every thread copies 16 whole 16-byte chunks to HBM (a half-wave = one 512-byte
  // output row).  Rounds 1-5 made four 64-row passes through an fp32 tile with eight barriers (12 300 of a block's 111 000 cycles).
  // A patch's GroupNorm partial for a group of 8 channels lives entirely in ONE wave (its 128 positions x 8 channels): a butterfly over
  // the 64 lanes, no LDS, no cross-wave merge.
  constexpr int EP = 512;
  char* const sT = smem;
  const float* const ssl = reinterpret_cast<const float*>(smem + SS_OFF);
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(a.out);
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  const bool relu = a.relu_nch > 0;
  {
    const SegDesc& sp = wm ? sd1 : sd0;
    const int ty = wm ? tl1.y : tl0.y;
    const int oy0 = ty >> 16, ox0 = ty & 0xffff;
    const int PW = sp.pw, NPOS = sp.ph * sp.pw;
    float vmask[4];
    int nvalid = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = i * 32 + l31;
      const int my = (int)(((unsigned)m * sp.inv_pw) >> 16);
      const bool ok = m < NPOS && oy0 + my < sp.out_H && ox0 + (m - my * PW) < sp.out_W;
      vmask[i] = ok ? 1.f : 0.f;
      nvalid += __popcll(__ballot(ok) & 0xffffffffull);
    }
    const int chl = wn * 64;  // first channel of this wave inside the block's 256
    // tile row r = wm * 128 + i * 32 + l31; 8-byte slot q of a row (4 channels) sits at slot q ^ (r & 31): the 32 rows a half-wave
    // writes at one q land on 32 distinct bank pairs, and the two slots of a 16-byte chunk stay inside one (aligned) chunk
    char* const wrow = sT + (wm * 128 + l31) * EP;
    float gS1[8], gS2[8], gPv[8];  // per (j, g): this lane's shifted sums; reduced over the wave after the loop (16 independent butterflies)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = chl + j * 32 + 8 * g + 4 * lh;
        const float4 s4v = *reinterpret_cast<const float4*>(ssl + cl);
        const float4 b4v = *reinterpret_cast<const float4*>(ssl + 256 + cl);
        const f32x2v s01 = {s4v.x, s4v.y}, s23 = {s4v.z, s4v.w}, b01 = {b4v.x, b4v.y}, b23 = {b4v.z, b4v.w};
        // pivot of the group's shifted sums: the conv bias of its first channel (what makes |mean| >> sigma in practice); lanes lh = 0 / 1
        // of a group must agree on it
        const float pvs = ssl[256 + chl + j * 32 + 8 * g];
        const f32x2v pv = {pvs, pvs};
        f32x2v s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
        const int q = (cl >> 2) ^ l31;  // (r & 31) == l31 for every i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x16& c = acc[i][j];
          f32x2v v01 = {c[4 * g], c[4 * g + 1]}, v23 = {c[4 * g + 2], c[4 * g + 3]};
          v01 = __builtin_elementwise_fma(v01, s01, b01);
          v23 = __builtin_elementwise_fma(v23, s23, b23);
          if (relu) {
            v01[0] = fmaxf(v01[0], 0.f); v01[1] = fmaxf(v01[1], 0.f); v23[0] = fmaxf(v23[0], 0.f); v23[1] = fmaxf(v23[1], 0.f);
          }
          if (a.gn_partial) {
            const f32x2v vm = {vmask[i], vmask[i]};
            const f32x2v d01 = (v01 - pv) * vm, d23 = (v23 - pv) * vm;
            s1 += d01; s1 += d23;
            s2 = __builtin_elementwise_fma(d01, d01, s2);
            s2 = __builtin_elementwise_fma(d23, d23, s2);
          }
          typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
          bf16x4 o;
          o[0] = (bf16_t)v01[0]; o[1] = (bf16_t)v01[1]; o[2] = (bf16_t)v23[0]; o[3] = (bf16_t)v23[1];
          *reinterpret_cast<bf16x4*>(wrow + i * 32 * EP + q * 8) = o;
        }
        gS1[j * 4 + g] = s1[0] + s1[1]; gS2[j * 4 + g] = s2[0] + s2[1]; gPv[j * 4 + g] = pvs;
      }
    if (a.gn_partial) {  // (n, mean, M2) of this patch and each of the wave's 8 groups: sums over the 64 lanes, fixed butterfly order
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { gS1[k] += __shfl_xor(gS1[k], d); gS2[k] += __shfl_xor(gS2[k], d); }
      }
      if (lane < 8) {
        float S1 = gS1[0], S2 = gS2[0], pvs = gPv[0];
#pragma unroll
        for (int k = 1; k < 8; ++k)
          if (lane == k) { S1 = gS1[k]; S2 = gS2[k]; pvs = gPv[k]; }
        const float N = 8.f * (float)nvalid;
        const float inv_n = N > 0.f ? 1.f / N : 0.f;
        const float m2 = S2 - S1 * S1 * inv_n;
        float* gp = a.gn_partial + ((size_t)(2 * mt + wm) * (a.Cout >> 3) + ((nt * 256 + chl) >> 3) + lane) * 3;  // group j * 4 + g = lane
        gp[0] = N; gp[1] = pvs + S1 * inv_n; gp[2] = m2 > 0.f ? m2 : 0.f;
      }
    }
  }
  lds_barrier();
  {
    const int c8 = tid & 31, rr = tid >> 5;
    const int n0 = nt * 256 + c8 * 8;
    const bool odd = rr & 1;  // rows of this thread: rr + 16 it -> all of rr's parity
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const SegDesc& sp = pp ? sd1 : sd0;
      const int ty = pp ? tl1.y : tl0.y;
      const int oy0 = ty >> 16, ox0 = ty & 0xffff;
      const int PW = sp.pw, NPOS = sp.ph * sp.pw;
      bf16_t* __restrict__ outn = out + (size_t)sp.out_row0 * a.out_ld + n0;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = rr + 16 * it;
        const int my = (int)(((unsigned)m * sp.inv_pw) >> 16);
        const int oy = oy0 + my, ox = ox0 + (m - my * PW);
        const uint4 raw = *reinterpret_cast<const uint4*>(sT + (pp * 128 + m) * EP + ((c8 ^ ((m & 31) >> 1)) << 4));
        if (m < NPOS && oy < sp.out_H && ox < sp.out_W) {
          const uint4 v = odd ? make_uint4(raw.z, raw.w, raw.x, raw.y) : raw;
#ifndef HP_NOEPI
          *reinterpret_cast<uint4*>(outn + (size_t)(oy * sp.out_W + ox) * a.out_ld) = v;
#else
          if (v.x == 0x12345678u) *reinterpret_cast<uint4*>(outn + (size_t)(oy * sp.out_W + ox) * a.out_ld) = v;
#endif
        }
      }
    }
  }
#ifdef HP_TIMING
  if (L >= 2048 && L < 2048 + 64 && (tid == 0 || tid == 256)) {
    const unsigned long long t3 = __builtin_readcyclecounter();
    printf("blk %d row %d: prologue %llu  K loop %llu  epilogue %llu cycles\n", L, tid >> 8, hp_t1 - hp_t0, hp_t2 - hp_t1, t3 - hp_t2);
  }
#endif
}

// [Cout][3][3][Cin] bf16 (conv_igemm layout) -> [Cout / 256][Cin / 32][9][256 rows][4 slots][8] with the stage swizzle applied
__global__ void hpipe_pack_weights_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk
  const size_t nchunks = (size_t)Cout * 9 * Cin / 8;
  if (i >= nchunks) return;
  const int ncc = Cin >> 5;
  const int s = (int)(i & 3), r = (int)((i >> 2) & 255);
  const size_t blk = i >> 10;
  const int t = (int)(blk % 9), c = (int)((blk / 9) % ncc), nt = (int)(blk / (9 * (size_t)ncc));
  const int chunk = s ^ ((r >> 2) & 3);
  const uint4 v = *reinterpret_cast<const uint4*>(w + ((size_t)(nt * 256 + r) * 9 + t) * Cin + c * 32 + chunk * 8);
  *reinterpret_cast<uint4*>(out + i * 8) = v;
}

int launch_hpipe_pack_weights(const void* w, void* out, int Cout, int Cin, hipStream_t s) {
  const size_t nchunks = (size_t)Cout * 9 * Cin / 8;
  hipLaunchKernelGGL(hpipe_pack_weights_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, s, (const bf16_t*)w, (bf16_t*)out, Cout, Cin);
  return (int)hipGetLastError();
}

bool conv_hpipe_ok(DType dt, bool out_f32, const ConvArgs& a) {
  if (a.gn_coef && (a.Cin > COEF_MAX_CIN || a.group_cout > 0)) return false;
  return dt == DT_BF16 && !out_f32 && !a.stem && !a.in2 && a.res_mode == 0 && a.mul_nch == 0 && a.KH == 3 && a.KW == 3 &&
         a.stride == 1 && a.pad == 1 && (a.relu_nch == 0 || a.relu_nch >= a.Cout) && a.Cout % 256 == 0 && a.Cin % 32 == 0 &&
         a.Cin >= 32 && a.ss_padded_host && (a.out_ld & 7) == 0 && a.zeros != nullptr;
}

#ifdef SYLPH_ABLATE
int launch_conv_hq(const ConvArgs& a, hipStream_t s);  // tools/probes/conv_hpipe4.hip (linked by tools/probes/build_hq.sh only)
#endif

int launch_conv_hpipe(const ConvArgs& a, hipStream_t s) {
#if defined(SYLPH_ABLATE) && defined(SYLPH_HQ_PROBE)
  static const int hq = SYLPH_AB_ENV("SYLPH_CONV_HQ", 0);  // A/B: the four-wave probe kernel on the same tile table and stage images
  if (hq) return launch_conv_hq(a, s);
#endif
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_hpipe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -7;
    if (hipFuncSetAttribute((const void*)conv_hpipe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -7;
    attr_set = true;
  }
  const int chunk = (a.n_mtiles + 7) / 8;
  const int grid = 8 * chunk * a.n_ntiles;
  if (a.gn_coef) hipLaunchKernelGGL(conv_hpipe_kernel<true>, dim3(grid), dim3(PNT), LDS_BYTES, s, a);
  else hipLaunchKernelGGL(conv_hpipe_kernel<false>, dim3(grid), dim3(PNT), LDS_BYTES, s, a);
  return (int)hipGetLastError();
}

}  // namespace sylph
