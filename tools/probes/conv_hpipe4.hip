// MEASUREMENT PROBE, not part of the product (round-2 experiment "four-wave 128x128-wave-tile variant", git 054d86d, rebuilt against the
// round-5 tree by tools/probes/build_hq.sh for profiles/r5_tower_four_wave.txt).
// Four-wave variant of conv_hpipe.hip (same tile, same LDS images, same weight stage images, same geometry tables):
//
// conv_hpipe_kernel's fragment-read segment is LDS-bandwidth bound: a 128 x 64 wave tile needs 12 ds_read_b128 per 16 MFMAs,
// four waves are in that segment at a time (48 KiB) while the weight / halo DMA writes another ~20 KiB -- ~530 cycles of LDS
// traffic per 512-cycle MFMA segment, measured MFMA utilisation 0.63.  Here the 256 x 256 tile is owned by FOUR waves (one per
// SIMD, up to 512 registers each) with 128 x 128 wave tiles = 4 x 4 MFMA 32x32x16 tiles (256 accumulator registers):
// 16 fragment reads per 32 MFMAs (LDS bytes per flop x 0.67), and ONE barrier per 32-MFMA phase instead of two per 16.
// There is no second wave on a SIMD to hide the fragment reads, so they are software pipelined at half-phase granularity with ONE
// fragment set (64 registers): the k-step-1 fragments are read under the 16 k-step-0 MFMAs, the NEXT phase's k-step-0 fragments
// under the 16 k-step-1 MFMAs.
//
// Phase q = 9 cc + t (half-slice cc, tap t), all waves in lock step:
//     s_waitcnt vmcnt(N)        this wave's share of the weight stage of phase q + 1 (issued in phase q - 2) has landed
//     s_barrier                 ... and everybody's; every wave is done READING stage (q - 1) & 3
//     global_load_lds           weights of phase q + 3 -> stage (q - 1) & 3; taps 0..3: two 4-KiB pieces of the next half-slice's halo
//     8 x ds_read_b128          k-step-1 fragments of phase q
//     16 x MFMA                 k-step 0 of phase q
//     8 x ds_read_b128          k-step-0 fragments of phase q + 1
//     16 x MFMA                 k-step 1 of phase q   (taps 3..6 with a fused input GroupNorm: + one landed halo piece per half)
#include <stdlib.h>

#include "../../sylph-few-shot-detection_amd/csrc/common.h"

namespace sylph {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

namespace {
constexpr int PNT = 256;
constexpr int BSTAGE = 256 * 64;
constexpr int NSTAGE = 4;
constexpr int HPROWS = 256;
constexpr int HBUF = 2 * HPROWS * 64;
constexpr int HALO_OFF = NSTAGE * BSTAGE;
constexpr int COEF_OFF = HALO_OFF + 2 * HBUF;
constexpr int COEF_MAX_CIN = 512;
constexpr int LDS_BYTES = COEF_OFF + 2 * COEF_MAX_CIN * 8;
constexpr int SCP = 256 + 4;
static_assert(64 * SCP * 4 <= LDS_BYTES, "epilogue tile must fit");
constexpr int NPIECE = 2 * HPROWS / 64;  // 8 block-wide halo loads (64 rows each) per half-slice: two on each of taps 0..3
constexpr int nload(int t) { return ((t % 9 + 9) % 9) < 4 ? 6 : 4; }  // DMA instructions a lane issues in phase tap t

#define HQ_FENCE __builtin_amdgcn_sched_barrier(0)
#define HQ_BAR()                                   \
  do {                                             \
    asm volatile("" ::: "memory");                 \
    HQ_FENCE;                                      \
    __builtin_amdgcn_s_barrier();                  \
    HQ_FENCE;                                      \
    asm volatile("" ::: "memory");                 \
  } while (0)
#define HQ_WAITV(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define HQ_WAITL() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
}  // namespace

template <bool GNIN>
__global__ __launch_bounds__(PNT, 1) void conv_hq_kernel(const ConvArgs a) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int L = blockIdx.x;
  const int xcd = L & 7, q0 = L >> 3;
  const int chunk = (a.n_mtiles + 7) >> 3;
  const int m_local = q0 / a.n_ntiles;
  const int nt = q0 - m_local * a.n_ntiles;
  const int mt = xcd * chunk + m_local;
  if (m_local >= chunk || mt >= a.n_mtiles) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;  // patch, 128-channel half
  const int l31 = lane & 31, lh = lane >> 5;

  const int2 tl0 = a.tiles[2 * mt], tl1 = a.tiles[2 * mt + 1];
  const SegDesc sd0 = a.segs[tl0.x], sd1 = a.segs[tl1.x];

  const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
  const T* __restrict__ wt = reinterpret_cast<const T*>(a.wt);
  const T* __restrict__ zero = reinterpret_cast<const T*>(a.zeros);
  const int Cin = a.Cin;
  const int ncc = Cin >> 5;
  const int goff = a.group_cout > 0 ? ((nt * 256) / a.group_cout) * a.group_in_off : 0;
  const int c0 = (mt + nt) % ncc;

  // ---- loader state: lane (r4, s4) of a block-wide global_load_lds fetches 16-byte slot s4 of LDS row (piece * 64 + r4) ----
  const int r4 = tid >> 2, s4 = tid & 3;
  const char* hptr[NPIECE];
  unsigned hmask = 0, hcs = 0;
#pragma unroll
  for (int g = 0; g < NPIECE; ++g) {
    const bool p1 = g >= NPIECE / 2;
    const SegDesc& sd = p1 ? sd1 : sd0;
    const int ty = p1 ? tl1.y : tl0.y;
    const int h = g * 64 + r4 - (p1 ? HPROWS : 0);
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
  const char* const wtile = reinterpret_cast<const char*>(wt) + (size_t)nt * ncc * 9 * BSTAGE;
  const unsigned wvo = (unsigned)tid * 16u;
  auto issue_halo = [&](int g, int buf) {
    char* d = smem + HALO_OFF + buf * HBUF + g * 4096 + wave * 1024;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)hptr[g], (lds_ptr_t)d, 16, 0, 0);
  };
  auto advance_halo = [&](int cc_next) {
    const int step = (c0 + cc_next == ncc) ? (32 - Cin) * 2 : 64;
#pragma unroll
    for (int g = 0; g < NPIECE; ++g) hptr[g] += ((hmask >> g) & 1u) ? step : 0;
  };
  auto issue_w = [&](int stage, int blk) {  // the 16-KiB stage image of weight block blk (= rotated half-slice * 9 + tap)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      char* d = smem + stage * BSTAGE + j * 4096 + wave * 1024;
      const char* src = wtile + (size_t)blk * BSTAGE + j * 4096 + wvo;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)d, 16, 0, 0);
    }
  };
  auto rot = [&](int cc) { const int c = c0 + cc; return c >= ncc ? c - ncc : c; };

  // ---- fused input GroupNorm (see conv_hpipe.hip): every lane rewrites the 16 bytes it fetched itself ----------------------
  constexpr bool gn_in = GNIN;
  const bool gn_relu = a.gn_relu != 0;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  u32x4 gx;
  f32x4v gc0, gc1, gc2, gc3;
  auto gn_addr = [&](int g, int cc_of_piece) { return lds0 + HALO_OFF + (cc_of_piece & 1) * HBUF + g * 4096 + tid * 16; };
  auto gn_read = [&](int g, int cc_of_piece) {
    const unsigned d = gn_addr(g, cc_of_piece);
    const int ch = rot(cc_of_piece) * 32 + (int)((hcs >> (2 * g)) & 3u) * 8;
    const unsigned cf = lds0 + COEF_OFF + ((g >= NPIECE / 2 ? Cin : 0) + ch) * 8;
    asm volatile("ds_read_b128 %0, %1" : "=v"(gx) : "v"(d));
    asm volatile("ds_read_b128 %0, %1" : "=v"(gc0) : "v"(cf));
    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(gc1) : "v"(cf));
    asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(gc2) : "v"(cf));
    asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(gc3) : "v"(cf));
  };
  auto gn_finish = [&](int g, int cc_of_piece) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gx), "+v"(gc0), "+v"(gc1), "+v"(gc2), "+v"(gc3));
    const unsigned live = ((hmask >> g) & 1u) ? 0xffffffffu : 0u;
    const f32x4v cs[4] = {gc0, gc1, gc2, gc3};
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
      if (gn_relu) {
        const s16x2v z = {0, 0};
        u = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2v, u), z));
      }
      y[e] = u & live;
    }
    asm volatile("ds_write_b128 %0, %1" ::"v"(gn_addr(g, cc_of_piece)), "v"(y) : "memory");
  };
  auto gn_piece = [&](int g, int cc_of_piece) { gn_read(g, cc_of_piece); gn_finish(g, cc_of_piece); };

  // ---- fragment addressing ------------------------------------------------------------------------------------------------
  const SegDesc& sdm = wm ? sd1 : sd0;
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
  for (int ks = 0; ks < 2; ++ks) offB[ks] = (wn * 128 + l31) * 64 + (((ks * 2 + lh) ^ ((l31 >> 2) & 3)) << 4);

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 fa[2][4], fb[2][4];  // [k-step][tile]: ONE set; each half is refilled under the other half's MFMAs

  // the 8 fragment reads of k-step ks of tap t of half-slice cc (phase q = 9 cc + t).  Inline asm, like the GroupNorm transform:
  // while an LDS-DMA is pending hipcc treats the LGKM counter as out of order and turns EVERY wait for a compiler-visible ds_read
  // into lgkmcnt(0), wherever it chooses to put it.  frags_wait() is the explicit wait, tied to the registers of the half that
  // must have arrived.
  auto ldfrag = [&](int ks, int cc, int t) {
    const int kh = t / 3, kw = t - 3 * kh;
    const unsigned bs = lds0 + ((cc + t) & 3) * BSTAGE + offB[ks];  // q & 3 == (cc + t) & 3
    const int f = ((l31 + kh * PWm + kw) >> 2) & 3;
    const unsigned hs = lds0 + HALO_OFF + (cc & 1) * HBUF + (kh * HW2m + kw) * 64 + (((ks * 2 + lh) ^ f) << 4);
    asm volatile("ds_read_b128 %0, %1" : "=v"(fb[ks][0]) : "v"(bs));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(fb[ks][1]) : "v"(bs));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fb[ks][2]) : "v"(bs));
    asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(fb[ks][3]) : "v"(bs));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(fa[ks][i]) : "v"(hs + a0[i]));
  };
  // the reads of k-step ks were issued right after the FIRST of the previous 16 MFMAs: they have long arrived, the wait is only
  // the ordering point (and nothing newer is outstanding at this point, so the count is 0)
  auto frags_wait = [&](int ks) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[ks][0]), "+v"(fa[ks][1]), "+v"(fa[ks][2]), "+v"(fa[ks][3]), "+v"(fb[ks][0]), "+v"(fb[ks][1]), "+v"(fb[ks][2]),
                   "+v"(fb[ks][3]));
  };
  // DMA of phase q + 3 (weights into stage (q + 3) & 3 = (q - 1) & 3: its last readers finished before this phase's barrier)
  // and, on taps 0..3, two halo pieces of the next half-slice
  auto issue_next = [&](int cc, int t) {
    const int t3 = (t + 3) % 9, cc3 = cc + (t + 3) / 9;
    const int blk = cc3 < ncc ? rot(cc3) * 9 + t3 : 0;  // past the end: re-read block 0 into a stage nobody reads (constant load count)
    if (t == 0 && cc + 1 < ncc) advance_halo(cc + 1);
    if (t < 4) { issue_halo(2 * t, (cc + 1) & 1); issue_halo(2 * t + 1, (cc + 1) & 1); }
    issue_w((cc + t + 3) & 3, blk);
  };
  // 16 MFMAs of k-step ks; with a fused input GroupNorm taps 3..6 also transform halo piece 2 (t - 3) + ks of the next half-slice
  // (issued in tap t - 3, landed: retired by the counted vmcnt of tap t - 1)
  // 16 MFMAs of k-step ks.  Everything else a phase has to do rides in their shadow (one wave per SIMD: an instruction issued
  // between two MFMAs costs nothing while the matrix pipe is busy, the same instruction at the phase top idles it):
  //   k-step 0:  after MFMA 0 the eight k-step-1 fragment reads, after MFMA 3 the DMA issue of phase q + 3;
  //   k-step 1:  after MFMA 0 the eight k-step-0 reads of phase q + 1 (the k-step-0 MFMAs have all been issued: their operand
  //              registers are free);
  //   with a fused input GroupNorm, taps 3..6: the transform of halo piece 2 (t - 3) + ks of the next half-slice around MFMAs 8..14.
  auto mma = [&](int ks, int cc, int t) {
    const bool xf = gn_in && t >= 3 && t < 7;
    int n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);  // D^T
        if (n == 0) {
          if (ks == 0) ldfrag(1, cc, t);
          else if (t < 8) ldfrag(0, cc, t + 1);
          else ldfrag(0, cc + 1, 0);
        }
        if (ks == 0 && n == 3) issue_next(cc, t);
        if (xf && n == 8) gn_read(2 * (t - 3) + ks, cc + 1);
        if (xf && n == 13) gn_finish(2 * (t - 3) + ks, cc + 1);
        ++n;
      }
  };

  // ---- prologue: halo of half-slice 0, weights of phases 0..2, k-step-0 fragments of phase 0 ----------------------------------
#pragma unroll
  for (int g = 0; g < NPIECE; ++g) issue_halo(g, 0);
#pragma unroll
  for (int t = 0; t < 3; ++t) issue_w(t, rot(0) * 9 + t);
  if (gn_in) {
    for (int idx = tid; idx < Cin; idx += PNT) {
      const int pch = idx >= Cin / 2 ? 1 : 0, ch = (idx - pch * (Cin / 2)) * 2;
      const int seg = pch ? tl1.x : tl0.x;
      const float2 c0v = a.gn_coef[(size_t)seg * a.in_ld + goff + ch], c1v = a.gn_coef[(size_t)seg * a.in_ld + goff + ch + 1];
      *reinterpret_cast<float4*>(smem + COEF_OFF + (pch * Cin + ch) * 8) = make_float4(c0v.x, c1v.x, c0v.y, c1v.y);
    }
  }
  HQ_WAITV(0);
  if (gn_in) {
    HQ_WAITL();
    HQ_BAR();
#pragma unroll
    for (int g = 0; g < NPIECE; ++g) gn_piece(g, 0);
    HQ_WAITL();
  }
  HQ_BAR();
  ldfrag(0, 0, 0);

  // vmcnt at the top of phase q: the stage of phase q + 1 (issued in phase q - 2) must have landed before its k-step-0 fragments
  // are read in this phase's second half; the only group issued after it is the one of phase q - 1.
#define HQ_PHASE(t)                                 \
  HQ_WAITV(nload((t) - 1));                         \
  HQ_BAR();                                         \
  mma(0, cc, t);                                    \
  frags_wait(1);                                    \
  mma(1, cc, t);                                    \
  frags_wait(0);

  for (int cc = 0; cc < ncc; ++cc) {
    HQ_PHASE(0) HQ_PHASE(1) HQ_PHASE(2) HQ_PHASE(3) HQ_PHASE(4) HQ_PHASE(5) HQ_PHASE(6) HQ_PHASE(7) HQ_PHASE(8)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  HQ_WAITL();
  __syncthreads();

  // ---- fused epilogue: per patch two 64-row passes through an fp32 LDS tile ---------------------------------------
  float* const sC = reinterpret_cast<float*>(smem);
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(a.out);
  const int c8 = tid & 31, rr = tid >> 5;
  const int n0 = nt * 256 + c8 * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 s4v = a.scale ? reinterpret_cast<const float4*>(a.scale + n0)[h] : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 b4v = a.shift ? reinterpret_cast<const float4*>(a.shift + n0)[h] : make_float4(0.f, 0.f, 0.f, 0.f);
    sc[4 * h] = s4v.x; sc[4 * h + 1] = s4v.y; sc[4 * h + 2] = s4v.z; sc[4 * h + 3] = s4v.w;
    sh[4 * h] = b4v.x; sh[4 * h + 1] = b4v.y; sh[4 * h + 2] = b4v.z; sh[4 * h + 3] = b4v.w;
  }
  const bool relu = a.relu_nch > 0;
#pragma unroll
#ifdef HQ_NOEPI
  for (int pp = 0; pp < (acc[0][0][0] == 1234.5f ? 2 : 0); ++pp) {
#else
  for (int pp = 0; pp < 2; ++pp) {  // patch
#endif
    const SegDesc& sp = pp ? sd1 : sd0;
    const int ty = pp ? tl1.y : tl0.y;
    const int oy0 = ty >> 16, ox0 = ty & 0xffff;
    const int PW = sp.pw, NPOS = sp.ph * sp.pw;
    bf16_t* __restrict__ outn = out + (size_t)sp.out_row0 * a.out_ld + n0;
    // GroupNorm partial sums of this patch about a pivot every lane of a group shares (the conv bias of the group's first
    // channel: what makes |mean| >> sigma in practice), so that lanes and waves merge by plain additions
    float gn_n = 0.f, gn_s1 = 0.f, gn_s2 = 0.f;
    const float gn_pv = sh[0];
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {  // 64-row half of the patch
      if (pp + hp > 0) lds_barrier();
      if (wm == pp) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x16& c = acc[2 * hp + ii][j];
              *reinterpret_cast<float4*>(sC + (ii * 32 + l31) * SCP + wn * 128 + j * 32 + 8 * g + 4 * lh) =
                  make_float4(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
            }
      }
      lds_barrier();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int rl = rr + 8 * it;
        const int m = hp * 64 + rl;
        const int my = (int)(((unsigned)m * sp.inv_pw) >> 16);
        const int oy = oy0 + my, ox = ox0 + (m - my * PW);
        if (m < NPOS && oy < sp.out_H && ox < sp.out_W) {
          float v[8];
          const float4 lo = *reinterpret_cast<const float4*>(sC + rl * SCP + c8 * 8);
          const float4 hi = *reinterpret_cast<const float4*>(sC + rl * SCP + c8 * 8 + 4);
          v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
          if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          if (a.gn_partial) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[e] - gn_pv; gn_s1 += d; gn_s2 = fmaf(d, d, gn_s2); }
            gn_n += 8.f;
          }
#ifndef HQ_NOEPI
          store8<bf16_t>(outn + (size_t)(oy * sp.out_W + ox) * a.out_ld, v);
#else
          if (v[0] == 1234.5f) store8<bf16_t>(outn + (size_t)(oy * sp.out_W + ox) * a.out_ld, v);
#endif
        }
      }
    }
    if (a.gn_partial) {  // one (n, mean, M2) partial per patch and 8-channel group, merged in a fixed order
      gn_n += __shfl_xor(gn_n, 32);  // lanes c8 and c8 + 32 of a wave hold the same group (rows rr, rr + 1)
      gn_s1 += __shfl_xor(gn_s1, 32);
      gn_s2 += __shfl_xor(gn_s2, 32);
      lds_barrier();
      float* red = sC;  // [4 waves][32 groups][3]
      if (lane < 32) {
        red[(wave * 32 + c8) * 3 + 0] = gn_n;
        red[(wave * 32 + c8) * 3 + 1] = gn_s1;
        red[(wave * 32 + c8) * 3 + 2] = gn_s2;
      }
      lds_barrier();
      if (tid < 32) {
        float N = 0.f, S1 = 0.f, S2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          N += red[(w * 32 + c8) * 3 + 0]; S1 += red[(w * 32 + c8) * 3 + 1]; S2 += red[(w * 32 + c8) * 3 + 2];
        }
        const float inv_n = N > 0.f ? 1.f / N : 0.f;
        const float m2 = S2 - S1 * S1 * inv_n;
        float* gp = a.gn_partial + ((size_t)(2 * mt + pp) * (a.Cout >> 3) + (n0 >> 3)) * 3;
        gp[0] = N; gp[1] = gn_pv + S1 * inv_n; gp[2] = m2 > 0.f ? m2 : 0.f;
      }
    }
  }
}

bool conv_hq_ok(const ConvArgs&) { return true; }
int launch_conv_hq(const ConvArgs& a, hipStream_t s);  // same scope as conv_hpipe_ok

int launch_conv_hq(const ConvArgs& a, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_hq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -7;
    if (hipFuncSetAttribute((const void*)conv_hq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -7;
    attr_set = true;
  }
  const int chunk = (a.n_mtiles + 7) / 8;
  const int grid = 8 * chunk * a.n_ntiles;
  if (a.gn_coef) hipLaunchKernelGGL(conv_hq_kernel<true>, dim3(grid), dim3(PNT), LDS_BYTES, s, a);
  else hipLaunchKernelGGL(conv_hq_kernel<false>, dim3(grid), dim3(PNT), LDS_BYTES, s, a);
  return (int)hipGetLastError();
}

}  // namespace sylph
