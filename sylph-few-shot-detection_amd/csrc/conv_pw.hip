// Persistent, software-pipelined POINTWISE (1x1) convolution on MFMA for gfx950: the bottleneck conv1 / conv3 (+ residual,
// + projection shortcut as a second K range) of res3..res5 and the FPN laterals (+ nearest-2x top-down add).
//
//   out[pos][n] = relu?( (sum_c in[src(pos)][c] w[n][c] + sum_c in2[src2(pos)][c] w[n][Cin + c]) * scale[n] + shift[n] + res[rsrc(pos)][n] )
//
// Why a second kernel beside conv_igemm: these layers are short-K GEMMs (K = 128 .. 2048) with a heavy epilogue; most of them
// are HBM-bound (AI 57-205 FLOP/B).  conv_igemm walks K with ONE LDS stage per block (load slice -> wait -> barrier ->
// MFMA -> barrier) and relies on 4 co-resident blocks to hide the round trip of every slice and of the residual: measured
// 0.42-0.63 of the HBM peak on unfused traffic.  Here the memory pipeline never drains:
//
//   * grid = 2 blocks per CU, PERSISTENT: a block walks its XCD's tiles (N index innermost, so the N tiles of one M tile
//     run back to back on the same XCD and share the A rows through that XCD's L2).
//   * K is walked in 32-channel phases through a 3-stage LDS ring (stage = A tile [BM][64 B] + W tile [BN][64 B] = 24 KiB)
//     filled by global_load_lds (LDS-DMA, no VGPR round trip).  The loads of phase q + 2 are issued at the top of phase q
//     and retired with COUNTED s_waitcnt vmcnt: one barrier per phase, never vmcnt(0) in the steady state.  The ring runs
//     ACROSS tiles: the first two phases of the next tile are in flight during the last two phases and the whole epilogue of
//     the current one, so a tile costs one exposed round trip less than a stand-alone launch of it, and the epilogue's stores
//     (left in flight: vmcnt counts them, the waits of the next tile's first two phases account for them) overlap the next
//     tile's MFMAs.
//   * tile 128 x 256 (or 256 x 128 when Cout % 256 != 0), 4 waves as 2 x 2, wave tile 64 x 128 (128 x 64): 12 ds_read_b128
//     per 16 MFMAs, 128 accumulator VGPRs -- the operand ratio of conv_hpipe.
//   * weights are re-packed once per layer into stage images ([n tile][phase] -> one contiguous BN x 64 B block, swizzle
//     applied), so a W stage is a linear copy; the A rows carry the same swizzle on the source side.  Blocks start the K walk
//     at different phases (rotation by tile index): concurrently running blocks do not hammer the same 64 bytes of every
//     2-4 KiB activation row (L2 channel camping).
//   * epilogue in the accumulator (D^T) layout -- a lane owns 4 consecutive channels of a position: scale/shift from an LDS
//     table, the residual by 8-byte loads (a lane pair covers 16 contiguous bytes, the 8 loads of a row cover its 128-byte
//     line), ReLU, bf16 -- then a wave-private 32 x 64 LDS transpose so that every store instruction writes whole 128-byte
//     lines.  The staging area is the ring stage the last phase just released; all LDS traffic of the epilogue is inline asm
//     with explicit lgkmcnt waits (for a compiler-visible LDS access hipcc would first drain the LDS-DMA queue, i.e. wait for
//     the prefetched stages of the next tile).
//   * rows past the end of a segment read the segment's last row (results discarded) and store to a trash slot: the number
//     of stores per tile is a constant, which the counted waits rely on.
//
// Numerics: bf16 operands, fp32 accumulate, v = fma(acc, scale, shift) (+ residual) (ReLU) -> bf16: the same rounding points
// as conv_igemm's epilogue (oracle/bf16.py conv_epilogue).
// Reference ops replaced: the 1x1 convs of detectron2's BottleneckBlock and FPN at the call site
// sylph/modeling/meta_arch/meta_one_stage_detector.py:181,273.
#include <stdlib.h>

#include "common.h"

namespace sylph {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

namespace {

#define PW_FENCE __builtin_amdgcn_sched_barrier(0)
#define PW_BAR()                        \
  do {                                  \
    asm volatile("" ::: "memory");      \
    PW_FENCE;                           \
    __builtin_amdgcn_s_barrier();       \
    PW_FENCE;                           \
    asm volatile("" ::: "memory");      \
  } while (0)
#define PW_WAITV(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void pw_wait_vm(int n) {
  switch (n) {
#define PW_C(N) case N: PW_WAITV(N); break;
    PW_C(0) PW_C(1) PW_C(2) PW_C(3) PW_C(4) PW_C(5) PW_C(6) PW_C(7) PW_C(8) PW_C(9) PW_C(10) PW_C(11) PW_C(12) PW_C(13) PW_C(14) PW_C(15)
    PW_C(16) PW_C(17) PW_C(18) PW_C(19) PW_C(20) PW_C(21) PW_C(22) PW_C(23) PW_C(24) PW_C(25) PW_C(26) PW_C(27) PW_C(28) PW_C(29) PW_C(30) PW_C(31)
    PW_C(32) PW_C(33) PW_C(34) PW_C(35) PW_C(36) PW_C(37) PW_C(38) PW_C(39) PW_C(40) PW_C(41) PW_C(42) PW_C(43) PW_C(44) PW_C(45) PW_C(46) PW_C(47)
    PW_C(48) PW_C(49) PW_C(50) PW_C(51) PW_C(52) PW_C(53) PW_C(54) PW_C(55) PW_C(56) PW_C(57) PW_C(58) PW_C(59) PW_C(60) PW_C(61) PW_C(62)
#undef PW_C
    default: PW_WAITV(63); break;
  }
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int pw_lds_bytes(int BM, int BN, int NST) { return NST * (BM + BN) * 64 + 2 * (2 * BN * 4); }
}  // namespace

// BM x BN tile, NW waves (4: two blocks per CU; 8: the 256 x 256 tile, one block per CU), NST ring stages, RES: 0 none / 1 residual
// of the output geometry / 2 nearest-2x upsampled residual, RELU
template <int BM, int BN, int NW, int NST, int RES, bool RELU>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void conv_pw_kernel(const ConvArgs a) {
  constexpr int NT = NW * 64, WGN = NW / 2;   // waves as 2 (M) x WGN (N)
  constexpr int WTM = BM / 2, WTN = BN / WGN, TM = WTM / 32, TN = WTN / 32;
  constexpr int RPI = NT / 4;                 // rows one block-wide LDS-DMA instruction lands (64-byte rows, 16 bytes per lane)
  constexpr int AR = BM / RPI, BR = BN / RPI; // LDS-DMA instructions per lane and phase for A / W
  constexpr int IB = RPI * 64;                // bytes of one block-wide instruction
  constexpr int NLOAD = AR + BR;             // per lane and phase; phase 0 of a tile: + 1 (scale / shift table)
  constexpr int STAGE = (BM + BN) * 64, RING = NST * STAGE, TAB = 2 * BN * 4;
  constexpr int ASZ = BM * 64;
  constexpr int NJH = WTN / 64;              // 64-channel chunks across the wave tile
  constexpr int NCH = TM * NJH;              // 32-row x 64-channel epilogue chunks per wave
  constexpr int NSTORE = 4 * NCH;            // 16-byte stores per lane and tile
  // small tiles: the whole residual tile of the wave (8 bytes per lane and (row block, 8-channel group)) is fetched at the START
  // of the tile and has landed long before the epilogue; large tiles fetch it chunk by chunk inside the epilogue
  constexpr bool HOIST = RES != 0 && TM * TN * 4 <= 16;
  constexpr int NRES = HOIST ? TM * TN * 4 : 0;
  static_assert(TM * TN >= 4 && NW * 4096 <= STAGE && NST >= 3 && WTN % 64 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int r4 = tid >> 2, s4 = tid & 3;

  // ---- persistent tile walk: XCD x owns M tiles [x * chunk, (x + 1) * chunk); its blocks stride over (m_local, nt), nt innermost.
  // A cursor is (m_local, nt); stepping by the XCD's block count is two adds and a carry (no division in the loop).
  const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, nbl = gridDim.x >> 3;
  const int chunk = (a.n_mtiles + 7) >> 3;
  const int n_nt = a.n_ntiles;
  const int step_m = nbl / n_nt, step_n = nbl - step_m * n_nt;
  auto valid_at = [&](int m_local) { return m_local < chunk && xcd * chunk + m_local < a.n_mtiles; };
  // tile descriptor through the SCALAR cache (a vector load would be the youngest entry of the in-order vmcnt queue: waiting
  // for it drains every prefetched stage and every store in flight)
  auto load_desc = [&](int mt, i32x8& d0, i32x8& d1) {
    const PwDesc* p = a.pw_desc + mt;
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1) : "s"(p));
  };

  const char* const in1 = reinterpret_cast<const char*>(a.in);
  const char* const in2 = reinterpret_cast<const char*>(a.in2);
  const int nk1 = a.Cin >> 5, nk2 = a.in2 ? (a.Cin2 >> 5) : 0, nk = nk1 + nk2;
  const int rot_mask = a.pw_rot_mask;  // (largest power of two <= nk) - 1: a tile starts its K walk at phase (tile row + N tile) & mask

  // ---- loader: runs NST - 1 phases ahead of the MFMAs, across tile boundaries ------------------------------------------------
  int ld_m = bl / n_nt, ld_nt = bl - ld_m * n_nt;  // one division per block
  bool ld_valid = valid_at(ld_m);
  if (!ld_valid) return;
  int ld_q = 0, ld_rot = 0, ld_par = 0;
  unsigned ld_off1[AR], ld_off2[AR];  // byte offsets of this lane's A rows (chunk swizzle included) inside the tile's image of `in` / `in2`
  size_t ld_img1 = 0, ld_img2 = 0;    // ... and the image's first byte (wave-uniform, 64 bits: no limit on the batch for the inputs)
  auto loader_setup = [&]() {
    const int mt = xcd * chunk + ld_m;
    i32x8 d0, d1;
    load_desc(mt, d0, d1);
    const int row0 = d0[0], seg_rows = d0[1], out_W = d0[2], in_row0 = d0[4], in_W = d0[5], in2_row0 = d0[6], in2_W = d0[7];
    ld_rot = ((row0 / BM) + ld_nt) & rot_mask;  // keyed on the tile's place inside its own image: results do not depend on the batch around it
    const bool direct1 = a.stride == 1 && in_W == out_W, direct2 = a.stride2 == 1 && in2_W == out_W;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int pos = row0 + r4 + RPI * i;
      pos = pos < seg_rows ? pos : seg_rows - 1;  // rows past the segment: re-read its last row (their results are discarded)
      const int cl = s4 ^ ((r4 >> 2) & 3);       // source-side swizzle: LDS slot s4 of row r holds logical chunk cl
      int oy = 0, ox = 0;
      if (!direct1 || (in2 && !direct2)) { oy = pos / out_W; ox = pos - oy * out_W; }
      const int row1 = direct1 ? pos : oy * a.stride * in_W + ox * a.stride;
      ld_off1[i] = (unsigned)row1 * (unsigned)(a.in_ld * 2) + cl * 16;
      const int row2 = direct2 ? pos : oy * a.stride2 * in2_W + ox * a.stride2;
      ld_off2[i] = in2 ? (unsigned)row2 * (unsigned)(a.in2_ld * 2) + cl * 16 : 0u;
    }
    ld_img1 = (size_t)(unsigned)in_row0 * (size_t)(a.in_ld * 2);
    ld_img2 = (size_t)(unsigned)in2_row0 * (size_t)(a.in2_ld * 2);
  };
  loader_setup();
  const unsigned wvo = (unsigned)tid * 16u;
  // issue the loads of the loader's next phase into ring stage `stage`; returns the number of loads per lane (0: nothing left)
  auto issue_one = [&](int stage) -> int {
    if (!ld_valid) return 0;
    int kp = ld_q + ld_rot;
    kp = kp >= nk ? kp - nk : kp;
    char* dA = smem + stage * STAGE + wave * 1024;  // wave-uniform; lane l lands at +16 l
    char* dB = dA + ASZ;
    const bool second = kp >= nk1;
    const char* base = second ? in2 + ld_img2 + (size_t)(kp - nk1) * 64 : in1 + ld_img1 + (size_t)kp * 64;  // wave-uniform
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const unsigned off = second ? ld_off2[i] : ld_off1[i];
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + off), (lds_ptr_t)(dA + i * IB), 16, 0, 0);
    }
    const char* wsrc = reinterpret_cast<const char*>(a.wt) + ((size_t)ld_nt * nk + kp) * (size_t)(BN * 64);
#pragma unroll
    for (int j = 0; j < BR; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc + j * IB + wvo), (lds_ptr_t)(dB + j * IB), 16, 0, 0);
    int n = NLOAD;
    if (ld_q == 0) {  // scale | shift of the tile's N block -> table slot ld_par (TAB bytes; every wave copies a 1-KiB piece of it)
      const int piece = (wave * 1024) % TAB;
      const char* tsrc = reinterpret_cast<const char*>(a.pw_table) + (size_t)ld_nt * TAB + piece + lane * 16;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)tsrc, (lds_ptr_t)(smem + RING + ld_par * TAB + piece), 16, 0, 0);
      n = NLOAD + 1;
    }
    if (++ld_q == nk) {  // on to the next tile of this block
      ld_q = 0; ld_par ^= 1;
      ld_m += step_m; ld_nt += step_n;
      if (ld_nt >= n_nt) { ld_nt -= n_nt; ++ld_m; }
      ld_valid = valid_at(ld_m);
      if (ld_valid) loader_setup();
    }
    return n;
  };

  // ---- fragment addressing (constant per lane) ----------------------------------------------------------------------------
  const int swz = (l31 >> 2) & 3;
  const int slot0 = ((0 + lh) ^ swz) << 4, slot1 = ((2 + lh) ^ swz) << 4;
  const int rowA = (wm * WTM + l31) * 64, rowB = ASZ + (wn * WTN + l31) * 64;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  int cur_m = ld_m, cur_nt = ld_nt, cur_par = 0;  // compute cursor (the loader has just been set up on the same tile)
  // In-flight LDS-DMA groups, oldest first: per-lane load count and the number of OTHER vector-memory operations (stores,
  // residual loads) issued after the group -- vmcnt retires in order, so a wait for group k may leave exactly the younger
  // groups and those operations outstanding.
  int grp[NST - 1], oth[NST - 1];
#pragma unroll
  for (int k = 0; k < NST - 1; ++k) { grp[k] = issue_one(k); oth[k] = 0; }
  auto note_other = [&](int n) {
#pragma unroll
    for (int k = 0; k < NST - 1; ++k) oth[k] += n;
  };
  int ring = 0;           // ring stage of the phase about to be computed

  const char* const resb = reinterpret_cast<const char*>(a.res);
  char* const outb = reinterpret_cast<char*>(a.out);

  while (true) {
    // ---- tile prologue: descriptor, residual rows (and, small tiles, the residual itself) -----------------------------------
    const int cur_mt = xcd * chunk + cur_m;
    i32x8 d0, d1;
    load_desc(cur_mt, d0, d1);
    const int row0 = d0[0], seg_rows = d0[1], out_W = d0[2], out_row0 = d0[3], res_row0 = d1[0], res_W = d1[1];
    const int colw = cur_nt * BN + wn * WTN;  // first channel of this wave's column block
    unsigned roff[TM];                        // residual row of accumulator row (i, l31)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int pos = row0 + wm * WTM + i * 32 + l31;
      pos = pos < seg_rows ? pos : seg_rows - 1;
      int rp = pos;
      if (RES == 2) {
        const int oy = pos / out_W, ox = pos - oy * out_W;
        rp = (oy >> 1) * res_W + (ox >> 1);
      }
      roff[i] = RES ? (unsigned)(res_row0 + rp) * (unsigned)(a.res_ld * 2) + (unsigned)(colw + 4 * lh) * 2 : 0u;
    }
    auto load_res = [&](int c, u32x2* rv) {  // the 8 residual pieces of chunk c (wave-tile rows i, channels jh * 64 ..)
      const int i = c / NJH, jh = c - i * NJH;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          rv[jj * 4 + g] = *reinterpret_cast<const u32x2*>(resb + roff[i] + (jh * 64 + jj * 32 + 8 * g) * 2);
    };
    u32x2 rall[HOIST ? NCH * 8 : 8];
    if (HOIST) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) load_res(c, rall + c * 8);
      note_other(NRES);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int q = 0; q < nk; ++q) {
      // the oldest group must have landed; the younger groups and the operations issued after it may stay outstanding
      int younger = oth[0];
#pragma unroll
      for (int k = 1; k < NST - 1; ++k) younger += grp[k];
      pw_wait_vm(younger < 63 ? younger : 63);
      PW_BAR();  // everyone's part of stage `ring` has landed; everyone is done reading the stage refilled next
      {
        int st2 = ring + NST - 1; st2 = st2 >= NST ? st2 - NST : st2;
#pragma unroll
        for (int k = 0; k + 1 < NST - 1; ++k) { grp[k] = grp[k + 1]; oth[k] = oth[k + 1]; }
        grp[NST - 2] = issue_one(st2);
        oth[NST - 2] = 0;
      }
      const char* tS = smem + ring * STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int so = ks ? slot1 : slot0;
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(tS + rowA + i * 2048 + so);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(tS + rowB + j * 2048 + so);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);  // D^T
      }
      ring = ring + 1 == NST ? 0 : ring + 1;
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    // The stage of the last phase is free once every wave has passed this barrier; the next tile's first NST - 1 phases keep
    // landing in the other stages meanwhile.  All LDS traffic below is inline asm (see the file header).
    const int last = ring == 0 ? NST - 1 : ring - 1;
    const unsigned stg = lds0 + last * STAGE + wave * 4096;  // 32 rows x 128 B, 16-byte chunk c of row r at slot c ^ (r & 7)
    const unsigned tab = lds0 + RING + cur_par * TAB;         // float scale[BN], shift[BN] (landed with the tile's phase 0)
    PW_BAR();
    u32x2 rchunk[8];
    if (RES && !HOIST) { load_res(0, rchunk); note_other(8); }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = c / NJH, jh = c - i * NJH;
      const u32x2* rcur = HOIST ? rall + c * 8 : rchunk;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = jh * 2 + jj;
        // scale / shift of this lane's 16 channels of MFMA tile j: 8 table reads, one wait
        f32x4v sc4[4], sh4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int chw = wn * WTN + j * 32 + 8 * g + 4 * lh;  // channel inside the N tile
          asm volatile("ds_read_b128 %0, %1" : "=v"(sc4[g]) : "v"(tab + chw * 4));
          asm volatile("ds_read_b128 %0, %1" : "=v"(sh4[g]) : "v"(tab + (BN + chw) * 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc4[0]), "+v"(sc4[1]), "+v"(sc4[2]), "+v"(sc4[3]), "+v"(sh4[0]), "+v"(sh4[1]), "+v"(sh4[2]), "+v"(sh4[3]));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[i][j][4 * g + e], sc4[g][e], sh4[g][e]);
          if (RES) {
            const u32x2 rr = rcur[jj * 4 + g];
            v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
            v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
          }
          if (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          bf16x2 p0, p1;
          p0[0] = (bf16_t)v[0]; p0[1] = (bf16_t)v[1]; p1[0] = (bf16_t)v[2]; p1[1] = (bf16_t)v[3];
          const u32x2 pk = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
          asm volatile("ds_write_b64 %0, %1" ::"v"(stg + l31 * 128 + (((jj * 4 + g) ^ (l31 & 7)) << 4) + lh * 8), "v"(pk) : "memory");
        }
      }
      if (RES && !HOIST && c + 1 < NCH) { load_res(c + 1, rchunk); note_other(8); }  // next chunk's residual: in flight during the stores below
      // read the 32 x 64 chunk back row-major: 8 lanes cover a row's 128 bytes
      u32x4 o[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = (lane >> 3) + 8 * it;
        asm volatile("ds_read_b128 %0, %1" : "=v"(o[it]) : "v"(stg + row * 128 + (((lane & 7) ^ (row & 7)) << 4)));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int pos = row0 + wm * WTM + i * 32 + (lane >> 3) + 8 * it;
        char* dst = pos < seg_rows ? outb + ((size_t)(out_row0 + pos) * a.out_ld + colw + jh * 64 + (lane & 7) * 8) * 2
                                   : reinterpret_cast<char*>(a.trash) + tid * 16;
        *reinterpret_cast<u32x4*>(dst) = o[it];
      }
    }
    note_other(NSTORE);

    // next tile of this block
    cur_m += step_m; cur_nt += step_n;
    if (cur_nt >= n_nt) { cur_nt -= n_nt; ++cur_m; }
    cur_par ^= 1;
    if (!valid_at(cur_m)) break;
  }
}

// per-layer scale / shift table: [Cout / BN][scale BN | shift BN] fp32; nullptr scale -> 1, shift -> 0
__global__ void pw_pack_table_kernel(const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ out, int Cout, int BN) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (nt, which, n)
  if (i >= 2 * Cout) return;
  const int nt = i / (2 * BN), r = i - nt * 2 * BN, w = r >= BN ? 1 : 0, n = r - w * BN;
  float v = w ? 0.f : 1.f;
  if (w == 0 && scale) v = scale[nt * BN + n];
  if (w == 1 && shift) v = shift[nt * BN + n];
  out[i] = v;
}

int launch_pw_pack_table(const float* scale, const float* shift, float* out, int Cout, int BN, hipStream_t s) {
  const int n = 2 * Cout;
  hipLaunchKernelGGL(pw_pack_table_kernel, dim3((n + 255) / 256), dim3(256), 0, s, scale, shift, out, Cout, BN);
  return (int)hipGetLastError();
}

// [Cout][K] bf16 (conv_igemm layout of a 1x1 layer, K = Cin (+ Cin2)) -> [Cout / BN][K / 32][BN rows][4 slots][8]: slot s of
// row r holds 16-byte chunk s ^ ((r >> 2) & 3) of that row's 32-channel phase
__global__ void pw_pack_weights_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int Cout, int K, int BN) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk
  const size_t nchunks = (size_t)Cout * K / 8;
  if (i >= nchunks) return;
  const int nk = K >> 5;
  const int s = (int)(i & 3);
  const size_t rr = i >> 2;
  const int r = (int)(rr % BN);
  const size_t blk = rr / BN;
  const int kp = (int)(blk % nk), nt = (int)(blk / nk);
  const int chunk = s ^ ((r >> 2) & 3);
  const uint4 v = *reinterpret_cast<const uint4*>(w + (size_t)(nt * BN + r) * K + kp * 32 + chunk * 8);
  *reinterpret_cast<uint4*>(out + i * 8) = v;
}

int launch_pw_pack_weights(const void* w, void* out, int Cout, int K, int BN, hipStream_t s) {
  const size_t nchunks = (size_t)Cout * K / 8;
  hipLaunchKernelGGL(pw_pack_weights_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, s, (const bf16_t*)w, (bf16_t*)out, Cout, K, BN);
  return (int)hipGetLastError();
}

// Tile shape of a pointwise layer (false: not eligible): 128 x 256 (256 x 128 when Cout % 256 != 0), 3 ring stages -- the operand
// ratio of conv_hpipe (12 fragment reads per 16 MFMAs) and 6 LDS-DMA instructions per wave and phase.  The 128 x 128 / 4-stage
// variant (whole residual tile prefetched at the tile start, three activation phases in flight) is kept for A/B runs
// (SYLPH_PW_TILE=1, -DSYLPH_ABLATE builds only): measured 5-25 % slower on every layer of the R-50 graph -- per 32-channel phase a wave pays one barrier,
// one counted wait and its LDS-DMA issue slots (~100 cycles each) for only 8 MFMAs.  SYLPH_PW_TILE=3: 256 x 256 tile, 8 lock-step
// waves, one block per CU (A/B only: equal to the default within 3 %).
bool conv_pw_tile(int cout, int k_total, bool has_res, int* BM, int* BN) {
  if (cout % 128 != 0) return false;
  (void)k_total; (void)has_res;
#ifdef SYLPH_ABLATE
  static const int force = SYLPH_AB_ENV("SYLPH_PW_TILE", 0);
  if (force == 3 && cout % 256 == 0) { *BM = 256; *BN = 256; return true; }  // experiment: 256 x 256 tile, 8 waves, one block per CU
  if (force == 1) { *BM = 128; *BN = 128; return true; }
#endif
  *BN = cout % 256 == 0 ? 256 : 128;
  *BM = 384 - *BN;
  return true;
}

bool conv_pw_ok(DType dt, bool out_f32, const ConvArgs& a) {
  return dt == DT_BF16 && !out_f32 && a.Cout % 128 == 0 && a.KH == 1 && a.KW == 1 && a.pad == 0 && !a.stem && !a.halo && a.group_cout == 0 &&
         a.mul_nch == 0 && (a.relu_nch == 0 || a.relu_nch >= a.Cout) && !a.gn_partial && !a.gn_coef && a.Cin % 32 == 0 &&
         (!a.in2 || a.Cin2 % 32 == 0) && (a.Cin + (a.in2 ? a.Cin2 : 0)) >= 128 && (a.out_ld & 7) == 0 && (a.in_ld & 7) == 0 &&
         (!a.in2 || (a.in2_ld & 7) == 0) && (a.res_mode == 0 || (a.res_ld & 3) == 0) && a.res_mode >= 0 && a.res_mode <= 2 &&
         a.trash != nullptr && a.pw_desc != nullptr && a.pw_table != nullptr;
}

template <int BM, int BN, int NST, int NW = 4>
static int launch_pw_t(const ConvArgs& a, int grid, hipStream_t s) {
  const bool relu = a.relu_nch > 0;
  constexpr int lds = pw_lds_bytes(BM, BN, NST);
#define PW_GO(R, L)                                                                                                                \
  do {                                                                                                                             \
    static bool attr = false;                                                                                                      \
    if (!attr) {                                                                                                                   \
      if (hipFuncSetAttribute((const void*)conv_pw_kernel<BM, BN, NW, NST, R, L>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -7; \
      attr = true;                                                                                                                 \
    }                                                                                                                              \
    hipLaunchKernelGGL((conv_pw_kernel<BM, BN, NW, NST, R, L>), dim3(grid), dim3(NW * 64), lds, s, a);                                   \
    return (int)hipGetLastError();                                                                                                 \
  } while (0)
  if (a.res_mode == 0) { if (relu) PW_GO(0, true); else PW_GO(0, false); }
  if (a.res_mode == 1) { if (relu) PW_GO(1, true); else PW_GO(1, false); }
  if (relu) PW_GO(2, true); else PW_GO(2, false);
#undef PW_GO
}

int launch_conv_pw(const ConvArgs& a_in, int BM, int BN, hipStream_t s) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return -7;
    n_cu = p.multiProcessorCount;
  }
  ConvArgs a = a_in;
  const int nk = (a.Cin + (a.in2 ? a.Cin2 : 0)) >> 5;
  int p2 = 1;
  while (p2 * 2 <= nk) p2 *= 2;
  a.pw_rot_mask = p2 - 1;
  const long tiles = (long)a.n_mtiles * a.n_ntiles;
  int grid = (2 * n_cu + 7) & ~7;
  const long need = ((tiles + 7) / 8) * 8;
  if (need < grid) grid = (int)need;
#ifdef SYLPH_ABLATE
  static const int nst = SYLPH_AB_ENV("SYLPH_PW_NST", 4);  // tuning knob: ring depth of the small tile
  if (BM == 128 && BN == 128) {
    if (nst == 3) { grid = (int)((3L * n_cu + 7) & ~7L); if (need < grid) grid = (int)need; return launch_pw_t<128, 128, 3>(a, grid, s); }
    return launch_pw_t<128, 128, 4>(a, grid, s);
  }
  if (BM == 256 && BN == 256) {  // 8 waves, one block per CU
    grid = (n_cu + 7) & ~7;
    if (need < grid) grid = (int)need;
    return launch_pw_t<256, 256, 3, 8>(a, grid, s);
  }
#endif
  if (BM == 128 && BN == 256) return launch_pw_t<128, 256, 3>(a, grid, s);
  if (BM == 256 && BN == 128) return launch_pw_t<256, 128, 3>(a, grid, s);
  return -1;
}

}  // namespace sylph
