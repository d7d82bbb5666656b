// Streaming pointwise conv with a same-geometry residual (round 5): bottleneck conv3 of the identity blocks of res3..res5,
//
//     out[pos][n] = relu( (sum_c in[pos][c] w[n][c]) * scale[n] + shift[n] + res[pos][n] )
//
// (detectron2 BottleneckBlock.conv3 + shortcut add + ReLU at the call site sylph/modeling/meta_arch/meta_one_stage_detector.py:181,273).
// These layers move 2 output-sized tensors (residual in, result out) per short-K GEMM: 1.24 GB for 141 GFLOP at res4.  On conv_igemm
// they ran at 3.5-3.9 TB/s: the ablation builds of round 5 (residual loads and / or stores compiled out) showed a compute skeleton of
// 221 us under a 349-us launch -- the residual is only requested in the epilogue, one pass at a time, and the epilogue goes through
// an fp32 LDS tile.  conv_pw could not do better: its residual loads are 8-byte fragment loads (32 requests of 16 bytes per
// instruction), only one chunk ahead, and vmcnt retires in order, so a residual load issued early would stall every wait for a ring
// stage behind its HBM latency.
//
// Here the two streams are taken off the waves that wait for ring stages:
//
//   * ONE persistent 512-thread block per CU, eight waves (two per SIMD) as 2 x 4 with 64 x 64 wave tiles on a 128 x 256 tile; K in
//     32-channel phases through conv_pw's 3-stage LDS ring (LDS-DMA, counted vmcnt, one barrier per phase, the ring running across
//     tiles).  Only waves 0-3 issue -- and wait for -- ring loads: their vmcnt queue holds nothing else.
//   * waves 4-7 own the traffic of a 64-KiB LDS tile buffer: in phase 0 of tile T they store the finished tile T - 1 out of it
//     (16-byte stores, 8 lanes per 128-byte line) and then DMA the residual rows of tile T into it (whole lines, swizzle on the source
//     side: piece p of row r at slot p ^ (r & 7)); they never wait for a ring stage (the phase barrier tells them it landed), only,
//     at the end of the K loop, for their own queue.  64 KiB of loads and 64 KiB of stores are in flight per CU under every K loop.
//   * epilogue of all waves in the accumulator (D^T) layout: fma(acc, scale, shift) + the residual piece read from the buffer
//     (ds_read_b64) -> ReLU -> bf16 -> written back IN PLACE (ds_write_b64): the buffer the residual came in is the store staging.
//   * barriers per tile (all eight waves): one per phase, one when the K loop is done and the residual has landed (the S waves wait
//     vmcnt(0) first), one when the epilogue is done.
//
// Round 6: the same structure for three more pointwise layers whose weights fit the registers (K <= 512 per 256-channel N tile) and that
// ran on conv_pw at 3.2-4.1 TB/s: RESM = 0 (no residual: the buffer is store staging only) with a STRIDED input (stride-2 conv1 of the
// first res4 block) and / or a SECOND K range from a second, strided input (conv3 + projection shortcut of the first res3 block as one
// GEMM, K = 128 + 256); RESM = 2 (the residual is the nearest-2x upsampled coarser map: FPN lateral3 + top-down add).  Inputs are
// addressed with a 64-bit base per image + 32-bit offsets inside it.
//
// Numerics: the rounding points of conv_pw / conv_igemm (fp32 accumulate, fma(acc, scale, shift) + residual, ReLU, bf16).
#include <stdlib.h>

#include "common.h"

namespace sylph {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

namespace {

// Ablation switches (SPW_NOSTORE, SPW_NORES: measurement aids) exist only in -DSYLPH_ABLATE builds (tools/build_variant.sh)
#ifndef SYLPH_ABLATE
#undef SPW_NOSTORE
#undef SPW_NORES
#endif

#define SP_FENCE __builtin_amdgcn_sched_barrier(0)
#define SP_BAR()                        \
  do {                                  \
    asm volatile("" ::: "memory");      \
    SP_FENCE;                           \
    __builtin_amdgcn_s_barrier();       \
    SP_FENCE;                           \
    asm volatile("" ::: "memory");      \
  } while (0)
#define SP_WAITV(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

__device__ __forceinline__ void sp_wait_vm(int n) {
  switch (n) {
#define SP_C(N) case N: SP_WAITV(N); break;
    SP_C(0) SP_C(1) SP_C(2) SP_C(3) SP_C(4) SP_C(5) SP_C(6) SP_C(7) SP_C(8) SP_C(9) SP_C(10) SP_C(11) SP_C(12) SP_C(13) SP_C(14) SP_C(15)
    SP_C(16) SP_C(17) SP_C(18) SP_C(19) SP_C(20) SP_C(21) SP_C(22) SP_C(23) SP_C(24) SP_C(25) SP_C(26) SP_C(27) SP_C(28) SP_C(29) SP_C(30) SP_C(31)
#undef SP_C
    default: SP_WAITV(32); break;
  }
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int BM = 128, BN = 256, NST = 4;
constexpr int STAGE = BM * 128, RING = NST * STAGE, TAB = 2 * BN * 4;   // a stage = A rows of one 64-channel phase: [128][128 B]
constexpr int BUF_OFF = RING + TAB, BUF = BM * BN * 2;
constexpr int SPW_LDS = BUF_OFF + BUF;  // 133 120
static_assert(SPW_LDS <= 160 * 1024, "LDS budget");
}  // namespace

// K = input channels (128 / 256 / 384 / 512; a.Cin of them from `in`, the rest from `in2`): the weight fragments a wave holds are indexed
// by compile-time k-steps.  RESM: 0 no residual, 1 same-geometry residual, 2 nearest-2x upsampled residual.
template <int K, int RESM, bool RELU>
__global__ __launch_bounds__(512, 1) void conv_spw_kernel(const ConvArgs a) {
  constexpr int NK = K / 64;                  // 64-channel phases per tile
  constexpr int KS = K / 16;                  // MFMA k-steps = weight fragments per wave (4 VGPRs each)
  constexpr int UNR_OUT = K >= 384 ? 1 : 4, UNR_IN = K >= 384 ? 2 : 16;  // streamer loops: K = 512 leaves no registers to unroll them
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave < 4;               // waves 0-3 fill the A ring (their vmcnt queue holds ring loads only)
  const bool streamer = !loader;              // waves 4-7 stream the tile buffer (residual in, result out): never wait for a ring stage
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- tile walk: the blocks of an XCD are (M stride) x (N tile); a block keeps ITS N tile for the whole launch (its weights stay in
  // registers) and walks the M tiles of the XCD's chunk; the n_nt blocks that share an M tile run it at the same time (A through L2)
  const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, nbl = gridDim.x >> 3;
  const int chunk = (a.n_mtiles + 7) >> 3;
  const int n_nt = a.n_ntiles;
  const int nt = bl % n_nt, m_step = nbl / n_nt;
  auto valid_at = [&](int m_local) { return m_local < chunk && xcd * chunk + m_local < a.n_mtiles; };
  auto load_desc = [&](int mt, i32x8& d0, i32x8& d1) {
    const PwDesc* p = a.pw_desc + mt;
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1) : "s"(p));
  };
  int cur_m = bl / n_nt;
  if (m_step == 0 || !valid_at(cur_m)) return;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // ---- this wave's weights -> registers: output channels nt * 256 + wave * 32 + l31, k-step ks = channels 16 ks + 8 lh .. (+7) --------
  bf16x8 wf[KS];
  {
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(a.wt) + (size_t)(nt * BN + wave * 32 + l31) * K + lh * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 16);
  }
  {  // scale | shift of the block's N tile -> LDS, once
    const float* tsrc = a.pw_table + (size_t)nt * (2 * BN);
    reinterpret_cast<float*>(smem + RING)[tid] = tsrc[tid];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- A ring loader (waves 0-3): stage = [128 rows][128 B] of one 64-channel phase, piece p of row r at slot p ^ (r & 7); the ring
  // runs NST - 1 phases ahead of the MFMAs, across tile boundaries --------------------------------------------------------------------
  const int r8 = lane >> 3, s8 = lane & 7;
  const char* const in1 = reinterpret_cast<const char*>(a.in);
  const char* const in2 = reinterpret_cast<const char*>(a.in2);
  const int nk1 = a.in2 ? (a.Cin >> 6) : NK;  // phases read from `in`; the rest from `in2`
  int ld_m = cur_m, ld_q = 0;
  bool ld_valid = true;
  unsigned ld_off[4], ld_off2[4];     // byte offsets of this lane's rows inside the tile's image of `in` / `in2` (swizzle included)
  size_t ld_img1 = 0, ld_img2 = 0;    // the image's first byte (wave-uniform)
  auto loader_setup = [&]() {
    i32x8 d0, d1;
    load_desc(xcd * chunk + ld_m, d0, d1);
    const int row0 = d0[0], seg_rows = d0[1], out_W = d0[2], in_row0 = d0[4], in_W = d0[5], in2_row0 = d0[6], in2_W = d0[7];
    const bool direct1 = a.stride == 1 && in_W == out_W, direct2 = a.stride2 == 1 && in2_W == out_W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rc = wave * 32 + 8 * k + r8;
      int pos = row0 + rc;
      pos = pos < seg_rows ? pos : seg_rows - 1;
      int oy = 0, ox = 0;
      if (!direct1 || (in2 && !direct2)) { oy = pos / out_W; ox = pos - oy * out_W; }
      const int row1 = direct1 ? pos : oy * a.stride * in_W + ox * a.stride;
      ld_off[k] = (unsigned)row1 * (unsigned)(a.in_ld * 2) + (unsigned)((s8 ^ (rc & 7)) << 4);
      const int row2 = direct2 ? pos : oy * a.stride2 * in2_W + ox * a.stride2;
      ld_off2[k] = in2 ? (unsigned)row2 * (unsigned)(a.in2_ld * 2) + (unsigned)((s8 ^ (rc & 7)) << 4) : 0u;
    }
    ld_img1 = (size_t)(unsigned)in_row0 * (size_t)(a.in_ld * 2);
    ld_img2 = (size_t)(unsigned)in2_row0 * (size_t)(a.in2_ld * 2);
  };
  if (loader) loader_setup();
  auto issue_one = [&](int stage) -> int {
    if (!ld_valid) return 0;
    char* d = smem + stage * STAGE + wave * 4096;
    const bool second = ld_q >= nk1;
    const char* base = second ? in2 + ld_img2 + (size_t)(ld_q - nk1) * 128 : in1 + ld_img1 + (size_t)ld_q * 128;
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + (second ? ld_off2[k] : ld_off[k])), (lds_ptr_t)(d + k * 1024), 16, 0, 0);
    if (++ld_q == NK) {
      ld_q = 0;
      ld_m += m_step;
      ld_valid = valid_at(ld_m);
      if (ld_valid) loader_setup();
    }
    return 4;
  };
  int grp[NST - 1];
#pragma unroll
  for (int k = 0; k < NST - 1; ++k) grp[k] = loader ? issue_one(k) : 0;

  // ---- tile-buffer streamer (waves 4-7): S wave s streams the 64-channel column group s - 4 of the tile: [128 rows][128 B] = 16 KiB ----
  const char* const resb = reinterpret_cast<const char*>(a.res);
  char* const outb = reinterpret_cast<char*>(a.out);
  const int sg = wave - 4;
  const int scol = nt * BN + sg * 64;
  int p_row0 = 0, p_seg_rows = 0, p_out_row0 = 0;
  bool have_prev = false;
  auto out_prev = [&]() {  // the finished tile out of the buffer: 16-byte stores, 8 lanes per 128-byte line
    const unsigned region = lds0 + BUF_OFF + sg * 16384;
#pragma unroll UNR_OUT
    for (int c = 0; c < 4; ++c) {  // (K = 512: not unrolled, a wave holds 128 weight registers)
      u32x4 o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(o[k]) : "v"(region + c * 4096 + k * 1024 + lane * 16));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int rc = c * 32 + 8 * k + r8, pos = p_row0 + rc;
#ifdef SPW_NOSTORE
        if (pos < p_seg_rows && o[k][0] == 0x12345678u)
#else
        if (pos < p_seg_rows)
#endif
          *reinterpret_cast<u32x4*>(outb + ((size_t)(p_out_row0 + pos) * a.out_ld + scol + ((s8 ^ (rc & 7)) << 3)) * 2) = o[k];
      }
    }
  };
  auto res_in = [&](int row0, int seg_rows, int res_row0, int out_W, int res_W) {  // this tile's residual rows into the buffer (whole lines, source-side swizzle)
    char* const region_p = smem + BUF_OFF + sg * 16384;
#pragma unroll UNR_IN
    for (int k = 0; k < 16; ++k) {
      const int rc = 8 * k + r8;
      int pos = row0 + rc;
      pos = pos < seg_rows ? pos : seg_rows - 1;
      if (RESM == 2) {  // nearest-2x upsample of the coarser map (FPN top-down path)
        const int oy = pos / out_W, ox = pos - oy * out_W;
        pos = (oy >> 1) * res_W + (ox >> 1);
      }
      const char* src = resb + ((size_t)(res_row0 + pos) * a.res_ld + scol + ((s8 ^ (rc & 7)) << 3)) * 2;
#ifdef SPW_NORES
      if (src == nullptr) *reinterpret_cast<volatile int*>(region_p) = 0;
#else
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(region_p + k * 1024), 16, 0, 0);
#endif
    }
  };

  // ---- fragment addressing (constant per lane): row 32 i + l31 of the stage, piece 2 ks' + lh ---------------------------------------
  unsigned aoff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) aoff[ks] = (unsigned)(l31 * 128 + (((2 * ks + lh) ^ (l31 & 7)) << 4));
  const int jh = wave >> 1, hb = wave & 1;  // this wave's 32 columns inside the tile buffer: column group jh, pieces 4 hb .. 4 hb + 3
  const unsigned region = lds0 + BUF_OFF + jh * 16384 + l31 * 128 + lh * 8;
  const unsigned tab = lds0 + RING;
  int ring = 0;

  while (true) {
    int row0 = 0, seg_rows = 0, out_row0 = 0, res_row0 = 0, out_W = 1, res_W = 1;
    if (streamer) {
      i32x8 d0, d1;
      load_desc(xcd * chunk + cur_m, d0, d1);
      row0 = d0[0]; seg_rows = d0[1]; out_W = d0[2]; out_row0 = d0[3]; res_row0 = d1[0]; res_W = d1[1];
    }
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < NK; ++q) {
      if (loader) {
        int younger = 0;
#pragma unroll
        for (int k = 1; k < NST - 1; ++k) younger += grp[k];
        sp_wait_vm(younger);
      }
      SP_BAR();  // everyone's part of stage `ring` has landed; everyone is done reading the stage refilled next
      if (loader) {
        int st2 = ring + NST - 1; st2 = st2 >= NST ? st2 - NST : st2;
#pragma unroll
        for (int k = 0; k + 1 < NST - 1; ++k) grp[k] = grp[k + 1];
        grp[NST - 2] = issue_one(st2);
      } else if (q == 0) {
        // phase 0: every wave is past barrier B of the previous tile, the buffer holds its result: out with it, then this tile's
        // residual in (both under this tile's K loop)
        if (have_prev) out_prev();
        if (RESM != 0) res_in(row0, seg_rows, res_row0, out_W, res_W);
      }
      const unsigned tS = lds0 + ring * STAGE;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 fa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(smem + (tS - lds0) + i * 4096 + aoff[ks]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (q == 0 && ks == 0) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], fa[i], z, 0, 0, 0);
          } else {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q * 4 + ks], fa[i], acc[i], 0, 0, 0);  // D^T: a lane holds 4 consecutive channels
          }
        }
      }
      ring = ring + 1 == NST ? 0 : ring + 1;
    }
    if (streamer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's residual has landed (and the previous tile is out)
    SP_BAR();  // A: K loop done everywhere, residual in the buffer
    if constexpr (K <= 384) {
      f32x4v sc4[4], sh4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int chw = wave * 32 + 8 * g + 4 * lh;
        asm volatile("ds_read_b128 %0, %1" : "=v"(sc4[g]) : "v"(tab + chw * 4));
        asm volatile("ds_read_b128 %0, %1" : "=v"(sh4[g]) : "v"(tab + (BN + chw) * 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned rg = region + i * 4096;
        u32x2 rv[4] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
        if (RESM != 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) asm volatile("ds_read_b64 %0, %1" : "=v"(rv[g]) : "v"(rg + (((4 * hb + g) ^ (l31 & 7)) << 4)));
        }
        if (i == 0)
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(sc4[0]), "+v"(sc4[1]), "+v"(sc4[2]), "+v"(sc4[3]), "+v"(sh4[0]), "+v"(sh4[1]), "+v"(sh4[2]), "+v"(sh4[3]), "+v"(rv[0]),
                         "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]));
        else
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[i][4 * g + e], sc4[g][e], sh4[g][e]);
          const u32x2 rr = rv[g];
          v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
          v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
          if (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          bf16x2 p0, p1;
          p0[0] = (bf16_t)v[0]; p0[1] = (bf16_t)v[1]; p1[0] = (bf16_t)v[2]; p1[1] = (bf16_t)v[3];
          const u32x2 pk = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
          asm volatile("ds_write_b64 %0, %1" ::"v"(rg + (((4 * hb + g) ^ (l31 & 7)) << 4)), "v"(pk) : "memory");
        }
      }
    } else {
      // (scale / shift are re-read per 8-channel group: K = 512 keeps 128 weight registers, there is no room to hold all 32 values)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned rg = region + i * 4096;
        u32x2 rv[4] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
        if (RESM != 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) asm volatile("ds_read_b64 %0, %1" : "=v"(rv[g]) : "v"(rg + (((4 * hb + g) ^ (l31 & 7)) << 4)));
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4v sc4, sh4;
          const int chw = wave * 32 + 8 * g + 4 * lh;
          asm volatile("ds_read_b128 %0, %1" : "=v"(sc4) : "v"(tab + chw * 4));
          asm volatile("ds_read_b128 %0, %1" : "=v"(sh4) : "v"(tab + (BN + chw) * 4));
          if (g == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc4), "+v"(sh4), "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]));
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc4), "+v"(sh4));
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[i][4 * g + e], sc4[e], sh4[e]);
          const u32x2 rr = rv[g];
          v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
          v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
          if (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          bf16x2 p0, p1;
          p0[0] = (bf16_t)v[0]; p0[1] = (bf16_t)v[1]; p1[0] = (bf16_t)v[2]; p1[1] = (bf16_t)v[3];
          const u32x2 pk = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
          asm volatile("ds_write_b64 %0, %1" ::"v"(rg + (((4 * hb + g) ^ (l31 & 7)) << 4)), "v"(pk) : "memory");
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SP_BAR();  // B: the buffer holds the finished tile
    if (streamer) { p_row0 = row0; p_seg_rows = seg_rows; p_out_row0 = out_row0; have_prev = true; }
    cur_m += m_step;
    if (!valid_at(cur_m)) break;
  }
  if (streamer) out_prev();
}

bool conv_spw_ok(DType dt, bool out_f32, const ConvArgs& a) {
  if (!(dt == DT_BF16 && !out_f32 && a.Cout % 256 == 0 && a.KH == 1 && a.KW == 1 && a.pad == 0 && !a.stem && !a.halo && a.group_cout == 0 && a.mul_nch == 0 &&
        (a.relu_nch == 0 || a.relu_nch >= a.Cout) && !a.gn_partial && !a.gn_coef && a.n_ntiles <= 8 && (a.out_ld & 7) == 0 && (a.in_ld & 7) == 0 &&
        a.pw_desc != nullptr && a.pw_table != nullptr))
    return false;
  const int K = a.Cin + (a.in2 ? a.Cin2 : 0);
  if (a.res_mode == 1)  // conv3 + residual of the identity blocks (round 5)
    return !a.in2 && a.stride == 1 && (K == 128 || K == 256 || K == 512) && a.res != nullptr && (a.res_ld & 7) == 0;
  if (a.res_mode == 2)  // FPN lateral + top-down add
    return !a.in2 && a.stride == 1 && K == 512 && a.relu_nch == 0 && a.res != nullptr && (a.res_ld & 7) == 0;
  if (a.in2)            // conv3 + projection shortcut of the first res3 block: K = 128 + 256, ReLU
    return a.Cin == 128 && a.Cin2 == 256 && a.stride == 1 && a.relu_nch > 0 && (a.in2_ld & 7) == 0;
  return K == 512 && a.relu_nch > 0;  // strided conv1 of the first res4 block
}

template <int K, int RESM, bool RELU>
static int launch_spw_k(const ConvArgs& a, int grid, hipStream_t s) {
  static PerDeviceOnce once;
  if (!once.run(current_device(), [] {
        return hipFuncSetAttribute((const void*)conv_spw_kernel<K, RESM, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, SPW_LDS) == hipSuccess;
      }))
    return -7;
  hipLaunchKernelGGL((conv_spw_kernel<K, RESM, RELU>), dim3(grid), dim3(512), SPW_LDS, s, a);
  return (int)hipGetLastError();
}

// a.wt: the layer's weights in the conv_igemm layout [Cout][K] (read once per block into registers); a.pw_table, a.pw_desc as conv_pw
int launch_conv_spw(const ConvArgs& a, hipStream_t s) {
  const int n_cu = device_cu_count(current_device());
  // blocks of an XCD = (M stride) x (N tiles): a multiple of 8 * n_ntiles, at most one block per CU
  const int per = 8 * a.n_ntiles;
  int grid = (n_cu / per) * per;
  if (grid == 0) return -1;
  const int K = a.Cin + (a.in2 ? a.Cin2 : 0);
  const bool relu = a.relu_nch > 0;
  if (a.res_mode == 1) {
    if (K == 128) return relu ? launch_spw_k<128, 1, true>(a, grid, s) : launch_spw_k<128, 1, false>(a, grid, s);
    if (K == 256) return relu ? launch_spw_k<256, 1, true>(a, grid, s) : launch_spw_k<256, 1, false>(a, grid, s);
    if (K == 512) return relu ? launch_spw_k<512, 1, true>(a, grid, s) : launch_spw_k<512, 1, false>(a, grid, s);
  } else if (a.res_mode == 2) {
    if (K == 512 && !relu) return launch_spw_k<512, 2, false>(a, grid, s);
  } else {
    if (K == 384 && relu) return launch_spw_k<384, 0, true>(a, grid, s);
    if (K == 512 && relu) return launch_spw_k<512, 0, true>(a, grid, s);
  }
  return -1;
}

}  // namespace sylph
