// Dual-output pointwise kernel for gfx950: the LAST 1x1 conv of one ResNet identity bottleneck (conv3 + FrozenBN + residual +
// ReLU) and the FIRST 1x1 conv of the next one (conv1' + FrozenBN + ReLU) in ONE pass over the rows:
//
//   y [pos][c]  = relu( (sum_k t2[pos][k] w3[c][k]) * s3[c] + b3[c] + x[pos][c] )          c < C    (written to HBM: the residual of the next block)
//   t1'[pos][m] = relu( (sum_c bf16(y[pos][c]) w1'[m][c]) * s1[m] + b1[m] )                  m < MID  (the next block's conv1 output)
//
// Why: the unfused pair reads y back from HBM for conv1' -- C * 2 bytes per position, 25 % of the traffic of a res3 / res4
// identity block (the pointwise + 3x3 bottleneck launches move 588 MB / image).  Here the y rows a wave has just
// produced become the B operand of the second GEMM STRAIGHT FROM ITS REGISTERS:
//
//   * a block owns 128 rows; wave w owns rows [32 w, 32 w + 32) for BOTH GEMMs, so no row of y ever crosses a wave.
//   * GEMM 1 runs over 128-channel chunks of C.  In the accumulator (D^T) layout a lane holds, for position l31, the channels
//     32 j + 8 g + 4 lh + e (j: MFMA tile, g, e < 4).  After the epilogue (FrozenBN, residual, ReLU, bf16 -- the value that is
//     stored) the 8 bf16 of (g = 2 s, 2 s + 1) ARE a valid B fragment of v_mfma_f32_32x32x16_bf16 for k-step s of tile j, with the
//     k slots permuted; conv1's weights are packed once with the same permutation (dual_pack_w1_kernel).  GEMM 2 accumulates
//     acc2[32 rows][MID] over the chunks in registers (MID / 2 VGPRs per lane) and is written once per row tile.
//   * both weight matrices stream from L2 through the LDS ring of conv_pw.hip (global_load_lds, counted vmcnt, one barrier per
//     phase, the ring running across chunks and row tiles); the t2 rows ride in the same stages.  The epilogue's LDS transpose
//     (whole 128-byte lines per store instruction) has its own staging area, so it needs no barrier.
//
// Numerics: the rounding points of the unfused pair (conv_pw / conv_igemm epilogue: v = fma(acc, scale, shift) (+ residual) (ReLU)
// -> bf16; the second GEMM consumes exactly the bf16 that is stored), fp32 accumulation.  Only the summation order of conv1'
// differs (128-channel chunks, permuted inside 16-channel groups): fp32 rounding noise below one bf16 ulp of the result.
// Reference ops replaced: detectron2 BottleneckBlock.conv3 (+ shortcut add + relu) of block i and BottleneckBlock.conv1 of block
// i + 1 at the call site sylph/modeling/meta_arch/meta_one_stage_detector.py:181,273.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace sylph {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

namespace {

#define DU_FENCE __builtin_amdgcn_sched_barrier(0)
#define DU_BAR()                        \
  do {                                  \
    asm volatile("" ::: "memory");      \
    DU_FENCE;                           \
    __builtin_amdgcn_s_barrier();       \
    DU_FENCE;                           \
    asm volatile("" ::: "memory");      \
  } while (0)
#define DU_WAITV(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

__device__ __forceinline__ void du_wait_vm(int n) {
  switch (n) {
#define DU_C(N) case N: DU_WAITV(N); break;
    DU_C(0) DU_C(1) DU_C(2) DU_C(3) DU_C(4) DU_C(5) DU_C(6) DU_C(7) DU_C(8) DU_C(9) DU_C(10) DU_C(11) DU_C(12) DU_C(13) DU_C(14) DU_C(15)
    DU_C(16) DU_C(17) DU_C(18) DU_C(19) DU_C(20) DU_C(21) DU_C(22) DU_C(23) DU_C(24) DU_C(25) DU_C(26) DU_C(27) DU_C(28) DU_C(29) DU_C(30) DU_C(31)
    DU_C(32) DU_C(33) DU_C(34) DU_C(35) DU_C(36) DU_C(37) DU_C(38) DU_C(39) DU_C(40) DU_C(41) DU_C(42) DU_C(43) DU_C(44) DU_C(45) DU_C(46) DU_C(47)
    DU_C(48) DU_C(49) DU_C(50) DU_C(51) DU_C(52) DU_C(53) DU_C(54) DU_C(55) DU_C(56) DU_C(57) DU_C(58) DU_C(59) DU_C(60) DU_C(61) DU_C(62)
#undef DU_C
    default: DU_WAITV(63); break;
  }
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// row swizzle of a stage: 16-byte chunk c of row r sits at slot c ^ swz(r); 64-byte rows (4 chunks) / 128-byte rows (8 chunks)
template <int CPR> __host__ __device__ __forceinline__ int du_swz(int r) { return CPR == 4 ? (r >> 2) & 3 : r & 7; }

constexpr int du_max(int a, int b) { return a > b ? a : b; }
constexpr int du_stage(int MID, int KP) { return du_max(256 * KP * 2, MID * KP * 2); }
constexpr int du_lds_bytes(int MID, int KP, int NST) { return NST * du_stage(MID, KP) + 16384 + 2 * 1024 + 2 * MID * 4; }

}  // namespace

// MID: channels of t2 / t1' (128: res3, 256: res4); KP: channels per ring phase (32 / 64); NST: ring stages; BPC: blocks per CU
template <int MID, int KP, int NST, int BPC>
__global__ __launch_bounds__(256, BPC) void conv_dual_kernel(const DualArgs a) {
  constexpr int BM = 128, BN = 128, NT = 256;
  constexpr int ROWB = KP * 2;               // bytes of one row of one phase
  constexpr int CPR = ROWB / 16;             // 16-byte chunks per row
  constexpr int RPI = NT / CPR;              // rows one block-wide LDS-DMA instruction lands
  constexpr int IB = NT * 16;                // bytes of one block-wide instruction
  constexpr int AR = BM / RPI, BR = BN / RPI, WR = MID / RPI;  // instructions per lane: t2 rows, w3 rows (GEMM 1), w1' rows (GEMM 2)
  constexpr int ASZ = BM * ROWB;
  constexpr int STAGE = du_stage(MID, KP), RING = NST * STAGE;
  constexpr int STG = RING;                  // epilogue staging: 4 waves x 4 KiB
  constexpr int TAB3 = 2 * BN * 4;           // scale | shift of one 128-channel chunk (double-buffered)
  constexpr int T3O = STG + 16384, T1O = T3O + 2 * TAB3;
  constexpr int NK1 = MID / KP, NK2 = BN / KP, NPH = NK1 + NK2;  // phases per chunk: GEMM 1, GEMM 2
  constexpr int KS = KP / 16;                // MFMA k-steps per phase
  constexpr int TN = BN / 32;                // MFMA tiles across a chunk
  constexpr int TM2 = MID / 32;              // MFMA tiles across t1'
  constexpr int JP = KP / 32;                // y tiles consumed per GEMM-2 phase
  static_assert((NK1 & (NK1 - 1)) == 0 && MID % 64 == 0 && (KP == 32 || KP == 64) && NST >= 3, "shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
#ifdef SYLPH_ABLATE
  const int ab = a.ablate;
#else
  constexpr int ab = 0;
#endif
  const int rI = tid / CPR, sI = tid % CPR;  // row / slot of this lane inside a block-wide LDS-DMA instruction

  // ---- persistent row-tile walk: XCD x owns tiles [x * chunk, (x + 1) * chunk); its blocks stride over them ----------------
  const int xcd = blockIdx.x & 7, bl = blockIdx.x >> 3, nbl = gridDim.x >> 3;
  const int chunk = (a.n_mtiles + 7) >> 3;
  const int n_ch = a.C >> 7;  // 128-channel chunks of C
  auto valid_at = [&](int m_local) { return m_local < chunk && xcd * chunk + m_local < a.n_mtiles; };
  auto load_desc = [&](int mt, i32x8& d0, i32x8& d1) {  // through the scalar cache (see conv_pw.hip)
    const PwDesc* p = a.desc + mt;
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1) : "s"(p));
  };

  // conv1' scale | shift: once per block, plain loads (before any LDS-DMA is in flight)
  {
    float* t1 = reinterpret_cast<float*>(smem + T1O);
    for (int i = tid; i < 2 * MID; i += NT) t1[i] = a.tab1[i];
  }
  __syncthreads();

  const char* const in1 = reinterpret_cast<const char*>(a.in);

  // ---- loader: NST - 1 phases ahead of the MFMAs, across chunks and row tiles -------------------------------------------------
  int ld_m = bl, ld_nt = 0, ld_q = 0, ld_par = 0, ld_row_key = 0;
  bool ld_valid = valid_at(ld_m);
  if (!ld_valid) return;
  unsigned ld_off[AR];  // byte offsets of this lane's t2 rows (chunk swizzle included)
  auto loader_setup = [&]() {
    i32x8 d0, d1;
    load_desc(xcd * chunk + ld_m, d0, d1);
    const int row0 = d0[0], seg_rows = d0[1], in_row0 = d0[4];
    ld_row_key = row0 / BM;  // the K rotation is keyed on the tile's place inside its own image: results do not depend on the batch
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int pos = row0 + rI + RPI * i;
      pos = pos < seg_rows ? pos : seg_rows - 1;  // rows past the segment re-read its last row (results discarded)
      const int cl = sI ^ du_swz<CPR>(rI + RPI * i);
      ld_off[i] = (unsigned)(in_row0 + pos) * (unsigned)(a.in_ld * 2) + cl * 16;
    }
  };
  loader_setup();
  const unsigned wvo = (unsigned)tid * 16u;
  auto issue_one = [&](int stage) -> int {
    if (!ld_valid) return 0;
    char* dst = smem + stage * STAGE + wave * 1024;  // wave-uniform; lane l lands at +16 l
    int n;
    if (ld_q < NK1) {
      const int kp = (ld_q + ld_row_key + ld_nt) & (NK1 - 1);
      const char* base = in1 + (size_t)kp * ROWB;
#pragma unroll
      for (int i = 0; i < AR; ++i)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + ld_off[i]), (lds_ptr_t)(dst + i * IB), 16, 0, 0);
      const char* wsrc = reinterpret_cast<const char*>(a.w3) + ((size_t)ld_nt * NK1 + kp) * (size_t)(BN * ROWB);
#pragma unroll
      for (int j = 0; j < BR; ++j)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc + j * IB + wvo), (lds_ptr_t)(dst + ASZ + j * IB), 16, 0, 0);
      n = AR + BR;
      if (ld_q == 0) {  // scale | shift of the chunk (1 KiB; every wave copies it: the per-lane load count stays uniform)
        const char* tsrc = reinterpret_cast<const char*>(a.tab3) + (size_t)ld_nt * TAB3 + lane * 16;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)tsrc, (lds_ptr_t)(smem + T3O + ld_par * TAB3), 16, 0, 0);
        n = AR + BR + 1;
      }
    } else {
      const int p = ld_q - NK1;
      const char* wsrc = reinterpret_cast<const char*>(a.w1) + ((size_t)ld_nt * NK2 + p) * (size_t)(MID * ROWB);
#pragma unroll
      for (int j = 0; j < WR; ++j)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc + j * IB + wvo), (lds_ptr_t)(dst + j * IB), 16, 0, 0);
      n = WR;
    }
    if (++ld_q == NPH) {
      ld_q = 0; ld_par ^= 1;
      if (++ld_nt == n_ch) {
        ld_nt = 0;
        ld_m += nbl;
        ld_valid = valid_at(ld_m);
        if (ld_valid) loader_setup();
      }
    }
    return n;
  };

  // ---- fragment addressing (constant per lane) --------------------------------------------------------------------------------
  const int swz = du_swz<CPR>(l31);
  const int rowA = (wave * 32 + l31) * ROWB, rowB = ASZ + l31 * ROWB, rowW = l31 * ROWB;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned stg = lds0 + STG + wave * 4096;  // this wave's 32 rows x 128 B; chunk c of row r at slot c ^ (r & 7)
  const unsigned tab1 = lds0 + T1O;

  int grp[NST - 1], oth[NST - 1];  // in-flight LDS-DMA groups, oldest first: loads per lane / other vector-memory ops issued after them
#pragma unroll
  for (int k = 0; k < NST - 1; ++k) { grp[k] = issue_one(k); oth[k] = 0; }
  int res_younger = 0;  // vector-memory operations issued after the residual loads in flight (their counted wait, see load_res)
  auto note_other = [&](int n) {
#pragma unroll
    for (int k = 0; k < NST - 1; ++k) oth[k] += n;
    res_younger += n;
  };
  int ring = 0;
  // top of a phase: the oldest group has landed (younger groups and the operations issued after it may stay outstanding), everyone
  // is done with the stage that is refilled next
  auto phase_top = [&]() {
    int younger = oth[0];
#pragma unroll
    for (int k = 1; k < NST - 1; ++k) younger += grp[k];
    du_wait_vm(younger < 63 ? younger : 63);
    DU_BAR();
    int st2 = ring + NST - 1; st2 = st2 >= NST ? st2 - NST : st2;
#pragma unroll
    for (int k = 0; k + 1 < NST - 1; ++k) { grp[k] = grp[k + 1]; oth[k] = oth[k + 1]; }
    grp[NST - 2] = issue_one(st2);
    oth[NST - 2] = 0;
    res_younger += grp[NST - 2];
  };

  const char* const resb = reinterpret_cast<const char*>(a.res);
  char* const outb = reinterpret_cast<char*>(a.out);
  char* const out2b = reinterpret_cast<char*>(a.out2);
  int cur_m = bl, cur_par = 0;

  // The residual x[rows of this wave][128-channel chunk] lives in registers (16 x 8 bytes per lane) and is fetched ONE CHUNK AHEAD:
  // the loads of chunk c + 1 (of the next row tile after the last chunk) are issued at the top of chunk c's first GEMM-2 phase, right
  // behind that phase's ring group -- vmcnt retires in order, so the first wait that forces them home is the one of the group issued
  // one phase later, NST phases on.  The loads are inline asm into ACCUMULATION registers and retired by a counted wait of our own
  // (wait_res): a compiler-visible load would be waited for with vmcnt(0), i.e. by draining the whole ring every chunk.
  u32x2 rres[16];
  auto res_row_off = [&](int m_local) -> unsigned {  // byte offset of this lane's accumulator row in the residual buffer
    i32x8 e0, e1;
    load_desc(xcd * chunk + m_local, e0, e1);
    int rp = e0[0] + wave * 32 + l31;
    rp = rp < e0[1] ? rp : e0[1] - 1;
    return (unsigned)(e1[0] + rp) * (unsigned)(a.res_ld * 2) + (unsigned)(4 * lh) * 2;
  };
  auto load_res = [&](unsigned off, int colw) {  // this lane's 16 pieces of channels [colw, colw + 128) of its row
    if (ab & 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) rres[k] = u32x2{0u, 0u};
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned vo = off + (unsigned)((colw + j * 32 + 8 * g) * 2);
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=a"(rres[j * 4 + g]) : "v"(vo), "s"(resb) : "memory");
      }
    note_other(16);
    res_younger = 0;
  };
  auto wait_res = [&]() {  // everything up to and including the residual loads has landed; younger operations stay in flight
    if (ab & 1) return;
    du_wait_vm(res_younger < 63 ? res_younger : 63);
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+a"(rres[k]));
  };
  load_res(res_row_off(cur_m), 0);

  while (true) {
    i32x8 d0, d1;
    load_desc(xcd * chunk + cur_m, d0, d1);
    const int row0 = d0[0], seg_rows = d0[1], out_row0 = d0[3], res_row0 = d1[0];
    int rpos = row0 + wave * 32 + l31;  // accumulator row of this lane
    rpos = rpos < seg_rows ? rpos : seg_rows - 1;
    const unsigned roff = (unsigned)(res_row0 + rpos) * (unsigned)(a.res_ld * 2) + (unsigned)(4 * lh) * 2;
    const bool more_tiles = valid_at(cur_m + nbl);

    f32x16 acc2[TM2];
#pragma unroll
    for (int m = 0; m < TM2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;

    for (int nt = 0; nt < n_ch; ++nt) {
      const int colw = nt * BN;
      // ---- GEMM 1: acc1[j] = w3[chunk rows 32 j ..][:] . t2[rows of this wave][:]^T ---------------------------------------------
      f32x16 acc1[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
#pragma unroll 1
      for (int q = 0; q < NK1; ++q) {
        phase_top();
        const char* tS = smem + ring * STAGE;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int so = ((2 * ks + lh) ^ swz) << 4;
          bf16x8 fb[TN];
          const bf16x8 fa = *reinterpret_cast<const bf16x8*>(tS + rowA + so);
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(tS + rowB + j * 32 * ROWB + so);
          if (!(ab & 8))
#pragma unroll
          for (int j = 0; j < TN; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa, acc1[j], 0, 0, 0);  // D^T
        }
        ring = ring + 1 == NST ? 0 : ring + 1;
      }

      // ---- epilogue 1: FrozenBN + residual + ReLU -> bf16: y to HBM (whole lines through the staging area) and, as packed
      // registers, the B operand of GEMM 2.  All LDS traffic is inline asm with explicit lgkmcnt (see conv_pw.hip).
      const unsigned tab = lds0 + T3O + cur_par * TAB3;
      u32x2 ypk[TN][4];
      wait_res();
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = jh * 2 + jj;
#pragma unroll
          for (int gh = 0; gh < 2; ++gh) {  // scale / shift of 8 of this lane's 16 channels of tile j at a time (register budget)
          f32x4v sc4[2], sh4[2];
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) {
            const int chw = j * 32 + 8 * (gh * 2 + g2) + 4 * lh;
            asm volatile("ds_read_b128 %0, %1" : "=v"(sc4[g2]) : "v"(tab + chw * 4));
            asm volatile("ds_read_b128 %0, %1" : "=v"(sh4[g2]) : "v"(tab + (BN + chw) * 4));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc4[0]), "+v"(sc4[1]), "+v"(sh4[0]), "+v"(sh4[1]));
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) {
            const int g = gh * 2 + g2;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc1[j][4 * g + e], sc4[g2][e], sh4[g2][e]);
            const u32x2 rr = rres[j * 4 + g];
            v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
            v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            bf16x2 p0, p1;
            p0[0] = (bf16_t)v[0]; p0[1] = (bf16_t)v[1]; p1[0] = (bf16_t)v[2]; p1[1] = (bf16_t)v[3];
            const u32x2 pk = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
            ypk[j][g] = pk;
            asm volatile("ds_write_b64 %0, %1" ::"v"(stg + l31 * 128 + (((jj * 4 + g) ^ (l31 & 7)) << 4) + lh * 8), "v"(pk) : "memory");
          }
          }
        }
        u32x4 o[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = (lane >> 3) + 8 * it;
          asm volatile("ds_read_b128 %0, %1" : "=v"(o[it]) : "v"(stg + row * 128 + (((lane & 7) ^ (row & 7)) << 4)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int pos = row0 + wave * 32 + (lane >> 3) + 8 * it;
          char* dst = pos < seg_rows ? outb + ((size_t)(out_row0 + pos) * a.out_ld + colw + jh * 64 + (lane & 7) * 8) * 2
                                     : reinterpret_cast<char*>(a.trash) + tid * 16;
          if (!(ab & 2)) *reinterpret_cast<u32x4*>(dst) = o[it];
        }
        if (!(ab & 2)) note_other(4);
      }

      // ---- GEMM 2: acc2[m] += w1'[rows 32 m ..][chunk channels] . y[rows of this wave][chunk channels]^T, y from registers ----------
#pragma unroll
      for (int p = 0; p < NK2; ++p) {
        phase_top();
        if (p == 0) {  // next chunk's residual (registers free since the epilogue above), behind this phase's ring group
          if (nt + 1 < n_ch) load_res(roff, colw + BN);
          else if (more_tiles) load_res(res_row_off(cur_m + nbl), 0);
        }
        const char* tS = smem + ring * STAGE;
#pragma unroll
        for (int jl = 0; jl < JP; ++jl) {
          const int j = p * JP + jl;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const u32x4 yv = {ypk[j][2 * s][0], ypk[j][2 * s][1], ypk[j][2 * s + 1][0], ypk[j][2 * s + 1][1]};
            const bf16x8 yf = __builtin_bit_cast(bf16x8, yv);
            const int so = ((2 * (jl * 2 + s) + lh) ^ swz) << 4;
            bf16x8 fw[TM2];
#pragma unroll
            for (int m = 0; m < TM2; ++m) fw[m] = *reinterpret_cast<const bf16x8*>(tS + rowW + m * 32 * ROWB + so);
            if (!(ab & 4))
#pragma unroll
            for (int m = 0; m < TM2; ++m) acc2[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[m], yf, acc2[m], 0, 0, 0);
          }
        }
        ring = ring + 1 == NST ? 0 : ring + 1;
      }
      cur_par ^= 1;
    }

    // ---- epilogue 2: t1' = relu(fma(acc2, s1, b1)) -> bf16, whole lines ------------------------------------------------------------
#pragma unroll
    for (int c = 0; c < TM2 / 2; ++c) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int m = c * 2 + jj;
#pragma unroll
        for (int gh = 0; gh < 2; ++gh) {
        f32x4v sc4[2], sh4[2];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const int chw = m * 32 + 8 * (gh * 2 + g2) + 4 * lh;
          asm volatile("ds_read_b128 %0, %1" : "=v"(sc4[g2]) : "v"(tab1 + chw * 4));
          asm volatile("ds_read_b128 %0, %1" : "=v"(sh4[g2]) : "v"(tab1 + (MID + chw) * 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc4[0]), "+v"(sc4[1]), "+v"(sh4[0]), "+v"(sh4[1]));
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const int g = gh * 2 + g2;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = __builtin_fmaf(acc2[m][4 * g + e], sc4[g2][e], sh4[g2][e]);
            v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          bf16x2 p0, p1;
          p0[0] = (bf16_t)v[0]; p0[1] = (bf16_t)v[1]; p1[0] = (bf16_t)v[2]; p1[1] = (bf16_t)v[3];
          const u32x2 pk = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
          asm volatile("ds_write_b64 %0, %1" ::"v"(stg + l31 * 128 + (((jj * 4 + g) ^ (l31 & 7)) << 4) + lh * 8), "v"(pk) : "memory");
        }
        }
      }
      u32x4 o[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = (lane >> 3) + 8 * it;
        asm volatile("ds_read_b128 %0, %1" : "=v"(o[it]) : "v"(stg + row * 128 + (((lane & 7) ^ (row & 7)) << 4)));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int pos = row0 + wave * 32 + (lane >> 3) + 8 * it;
        char* dst = pos < seg_rows ? out2b + ((size_t)(out_row0 + pos) * a.out2_ld + c * 64 + (lane & 7) * 8) * 2
                                   : reinterpret_cast<char*>(a.trash) + tid * 16;
        if (!(ab & 16)) *reinterpret_cast<u32x4*>(dst) = o[it];
      }
      if (!(ab & 16)) note_other(4);
    }

    cur_m += nbl;
    if (!valid_at(cur_m)) break;
  }
}

// conv3 weights [C][MID] bf16 (row-major, the conv_igemm layout of a 1x1 layer) -> stage images [C / 128][MID / KP][128 rows][KP * 2 B],
// 16-byte chunk c of row r at slot c ^ swz(r)
template <int KP>
__global__ void dual_pack_w3_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int C, int MID) {
  constexpr int CPR = KP / 8;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk
  if (i >= (size_t)C * MID / 8) return;
  const int s = (int)(i % CPR);
  const size_t rr = i / CPR;
  const int r = (int)(rr % 128);
  const size_t blk = rr / 128;
  const int nk = MID / KP, kp = (int)(blk % nk), nt = (int)(blk / nk);
  const int cl = s ^ du_swz<CPR>(r);
  *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(w + (size_t)(nt * 128 + r) * MID + kp * KP + cl * 8);
}

// conv1' weights [MID][C] bf16 -> stage images [C / KP][MID rows][KP * 2 B] in the k order of the register-resident y operand:
// logical chunk u = 2 * step + lh (step = 2 * (32-channel tile inside the phase) + s) holds the channels
// 32 tile + 16 s + 4 lh + {0, 1, 2, 3, 8, 9, 10, 11} of the phase -- what lane half lh of the accumulator layout owns for k-step s
template <int KP>
__global__ void dual_pack_w1_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int MID, int C) {
  constexpr int CPR = KP / 8;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)C * MID / 8) return;
  const int s = (int)(i % CPR);
  const size_t rr = i / CPR;
  const int r = (int)(rr % MID);
  const int P = (int)(rr / MID);
  const int u = s ^ du_swz<CPR>(r);
  const int step = u >> 1, lh = u & 1, jl = step >> 1, sb = step & 1;
  const bf16_t* src = w + (size_t)r * C + P * KP + 32 * jl + 16 * sb + 4 * lh;
  bf16_t v[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = src[e]; v[4 + e] = src[8 + e]; }
  *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(v);
}

int dual_phase_channels(int mid) { return SYLPH_AB_ENV("SYLPH_DUAL_CFG", 0) == 2 ? 32 : 64; }

int launch_dual_pack(const void* w3, const void* w1, void* w3_out, void* w1_out, int C, int mid, hipStream_t s) {
  if ((mid != 128 && mid != 256) || C % 128 != 0) return -1;
  const size_t n = (size_t)C * mid / 8;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dual_phase_channels(mid) == 32) {
    hipLaunchKernelGGL(dual_pack_w3_kernel<32>, grid, dim3(256), 0, s, (const bf16_t*)w3, (bf16_t*)w3_out, C, mid);
    hipLaunchKernelGGL(dual_pack_w1_kernel<32>, grid, dim3(256), 0, s, (const bf16_t*)w1, (bf16_t*)w1_out, mid, C);
  } else {
    hipLaunchKernelGGL(dual_pack_w3_kernel<64>, grid, dim3(256), 0, s, (const bf16_t*)w3, (bf16_t*)w3_out, C, mid);
    hipLaunchKernelGGL(dual_pack_w1_kernel<64>, grid, dim3(256), 0, s, (const bf16_t*)w1, (bf16_t*)w1_out, mid, C);
  }
  return (int)hipGetLastError();
}

bool conv_dual_ok(const DualArgs& a, int mid) {
  return (mid == 128 || mid == 256) && a.C % 128 == 0 && a.C >= 128 && (a.in_ld & 7) == 0 && (a.res_ld & 3) == 0 && (a.out_ld & 7) == 0 &&
         (a.out2_ld & 7) == 0 && a.in && a.w3 && a.tab3 && a.res && a.out && a.w1 && a.tab1 && a.out2 && a.trash && a.desc && a.n_mtiles > 0;
}

template <int MID, int KP, int NST, int BPC>
static int launch_dual_t(const DualArgs& a, int n_cu, hipStream_t s) {
  constexpr int lds = du_lds_bytes(MID, KP, NST);
  static_assert(lds * BPC <= 160 * 1024, "LDS budget");
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)conv_dual_kernel<MID, KP, NST, BPC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -7;
    attr = true;
  }
  int grid = (BPC * n_cu + 7) & ~7;
  const int need = (a.n_mtiles + 7) & ~7;
  if (need < grid) grid = need;
  hipLaunchKernelGGL((conv_dual_kernel<MID, KP, NST, BPC>), dim3(grid), dim3(256), lds, s, a);
  return (int)hipGetLastError();
}

int launch_conv_dual(const DualArgs& a, int mid, hipStream_t s) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return -7;
    n_cu = p.multiProcessorCount;
  }
  if (!conv_dual_ok(a, mid)) return -1;
#ifdef SYLPH_ABLATE
  DualArgs b = a;
  b.ablate = SYLPH_AB_ENV("SYLPH_DUAL_ABLATE", 0);
  const int cfg = SYLPH_AB_ENV("SYLPH_DUAL_CFG", 0);
  if (cfg == 2) {  // A/B: 32-channel phases, two blocks per CU (res3) / three stages (res4)
    if (mid == 128) return launch_dual_t<128, 32, 3, 2>(b, n_cu, s);
    return launch_dual_t<256, 32, 5, 1>(b, n_cu, s);
  }
  if (mid == 128) return launch_dual_t<128, 64, 4, 1>(b, n_cu, s);
  return launch_dual_t<256, 64, 4, 1>(b, n_cu, s);
#endif
  if (mid == 128) return launch_dual_t<128, 64, 4, 1>(a, n_cu, s);
  return launch_dual_t<256, 64, 4, 1>(a, n_cu, s);
}

}  // namespace sylph
