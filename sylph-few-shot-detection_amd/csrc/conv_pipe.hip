// Deep-pipelined implicit-GEMM convolution for the MFMA-bound layers (FCOS towers, FPN output convs).
//
// Same GEMM view, operand layouts and fused epilogue as conv_igemm.hip; different machine mapping.
// conv_igemm's 128x128 tile + two barriers per K-slice tops out near 0.35 of the bf16 MFMA peak: each
// block drains its loads before every barrier and the matrix pipe idles while a wave reads fragments.
// Here one 512-thread block per CU owns a 256x256 tile and keeps the matrix pipe fed by construction:
//
//  * 8 waves = 2 (M) x 4 (N), wave tile 128x64 = 4x2 MFMA tiles (128 accumulator VGPRs): 6 fragment
//    reads per 8 MFMAs instead of 4 per 4 (LDS read traffic per flop x0.75, L2->LDS traffic x0.5).
//  * K walks in 64-byte half-slices (32 bf16): one PHASE = 12 ds_read_b128 (L) then 16 MFMAs (M).
//    Four half-slice stages (4 x 32 KiB) rotate through LDS; the loads of phase q+3 (global_load_lds, 4
//    per lane) are issued BETWEEN the MFMAs of phase q, where the wave is parked on the matrix pipe
//    anyway, and are only waited for two phases later with a COUNTED s_waitcnt vmcnt(8 / 4): never a
//    drain in the steady state.
//  * The two M-halves of the block (waves 0-3 / 4-7: the two waves that share a SIMD) run staggered by
//    one barrier: while one half issues its 16 MFMAs (s_setprio 1), the other does its fragment
//    reads, address arithmetic and global_load_lds for the same phase.  Raw s_barrier + explicit
//    waitcnts only (a __syncthreads would drain the LDS-DMA queue).
//
// Hazards (B_n = n-th barrier, seg n = between B_n and B_n+1; half 0 does L(q) in seg 2q and M(q) in
// seg 2q+1, half 1 one segment later):
//   RAW  stage of phase q+1 is read from seg 2q+2 on; both halves execute their vmcnt wait for their own
//        phase-(q+1) loads in seg 2q+1, i.e. before B_{2q+2}.
//   WAR  stage (q-1)&3 is refilled (phase q+3) by half 0 in seg 2q+1 and by half 1 in seg 2q+2; its last
//        reader (half 1, L(q-1), seg 2q-1) retires its ds_reads with lgkmcnt(0) before B_{2q}.
//
// LDS image of a half-slice stage: [A: 256 rows x 64 B][B: 256 rows x 64 B]; a wave-level
// global_load_lds lands 16 rows x 64 B lane-linearly, so the bank swizzle is applied on the SOURCE
// address: slot s of row r holds logical 16-byte chunk s ^ ((r >> 2) & 3), and fragment reads use the
// same XOR (each ds_read_b128 lane group then covers all 64 banks exactly once).
//
// Scope (launch_conv checks): bf16 in/out, no residual, no per-segment Scale, ReLU on all channels or
// none, Cout % 256 == 0, padded scale/shift; optional fused GroupNorm partial statistics.
#include <stdlib.h>

#include "common.h"

namespace sylph {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

namespace {
constexpr int PBM = 256, PBN = 256, PNT = 512;
constexpr int HS = (PBM + PBN) * 64;  // bytes of one half-slice stage
constexpr int NSTAGE = 4;
constexpr int SCP = PBN + 4;          // fp32 pitch of the epilogue tile


#ifdef PIPE_NO_SCHED_FENCE
#define PIPE_SCHED_FENCE
#else
#define PIPE_SCHED_FENCE __builtin_amdgcn_sched_barrier(0)
#endif
#define SYLPH_BAR()                                \
  do {                                             \
    asm volatile("" ::: "memory");                 \
    PIPE_SCHED_FENCE;                              \
    __builtin_amdgcn_s_barrier();                  \
    PIPE_SCHED_FENCE;                              \
    asm volatile("" ::: "memory");                 \
  } while (0)

}  // namespace

__global__ __launch_bounds__(PNT, 1) void conv_pipe_kernel(const ConvArgs a) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware block -> tile map (as conv_igemm.hip)
  const int L = blockIdx.x;
  const int xcd = L & 7, q0 = L >> 3;
  const int chunk = (a.n_mtiles + 7) >> 3;
  const int m_local = q0 / a.n_ntiles;
  const int nt = q0 - m_local * a.n_ntiles;
  const int mt = xcd * chunk + m_local;
  if (m_local >= chunk || mt >= a.n_mtiles) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;

  const int2 tile = a.tiles[mt];
  const SegDesc sd = a.segs[tile.x];
  const int seg_rows = sd.out_H * sd.out_W;

  const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
  const T* __restrict__ wt = reinterpret_cast<const T*>(a.wt);
  const T* __restrict__ zero = reinterpret_cast<const T*>(a.zeros);
  const int Cin = a.Cin, KW = a.KW, ntaps = a.KH * a.KW;
  const int hp = Cin >> 5;       // half-slices (32 channels = 64 bytes per row) per tap
  const int NP = ntaps * hp;     // phases
  const int Ktot = ntaps * Cin;

  // ---- loader state: lane (r4, s4) fetches logical chunk cs of rows r4, r4 + 128 of A and of B ------
  const int r4 = tid >> 2, s4 = tid & 3;
  const int cs = s4 ^ ((r4 >> 2) & 3);
  int pbase[2];
  uint32_t pmask[2];
  const int goff = a.group_cout > 0 ? ((nt * PBN) / a.group_cout) * a.group_in_off : 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pos = tile.y + r4 + 128 * i;
    const bool rv = pos < seg_rows;
    const int oy = pos / sd.out_W, ox = pos - oy * sd.out_W;
    const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
    pbase[i] = (sd.in_row0 + iy0 * sd.in_W + ix0) * a.in_ld + cs * 8 + goff;
    uint32_t colm = 0, m = 0;
    for (int kx = 0; kx < KW; ++kx) colm |= ((unsigned)(ix0 + kx) < (unsigned)sd.in_W ? 1u : 0u) << kx;
    int sh = 0;
    for (int ky = 0; ky < a.KH; ++ky, sh += KW)
      if ((unsigned)(iy0 + ky) < (unsigned)sd.in_H) m |= colm << sh;
    pmask[i] = rv ? m : 0u;
  }
  int wbase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) wbase[j] = (nt * PBN + r4 + 128 * j) * Ktot + cs * 8;

  int itap = 0, icc = 0, ikh = 0, ikw = 0;  // the NEXT phase to issue
  // One load of the next phase: k = 0,1 the lane's two A rows, k = 2,3 its two B rows.  `valid` is false
  // past the last phase: the lane then fetches the zero page into a stage nobody reads any more, which
  // keeps the per-phase load count (and with it every vmcnt immediate) constant.
  auto issue_one = [&](int stage, int k, bool valid) {
    char* d = smem + stage * HS + wave * 1024 + (k & 1) * (128 * 64) + (k >> 1) * (PBM * 64);  // wave-uniform
    const T* zp = zero + cs * 8;
    const T* src;
    if (k < 2) {
      const int aoff = (ikh * sd.in_W + ikw) * a.in_ld + icc * 32;
      const T* real = in + (pbase[k] + aoff);
      src = (valid && ((pmask[k] >> itap) & 1u)) ? real : zp;
    } else {
      const T* real = wt + (wbase[k - 2] + itap * Cin + icc * 32);
      src = valid ? real : zp;
    }
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)d, 16, 0, 0);
  };
  auto advance = [&]() {  // select form: no branches
    const int c1 = icc + 1;
    const bool wrap = c1 == hp;
    icc = wrap ? 0 : c1;
    const int kw1 = ikw + 1;
    const bool wrapw = wrap && kw1 == KW;
    itap = wrap ? itap + 1 : itap;
    ikw = wrap ? (wrapw ? 0 : kw1) : ikw;
    ikh = wrapw ? ikh + 1 : ikh;
  };

  // ---- fragment addressing ---------------------------------------------------------------------
  int offA[2], offB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int slot = (ks * 2 + lh) ^ ((l31 >> 2) & 3);
    offA[ks] = (wm * 128 + l31) * 64 + slot * 16;
    offB[ks] = PBM * 64 + (wn * 64 + l31) * 64 + slot * 16;
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 fa[2][4], fb[2][2];

  auto ldfrag = [&](int stage) {
    const char* b = smem + stage * HS;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(b + offB[ks] + j * 2048);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(b + offA[ks] + i * 2048);
    }
  };

  // 16 MFMAs of one phase with the 4 loads of phase q+3 (and their address arithmetic) issued in their
  // shadow: the wave is parked on the matrix pipe between MFMAs anyway.  Straight-line code: every
  // scalar branch here costs the matrix pipe an instruction-fetch bubble.
  auto mma = [&](int stage, bool valid) {
#ifndef PIPE_NO_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
    int n = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);  // D^T
          if ((n & 3) == 1) issue_one(stage, n >> 2, valid);
          ++n;
        }
    advance();
#ifndef PIPE_NO_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  // ---- prologue: phases 0..2 in flight (phase q+3 is issued during M(q)) -------------------------------
  for (int s = 0; s < 3; ++s) {
#pragma unroll
    for (int k = 0; k < 4; ++k) issue_one(s, k, s < NP);
    advance();
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // phase 0 landed
  SYLPH_BAR();  // B_0

  if (wm == 0) {
    for (int q = 0; q < NP; ++q) {
      ldfrag(q & 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SYLPH_BAR();  // B_{2q+1}
      mma((q + 3) & 3, q + 3 < NP);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // own loads of phase q+1 landed (q+2, q+3 in flight)
      SYLPH_BAR();  // B_{2q+2}
    }
    SYLPH_BAR();    // B_{2NP+1}
  } else {
    SYLPH_BAR();    // B_1: the stagger
    for (int q = 0; q < NP; ++q) {
      ldfrag(q & 3);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // own loads of phase q+1 landed (q+2 in flight)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SYLPH_BAR();  // B_{2q+2}
      mma((q + 3) & 3, q + 3 < NP);
      SYLPH_BAR();  // B_{2q+3}
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zero-page tail loads must not land in the epilogue tile
  __syncthreads();

  // ---- fused epilogue: four 64-row passes through an fp32 LDS tile -----------------------------------
  float* const sC = reinterpret_cast<float*>(smem);
  bf16_t* __restrict__ out = reinterpret_cast<bf16_t*>(a.out);
  const int c8 = tid & 31, rr = tid >> 5;
  const int n0 = nt * PBN + c8 * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 s4v = a.scale ? reinterpret_cast<const float4*>(a.scale + n0)[h] : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 b4v = a.shift ? reinterpret_cast<const float4*>(a.shift + n0)[h] : make_float4(0.f, 0.f, 0.f, 0.f);
    sc[4 * h] = s4v.x; sc[4 * h + 1] = s4v.y; sc[4 * h + 2] = s4v.z; sc[4 * h + 3] = s4v.w;
    sh[4 * h] = b4v.x; sh[4 * h + 1] = b4v.y; sh[4 * h + 2] = b4v.z; sh[4 * h + 3] = b4v.w;
  }
  const bool relu = a.relu_nch > 0;
  bf16_t* __restrict__ outn = out + (size_t)sd.out_row0 * a.out_ld + n0;
  float gn_n = 0.f, gn_pv = 0.f, gn_s1 = 0.f, gn_s2 = 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (p > 0) lds_barrier();
    if (wm == (p >> 1)) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x16& c = acc[2 * (p & 1) + ii][j];
            *reinterpret_cast<float4*>(sC + (ii * 32 + l31) * SCP + wn * 64 + j * 32 + 8 * g + 4 * lh) =
                make_float4(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
          }
    }
    lds_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = rr + 16 * it;
      const int pos = tile.y + p * 64 + rl;
      if (pos < seg_rows) {
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
        if (a.gn_partial) {  // shifted sums about the lane's first row mean (exact enough in fp32, no divisions)
          if (gn_n == 0.f) gn_pv = 0.125f * (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = v[e] - gn_pv; gn_s1 += d; gn_s2 = fmaf(d, d, gn_s2); }
          gn_n += 8.f;
        }
        store8<bf16_t>(outn + (size_t)pos * a.out_ld, v);
      }
    }
  }
  if (a.gn_partial) {
    lds_barrier();
    float* red = sC;  // [16][32][3]
    const float inv_n = gn_n > 0.f ? 1.f / gn_n : 0.f;
    red[(rr * 32 + c8) * 3 + 0] = gn_n;
    red[(rr * 32 + c8) * 3 + 1] = gn_pv + gn_s1 * inv_n;       // lane mean
    red[(rr * 32 + c8) * 3 + 2] = gn_s2 - gn_s1 * gn_s1 * inv_n;  // lane M2
    lds_barrier();
    if (rr == 0) {
      float N = 0.f, M = 0.f, Q = 0.f;
      for (int r = 0; r < 16; ++r) {
        const float nb = red[(r * 32 + c8) * 3 + 0];
        if (nb > 0.f) {
          const float mb = red[(r * 32 + c8) * 3 + 1], qb = red[(r * 32 + c8) * 3 + 2];
          const float nn = N + nb, delta = mb - M;
          M += delta * (nb / nn);
          Q += qb + delta * delta * (N * nb / nn);
          N = nn;
        }
      }
      float* gp = a.gn_partial + ((size_t)mt * (a.Cout >> 3) + (n0 >> 3)) * 3;
      gp[0] = N; gp[1] = M; gp[2] = Q;
    }
  }
}

bool conv_pipe_ok(DType dt, bool out_f32, const ConvArgs& a) {
  return dt == DT_BF16 && !out_f32 && !a.stem && !a.in2 && a.res_mode == 0 && a.mul_nch == 0 &&
         (a.relu_nch == 0 || a.relu_nch >= a.Cout) && a.Cout % PBN == 0 && a.Cin % 32 == 0 && a.KH * a.KW <= 31 &&
         a.ss_padded_host && (a.out_ld & 7) == 0 && a.zeros != nullptr;
}

int launch_conv_pipe(const ConvArgs& a_in, hipStream_t s) {
  const ConvArgs& a = a_in;
  static bool attr_set = false;
  const int lds = NSTAGE * HS;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -7;
    attr_set = true;
  }
  const int chunk = (a.n_mtiles + 7) / 8;
  const int grid = 8 * chunk * a.n_ntiles;
  hipLaunchKernelGGL(conv_pipe_kernel, dim3(grid), dim3(PNT), lds, s, a);
  return (int)hipGetLastError();
}

}  // namespace sylph
