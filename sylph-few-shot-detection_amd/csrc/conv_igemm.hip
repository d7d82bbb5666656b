// Implicit-GEMM convolution on MFMA for gfx950.
//
//   out[pos][n] = epi( sum_{kh,kw,c} in[pos*stride + (kh,kw) - pad][c] * wt[n][kh][kw][c] )
//
// GEMM view: M = output positions (rows of the position-major activation buffer), N = Cout,
// K = KH*KW*Cin walked tap-major in 128-byte slices (64 bf16 / 32 fp32), so a K-slice never
// straddles a tap and the A operand of a slice is one 128-byte span of one input row: the im2col
// gather is a per-row address + a validity mask computed once per tile.  Both operands are
// K-contiguous ([pos][C] activations, [n][kh][kw][c] weights): the layout the 32x32 MFMA
// fragments want (8 bf16 / 1 fp32 along K per lane).
//
// Block = 256 threads = 4 wave64 arranged WGM x WGN; wave tile (BM/WGM) x (BN/WGN) made of
// 32x32 MFMA tiles (v_mfma_f32_32x32x16_bf16, or the exact-fp32 v_mfma_f32_32x32x2_f32 in parity mode).
//
// Staging: global_load_lds_dwordx4 (HBM/L2 -> LDS directly, no VGPR round trip, no ds_write).
// One wave instruction lands 64 x 16 B = 8 LDS rows of 128 B.  The LDS image is lane-linear, so the
// bank-conflict swizzle lives on the SOURCE side: lane (row r, slot s) fetches logical 16-byte
// chunk s ^ ((r>>1)&7) of its row, and the ds_read_b128 fragment reads apply the same XOR, so
// the 16 lanes of a read group hit 16 distinct slots.  Out-of-image taps (and rows past the end
// of a segment) fetch from a zero page instead of branching.
// NBUF = 1: one 128-byte slice resident (LDS (BM+BN)*128 B, two barriers per slice, latency hidden
// by 3-4 co-resident blocks per CU); NBUF = 2: next slice in flight during the MFMAs.
//
// HALO mode (3x3 stride-1 pad-1 layers): the M tile is an 8 x 16 patch of output positions; its 10 x 18 input halo is
// staged ONCE per 64-channel slice and the nine taps read shifted rows of it, so only the weight tile is fetched per
// tap (K order: channel slice outer, taps inner).  The halo image is swizzled by halo COLUMN, which keeps a fragment
// read that spans two patch rows conflict-free.
//
// MFMA operands are passed swapped (D^T = W * A^T): a lane then owns 4 consecutive CHANNELS of one position, and the
// fp32 epilogue tile is written with ds_write_b128.
//
// Epilogue (fused): accumulators -> LDS (fp32) -> per lane 8 consecutive channels: per-channel
// scale/shift (FrozenBN or bias), residual add (optionally through a nearest 2x upsample: the FPN
// top-down path), per-segment Scale_l, ReLU, GroupNorm partial statistics, 16-byte stores.  The FAST instantiation
// (bf16 in/out, full tiles) has a branch-free row loop: the short-K tiles are instruction-issue bound.
// Reference ops replaced: every F.conv2d / nn.Conv2d on the path: ResNet stem and
// bottleneck convs + FrozenBN (detectron2), FPN lateral/output/P6/P7, FCOS towers (fcos.py:72-122),
// bbox_pred/ctrness (fcos.py:656-664), CondConvBasic (head_utils.py:60-81), code-generator
// convs (code_generator.py:509-688).
#include <stdlib.h>

#include "common.h"

namespace sylph {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <typename T> struct Mma;

template <> struct Mma<bf16_t> {
  static constexpr int KSTEPS = 4;  // 64 bf16 per slice / 16 per MFMA
  typedef bf16x8 frag_t;
  static __device__ __forceinline__ frag_t load(const char* tile, int row, int ks, int lane) {
    const int chunk = ks * 2 + (lane >> 5);
    const int sw = chunk ^ ((row >> 1) & 7);
    return *reinterpret_cast<const frag_t*>(tile + row * 128 + sw * 16);
  }
  static __device__ __forceinline__ frag_t load_sw(const char* tile, int row, int ks, int lane, int swz) {
    return *reinterpret_cast<const frag_t*>(tile + row * 128 + ((ks * 2 + (lane >> 5)) ^ swz) * 16);
  }
  static __device__ __forceinline__ f32x16 mma(frag_t a, frag_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <> struct Mma<float> {
  static constexpr int KSTEPS = 16;  // 32 fp32 per slice / 2 per MFMA
  typedef float frag_t;
  static __device__ __forceinline__ frag_t load(const char* tile, int row, int ks, int lane) {
    const int k = ks * 2 + (lane >> 5);
    const int sw = (k >> 2) ^ ((row >> 1) & 7);
    return *reinterpret_cast<const float*>(tile + row * 128 + sw * 16 + (k & 3) * 4);
  }
  static __device__ __forceinline__ frag_t load_sw(const char* tile, int row, int ks, int lane, int swz) {
    const int k = ks * 2 + (lane >> 5);
    return *reinterpret_cast<const float*>(tile + row * 128 + ((k >> 2) ^ swz) * 16 + (k & 3) * 4);
  }
  static __device__ __forceinline__ f32x16 mma(frag_t a, frag_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

// Split-bf16 parity mode (DT_F32S): fp32 activations and fp32-sized weights in HBM / LDS exactly as in the fp32 mode, but the products
// run on the bf16 pipe at 3/16 of the fp32-MFMA cost.  x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (both round-to-nearest-even:
// |x - hi - lo| <= 2^-18 |x|), and x * w ~ hi_x hi_w + hi_x lo_w + lo_x hi_w into the same fp32 accumulator (the dropped lo * lo term
// is 2^-18 relative).  Activations are split in registers when a fragment is read (8 consecutive fp32 of one row = two ds_read_b128);
// weights are split ONCE on the host: each 32-element K-slice of a packed row is stored as [32 bf16 hi | 32 bf16 lo] -- the same 128
// bytes, so the staging code does not know the difference.
struct MmaSplit {
  static constexpr int KSTEPS = 2;  // 32 fp32 per slice / 16 per MFMA
  static __device__ __forceinline__ void load_a(const char* tile, int row, int ks, int lane, bf16x8& hi, bf16x8& lo) {
    const int c0 = ks * 4 + (lane >> 5) * 2, sw = (row >> 1) & 7;
    const f32x4 p = *reinterpret_cast<const f32x4*>(tile + row * 128 + ((c0 ^ sw) << 4));
    const f32x4 q = *reinterpret_cast<const f32x4*>(tile + row * 128 + (((c0 + 1) ^ sw) << 4));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bf16_t hp = (bf16_t)p[e], hq = (bf16_t)q[e];
      hi[e] = hp; hi[4 + e] = hq;
      lo[e] = (bf16_t)(p[e] - (float)hp); lo[4 + e] = (bf16_t)(q[e] - (float)hq);
    }
  }
  static __device__ __forceinline__ void load_w(const char* tile, int row, int ks, int lane, bf16x8& hi, bf16x8& lo) {
    const int c = ks * 2 + (lane >> 5), sw = (row >> 1) & 7;
    hi = *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((c ^ sw) << 4));
    lo = *reinterpret_cast<const bf16x8*>(tile + row * 128 + (((c + 4) ^ sw) << 4));
  }
};

// Per-tile GroupNorm partial: every lane holds shifted sums over its rows of one 8-channel group; the RPP lanes of a
// group are merged (Chan) in a fixed order by lane row 0 and written as (n, mean, M2).
template <int RPP, int TPR>
__device__ __forceinline__ void gn_tile_reduce(float* red, int rr, int c8, float gn_n, float gn_pv, float gn_s1, float gn_s2,
                                               float* gp, bool active) {
  lds_barrier();
  const float inv_n = gn_n > 0.f ? 1.f / gn_n : 0.f;
  red[(rr * TPR + c8) * 3 + 0] = gn_n;
  red[(rr * TPR + c8) * 3 + 1] = gn_pv + gn_s1 * inv_n;          // lane mean
  red[(rr * TPR + c8) * 3 + 2] = gn_s2 - gn_s1 * gn_s1 * inv_n;  // lane M2
  lds_barrier();
  if (rr == 0 && active) {
    float N = 0.f, M = 0.f, Q = 0.f;
    for (int r = 0; r < RPP; ++r) {
      const float nb = red[(r * TPR + c8) * 3 + 0];
      if (nb > 0.f) {
        const float mb = red[(r * TPR + c8) * 3 + 1], qb = red[(r * TPR + c8) * 3 + 2];
        const float nn = N + nb, delta = mb - M;
        M += delta * (nb / nn);
        Q += qb + delta * delta * (N * nb / nn);
        N = nn;
      }
    }
    gp[0] = N; gp[1] = M; gp[2] = Q;
  }
}

template <typename T, typename OutT, int BM, int BN, int WGM, int WGN, int NBUF, bool FAST, bool HALO, bool SPLIT = false>
__global__ __launch_bounds__(WGM * WGN * 64, (SPLIT ? (BM * BN == 128 * 128 ? 3 : (BM * BN == 128 * 64 && WGN == 2 ? 4 : 5)) : WGM * WGN == 8 ? 4 : (BM * BN <= 128 * 64 ? 5 : (BM * BN >= 256 * 128 ? 2 : (BM * BN == 128 * 128 ? 4 : 1))))) void conv_igemm_kernel(const ConvArgs a) {
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BK = 8 * EPC;                // elements per 128-byte K-slice
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int NT = WGM * WGN * 64;         // threads per block (4 or 8 waves)
  constexpr int RS = NT / 8;                 // tile rows staged by one load instruction of the whole block
  constexpr int AR = BM / RS, BR = BN / RS;  // wave-level loads per slice (8 rows each)
  // HALO (3x3 s1 p1): the M tile is an 8 x 16 patch of output positions; its 10 x 18 input halo (192 LDS rows with the
  // tail of the last load round) is loaded ONCE per 64-channel slice and the 9 taps read shifted rows of it.
  // The patch shape (ph x pw <= BM positions, (ph + 2) * (pw + 2) <= HALLOC halo rows) is chosen per segment on the host
  // (api_conv.hip pick_patch: 10 x 12 for the 100 x 168 / 50 x 84 maps, 8 x 16 for 200 x 336, ...).
  constexpr int HALLOC = 184;                          // LDS rows reserved for the halo (whole 8-row wave loads)
  constexpr int HRND = (HALLOC + NT / 8 - 1) / (NT / 8);
  constexpr int STAGE = HALO ? (BN + HALLOC) * 128 : (BM + BN) * 128;
  static_assert(!HALO || (NBUF == 1 && BM == 128), "halo mode: 128-position patches, single stage");
  static_assert(!SPLIT || (sizeof(T) == 4 && !HALO && !FAST), "split-bf16 mode: fp32 storage, generic epilogue");
  static_assert((WGM * WGN == 4 || WGM * WGN == 8) && TM >= 1 && TN >= 1 && AR >= 1 && BR >= 1, "bad tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware block -> tile map: XCD x owns a contiguous chunk of M-tiles; inside it the N index
  // is innermost so consecutive blocks of one XCD share the A tile through that XCD's L2.
  const int L = blockIdx.x;
  const int xcd = L & 7, q = L >> 3;
  const int chunk = (a.n_mtiles + 7) >> 3;
  const int m_local = q / a.n_ntiles;
  const int nt = q - m_local * a.n_ntiles;
  const int mt = xcd * chunk + m_local;
  if (m_local >= chunk || mt >= a.n_mtiles) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int c16 = tid & 7, r0 = tid >> 3;
  const int cl = c16 ^ ((r0 >> 1) & 7);  // logical chunk this lane fetches (source-side swizzle)

  const int2 tile = a.tiles[mt];
  const SegDesc sd = a.segs[tile.x];
  const int seg_rows = sd.out_H * sd.out_W;
  const int oy0 = HALO ? (tile.y >> 16) : 0, ox0 = HALO ? (tile.y & 0xffff) : 0;  // HALO: tile.y packs the patch origin
  // HALO patch geometry (wave-uniform).  Divisions by pw / pw + 2 of values < 256 use a 16-bit reciprocal (exact there).
  const int HPW = HALO ? sd.pw : 16, HW2 = HALO ? sd.hpitch : 18, HROWS = HALO ? (sd.ph + 2) * HW2 : 0, HPOS = HALO ? sd.ph * sd.pw : 0;
  const unsigned inv_pw = sd.inv_pw, inv_hw2 = sd.inv_hw2;

  const T* __restrict__ in = reinterpret_cast<const T*>(a.in);
  const T* __restrict__ wt = reinterpret_cast<const T*>(a.wt);
  const T* __restrict__ zero = reinterpret_cast<const T*>(a.zeros);
  const int Cin = a.Cin, KW = a.KW, ntaps = a.KH * a.KW;
  const int cpt = Cin / BK;            // K-slices per tap
  // Dual-source pointwise mode (a.in2): K = [Cin channels of `in` | Cin2 channels of `in2`], the
  // second source sampled with its own spatial stride (bottleneck conv3 + projection shortcut in
  // one GEMM; FrozenBN scales are pre-folded into the weights, shifts summed).
  const T* __restrict__ in2 = reinterpret_cast<const T*>(a.in2);
  const int cpt2 = in2 ? a.Cin2 / BK : 0;
  const int nk = ntaps * cpt + cpt2;
  const int Ktot = ntaps * Cin + (in2 ? a.Cin2 : 0);
  // Split K (small launches whose K loop is latency-bound and whose tiles do not fill the chip): blockIdx.y owns the K-slices
  // [kt0, kt1) and writes its fp32 partial sums to its own plane of a.out; splitk_finish_kernel adds the planes in a fixed order
  // and applies the epilogue.  Host guarantees (add_conv): no halo mode, no second K source, fp32 out, identity epilogue.
  int kt0 = 0, kt1 = nk;
  if (a.ksplit > 1) {
    kt0 = (int)((long)blockIdx.y * nk / a.ksplit);
    kt1 = (int)((long)(blockIdx.y + 1) * nk / a.ksplit);
  }

  int abase[AR], abase2[AR];
  uint32_t amask[AR];
  // Pointwise stride-1 convs (most launches; their tiles are short, so per-tile setup is a visible
  // share of the instruction stream) address the input by output position: no div, no tap loop.
  const bool pw_fast = ntaps == 1 && a.pad == 0 && a.stride == 1 && !a.stem && sd.in_W == sd.out_W;
  const bool in2_direct = in2 && a.stride2 == 1 && sd.in2_W == sd.out_W;
#pragma unroll
  for (int i = 0; i < (HALO ? 0 : AR); ++i) {
    const int pos = tile.y + r0 + RS * i;
    const bool rv = pos < seg_rows;
    int oy = 0, ox = 0;
    if (!pw_fast || (in2 && !in2_direct)) { oy = pos / sd.out_W; ox = pos - oy * sd.out_W; }
    uint32_t m = 0;
    if (pw_fast) {
      abase[i] = (sd.in_row0 + pos) * a.in_ld + cl * EPC;
      m = rv ? 1u : 0u;
    } else if (a.stem) {
      // ResNet stem (7x7 s2 p3 over a 4-channel-padded image): a K-slice is RPS kernel rows of an
      // 8-pixel window [2*ox-4, 2*ox+4) x 4 channels (window pixel 0 and channel 3 carry zero
      // weights), so every 16-byte chunk is PPC whole, aligned pixels of one input row.
      constexpr int PPC = EPC / 4, CPR = 8 / PPC, RPS = 8 / CPR;
      const int khl = cl / CPR, pxo = (cl % CPR) * PPC;
      const int iy0 = oy * 2 - 3 + khl, ixw = ox * 2 - 4 + pxo;
      abase[i] = (sd.in_row0 + iy0 * sd.in_W + ixw) * 4;
      for (int t = 0; t < ntaps; ++t) {
        const bool ok = rv && (unsigned)(iy0 + t * RPS) < (unsigned)sd.in_H && (t * RPS + khl) < 7 &&
                        (unsigned)ixw < (unsigned)sd.in_W;
        m |= (ok ? 1u : 0u) << t;
      }
    } else {
      const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
      abase[i] = (sd.in_row0 + iy0 * sd.in_W + ix0) * a.in_ld + cl * EPC;
      uint32_t colm = 0;  // taps of one kernel row whose column is inside the image
      for (int kx = 0; kx < KW; ++kx) colm |= ((unsigned)(ix0 + kx) < (unsigned)sd.in_W ? 1u : 0u) << kx;
      int sh = 0;
      for (int ky = 0; ky < a.KH; ++ky, sh += KW)
        if ((unsigned)(iy0 + ky) < (unsigned)sd.in_H) m |= colm << sh;
      if (!rv) m = 0;
    }
    amask[i] = m;
    if (in2) {
      abase2[i] = in2_direct ? (sd.in2_row0 + pos) * a.in2_ld + cl * EPC
                             : (sd.in2_row0 + (oy * a.stride2) * sd.in2_W + ox * a.stride2) * a.in2_ld + cl * EPC;
      if (rv) amask[i] |= 1u << 31;
    }
  }
  // HALO: source of LDS halo row t*32 + r0 (input position (oy0-1+hy, ox0-1+hx)); rows outside the image, and the
  // tail rows past the 10 x 18 halo, come from the zero page.
  int hsrc[HRND];
  uint32_t hmask = 0;
  int hb[TM];
  if constexpr (HALO) {
    const int goff_h = a.group_cout > 0 ? ((nt * BN) / a.group_cout) * a.group_in_off : 0;
#pragma unroll
    for (int t = 0; t < HRND; ++t) {
      const int hrow = t * RS + r0;
      const int hy = (int)(((unsigned)hrow * inv_hw2) >> 16), hx = hrow - hy * HW2;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      const bool ok = hrow < HROWS && (unsigned)iy < (unsigned)sd.in_H && (unsigned)ix < (unsigned)sd.in_W;
      // bank swizzle keyed on k = hy * pw + hx: the reader of tap (kh, kw) at patch position m sits on k = m + kh * pw + kw,
      // so the 16 lanes of a ds_read_b128 group (16 consecutive m) see 16 distinct k mod 16, and k has the parity of the
      // halo row index (row - k = 2 * hy): (row & 1, (k >> 1) & 7) is a distinct 16-byte bank slot for every lane
      const int clh = c16 ^ (((hy * HPW + hx) >> 1) & 7);
      hsrc[t] = (sd.in_row0 + iy * sd.in_W + ix) * a.in_ld + clh * EPC + goff_h;
      hmask |= (ok ? 1u : 0u) << t;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = wm * WTM + i * 32 + (lane & 31);
      const int my = (int)(((unsigned)m * inv_pw) >> 16);
      hb[i] = my * HW2 + (m - my * HPW);
    }
  }
  if (a.group_cout > 0) {  // grouped conv: this N tile's group reads its own input-channel window
    const int goff = ((nt * BN) / a.group_cout) * a.group_in_off;
#pragma unroll
    for (int i = 0; i < AR; ++i) abase[i] += goff;
  }
  size_t bbase[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) bbase[j] = (size_t)(nt * BN + r0 + RS * j) * Ktot + cl * EPC;

  int tap = 0, cc = 0, kh = 0, kw = 0;  // state of the NEXT slice to fetch
  if (kt0 > 0) { tap = kt0 / cpt; cc = kt0 - tap * cpt; kh = tap / KW; kw = tap - kh * KW; }  // split K: taps outer, channel slices inner
  auto issue = [&](int buf) {
    if constexpr (HALO) {
      // LDS: [B tile][halo].  Slice order: 64-channel slice outer, taps inner; the halo is fetched on tap 0.
      char* dBh = smem + wave * 8 * 128;
      char* dH = dBh + BN * 128;
      if (tap == 0) {
#pragma unroll
        for (int t = 0; t < HRND; ++t) {
          if (t * RS + wave * 8 >= HALLOC) continue;  // wave-uniform: nothing of the halo in this wave's 8 rows
          const T* src = ((hmask >> t) & 1u) ? in + (hsrc[t] + cc * BK) : zero + cl * EPC;
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dH + t * RS * 128), 16, 0, 0);
        }
      }
      const int boff = tap * Cin + cc * BK;
#pragma unroll
      for (int j = 0; j < BR; ++j)
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wt + (bbase[j] + boff)), (lds_ptr_t)(dBh + j * RS * 128), 16, 0, 0);
      ++tap;
      if (++kw == KW) { kw = 0; ++kh; }
      if (tap == ntaps) { tap = 0; kh = 0; kw = 0; ++cc; }
      return;
    }
    char* dA = smem + buf * STAGE + wave * 8 * 128;  // wave-uniform: lane l lands at +16*l
    char* dB = dA + BM * 128;
    const bool second = tap >= ntaps;  // only in dual-source mode
    const int aoff = second ? cc * BK : (kh * a.tap_dy * sd.in_W + kw) * a.in_ld + cc * BK;
    const int boff = tap * Cin + cc * BK;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const T* src;
      if (second) src = (amask[i] >> 31) ? in2 + (abase2[i] + aoff) : zero + cl * EPC;
      else src = ((amask[i] >> tap) & 1u) ? in + (abase[i] + aoff) : zero + cl * EPC;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dA + i * RS * 128), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BR; ++j)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wt + (bbase[j] + boff)), (lds_ptr_t)(dB + j * RS * 128), 16, 0, 0);
    // K order: taps outer, channel slices inner.  (Slices outer / taps inner would re-read the same lines
    // back to back, but every CU then hammers lines with equal (address >> 7) & 3 at the same time and the
    // L2 channels camp: measured 10-20 % slower on the 3x3 layers.)
    if (++cc == cpt && !second) { cc = 0; ++tap; if (++kw == KW) { kw = 0; ++kh; } }
  };

  // Every HBM access of the block is issued up front so that the only exposed memory round trip is
  // the first operand slice:
  //  * residual tile via global_load_lds (narrow HBM-bound tiles): piece-major LDS image, piece q =
  //    bytes [128q, 128q+128) of every row at 128-byte pitch, read back by the epilogue;
  //  * per-channel scale/shift: one float4 per lane of the first BN/2 lanes, parked in LDS later.
  constexpr int SCP = BN + 4;                 // fp32 pitch of the epilogue tile (+16 B: conflict-free b128 writes)
  constexpr int SC_BYTES = WTM * SCP * 4;
  constexpr int RES_OFF = (NBUF * STAGE > SC_BYTES) ? NBUF * STAGE : SC_BYTES;
  constexpr int RES_PIECES = BN * (int)sizeof(T) / 128;
  constexpr bool SS_IN_STAGE = NBUF * STAGE >= SC_BYTES + BN * 8;  // scale/shift fit behind the epilogue tile
  const int ss_off = SS_IN_STAGE ? SC_BYTES : RES_OFF + (a.res_lds ? BM * BN * (int)sizeof(T) : 0);
  float4 ssv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (BN <= 64 && a.ss_padded) {
    if (tid < BN / 4) ssv = a.scale ? reinterpret_cast<const float4*>(a.scale)[nt * (BN / 4) + tid] : make_float4(1.f, 1.f, 1.f, 1.f);
    else if (tid < BN / 2 && a.shift) ssv = reinterpret_cast<const float4*>(a.shift)[nt * (BN / 4) + tid - BN / 4];
  }
  if (a.res_lds) {
    const T* __restrict__ resg = reinterpret_cast<const T*>(a.res);
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int pos = tile.y + r0 + RS * i;
      int rp = pos;
      if (a.res_mode == 2) {
        const int oy = pos / sd.out_W, ox = pos - oy * sd.out_W;
        rp = (oy >> 1) * sd.res_W + (ox >> 1);
      }
      const T* rowp = resg + (size_t)(sd.res_row0 + rp) * a.res_ld + nt * BN + c16 * EPC;
#pragma unroll
      for (int q = 0; q < RES_PIECES; ++q) {
        const T* src = pos < seg_rows ? rowp + q * (128 / (int)sizeof(T)) : zero + c16 * EPC;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(smem + RES_OFF + q * BM * 128 + (wave * 8 + i * RS) * 128),
                                         16, 0, 0);
      }
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int buf, int hoff) {
    const char* tA = HALO ? smem + BN * 128 : smem + buf * STAGE;
    const char* tB = HALO ? smem : tA + BM * 128;
    if constexpr (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < MmaSplit::KSTEPS; ++ks) {
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) MmaSplit::load_a(tA, wm * WTM + i * 32 + (lane & 31), ks, lane, ah[i], al[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) MmaSplit::load_w(tB, wn * WTN + j * 32 + (lane & 31), ks, lane, bh[j], bl[j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {  // the two cross terms first, the leading term last
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
          }
      }
      return;
    }
#pragma unroll
    for (int ks = 0; ks < Mma<T>::KSTEPS; ++ks) {
      typename Mma<T>::frag_t fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (HALO) fa[i] = Mma<T>::load_sw(tA, hb[i] + (hoff >> 8), ks, lane, (((lane & 15) + (hoff & 255)) >> 1) & 7);
        else fa[i] = Mma<T>::load(tA, wm * WTM + i * 32 + (lane & 31), ks, lane);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = Mma<T>::load(tB, wn * WTN + j * 32 + (lane & 31), ks, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::mma(fb[j], fa[i], acc[i][j]);  // D^T: a lane holds 4 consecutive channels
    }
  };

  if constexpr (NBUF == 1) {
    for (int kt = kt0; kt < kt1; ++kt) {
      const int hoff = ((kh * HW2 + kw) << 8) | ((kh * HPW + kw) & 15);  // halo row offset (and swizzle-key shift) of the tap about to be fetched
      issue(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(0, hoff);
      __syncthreads();
    }
  } else if constexpr (NBUF == 2) {
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = (kt - kt0) & 1;
      if (kt + 1 < kt1) issue(buf ^ 1);
      compute(buf, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // Ring of NBUF stages (3; 4 measured equal), NBUF - 1 slices in flight, for launches of about one block per CU: nothing else hides a slice's
    // round trip there.  Every wave issues LPS loads per slice and vmcnt retires in order: "slice kt has landed" = at most (slices
    // issued after it) * LPS of this wave's loads outstanding.  The barrier that publishes slice kt also says every wave is done with
    // slice kt - 1, whose stage the next issue overwrites.  lds_barrier(), not __syncthreads(): its fence would drain the slices in flight.
    constexpr int LPS = AR + BR;
    static_assert((NBUF - 2) * LPS <= 63, "vmcnt range");
    constexpr int W1 = LPS, W2 = 2 * LPS;
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
      if (kt0 + s < kt1) issue(s);
    int cur = 0, nxt = NBUF - 1;
    for (int kt = kt0; kt < kt1; ++kt) {
      const int after = kt1 - 1 - kt;  // slices issued behind this one (capped at NBUF - 2)
      if (after == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (after == 1 || NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W2) : "memory");
      lds_barrier();
      if (kt + NBUF - 1 < kt1) issue(nxt);
      compute(cur, 0);
      cur = cur + 1 == NBUF ? 0 : cur + 1;
      nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
    }
    __syncthreads();
  }

  // ---- fused epilogue ---------------------------------------------------------------------------
  // One pass per wave-row: its waves park their accumulators in LDS as fp32 [WTM][BN+4] (aliasing the
  // staging buffers; the K loop ended on a barrier), then every lane owns 8 consecutive channels of
  // one row: 16-byte residual loads, 16-byte (bf16) / 32-byte (fp32) stores, 256 B per 16 lanes.
  float* const sC = reinterpret_cast<float*>(smem);
  constexpr int TPR = BN / 8;     // lanes per output row
  constexpr int RPP = NT / TPR;   // rows per sweep
  OutT* __restrict__ out = reinterpret_cast<OutT*>(a.out) + (a.ksplit > 1 ? (size_t)blockIdx.y * (size_t)a.split_stride : (size_t)0);
  const T* __restrict__ res = reinterpret_cast<const T*>(a.res);
  const int c8 = tid % TPR, rr = tid / TPR;
  const int n0 = nt * BN + c8 * 8;
  const bool active = n0 < a.Cout;
  const bool vec = (n0 + 8 <= a.Cout) && ((a.out_ld & 7) == 0) && (a.res_mode == 0 || (a.res_ld & 7) == 0);
  float sc[8], sh[8];
  if (BN <= 64 && a.ss_padded) {
    if (tid < BN / 2) *reinterpret_cast<float4*>(smem + ss_off + tid * 16) = ssv;  // visible after the barrier below
  } else if (!FAST) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool nv = n0 + e < a.Cout;
      sc[e] = (nv && a.scale) ? a.scale[n0 + e] : 1.f;
      sh[e] = (nv && a.shift) ? a.shift[n0 + e] : 0.f;
    }
  }
  if constexpr (FAST) {
    static_assert(sizeof(T) == 2, "FAST epilogue: bf16 activations");
    // Launch-time guarantees (launch_conv): Cout % BN == 0, 16-byte aligned rows, padded scale/shift, no
    // per-segment Scale, no GroupNorm partials, ReLU on all channels or none.  The row loop is then
    // branch-free apart from the residual mode: these short-K tiles are instruction-issue bound.
    if (BN > 64) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 s4 = a.scale ? reinterpret_cast<const float4*>(a.scale + n0)[h] : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 b4 = a.shift ? reinterpret_cast<const float4*>(a.shift + n0)[h] : make_float4(0.f, 0.f, 0.f, 0.f);
        sc[4 * h] = s4.x; sc[4 * h + 1] = s4.y; sc[4 * h + 2] = s4.z; sc[4 * h + 3] = s4.w;
        sh[4 * h] = b4.x; sh[4 * h + 1] = b4.y; sh[4 * h + 2] = b4.z; sh[4 * h + 3] = b4.w;
      }
    }
    const bool relu = a.relu_nch > 0;
    float gn_n = 0.f, gn_pv = 0.f, gn_s1 = 0.f, gn_s2 = 0.f;  // fused GroupNorm statistics (see the generic path)
    const T* __restrict__ resn = res + n0;
    OutT* __restrict__ outn = out + (size_t)sd.out_row0 * a.out_ld + n0;
#pragma unroll
    for (int p = 0; p < WGM; ++p) {
      if (p > 0) lds_barrier();
      if (wm == p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(sC + (i * 32 + (lane & 31)) * SCP + wn * WTN + j * 32 + 8 * g + 4 * (lane >> 5)) =
                  make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
      }
      lds_barrier();
      if (p == 0 && BN <= 64) {
        const float* ssl = reinterpret_cast<const float*>(smem + ss_off);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = ssl[c8 * 8 + e]; sh[e] = ssl[BN + c8 * 8 + e]; }
      }
      // Two sweeps over this pass's rows: first every residual load is issued, then the math and the stores.
      // (vmcnt retires in order, so a residual load issued after a store would also wait for that store.)
      constexpr int ITERS = (WTM + RPP - 1) / RPP;
      int posv[ITERS];
      bool pvv[ITERS];
      uint4 rraw[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int rl = rr + it * RPP;
        int pos = tile.y + p * WTM + rl;
        bool pv = pos < seg_rows;
        if constexpr (HALO) {
          const int m = p * WTM + rl, my = (int)(((unsigned)m * inv_pw) >> 16);
          const int oy = oy0 + my, ox = ox0 + (m - my * HPW);
          pos = oy * sd.out_W + ox;
          pv = m < HPOS && oy < sd.out_H && ox < sd.out_W;
        }
        posv[it] = pos;
        pvv[it] = (WTM % RPP == 0 || rl < WTM) && pv;
        rraw[it] = make_uint4(0u, 0u, 0u, 0u);
        if (a.res_mode != 0 && pvv[it]) {
          int rp = pos;
          if (a.res_mode == 2) {
            const int oy = pos / sd.out_W, ox = pos - oy * sd.out_W;
            rp = (oy >> 1) * sd.res_W + (ox >> 1);
          }
          rraw[it] = *reinterpret_cast<const uint4*>(resn + (size_t)(sd.res_row0 + rp) * a.res_ld);
        }
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int rl = rr + it * RPP;
        if (pvv[it]) {
          float v[8];
          const float4 lo = *reinterpret_cast<const float4*>(sC + rl * SCP + c8 * 8);
          const float4 hi = *reinterpret_cast<const float4*>(sC + rl * SCP + c8 * 8 + 4);
          v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
          if (a.res_mode != 0) {  // FAST is bf16 only: 8 residual channels = one 16-byte load
            const uint4 r4 = rraw[it];
            v[0] += __uint_as_float(r4.x << 16); v[1] += __uint_as_float(r4.x & 0xffff0000u);
            v[2] += __uint_as_float(r4.y << 16); v[3] += __uint_as_float(r4.y & 0xffff0000u);
            v[4] += __uint_as_float(r4.z << 16); v[5] += __uint_as_float(r4.z & 0xffff0000u);
            v[6] += __uint_as_float(r4.w << 16); v[7] += __uint_as_float(r4.w & 0xffff0000u);
          }
          if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          if (a.gn_partial) {
            if (gn_n == 0.f) gn_pv = 0.125f * (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[e] - gn_pv; gn_s1 += d; gn_s2 = fmaf(d, d, gn_s2); }
            gn_n += 8.f;
          }
          store8<OutT>(outn + (size_t)posv[it] * a.out_ld, v);
        }
      }
    }
    if (a.gn_partial) gn_tile_reduce<RPP, TPR>(sC, rr, c8, gn_n, gn_pv, gn_s1, gn_s2, a.gn_partial + ((size_t)mt * (a.Cout >> 3) + (n0 >> 3)) * 3, true);
    return;
  }
  // GroupNorm(32 x 8 channels) statistics of the fp32 outputs, fused: a lane's 8 channels are exactly
  // one group; per-lane shifted sums -> (count, mean, M2), combined (Chan) in a fixed order below.
  float gn_n = 0.f, gn_pv = 0.f, gn_s1 = 0.f, gn_s2 = 0.f;
  for (int p = 0; p < WGM; ++p) {
    if (p > 0) lds_barrier();
    if (wm == p) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = i * 32 + (lane & 31);
            const int col = wn * WTN + j * 32 + 8 * g + 4 * (lane >> 5);
            *reinterpret_cast<float4*>(sC + row * SCP + col) =
                make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          }
    }
    lds_barrier();
    if (p == 0 && BN <= 64 && a.ss_padded) {
      const float* ssl = reinterpret_cast<const float*>(smem + ss_off);
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = ssl[c8 * 8 + e]; sh[e] = ssl[BN + c8 * 8 + e]; }
    }
    if (!active) continue;
    for (int rl = rr; rl < WTM; rl += RPP) {
      int pos = tile.y + p * WTM + rl;
      if constexpr (HALO) {
        const int m = p * WTM + rl, my = (int)(((unsigned)m * inv_pw) >> 16);
        const int oy = oy0 + my, ox = ox0 + (m - my * HPW);
        if (m >= HPOS || oy >= sd.out_H || ox >= sd.out_W) continue;
        pos = oy * sd.out_W + ox;
      } else {
        if (pos >= seg_rows) break;
      }
      float v[8];
      {
        const float4 lo = *reinterpret_cast<const float4*>(sC + rl * SCP + c8 * 8);
        const float4 hi = *reinterpret_cast<const float4*>(sC + rl * SCP + c8 * 8 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
      if (a.res_mode != 0) {
        int rp = pos;
        if (a.res_mode == 2) {
          const int oy = pos / sd.out_W, ox = pos - oy * sd.out_W;
          rp = (oy >> 1) * sd.res_W + (ox >> 1);
        }
        const T* rptr = res + (size_t)(sd.res_row0 + rp) * a.res_ld + n0;
        if (a.res_lds) {
          float rv[8];
          const int row = p * WTM + rl, byte = c8 * 8 * (int)sizeof(T);
          load8<T>(reinterpret_cast<const T*>(smem + RES_OFF + (byte >> 7) * BM * 128 + row * 128 + (byte & 127)), rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        } else if (vec) {
          float rv[8];
          load8<T>(rptr, rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        } else {
          for (int e = 0; e < 8; ++e)
            if (n0 + e < a.Cout) v[e] += Cvt<T>::to_f(rptr[e]);
        }
      }
      if (a.mul_nch > 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (n0 + e < a.mul_nch) v[e] *= sd.mul;
      }
      if (a.relu_nch >= a.Cout) {  // channels past Cout are never stored
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      } else if (a.relu_nch > 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (n0 + e < a.relu_nch) v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      if (a.gn_partial) {  // shifted sums about the lane's first row mean (fp32-exact enough, no divisions)
        if (gn_n == 0.f) gn_pv = 0.125f * (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[e] - gn_pv; gn_s1 += d; gn_s2 = fmaf(d, d, gn_s2); }
        gn_n += 8.f;
      }
      OutT* optr = out + (size_t)(sd.out_row0 + pos) * a.out_ld + n0;
      if (vec) {
        store8<OutT>(optr, v);
      } else {
        for (int e = 0; e < 8; ++e)
          if (n0 + e < a.Cout) optr[e] = Cvt<OutT>::from_f(v[e]);
      }
    }
  }
  if (a.gn_partial)
    gn_tile_reduce<RPP, TPR>(sC, rr, c8, gn_n, gn_pv, gn_s1, gn_s2, a.gn_partial + ((size_t)mt * (a.Cout >> 3) + (n0 >> 3)) * 3, active);
}

template <typename T, typename OutT, int BM, int BN, int WGM, int WGN, int NBUF, bool FAST, bool HALO = false, bool SPLIT = false>
static int launch_cfg(const ConvArgs& a, hipStream_t s) {
  const int chunk = (a.n_mtiles + 7) / 8;
  const int grid = 8 * chunk * a.n_ntiles;
  const size_t stage = HALO ? (size_t)(BN + 184) * 128 : (size_t)NBUF * (BM + BN) * 128, epi = (size_t)(BM / WGM) * (BN + 4) * 4;
  size_t lds = stage > epi ? stage : epi;
  if (a.res_lds) lds += (size_t)BM * BN * sizeof(T);
  if (!(stage >= epi + (size_t)BN * 8)) lds += (size_t)BN * 8;  // scale/shift parked behind everything else
  if (lds > 65536) (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, OutT, BM, BN, WGM, WGN, NBUF, FAST, HALO, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  auto kern = conv_igemm_kernel<T, OutT, BM, BN, WGM, WGN, NBUF, FAST, HALO, SPLIT>;
  hipLaunchKernelGGL(kern, dim3(grid, a.ksplit > 1 ? a.ksplit : 1), dim3(WGM * WGN * 64), lds, s, a);
  return (int)hipGetLastError();
}

// Second half of a split-K conv: out[dst row][n] = epilogue(sum over the planes, in plane order, of partial[plane][src row][n]) --
// v = fma(sum, scale, shift) (+ residual of the output geometry) (ReLU) -> bf16, the rounding points of the unsplit epilogue.
// One thread = 8 channels of one row; blockIdx.y = segment.
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ partial, int ksplit, size_t plane, int ld, int Cout,
                                                            const SplitSeg* __restrict__ segs, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const bf16_t* __restrict__ res, int res_ld,
                                                            int relu_nch, bf16_t* __restrict__ out, int out_ld) {
  const SplitSeg sg = segs[blockIdx.y];
  const int cpr = Cout >> 3;  // 8-channel chunks per row
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int row = idx / cpr, n0 = (idx - row * cpr) * 8;
  if (row >= sg.nrows) return;
  const float* src = partial + (size_t)(sg.src_row0 + row) * ld + n0;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  for (int k = 0; k < ksplit; ++k) {
    const float4 lo = *reinterpret_cast<const float4*>(src + k * plane), hi = *reinterpret_cast<const float4*>(src + k * plane + 4);
    v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = v[e] * (scale ? scale[n0 + e] : 1.f) + (shift ? shift[n0 + e] : 0.f);
  if (res) {
    float rv[8];
    load8<bf16_t>(res + (size_t)(sg.res_row0 + row) * res_ld + n0, rv);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += rv[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (n0 + e < relu_nch) v[e] = v[e] > 0.f ? v[e] : 0.f;
  store8<bf16_t>(out + (size_t)(sg.dst_row0 + row) * out_ld + n0, v);
}

int launch_splitk_finish(const float* partial, int ksplit, size_t plane, int ld, int Cout, const SplitSeg* segs_dev, int nseg, int max_rows,
                         const float* scale, const float* shift, const void* res, int res_ld, int relu_nch, void* out, int out_ld, hipStream_t s) {
  if ((Cout & 7) || (ld & 3) || (out_ld & 7) || (res && (res_ld & 7)) || nseg < 1) return -1;
  const long n = (long)max_rows * (Cout >> 3);
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((n + 255) / 256), nseg), dim3(256), 0, s, partial, ksplit, plane, ld, Cout, segs_dev, scale,
                     shift, (const bf16_t*)res, res_ld, relu_nch, (bf16_t*)out, out_ld);
  return (int)hipGetLastError();
}

#ifdef SYLPH_ABLATE
static int g_nbuf = 1;  // A/B knob: a second LDS stage (rejected)
void conv_set_nbuf(int n) { g_nbuf = n == 2 ? 2 : 1; }
#else
static constexpr int g_nbuf = 1;
void conv_set_nbuf(int) {}
#endif

template <typename T, typename OutT, int NBUF, bool FAST>
static int launch_n(const ConvArgs& a, int BM, int BN, hipStream_t s) {
#ifdef SYLPH_ABLATE  // reachable only through SYLPH_CONV_FORCE_BM / _BN
  if (BM == 256 && BN == 128) return launch_cfg<T, OutT, 256, 128, 2, 2, NBUF, FAST>(a, s);
  if (BM == 128 && BN == 256) return launch_cfg<T, OutT, 128, 256, 2, 2, NBUF, FAST>(a, s);
#endif
  if (BM == 128 && BN == 128) return launch_cfg<T, OutT, 128, 128, 2, 2, NBUF, FAST>(a, s);
  if (BM == 128 && BN == 64) return launch_cfg<T, OutT, 128, 64, 2, 2, NBUF, FAST>(a, s);
  if (BM == 128 && BN == 32) return launch_cfg<T, OutT, 128, 32, 4, 1, NBUF, FAST>(a, s);
  if (BM == 64 && BN == 128) return launch_cfg<T, OutT, 64, 128, 2, 2, NBUF, FAST>(a, s);
  if (BM == 64 && BN == 64) return launch_cfg<T, OutT, 64, 64, 2, 2, NBUF, FAST>(a, s);
  return -1;
}

// FAST epilogue: see the kernel.  Only the production dtype (bf16 in, bf16 out, single stage) gets it.
static bool fast_ok(const ConvArgs& a, int BN) {
  static const int on = SYLPH_AB_ENV("SYLPH_CONV_FAST", 1);
  return on && a.ss_padded_host && (BN > 64 || a.ss_padded) && a.Cout % BN == 0 && (a.out_ld & 7) == 0 && (a.res_mode == 0 || (a.res_ld & 7) == 0) &&
         a.mul_nch == 0 && (a.relu_nch == 0 || a.relu_nch >= a.Cout);
}

template <typename T, typename OutT>
static int launch_t(const ConvArgs& a, int BM, int BN, hipStream_t s) {
#ifdef SYLPH_ABLATE
  if (g_nbuf == 2) return launch_n<T, OutT, 2, false>(a, BM, BN, s);
#endif
  return launch_n<T, OutT, 1, false>(a, BM, BN, s);
}

// Tile choice: widest N tile the layer fills (MFMA-bound 3x3 convs); HBM-bound pointwise convs
// prefer 128x64 (fewer registers -> more co-resident blocks -> more loads in flight: measured
// 3-5 % faster on the bottleneck 1x1 layers); drop to BM=64 when the grid would not fill 256 CUs.
void conv_pick_tile(int rows_total, int cout, int ntaps, int* BM, int* BN) {
  int bn = cout >= 128 ? 128 : (cout > 32 ? 64 : 32);
  // (pointwise convs used to prefer 128x64 for occupancy; since the per-tile instruction diet 128x128 is equal or
  // better on every bottleneck 1x1: less LDS-DMA traffic per flop)
#ifdef SYLPH_ABLATE
  if (const char* f = getenv("SYLPH_CONV_FORCE_BN")) {  // tuning knob
    const int v = atoi(f);
    if ((v == 64 || v == 128 || v == 256) && cout % v == 0) bn = v;
  }
#endif
  int bm = 128;
  if (bn != 32) {
    const long blocks128 = (long)((rows_total + 127) / 128) * ((cout + bn - 1) / bn);
    if (blocks128 < 1024) bm = 64;
    // Round 6: 64 x 64 tiles when even the 64 x 128 grid leaves a third of the CUs idle (<= 160 tiles: res4 / res5 of one 800 x 1333
    // image): twice the blocks, half the weight stage per block.  From 264 tiles on (res3 at batch 1, res4 at batch 2) measured slower.
    static const int bn64_max = getenv("SYLPH_CONV_BN64_MAX") ? atoi(getenv("SYLPH_CONV_BN64_MAX")) : 160;
    if (bm == 64 && bn == 128 && cout % 64 == 0 && (long)((rows_total + 63) / 64) * (cout / 128) <= bn64_max) bn = 64;
  }
#ifdef SYLPH_ABLATE
  if (const char* f = getenv("SYLPH_CONV_FORCE_BM")) {  // tuning knob
    const int v = atoi(f);
    if ((v == 64 || v == 128 || v == 256) && bn != 32) bm = v;
  }
#endif
  if (bn == 256) bm = 128;
  if (bm == 256 && bn != 128) bm = 128;
  *BM = bm;
  *BN = bn;
}

int launch_conv(DType dt, bool out_f32, const ConvArgs& a_in, int BM, int BN, hipStream_t s) {
  ConvArgs a = a_in;
  // stage the residual through LDS on the narrow (HBM-bound) tiles; measured 1.4 % slower (LDS occupancy 5 -> 4 blocks), so off; SYLPH_CONV_RES_LDS=1 enables
  static const int res_lds_on = SYLPH_AB_ENV("SYLPH_CONV_RES_LDS", 0);
  static const int ss_on = SYLPH_AB_ENV("SYLPH_CONV_SS_LDS", 1);
  a.res_lds = (res_lds_on && a.res_mode != 0 && BN == 64 && a.Cout % 64 == 0 && (a.res_ld & 7) == 0) ? 1 : 0;
  a.ss_padded_host = a.ss_padded;
  if (!ss_on || BN > 64) a.ss_padded = 0;  // the wide tiles are MFMA-bound and have no VGPRs to spare for the prefetch
  if (BM == 256 && BN == 256) return conv_hpipe_ok(dt, out_f32, a) ? launch_conv_hpipe(a, s) : -8;
  if (a.KH * a.KW > 31) return -3;
  const int bk = dt == DT_BF16 ? 64 : 32;
  if (a.Cin % bk != 0) return -4;
  if (a.in2 && (a.Cin2 % bk != 0 || a.KH * a.KW != 1)) return -6;
  if (!a.zeros) return -5;
  if (a.ksplit > 1 && (a.halo || a.in2 || !out_f32 || a.stem || a.gn_partial || a.res_mode != 0 || g_nbuf != 1)) return -10;
  if (dt == DT_BF16) {
    if (a.halo) {  // 3x3 s1 p1 with patch tiles (api_conv.hip builds the matching tile table)
      if (g_nbuf != 1 || BM != 128 || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.stem || a.in2) return -9;
      if (BN == 32) {  // narrow prediction convs (bbox/ctrness, code-generator heads): fp32 or bf16 out
        return out_f32 ? launch_cfg<bf16_t, float, 128, 32, 4, 1, 1, false, true>(a, s)
                       : launch_cfg<bf16_t, bf16_t, 128, 32, 4, 1, 1, false, true>(a, s);
      }
      if (out_f32 || (BN != 128 && BN != 64)) return -9;
      const bool fast = fast_ok(a, BN);
      if (BN == 128) return fast ? launch_cfg<bf16_t, bf16_t, 128, 128, 2, 2, 1, true, true>(a, s)
                                 : launch_cfg<bf16_t, bf16_t, 128, 128, 2, 2, 1, false, true>(a, s);
      return fast ? launch_cfg<bf16_t, bf16_t, 128, 64, 2, 2, 1, true, true>(a, s)
                  : launch_cfg<bf16_t, bf16_t, 128, 64, 2, 2, 1, false, true>(a, s);
    }
#define SYLPH_IGEMM_RING(NB)                                                                                                                       \
  do {                                                                                                                                             \
    const bool fast = !out_f32 && fast_ok(a, BN);                                                                                                  \
    if (BN == 128) return out_f32 ? launch_cfg<bf16_t, float, 64, 128, 2, 2, NB, false>(a, s)                                                      \
                                  : (fast ? launch_cfg<bf16_t, bf16_t, 64, 128, 2, 2, NB, true>(a, s) : launch_cfg<bf16_t, bf16_t, 64, 128, 2, 2, NB, false>(a, s)); \
    return out_f32 ? launch_cfg<bf16_t, float, 64, 64, 2, 2, NB, false>(a, s)                                                                      \
                   : (fast ? launch_cfg<bf16_t, bf16_t, 64, 64, 2, 2, NB, true>(a, s) : launch_cfg<bf16_t, bf16_t, 64, 64, 2, 2, NB, false>(a, s)); \
  } while (0)
    if (a.nbuf2 == 3 && g_nbuf == 1 && BM == 64 && (BN == 128 || BN == 64)) SYLPH_IGEMM_RING(3);
#undef SYLPH_IGEMM_RING
    if (a.nbuf2 && g_nbuf == 1 && BM == 64 && (BN == 128 || BN == 64)) {
      // small launches (fewer blocks than it takes to hide a slice's round trip by occupancy): two LDS stages, the next slice's loads
      // in flight under the current slice's MFMAs
      const bool fast = !out_f32 && fast_ok(a, BN);
      if (BN == 128) return out_f32 ? launch_cfg<bf16_t, float, 64, 128, 2, 2, 2, false>(a, s)
                                    : (fast ? launch_cfg<bf16_t, bf16_t, 64, 128, 2, 2, 2, true>(a, s) : launch_cfg<bf16_t, bf16_t, 64, 128, 2, 2, 2, false>(a, s));
      return out_f32 ? launch_cfg<bf16_t, float, 64, 64, 2, 2, 2, false>(a, s)
                     : (fast ? launch_cfg<bf16_t, bf16_t, 64, 64, 2, 2, 2, true>(a, s) : launch_cfg<bf16_t, bf16_t, 64, 64, 2, 2, 2, false>(a, s));
    }
    if (!out_f32 && g_nbuf == 1 && fast_ok(a, BN)) return launch_n<bf16_t, bf16_t, 1, true>(a, BM, BN, s);
    return out_f32 ? launch_t<bf16_t, float>(a, BM, BN, s) : launch_t<bf16_t, bf16_t>(a, BM, BN, s);
  }
  if (dt == DT_F32S) {  // split-bf16 parity mode: fp32 storage, three bf16 MFMAs per product (MmaSplit)
    if (BM == 128 && BN == 128) return launch_cfg<float, float, 128, 128, 2, 2, 1, false, false, true>(a, s);
    if (BM == 128 && BN == 64) return launch_cfg<float, float, 128, 64, 2, 2, 1, false, false, true>(a, s);
    if (BM == 128 && BN == 32) return launch_cfg<float, float, 128, 32, 4, 1, 1, false, false, true>(a, s);
    if (BM == 64 && BN == 128) return launch_cfg<float, float, 64, 128, 2, 2, 1, false, false, true>(a, s);
    if (BM == 64 && BN == 64) return launch_cfg<float, float, 64, 64, 2, 2, 1, false, false, true>(a, s);
    return -1;
  }
  return launch_t<float, float>(a, BM, BN, s);
}

}  // namespace sylph
