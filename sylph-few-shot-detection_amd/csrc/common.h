// Shared device/host declarations for the Sylph MI355X (gfx950) inference library.
// Activations live in HBM as position-major ("NHWC") rows: one row = one spatial position,
// C contiguous channels, either bf16 (throughput mode) or fp32 (parity mode).  A feature map is a
// list of SEGMENTS (one per image, or one per image x FPN level); segment s owns rows
// [row0, row0 + H*W).  Every kernel takes its geometry from a small device-resident table.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

// A/B knobs of variants that were measured and rejected exist only in builds made with -DSYLPH_ABLATE
// (tools/build_variant.sh): the product library reads no environment for them and carries no instantiation of them.
#ifdef SYLPH_ABLATE
#include <stdlib.h>
#define SYLPH_AB_ENV(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define SYLPH_AB_ENV(name, dflt) (dflt)
#endif

namespace sylph {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// DT_F32S: fp32 storage like DT_F32 (every kernel that only asks "bf16 or not" treats it as fp32); the convs split their operands into
// bf16 hi + lo parts and run three bf16 MFMAs per product (conv_igemm.hip MmaSplit)
enum DType { DT_F32 = 0, DT_BF16 = 1, DT_F32S = 2 };

// One segment of a convolution problem.  All *_row0 are row indices into the respective buffers.
struct SegDesc {
  int in_row0, in_H, in_W;
  int out_row0, out_H, out_W;
  int res_row0, res_H, res_W;  // residual source geometry (res_mode 2: half resolution)
  float mul;                   // per-segment multiplier (FCOS Scale_l), applied to channels < mul_nch
  int in2_row0, in2_W;         // second input (dual-source pointwise conv): first row, row width
  int ph, pw;                  // halo-tile mode: patch height / width of this segment (ph * pw <= M tile rows)
  int hpitch;                  // halo-tile mode: LDS rows per halo row (pw + 2; conv_hpipe.hip: pw + 4, see there)
  unsigned inv_pw, inv_hw2;    // ceil(65536 / pw), ceil(65536 / hpitch): n / d == (n * inv) >> 16 for n < 256
};

struct ConvArgs {
  const void* in;
  const void* wt;     // packed [Cout_pad][KH][KW][Cin] in the compute dtype
  void* out;
  const void* res;    // optional residual (compute dtype), row stride res_ld
  const float* scale; // per-output-channel epilogue scale (nullptr -> 1)
  const float* shift; // per-output-channel epilogue shift (nullptr -> 0)
  const void* in2;    // optional second input of a dual-source pointwise conv (see conv_igemm.hip)
  const void* zeros;  // >= 128 B of zeros: source of out-of-image taps (global_load_lds cannot predicate)
  const SegDesc* segs;
  const int2* tiles;  // tiles[t] = {segment, first row of the tile inside the segment}
  int n_mtiles, n_ntiles;
  int Cin, Cout, KH, KW, stride, pad;
  int in_ld, out_ld, res_ld;  // row strides in elements
  int Cin2, in2_ld, stride2;  // second input: channels, row stride, spatial stride
  int group_cout, group_in_off;  // grouped conv: output channels per group, input-channel offset per group (0 = off)
  int relu_nch;               // ReLU on output channels < relu_nch
  int mul_nch;                // seg.mul on output channels < mul_nch
  int res_mode;               // 0 none, 1 same geometry, 2 nearest-neighbour 2x upsample of res
  float* gn_partial;          // optional [n_mtiles][Cout/8][3] per-tile GroupNorm partials (n, mean, M2)
  int halo;                   // 3x3 s1 p1 halo-tile mode: tiles[].y = (patch row << 16) | patch col (seg.ph x seg.pw patches)
  int res_lds;                // set by launch_conv: residual tile staged through LDS
  int ss_padded_host;         // as given by the caller (ss_padded is cleared for wide tiles)
  int ss_padded;              // scale/shift arrays are padded to a multiple of the N tile (vector prefetch allowed)
  int stem;                   // ResNet-stem A loader (see conv_igemm.hip)
  int tap_dy;                 // input rows advanced per kernel-row tap (1; stem: rows per K-slice)
  const float2* gn_coef;      // conv_hpipe.hip: fused GroupNorm(+ReLU) of the INPUT: per (segment, input channel) (a, b),
  int gn_relu;                //   x <- relu?(a * x + b) applied to the landed halo in LDS; row stride of gn_coef = in_ld
  void* trash;                // conv_pw.hip: >= 4 KiB scratch where lanes without a valid output row put their (fixed number of) stores
  const struct PwDesc* pw_desc;  // conv_pw.hip: one 64-byte descriptor per M tile (read through the scalar cache)
  const float* pw_table;      // conv_pw.hip: [n_ntiles][scale BN | shift BN] fp32
  int pw_rot_mask;            // set by launch_conv_pw
  int ksplit;                 // conv_igemm.hip split K: > 1 = grid.y K ranges, fp32 partial planes in `out` (plane stride split_stride elements)
  long long split_stride;
  int nbuf2;                  // conv_igemm.hip: LDS stages of the K loop of small launches (64-row tiles): 0 = one, 2, 3
};

// one segment of a split-K finish pass: rows [src_row0, +nrows) of the partial planes -> rows [dst_row0, ..) of the output
struct SplitSeg { int src_row0, dst_row0, res_row0, nrows; };

// conv_pw.hip tile descriptor (two s_load_dwordx8): geometry of the tile's segment + the tile's first row
struct PwDesc { int row0, seg_rows, out_W, out_row0, in_row0, in_W, in2_row0, in2_W, res_row0, res_W, pad[6]; };

// fused identity bottleneck (bottleneck.hip): x, y [pos][256] bf16; w1 [64][256], w2 [64][3][3][64], w3 [256][64] bf16
// (the conv_igemm weight layouts); FrozenBN scale / shift per conv (fp32)
struct BkTile { int row0, H, W, yx, ph, pw; unsigned inv_pw, inv_hw2; };  // one 32-byte descriptor per patch (s_load_dwordx8)
struct BottleneckArgs {
  const void* x;
  void* y;
  const __bf16 *w1, *w2, *w3;
  const float *s1, *b1, *s2, *b2, *s3, *b3;
  const void* zeros;
  void* trash;  // >= 256 CUs x 256 threads x 128 B: where lanes without a valid output position put their (fixed number of) stores
  const BkTile* bk;
  int n_tiles;
};

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
  static __device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
  static __device__ __forceinline__ bf16_t from_f(float v) { return (bf16_t)v; }
};

__device__ __forceinline__ float bf16_bits_to_f(uint32_t hi16) { return __uint_as_float(hi16 << 16); }

// load 8 consecutive elements of T as floats (16-byte aligned for bf16, 32-byte span for f32)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x8*>(p) = o;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for every global
// store (and prefetch load) still in flight: ~1-2 us per barrier in an epilogue that has just issued its stores.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Per-device launch set-up of the persistent kernels (function-local statics took the CU count and the LDS attribute from
// whichever device launched first, unsynchronised).  Both are keyed by the CURRENT device; racing threads at worst repeat an idempotent call.
inline int current_device() {
  int dev = 0;
  return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) ? dev : 0;
}
inline int device_cu_count(int dev) {
  static std::atomic<int> cus[64];
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    hipDeviceProp_t p;
    n = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
struct PerDeviceOnce {  // `static PerDeviceOnce once;  if (!once.run(dev, [&] { return hipFuncSetAttribute(...) == hipSuccess; })) return -7;`
  std::atomic<unsigned long long> done{0};
  template <typename F> bool run(int dev, F&& f) {
    const unsigned long long bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return true;
    if (!f()) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
  }
};

// ---- launcher prototypes (one per translation unit) -----------------------------------------
// conv_igemm.hip
int launch_conv(DType dt, bool out_f32, const ConvArgs& a, int BM, int BN, hipStream_t s);
void conv_pick_tile(int rows_total, int cout, int ntaps, int* BM, int* BN);
int launch_splitk_finish(const float* partial, int ksplit, size_t plane, int ld, int Cout, const SplitSeg* segs_dev, int nseg, int max_rows,
                         const float* scale, const float* shift, const void* res, int res_ld, int relu_nch, void* out, int out_ld, hipStream_t s);
// conv_hpipe.hip: 256x256 deep-pipelined halo-operand 3x3 kernel (BM == BN == 256 selects it in launch_conv; the tile
// table then holds PAIRS of patches and n_mtiles counts the pairs)
bool conv_hpipe_ok(DType dt, bool out_f32, const ConvArgs& a);
int launch_conv_hpipe(const ConvArgs& a, hipStream_t s);
int launch_hpipe_pack_weights(const void* w_igemm, void* w_hpipe, int Cout, int Cin, hipStream_t s);  // a.wt of an hpipe launch
int launch_bottleneck64(const BottleneckArgs& a, int small, hipStream_t s);   // bottleneck.hip: identity block, persistent, weights in registers; small: 64-position patches, double-buffered halo
int launch_bottleneck64p(const BottleneckArgs& a, hipStream_t s);  // first block of res2: x [pos][64], w3 = [256][128] packed [W3 | Wsc], y = relu(acc + b3)
int launch_conv_rw3(const BottleneckArgs& a, hipStream_t s);  // conv_rw3.hip: 3x3 s1 128 -> 128 + FrozenBN + ReLU, weights in registers (x, y, w2, s2, b2, bk, n_tiles)
bool conv_rw3_patch_ok(int ph, int pw);
void conv_set_nbuf(int n);  // 1: single LDS stage (max occupancy), 2: double-buffered
// conv_pw.hip: persistent pipelined pointwise (1x1) conv, bf16; a.wt = the layer's stage-image weights (launch_pw_pack_weights),
// a.tiles = BM-row tiles, a.pw_desc / a.pw_table as below; (BM, BN) from conv_pw_tile (false: not eligible)
bool conv_pw_tile(int cout, int k_total, bool has_res, int* BM, int* BN);
bool conv_pw_ok(DType dt, bool out_f32, const ConvArgs& a);
int launch_conv_pw(const ConvArgs& a, int BM, int BN, hipStream_t s);
int launch_pw_pack_weights(const void* w_igemm, void* w_pw, int Cout, int K, int BN, hipStream_t s);
// conv_spw.hip: streaming pointwise conv with a same-geometry residual (bottleneck conv3 of identity blocks): conv_pw's operands
// (stage-image weights for BN = 256, table, one PwDesc per 128-row M tile), four MFMA waves + four streaming waves per CU
bool conv_spw_ok(DType dt, bool out_f32, const ConvArgs& a);
int launch_conv_spw(const ConvArgs& a, hipStream_t s);
int launch_pw_pack_table(const float* scale, const float* shift, float* out, int Cout, int BN, hipStream_t s);

}  // namespace sylph
