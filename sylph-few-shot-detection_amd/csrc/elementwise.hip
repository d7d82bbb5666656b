// HBM-bound kernels of the Sylph inference path (gfx950): pixel normalisation + padding,
// 3x3 s2 max-pool, row-wise ReLU copy,
// GroupNorm(32 groups x 8 channels) statistics / apply(+ReLU), layout import/export.
// All activation traffic is 16 bytes per lane (8 bf16 or 2x float4), rows are channel-contiguous.
//
// Reference ops replaced (paths relative to /root/reference):
//   preprocess  : sylph/modeling/meta_arch/meta_one_stage_detector.py:174-178 (+ d2 ImageList.from_tensors)
//   maxpool     : detectron2 BasicStem max_pool2d(3, 2, 1), via meta_one_stage_detector.py:181,273
//   relu_rows   : the F.relu(p6) feeding P7 in AdelaiDet LastLevelP6P7
//   GroupNorm   : nn.GroupNorm(32, C) in sylph/modeling/meta_fcos/fcos.py:97-98 and
//                 sylph/modeling/code_generator/code_generator.py:648-688 (build_fpn_norm "GN")
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace sylph {

// ------------------------------------------------------------------------------------------------
// preprocess: list of (3,h,w) fp32 planes -> [B][H][W][4] (4th channel = 0), zero padded.
template <typename T>
__global__ void preprocess_kernel(const ImageDesc* imgs, T* out, int H, int W, float m0, float m1, float m2,
                                  float is0, float is1, float is2) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= W) return;
  const ImageDesc d = imgs[b];
  float v0 = 0.f, v1 = 0.f, v2 = 0.f;
  if (y < d.h && x < d.w) {
    const size_t plane = (size_t)d.h * d.w;
    const float* p = d.ptr + (size_t)y * d.w + x;
    v0 = (p[0] - m0) * is0;
    v1 = (p[plane] - m1) * is1;
    v2 = (p[2 * plane] - m2) * is2;
  }
  T* o = out + (((size_t)b * H + y) * W + x) * 4;
  o[0] = Cvt<T>::from_f(v0); o[1] = Cvt<T>::from_f(v1); o[2] = Cvt<T>::from_f(v2); o[3] = Cvt<T>::from_f(0.f);
}

int launch_preprocess(DType dt, const ImageDesc* imgs_dev, void* out, int B, int H, int W, const float* mean,
                      const float* stdv, hipStream_t s) {
  dim3 grid((W + 255) / 256, H, B), block(256);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(preprocess_kernel<bf16_t>, grid, block, 0, s, imgs_dev, (bf16_t*)out, H, W, mean[0], mean[1],
                       mean[2], 1.f / stdv[0], 1.f / stdv[1], 1.f / stdv[2]);
  else
    hipLaunchKernelGGL(preprocess_kernel<float>, grid, block, 0, s, imgs_dev, (float*)out, H, W, mean[0], mean[1],
                       mean[2], 1.f / stdv[0], 1.f / stdv[1], 1.f / stdv[2]);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Fused input pipeline: uint8 HWC image -> ResizeShortestEdge target size with PIL's BILINEAR
// resampling (what detectron2's ResizeTransform applies to uint8 images: sylph/predictor.py:117-120,259-269) -> optional
// RGB->BGR -> (x - mean) / std -> zero pad -> [B][H][W][4] in the compute dtype, in ONE pass.
// Pillow semantics (Resample.c, 8 bits per channel), reproduced bit for bit: two separable passes, horizontal first, each
// with a triangle filter whose support is scaled by the down-sampling factor (antialiasing), coefficients normalised per
// output pixel and rounded to 22-bit fixed point, the horizontal result ROUNDED TO uint8 before the vertical pass.
// The coefficient tables are built on the host in double exactly as Pillow does (api_backbone.hip pil_bilinear_coeffs); this
// kernel does the integer arithmetic: out = clip8((2^21 + sum_j kv[j] * clip8((2^21 + sum_i kh[i] * px) >> 22)) >> 22).
__device__ __forceinline__ int clip8(int v) {
  v >>= 22;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

template <typename T>
__global__ void resize_preprocess_kernel(const ResizeDesc* descs, const int* __restrict__ tab, T* out, int H, int W, float m0,
                                         float m1, float m2, float is0, float is1, float is2, int rgb_input) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= W) return;
  const ResizeDesc d = descs[b];
  float v0 = 0.f, v1 = 0.f, v2 = 0.f;
  if (y < d.new_h && x < d.new_w) {
    const int xmin = tab[d.hb_off + 2 * x], xn = tab[d.hb_off + 2 * x + 1];
    const int ymin = tab[d.vb_off + 2 * y], yn = tab[d.vb_off + 2 * y + 1];
    const int* kh = tab + d.hk_off + x * d.ksh;
    const int* kv = tab + d.vk_off + y * d.ksv;
    int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
    for (int j = 0; j < yn; ++j) {
      const unsigned char* row = d.src + ((size_t)(ymin + j) * d.w + xmin) * 3;
      int h0 = 1 << 21, h1 = 1 << 21, h2 = 1 << 21;
      for (int i = 0; i < xn; ++i) {
        const int k = kh[i];
        h0 += row[3 * i] * k; h1 += row[3 * i + 1] * k; h2 += row[3 * i + 2] * k;
      }
      const int w = kv[j];
      a0 += clip8(h0) * w; a1 += clip8(h1) * w; a2 += clip8(h2) * w;
    }
    int c0 = clip8(a0), c1 = clip8(a1), c2 = clip8(a2);
    if (rgb_input) { const int t = c0; c0 = c2; c2 = t; }  // the model works in BGR (cfg.INPUT.FORMAT, predictor.py:259-262)
    v0 = ((float)c0 - m0) * is0;
    v1 = ((float)c1 - m1) * is1;
    v2 = ((float)c2 - m2) * is2;
  }
  T* o = out + (((size_t)b * H + y) * W + x) * 4;
  o[0] = Cvt<T>::from_f(v0); o[1] = Cvt<T>::from_f(v1); o[2] = Cvt<T>::from_f(v2); o[3] = Cvt<T>::from_f(0.f);
}

int launch_resize_preprocess(DType dt, const ResizeDesc* descs_dev, const int* tab_dev, void* out, int B, int H, int W,
                             const float* mean, const float* stdv, int rgb_input, hipStream_t s) {
  dim3 grid((W + 127) / 128, H, B), block(128);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(resize_preprocess_kernel<bf16_t>, grid, block, 0, s, descs_dev, tab_dev, (bf16_t*)out, H, W, mean[0], mean[1],
                       mean[2], 1.f / stdv[0], 1.f / stdv[1], 1.f / stdv[2], rgb_input);
  else
    hipLaunchKernelGGL(resize_preprocess_kernel<float>, grid, block, 0, s, descs_dev, tab_dev, (float*)out, H, W, mean[0], mean[1],
                       mean[2], 1.f / stdv[0], 1.f / stdv[1], 1.f / stdv[2], rgb_input);
  return (int)hipGetLastError();
}

// [B][H][W][4] compute dtype -> (B,3,H,W) fp32 NCHW (test boundary: the preprocessed network input)
template <typename T>
__global__ void export_input_kernel(const T* __restrict__ x, float* __restrict__ out, int H, int W) {
  const int b = blockIdx.z, y = blockIdx.y, xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= W) return;
  const T* p = x + (((size_t)b * H + y) * W + xx) * 4;
  for (int c = 0; c < 3; ++c) out[(((size_t)b * 3 + c) * H + y) * W + xx] = Cvt<T>::to_f(p[c]);
}

int launch_export_input(DType dt, const void* x, float* out, int B, int H, int W, hipStream_t s) {
  dim3 grid((W + 127) / 128, H, B), block(128);
  if (dt == DT_BF16) hipLaunchKernelGGL(export_input_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, out, H, W);
  else hipLaunchKernelGGL(export_input_kernel<float>, grid, block, 0, s, (const float*)x, out, H, W);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// max-pool 3x3 stride 2 pad 1 over [B][H][W][C]; one thread = 8 channels of one output position.
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C, int Ho,
                               int Wo) {
  const int cg = C / 8;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * Ho * Wo * cg;
  if (idx >= total) return;
  const int c8 = (int)(idx % cg);
  size_t p = idx / cg;
  const int ox = (int)(p % Wo); p /= Wo;
  const int oy = (int)(p % Ho);
  const int b = (int)(p / Ho);
  float m[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = oy * 2 - 1 + dy;
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ox * 2 - 1 + dx;
      if ((unsigned)ix >= (unsigned)W) continue;
      float v[8];
      load8<T>(in + (((size_t)b * H + iy) * W + ix) * C + c8 * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
    }
  }
  store8<T>(out + (((size_t)b * Ho + oy) * Wo + ox) * C + c8 * 8, m);
}

int launch_maxpool(DType dt, const void* in, void* out, int B, int H, int W, int C, int Ho, int Wo, hipStream_t s) {
  const size_t total = (size_t)B * Ho * Wo * (C / 8);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(maxpool_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)in, (bf16_t*)out, B, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool_kernel<float>, grid, block, 0, s, (const float*)in, (float*)out, B, H, W, C, Ho, Wo);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(32, 256): 8 channels per group == one 16-byte (bf16) lane load.
// Pass 1: block (chunk, seg): 256 threads = 32 groups x 8 row-lanes; shifted sums per thread, Chan
// combine across the 8 row-lanes in LDS -> partial (n, mean, M2) per (seg, chunk, group).
// Pass 2: one thread per (seg, group) folds the chunks in a fixed order (deterministic) -> mean, rstd.
// Pass 3: y = relu((x - mean) * rstd * gamma + beta), in place.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, const RowSeg* segs, int ld,
                                                       int rows_per_chunk, int max_chunks,
                                                       float* __restrict__ partial) {
  const int seg = blockIdx.y, chunk = blockIdx.x;
  const RowSeg sg = segs[seg];
  const int r_begin = chunk * rows_per_chunk;
  if (r_begin >= sg.nrows) return;
  const int r_end = min(sg.nrows, r_begin + rows_per_chunk);
  const int g = threadIdx.x & 31, rl = threadIdx.x >> 5;
  float n = 0.f, s1 = 0.f, s2 = 0.f, K = 0.f;
  bool first = true;
  for (int r = r_begin + rl; r < r_end; r += 8) {
    float v[8];
    load8<T>(x + (size_t)(sg.row0 + r) * ld + g * 8, v);
    if (first) { K = v[0]; first = false; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = v[j] - K;
      s1 += d;
      s2 = fmaf(d, d, s2);
    }
    n += 8.f;
  }
  float mean = 0.f, m2 = 0.f;
  if (n > 0.f) {
    const float md = s1 / n;
    mean = K + md;
    m2 = fmaxf(s2 - s1 * md, 0.f);
  }
  __shared__ float sh[3][8][32];
  sh[0][rl][g] = n; sh[1][rl][g] = mean; sh[2][rl][g] = m2;
  __syncthreads();
  if (threadIdx.x < 32) {
    float N = 0.f, M = 0.f, Q = 0.f;
    for (int i = 0; i < 8; ++i) {
      const float nb = sh[0][i][g];
      if (nb > 0.f) {
        const float mb = sh[1][i][g], qb = sh[2][i][g];
        const float nn = N + nb, delta = mb - M;
        M += delta * (nb / nn);
        Q += qb + delta * delta * (N * nb / nn);
        N = nn;
      }
    }
    float* p = partial + (((size_t)seg * max_chunks + chunk) * 32 + g) * 3;
    p[0] = N; p[1] = M; p[2] = Q;
  }
}

__global__ void gn_finalize_kernel(const float* __restrict__ partial, const RowSeg* segs, int nseg,
                                   int rows_per_chunk, int max_chunks, float eps, float2* __restrict__ stats) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nseg * 32) return;
  const int seg = idx >> 5, g = idx & 31;
  const int nchunks = (segs[seg].nrows + rows_per_chunk - 1) / rows_per_chunk;
  double N = 0.0, M = 0.0, Q = 0.0;
  for (int c = 0; c < nchunks; ++c) {
    const float* p = partial + (((size_t)seg * max_chunks + c) * 32 + g) * 3;
    const double nb = p[0], mb = p[1], qb = p[2];
    if (nb > 0.0) {
      const double nn = N + nb, delta = mb - M;
      M += delta * (nb / nn);
      Q += qb + delta * delta * (N * nb / nn);
      N = nn;
    }
  }
  const double var = N > 0.0 ? Q / N : 0.0;
  stats[idx] = make_float2((float)M, (float)(1.0 / sqrt(var + (double)eps)));
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(T* __restrict__ x, const RowSeg* segs, int ld,
                                                       int rows_per_chunk, const float2* __restrict__ stats,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu) {
  const int seg = blockIdx.y, chunk = blockIdx.x;
  const RowSeg sg = segs[seg];
  const int r_begin = chunk * rows_per_chunk;
  if (r_begin >= sg.nrows) return;
  const int r_end = min(sg.nrows, r_begin + rows_per_chunk);
  const int g = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const float2 st = stats[seg * 32 + g];
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = st.y * gamma[g * 8 + j];
    b[j] = beta[g * 8 + j] - st.x * a[j];
  }
  for (int r = r_begin + rl; r < r_end; r += 8) {
    T* p = x + (size_t)(sg.row0 + r) * ld + g * 8;
    float v[8];
    load8<T>(p, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = fmaf(v[j], a[j], b[j]);
      if (relu) t = t > 0.f ? t : 0.f;
      v[j] = t;
    }
    store8<T>(p, v);
  }
}

// Fused variant: the conv epilogue already left (n, mean, M2) per (M-tile, group); every block folds
// the partials of its segment in tile order (deterministic, fp64) and applies the affine (+ReLU).
// ngroups = C/8 (32 for one tower, 64 for the paired cls|bbox towers held side by side).
// Finalize: one block per segment folds the conv epilogue's per-tile (n, mean, M2) partials of every group in a
// fixed order (fp64 Chan merges: 1024/ngroups interleaved chains per group, then the chains in index order) and
// leaves (mean, rstd) per (segment, group).  Done once per layer: the fp64 chain (two divisions per tile) is far
// too slow to repeat in every block of the streaming pass.
// With `coef`: also the per (segment, channel) (a, b) of y = a * x + b (the apply pass of the NEXT conv: conv_hpipe.hip
// transforms its input halo with these instead of a separate streaming pass over the tensor) -- one launch, not two.
__global__ __launch_bounds__(1024) void gn_finalize_partials_kernel(const GnSeg* segs, int ngroups, const float* __restrict__ partial,
                                                                   float eps, float2* __restrict__ stats,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float2* __restrict__ coef) {
  const GnSeg sg = segs[blockIdx.x];
  const int g = threadIdx.x % ngroups, sub = threadIdx.x / ngroups, nsub = 1024 / ngroups;
  __shared__ double sh[1024 * 3];
  double N = 0.0, M = 0.0, Q = 0.0;
  // the chain's first PF partials are fetched up front (one memory round trip instead of one per link: a 140-tile P3 segment is 5 links
  // per chain, and at small batches this launch sits on the critical path between two tower layers); same links, same order
  constexpr int PF = 8;
  float pn[PF], pm[PF], pq[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int t = sub + i * nsub;
    pn[i] = 0.f; pm[i] = 0.f; pq[i] = 0.f;
    if (t < sg.ntiles) {
      const float* p = partial + ((size_t)(sg.tile0 + t) * ngroups + g) * 3;
      pn[i] = p[0]; pm[i] = p[1]; pq[i] = p[2];
    }
  }
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const double nb = pn[i], mb = pm[i], qb = pq[i];
    if (nb > 0.0) {  // (a link past the segment's end carries n = 0, like an empty tile)
      const double nn = N + nb, delta = mb - M;
      M += delta * (nb / nn);
      Q += qb + delta * delta * (N * nb / nn);
      N = nn;
    }
  }
  for (int t = sub + PF * nsub; t < sg.ntiles; t += nsub) {
    const float* p = partial + ((size_t)(sg.tile0 + t) * ngroups + g) * 3;
    const double nb = p[0], mb = p[1], qb = p[2];
    if (nb > 0.0) {
      const double nn = N + nb, delta = mb - M;
      M += delta * (nb / nn);
      Q += qb + delta * delta * (N * nb / nn);
      N = nn;
    }
  }
  sh[threadIdx.x * 3 + 0] = N; sh[threadIdx.x * 3 + 1] = M; sh[threadIdx.x * 3 + 2] = Q;
  __syncthreads();
  if (sub == 0) {
    for (int u = 1; u < nsub; ++u) {
      const double nb = sh[(u * ngroups + g) * 3 + 0], mb = sh[(u * ngroups + g) * 3 + 1], qb = sh[(u * ngroups + g) * 3 + 2];
      if (nb > 0.0) {
        const double nn = N + nb, delta = mb - M;
        M += delta * (nb / nn);
        Q += qb + delta * delta * (N * nb / nn);
        N = nn;
      }
    }
    const double var = N > 0.0 ? Q / N : 0.0;
    stats[(size_t)blockIdx.x * ngroups + g] = make_float2((float)M, (float)(1.0 / sqrt(var + (double)eps)));
  }
  if (coef) {
    __syncthreads();  // the stats of this block's segment are visible to the block
    const int c = threadIdx.x, C = ngroups * 8;
    if (c < C) {
      const float2 st = stats[(size_t)blockIdx.x * ngroups + (c >> 3)];
      const float a = st.y * gamma[c];
      coef[(size_t)blockIdx.x * C + c] = make_float2(a, beta[c] - st.x * a);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_partials_kernel(T* __restrict__ x, const GnSeg* segs, int ld, int ngroups,
                                                                int rows_per_chunk, const float2* __restrict__ stats,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int relu) {
  const int seg = blockIdx.y, chunk = blockIdx.x;
  const GnSeg sg = segs[seg];
  const int r_begin = chunk * rows_per_chunk;
  if (r_begin >= sg.nrows) return;
  const int r_end = min(sg.nrows, r_begin + rows_per_chunk);
  const int g = threadIdx.x % ngroups, rl = threadIdx.x / ngroups, rstep = 256 / ngroups;
  const float2 st = stats[(size_t)seg * ngroups + g];
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = st.y * gamma[g * 8 + j];
    b[j] = beta[g * 8 + j] - st.x * a[j];
  }
  // 4 independent rows per iteration: 4 x 16-byte loads in flight per lane (HBM-bound pass)
  for (int r = r_begin + rl; r < r_end; r += 4 * rstep) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r + u * rstep < r_end) load8<T>(x + (size_t)(sg.row0 + r + u * rstep) * ld + g * 8, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + u * rstep < r_end) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float t = fmaf(v[u][j], a[j], b[j]);
          if (relu) t = t > 0.f ? t : 0.f;
          v[u][j] = t;
        }
        store8<T>(x + (size_t)(sg.row0 + r + u * rstep) * ld + g * 8, v[u]);
      }
    }
  }
}

int launch_gn_apply_partials(DType dt, void* x, int ld, int ngroups, const GnSeg* segs_dev, int nseg, int max_rows,
                             const float* partial, float2* stats_ws, const float* gamma, const float* beta, float eps, int relu,
                             hipStream_t s) {
  if (ngroups != 32 && ngroups != 64) return -1;
  const int rpc = GN_ROWS_PER_CHUNK;
  hipLaunchKernelGGL(gn_finalize_partials_kernel, dim3(nseg), dim3(1024), 0, s, segs_dev, ngroups, partial, eps, stats_ws,
                     (const float*)nullptr, (const float*)nullptr, (float2*)nullptr);
  dim3 grid((max_rows + rpc - 1) / rpc, nseg), block(256);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(gn_apply_partials_kernel<bf16_t>, grid, block, 0, s, (bf16_t*)x, segs_dev, ld, ngroups, rpc, stats_ws, gamma,
                       beta, relu);
  else
    hipLaunchKernelGGL(gn_apply_partials_kernel<float>, grid, block, 0, s, (float*)x, segs_dev, ld, ngroups, rpc, stats_ws, gamma,
                       beta, relu);
  return (int)hipGetLastError();
}

int launch_gn_finalize_coef(int ngroups, const GnSeg* segs_dev, int nseg, const float* partial, float2* stats_ws, const float* gamma,
                            const float* beta, float eps, float2* coef, hipStream_t s) {
  if (ngroups != 32 && ngroups != 64) return -1;
  hipLaunchKernelGGL(gn_finalize_partials_kernel, dim3(nseg), dim3(1024), 0, s, segs_dev, ngroups, partial, eps, stats_ws, gamma, beta,
                     coef);
  return (int)hipGetLastError();
}

int launch_groupnorm(DType dt, void* x, const RowSeg* segs_dev, int nseg, int max_rows, int ld, const float* gamma,
                     const float* beta, float eps, int relu, float* partial, float2* stats, hipStream_t s) {
  const int rpc = GN_ROWS_PER_CHUNK;
  const int max_chunks = (max_rows + rpc - 1) / rpc;
  dim3 grid(max_chunks, nseg), block(256);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, segs_dev, ld, rpc, max_chunks,
                       partial);
  else
    hipLaunchKernelGGL(gn_stats_kernel<float>, grid, block, 0, s, (const float*)x, segs_dev, ld, rpc, max_chunks,
                       partial);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((nseg * 32 + 255) / 256), dim3(256), 0, s, partial, segs_dev, nseg, rpc,
                     max_chunks, eps, stats);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, block, 0, s, (bf16_t*)x, segs_dev, ld, rpc, stats, gamma, beta,
                       relu);
  else
    hipLaunchKernelGGL(gn_apply_kernel<float>, grid, block, 0, s, (float*)x, segs_dev, ld, rpc, stats, gamma, beta,
                       relu);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// dst rows = relu(src rows), 8 channels per lane; grid (chunks, segments)
template <typename T>
__global__ void relu_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, int ld, const CopySeg* segs) {
  const CopySeg sg = segs[blockIdx.y];
  const int per_row = ld / 8;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= sg.nrows * per_row) return;
  const int r = idx / per_row, c = (idx - r * per_row) * 8;
  float v[8];
  load8<T>(src + (size_t)(sg.src_row0 + r) * ld + c, v);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
  store8<T>(dst + (size_t)(sg.dst_row0 + r) * ld + c, v);
}

int launch_relu_rows(DType dt, const void* src, void* dst, int ld, const CopySeg* segs_dev, int nseg, int max_rows,
                     hipStream_t s) {
  dim3 grid((max_rows * (ld / 8) + 255) / 256, nseg), block(256);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(relu_rows_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)src, (bf16_t*)dst, ld, segs_dev);
  else
    hipLaunchKernelGGL(relu_rows_kernel<float>, grid, block, 0, s, (const float*)src, (float*)dst, ld, segs_dev);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// pseudo-random fill in [-1, 1) (kernel micro-benchmarks: random operands, never zeros -- DVFS)
template <typename T>
__global__ void fill_random_kernel(T* __restrict__ p, size_t n, unsigned seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)(i * 2654435761u) ^ seed;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  p[i] = Cvt<T>::from_f((float)(h & 0xffffff) * (2.0f / 16777216.0f) - 1.0f);
}

int launch_fill_random(DType dt, void* p, size_t n, unsigned seed, hipStream_t s) {
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(fill_random_kernel<bf16_t>, grid, block, 0, s, (bf16_t*)p, n, seed);
  else
    hipLaunchKernelGGL(fill_random_kernel<float>, grid, block, 0, s, (float*)p, n, seed);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// layout conversion (API boundary + tests): NCHW fp32 <-> position-major rows
template <typename T>
__global__ void import_nchw_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int HW, int row0,
                                   int ld) {
  // one block = 64 positions x 64 channels through LDS (transpose)
  __shared__ float t[64][65];
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 rows at a time
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, p = p0 + tx;
    t[i][tx] = (c < C && p < HW) ? src[(size_t)c * HW + p] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int p = p0 + i, c = c0 + tx;
    if (p < HW && c < C) dst[(size_t)(row0 + p) * ld + c] = Cvt<T>::from_f(t[tx][i]);
  }
}

template <typename T>
__global__ void export_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int C, int HW, int row0,
                                   int ld) {
  __shared__ float t[64][65];
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int p = p0 + i, c = c0 + tx;
    t[i][tx] = (p < HW && c < C) ? Cvt<T>::to_f(src[(size_t)(row0 + p) * ld + c]) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, p = p0 + tx;
    if (c < C && p < HW) dst[(size_t)c * HW + p] = t[tx][i];
  }
}

int launch_import_nchw(DType dt, const float* src, void* dst, int C, int HW, int row0, int ld, hipStream_t s) {
  dim3 grid((HW + 63) / 64, (C + 63) / 64), block(256);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(import_nchw_kernel<bf16_t>, grid, block, 0, s, src, (bf16_t*)dst, C, HW, row0, ld);
  else
    hipLaunchKernelGGL(import_nchw_kernel<float>, grid, block, 0, s, src, (float*)dst, C, HW, row0, ld);
  return (int)hipGetLastError();
}

int launch_export_nchw(DType dt, const void* src, float* dst, int C, int HW, int row0, int ld, hipStream_t s) {
  dim3 grid((HW + 63) / 64, (C + 63) / 64), block(256);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(export_nchw_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)src, dst, C, HW, row0, ld);
  else
    hipLaunchKernelGGL(export_nchw_kernel<float>, grid, block, 0, s, (const float*)src, dst, C, HW, row0, ld);
  return (int)hipGetLastError();
}

// fp32 rows (head outputs) -> NCHW fp32, channel window [ch0, ch0+C)
__global__ void export_nchw_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW, int row0,
                                       int ld, int ch0) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (p < HW && c < C) dst[(size_t)c * HW + p] = src[(size_t)(row0 + p) * ld + ch0 + c];
}

int launch_export_nchw_f32(const float* src, float* dst, int C, int HW, int row0, int ld, int ch0, hipStream_t s) {
  hipLaunchKernelGGL(export_nchw_f32_kernel, dim3((HW + 255) / 256, C), dim3(256), 0, s, src, dst, C, HW, row0, ld,
                     ch0);
  return (int)hipGetLastError();
}

// class codes (N,C) fp32 -> packed weight rows [Npad][C] in the compute dtype (zero padded); the same launch leaves the class biases
// zero-padded to Npad in bias_pad (nullptr bias: zeros) and, with bias_scan, a copy with -inf from class N on (logits_scan_kernel:
// padded classes never pass the threshold) -- one dispatch instead of a kernel, a memset and one or two device copies per head call.
// SPLIT: every 32-element K-slice as [32 bf16 hi | 32 bf16 lo] (conv_igemm.hip MmaSplit)
template <typename T, bool SPLIT>
__global__ void pack_codes_kernel(const float* __restrict__ w, int N, int C, int Npad, T* __restrict__ out, const float* __restrict__ bias,
                                  float* __restrict__ bias_pad, float* __restrict__ bias_scan) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Npad) {
    const float b = (bias && i < N) ? bias[i] : 0.f;
    if (bias_pad) bias_pad[i] = b;
    if (bias_scan) bias_scan[i] = i < N ? b : __uint_as_float(0xff800000u);
  }
  if (i >= Npad * C) return;
  const float v = i / C < N ? w[i] : 0.f;
  if constexpr (SPLIT) {
    const bf16_t hi = (bf16_t)v;
    const size_t o = (size_t)(i >> 5) * 64 + (i & 31);  // C % 32 == 0: slices do not straddle rows
    out[o] = hi;
    out[o + 32] = (bf16_t)(v - (float)hi);
  } else {
    out[i] = Cvt<T>::from_f(v);
  }
}

int launch_pack_codes(DType dt, const float* w, int N, int C, int Npad, void* out, const float* bias, float* bias_pad, float* bias_scan,
                      hipStream_t s) {
  dim3 grid((Npad * C + 255) / 256), block(256);
  if (dt == DT_F32S) {
    if (C % 32 != 0) return -1;
    hipLaunchKernelGGL((pack_codes_kernel<bf16_t, true>), grid, block, 0, s, w, N, C, Npad, (bf16_t*)out, bias, bias_pad, bias_scan);
  } else if (dt == DT_BF16)
    hipLaunchKernelGGL((pack_codes_kernel<bf16_t, false>), grid, block, 0, s, w, N, C, Npad, (bf16_t*)out, bias, bias_pad, bias_scan);
  else
    hipLaunchKernelGGL((pack_codes_kernel<float, false>), grid, block, 0, s, w, N, C, Npad, (float*)out, bias, bias_pad, bias_scan);
  return (int)hipGetLastError();
}

}  // namespace sylph
