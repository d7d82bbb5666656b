// FCOS decode, pre-NMS top-k, class-aware NMS, post-NMS keep and postprocess, all on device with no
// host round trip (the reference loops levels x images in Python with .item()/.cpu() syncs).
// Compiled with -ffp-contract=off: box and IoU arithmetic must round exactly like the fp32 reference.
//
// Pipeline (one launch each, batch-wide):
//   scan    : sigmoid(logit) > thr -> candidate (score bits, loc*N+cls) appended per (image, level)
//   select  : per (image, level): if count > pre_nms_topk, exact radix-select of the k largest
//             (score, lower-index-first) composites; survivors go to the per-image pool with key
//             (sqrt(score) bits << 32 | ~ordinal), ordinal = (level, location, class) rank
//   sort    : per image bitonic sort (descending) in LDS, then box decode of the sorted candidates
//   mask    : 64x64 blocks of the upper-triangular suppression matrix (same class && IoU > thr)
//   reduce  : one wave per image walks 64-box chunks, keeps/suppresses, stops once post_nms_topk
//             (+ score ties) are kept, applies rescale/clip/non-empty filter, writes the outputs.
//
// Reference arithmetic followed (paths relative to /root/reference):
//   sylph/modeling/meta_fcos/fcos_outputs.py:904-1008 forward_for_single_feature_map
//   sylph/modeling/meta_fcos/fcos_outputs.py:743-812  predict_proposals (reg * stride, level concat)
//   sylph/modeling/meta_fcos/fcos_outputs.py:1010-1028 select_over_all_levels (ml_nms, kthvalue keep)
//   sylph/modeling/meta_arch/meta_one_stage_detector.py:288-296 detector_postprocess
//   sylph/modeling/meta_fcos/fcos.py:270-282 compute_locations
#include "common.h"
#include "kernels.h"

namespace sylph {

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float quality_of(const float* pred_row, int mode) {
  if (mode == 0) return sigmoid_f(pred_row[4]);
  if (mode == 1) return sigmoid_f(pred_row[5]);
  return sqrtf(sigmoid_f(pred_row[5]) * sigmoid_f(pred_row[4]));
}

constexpr int SCAN_ROWS = 64;

__global__ __launch_bounds__(256) void decode_scan_kernel(const DecodeCfg cfg, const DecodeSeg* __restrict__ segs,
                                                          const float* __restrict__ logits,
                                                          const float* __restrict__ pred, int pred_ld,
                                                          const DecodeBuffers buf) {
  const int seg = blockIdx.y;
  const DecodeSeg sg = segs[seg];
  const int r_begin = blockIdx.x * SCAN_ROWS;
  if (r_begin >= sg.nloc) return;
  const int r_end = min(sg.nloc, r_begin + SCAN_ROWS);
  const int N = cfg.num_classes;
  const int total = (r_end - r_begin) * N;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int r = e / N, c = e - r * N;
    const int loc = r_begin + r;
    const size_t row = (size_t)sg.row0 + loc;
    float p = sigmoid_f(logits[row * cfg.logits_ld + c]);
    float s;
    bool pass;
    if (cfg.thresh_with_ctr) {
      p = p * quality_of(pred + row * pred_ld, cfg.quality_mode);
      pass = p > cfg.pre_nms_thresh;
      s = p;
    } else {
      pass = p > cfg.pre_nms_thresh;
      s = pass ? p * quality_of(pred + row * pred_ld, cfg.quality_mode) : 0.f;
    }
    if (pass) {
      const unsigned pos = atomicAdd(&buf.cand_count[seg], 1u);
      if (pos < (unsigned)cfg.cand_cap) {
        buf.cand_key[(size_t)seg * cfg.cand_cap + pos] = __float_as_uint(s);
        buf.cand_idx[(size_t)seg * cfg.cand_cap + pos] = (unsigned)(loc * N + c);
      }
    }
  }
}

__global__ __launch_bounds__(1024) void decode_select_kernel(const DecodeCfg cfg, const DecodeSeg* __restrict__ segs,
                                                             const DecodeBuffers buf) {
  const int seg = blockIdx.x, tid = threadIdx.x;
  const DecodeSeg sg = segs[seg];
  unsigned n = buf.cand_count[seg];
  if (n > (unsigned)cfg.cand_cap) {
    if (tid == 0) atomicOr(buf.status, 1);
    n = cfg.cand_cap;
  }
  const unsigned* key = buf.cand_key + (size_t)seg * cfg.cand_cap;
  const unsigned* idx = buf.cand_idx + (size_t)seg * cfg.cand_cap;
  const unsigned k = (unsigned)cfg.pre_nms_topk;
  __shared__ unsigned hist[256];
  __shared__ unsigned long long sh_prefix;
  __shared__ unsigned sh_remain;
  unsigned long long thresh = 0ull;
  if (n > k) {
    unsigned long long prefix = 0ull;
    unsigned remain = k;
    for (int shift = 56; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      const unsigned long long hi_mask = shift == 56 ? 0ull : (~0ull << (shift + 8));
      for (unsigned i = tid; i < n; i += 1024) {
        const unsigned long long comp = ((unsigned long long)key[i] << 32) | (unsigned long long)(~idx[i]);
        if ((comp & hi_mask) == prefix) atomicAdd(&hist[(unsigned)(comp >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned cum = 0;
        int d = 255;
        for (; d > 0; --d) {
          if (cum + hist[d] >= remain) break;
          cum += hist[d];
        }
        sh_prefix = prefix | ((unsigned long long)d << shift);
        sh_remain = remain - cum;
      }
      __syncthreads();
      prefix = sh_prefix;
      remain = sh_remain;
    }
    thresh = prefix;
  }
  const int img = sg.image;
  for (unsigned i = tid; i < n; i += 1024) {
    const unsigned long long comp = ((unsigned long long)key[i] << 32) | (unsigned long long)(~idx[i]);
    if (comp >= thresh) {
      const float sq = sqrtf(__uint_as_float(key[i]));
      const unsigned ord = sg.loc_base * (unsigned)cfg.num_classes + idx[i];
      const unsigned pos = atomicAdd(&buf.pool_count[img], 1u);
      if (pos < (unsigned)cfg.pool_cap)
        buf.pool_key[(size_t)img * cfg.pool_cap + pos] =
            ((unsigned long long)__float_as_uint(sq) << 32) | (unsigned long long)(~ord);
    }
  }
}

// per image: bitonic sort (descending) of the pool keys in LDS, then decode the sorted candidates
__global__ __launch_bounds__(1024) void decode_sort_kernel(const DecodeCfg cfg, const DecodeSeg* __restrict__ segs,
                                                           const float* __restrict__ pred, int pred_ld,
                                                           const DecodeBuffers buf) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const int img = blockIdx.x, tid = threadIdx.x;
  unsigned n = buf.pool_count[img];
  if (n > (unsigned)cfg.pool_cap) n = cfg.pool_cap;
  unsigned P = 64;
  while (P < n) P <<= 1;
  const unsigned long long* src = buf.pool_key + (size_t)img * cfg.pool_cap;
  for (unsigned i = tid; i < P; i += 1024) keys[i] = i < n ? src[i] : 0ull;
  __syncthreads();
  for (unsigned size = 2; size <= P; size <<= 1) {
    for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
      for (unsigned t = tid; t < (P >> 1); t += 1024) {
        const unsigned lo = ((t / stride) * (stride << 1)) + (t % stride);
        const unsigned hi = lo + stride;
        const bool desc = ((lo & size) == 0);  // descending overall
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  const int N = cfg.num_classes, L = cfg.nlevels;
  const DecodeSeg* isegs = segs + (size_t)img * L;
  for (unsigned i = tid; i < n; i += 1024) {
    const unsigned long long kk = keys[i];
    const unsigned ord = ~(unsigned)(kk & 0xffffffffull);
    const float score = __uint_as_float((unsigned)(kk >> 32));
    int l = L - 1;
    while (l > 0 && ord < isegs[l].loc_base * (unsigned)N) --l;
    const DecodeSeg sg = isegs[l];
    const unsigned idx = ord - sg.loc_base * (unsigned)N;
    const int loc = (int)(idx / (unsigned)N), cls = (int)(idx - (unsigned)loc * N);
    const int ly = loc / sg.W, lx = loc - ly * sg.W;
    const float x = (float)(lx * sg.stride + sg.stride / 2), y = (float)(ly * sg.stride + sg.stride / 2);
    const float* pr = pred + ((size_t)sg.row0 + loc) * pred_ld;
    const float st = (float)sg.stride;
    const float r0 = pr[0] * st, r1 = pr[1] * st, r2 = pr[2] * st, r3 = pr[3] * st;
    const size_t o = (size_t)img * cfg.pool_cap + i;
    buf.s_box[o * 4 + 0] = x - r0;
    buf.s_box[o * 4 + 1] = y - r1;
    buf.s_box[o * 4 + 2] = x + r2;
    buf.s_box[o * 4 + 3] = y + r3;
    buf.s_score[o] = score;
    buf.s_cls[o] = cls;
    buf.s_level[o] = sg.level;
    buf.s_loc[o * 2 + 0] = x;
    buf.s_loc[o * 2 + 1] = y;
    buf.s_ord[o] = ord;
  }
}

// suppression bits: block (cj, ci, img), 64 threads; thread t = box i = ci*64+t vs boxes of chunk cj
__global__ __launch_bounds__(64) void nms_mask_kernel(const DecodeCfg cfg, const DecodeBuffers buf) {
  // grid.x enumerates the upper triangle (ci <= cj) only: t = cj (cj + 1) / 2 + ci  (a square grid spent half of its
  // 400 000 one-wave blocks on an immediate return: the launch was dispatch bound)
  const int tri = blockIdx.x, img = blockIdx.z;
  int cj = (int)((sqrtf(8.f * (float)tri + 1.f) - 1.f) * 0.5f);
  while ((cj + 1) * (cj + 2) / 2 <= tri) ++cj;
  while (cj * (cj + 1) / 2 > tri) --cj;
  const int ci = tri - cj * (cj + 1) / 2;
  unsigned n = buf.pool_count[img];
  if (n > (unsigned)cfg.pool_cap) n = cfg.pool_cap;
  if ((unsigned)ci * 64u >= n || (unsigned)cj * 64u >= n) return;
  __shared__ float jb[64][4];
  __shared__ int jc[64];
  const int t = threadIdx.x;
  const size_t base = (size_t)img * cfg.pool_cap;
  const unsigned j = cj * 64 + t;
  if (j < n) {
    jb[t][0] = buf.s_box[(base + j) * 4 + 0]; jb[t][1] = buf.s_box[(base + j) * 4 + 1];
    jb[t][2] = buf.s_box[(base + j) * 4 + 2]; jb[t][3] = buf.s_box[(base + j) * 4 + 3];
    jc[t] = buf.s_cls[base + j];
  } else {
    jc[t] = -1;
  }
  __syncthreads();
  const unsigned i = ci * 64 + t;
  if (i >= n) return;
  const float x1 = buf.s_box[(base + i) * 4 + 0], y1 = buf.s_box[(base + i) * 4 + 1];
  const float x2 = buf.s_box[(base + i) * 4 + 2], y2 = buf.s_box[(base + i) * 4 + 3];
  const int ic = buf.s_cls[base + i];
  const float iarea = (x2 - x1) * (y2 - y1);
  unsigned long long bits = 0ull;
  const int jstart = (ci == cj) ? t + 1 : 0;
  for (int jj = jstart; jj < 64; ++jj) {
    if (jc[jj] != ic) continue;
    const float xx1 = fmaxf(x1, jb[jj][0]), yy1 = fmaxf(y1, jb[jj][1]);
    const float xx2 = fminf(x2, jb[jj][2]), yy2 = fminf(y2, jb[jj][3]);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    const float jarea = (jb[jj][2] - jb[jj][0]) * (jb[jj][3] - jb[jj][1]);
    const float ovr = inter / (iarea + jarea - inter);
    if (ovr > cfg.nms_thresh) bits |= 1ull << jj;
  }
  buf.mask[(base + i) * (size_t)(cfg.pool_cap / 64) + cj] = bits;
}

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
  const unsigned lo = __shfl((unsigned)(v & 0xffffffffull), src);
  const unsigned hi = __shfl((unsigned)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void nms_reduce_kernel(const DecodeCfg cfg, const DecodeBuffers buf,
                                                        const ImageOut* __restrict__ img_out, float* out_boxes,
                                                        float* out_scores, int* out_classes, int* out_levels,
                                                        float* out_locations, int* out_cand, int* out_counts) {
  const int img = blockIdx.x, lane = threadIdx.x;
  unsigned n = buf.pool_count[img];
  if (n > (unsigned)cfg.pool_cap) n = cfg.pool_cap;
  const int nw = (int)((n + 63) / 64);
  const size_t Wd = cfg.pool_cap / 64;
  const size_t base = (size_t)img * cfg.pool_cap;
  const ImageOut io = img_out[img];
  const int K = cfg.post_nms_topk;
  unsigned long long rem0 = 0ull, rem1 = 0ull;
  int kept_total = 0, nout = 0;
  float kth = -1.f;
  bool truncated = false;
  for (int c = 0; c < nw; ++c) {
    const unsigned b0 = c * 64;
    const unsigned bi = b0 + lane;
    const bool valid = bi < n;
    const unsigned long long d = (valid && cfg.nms_thresh > 0.f) ? buf.mask[(base + bi) * Wd + c] : 0ull;
    unsigned long long cur = c < 64 ? shfl_u64(rem0, c) : shfl_u64(rem1, c - 64);
    if (n - b0 < 64u) cur |= ~0ull << (n - b0);
    unsigned long long keptmask = 0ull;
    for (int b = 0; b < 64; ++b) {
      const unsigned long long db = shfl_u64(d, b);
      if (!((cur >> b) & 1ull)) {
        keptmask |= 1ull << b;
        cur |= db;
      }
    }
    if (cfg.nms_thresh > 0.f) {
      unsigned long long km = keptmask;
      while (km) {
        const int b = __ffsll((long long)km) - 1;
        km &= km - 1;
        const unsigned long long* mr = buf.mask + (base + b0 + b) * Wd;
        if (lane < nw && lane > c) rem0 |= mr[lane];
        if (lane + 64 < nw && lane + 64 > c) rem1 |= mr[lane + 64];
      }
    }
    const int nk = __popcll(keptmask);
    const bool is_kept = (keptmask >> lane) & 1ull;
    const int rank = kept_total + __popcll(keptmask & ((1ull << lane) - 1ull));
    const float score = valid ? buf.s_score[base + bi] : 0.f;
    if (K > 0 && kth < 0.f && kept_total + nk >= K) {
      const unsigned long long sel = __ballot(is_kept && rank == K - 1);
      const int src = __ffsll((long long)sel) - 1;
      kth = __shfl(score, src);
    }
    bool emit = is_kept && (K <= 0 || rank < K || score >= kth);
    float bx1 = 0.f, by1 = 0.f, bx2 = 0.f, by2 = 0.f;
    if (emit) {
      bx1 = buf.s_box[(base + bi) * 4 + 0] * io.sx; by1 = buf.s_box[(base + bi) * 4 + 1] * io.sy;
      bx2 = buf.s_box[(base + bi) * 4 + 2] * io.sx; by2 = buf.s_box[(base + bi) * 4 + 3] * io.sy;
      bx1 = fminf(fmaxf(bx1, 0.f), io.out_w); by1 = fminf(fmaxf(by1, 0.f), io.out_h);
      bx2 = fminf(fmaxf(bx2, 0.f), io.out_w); by2 = fminf(fmaxf(by2, 0.f), io.out_h);
      emit = (bx2 - bx1) > 0.f && (by2 - by1) > 0.f;
    }
    const unsigned long long em = __ballot(emit);
    const int slot = nout + __popcll(em & ((1ull << lane) - 1ull));
    if (emit) {
      if (slot < cfg.max_out) {
        const size_t o = (size_t)img * cfg.max_out + slot;
        out_boxes[o * 4 + 0] = bx1; out_boxes[o * 4 + 1] = by1; out_boxes[o * 4 + 2] = bx2; out_boxes[o * 4 + 3] = by2;
        out_scores[o] = score;
        out_classes[o] = buf.s_cls[base + bi];
        out_levels[o] = buf.s_level[base + bi];
        out_locations[o * 2 + 0] = buf.s_loc[(base + bi) * 2 + 0];
        out_locations[o * 2 + 1] = buf.s_loc[(base + bi) * 2 + 1];
        out_cand[o] = (int)buf.s_ord[base + bi];
      } else {
        truncated = true;
      }
    }
    nout += __popcll(em);
    kept_total += nk;
    if (K > 0 && kept_total >= K && c + 1 < nw) {
      const float next_score = buf.s_score[base + b0 + 64];
      if (next_score < kth) break;
    }
  }
  if (__any(truncated) && lane == 0) atomicOr(buf.status, 2);
  if (lane == 0) out_counts[img] = nout < cfg.max_out ? nout : cfg.max_out;
}

int launch_decode(const DecodeCfg& cfg, const DecodeSeg* segs_dev, int nseg, int max_nloc, int B, int nw_bound,
                  const float* logits, const float* pred, int pred_ld, const DecodeBuffers& buf,
                  const ImageOut* img_out_dev, float* out_boxes, float* out_scores, int* out_classes,
                  int* out_levels, float* out_locations, int* out_cand, int* out_counts, hipStream_t s) {
  (void)hipMemsetAsync(buf.cand_count, 0, sizeof(unsigned) * nseg, s);
  (void)hipMemsetAsync(buf.pool_count, 0, sizeof(unsigned) * B, s);
  (void)hipMemsetAsync(buf.status, 0, sizeof(int), s);
  dim3 g1((max_nloc + SCAN_ROWS - 1) / SCAN_ROWS, nseg);
  hipLaunchKernelGGL(decode_scan_kernel, g1, dim3(256), 0, s, cfg, segs_dev, logits, pred, pred_ld, buf);
  hipLaunchKernelGGL(decode_select_kernel, dim3(nseg), dim3(1024), 0, s, cfg, segs_dev, buf);
  hipLaunchKernelGGL(decode_sort_kernel, dim3(B), dim3(1024), sizeof(unsigned long long) * cfg.pool_cap, s, cfg,
                     segs_dev, pred, pred_ld, buf);
  if (cfg.nms_thresh > 0.f) {
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nw_bound * (nw_bound + 1) / 2, 1, B), dim3(64), 0, s, cfg, buf);
  }
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(64), 0, s, cfg, buf, img_out_dev, out_boxes, out_scores,
                     out_classes, out_levels, out_locations, out_cand, out_counts);
  return (int)hipGetLastError();
}

}  // namespace sylph
