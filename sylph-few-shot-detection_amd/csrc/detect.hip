// FCOS decode, pre-NMS top-k, class-aware NMS, post-NMS keep and postprocess, all on device with no
// host round trip (the reference loops levels x images in Python with .item()/.cpu() syncs).
// Compiled with -ffp-contract=off: box and IoU arithmetic must round exactly like the fp32 reference.
//
// Pipeline (one launch each, batch-wide):
//   scan    : sigmoid(logit) > thr -> candidate (score bits, loc*N+cls) appended per (image, level); with more than 32
//             classes (bf16) the class-conditional conv and this scan are ONE kernel and the logits are never stored
//             (logits_scan_kernel, below)
//   select  : per (image, level): if count > pre_nms_topk, the k largest (score, lower-index-first) composites exactly:
//             score histogram -> partition around the bin of the k-th largest -> radix select inside that bin
//             (decode_hist / decode_partition / decode_finish); survivors go to the per-image pool with key
//             (sqrt(score) bits << 32 | ~ordinal), ordinal = (level, location, class) rank
//   sort    : per image bitonic sort (descending) in LDS, then box decode of the sorted candidates
//   nms     : one block per image walks 64-box chunks of the sorted pool: suppression by the boxes kept so far (same class &&
//             IoU > thr) computed for the visited chunks only, stops once post_nms_topk (+ score ties) are kept, applies
//             rescale/clip/non-empty filter, writes the outputs.
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

// box quality of a location (fcos_outputs.py:938-959): centerness, IoU, or sqrt(IoU * centerness)
__device__ __forceinline__ float quality_from(float ctr_logit, float iou_logit, int mode) {
  if (mode == 0) return sigmoid_f(ctr_logit);
  if (mode == 1) return sigmoid_f(iou_logit);
  return sqrtf(sigmoid_f(iou_logit) * sigmoid_f(ctr_logit));
}

constexpr int SCAN_ROWS = 64;     // minimum locations per block
constexpr int SCAN_UNROLL = 4;    // loads in flight per thread (16 B each on the vector path)
constexpr int SCAN_STAGE = 2048;  // candidates a block collects in LDS before it reserves a range of the (image, level) buffer

// Candidates are appended per (image, level).  One global atomicAdd per CANDIDATE on that one counter serialises: at 866 classes
// (LVIS) and a few per cent of the 14.5 M scores of a P3 level above the threshold the scan took 32 ms for 16 images.  Here a
// block collects its candidates in LDS (wave-aggregated LDS atomics: one per wave and iteration) and reserves global ranges with
// one atomicAdd per flush -- a handful per block.  The order of the buffer is irrelevant: selection and sort work on the
// composite (score, ~ordinal) keys.  The logits are the only HBM traffic that matters (4 bytes per score); a thread keeps
// SCAN_UNROLL 16-byte loads (4 classes of a location each) in flight before it looks at any of them.
// The LDS list is sized for occupancy (16 KiB: 8 blocks per CU), not for the worst case: it is flushed once half full, and
// the part of a wave's candidates that still does not fit (density above ~50 % of a round) is appended directly with one
// global atomicAdd per wave.
__global__ __launch_bounds__(256) void decode_scan_kernel(const DecodeCfg cfg, const DecodeSeg* __restrict__ segs,
                                                          const float* __restrict__ logits,
                                                          const float* __restrict__ pred, int pred_ld,
                                                          const DecodeBuffers buf, int rows_per_block) {
  const int seg = blockIdx.y;
  const DecodeSeg sg = segs[seg];
  const int r_begin = blockIdx.x * rows_per_block;
  if (r_begin >= sg.nloc) return;
  const int r_end = min(sg.nloc, r_begin + rows_per_block);
  constexpr int W = 4;  // classes per load
  const int N = cfg.num_classes;
  const int G = (N + W - 1) / W;  // W-wide class groups per location
  const int total = (r_end - r_begin) * G;
  __shared__ unsigned s_key[SCAN_STAGE], s_idx[SCAN_STAGE];
  __shared__ unsigned s_n, s_base;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) s_n = 0u;
  __syncthreads();
  auto flush = [&]() {  // block-uniform
    __syncthreads();
    const unsigned n = min(s_n, (unsigned)SCAN_STAGE);
    if (n > 0u) {
      if (tid == 0) s_base = atomicAdd(&buf.cand_count[seg], n);
      __syncthreads();
      const unsigned base = s_base;
      for (unsigned i = tid; i < n; i += 256) {
        const unsigned pos = base + i;
        if (pos < (unsigned)cfg.cand_cap) {
          buf.cand_key[(size_t)seg * cfg.cand_cap + pos] = s_key[i];
          buf.cand_idx[(size_t)seg * cfg.cand_cap + pos] = s_idx[i];
        }
      }
      __syncthreads();
      if (tid == 0) s_n = 0u;
      __syncthreads();
    }
  };
  // (location, group) of the thread's element without a division per element: it advances by (256 / G, 256 % G) per step
  const int step_r = 256 / G, step_g = 256 - step_r * G;
  int r = tid / G, g = tid - r * G;
  const int iters = (total + 255) / 256;
  // sigmoid(x) > thr needs x > logit(thr) (and so does sigmoid(x) * quality > thr, quality <= 1): scores whose logit is below
  // that bound by more than 1e-2 -- five orders of magnitude above the rounding error of either form -- skip the exp and the
  // division, which is ~95 % of a many-way level.  The exact fp32 test on the sigmoid still decides everything near the bound.
  const float thr = cfg.pre_nms_thresh;
  const float x_min = (thr > 0.f && thr < 1.f) ? logf(thr / (1.f - thr)) - 1e-2f : (thr >= 1.f ? INFINITY : -INFINITY);
  for (int it0 = 0; it0 < iters; it0 += SCAN_UNROLL) {
    if (it0 > 0) {
      __syncthreads();
      const unsigned n_now = s_n;
      __syncthreads();  // every thread has read the same value before any wave appends again: block-uniform decision
      if (n_now > (unsigned)(SCAN_STAGE / 2)) flush();
    }
    float x[SCAN_UNROLL][W], qa[SCAN_UNROLL], qb[SCAN_UNROLL];
    int loc_u[SCAN_UNROLL], c_u[SCAN_UNROLL];
#pragma unroll
    for (int u = 0; u < SCAN_UNROLL; ++u) {
      const int e = (it0 + u) * 256 + tid;
      loc_u[u] = r_begin + r;
      c_u[u] = e < total ? g * W : N;  // N: nothing of this slot is a class
      if (e < total) {
        const float* src = logits + ((size_t)sg.row0 + loc_u[u]) * cfg.logits_ld + g * W;
        // the location's centerness / IoU logits ride along (a cache hit for all but the first group of a location): a
        // dependent load after the threshold test would stall every round on HBM latency
        const float* prow = pred + ((size_t)sg.row0 + loc_u[u]) * pred_ld;
        qa[u] = cfg.quality_mode != 1 ? prow[4] : 0.f;
        qb[u] = cfg.quality_mode != 0 ? prow[5] : 0.f;
        const float4 v = *reinterpret_cast<const float4*>(src);
        x[u][0] = v.x; x[u][1] = v.y; x[u][2] = v.z; x[u][3] = v.w;
      }
      r += step_r; g += step_g;
      if (g >= G) { g -= G; ++r; }
    }
#pragma unroll
    for (int u = 0; u < SCAN_UNROLL; ++u) {
      bool pass[W];
      float sc[W];
      bool any = false;
#pragma unroll
      for (int w = 0; w < W; ++w) {
        pass[w] = false; sc[w] = 0.f;
        if (c_u[u] + w < N) {
          const float p = x[u][w] > x_min ? sigmoid_f(x[u][w]) : 0.f;
          if (p > thr) { sc[w] = p; pass[w] = true; any = true; }
        }
      }
      if (any) {  // the quality (centerness / IoU) of the location: p * quality <= p, so it is only needed above the threshold
        const float q = quality_from(qa[u], qb[u], cfg.quality_mode);
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const float pq = sc[w] * q;
          if (cfg.thresh_with_ctr) pass[w] = pass[w] && pq > thr;
          sc[w] = pq;
        }
      }
      unsigned long long m[W];
      unsigned cnt = 0u;
#pragma unroll
      for (int w = 0; w < W; ++w) { m[w] = __ballot(pass[w]); cnt += (unsigned)__popcll(m[w]); }
      if (cnt != 0u) {  // wave-uniform
        unsigned wbase = 0u;
        if (lane == 0) wbase = atomicAdd(&s_n, cnt);
        wbase = __shfl(wbase, 0);
        // positions [wbase, wbase + cnt): those below SCAN_STAGE are LDS slots, the rest one reserved global range
        const unsigned first_over = max(wbase, (unsigned)SCAN_STAGE);
        unsigned gbase = 0u;
        if (wbase + cnt > first_over) {  // wave-uniform
          if (lane == 0) gbase = atomicAdd(&buf.cand_count[seg], wbase + cnt - first_over);
          gbase = __shfl(gbase, 0);
        }
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int w = 0; w < W; ++w) {
          if (pass[w]) {
            const unsigned pos = wbase + (unsigned)__popcll(m[w] & below);
            const unsigned kb = __float_as_uint(sc[w]), ix = (unsigned)(loc_u[u] * N + c_u[u] + w);
            if (pos < (unsigned)SCAN_STAGE) {
              s_key[pos] = kb;
              s_idx[pos] = ix;
            } else {
              const unsigned gpos = gbase + (pos - first_over);
              if (gpos < (unsigned)cfg.cand_cap) {
                buf.cand_key[(size_t)seg * cfg.cand_cap + gpos] = kb;
                buf.cand_idx[(size_t)seg * cfg.cand_cap + gpos] = ix;
              }
            }
          }
          wbase += (unsigned)__popcll(m[w]);
        }
      }
    }
  }
  flush();
}

constexpr int SEL_UNROLL = 8;   // key loads in flight per thread

__device__ __forceinline__ unsigned sel_bin_of(unsigned kbits) { return min(kbits >> 19, (unsigned)(SEL_BINS - 1)); }

__device__ __forceinline__ unsigned long long pool_composite(const DecodeCfg& cfg, const DecodeSeg& sg, unsigned kbits, unsigned ix) {
  const float sq = sqrtf(__uint_as_float(kbits));
  const unsigned ord = sg.loc_base * (unsigned)cfg.num_classes + ix;
  return ((unsigned long long)__float_as_uint(sq) << 32) | (unsigned long long)(~ord);
}

// Block-level append to a global list with ONE atomicAdd on its counter per block: returning atomics on one address take
// ~80 ns each at L2, and the five levels of an image share pool_count[image] (5000 appends = 0.4 ms when done one by one).
// Entries collect in LDS; the few that do not fit (cap < entries of this block) are appended directly.
template <int CAP>
struct StagedAppend {
  unsigned long long* buf;  // LDS [CAP]
  unsigned* cnt;            // LDS
  unsigned* base;           // LDS
  unsigned* g_count;
  unsigned long long* g_dst;
  unsigned g_cap;
  __device__ __forceinline__ void init() { if (threadIdx.x == 0) *cnt = 0u; }  // followed by a __syncthreads() of the caller
  __device__ __forceinline__ void push(unsigned long long v) {
    const unsigned pos = atomicAdd(cnt, 1u);
    if (pos < (unsigned)CAP) { buf[pos] = v; return; }
    const unsigned gpos = atomicAdd(g_count, 1u);
    if (gpos < g_cap) g_dst[gpos] = v;
  }
  __device__ __forceinline__ void flush() {  // block-uniform
    __syncthreads();
    const unsigned n = min(*cnt, (unsigned)CAP);
    if (n == 0u) return;
    if (threadIdx.x == 0) *base = atomicAdd(g_count, n);
    __syncthreads();
    const unsigned b = *base;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
      if (b + i < g_cap) g_dst[b + i] = buf[i];
  }
};

// exact radix select (8 bits per pass, most significant first) of the `remain`-th largest of the composites load(i), i < n;
// every thread of the 1024-thread block calls it and gets the threshold composite
template <typename Load>
__device__ __forceinline__ unsigned long long radix_select_desc(unsigned n, unsigned remain, unsigned* hist,
                                                                unsigned long long* sh_prefix, unsigned* sh_remain,
                                                                Load load) {
  const int tid = threadIdx.x;
  unsigned long long prefix = 0ull;
  for (int shift = 56; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const unsigned long long hi_mask = shift == 56 ? 0ull : (~0ull << (shift + 8));
    for (unsigned i = tid; i < n; i += 1024) {
      const unsigned long long comp = load(i);
      if ((comp & hi_mask) == prefix) atomicAdd(&hist[(unsigned)(comp >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      // digit d = the largest with count(digits > d) < remain <= count(digits >= d); lane l owns digits 4l .. 4l + 3
      const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
      const unsigned mine = h0 + h1 + h2 + h3;
      unsigned above = 0u;  // count of the digits owned by higher lanes
      for (int l = 0; l < 64; ++l) {
        const unsigned v = __shfl(mine, l);
        if (l > tid) above += v;
      }
      const bool here = above < remain && above + mine >= remain;
      const unsigned long long found = __ballot(here);
      if (here) {
        unsigned cum = above;
        int d = 4 * tid + 3;
        if (cum + h3 < remain) { cum += h3; --d;
          if (cum + h2 < remain) { cum += h2; --d;
            if (cum + h1 < remain) { cum += h1; --d; } } }
        *sh_prefix = prefix | ((unsigned long long)d << shift);
        *sh_remain = remain - cum;
      } else if (found == 0ull && tid == 0) {  // fewer than `remain` composites under this prefix (not reached by the callers)
        *sh_prefix = prefix;
        *sh_remain = remain - above - (mine - h0);
      }
    }
    __syncthreads();
    prefix = *sh_prefix;
    remain = *sh_remain;
    __syncthreads();
  }
  return prefix;
}

// Pre-NMS top-k per (image, level): keep exactly the k largest (score, lower index first) of more than k candidates.
// A many-way level holds ~10^6 candidates (866 classes x 16 800 locations x a few per cent), so the k-th largest is located
// in two reads of the scores instead of eight of scores and indices, by `parts` blocks per level:
//   hist      : 4096-bin histogram of the score's exponent and top mantissa bits (block-local in LDS, merged with atomics)
//   partition : the bin b* the k-th largest falls in is read off the histogram; candidates of higher bins go to the pool
//               outright, the boundary bin's (a few hundred) to a per-level list
//   finish    : exact radix select of the list; a boundary bin too full for the list (scores piled on one value) is the one
//               case nothing was partitioned and the radix select runs over the whole buffer.
// Levels with at most k candidates skip hist and finish; partition copies them to the pool.
__device__ __forceinline__ unsigned sel_count(const DecodeCfg& cfg, const DecodeBuffers& buf, int seg) {
  const unsigned n = buf.cand_count[seg];
  return n > (unsigned)cfg.cand_cap ? (unsigned)cfg.cand_cap : n;
}

__global__ __launch_bounds__(1024) void decode_hist_kernel(const DecodeCfg cfg, const DecodeBuffers buf) {
  const int seg = blockIdx.y, tid = threadIdx.x;
  const unsigned n = sel_count(cfg, buf, seg);
  if (n <= (unsigned)cfg.pre_nms_topk) return;
  const unsigned slice = (n + gridDim.x - 1) / gridDim.x;
  const unsigned i_begin = blockIdx.x * slice, i_end = min(n, i_begin + slice);
  const unsigned* key = buf.cand_key + (size_t)seg * cfg.cand_cap;
  __shared__ unsigned hist[SEL_BINS];
  for (int i = tid; i < SEL_BINS; i += 1024) hist[i] = 0u;
  __syncthreads();
  for (unsigned i0 = i_begin + tid; i0 < i_end; i0 += 1024 * SEL_UNROLL) {
    unsigned kb[SEL_UNROLL];
#pragma unroll
    for (int u = 0; u < SEL_UNROLL; ++u) kb[u] = i0 + u * 1024 < i_end ? key[i0 + u * 1024] : 0u;
#pragma unroll
    for (int u = 0; u < SEL_UNROLL; ++u)
      if (i0 + u * 1024 < i_end) atomicAdd(&hist[sel_bin_of(kb[u])], 1u);
  }
  __syncthreads();
  unsigned* ghist = buf.sel_ws + (size_t)seg * SEL_WS;
  for (int i = tid; i < SEL_BINS; i += 1024)
    if (hist[i] != 0u) atomicAdd(&ghist[i], hist[i]);
}

__global__ __launch_bounds__(1024) void decode_partition_kernel(const DecodeCfg cfg, const DecodeSeg* __restrict__ segs,
                                                                const DecodeBuffers buf) {
  const int seg = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const DecodeSeg sg = segs[seg];
  if (blockIdx.x == 0 && tid == 0 && buf.cand_count[seg] > (unsigned)cfg.cand_cap) atomicOr(buf.status, 1);
  const unsigned n = sel_count(cfg, buf, seg);
  const unsigned k = (unsigned)cfg.pre_nms_topk;
  const unsigned slice = (n + gridDim.x - 1) / gridDim.x;
  const unsigned i_begin = blockIdx.x * slice, i_end = min(n, i_begin + slice);
  const unsigned* key = buf.cand_key + (size_t)seg * cfg.cand_cap;
  const unsigned* idx = buf.cand_idx + (size_t)seg * cfg.cand_cap;
  unsigned* ghist = buf.sel_ws + (size_t)seg * SEL_WS;
  unsigned* state = ghist + SEL_BINS;
  __shared__ unsigned hist[SEL_BINS];
  __shared__ unsigned sh_bin, sh_need;
  __shared__ unsigned long long pool_stage[2048], tie_stage[1024];
  __shared__ unsigned pool_n, pool_base, tie_n, tie_base;
  StagedAppend<2048> pool{pool_stage, &pool_n, &pool_base, &buf.pool_count[sg.image],
                          buf.pool_key + (size_t)sg.image * cfg.pool_cap, (unsigned)cfg.pool_cap};
  StagedAppend<1024> ties{tie_stage, &tie_n, &tie_base, &state[0], buf.sel_tie + (size_t)seg * SEL_TIE, (unsigned)SEL_TIE};
  pool.init();
  ties.init();
  if (n <= k) {
    __syncthreads();
    for (unsigned i = i_begin + tid; i < i_end; i += 1024) pool.push(pool_composite(cfg, sg, key[i], idx[i]));
    pool.flush();
    return;
  }
  for (int i = tid; i < SEL_BINS; i += 1024) hist[i] = ghist[i];
  __syncthreads();
  if (tid < 64) {
    // lane c owns bins [64c, 64c+64); the rotated order keeps the 64 lanes on 64 different LDS banks
    unsigned sum = 0u;
    for (int j = 0; j < 64; ++j) sum += hist[lane * 64 + ((j + lane) & 63)];
    unsigned above = 0u;  // candidates in the chunks above this lane's
    for (int c = 0; c < 64; ++c) {
      const unsigned v = __shfl(sum, c);
      if (c > lane) above += v;
    }
    if (above < k && above + sum >= k) {  // exactly one lane: the chunk that holds the k-th largest
      int b = lane * 64 + 63;
      for (;; --b) {
        if (above + hist[b] >= k) break;
        above += hist[b];
      }
      sh_bin = (unsigned)b;
      sh_need = k - above;  // how many of bin b's candidates are kept
    }
  }
  __syncthreads();
  const unsigned b_star = sh_bin, n_b = hist[b_star];
  if (blockIdx.x == 0 && tid == 0) { state[1] = b_star; state[2] = sh_need; state[3] = n_b; }
  if (n_b > (unsigned)SEL_TIE) return;  // finish selects over the whole buffer
  for (unsigned i0 = i_begin + tid; i0 < i_end; i0 += 1024 * SEL_UNROLL) {
    unsigned kb[SEL_UNROLL];
#pragma unroll
    for (int u = 0; u < SEL_UNROLL; ++u) kb[u] = i0 + u * 1024 < i_end ? key[i0 + u * 1024] : 0u;
#pragma unroll
    for (int u = 0; u < SEL_UNROLL; ++u) {
      const unsigned i = i0 + u * 1024;
      const unsigned b = sel_bin_of(kb[u]);
      if (i >= i_end || b < b_star) continue;  // the indices of the (vast) rest are never read
      const unsigned ix = idx[i];
      if (b > b_star) pool.push(pool_composite(cfg, sg, kb[u], ix));
      else ties.push(((unsigned long long)kb[u] << 32) | (unsigned long long)(~ix));
    }
  }
  pool.flush();
  ties.flush();
}

__global__ __launch_bounds__(1024) void decode_finish_kernel(const DecodeCfg cfg, const DecodeSeg* __restrict__ segs,
                                                             const DecodeBuffers buf) {
  const int seg = blockIdx.x, tid = threadIdx.x;
  const unsigned n = sel_count(cfg, buf, seg);
  const unsigned k = (unsigned)cfg.pre_nms_topk;
  if (n <= k) return;
  const DecodeSeg sg = segs[seg];
  const unsigned* state = buf.sel_ws + (size_t)seg * SEL_WS + SEL_BINS;
  const unsigned need = state[2], n_b = state[3];
  __shared__ unsigned hist[256];
  __shared__ unsigned long long sh_prefix;
  __shared__ unsigned sh_remain;
  __shared__ unsigned long long pool_stage[2048];
  __shared__ unsigned pool_n, pool_base;
  StagedAppend<2048> pool{pool_stage, &pool_n, &pool_base, &buf.pool_count[sg.image],
                          buf.pool_key + (size_t)sg.image * cfg.pool_cap, (unsigned)cfg.pool_cap};
  pool.init();
  __syncthreads();
  if (n_b <= (unsigned)SEL_TIE) {
    const unsigned long long* tie = buf.sel_tie + (size_t)seg * SEL_TIE;
    const unsigned long long thresh =
        need >= n_b ? 0ull : radix_select_desc(n_b, need, hist, &sh_prefix, &sh_remain, [&](unsigned i) { return tie[i]; });
    for (unsigned i = tid; i < n_b; i += 1024) {
      const unsigned long long comp = tie[i];
      if (comp >= thresh) pool.push(pool_composite(cfg, sg, (unsigned)(comp >> 32), ~(unsigned)(comp & 0xffffffffull)));
    }
  } else {
    const unsigned* key = buf.cand_key + (size_t)seg * cfg.cand_cap;
    const unsigned* idx = buf.cand_idx + (size_t)seg * cfg.cand_cap;
    const unsigned long long thresh = radix_select_desc(n, k, hist, &sh_prefix, &sh_remain, [&](unsigned i) {
      return ((unsigned long long)key[i] << 32) | (unsigned long long)(~idx[i]);
    });
    for (unsigned i = tid; i < n; i += 1024) {
      const unsigned long long comp = ((unsigned long long)key[i] << 32) | (unsigned long long)(~idx[i]);
      if (comp >= thresh) pool.push(pool_composite(cfg, sg, key[i], idx[i]));
    }
  }
  pool.flush();
}

// ---- many-way episodes: last cls-tower GroupNorm + ReLU + class-conditional conv + scan in ONE pass ------------------------------
// With hundreds of classes the logits are the largest tensor of the step (866 classes x 22 400 locations x 4 B = 77.6 MB per
// image) and exist only to be thresholded: written by the conv, read back by the scan.  Here a wave keeps the normalised 32 x 256
// activations of 32 locations in registers (the operand scheme of gn_logits_kernel: HBM -> registers -> MFMA, no LDS), walks the
// class codes in 32-class tiles (16 MFMAs each, code fragments straight from L2, the next tile's loaded into the registers the
// MFMAs have just consumed) and thresholds the 32 x 32 logits in the accumulator layout:
//   phase A  per accumulator register: logit + bias above logit(thr) - 1e-2 ?  (one v_cmp = the wave's hit mask) -> hits are
//            compacted into a wave-private LDS queue
//   phase B  the queue is processed densely, 64 hits at a time: sigmoid, quality, the exact fp32 threshold tests of
//            decode_scan_kernel, candidates into a wave-private LDS list that reserves a range of the (image, level) buffer with
//            one global atomicAdd per ~1000 candidates.
// No block-level synchronisation at all: same-wave LDS traffic is ordered.  The arithmetic is that of gn_apply + conv_igemm +
// decode_scan (same fma / rounding of the GroupNorm, same MFMA and K order, same sigmoid): the candidate set is identical, which
// tests/test_hip_parity.py::test_many_way_fused_scan_equals_unfused checks.  The logits buffer is not written; sylph_export_head
// runs the unfused conv on demand.
// packed code rows [>= 32 n_ct][256] -> MFMA fragment order: the 1 KiB a wave loads for (class tile ct, K step ks) is contiguous,
// 16 bytes per lane (lane = 32 lh + l31 holds class 32 ct + l31, channels 16 ks + 8 lh .. + 7).  Read row-wise, the same load
// touches 32 cache lines for 32 bytes each.
__global__ void pack_code_fragments_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wf, int n_frag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte piece
  if (i >= n_frag * 64) return;
  const int lane = i & 63, f = i >> 6, ks = f & 15, ct = f >> 4, l31 = lane & 31, lh = lane >> 5;
  *reinterpret_cast<bf16x8*>(wf + (size_t)i * 8) =
      *reinterpret_cast<const bf16x8*>(w + ((size_t)ct * 32 + l31) * 256 + ks * 16 + lh * 8);
}

constexpr int LS_QCAP = 1024;  // hits of one class tile of a wave: 64 lanes x 16 scores
constexpr int LS_CCAP = 1024;  // candidates a wave collects before it reserves a global range

__global__ __launch_bounds__(256, 2) void logits_scan_kernel(const bf16_t* __restrict__ x, int ld, const float2* __restrict__ coef,
                                                             const bf16_t* __restrict__ w /* fragment order */,
                                                             const float* __restrict__ bias_scan /* -inf past N */,
                                                             const SegDesc* __restrict__ segs, const int2* __restrict__ tiles,
                                                             int n_tiles, const float* __restrict__ pred, int pred_ld,
                                                             const DecodeCfg cfg, const DecodeBuffers buf) {
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  typedef short s16x2v __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  __shared__ __attribute__((aligned(16))) float cf[4][512];  // per wave: (a0, a1, b0, b1) per channel pair of its current segment
  __shared__ unsigned q_x[4][LS_QCAP], q_id[4][LS_QCAP];      // hit queue: logit bits, (row of the group << 16) | class
  __shared__ unsigned c_key[4][LS_CCAP], c_idx[4][LS_CCAP];   // candidates of the wave's current segment
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int N = cfg.num_classes, n_ct = (N + 31) >> 5;
  const float thr = cfg.pre_nms_thresh;
  const float x_min = (thr > 0.f && thr < 1.f) ? logf(thr / (1.f - thr)) - 1e-2f : (thr >= 1.f ? INFINITY : -INFINITY);
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned cn = 0u;  // candidates in the wave's list (wave-uniform)
  int cur_seg = -1;
  auto flush = [&]() {  // wave-uniform
    if (cn == 0u) return;
    unsigned base = 0u;
    if (lane == 0) base = atomicAdd(&buf.cand_count[cur_seg], cn);
    base = __shfl(base, 0);
    for (unsigned i = lane; i < cn; i += 64) {
      const unsigned pos = base + i;
      if (pos < (unsigned)cfg.cand_cap) {
        buf.cand_key[(size_t)cur_seg * cfg.cand_cap + pos] = c_key[wave][i];
        buf.cand_idx[(size_t)cur_seg * cfg.cand_cap + pos] = c_idx[wave][i];
      }
    }
    cn = 0u;
  };
  bf16x8 Wf[16];  // A operand: lane (class l31 of the tile, k half lh); class tile 0 now, then always the NEXT tile's (see below)
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) Wf[ks] = *reinterpret_cast<const bf16x8*>(w + ((size_t)ks * 64 + lane) * 8);
  float4 bnext[4];  // biases of the next class tile, in the accumulator layout
#pragma unroll
  for (int q = 0; q < 4; ++q) bnext[q] = *reinterpret_cast<const float4*>(bias_scan + 8 * q + 4 * lh);
  const int n_groups = n_tiles * 4, stride = gridDim.x * 4;
  for (int g = blockIdx.x * 4 + wave; g < n_groups; g += stride) {
    const int2 tl = tiles[g >> 2];
    const int seg = tl.x, r0 = tl.y + (g & 3) * 32;
    const SegDesc& sd = segs[seg];
    const int nrows = sd.out_H * sd.out_W;
    if (r0 >= nrows) continue;  // wave-uniform
    if (seg != cur_seg) {
      flush();
      cur_seg = seg;
      const float2* cp = coef + (size_t)seg * 256;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pr = lane + 64 * i;  // channel pair
        const float2 c0 = cp[2 * pr], c1 = cp[2 * pr + 1];
        *reinterpret_cast<float4*>(&cf[wave][4 * pr]) = make_float4(c0.x, c1.x, c0.y, c1.y);
      }
    }
    const int row = r0 + l31;
    const bool valid = row < nrows;
    const size_t grow = (size_t)(sd.out_row0 + (valid ? row : nrows - 1));
    const bf16_t* xp = x + grow * ld + lh * 8;
    u32x4 yv[16];  // relu(GN(x)) of (row l31, channels 16 ks + 8 lh ..) as bf16: the B operand of every class tile
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) yv[ks] = *reinterpret_cast<const u32x4*>(xp + ks * 16);
    // quality (centerness / IoU) of the lane's row, computed once per row group; rows past the segment never hit
    const float q_row = quality_from(cfg.quality_mode != 1 ? pred[grow * pred_ld + 4] : 0.f,
                                     cfg.quality_mode != 0 ? pred[grow * pred_ld + 5] : 0.f, cfg.quality_mode);
    const float x_min_row = valid ? x_min : INFINITY;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float* cq = &cf[wave][(ks * 16 + lh * 8) * 2];  // 4 channel pairs x (a0, a1, b0, b1)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 c4 = *reinterpret_cast<const float4*>(cq + 4 * e);
        const f32x2v xf = {__uint_as_float(yv[ks][e] << 16), __uint_as_float(yv[ks][e] & 0xffff0000u)};
        const f32x2v av = {c4.x, c4.y}, bv = {c4.z, c4.w};
        const f32x2v r = __builtin_elementwise_fma(xf, av, bv);
        bf16x2 pk;
        pk[0] = (bf16_t)r[0];
        pk[1] = (bf16_t)r[1];
        const s16x2v z = {0, 0};
        yv[ks][e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2v, pk), z));  // ReLU on the bf16 pair
      }
    }
    for (int ct = 0; ct < n_ct; ++ct) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int ct_next = ct + 1 < n_ct ? ct + 1 : 0;  // the last tile reloads tile 0: the next row group starts with it
      const bf16_t* wn = w + ((size_t)ct_next * 16 * 64 + lane) * 8;
      float4 bcur[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bcur[q] = bnext[q];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[ks], __builtin_bit_cast(bf16x8, yv[ks]), acc, 0, 0, 0);
        Wf[ks] = *reinterpret_cast<const bf16x8*>(wn + (size_t)ks * 64 * 8);
      }
      // the NEXT tile's biases ride behind its code fragments: nothing the epilogue below waits for is younger than them
#pragma unroll
      for (int q = 0; q < 4; ++q) bnext[q] = *reinterpret_cast<const float4*>(bias_scan + ct_next * 32 + 8 * q + 4 * lh);
      // phase A: D^T layout, register 4q + e of a lane is class 32 ct + 8q + 4 lh + e of row l31.  Classes past N carry a bias
      // of -inf (bias_scan) and rows past the segment a bound of +inf: one add and one compare per score.
      unsigned qn = 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float bq[4] = {bcur[q].x, bcur[q].y, bcur[q].z, bcur[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xv = acc[4 * q + e] + bq[e];
          const bool hit = xv > x_min_row;
          const unsigned long long m = __ballot(hit);
          if (hit) {
            const unsigned pos = qn + (unsigned)__popcll(m & below);
            q_x[wave][pos] = __float_as_uint(xv);
            q_id[wave][pos] = ((unsigned)l31 << 16) | (unsigned)(ct * 32 + 8 * q + 4 * lh + e);
          }
          qn += (unsigned)__popcll(m);
        }
      }
      // phase B
      for (unsigned i0 = 0; i0 < qn; i0 += 64) {
        const unsigned i = i0 + lane;
        const bool act = i < qn;
        const unsigned id = act ? q_id[wave][i] : 0u;
        const float xv = act ? __uint_as_float(q_x[wave][i]) : 0.f;
        const int rl = (int)(id >> 16), cls = (int)(id & 0xffffu);
        const float qr = __shfl(q_row, rl);
        const float p = sigmoid_f(xv);
        bool pass = act && p > thr;
        const float sc = p * qr;
        if (cfg.thresh_with_ctr) pass = pass && sc > thr;
        const unsigned long long m = __ballot(pass);
        const unsigned cnt = (unsigned)__popcll(m);
        if (cnt == 0u) continue;
        if (cn + cnt > (unsigned)LS_CCAP) flush();
        if (pass) {
          const unsigned pos = cn + (unsigned)__popcll(m & below);
          c_key[wave][pos] = __float_as_uint(sc);
          c_idx[wave][pos] = (unsigned)((r0 + rl) * N + cls);
        }
        cn += cnt;
      }
    }
  }
  flush();
}

int launch_logits_scan(const void* x, int ld, const float2* coef, const void* w, void* wf_ws, const float* bias_scan,
                       const SegDesc* segs, const int2* tiles, int n_tiles, const float* pred, int pred_ld, const DecodeCfg& cfg,
                       const DecodeBuffers& buf, int nseg, hipStream_t s) {
  if (cfg.num_classes <= 0 || cfg.num_classes >= 65536 || n_tiles <= 0) return (int)hipErrorInvalidValue;
  (void)hipMemsetAsync(buf.cand_count, 0, sizeof(unsigned) * nseg, s);
  const int n_frag = ((cfg.num_classes + 31) / 32) * 16;
  hipLaunchKernelGGL(pack_code_fragments_kernel, dim3((n_frag * 64 + 255) / 256), dim3(256), 0, s, (const bf16_t*)w, (bf16_t*)wf_ws,
                     n_frag);
  const int grid = n_tiles < 2048 ? n_tiles : 2048;
  hipLaunchKernelGGL(logits_scan_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ld, coef, (const bf16_t*)wf_ws, bias_scan,
                     segs, tiles, n_tiles, pred, pred_ld, cfg, buf);
  return (int)hipGetLastError();
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
  // Bitonic network, descending.  Round 6: the strides 4, 2, 1 of every merge size run in REGISTERS on chunks of 8 consecutive keys
  // (P >= 64 is a multiple of 8; a chunk is one thread's): 66 barrier-separated passes over LDS instead of 91 for 8 192 keys, and the
  // three short strides cost one read + one write of the chunk instead of three.  The network -- and so the result: the keys are
  // unique -- is the one of the plain loop.
  auto cx = [](unsigned long long& a, unsigned long long& b, bool desc) {
    if ((a < b) == desc) { const unsigned long long t = a; a = b; b = t; }
  };
  auto chunk_pass = [&](unsigned size, unsigned first_stride) {  // strides first_stride (<= 4), ..., 1 of merge size `size` on every chunk
    for (unsigned c = tid; c < (P >> 3); c += 1024) {
      unsigned long long k[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) k[e] = keys[8 * c + e];
#pragma unroll
      for (unsigned stride = 4; stride > 0; stride >>= 1) {
        if (stride > first_stride) continue;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (!(e & stride)) cx(k[e], k[e + stride], (((8 * c + e) & size) == 0));
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) keys[8 * c + e] = k[e];
    }
  };
  chunk_pass(2, 1);
  chunk_pass(4, 2);   // (a thread re-reads only its own chunk: no barrier between the three register passes)
  chunk_pass(8, 4);
  __syncthreads();
  for (unsigned size = 16; size <= P; size <<= 1) {
    for (unsigned stride = size >> 1; stride >= 8; stride >>= 1) {
      for (unsigned t = tid; t < (P >> 1); t += 1024) {
        const unsigned lo = ((t / stride) * (stride << 1)) + (t % stride);
        const unsigned hi = lo + stride;
        const bool desc = ((lo & size) == 0);  // descending overall
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
    chunk_pass(size, 4);
    __syncthreads();
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

__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
  const unsigned lo = __shfl((unsigned)(v & 0xffffffffull), src);
  const unsigned hi = __shfl((unsigned)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}

// IoU > thr for two boxes of the same class: torchvision's nms arithmetic (inter / (area_i + area_j - inter), no +1)
__device__ __forceinline__ bool nms_overlap(float x1, float y1, float x2, float y2, float iarea, float4 b, float thr) {
  const float xx1 = fmaxf(x1, b.x), yy1 = fmaxf(y1, b.y);
  const float xx2 = fminf(x2, b.z), yy2 = fminf(y2, b.w);
  const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
  const float inter = w * h;
  const float jarea = (b.z - b.x) * (b.w - b.y);
  return inter / (iarea + jarea - inter) > thr;
}

// Class-aware greedy NMS + post-NMS keep + postprocess, one 1024-thread block per image, walking the score-sorted pool in chunks of
// 64 boxes.  Per chunk:
//   1  suppression by the boxes kept in EARLIER chunks: thread (box b = t & 63, slice t >> 6) tests box b against every 16th kept
//      box (class first; the kept box is one broadcast 16-byte load per wave), the 16 per-wave ballots are OR-ed through LDS;
//   2  the 64 x 64 suppression bits inside the chunk (four pairs per thread);
//   3  wave 0 resolves the chunk serially (a box is kept iff no kept box before it suppresses it), applies the post-NMS keep
//      (first post_nms_topk kept boxes + every later one whose score ties the K-th), rescales / clips / drops empty boxes and
//      writes the outputs; the walk stops as soon as the next chunk's best score is below the K-th kept score.
// Only the chunks the walk visits cost anything (typically 3-10 of up to 79), and there is no B x pool x pool / 64 suppression
// matrix in HBM: rounds 1-3 built that matrix for ALL chunk pairs in a separate launch (0.19 ms at the headline shape, 0.5 GB).
__global__ __launch_bounds__(1024) void nms_kernel(const DecodeCfg cfg, const DecodeBuffers buf, const ImageOut* __restrict__ img_out,
                                                   float* out_boxes, float* out_scores, int* out_classes, int* out_levels,
                                                   float* out_locations, int* out_cand, int* out_counts, int* status_out, int clear_cands) {
  extern __shared__ unsigned short kept_pos[];  // [pool_cap]: pool positions of the boxes kept so far
  __shared__ unsigned long long s_part[16];
  __shared__ unsigned s_d[64][2];                // suppression bits inside the chunk: word i = boxes of the chunk suppressed by box i
  __shared__ float4 s_box4[64];
  __shared__ int s_cls4[64];
  __shared__ int s_kept_total, s_stop;
  const int img = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned n = buf.pool_count[img];
  if (n > (unsigned)cfg.pool_cap) n = cfg.pool_cap;
  const int nw = (int)((n + 63) / 64);
  const size_t base = (size_t)img * cfg.pool_cap;
  const ImageOut io = img_out[img];
  const int K = cfg.post_nms_topk;
  const float thr = cfg.nms_thresh;
  const bool nms_on = thr > 0.f;
  int kept_total = 0, nout = 0;  // wave 0 keeps the running state; kept_total is mirrored in LDS for the other waves
  float kth = -1.f;
  bool truncated = false;
  if (t == 0) { s_kept_total = 0; s_stop = 0; }
  __syncthreads();
  for (int c = 0; c < nw; ++c) {
    const unsigned b0 = c * 64;
    // the chunk's boxes -> LDS (wave 0), suppression words cleared
    if (wave == 0) {
      const unsigned bi = b0 + lane;
      if (bi < n) {
        s_box4[lane] = *reinterpret_cast<const float4*>(buf.s_box + (base + bi) * 4);
        s_cls4[lane] = buf.s_cls[base + bi];
      } else {
        s_box4[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_cls4[lane] = -1 - lane;  // matches nothing
      }
      s_d[lane][0] = 0u; s_d[lane][1] = 0u;
    }
    __syncthreads();
    const int ktot = s_kept_total;
    const float4 mb = s_box4[lane];
    const int mc = s_cls4[lane];
    bool sup = false;
    if (nms_on) {
      // 1: against the kept boxes of earlier chunks (kept box = the earlier, higher-scored one: "i" of the pair)
      for (int k = wave; k < ktot; k += 16) {
        const size_t kp = base + kept_pos[k];
        if (buf.s_cls[kp] != mc) continue;
        const float4 kb = *reinterpret_cast<const float4*>(buf.s_box + kp * 4);
        const float karea = (kb.z - kb.x) * (kb.w - kb.y);
        sup = sup || nms_overlap(kb.x, kb.y, kb.z, kb.w, karea, mb, thr);
      }
      // 2: inside the chunk: thread -> box i = t >> 4 against boxes j = 4 (t & 15) .. + 3, j > i
      const int i = t >> 4, jq = t & 15;
      const float4 ib = s_box4[i];
      const int ic = s_cls4[i];
      const float iarea = (ib.z - ib.x) * (ib.w - ib.y);
      unsigned bits = 0u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = 4 * jq + e;
        if (j > i && s_cls4[j] == ic && nms_overlap(ib.x, ib.y, ib.z, ib.w, iarea, s_box4[j], thr)) bits |= 1u << (j & 31);
      }
      if (bits) atomicOr(&s_d[i][jq >> 3], bits);
    }
    const unsigned long long part = __ballot(sup);
    if (lane == 0) s_part[wave] = part;
    __syncthreads();
    if (wave == 0) {
      // 3: serial resolve + emit (the arithmetic and order of the former nms_reduce_kernel)
      const unsigned bi = b0 + lane;
      const bool valid = bi < n;
      unsigned long long cur = 0ull;
#pragma unroll
      for (int w = 0; w < 16; ++w) cur |= s_part[w];
      if (n - b0 < 64u) cur |= ~0ull << (n - b0);
      const unsigned long long d = ((unsigned long long)s_d[lane][1] << 32) | (unsigned long long)s_d[lane][0];
      unsigned long long keptmask = 0ull;
      for (int b = 0; b < 64; ++b) {
        const unsigned long long db = shfl_u64(d, b);
        if (!((cur >> b) & 1ull)) {
          keptmask |= 1ull << b;
          cur |= db;
        }
      }
      const int nk = __popcll(keptmask);
      const bool is_kept = (keptmask >> lane) & 1ull;
      const int rank = kept_total + __popcll(keptmask & ((1ull << lane) - 1ull));
      if (is_kept) kept_pos[rank] = (unsigned short)bi;
      const float score = valid ? buf.s_score[base + bi] : 0.f;
      if (K > 0 && kth < 0.f && kept_total + nk >= K) {
        const unsigned long long sel = __ballot(is_kept && rank == K - 1);
        const int src = __ffsll((long long)sel) - 1;
        kth = __shfl(score, src);
      }
      bool emit = is_kept && (K <= 0 || rank < K || score >= kth);
      float bx1 = 0.f, by1 = 0.f, bx2 = 0.f, by2 = 0.f;
      if (emit) {
        bx1 = mb.x * io.sx; by1 = mb.y * io.sy;
        bx2 = mb.z * io.sx; by2 = mb.w * io.sy;
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
          out_classes[o] = mc;
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
      bool stop = false;
      if (K > 0 && kept_total >= K && c + 1 < nw) {
        const float next_score = buf.s_score[base + b0 + 64];
        stop = next_score < kth;
      }
      if (lane == 0) { s_kept_total = kept_total; s_stop = stop ? 1 : 0; }
    }
    __syncthreads();
    if (s_stop) break;
  }
  if (wave == 0) {
    if (__any(truncated) && lane == 0) atomicOr(buf.status, 2);
    if (lane == 0) out_counts[img] = nout < cfg.max_out ? nout : cfg.max_out;
  }
  // Self-cleaning workspace (round 6): this launch is the last reader of the image's counters and selection histograms, so it leaves
  // them zero for the next decode of the plan -- the four hipMemsetAsync dispatches in front of the scan / the histogram (about 35 us of
  // a batch-1 step) are gone.  The status word is handed to the caller and cleared by the last block to finish.
  const int L = cfg.nlevels;
  __syncthreads();
  if (t == 0) buf.pool_count[img] = 0u;
  if (clear_cands && t < L) buf.cand_count[(size_t)img * L + t] = 0u;  // (fused many-way scan: its launcher clears them itself)
  unsigned* ws = buf.sel_ws + (size_t)img * L * SEL_WS;
  for (int i = t; i < L * SEL_WS; i += 1024) ws[i] = 0u;
  if (t == 0) {
    __threadfence();
    const int done = atomicAdd(buf.status + 1, 1);
    if (done == (int)gridDim.x - 1) {
      __threadfence();
      const int st = atomicExch(buf.status, 0);
      if (status_out) *status_out = st;
      buf.status[1] = 0;
    }
  }
}

int launch_decode(const DecodeCfg& cfg, const DecodeSeg* segs_dev, int nseg, int max_nloc, int B, int nw_bound,
                  const float* logits, const float* pred, int pred_ld, const DecodeBuffers& buf,
                  const ImageOut* img_out_dev, float* out_boxes, float* out_scores, int* out_classes,
                  int* out_levels, float* out_locations, int* out_cand, int* out_counts, int* status_out, bool candidates_ready, hipStream_t s) {
  // pool_count, cand_count, sel_ws and the status word are zero on entry: zeroed when the plan's decode buffers are built and left zero
  // by the previous decode's nms_kernel
  if (!candidates_ready) {  // else logits_scan_kernel has filled the candidate buffers of this batch
    // locations per block: at least SCAN_ROWS, and enough for ~16 K scores (a 5-class episode has 8 score slots per location: 64
    // locations would be half a round of the block, and 84 000 such blocks a dispatch-bound launch)
    const int groups = (cfg.num_classes + 3) / 4;
    int rows_per_block = (4096 + groups - 1) / groups;
    // small batches (round 6): at 2 048 locations per block one 800 x 1333 image is 13 blocks (23 us); aim at ~512 blocks instead
    // (the order of the candidate buffer is irrelevant, see the kernel).  max_nloc is the P3 level: a pyramid is ~4/3 of it per image
    static const int scan_blocks = getenv("SYLPH_SCAN_BLOCKS") ? atoi(getenv("SYLPH_SCAN_BLOCKS")) : 512;  // 0: the 2 048-location blocks everywhere
    const long est_rows = (long)max_nloc * B * 4 / 3;
    const int rpb_small = scan_blocks > 0 ? (int)((est_rows / scan_blocks + 63) / 64 * 64) : rows_per_block;
    if (rpb_small < rows_per_block) rows_per_block = rpb_small;
    if (rows_per_block < SCAN_ROWS) rows_per_block = SCAN_ROWS;
    dim3 g1((max_nloc + rows_per_block - 1) / rows_per_block, nseg);
    // the scan reads the logits 16 bytes at a time: rows are padded to a multiple of 32 classes by ensure_logits()
    if ((cfg.logits_ld & 3) != 0 || (reinterpret_cast<uintptr_t>(logits) & 15) != 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(decode_scan_kernel, g1, dim3(256), 0, s, cfg, segs_dev, logits, pred, pred_ld, buf, rows_per_block);
  }
  // blocks per level of the selection: one per 64 Ki candidate slots, so a few-way plan (84 000 slots) runs 2 and an
  // 866-way plan (1.8 M slots) 28
  int parts = (cfg.cand_cap + 65535) / 65536;
  parts = parts < 1 ? 1 : (parts > 32 ? 32 : parts);
  hipLaunchKernelGGL(decode_hist_kernel, dim3(parts, nseg), dim3(1024), 0, s, cfg, buf);
  hipLaunchKernelGGL(decode_partition_kernel, dim3(parts, nseg), dim3(1024), 0, s, cfg, segs_dev, buf);
  hipLaunchKernelGGL(decode_finish_kernel, dim3(nseg), dim3(1024), 0, s, cfg, segs_dev, buf);
  hipLaunchKernelGGL(decode_sort_kernel, dim3(B), dim3(1024), sizeof(unsigned long long) * cfg.pool_cap, s, cfg,
                     segs_dev, pred, pred_ld, buf);
  (void)nw_bound;
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(1024), sizeof(unsigned short) * cfg.pool_cap, s, cfg, buf, img_out_dev, out_boxes, out_scores,
                     out_classes, out_levels, out_locations, out_cand, out_counts, status_out, candidates_ready ? 0 : 1);
  return (int)hipGetLastError();
}

}  // namespace sylph
