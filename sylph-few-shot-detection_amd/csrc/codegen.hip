// Support-set path of the Sylph hypernetwork ("code generator") on gfx950: FPN level assignment +
// ROIAlignV2 gather, per-class code aggregation over positions and shots (wavefront/block
// reductions), and class-code normalisation (GroupNorm(32) on a 256-vector, L2 normalise, scale).
// Compiled with -ffp-contract=off so sampling coordinates round like the fp32 reference.
//
// Reference arithmetic followed (paths relative to /root/reference):
//   sylph/modeling/code_generator/code_generator.py:341-348,928-930  ROIPooler(7, scales, 0, "ROIAlignV2")
//     (detectron2 assign_boxes_to_levels + torchvision roi_align, aligned=True, adaptive sampling)
//   sylph/modeling/code_generator/code_generator.py:954-967          conv/bias features, BIAS_L2_NORM
//   sylph/modeling/code_generator/utils.py:51-67                     GlobalAdaptiveAvgPool2d
//   sylph/modeling/code_generator/code_generator.py:778-829          compute_code (uniform 1/S)
//   sylph/modeling/code_generator/code_generator.py:832-875          normalize_code / process_bias
#include "common.h"
#include "kernels.h"

namespace sylph {

template <typename T>
__device__ __forceinline__ float bilinear_at(const T* __restrict__ feat, int ld, int row0, int H, int W, float y,
                                             float x, int c) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - (float)y_low, lx = x - (float)x_low;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  const float v1 = Cvt<T>::to_f(feat[(size_t)(row0 + y_low * W + x_low) * ld + c]);
  const float v2 = Cvt<T>::to_f(feat[(size_t)(row0 + y_low * W + x_high) * ld + c]);
  const float v3 = Cvt<T>::to_f(feat[(size_t)(row0 + y_high * W + x_low) * ld + c]);
  const float v4 = Cvt<T>::to_f(feat[(size_t)(row0 + y_high * W + x_high) * ld + c]);
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

// grid (out*out, S), block C threads (channel per thread: channel-contiguous gathers)
template <typename T>
__global__ void roi_align_kernel(const T* __restrict__ feats, int ld, const LevelDesc* __restrict__ lv, int nlevels,
                                 const float* __restrict__ boxes, int out_size, int C, T* __restrict__ out) {
  const int s = blockIdx.y, bin = blockIdx.x;
  const int ph = bin / out_size, pw = bin - ph * out_size;
  const float bx1 = boxes[s * 4 + 0], by1 = boxes[s * 4 + 1], bx2 = boxes[s * 4 + 2], by2 = boxes[s * 4 + 3];
  int lvl = 0;
  if (nlevels > 1) {
    const float area = (bx2 - bx1) * (by2 - by1);
    const float sz = sqrtf(area);
    float l = floorf(4.0f + log2f(sz / 224.0f + 1e-8f));
    const float min_level = -log2f(lv[0].scale);  // 3 for stride 8
    l = fminf(fmaxf(l, min_level), min_level + (float)(nlevels - 1));
    lvl = (int)(l - min_level);
  }
  const LevelDesc d = lv[s * nlevels + lvl];
  const float x1 = bx1 * d.scale - 0.5f, y1 = by1 * d.scale - 0.5f;
  const float x2 = bx2 * d.scale - 0.5f, y2 = by2 * d.scale - 0.5f;
  const float rw = x2 - x1, rh = y2 - y1;
  const float bw = rw / (float)out_size, bh = rh / (float)out_size;
  const int gh = (int)ceilf(bh), gw = (int)ceilf(bw);
  const float count = (float)max(gh * gw, 1);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      const float yy = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float xx = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
        acc += bilinear_at<T>(feats, ld, d.row0, d.H, d.W, yy, xx, c);
      }
    }
    out[((size_t)s * out_size * out_size + bin) * C + c] = Cvt<T>::from_f(acc / count);
  }
}

int launch_roi_align(DType dt, const void* feats, int ld, const LevelDesc* lv_dev, int nlevels,
                     const float* boxes_dev, int S, int out_size, void* out, hipStream_t s) {
  const int C = 256;
  dim3 grid(out_size * out_size, S), block(256);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(roi_align_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)feats, ld, lv_dev, nlevels,
                       boxes_dev, out_size, C, (bf16_t*)out);
  else
    hipLaunchKernelGGL(roi_align_kernel<float>, grid, block, 0, s, (const float*)feats, ld, lv_dev, nlevels, boxes_dev,
                       out_size, C, (float*)out);
  return (int)hipGetLastError();
}

// conv_out [ncls*S*npos][conv_ld] fp32, aux_out [ncls*S*npos][aux_ld] fp32 (channel ib: bias head, iw: shot-weight head, is: class-scale
// head; -1 = absent) -> code_out[ncls][C+1] (+ wnorm_out[ncls] with a scale head); one block per class (its S consecutive support
// images: the arithmetic of a class does not depend on how many classes share the batch).
// code_generator.py:766-829: per shot the heads are global-average-pooled, the shots are combined with uniform weights 1/S or, with a
// WEIGHT_LAYER, with softmax(pooled shot-weight logits) over the shots of the class.
__global__ __launch_bounds__(256) void codegen_tail_kernel(const float* __restrict__ conv_out, int conv_ld,
                                                           const float* __restrict__ aux_out, int aux_ld, int ib, int iw, int is, int S,
                                                           int npos, int C, int bias_l2_norm, float* __restrict__ code_out,
                                                           float* __restrict__ wnorm_out) {
  conv_out += (size_t)blockIdx.x * S * npos * conv_ld;
  aux_out += (size_t)blockIdx.x * S * npos * aux_ld;
  code_out += (size_t)blockIdx.x * (C + 1);
  __shared__ float wsh[64];  // per-shot weights (S <= 64)
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (iw >= 0) {  // softmax over the shots of the pooled logits (torch.nn.Softmax(dim=1), fp32)
      float lg = -INFINITY;
      if (lane < S) {
        float sum = 0.f;
        for (int p = 0; p < npos; ++p) sum += aux_out[((size_t)lane * npos + p) * aux_ld + iw];
        lg = sum / (float)npos;
      }
      float mx = lg;
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const float e = lane < S ? expf(lg - mx) : 0.f;
      float den = e;
      for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
      wsh[lane] = e / den;
    } else {
      wsh[lane] = 1.0f / (float)S;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float code = 0.f;
    for (int s = 0; s < S; ++s) {
      float sum = 0.f;
      for (int p = 0; p < npos; ++p) sum += conv_out[((size_t)s * npos + p) * conv_ld + c];
      code += wsh[s] * (sum / (float)npos);
    }
    code_out[c] = code;
  }
  if (threadIdx.x < 64) {  // one wave: bias and class-scale heads
    const int lane = threadIdx.x;
    float bias = 0.f, wn = 0.f;
    for (int s = 0; s < S; ++s) {
      if (ib >= 0) {
        const float v = lane < npos ? aux_out[((size_t)s * npos + lane) * aux_ld + ib] : 0.f;  // npos <= 64
        float nrm = 1.f;
        if (bias_l2_norm) {
          float sq = v * v;
          for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
          nrm = fmaxf(sqrtf(sq), 1e-12f);
        }
        float t = v / nrm;
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        bias += wsh[s] * (t / (float)npos);
      }
      if (is >= 0) {
        float t = lane < npos ? aux_out[((size_t)s * npos + lane) * aux_ld + is] : 0.f;
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        wn += wsh[s] * (t / (float)npos);
      }
    }
    if (lane == 0) {
      code_out[C] = bias;
      if (wnorm_out) wnorm_out[blockIdx.x] = wn;
    }
  }
}

int launch_codegen_tail(const float* conv_out, int conv_ld, const float* aux_out, int aux_ld, int ib, int iw, int is, int ncls, int S, int npos,
                        int C, int bias_l2_norm, float* code_out, float* wnorm_out, hipStream_t s) {
  if (npos > 64 || ncls < 1 || S > 64) return -1;
  hipLaunchKernelGGL(codegen_tail_kernel, dim3(ncls), dim3(256), 0, s, conv_out, conv_ld, aux_out, aux_ld, ib, iw, is, S, npos, C, bias_l2_norm,
                     code_out, is >= 0 ? wnorm_out : nullptr);
  return (int)hipGetLastError();
}

// codes [ncodes][C+1] in place; one block of 256 threads per code, C == 256 (8 channels / group)
__global__ __launch_bounds__(256) void normalize_codes_kernel(float* __restrict__ codes, int C,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int post_norm,
                                                              int l2_norm, float conv_scale, float bias_scale,
                                                              float bias_prior, const float* __restrict__ weight_norm) {
  float* code = codes + (size_t)blockIdx.x * (C + 1);
  const int c = threadIdx.x;
  float v = c < C ? code[c] : 0.f;
  if (post_norm && (C % 32 == 0)) {
    // groups of C/32 = 8 consecutive channels -> 8 consecutive lanes
    float s = v;
    for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / 8.f;
    const float d = v - mean;
    float q = d * d;
    for (int o = 4; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / 8.f + 1e-5f);
    v = d * rstd * gamma[c] + beta[c];
  }
  if (l2_norm) {
    __shared__ float part[4];
    float sq = v * v;
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((c & 63) == 0) part[c >> 6] = sq;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(part[0] + part[1] + part[2] + part[3]), 1e-12f);
    v = v / nrm;
  }
  if (weight_norm) v = v * weight_norm[blockIdx.x];  // x cls_weight_norm (code_generator.py:838-840)
  v = v * conv_scale;
  if (c < C) code[c] = v;
  if (c == 0) code[C] = code[C] * bias_scale + bias_prior;
}

int launch_normalize_codes(float* codes, int ncodes, int C, const float* gn_gamma, const float* gn_beta,
                           int post_norm, int l2_norm, float conv_scale, float bias_scale, float bias_prior,
                           const float* weight_norm, hipStream_t s) {
  if (C != 256) return -1;
  hipLaunchKernelGGL(normalize_codes_kernel, dim3(ncodes), dim3(256), 0, s, codes, C, gn_gamma, gn_beta, post_norm,
                     l2_norm, conv_scale, bias_scale, bias_prior, weight_norm);
  return (int)hipGetLastError();
}

// reduce_class_code (sylph/modeling/code_generator/utils.py:397-427) on packed rows, one block per class id.
// Row layout (fp32, ld >= 262): [0,256) cls_conv | 256 cls_bias | 257 acc_weight | 258 class id | 259 valid |
// 260 cls_weight_norm | 261 has_weight_norm | [262, ld) opaque payload (copied from the class's first row).
// The chunk codes of a class (already weighted by len / total_len, meta_learn_evaluation.py:176-188) are summed in ROW
// ORDER (= rank order, then arrival order: the reference's list order), acc_weight in double like the reference's Python
// floats; with divide_by_acc the sums are divided by acc_weight when |1 - acc| > 1e-6 (the cross-rank reduce), without it
// the row keeps the accumulated weight (the per-rank accumulation).  Output row c = class id c (valid 0 if absent).
__global__ __launch_bounds__(256) void reduce_codes_kernel(const float* __restrict__ rows, int n, int ld,
                                                           float* __restrict__ out, int num_classes, int divide_by_acc) {
  const int cid = blockIdx.x, t = threadIdx.x;
  __shared__ double s_acc;
  __shared__ int s_first, s_cnt;
  float sum = 0.f, sum_bias = 0.f, sum_wn = 0.f, has_wn = 0.f;
  double acc = 0.0;
  int first = -1, cnt = 0;
  for (int i = 0; i < n; ++i) {
    const float* r = rows + (size_t)i * ld;
    if (r[259] == 0.f || (int)r[258] != cid) continue;
    if (first < 0) first = i;
    ++cnt;
    sum += r[t];
    if (t == 0) { sum_bias += r[256]; acc += (double)r[257]; sum_wn += r[260]; has_wn = fmaxf(has_wn, r[261]); }
  }
  if (t == 0) { s_acc = acc; s_first = first; s_cnt = cnt; }
  __syncthreads();
  float* o = out + (size_t)cid * ld;
  if (s_cnt == 0) {
    for (int k = t; k < ld; k += 256) o[k] = 0.f;
    if (t == 0) o[258] = (float)cid;
    return;
  }
  const double a = s_acc;
  const bool div = divide_by_acc && fabs(1.0 - a) > 1e-6;
  const float af = (float)a;
  o[t] = div ? sum / af : sum;
  if (t == 0) {
    o[256] = div ? sum_bias / af : sum_bias;
    o[257] = divide_by_acc ? 1.f : af;  // plain accumulation keeps the accumulated weight for the next (cross-rank) reduce
    o[258] = (float)cid;
    o[259] = 1.f;
    o[260] = div ? sum_wn / af : sum_wn;
    o[261] = has_wn;
  }
  const float* r0 = rows + (size_t)s_first * ld;
  for (int k = 262 + t; k < ld; k += 256) o[k] = r0[k];
}

int launch_reduce_codes(const float* rows, int n, int ld, float* out, int num_classes, int divide_by_acc, hipStream_t s) {
  if (ld < 262 || num_classes <= 0) return -1;
  hipLaunchKernelGGL(reduce_codes_kernel, dim3(num_classes), dim3(256), 0, s, rows, n, ld, out, num_classes, divide_by_acc);
  return (int)hipGetLastError();
}

}  // namespace sylph
