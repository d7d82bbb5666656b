// ROIEncoder code generator (LVIS variant) on gfx950: the pieces that are not convolutions.
// All of it is tiny (the shots of a few classes, 49 positions, 256 channels): fp32 VALU math with
// wavefront/LDS reductions, one workgroup per support shot / token / output feature.
//
// Reference arithmetic followed (paths relative to /root/reference):
//   sylph/modeling/code_generator/utils.py:143-165   context = mean_l adaptive_avg_pool_7x7(feature_l)
//   sylph/modeling/code_generator/utils.py:70-103    MS_CAM: x * sigmoid(local(ctx) + global(avgpool(ctx)))
//   sylph/modeling/code_generator/roi_encoder.py:26-115  Tokenizer FC / HyperNetworkHead FC stacks
//   torch.nn.TransformerEncoderLayer (post-norm, ReLU): residual + LayerNorm; with the reference's
//   batch_first=False call the attention runs over the class axis, which has length 1 at inference
//   (roi_encoder.py:184-186; forward_class_code asserts one class per call), so softmax == 1 and
//   self-attention reduces to out_proj(v_proj(x)) (folded into one matrix on the host).
#include "common.h"
#include "kernels.h"

namespace sylph {

// ---- context: grid (49, S), block 256 (channel per thread) ---------------------------------------
template <typename T>
__global__ void adaptive_context_kernel(const T* __restrict__ feats, int ld, const LevelDesc* __restrict__ lv,
                                        int nlevels, int out_size, int C, float* __restrict__ ctx) {
  const int s = blockIdx.y, bin = blockIdx.x;
  const int ph = bin / out_size, pw = bin - ph * out_size;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (int l = 0; l < nlevels; ++l) {
      const LevelDesc d = lv[s * nlevels + l];
      const int ys = (ph * d.H) / out_size, ye = ((ph + 1) * d.H + out_size - 1) / out_size;
      const int xs = (pw * d.W) / out_size, xe = ((pw + 1) * d.W + out_size - 1) / out_size;
      float sum = 0.f;
      for (int y = ys; y < ye; ++y)
        for (int x = xs; x < xe; ++x) sum += Cvt<T>::to_f(feats[(size_t)(d.row0 + y * d.W + x) * ld + c]);
      acc += sum / (float)((ye - ys) * (xe - xs));
    }
    ctx[((size_t)s * out_size * out_size + bin) * C + c] = acc / (float)nlevels;
  }
}

int launch_adaptive_context(DType dt, const void* feats, int ld, const LevelDesc* lv_dev, int nlevels, int S,
                            int out_size, float* ctx, hipStream_t s) {
  dim3 grid(out_size * out_size, S), block(256);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(adaptive_context_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)feats, ld, lv_dev, nlevels,
                       out_size, 256, ctx);
  else
    hipLaunchKernelGGL(adaptive_context_kernel<float>, grid, block, 0, s, (const float*)feats, ld, lv_dev, nlevels,
                       out_size, 256, ctx);
  return (int)hipGetLastError();
}

// ---- MS-CAM gate: one block (256 threads) per shot, C = 256, inter = 64, 49 positions -------------
constexpr int MC = 256, MI = 64, MP = 49;

__device__ __forceinline__ float gn8_lane(float v, float gamma, float beta) {  // 8 consecutive lanes = one group
  float s = v;
  for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s * 0.125f, d = v - mean;
  float q = d * d;
  for (int o = 4; o > 0; o >>= 1) q += __shfl_xor(q, o);
  return d * (1.0f / sqrtf(q * 0.125f + 1e-5f)) * gamma + beta;
}

template <typename T>
__global__ __launch_bounds__(256) void mscam_kernel(const float* __restrict__ ctx_g, T* __restrict__ x,
                                                    const MsCamWeights w) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* ctx = sm;                 // [49][256]
  float* l1 = ctx + MP * MC;       // [49][64]
  float* l2 = l1 + MP * MI;        // [49][256]
  float* vec = l2 + MP * MC;       // [256] scratch
  float* stat = vec + MC;          // [64]
  const int s = blockIdx.x, t = threadIdx.x;
  const float* cg = ctx_g + (size_t)s * MP * MC;
  for (int i = t; i < MP * MC; i += 256) ctx[i] = cg[i];
  __syncthreads();
  // global descriptor: mean over positions
  {
    float g = 0.f;
    for (int p = 0; p < MP; ++p) g += ctx[p * MC + t];
    vec[t] = g / (float)MP;
  }
  // local conv1x1 256->64: thread (j = t&63, pg = t>>6), positions pg, pg+4, ...
  {
    const int j = t & 63, pg = t >> 6;
    float acc[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) acc[i] = 0.f;
    const float* wr = w.l_w1 + (size_t)j * MC;
    for (int k = 0; k < MC; ++k) {
      const float wk = wr[k];
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        const int p = pg + 4 * i;
        if (p < MP) acc[i] = fmaf(ctx[p * MC + k], wk, acc[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      const int p = pg + 4 * i;
      if (p < MP) l1[p * MI + j] = acc[i] + w.l_b1[j];
    }
  }
  __syncthreads();
  // GroupNorm(32, 64): 2 channels x 49 positions per group
  if (t < 32) {
    float s1 = 0.f;
    for (int p = 0; p < MP; ++p) s1 += l1[p * MI + 2 * t] + l1[p * MI + 2 * t + 1];
    const float mean = s1 / (2.f * MP);
    float q = 0.f;
    for (int p = 0; p < MP; ++p) {
      const float a = l1[p * MI + 2 * t] - mean, b = l1[p * MI + 2 * t + 1] - mean;
      q += a * a + b * b;
    }
    stat[2 * t] = mean;
    stat[2 * t + 1] = 1.0f / sqrtf(q / (2.f * MP) + 1e-5f);
  }
  __syncthreads();
  for (int i = t; i < MP * MI; i += 256) {
    const int j = i & 63, g = j >> 1;
    const float v = (l1[i] - stat[2 * g]) * stat[2 * g + 1] * w.l_g1[j] + w.l_be1[j];
    l1[i] = v > 0.f ? v : 0.f;
  }
  __syncthreads();
  // local conv1x1 64->256: thread = output channel
  {
    float wr[MI];
    const float* wp = w.l_w2 + (size_t)t * MI;
#pragma unroll
    for (int k = 0; k < MI; ++k) wr[k] = wp[k];
    const float b = w.l_b2[t];
    for (int p = 0; p < MP; ++p) {
      float a = b;
#pragma unroll
      for (int k = 0; k < MI; ++k) a = fmaf(l1[p * MI + k], wr[k], a);
      l2[p * MC + t] = a;
    }
  }
  // GroupNorm(32, 256) on l2: 8 channels x 49 positions; per-channel sums then 8-lane reduce
  float lmean, lrstd;
  {
    float s1 = 0.f;
    for (int p = 0; p < MP; ++p) s1 += l2[p * MC + t];
    for (int o = 4; o > 0; o >>= 1) s1 += __shfl_xor(s1, o);
    lmean = s1 / (8.f * MP);
    float q = 0.f;
    for (int p = 0; p < MP; ++p) { const float d = l2[p * MC + t] - lmean; q = fmaf(d, d, q); }
    for (int o = 4; o > 0; o >>= 1) q += __shfl_xor(q, o);
    lrstd = 1.0f / sqrtf(q / (8.f * MP) + 1e-5f);
  }
  // global branch on vec (256) -> 64 -> 256, GroupNorm over (C/32) x 1 x 1 elements
  __syncthreads();
  float g1 = 0.f;
  if (t < MI) {
    const float* wr = w.g_w1 + (size_t)t * MC;
    float a = w.g_b1[t];
    for (int k = 0; k < MC; ++k) a = fmaf(vec[k], wr[k], a);
    const float other = __shfl_xor(a, 1);
    const float mean = 0.5f * (a + other);
    const float var = 0.5f * ((a - mean) * (a - mean) + (other - mean) * (other - mean));
    g1 = (a - mean) * (1.0f / sqrtf(var + 1e-5f)) * w.g_g1[t] + w.g_be1[t];
    g1 = g1 > 0.f ? g1 : 0.f;
    stat[t] = g1;
  }
  __syncthreads();
  float g2;
  {
    const float* wr = w.g_w2 + (size_t)t * MI;
    float a = w.g_b2[t];
    for (int k = 0; k < MI; ++k) a = fmaf(stat[k], wr[k], a);
    g2 = gn8_lane(a, w.g_g2[t], w.g_be2[t]);
  }
  // gate
  const float ga = lrstd * w.l_g2[t], gb = w.l_be2[t] - lmean * ga;
  T* xs = x + (size_t)s * MP * MC;
  for (int p = 0; p < MP; ++p) {
    const float z = fmaf(l2[p * MC + t], ga, gb) + g2;
    const float wei = 1.f / (1.f + expf(-z));
    xs[p * MC + t] = Cvt<T>::from_f(Cvt<T>::to_f(xs[p * MC + t]) * wei);
  }
}

int launch_mscam(DType dt, const float* ctx, void* x, int S, const MsCamWeights& w, hipStream_t s) {
  const size_t lds = (size_t)(MP * MC * 2 + MP * MI + MC + 64) * sizeof(float);
  if (dt == DT_BF16) {
    (void)hipFuncSetAttribute((const void*)mscam_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mscam_kernel<bf16_t>, dim3(S), dim3(256), lds, s, ctx, (bf16_t*)x, w);
  } else {
    (void)hipFuncSetAttribute((const void*)mscam_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mscam_kernel<float>, dim3(S), dim3(256), lds, s, ctx, (float*)x, w);
  }
  return (int)hipGetLastError();
}

// ---- y[s][o] = act(x[s][:] . W[o][:] + b[o]) (+ add); one block per output feature and 16 rows ----
template <typename XT>
__global__ __launch_bounds__(256) void linear_kernel(const XT* __restrict__ x, int ldx, int S,
                                                     const float* __restrict__ W, const float* __restrict__ b,
                                                     int K, float* __restrict__ y, int ldy, int relu, float add) {
  const int o = blockIdx.x, t = threadIdx.x;
  {  // rows [16 blockIdx.y, +16): a row's arithmetic does not depend on how many rows the call has
    const int s0 = blockIdx.y * 16;
    x += (size_t)s0 * ldx;
    y += (size_t)s0 * ldy;
    S = min(16, S - s0);
  }
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const float* wr = W + (size_t)o * K;
  for (int k = t; k < K; k += 256) {
    const float wk = wr[k];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < S) acc[i] = fmaf(Cvt<XT>::to_f(x[(size_t)i * ldx + k]), wk, acc[i]);
  }
  __shared__ float red[4][16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float v = acc[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((t & 63) == 0) red[t >> 6][i] = v;
  }
  __syncthreads();
  if (t < S) {
    float v = red[0][t] + red[1][t] + red[2][t] + red[3][t] + (b ? b[o] : 0.f);
    if (relu) v = v > 0.f ? v : 0.f;
    y[(size_t)t * ldy + o] = v + add;
  }
}

int launch_linear(int x_is_bf16, const void* x, int ldx, int S, const float* W, const float* b, int K, int O, float* y,
                  int ldy, int relu, float add, hipStream_t s) {
  if (S < 1) return -1;
  const dim3 grid(O, (S + 15) / 16);
  if (x_is_bf16)
    hipLaunchKernelGGL(linear_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ldx, S, W, b, K, y, ldy, relu, add);
  else
    hipLaunchKernelGGL(linear_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ldx, S, W, b, K, y, ldy, relu, add);
  return (int)hipGetLastError();
}

// ---- x[s][:] = LayerNorm(x[s][:] + r[s][:]) over E = 256; block per token -------------------------
__global__ __launch_bounds__(256) void add_layernorm_kernel(float* __restrict__ x, const float* __restrict__ r,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta) {
  const int s = blockIdx.x, t = threadIdx.x;
  __shared__ float part[4];
  const float v = x[(size_t)s * 256 + t] + r[(size_t)s * 256 + t];
  float a = v;
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if ((t & 63) == 0) part[t >> 6] = a;
  __syncthreads();
  const float mean = (part[0] + part[1] + part[2] + part[3]) * (1.f / 256.f);
  __syncthreads();
  const float d = v - mean;
  float q = d * d;
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  if ((t & 63) == 0) part[t >> 6] = q;
  __syncthreads();
  const float var = (part[0] + part[1] + part[2] + part[3]) * (1.f / 256.f);
  x[(size_t)s * 256 + t] = d * (1.0f / sqrtf(var + 1e-5f)) * gamma[t] + beta[t];
}

int launch_add_layernorm(float* x, const float* r, int S, const float* gamma, const float* beta, hipStream_t s) {
  hipLaunchKernelGGL(add_layernorm_kernel, dim3(S), dim3(256), 0, s, x, r, gamma, beta);
  return (int)hipGetLastError();
}

// ---- out[:] = mean_s x[s][:] (E = 256) --------------------------------------------------------------
// one block per class: the S tokens of class c are rows [c S, (c + 1) S)
__global__ void mean_tokens_kernel(const float* __restrict__ x, int S, float* __restrict__ out) {
  const int t = threadIdx.x, c = blockIdx.x;
  float a = 0.f;
  for (int s = 0; s < S; ++s) a += x[((size_t)c * S + s) * 256 + t];
  out[(size_t)c * 256 + t] = a / (float)S;
}

int launch_mean_tokens(const float* x, int n_classes, int S, float* out, hipStream_t s) {
  hipLaunchKernelGGL(mean_tokens_kernel, dim3(n_classes), dim3(256), 0, s, x, S, out);
  return (int)hipGetLastError();
}

}  // namespace sylph
