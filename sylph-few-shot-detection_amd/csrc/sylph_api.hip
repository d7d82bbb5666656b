// Host side of libsylph_hip.so: context, weight packing, per-batch-shape execution plans (lists of
// fully resolved kernel launches over context-owned HBM buffers) and the C ABI of include/sylph_hip.h.
// No torch types, no CPU compute fallback: every stage is a HIP kernel from this directory.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/sylph_hip.h"
#include "common.h"
#include "kernels.h"

using namespace sylph;

static thread_local std::string g_err;
static int fail(const std::string& m) {
  g_err = m;
  return 1;
}
// collective.hip
hipStream_t sylph_internal_stream(sylph_ctx* c);
int sylph_internal_fail(const std::string& m) { return fail(m); }
#define HIPCHK(x)                                                                                  \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_));             \
  } while (0)
#define RET(x)                     \
  do {                             \
    int r_ = (x);                  \
    if (r_ != 0) return r_;        \
  } while (0)
// build step of plan P: its allocations belong to P; a failed build releases the whole plan so that a retry starts clean
#define BUILD(x, P)                                  \
  do {                                               \
    int r_ = (x);                                    \
    if (r_ != 0) { drop_plan(c, (P)); return r_; }   \
  } while (0)
#define KCHK(x, what)                                                                              \
  do {                                                                                             \
    int r_ = (x);                                                                                  \
    if (r_ != 0) return fail(std::string(what) + ": launch failed (" + std::to_string(r_) + ")");  \
  } while (0)

static inline uint16_t f2bf_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

typedef std::function<int(hipStream_t)> OpFn;

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct ConvLayer {
  void* w = nullptr;       // packed [Cout_pad][KH][KW][Cin]
  float* scale = nullptr;  // device, Cout_pad (may be null)
  float* shift = nullptr;
  int Cin = 0, Cout = 0, Cout_pad = 0, KH = 1, KW = 1;
};
struct GNLayer {
  float* gamma = nullptr;
  float* beta = nullptr;
};

struct Plan;

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (libImaging/Resample.c), in double like the
// original: per output index the first input index, the tap count and `ksize` 22-bit fixed-point weights.
struct PilCoeffs {
  int ksize = 0;
  std::vector<int> bounds;  // [out][2] = (first, count)
  std::vector<int> kk;      // [out][ksize]
};
static std::shared_ptr<PilCoeffs> pil_bilinear_coeffs(int in_size, int out_size) {
  auto pc = std::make_shared<PilCoeffs>();
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;  // bilinear filter support = 1
  const int ksize = (int)ceil(support) * 2 + 1;
  pc->ksize = ksize;
  pc->bounds.assign((size_t)out_size * 2, 0);
  pc->kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> k((size_t)ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      const double w = a < 1.0 ? 1.0 - a : 0.0;
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      pc->kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (double)(1 << 22)) : (int)(0.5 + k[x] * (double)(1 << 22));
    }
    pc->bounds[2 * xx] = xmin;
    pc->bounds[2 * xx + 1] = xmax;
  }
  return pc;
}

struct sylph_ctx {
  int device = 0;
  DType dt = DT_BF16;
  hipStream_t stream = nullptr;
  sylph_config cfg;
  bool finalized = false;
  int64_t bytes = 0;
  std::vector<void*> allocs;
  std::map<std::string, HostTensor> host_w;
  // packed model
  ConvLayer stem;  // 7x7 s2 stem packed for the implicit-GEMM stem loader
  void* stem_wp = nullptr;  // bf16 [64][7][8][4] for the dedicated stem kernel (stem_conv.hip)
  struct Block { ConvLayer c1, c2, c3, sc, c3sc; bool has_sc = false, fused_sc = false; };
  std::vector<std::vector<Block>> stages;  // res2..res5
  ConvLayer fpn_lat[3], fpn_out[3], p6, p7;  // index 0..2 = stage 3..5
  std::vector<ConvLayer> cls_tower, box_tower;
  std::vector<GNLayer> cls_gn, box_gn;
  std::vector<ConvLayer> pair_tower;  // cls|bbox towers stacked on Cout (layer 0 shares the input, then grouped)
  std::vector<GNLayer> pair_gn;
  bool paired = false;
  ConvLayer pred;  // bbox_pred(4) + ctrness(1) + iou_overlap(1)
  ConvLayer cls_logits;  // the base detector's own classifier (fcos.py:418-427), 1x1 or 3x3, when the checkpoint carries it
  bool has_cls_logits = false;
  void* pred_taps = nullptr;  // bf16 [64][256]: row kh * sw + kw * Cout + n = pred weight W[n][kh][kw][:] (head_fused.hip), bf16 mode only
  std::vector<float> level_scales;
  std::vector<ConvLayer> cg_tower;
  std::vector<GNLayer> cg_gn;
  ConvLayer cg_cls, cg_bias;  // cg_bias: the 1-channel heads stacked on Cout: [bias][shot weight][class scale] (those the config has)
  int cg_naux = 0, cg_ib = -1, cg_iw = -1, cg_is = -1;  // channel of each head in cg_bias's output (-1: absent)
  GNLayer cg_post;
  float cg_conv_scale = 1.f, cg_bias_scale = 1.f;
  float cg_bias_prior = 0.f;  // bias_value: -log((1 - PRIOR_PROB) / PRIOR_PROB), or the learned parameter with META_BIAS
  // ROIEncoder variant
  struct Lin { float* W = nullptr; float* b = nullptr; int K = 0, O = 0; };
  struct EncLayer { Lin attn, l1, l2; GNLayer n1, n2; };
  struct RoiEnc {
    ConvLayer pool_conv; GNLayer pool_gn;
    MsCamWeights cam;
    std::vector<ConvLayer> tok_conv; std::vector<GNLayer> tok_gn;
    std::vector<Lin> tok_fc, wh, bh;
    std::vector<EncLayer> layers;
  } re;
  bool has_backbone = false, has_head = false, has_codegen = false, has_roienc = false;
  // plans
  std::map<std::tuple<int, int, int>, std::unique_ptr<Plan>> plans;
  std::map<std::pair<int, int>, std::shared_ptr<struct PilCoeffs>> pil_cache;  // (in size, out size) -> resampling tables
  std::map<const void*, void*> hp_weights;  // conv_hpipe.hip re-packed copies of 3x3 weights, keyed by the igemm-layout pointer
  std::map<const void*, std::pair<void*, float*>> pw_weights;  // conv_pw.hip stage-image copies of 1x1 weights + scale/shift tables, keyed by the igemm-layout pointer
  void* pw_trash = nullptr;                 // conv_pw.hip trash slots (4 KiB)
  Plan* cur = nullptr;
  void* zeros = nullptr;  // 256 B of zeros (conv out-of-image taps)
  // optional per-launch timing of the MFMA conv kernel (bench.py roofline): HIP events on the launch stream
  bool prof = false;
  struct ProfRec { hipEvent_t a, b; double flops; const char* kern; };
  std::vector<ProfRec> prof_recs;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;

  // Plan cache policy: every allocation made while a plan is being built / grown is owned by that plan, and
  // plans are evicted least-recently-used first once their count or their bytes exceed the budget, so a stream of
  // distinct padded shapes (real COCO / LVIS episodes) cannot grow HBM without bound.
  Plan* alloc_owner = nullptr;
  bool debug_taps = false;  // sylph_set_debug_taps: tower layers keep their outputs in separate buffers (parity tests)
  uint64_t use_clock = 0;
  size_t max_plans = 32;
  int64_t plan_byte_budget = 0;  // 0 = set from the device size at context creation
  int dalloc(void** p, size_t n);
  void dfree_nosync(void* p) {
    if (!p) return;
    for (size_t i = 0; i < allocs.size(); ++i)
      if (allocs[i] == p) { allocs[i] = allocs.back(); allocs.pop_back(); break; }
    auto it = alloc_bytes.find(p);
    if (it != alloc_bytes.end()) { bytes -= (int64_t)it->second; alloc_bytes.erase(it); }
    (void)hipFree(p);
  }
  std::map<void*, size_t> alloc_bytes;
  void dfree(void* p);  // release one dalloc'ed buffer (the stream may still use it: drained first)
  size_t esz() const { return dt == DT_BF16 ? 2 : 4; }
};

struct Plan {
  std::vector<void*> allocs;  // device buffers owned by this plan (freed on eviction)
  int64_t bytes = 0;
  uint64_t last_use = 0;
  int B = 0, H = 0, W = 0;
  int hl[8], wl[8], off[8], Ltot = 0;
  std::vector<int> img_h, img_w;
  // backbone
  void *x0 = nullptr, *stem_out = nullptr, *pool_out = nullptr;
  void* F = nullptr;  // pyramid [B*Ltot][256]
  void* bk_trash = nullptr;  // trash slots of the fused bottleneck kernels
  // parity taps (sylph_export_stage / sylph_export_tower): where the stage outputs res2..res5 and, with debug taps on, every
  // tower layer's stored conv output and GroupNorm coefficient table live
  const void* stage_out[4] = {nullptr, nullptr, nullptr, nullptr};
  int stage_h[4] = {0, 0, 0, 0}, stage_w[4] = {0, 0, 0, 0};
  std::vector<const void*> tap_out[2];      // [cls | bbox][layer]: conv output [rows][256] (pre-GroupNorm when tap_coef is set)
  std::vector<const float2*> tap_coef[2];   // [cls | bbox][layer]: (a, b) per (segment, channel), nullptr if applied in place
  std::vector<OpFn> backbone_ops, head_ops, support_ops;
  bool backbone_built = false, head_built = false, support_built = false;
  ImageDesc* img_desc_dev = nullptr;
  ImageDesc* img_desc_host = nullptr;
  hipEvent_t img_desc_ev = nullptr;  // recorded after the H2D copy of img_desc_host / rz_host (guards their reuse)
  // fused resize input pipeline: per-image descriptors + PIL coefficient tables (pinned host staging, device copy)
  ResizeDesc* rz_desc_dev = nullptr;
  char* rz_host = nullptr;  // [B descs][int table]
  int* rz_tab_dev = nullptr;
  size_t rz_tab_cap = 0;
  // head
  void *tA = nullptr, *tB = nullptr, *tC = nullptr, *tD = nullptr;
  void* cls_feat = nullptr;  // output of the cls tower (input of the class-conditional conv)
  int cls_ld = 256;
  // bf16, un-paired towers: the last cls-tower GroupNorm is NOT applied by the head ops; sylph_fcos_head either fuses it into the
  // class-conditional conv (N <= 32: head_fused.hip) or runs cls_apply first
  const float2* cls_coef = nullptr;
  std::function<int(hipStream_t)> cls_apply;
  float* pred = nullptr;    // [rows][8]
  float* logits = nullptr;  // [rows][logits_ld]
  int logits_ld = 0, logits_cap_ld = 0, ncls = 0;
  void* code_w = nullptr;   // packed class codes [Npad][256]
  void* code_wf = nullptr;  // the same in MFMA fragment order (logits_scan_kernel), same capacity
  int code_w_cap = 0;
  const SegDesc* head_segs = nullptr;
  const int2 *head_tiles = nullptr, *head_tiles32 = nullptr;
  int head_mtiles = 0, head_BM = 128, head_mtiles32 = 0;
  RowSeg* head_rowsegs = nullptr;
  float* gn_partial = nullptr;
  float2* gn_stats = nullptr;
  // decode
  DecodeSeg* dsegs = nullptr;
  DecodeBuffers dbuf;
  bool decode_built = false;
  std::vector<OpFn> cls_logits_ops;  // the checkpoint's cls_logits conv on this plan's cls tower output (sylph_fcos_head_pretrained)
  const float* cls_logits_dst = nullptr;  // the logits buffer those ops were built for
  int cand_cap = 0, pool_cap = 0;
  bool stem_takes_raw = false;  // this plan's first backbone op is the fused stem + pool kernel (bf16): it can read raw images
  bool raw_input = false;     // the batch came in through sylph_preprocess and its normalisation is fused into the stem kernel
                              // (launch_stem_pool_raw reads the caller's images through img_desc_dev): x0 has NOT been written
  bool scan_fused = false;    // the candidate buffers were filled by logits_scan_kernel (many-way head): decode skips its scan
  bool logits_stale = false;  // ... and the logits buffer was not written: sylph_export_head runs the unfused conv first
  float* bias_pad = nullptr;  // fp32 class biases of the last sylph_fcos_head: [0, cap) zero-padded to the packed code rows; [cap, 2 cap) the
                              // same with -inf from class N on (logits_scan_kernel: padded classes never pass the threshold)
  int bias_pad_cap = 0;
  bool has_bias = false;
  ImageOut* img_out_dev = nullptr;
  ImageOut* img_out_host = nullptr;
  hipEvent_t img_out_ev = nullptr;  // recorded after the H2D copy of img_out_host (guards its reuse without a stream sync)
  // support
  LevelDesc* lv_dev = nullptr;
  void *roi = nullptr, *cgA = nullptr, *cgB = nullptr;
  float *cg_conv_out = nullptr, *cg_bias_out = nullptr, *cg_wnorm = nullptr;  // cg_wnorm: cls_weight_norm per class of the last call
  float *re_ctx = nullptr, *re_tok = nullptr, *re_tmp = nullptr, *re_hid = nullptr, *re_cls = nullptr, *re_h = nullptr;
  const float* cur_boxes = nullptr;
  int cur_shots = 0;  // support images per class of the current sylph_codegen[_classes] call (B = classes x shots)
  float* cur_code_out = nullptr;
};

hipStream_t sylph_internal_stream(sylph_ctx* c) {
  (void)hipSetDevice(c->device);
  return c->stream;
}

int sylph_ctx::dalloc(void** p, size_t n) {
  if (n == 0) n = 16;
  hipError_t e = hipMalloc(p, n);
  if (e != hipSuccess) return fail(std::string("hipMalloc(") + std::to_string(n) + "): " + hipGetErrorString(e));
  allocs.push_back(*p);
  alloc_bytes[*p] = n;
  bytes += (int64_t)n;
  if (alloc_owner) { alloc_owner->allocs.push_back(*p); alloc_owner->bytes += (int64_t)n; }
  return 0;
}

void sylph_ctx::dfree(void* p) {
  if (!p) return;
  (void)hipStreamSynchronize(stream);
  for (auto& kv : plans) {
    auto& v = kv.second->allocs;
    for (size_t i = 0; i < v.size(); ++i)
      if (v[i] == p) {
        auto it = alloc_bytes.find(p);
        if (it != alloc_bytes.end()) kv.second->bytes -= (int64_t)it->second;
        v[i] = v.back(); v.pop_back();
        break;
      }
  }
  dfree_nosync(p);
}

// allocations made inside the scope belong to plan P (nullptr: to the context, e.g. re-packed weights)
struct OwnerScope {
  sylph_ctx* c; Plan* prev;
  OwnerScope(sylph_ctx* c_, Plan* P) : c(c_), prev(c_->alloc_owner) { c->alloc_owner = P; }
  ~OwnerScope() { c->alloc_owner = prev; }
};

static void free_plan(sylph_ctx* c, Plan* P) {
  (void)hipStreamSynchronize(c->stream);
  for (void* p : P->allocs) c->dfree_nosync(p);
  P->allocs.clear();
  if (P->img_desc_host) (void)hipHostFree(P->img_desc_host);
  if (P->img_out_host) (void)hipHostFree(P->img_out_host);
  if (P->rz_host) (void)hipHostFree(P->rz_host);
  if (P->img_out_ev) (void)hipEventDestroy(P->img_out_ev);
  if (P->img_desc_ev) (void)hipEventDestroy(P->img_desc_ev);
  if (c->cur == P) c->cur = nullptr;
}

// ------------------------------------------------------------------------------------------------
static int upload(sylph_ctx* c, void** dev, const void* host, size_t n) {
  RET(c->dalloc(dev, n));
  HIPCHK(hipMemcpy(*dev, host, n, hipMemcpyHostToDevice));
  return 0;
}

static const HostTensor* find_w(sylph_ctx* c, const std::string& k) {
  auto it = c->host_w.find(k);
  return it == c->host_w.end() ? nullptr : &it->second;
}

// pack (Cout,Cin,KH,KW) fp32 -> [Cout_pad][KH][KW][Cin] compute dtype; several tensors may be stacked on Cout
static int pack_conv(sylph_ctx* c, const std::vector<const HostTensor*>& ws, ConvLayer* L) {
  const HostTensor* w0 = ws[0];
  if (w0->shape.size() != 4) return fail("conv weight must be 4-D");
  const int Cin = (int)w0->shape[1], KH = (int)w0->shape[2], KW = (int)w0->shape[3];
  int Cout = 0;
  for (auto* w : ws) {
    if ((int)w->shape[1] != Cin || (int)w->shape[2] != KH || (int)w->shape[3] != KW) return fail("stacked conv mismatch");
    Cout += (int)w->shape[0];
  }
  const int pad_to = Cout >= 128 ? 128 : (Cout > 32 ? 64 : 32);
  const int Cout_pad = (Cout + pad_to - 1) / pad_to * pad_to;
  const size_t K = (size_t)KH * KW * Cin;
  std::vector<float> packed((size_t)Cout_pad * K, 0.f);
  int n0 = 0;
  for (auto* w : ws) {
    const int co = (int)w->shape[0];
    for (int n = 0; n < co; ++n)
      for (int ci = 0; ci < Cin; ++ci)
        for (int kh = 0; kh < KH; ++kh)
          for (int kw = 0; kw < KW; ++kw)
            packed[(size_t)(n0 + n) * K + ((size_t)kh * KW + kw) * Cin + ci] =
                w->data[(((size_t)n * Cin + ci) * KH + kh) * KW + kw];
    n0 += co;
  }
  L->Cin = Cin; L->Cout = Cout; L->Cout_pad = Cout_pad; L->KH = KH; L->KW = KW;
  if (c->dt == DT_BF16) {
    std::vector<uint16_t> h(packed.size());
    for (size_t i = 0; i < packed.size(); ++i) h[i] = f2bf_host(packed[i]);
    RET(upload(c, &L->w, h.data(), h.size() * 2));
  } else {
    RET(upload(c, &L->w, packed.data(), packed.size() * 4));
  }
  return 0;
}

static int upload_vec(sylph_ctx* c, float** dev, const std::vector<float>& v, int pad_to) {
  std::vector<float> t(v);
  t.resize((size_t)pad_to, 0.f);
  return upload(c, (void**)dev, t.data(), t.size() * 4);
}

// conv + FrozenBN folded into per-channel scale/shift (detectron2 FrozenBatchNorm2d, eps 1e-5)
static int make_conv_bn(sylph_ctx* c, const std::string& name, ConvLayer* L) {
  const HostTensor* w = find_w(c, name + ".weight");
  const HostTensor *g = find_w(c, name + ".norm.weight"), *b = find_w(c, name + ".norm.bias");
  const HostTensor *rm = find_w(c, name + ".norm.running_mean"), *rv = find_w(c, name + ".norm.running_var");
  if (!w || !g || !b || !rm || !rv) return fail("missing weights for " + name);
  RET(pack_conv(c, {w}, L));
  std::vector<float> sc(L->Cout), sh(L->Cout);
  for (int i = 0; i < L->Cout; ++i) {
    const float s = g->data[i] * (1.0f / sqrtf(rv->data[i] + 1e-5f));
    sc[i] = s;
    sh[i] = b->data[i] - rm->data[i] * s;
  }
  RET(upload_vec(c, &L->scale, sc, L->Cout_pad));
  RET(upload_vec(c, &L->shift, sh, L->Cout_pad));
  return 0;
}

static int make_conv_bias(sylph_ctx* c, const std::vector<std::string>& names, ConvLayer* L) {
  std::vector<const HostTensor*> ws;
  std::vector<float> bias;
  for (auto& n : names) {
    const HostTensor *w = find_w(c, n + ".weight"), *b = find_w(c, n + ".bias");
    if (!w || !b) return fail("missing weights for " + n);
    ws.push_back(w);
    bias.insert(bias.end(), b->data.begin(), b->data.end());
  }
  RET(pack_conv(c, ws, L));
  RET(upload_vec(c, &L->shift, bias, L->Cout_pad));
  return 0;
}

// conv3 + projection shortcut as one pointwise layer over K = [conv3 inputs | shortcut inputs]: the two FrozenBN scales
// are folded into the weights in fp32 (before the dtype cast), the shifts are summed; no epilogue scale.
static int make_c3sc(sylph_ctx* c, const HostTensor& w3, const float* s3, const float* h3, const HostTensor& ws, const float* ss,
                     const float* hs, ConvLayer* L) {
  const int co = (int)w3.shape[0], k3 = (int)w3.shape[1], ks = (int)ws.shape[1];
  HostTensor hc;
  hc.shape = {co, k3 + ks, 1, 1};
  hc.data.resize((size_t)co * (k3 + ks));
  std::vector<float> shift(co);
  for (int i = 0; i < co; ++i) {
    for (int k = 0; k < k3; ++k) hc.data[(size_t)i * (k3 + ks) + k] = w3.data[(size_t)i * k3 + k] * s3[i];
    for (int k = 0; k < ks; ++k) hc.data[(size_t)i * (k3 + ks) + k3 + k] = ws.data[(size_t)i * ks + k] * ss[i];
    shift[i] = h3[i] + hs[i];
  }
  RET(pack_conv(c, {&hc}, L));
  RET(upload_vec(c, &L->shift, shift, L->Cout_pad));
  return 0;
}

static int make_gn(sylph_ctx* c, const std::string& name, GNLayer* G) {
  const HostTensor *g = find_w(c, name + ".weight"), *b = find_w(c, name + ".bias");
  if (!g || !b) return fail("missing weights for " + name);
  if (g->data.size() != 256) return fail("GroupNorm layers must have 256 channels: " + name);
  RET(upload_vec(c, &G->gamma, g->data, 256));
  RET(upload_vec(c, &G->beta, b->data, 256));
  return 0;
}

static int upload_f32(sylph_ctx* c, const float** dev, const HostTensor* t, const std::string& what, size_t expect = 0) {
  if (!t) return fail("missing weights for " + what);
  if (expect && t->data.size() != expect) return fail("unexpected size for " + what);
  void* d;
  RET(upload(c, &d, t->data.data(), t->data.size() * 4));
  *dev = (const float*)d;
  return 0;
}

static int make_lin(sylph_ctx* c, const std::string& name, sylph_ctx::Lin* L) {
  const HostTensor *w = find_w(c, name + ".weight"), *b = find_w(c, name + ".bias");
  if (!w || !b || w->shape.size() != 2) return fail("missing weights for " + name);
  L->O = (int)w->shape[0]; L->K = (int)w->shape[1];
  RET(upload(c, (void**)&L->W, w->data.data(), w->data.size() * 4));
  RET(upload(c, (void**)&L->b, b->data.data(), b->data.size() * 4));
  return 0;
}

static int make_ln(sylph_ctx* c, const std::string& name, GNLayer* G) {
  const HostTensor *g = find_w(c, name + ".weight"), *b = find_w(c, name + ".bias");
  if (!g || !b || g->data.size() != 256) return fail("missing weights for " + name);
  RET(upload_vec(c, &G->gamma, g->data, 256));
  RET(upload_vec(c, &G->beta, b->data, 256));
  return 0;
}

static bool has_prefix(sylph_ctx* c, const std::string& p) {
  auto it = c->host_w.lower_bound(p);
  return it != c->host_w.end() && it->first.compare(0, p.size(), p) == 0;
}

// ------------------------------------------------------------------------------------------------
static void level_dims(const sylph_config& cfg, int H, int W, int* hl, int* wl, int* off, int* Ltot) {
  // stride 8/16/32 from the bottom-up path, then P6/P7 by 3x3 s2 p1 convs (ceil(x/2))
  int h = H / 8, w = W / 8;
  int o = 0;
  for (int l = 0; l < cfg.nlevels; ++l) {
    hl[l] = h; wl[l] = w; off[l] = o;
    o += h * w;
    if (l < 2) { h = (h - 1) / 2 + 1; w = (w - 1) / 2 + 1; }
    else { h = (h + 1) / 2; w = (w + 1) / 2; }
  }
  *Ltot = o;
}

struct Geom {
  const SegDesc* segs;
  const int2* tiles;
  int n_mtiles;
  std::vector<int2> seg_tiles;  // per segment: {first tile, tile count}
  float* gn_partial = nullptr;  // per-tile GroupNorm partials written by the conv epilogue (want_gn)
};

static int make_geom(sylph_ctx* c, const std::vector<SegDesc>& segs, int BM, Geom* g) {
  std::vector<int2> tiles;
  for (size_t s = 0; s < segs.size(); ++s) {
    const int rows = segs[s].out_H * segs[s].out_W;
    const int t0 = (int)tiles.size();
    for (int r = 0; r < rows; r += BM) tiles.push_back(make_int2((int)s, r));
    g->seg_tiles.push_back(make_int2(t0, (int)tiles.size() - t0));
  }
  void *ds = nullptr, *dtl = nullptr;
  RET(upload(c, &ds, segs.data(), segs.size() * sizeof(SegDesc)));
  RET(upload(c, &dtl, tiles.data(), tiles.size() * sizeof(int2)));
  g->segs = (const SegDesc*)ds;
  g->tiles = (const int2*)dtl;
  g->n_mtiles = (int)tiles.size();
  return 0;
}

// Patch shape of the halo-tile conv modes for an H x W map: ph x pw <= max_pos output positions whose (ph + 2) x (pw + 2)
// input halo fits in halo_rows LDS rows of pitch pw + xpad, chosen to minimise the number of patches (= padded positions).
// The fragment reads of the pad positions m in [ph * pw, max_pos) must stay inside the halo allocation too.
// 800 x 1344 pyramid: 100 x 168 and 50 x 84 -> 10 x 12 (no ragged edge), 25 x 42 -> 9 x 14, 13 x 21 -> 13 x 7 / 7 x 11 -> 7 x 11;
// per image 188 patches of 128 = 24 064 positions for 22 400 real ones (8 x 16 everywhere: 202 patches).
static void pick_patch(int H, int W, int max_pos, int halo_rows, int xpad, int* ph_out, int* pw_out) {
  long best_n = -1;
  int bh = 8, bw = 16, best_halo = 0;
  for (int w = 4; w <= 32; ++w) {
    for (int h = 1; h * w <= max_pos; ++h) {
      if ((h + 2) * (w + xpad) > halo_rows) continue;
      if (((max_pos - 1) / w + 2) * (w + xpad) + (max_pos - 1) % w + 2 >= halo_rows) continue;
      const long n = (long)((H + h - 1) / h) * ((W + w - 1) / w);
      const int halo = (h + 2) * (w + 2);
      const bool better = best_n < 0 || n < best_n || (n == best_n && (halo < best_halo || (halo == best_halo && w == 16)));
      if (better) { best_n = n; bh = h; bw = w; best_halo = halo; }
    }
  }
  *ph_out = bh; *pw_out = bw;
}

static void set_patch(SegDesc* s, int ph, int pw, int xpad) {
  s->ph = ph; s->pw = pw; s->hpitch = pw + xpad;
  s->inv_pw = (65536u + pw - 1) / pw;
  s->inv_hw2 = (65536u + s->hpitch - 1) / s->hpitch;
}

// 3x3 s1 p1 halo modes: M tiles are ph x pw patches of one segment, tile.y = (row << 16) | col.  `pair`: the tile list is
// padded to an even length with an empty patch (conv_halo_pipe.hip works on two patches per block).
static int make_geom_patch(sylph_ctx* c, std::vector<SegDesc> segs, int max_pos, int halo_rows, int xpad, bool pair, Geom* g) {
  std::vector<int2> tiles;
  static const int fixed = SYLPH_AB_ENV("SYLPH_CONV_PATCH_8X16", 0);  // A/B knob (-DSYLPH_ABLATE builds): the round-1 geometry
  for (size_t s = 0; s < segs.size(); ++s) {
    int ph, pw;
    if (fixed) { ph = max_pos / 16; pw = 16; }
    else pick_patch(segs[s].out_H, segs[s].out_W, max_pos, halo_rows, xpad, &ph, &pw);
    set_patch(&segs[s], ph, pw, xpad);
    const int t0 = (int)tiles.size();
    for (int y = 0; y < segs[s].out_H; y += ph)
      for (int x = 0; x < segs[s].out_W; x += pw) tiles.push_back(make_int2((int)s, (y << 16) | x));
    g->seg_tiles.push_back(make_int2(t0, (int)tiles.size() - t0));
  }
  g->n_mtiles = (int)tiles.size();
  if (pair && (tiles.size() & 1)) tiles.push_back(make_int2(0, 0x7fff << 16));  // origin below every map: nothing loaded, nothing stored
  void *ds = nullptr, *dtl = nullptr;
  RET(upload(c, &ds, segs.data(), segs.size() * sizeof(SegDesc)));
  RET(upload(c, &dtl, tiles.data(), tiles.size() * sizeof(int2)));
  g->segs = (const SegDesc*)ds;
  g->tiles = (const int2*)dtl;
  return 0;
}

static long patch_count(const std::vector<SegDesc>& segs, int max_pos, int halo_rows, int xpad) {
  long n = 0;
  for (auto& sg : segs) {
    int ph, pw;
    pick_patch(sg.out_H, sg.out_W, max_pos, halo_rows, xpad, &ph, &pw);
    n += (long)((sg.out_H + ph - 1) / ph) * ((sg.out_W + pw - 1) / pw);
  }
  return n;
}

// launch the conv kernel, optionally bracketed by HIP events on the same stream
static int timed_conv(sylph_ctx* c, DType dt, bool of32, const ConvArgs& a, int BM, int BN, double flops,
                      hipStream_t s) {
  if (!c->prof) return launch_conv(dt, of32, a, BM, BN, s);
  sylph_ctx::ProfRec r;
  if (!c->prof_free.empty()) {
    r.a = c->prof_free.back().first; r.b = c->prof_free.back().second;
    c->prof_free.pop_back();
  } else {
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -100;
  }
  r.flops = flops;
  r.kern = (BM == 256 && BN == 256) ? (a.gn_coef ? "conv_hpipe_kernel<true>" : "conv_hpipe_kernel<false>") : "conv_igemm_kernel";  // the names rocprofv3 prints
  (void)hipEventRecord(r.a, s);
  const int rc = launch_conv(dt, of32, a, BM, BN, s);
  (void)hipEventRecord(r.b, s);
  c->prof_recs.push_back(r);
  return rc;
}

// any other launch that should count as conv work in the profile (dedicated stem kernel)
static int timed_op(sylph_ctx* c, const char* kern, double flops, hipStream_t s, const std::function<int(hipStream_t)>& fn) {
  if (!c->prof) return fn(s);
  sylph_ctx::ProfRec r;
  if (!c->prof_free.empty()) {
    r.a = c->prof_free.back().first; r.b = c->prof_free.back().second;
    c->prof_free.pop_back();
  } else {
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -100;
  }
  r.flops = flops;
  r.kern = kern;
  (void)hipEventRecord(r.a, s);
  const int rc = fn(s);
  (void)hipEventRecord(r.b, s);
  c->prof_recs.push_back(r);
  return rc;
}

struct ConvOpts {
  int stride = 1, pad = 0;
  int relu_nch = 0, mul_nch = 0;
  const void* res = nullptr;
  int res_ld = 0, res_mode = 0;
  bool out_f32 = false;
  int cout_override = -1;  // logical Cout (class-conditional conv)
  int group_cout = 0, group_in_off = 0;  // grouped conv (paired FCOS towers)
  int stem = 0;            // ResNet stem loader
  int want_gn = 0;         // leave per-tile GroupNorm partials in the epilogue
  const void* in2 = nullptr;  // dual-source pointwise conv: second input, its row stride / channels / stride
  int in2_ld = 0, Cin2 = 0, stride2 = 1;
  double flops = -1.0;     // algorithmic FLOPs of the launch when they differ from 2*M*N*K (stem padding)
  const float2* gn_coef = nullptr;  // fused GroupNorm(+ReLU) of the INPUT (conv_hpipe.hip): (a, b) per (segment, input channel)
  int gn_relu = 0;
};

// Will add_conv route this 3x3 layer to the deep-pipelined halo kernel (conv_hpipe.hip)?  MFMA-bound 3x3 stride-1 layers
// with Cout % 256 == 0 (FCOS towers, FPN outputs, res4/res5 conv2) when its rounds fill the chip: at least two rounds over
// the 256 CUs and >= 80 % of the last one used.
static long patch_count(const std::vector<SegDesc>& segs, int max_pos, int halo_rows, int xpad);
static bool use_hpipe(sylph_ctx* c, const ConvLayer& L, const std::vector<SegDesc>& segs, const ConvOpts& o) {
  static const int hp_on = getenv("SYLPH_CONV_HPIPE") ? atoi(getenv("SYLPH_CONV_HPIPE")) : 1;
  const int cout_l = o.cout_override >= 0 ? o.cout_override : L.Cout;
  const bool k3s1 = L.KH == 3 && L.KW == 3 && o.stride == 1 && o.pad == 1 && !o.stem && !o.in2;
  if (!(hp_on && k3s1 && c->dt == DT_BF16 && !o.out_f32 && !o.res && o.res_mode == 0 && o.mul_nch == 0 && o.cout_override < 0 &&
        (o.relu_nch == 0 || o.relu_nch >= cout_l) && L.Cin % 32 == 0 && L.Cout % 256 == 0 && L.Cout == L.Cout_pad))
    return false;
  const long blocks = (patch_count(segs, 128, 256, 4) + 1) / 2 * (L.Cout / 256), rounds = (blocks + 255) / 256;
  return hp_on == 2 || (blocks >= 512 && blocks * 10 >= rounds * 256 * 8);
}

static int add_conv(sylph_ctx* c, std::vector<OpFn>& ops, const ConvLayer& L, const void* in, int in_ld, void* out,
                    int out_ld, const std::vector<SegDesc>& segs, const ConvOpts& o, Geom* geom_out = nullptr) {
  long rows = 0;
  for (auto& s : segs) rows += (long)s.out_H * s.out_W;
  int BM, BN;
  conv_pick_tile((int)rows, L.Cout_pad, o.stem ? 49 : L.KH * L.KW, &BM, &BN);
  if (L.Cout_pad % BN != 0) return fail("Cout_pad not a multiple of BN");
  const bool k3s1 = L.KH == 3 && L.KW == 3 && o.stride == 1 && o.pad == 1 && !o.stem && !o.in2;
  const bool hpipe = use_hpipe(c, L, segs, o);
  if (o.gn_coef && !hpipe) return fail("a fused input GroupNorm needs the conv_hpipe kernel (internal)");
  if (hpipe) { BM = 256; BN = 256; }
  // other 3x3 stride-1 convs on 128-row tiles: halo mode (input patch staged once per channel slice, 9 taps read it)
  static const int halo_on = getenv("SYLPH_CONV_HALO") ? atoi(getenv("SYLPH_CONV_HALO")) : 1;
  bool halo = !hpipe && halo_on && c->dt == DT_BF16 && BM == 128 && (BN == 32 || (!o.out_f32 && (BN == 128 || BN == 64))) && k3s1 &&
              L.Cin % 64 == 0;
  if (halo) {  // patches must not waste much of the launch on ragged edges (tiny maps are cheap anyway)
    const long patch_rows = patch_count(segs, 128, 184, 2) * 128;
    if (halo_on != 2 && patch_rows * 10 > rows * 15) halo = false;
  }
  // pointwise bf16 layers: persistent pipelined kernel (conv_pw.hip) when the launch has at least one tile per block slot
  static const int pw_on = getenv("SYLPH_CONV_PW") ? atoi(getenv("SYLPH_CONV_PW")) : 1;
  bool pw = false;
  int pw_bm = 0, pw_bn = 0;
  if (pw_on && !hpipe && !halo && c->dt == DT_BF16 && !o.out_f32 && L.KH == 1 && L.KW == 1 && o.pad == 0 && !o.stem && o.group_cout == 0 &&
      o.mul_nch == 0 && !o.want_gn && !o.gn_coef && o.cout_override < 0 && L.Cout == L.Cout_pad &&
      (o.relu_nch == 0 || o.relu_nch >= L.Cout) && L.Cin % 32 == 0 && L.Cin >= 128 && (!o.in2 || o.Cin2 % 32 == 0) &&
      conv_pw_tile(L.Cout, L.Cin, o.res_mode != 0, &pw_bm, &pw_bn)) {
    const int bn = pw_bn, bm = pw_bm;
    const long tiles = ((rows + bm - 1) / bm) * (L.Cout / bn);
    // 32-bit byte offsets into the activation buffers
    long in_rows = 0, in2_rows = 0, res_rows = 0;
    for (auto& sg : segs) {
      in_rows = std::max(in_rows, (long)sg.in_row0 + (long)sg.in_H * sg.in_W);
      in2_rows = std::max(in2_rows, (long)sg.in2_row0 + (long)sg.out_H * o.stride2 * sg.in2_W);
      res_rows = std::max(res_rows, (long)sg.res_row0 + (long)sg.res_H * sg.res_W);
    }
    const bool fits = in_rows * in_ld * 2 < (1L << 32) && (!o.in2 || in2_rows * o.in2_ld * 2 < (1L << 32)) &&
                      (!o.res || res_rows * o.res_ld * 2 < (1L << 32));
    // Where it pays (in-situ timeline at B = 64, profiles/r3_*): every pointwise layer without a same-geometry residual -- bottleneck
    // conv1 (-13 ... -20 %), conv3 + projection as one GEMM (-14 ... -20 %), FPN laterals incl. the top-down add (-10 ... -15 %) --
    // except the res3-shaped identity conv1 (N = 128, stride 1: already at 5 TB/s in conv_igemm).  With a residual tile to fetch the
    // two kernels are equal (res4 / res5) or conv_igemm's five co-resident blocks win (res3, K = 128): those stay there.
    const bool pays = o.res_mode != 1 && (L.Cout % 256 == 0 || o.stride != 1);
    pw = fits && (pw_on == 2 || (tiles >= 256 && pays));
    if (pw) { BM = bm; BN = bn; }
  }
  Geom g;
  if (hpipe) RET(make_geom_patch(c, segs, 128, 256, 4, true, &g));
  else if (halo) RET(make_geom_patch(c, segs, 128, 184, 2, false, &g));
  else RET(make_geom(c, segs, BM, &g));
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.halo = hpipe ? 2 : (halo ? 1 : 0);
  a.in = in; a.wt = L.w; a.out = out; a.res = o.res;
  if (hpipe) {  // stage-image weight layout of conv_hpipe.hip, packed once per layer
    auto it = c->hp_weights.find(L.w);
    if (it == c->hp_weights.end()) {
      void* wp = nullptr;
      OwnerScope ctx_owned(c, nullptr);  // a layer's re-packed weights outlive the plan that first needed them
      RET(c->dalloc(&wp, (size_t)L.Cout * 9 * L.Cin * 2));
      KCHK(launch_hpipe_pack_weights(L.w, wp, L.Cout, L.Cin, c->stream), "hpipe_pack_weights");
      HIPCHK(hipStreamSynchronize(c->stream));
      it = c->hp_weights.emplace(L.w, wp).first;
    }
    a.wt = it->second;
  }
  if (pw) {  // stage-image weight layout + scale/shift table of conv_pw.hip, packed once per layer; one descriptor per M tile
    auto it = c->pw_weights.find(L.w);
    if (it == c->pw_weights.end()) {
      void* wp = nullptr;
      float* tb = nullptr;
      OwnerScope ctx_owned(c, nullptr);
      RET(c->dalloc(&wp, (size_t)L.Cout * L.Cin * 2));
      RET(c->dalloc((void**)&tb, (size_t)2 * L.Cout * sizeof(float)));
      KCHK(launch_pw_pack_weights(L.w, wp, L.Cout, L.Cin, BN, c->stream), "pw_pack_weights");
      KCHK(launch_pw_pack_table(L.scale, L.shift, tb, L.Cout, BN, c->stream), "pw_pack_table");
      HIPCHK(hipStreamSynchronize(c->stream));
      it = c->pw_weights.emplace(L.w, std::make_pair(wp, tb)).first;
    }
    a.wt = it->second.first;
    a.pw_table = it->second.second;
    if (!c->pw_trash) {
      OwnerScope ctx_owned(c, nullptr);
      RET(c->dalloc(&c->pw_trash, 8192));
    }
    a.trash = c->pw_trash;
    std::vector<PwDesc> pd;
    for (size_t sgi = 0; sgi < segs.size(); ++sgi) {
      const SegDesc& sg = segs[sgi];
      const int nrows = sg.out_H * sg.out_W;
      for (int r = 0; r < nrows; r += BM) {
        PwDesc d;
        memset(&d, 0, sizeof(d));
        d.row0 = r; d.seg_rows = nrows; d.out_W = sg.out_W; d.out_row0 = sg.out_row0; d.in_row0 = sg.in_row0; d.in_W = sg.in_W;
        d.in2_row0 = sg.in2_row0; d.in2_W = o.in2 ? sg.in2_W : sg.out_W; d.res_row0 = sg.res_row0; d.res_W = sg.res_W;
        pd.push_back(d);
      }
    }
    if ((int)pd.size() != g.n_mtiles) return fail("internal: conv_pw descriptor count");
    void* pdd = nullptr;
    RET(upload(c, &pdd, pd.data(), pd.size() * sizeof(PwDesc)));
    a.pw_desc = (const PwDesc*)pdd;
  }
  a.scale = L.scale; a.shift = L.shift; a.zeros = c->zeros;
  a.segs = g.segs; a.tiles = g.tiles; a.n_mtiles = hpipe ? (g.n_mtiles + 1) / 2 : g.n_mtiles; a.n_ntiles = L.Cout_pad / BN;
  a.Cin = L.Cin; a.Cout = o.cout_override >= 0 ? o.cout_override : L.Cout;
  a.KH = L.KH; a.KW = L.KW; a.stride = o.stride; a.pad = o.pad;
  a.in_ld = in_ld; a.out_ld = out_ld; a.res_ld = o.res_ld;
  a.relu_nch = o.relu_nch; a.mul_nch = o.mul_nch; a.res_mode = o.res_mode;
  a.stem = o.stem; a.tap_dy = o.stem ? L.Cin / 32 : 1;
  a.group_cout = o.group_cout; a.group_in_off = o.group_in_off;
  a.gn_coef = o.gn_coef; a.gn_relu = o.gn_relu;
  a.ss_padded = 1;  // ConvLayer scale/shift are zero-padded to Cout_pad
  if (o.in2) {
    a.in2 = o.in2; a.in2_ld = o.in2_ld; a.Cin2 = o.Cin2; a.stride2 = o.stride2;
    a.Cin = L.Cin - o.Cin2;  // the packed weights hold both K ranges back to back
  }
  if (o.want_gn) {
    if (L.Cout != 256 && L.Cout != 512) return fail("fused GroupNorm statistics need Cout == 256 or 512");
    RET(c->dalloc((void**)&a.gn_partial, (size_t)(g.n_mtiles + 1) * (L.Cout / 8) * 3 * sizeof(float)));  // +1: the pad patch of an odd pair list
  }
  if (geom_out) { *geom_out = g; geom_out->gn_partial = a.gn_partial; }
  const DType dt = c->dt;
  const bool of32 = o.out_f32;
  const double flops = o.flops >= 0.0 ? o.flops : 2.0 * (double)rows * (double)a.Cout * (double)(L.KH * L.KW) * (double)L.Cin;
  if (pw) {
    if (!conv_pw_ok(dt, of32, a)) return fail("internal: conv_pw selected for a layer it cannot run");
    ops.push_back([a, BM, BN, c, flops](hipStream_t s) { return timed_op(c, "conv_pw_kernel", flops, s, [=](hipStream_t st) { return launch_conv_pw(a, BM, BN, st); }); });
    return 0;
  }
  ops.push_back([a, BM, BN, dt, of32, c, flops](hipStream_t s) { return timed_conv(c, dt, of32, a, BM, BN, flops, s); });
  return 0;
}

// conv + GroupNorm(32, 256)(+ReLU): statistics fused into the conv epilogue, one in-place apply pass
// coef_out != nullptr: no apply pass; the (a, b) table of this layer's GroupNorm is left for the NEXT conv, which applies
// it (+ ReLU) to its input halo in LDS (ConvOpts::gn_coef).
static int add_conv_gn(sylph_ctx* c, std::vector<OpFn>& ops, const ConvLayer& L, const void* in, int in_ld, void* out,
                       const std::vector<SegDesc>& segs, ConvOpts o, const GNLayer& G, int relu, const float2** coef_out = nullptr,
                       OpFn* apply_out = nullptr) {
  o.want_gn = 1;
  Geom g;
  const int ld = L.Cout, ngroups = L.Cout / 8;
  RET(add_conv(c, ops, L, in, in_ld, out, ld, segs, o, &g));
  const float* partial = g.gn_partial;
  std::vector<GnSeg> gs;
  int max_rows = 0;
  for (size_t s = 0; s < segs.size(); ++s) {
    const int rows = segs[s].out_H * segs[s].out_W;
    gs.push_back(GnSeg{segs[s].out_row0, rows, g.seg_tiles[s].x, g.seg_tiles[s].y});
    max_rows = rows > max_rows ? rows : max_rows;
  }
  GnSeg* gsd;
  RET(upload(c, (void**)&gsd, gs.data(), gs.size() * sizeof(GnSeg)));
  const DType dt = c->dt;
  const int nseg = (int)gs.size();
  const float *ga = G.gamma, *be = G.beta;
  float2* stats_ws = nullptr;
  RET(c->dalloc((void**)&stats_ws, (size_t)nseg * ngroups * sizeof(float2)));
  if (coef_out) {
    float2* coef = nullptr;
    RET(c->dalloc((void**)&coef, (size_t)nseg * ld * sizeof(float2)));
    ops.push_back([=](hipStream_t s) { return launch_gn_finalize_coef(ngroups, gsd, nseg, partial, stats_ws, ga, be, 1e-5f, coef, s); });
    *coef_out = coef;
    if (apply_out)  // the stand-alone apply of the same layer, for a consumer that cannot take the coefficients
      *apply_out = [=](hipStream_t s) {
        return launch_gn_apply_partials(dt, out, ld, ngroups, gsd, nseg, max_rows, partial, stats_ws, ga, be, 1e-5f, relu, s);
      };
    return 0;
  }
  ops.push_back([=](hipStream_t s) {
    return launch_gn_apply_partials(dt, out, ld, ngroups, gsd, nseg, max_rows, partial, stats_ws, ga, be, 1e-5f, relu, s);
  });
  return 0;
}

static std::vector<SegDesc> image_segs(int B, int Hin, int Win, int Hout, int Wout, int resH = 0, int resW = 0) {
  std::vector<SegDesc> v((size_t)B);
  for (int b = 0; b < B; ++b) {
    SegDesc s;
    memset(&s, 0, sizeof(s));
    s.in_row0 = b * Hin * Win; s.in_H = Hin; s.in_W = Win;
    s.out_row0 = b * Hout * Wout; s.out_H = Hout; s.out_W = Wout;
    s.res_H = resH ? resH : Hout; s.res_W = resW ? resW : Wout;
    s.res_row0 = b * s.res_H * s.res_W;
    s.mul = 1.f;
    v[b] = s;
  }
  return v;
}

static void evict_plans(sylph_ctx* c, const Plan* keep) {
  auto total = [&]() { int64_t t = 0; for (auto& kv : c->plans) t += kv.second->bytes; return t; };
  while (c->plans.size() > 1 && (c->plans.size() >= c->max_plans || (c->plan_byte_budget > 0 && total() > c->plan_byte_budget))) {
    auto victim = c->plans.end();
    for (auto it = c->plans.begin(); it != c->plans.end(); ++it)
      if (it->second.get() != keep && it->second.get() != c->cur &&
          (victim == c->plans.end() || it->second->last_use < victim->second->last_use)) victim = it;
    if (victim == c->plans.end()) break;
    free_plan(c, victim->second.get());
    c->plans.erase(victim);
  }
}

static void drop_plan(sylph_ctx* c, Plan* P) {  // a plan whose build failed half way: release it so that a retry starts clean
  for (auto it = c->plans.begin(); it != c->plans.end(); ++it)
    if (it->second.get() == P) { free_plan(c, P); c->plans.erase(it); return; }
}

static Plan* get_plan(sylph_ctx* c, int B, int H, int W) {
  auto key = std::make_tuple(B, H, W);
  auto it = c->plans.find(key);
  if (it != c->plans.end()) { it->second->last_use = ++c->use_clock; return it->second.get(); }
  evict_plans(c, nullptr);
  std::unique_ptr<Plan> p(new Plan());
  p->last_use = ++c->use_clock;
  p->B = B; p->H = H; p->W = W;
  level_dims(c->cfg, H, W, p->hl, p->wl, p->off, &p->Ltot);
  p->img_h.assign(B, H);
  p->img_w.assign(B, W);
  memset(&p->dbuf, 0, sizeof(p->dbuf));
  Plan* raw = p.get();
  c->plans[key] = std::move(p);
  return raw;
}

static int ensure_pyramid(sylph_ctx* c, Plan* P) {
  if (!P->F) RET(c->dalloc(&P->F, (size_t)P->B * P->Ltot * 256 * c->esz()));
  return 0;
}

// One ResNet bottleneck block (detectron2 BottleneckBlock: 1x1 -> 3x3 -> 1x1, FrozenBN folded, residual / projection shortcut)
// appended to `ops`: X [B][Hin*Win][Cin] -> Y [B][Ho*Wo][cout].  t1 / t2 / sc are scratch activations of the stage.
// Shared by build_backbone and the single-block parity entry sylph_bottleneck, so both run the same kernels.
struct BkScratch { void *t1, *t2, *sc; void** trash; };
static int add_bottleneck(sylph_ctx* c, std::vector<OpFn>& ops, const sylph_ctx::Block& blk, int B, const void* X, int Cin, int Hin, int Win,
                          int stride, int mid, int cout, void* Y, const BkScratch& scr) {
  const DType dt = c->dt;
  const int s1 = c->cfg.stride_in_1x1 ? stride : 1, s3 = c->cfg.stride_in_1x1 ? 1 : stride;
  const int H1 = (Hin - 1) / s1 + 1, W1 = (Win - 1) / s1 + 1;
  const int Ho = (Hin - 1) / stride + 1, Wo = (Win - 1) / stride + 1;
  void *t1 = scr.t1, *t2 = scr.t2, *sc = scr.sc;
  // res2 identity blocks (C 256, mid 64, stride 1, no projection), bf16: ONE fused kernel (bottleneck.hip): the two
  // 64-channel intermediates and the second read of x never reach HBM (2 048 -> 1 024 B per position)
  static const int fuse_bn = getenv("SYLPH_FUSE_BOTTLENECK") ? atoi(getenv("SYLPH_FUSE_BOTTLENECK")) : 1;
  const bool fuse_id = fuse_bn && dt == DT_BF16 && !blk.has_sc && stride == 1 && mid == 64 && Cin == 256 && cout == 256 &&
                       blk.c1.Cout_pad == 64 && blk.c2.Cout_pad == 64 && blk.c3.Cout_pad == 256;
  // first block of res2 (64 -> 64 -> 64 -> 256, projection folded into conv3's GEMM, stride 1): one fused kernel too
  const bool fuse_pr = fuse_bn && dt == DT_BF16 && blk.fused_sc && stride == 1 && mid == 64 && Cin == 64 && cout == 256 &&
                       blk.c1.Cout_pad == 64 && blk.c2.Cout_pad == 64 && blk.c3sc.Cout_pad == 256 && blk.c3sc.Cin == 128 && !blk.c3sc.scale;
  if ((fuse_id || fuse_pr) && (size_t)B * Hin * Win * 512 < ((size_t)1 << 32)) {  // the kernels address x with 32-bit byte offsets
    BottleneckArgs ba;
    memset(&ba, 0, sizeof(ba));
    ba.x = X; ba.y = Y;
    ba.w1 = (const __bf16*)blk.c1.w; ba.w2 = (const __bf16*)blk.c2.w; ba.w3 = (const __bf16*)(fuse_id ? blk.c3.w : blk.c3sc.w);
    ba.s1 = blk.c1.scale; ba.b1 = blk.c1.shift; ba.s2 = blk.c2.scale; ba.b2 = blk.c2.shift;
    ba.s3 = fuse_id ? blk.c3.scale : nullptr; ba.b3 = fuse_id ? blk.c3.shift : blk.c3sc.shift;
    ba.zeros = c->zeros;
    if (!*scr.trash) RET(c->dalloc(scr.trash, (size_t)1024 * 256 * 128));  // per-thread trash slots (grid <= CU count <= 1024)
    ba.trash = *scr.trash;
    std::vector<SegDesc> sg = image_segs(B, Hin, Win, Hin, Win);
    std::vector<BkTile> bt;
    int ph, pw;
    // SYLPH_BK_SMALL=1 (A/B knob): identity block on 64-position patches (8 x 8 on the 200 x 336 map) whose 128-row halo is
    // double-buffered and prefetched a whole tile ahead (bottleneck.hip <2, 1, 2, true>).  Measured 1.56 ms vs 1.31 ms per launch at
    // B = 64: the halo round trip is hidden, but 2.2 x as many tiles pay the per-tile fixed costs (five barriers, three pipeline
    // fills / MFMA drains, descriptor and address set-up: ~2.3 us per tile) -- the <= 128-position geometry stays the default.
    static const int bk_small_on = SYLPH_AB_ENV("SYLPH_BK_SMALL", 0);
    const int bk_small = fuse_id && bk_small_on;
    if (bk_small) pick_patch(Hin, Win, 64, 128, 2, &ph, &pw);
    else pick_patch(Hin, Win, 128, 184, 2, &ph, &pw);
    for (size_t si2 = 0; si2 < sg.size(); ++si2)
      for (int yy = 0; yy < Hin; yy += ph)
        for (int xx = 0; xx < Win; xx += pw)
          bt.push_back(BkTile{sg[si2].in_row0, Hin, Win, (yy << 16) | xx, ph, pw, (65536u + pw - 1) / pw, (65536u + pw + 2 - 1) / (pw + 2)});
    void* btd = nullptr;
    RET(upload(c, &btd, bt.data(), bt.size() * sizeof(BkTile)));
    ba.bk = (const BkTile*)btd;
    ba.n_tiles = (int)bt.size();
    const double fl = 2.0 * (double)B * Hin * Win * (fuse_id ? (256.0 * 64 + 64.0 * 576 + 64.0 * 256) : (64.0 * 64 + 64.0 * 576 + 128.0 * 256));
    if (fuse_id) ops.push_back([=](hipStream_t s) { return timed_op(c, "bottleneck64_kernel", fl, s, [=](hipStream_t st) { return launch_bottleneck64(ba, bk_small, st); }); });
    else ops.push_back([=](hipStream_t s) { return timed_op(c, "bottleneck64p_kernel", fl, s, [=](hipStream_t st) { return launch_bottleneck64p(ba, st); }); });
    return 0;
  }
  ConvOpts o1; o1.stride = s1; o1.relu_nch = 1 << 30;
  RET(add_conv(c, ops, blk.c1, X, Cin, t1, mid, image_segs(B, Hin, Win, H1, W1), o1));
  ConvOpts o2; o2.stride = s3; o2.pad = 1; o2.relu_nch = 1 << 30;
  RET(add_conv(c, ops, blk.c2, t1, mid, t2, mid, image_segs(B, H1, W1, Ho, Wo), o2));
  if (blk.fused_sc) {
    // conv3 + projection shortcut as ONE pointwise GEMM over K = [t2 | X(strided)]: the shortcut
    // tensor is never written to / re-read from HBM
    std::vector<SegDesc> sg = image_segs(B, Ho, Wo, Ho, Wo);
    for (int b = 0; b < B; ++b) { sg[b].in2_row0 = b * Hin * Win; sg[b].in2_W = Win; }
    ConvOpts o3; o3.relu_nch = 1 << 30; o3.in2 = X; o3.in2_ld = Cin; o3.Cin2 = Cin; o3.stride2 = stride;
    RET(add_conv(c, ops, blk.c3sc, t2, mid, Y, cout, sg, o3));
  } else {
    const void* resid = X;
    if (blk.has_sc) {
      ConvOpts os; os.stride = stride;
      RET(add_conv(c, ops, blk.sc, X, Cin, sc, cout, image_segs(B, Hin, Win, Ho, Wo), os));
      resid = sc;
    }
    ConvOpts o3; o3.relu_nch = 1 << 30; o3.res = resid; o3.res_ld = cout; o3.res_mode = 1;
    RET(add_conv(c, ops, blk.c3, t2, mid, Y, cout, image_segs(B, Ho, Wo, Ho, Wo), o3));
  }
  return 0;
}

static int build_backbone(sylph_ctx* c, Plan* P) {
  if (P->backbone_built) return 0;
  if (!c->has_backbone) return fail("backbone weights were not loaded");
  const int B = P->B, H = P->H, W = P->W;
  const size_t e = c->esz();
  RET(ensure_pyramid(c, P));
  RET(c->dalloc(&P->x0, (size_t)B * H * W * 4 * e));
  const int H2 = (H - 1) / 2 + 1, W2 = (W - 1) / 2 + 1;
  const int H4 = (H2 - 1) / 2 + 1, W4 = (W2 - 1) / 2 + 1;
  RET(c->dalloc(&P->stem_out, (size_t)B * H2 * W2 * 64 * e));
  RET(c->dalloc(&P->pool_out, (size_t)B * H4 * W4 * 64 * e));
  RET(c->dalloc((void**)&P->img_desc_dev, sizeof(ImageDesc) * B));
  HIPCHK(hipHostMalloc((void**)&P->img_desc_host, sizeof(ImageDesc) * B));
  auto& ops = P->backbone_ops;
  const DType dt = c->dt;
  {
    void *so = P->stem_out, *po = P->pool_out;
    ConvOpts os; os.stem = 1; os.relu_nch = 1 << 30;
    os.flops = 2.0 * (double)B * H2 * W2 * 64.0 * 147.0;
    static const int stem_fast = getenv("SYLPH_STEM_KERNEL") ? atoi(getenv("SYLPH_STEM_KERNEL")) : 1;
    if (dt == DT_BF16 && c->stem_wp && stem_fast) {
      const void *x0 = P->x0, *wp = c->stem_wp;
      const float *scl = c->stem.scale, *shf = c->stem.shift;
      const double fl = os.flops;
      // stem + max-pool in one kernel: the 64-channel stem output never reaches HBM (stem_conv.hip)
      static const int fuse_pool = getenv("SYLPH_FUSE_STEM_POOL") ? atoi(getenv("SYLPH_FUSE_STEM_POOL")) : 1;
      if (fuse_pool) {
        void* trash = nullptr;
        RET(c->dalloc(&trash, (size_t)512 * 256 * 16));
        const Plan* PP = P;
        ops.push_back([=](hipStream_t s) {
          return timed_op(c, "stem_pool_kernel", fl, s, [=](hipStream_t st) {
            if (PP->raw_input)  // (p - mean) / std applied on the way into the stem's LDS patch: no normalised copy of the batch
              return launch_stem_pool_raw(PP->img_desc_dev, c->cfg.pixel_mean, c->cfg.pixel_std, wp, scl, shf, po, trash, B, H, W, H2, W2, H4, W4, st);
            return launch_stem_pool(x0, wp, scl, shf, po, trash, B, H, W, H2, W2, H4, W4, st);
          });
        });
        P->stem_takes_raw = B <= STEM_RAW_MAX_BATCH;
      } else {
        ops.push_back([=](hipStream_t s) {
          return timed_op(c, "stem_conv_kernel", fl, s, [=](hipStream_t st) { return launch_stem_conv(x0, wp, scl, shf, so, B, H, W, H2, W2, st); });
        });
        ops.push_back([=](hipStream_t s) { return launch_maxpool(dt, so, po, B, H2, W2, 64, H4, W4, s); });
      }
    } else {
      RET(add_conv(c, ops, c->stem, P->x0, 4, so, 64, image_segs(B, H, W, H2, W2), os));
      ops.push_back([=](hipStream_t s) { return launch_maxpool(dt, so, po, B, H2, W2, 64, H4, W4, s); });
    }
  }
  const void* X = P->pool_out;
  int Hin = H4, Win = W4, Cin = 64;
  const void* stage_out[4] = {nullptr, nullptr, nullptr, nullptr};
  int stage_h[4], stage_w[4];
  for (int si = 0; si < 4; ++si) {
    const int mid = 64 << si, cout = 256 << si;
    const int first_stride = si == 0 ? 1 : 2;
    const int Hs = (Hin - 1) / first_stride + 1, Ws = (Win - 1) / first_stride + 1;
    void *t1, *t2, *sc, *Ya, *Yb;
    // t1 may still be at the input resolution when the stride sits on the 3x3
    RET(c->dalloc(&t1, (size_t)B * Hin * Win * mid * e));
    RET(c->dalloc(&t2, (size_t)B * Hs * Ws * mid * e));
    RET(c->dalloc(&sc, (size_t)B * Hs * Ws * cout * e));
    RET(c->dalloc(&Ya, (size_t)B * Hs * Ws * cout * e));
    RET(c->dalloc(&Yb, (size_t)B * Hs * Ws * cout * e));
    auto& blocks = c->stages[si];
    void* Y = nullptr;
    BkScratch scr{t1, t2, sc, &P->bk_trash};
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
      const int stride = bi == 0 ? first_stride : 1;
      Y = (Y == Ya) ? Yb : Ya;
      RET(add_bottleneck(c, ops, blocks[bi], B, X, Cin, Hin, Win, stride, mid, cout, Y, scr));
      X = Y; Hin = (Hin - 1) / stride + 1; Win = (Win - 1) / stride + 1; Cin = cout;
    }
    stage_out[si] = X; stage_h[si] = Hin; stage_w[si] = Win;
    P->stage_out[si] = X; P->stage_h[si] = Hin; P->stage_w[si] = Win;
  }
  // FPN (res3..res5 -> p3..p5), top-down with nearest 2x upsample fused as a residual, then P6/P7
  void* lat[3] = {nullptr, nullptr, nullptr};
  for (int k = 2; k >= 0; --k) {
    const int si = k + 1, h = stage_h[si], w = stage_w[si], cin = 256 << si;
    if (h != P->hl[k] || w != P->wl[k]) return fail("internal: level geometry mismatch");
    RET(c->dalloc(&lat[k], (size_t)B * h * w * 256 * e));
    ConvOpts ol;
    std::vector<SegDesc> segs = image_segs(B, h, w, h, w);
    if (k < 2) {
      ol.res = lat[k + 1]; ol.res_ld = 256; ol.res_mode = 2;
      segs = image_segs(B, h, w, h, w, stage_h[si + 1], stage_w[si + 1]);
      if (h != 2 * stage_h[si + 1] || w != 2 * stage_w[si + 1]) return fail("FPN needs exact 2x level sizes");
    }
    RET(add_conv(c, ops, c->fpn_lat[k], stage_out[si], cin, lat[k], 256, segs, ol));
    std::vector<SegDesc> so = image_segs(B, h, w, h, w);
    for (int b = 0; b < B; ++b) so[b].out_row0 = b * P->Ltot + P->off[k];
    ConvOpts oo; oo.pad = 1;
    RET(add_conv(c, ops, c->fpn_out[k], lat[k], 256, P->F, 256, so, oo));
  }
  for (int k = 3; k < c->cfg.nlevels && k < 5; ++k) {
    std::vector<SegDesc> sg = image_segs(B, P->hl[k - 1], P->wl[k - 1], P->hl[k], P->wl[k]);
    for (int b = 0; b < B; ++b) {
      sg[b].in_row0 = b * P->Ltot + P->off[k - 1];
      sg[b].out_row0 = b * P->Ltot + P->off[k];
    }
    ConvOpts op; op.stride = 2; op.pad = 1;
    const void* src = P->F;
    if (k == 4) {  // P7 = conv(relu(P6)): rectified copy of the P6 rows
      const int n6 = P->hl[3] * P->wl[3];
      void* p6r;
      RET(c->dalloc(&p6r, (size_t)B * n6 * 256 * e));
      std::vector<CopySeg> cs;
      for (int b = 0; b < B; ++b) {
        cs.push_back(CopySeg{b * P->Ltot + P->off[3], b * n6, n6});
        sg[b].in_row0 = b * n6;
      }
      CopySeg* csd;
      RET(upload(c, (void**)&csd, cs.data(), cs.size() * sizeof(CopySeg)));
      const void* F = P->F;
      ops.push_back([=](hipStream_t s) { return launch_relu_rows(dt, F, p6r, 256, csd, B, n6, s); });
      src = p6r;
    }
    RET(add_conv(c, ops, k == 3 ? c->p6 : c->p7, src, 256, P->F, 256, sg, op));
  }
  P->backbone_built = true;
  return 0;
}

static std::vector<SegDesc> pyramid_segs(sylph_ctx* c, Plan* P) {
  std::vector<SegDesc> v;
  for (int b = 0; b < P->B; ++b)
    for (int l = 0; l < c->cfg.nlevels; ++l) {
      SegDesc s;
      memset(&s, 0, sizeof(s));
      s.in_row0 = s.out_row0 = s.res_row0 = b * P->Ltot + P->off[l];
      s.in_H = s.out_H = s.res_H = P->hl[l];
      s.in_W = s.out_W = s.res_W = P->wl[l];
      s.mul = c->cfg.use_scale ? c->level_scales[l] : 1.f;
      v.push_back(s);
    }
  return v;
}

static int add_gn(sylph_ctx* c, Plan* P, std::vector<OpFn>& ops, void* x, const RowSeg* segs_dev, int nseg,
                  int max_rows, const GNLayer& G, int relu) {
  const DType dt = c->dt;
  float* partial = P->gn_partial;
  float2* stats = P->gn_stats;
  const float *ga = G.gamma, *be = G.beta;
  ops.push_back([=](hipStream_t s) {
    return launch_groupnorm(dt, x, segs_dev, nseg, max_rows, 256, ga, be, 1e-5f, relu, partial, stats, s);
  });
  return 0;
}

static int ensure_gn_ws(sylph_ctx* c, Plan* P, int nseg, int max_rows) {
  if (P->gn_partial) return 0;
  const int max_chunks = (max_rows + GN_ROWS_PER_CHUNK - 1) / GN_ROWS_PER_CHUNK;
  RET(c->dalloc((void**)&P->gn_partial, (size_t)nseg * max_chunks * 32 * 3 * sizeof(float)));
  RET(c->dalloc((void**)&P->gn_stats, (size_t)nseg * 32 * sizeof(float2)));
  return 0;
}

static int build_head(sylph_ctx* c, Plan* P) {
  if (P->head_built) return 0;
  if (!c->has_head) return fail("FCOS head weights were not loaded");
  RET(ensure_pyramid(c, P));
  const size_t e = c->esz();
  const size_t rows = (size_t)P->B * P->Ltot;
  const int L = c->cfg.nlevels, nseg = P->B * L;
  RET(c->dalloc(&P->tA, rows * 512 * e));  // paired towers: [rows][512]; unpaired: tA/tB = halves
  RET(c->dalloc(&P->tC, rows * 512 * e));
  P->tB = (char*)P->tA + rows * 256 * e;
  P->tD = (char*)P->tC + rows * 256 * e;
  RET(c->dalloc((void**)&P->pred, rows * 8 * sizeof(float)));
  std::vector<RowSeg> rs;
  for (int b = 0; b < P->B; ++b)
    for (int l = 0; l < L; ++l) rs.push_back(RowSeg{b * P->Ltot + P->off[l], P->hl[l] * P->wl[l]});
  RET(upload(c, (void**)&P->head_rowsegs, rs.data(), rs.size() * sizeof(RowSeg)));
  const int max_rows = P->hl[0] * P->wl[0];
  // the support plan of the same shape may already own a GN workspace sized for fewer segments
  if (P->gn_partial) { P->gn_partial = nullptr; P->gn_stats = nullptr; }
  RET(ensure_gn_ws(c, P, nseg > P->B ? nseg : P->B, max_rows));
  const std::vector<SegDesc> segs = pyramid_segs(c, P);
  auto& ops = P->head_ops;
  auto tower = [&](int which, const std::vector<ConvLayer>& convs, const std::vector<GNLayer>& gns, void* b0, void* b1,
                   void** last, const float2** coef_last, OpFn* apply_last) -> int {
    const bool defer_last = coef_last != nullptr;
    const void* in = P->F;
    void* out = b0;
    // GroupNorm + ReLU of layers 0 .. n-2 are applied by the NEXT layer's conv to its input halo in LDS (conv_hpipe.hip):
    // no separate streaming pass over those tensors.  The last layer keeps its apply pass (its readers are the
    // prediction convs and the class-conditional 1x1 conv).
    static const int gn_fuse_on = getenv("SYLPH_GN_FUSE") ? atoi(getenv("SYLPH_GN_FUSE")) : 1;
    ConvOpts probe; probe.pad = 1;
    const bool fuse = gn_fuse_on && convs.size() > 1 && convs[0].Cin <= 512 && use_hpipe(c, convs[1], segs, probe);
    const float2* coef_prev = nullptr;
    for (size_t i = 0; i < convs.size(); ++i) {
      ConvOpts o; o.pad = 1;
      if (coef_prev) { o.gn_coef = coef_prev; o.gn_relu = 1; }
      const float2* coef = nullptr;
      const bool is_last = i + 1 == convs.size();
      const bool defer = (fuse && !is_last) || (is_last && defer_last);
      OpFn apply;
      RET(add_conv_gn(c, ops, convs[i], in, 256, out, segs, o, gns[i], 1, defer ? &coef : nullptr, (is_last && defer_last) ? &apply : nullptr));
      if (is_last && defer_last) { *coef_last = coef; *apply_last = apply; }
      coef_prev = coef;
      P->tap_out[which].push_back(out);
      P->tap_coef[which].push_back(coef);
      in = out;
      out = (out == b0) ? b1 : b0;
      if (c->debug_taps && !is_last) RET(c->dalloc(&out, rows * 256 * e));  // keep every layer's output (same kernels, other destination)
    }
    *last = const_cast<void*>(in);
    return 0;
  };
  void *cls_feat = nullptr, *box_feat = nullptr;
  int feat_ld = 256;
  const float2* box_coef = nullptr;
  OpFn box_apply;
  bool box_defer = false;
  if (c->paired) {
    // tA|tB and tC|tD are used as two [rows][512] ping-pong buffers.  The towers run image-chunk by
    // image-chunk (depth first): a chunk's [rows][512] layer output (~23 MB per 800x1344 image) is
    // normalised and consumed by the next layer while it is still resident in the 256 MiB Infinity Cache.
    int chunk_imgs = P->B;
    if (const char* cz = getenv("SYLPH_HEAD_CHUNK")) chunk_imgs = atoi(cz) > 0 ? atoi(cz) : P->B;
    const void* in = nullptr;
    for (int b0 = 0; b0 < P->B; b0 += chunk_imgs) {
      const int b1 = b0 + chunk_imgs < P->B ? b0 + chunk_imgs : P->B;
      const std::vector<SegDesc> csegs(segs.begin() + (size_t)b0 * L, segs.begin() + (size_t)b1 * L);
      in = P->F;
      int in_ld = 256;
      void* out = P->tA;
      for (size_t i = 0; i < c->pair_tower.size(); ++i) {
        ConvOpts o; o.pad = 1;
        if (i > 0) { o.group_cout = 256; o.group_in_off = 256; }
        RET(add_conv_gn(c, ops, c->pair_tower[i], in, in_ld, out, csegs, o, c->pair_gn[i], 1));
        in = out; in_ld = 512;
        out = (out == P->tA) ? P->tC : P->tA;
      }
    }
    cls_feat = const_cast<void*>(in);
    box_feat = (char*)cls_feat + 256 * e;
    feat_ld = 512;
  } else {
    // the cls tower's last GroupNorm is left to sylph_fcos_head (fused into the class-conditional conv when N <= 32)
    static const int gn_logits_on = getenv("SYLPH_FUSE_GN_LOGITS") ? atoi(getenv("SYLPH_FUSE_GN_LOGITS")) : 1;
    P->cls_coef = nullptr; P->cls_apply = nullptr;
    const bool defer = gn_logits_on && c->dt == DT_BF16;
    OpFn cls_apply;
    RET(tower(0, c->cls_tower, c->cls_gn, P->tA, P->tB, &cls_feat, (defer && !c->cls_tower.empty()) ? &P->cls_coef : nullptr, &cls_apply));
    P->cls_apply = cls_apply;
    box_defer = defer && c->pred_taps && !c->box_tower.empty();
    RET(tower(1, c->box_tower, c->box_gn, P->tC, P->tD, &box_feat, box_defer ? &box_coef : nullptr, &box_apply));
  }
  P->cls_ld = feat_ld;
  Geom g32;  // 128-row pointwise tiles of the pyramid (class-conditional conv with N <= 32, fused GN + prediction pass)
  RET(make_geom(c, segs, 128, &g32));
  if (box_defer && box_coef) {
    // last bbox-tower GroupNorm + the 3x3 prediction convs: one streaming pass for the nine tap responses + a gather (head_fused.hip)
    const int cp = c->pred.Cout, sw = (3 * cp + 3) & ~3;
    float* taps_ws = nullptr;
    const size_t plane_rows = rows;
    RET(c->dalloc((void**)&taps_ws, (size_t)3 * rows * sw * sizeof(float)));
    const void *xin = box_feat, *wt = c->pred_taps;
    const float* bias = c->pred.shift;
    float* pout = P->pred;
    const SegDesc* sgd = g32.segs; const int2* tld = g32.tiles; const int ntl = g32.n_mtiles;
    const float2* bc = box_coef;
    const double fl = 2.0 * (double)rows * cp * 9.0 * 256.0;
    ops.push_back([=](hipStream_t s) {
      return timed_op(c, "gn_taps_kernel+tap_gather_kernel", fl, s, [=](hipStream_t st) { return launch_gn_pred_taps(xin, 256, bc, wt, cp, bias, 4, 4, taps_ws, plane_rows, pout, 8, sgd, tld, ntl, st); });
    });
  } else {
    ConvOpts op; op.pad = 1; op.relu_nch = 4; op.mul_nch = 4; op.out_f32 = true;
    RET(add_conv(c, ops, c->pred, box_feat, feat_ld, P->pred, 8, segs, op));
  }
  // geometry for the class-conditional 1x1 conv (weights arrive per call)
  {
    Geom g;
    int BM, BN;
    conv_pick_tile((int)rows, 128, 9, &BM, &BN);  // geometry for BN in {64,128}; BM from the 128-wide rule
    RET(make_geom(c, segs, BM, &g));
    P->head_segs = g.segs; P->head_tiles = g.tiles; P->head_mtiles = g.n_mtiles; P->head_BM = BM;
    P->head_tiles32 = g32.tiles; P->head_mtiles32 = g32.n_mtiles;
  }
  P->cls_feat = cls_feat;
  P->head_built = true;
  return 0;
}

// Per-(image, level) capacity of the decode candidate buffers.  The reference has no cap (boolean-mask indexing,
// fcos_outputs.py:960-990); here the scan compacts into a fixed buffer and overflow is reported.  Up to 262 144 slots the
// buffer holds EVERY (location, class) score of the largest level (5-way: 84 000, 20-way: 262 144 of 336 000), i.e. it cannot
// overflow for few-shot class counts; many-way episodes get 1/8 of the scores (LVIS 866-way: 1.8 M), at most 4 M
// (HBM is plentiful: 8 bytes per slot).
static int want_cand_cap(const sylph_ctx* c, const Plan* P) {
  if (c->cfg.cand_cap > 0) return c->cfg.cand_cap;
  const long all = (long)P->hl[0] * P->wl[0] * (long)(P->ncls > 0 ? P->ncls : 1);
  long w = all <= 262144 ? all : (all / 8 > 262144 ? all / 8 : 262144);
  if (w < 4096) w = 4096;
  if (w > (1L << 22)) w = 1L << 22;
  return (int)w;
}

static int build_decode(sylph_ctx* c, Plan* P) {
  if (P->decode_built) return 0;
  const int L = c->cfg.nlevels, B = P->B, nseg = B * L;
  std::vector<DecodeSeg> ds;
  for (int b = 0; b < B; ++b) {
    unsigned lb = 0;
    for (int l = 0; l < L; ++l) {
      DecodeSeg d;
      d.row0 = b * P->Ltot + P->off[l]; d.nloc = P->hl[l] * P->wl[l]; d.W = P->wl[l];
      d.stride = c->cfg.strides[l]; d.level = l; d.image = b; d.loc_base = lb; d.pad = 0;
      lb += (unsigned)d.nloc;
      ds.push_back(d);
    }
  }
  RET(upload(c, (void**)&P->dsegs, ds.data(), ds.size() * sizeof(DecodeSeg)));
  int pool = 64;
  while (pool < L * c->cfg.pre_nms_topk) pool <<= 1;
  if (pool > 8192) return fail("levels * PRE_NMS_TOPK exceeds the 8192-entry on-chip sort capacity");
  P->pool_cap = pool;
  P->cand_cap = want_cand_cap(c, P);
  DecodeBuffers& d = P->dbuf;
  RET(c->dalloc((void**)&d.cand_key, (size_t)nseg * P->cand_cap * 4));
  RET(c->dalloc((void**)&d.cand_idx, (size_t)nseg * P->cand_cap * 4));
  RET(c->dalloc((void**)&d.cand_count, (size_t)nseg * 4));
  RET(c->dalloc((void**)&d.sel_ws, (size_t)nseg * SEL_WS * 4));
  RET(c->dalloc((void**)&d.sel_tie, (size_t)nseg * SEL_TIE * 8));
  RET(c->dalloc((void**)&d.pool_key, (size_t)B * pool * 8));
  RET(c->dalloc((void**)&d.pool_count, (size_t)B * 4));
  RET(c->dalloc((void**)&d.s_box, (size_t)B * pool * 16));
  RET(c->dalloc((void**)&d.s_score, (size_t)B * pool * 4));
  RET(c->dalloc((void**)&d.s_cls, (size_t)B * pool * 4));
  RET(c->dalloc((void**)&d.s_level, (size_t)B * pool * 4));
  RET(c->dalloc((void**)&d.s_loc, (size_t)B * pool * 8));
  RET(c->dalloc((void**)&d.s_ord, (size_t)B * pool * 4));
  RET(c->dalloc((void**)&d.status, 4));
  RET(c->dalloc((void**)&P->img_out_dev, sizeof(ImageOut) * B));
  HIPCHK(hipHostMalloc((void**)&P->img_out_host, sizeof(ImageOut) * B));
  P->decode_built = true;
  return 0;
}

// more classes than when the decode buffers were built: grow the candidate buffers
static int ensure_cand_cap(sylph_ctx* c, Plan* P) {
  if (want_cand_cap(c, P) <= P->cand_cap) return 0;
  const int nseg = P->B * c->cfg.nlevels;
  P->cand_cap = want_cand_cap(c, P);
  c->dfree(P->dbuf.cand_key); c->dfree(P->dbuf.cand_idx);
  P->dbuf.cand_key = nullptr; P->dbuf.cand_idx = nullptr;
  RET(c->dalloc((void**)&P->dbuf.cand_key, (size_t)nseg * P->cand_cap * 4));
  RET(c->dalloc((void**)&P->dbuf.cand_idx, (size_t)nseg * P->cand_cap * 4));
  return 0;
}

static DecodeCfg decode_cfg(const sylph_ctx* c, const Plan* P, int max_out) {
  DecodeCfg d;
  d.num_classes = P->ncls; d.logits_ld = P->logits_ld; d.pre_nms_thresh = c->cfg.pre_nms_thresh;
  d.pre_nms_topk = c->cfg.pre_nms_topk; d.nms_thresh = c->cfg.nms_thresh; d.post_nms_topk = c->cfg.post_nms_topk;
  d.thresh_with_ctr = c->cfg.thresh_with_ctr; d.quality_mode = c->cfg.quality_mode; d.cand_cap = P->cand_cap;
  d.pool_cap = P->pool_cap; d.nlevels = c->cfg.nlevels; d.max_out = max_out;
  return d;
}

static int build_support(sylph_ctx* c, Plan* P) {
  if (P->support_built) return 0;
  if (!c->has_codegen) return fail("code generator weights were not loaded");
  RET(ensure_pyramid(c, P));
  const size_t e = c->esz();
  const int S = P->B, L = c->cfg.nlevels, npos = 49;
  std::vector<LevelDesc> lv;
  for (int b = 0; b < S; ++b)
    for (int l = 0; l < L; ++l)
      lv.push_back(LevelDesc{b * P->Ltot + P->off[l], P->hl[l], P->wl[l], 1.0f / (float)c->cfg.strides[l]});
  RET(upload(c, (void**)&P->lv_dev, lv.data(), lv.size() * sizeof(LevelDesc)));
  RET(c->dalloc(&P->roi, (size_t)S * npos * 256 * e));
  RET(c->dalloc(&P->cgA, (size_t)S * npos * 256 * e));
  RET(c->dalloc(&P->cgB, (size_t)S * npos * 256 * e));
  RET(c->dalloc((void**)&P->cg_conv_out, (size_t)S * npos * 256 * 4));
  RET(c->dalloc((void**)&P->cg_bias_out, (size_t)S * npos * 4 * (c->cg_naux > 0 ? c->cg_naux : 1)));
  RET(c->dalloc((void**)&P->cg_wnorm, (size_t)S * 4));
  RET(ensure_gn_ws(c, P, S, P->hl[0] * P->wl[0]));
  std::vector<RowSeg> rs;
  for (int s = 0; s < S; ++s) rs.push_back(RowSeg{s * npos, npos});
  RowSeg* rs_dev = nullptr;
  RET(upload(c, (void**)&rs_dev, rs.data(), rs.size() * sizeof(RowSeg)));
  const std::vector<SegDesc> segs = image_segs(S, 7, 7, 7, 7);
  auto& ops = P->support_ops;
  const DType dt = c->dt;
  Plan* PP = P;
  {
    const void* F = P->F;
    const LevelDesc* lvd = P->lv_dev;
    void* roi = P->roi;
    ops.push_back([=](hipStream_t s) { return launch_roi_align(dt, F, 256, lvd, L, PP->cur_boxes, S, 7, roi, s); });
  }
  const void* in = P->roi;
  void* out = P->cgA;
  for (size_t i = 0; i < c->cg_tower.size(); ++i) {
    ConvOpts o; o.pad = 1;
    RET(add_conv_gn(c, ops, c->cg_tower[i], in, 256, out, segs, o, c->cg_gn[i], 1));
    in = out;
    out = (out == P->cgA) ? P->cgB : P->cgA;
  }
  ConvOpts oc; oc.pad = 1; oc.out_f32 = true;
  RET(add_conv(c, ops, c->cg_cls, in, 256, P->cg_conv_out, 256, segs, oc));
  const int naux = c->cg_naux;
  if (naux > 0) RET(add_conv(c, ops, c->cg_bias, in, 256, P->cg_bias_out, naux, segs, oc));
  {
    const float *co = P->cg_conv_out, *bo = P->cg_bias_out;
    const int l2 = c->cfg.cg_bias_l2_norm, ib = c->cg_ib, iw = c->cg_iw, is = c->cg_is;
    float* wn = P->cg_wnorm;
    ops.push_back([=](hipStream_t s) {
      const int shots = PP->cur_shots > 0 ? PP->cur_shots : S;
      return launch_codegen_tail(co, 256, bo, naux > 0 ? naux : 1, ib, iw, is, S / shots, shots, npos, 256, l2, PP->cur_code_out, wn, s);
    });
  }
  P->support_built = true;
  return 0;
}

static int build_support_roienc(sylph_ctx* c, Plan* P) {
  if (P->support_built) return 0;
  if (!c->has_roienc) return fail("ROIEncoder weights were not loaded");
  RET(ensure_pyramid(c, P));
  const size_t e = c->esz();
  // S support images = one or several classes of P->cur_shots images each (sylph_codegen_classes).  Everything up to the encoder
  // is per image; the reference's encoder attends over the CLASS axis of a (classes, shots, C) tensor (roi_encoder.py:184-186)
  // and always sees one class per call at inference, i.e. a length-1 sequence: here too a class never sees another one.
  const int S = P->B, L = c->cfg.nlevels, npos = 49;
  std::vector<LevelDesc> lv;
  for (int b = 0; b < S; ++b)
    for (int l = 0; l < L; ++l)
      lv.push_back(LevelDesc{b * P->Ltot + P->off[l], P->hl[l], P->wl[l], 1.0f / (float)c->cfg.strides[l]});
  RET(upload(c, (void**)&P->lv_dev, lv.data(), lv.size() * sizeof(LevelDesc)));
  RET(c->dalloc(&P->roi, (size_t)S * npos * 256 * e));
  RET(c->dalloc(&P->cgA, (size_t)S * npos * 256 * e));
  RET(c->dalloc(&P->cgB, (size_t)S * npos * 256 * e));
  RET(c->dalloc((void**)&P->re_ctx, (size_t)S * npos * 256 * 4));
  RET(c->dalloc((void**)&P->re_tok, (size_t)S * 256 * 4));
  RET(c->dalloc((void**)&P->re_tmp, (size_t)S * 256 * 4));
  int maxhid = 1024;
  for (auto& l : c->re.layers) maxhid = l.l1.O > maxhid ? l.l1.O : maxhid;
  RET(c->dalloc((void**)&P->re_hid, (size_t)S * maxhid * 4));
  const int hdim = c->cfg.head_fc_dim > 256 ? c->cfg.head_fc_dim : 256;
  RET(c->dalloc((void**)&P->re_cls, (size_t)S * 256 * 4));
  RET(c->dalloc((void**)&P->re_h, (size_t)2 * S * hdim * 4));
  const std::vector<SegDesc> segs = image_segs(S, 7, 7, 7, 7);
  auto& ops = P->support_ops;
  const DType dt = c->dt;
  const int xbf = dt == DT_BF16 ? 1 : 0;
  Plan* PP = P;
  auto& R = c->re;
  {
    const void* F = P->F;
    const LevelDesc* lvd = P->lv_dev;
    void* roi = P->roi;
    float* ctx = P->re_ctx;
    ops.push_back([=](hipStream_t s) { return launch_roi_align(dt, F, 256, lvd, L, PP->cur_boxes, S, 7, roi, s); });
    ops.push_back([=](hipStream_t s) { return launch_adaptive_context(dt, F, 256, lvd, L, S, 7, ctx, s); });
  }
  ConvOpts o; o.pad = 1;
  RET(add_conv_gn(c, ops, R.pool_conv, P->roi, 256, P->cgA, segs, o, R.pool_gn, 1));
  {
    const float* ctx = P->re_ctx;
    void* x = P->cgA;
    const MsCamWeights w = R.cam;
    ops.push_back([=](hipStream_t s) { return launch_mscam(dt, ctx, x, S, w, s); });
  }
  void* cur = P->cgA;
  void* nxt = P->cgB;
  for (size_t k = 0; k < R.tok_conv.size(); ++k) {
    RET(add_conv_gn(c, ops, R.tok_conv[k], cur, 256, nxt, segs, o, R.tok_gn[k], 1));
    std::swap(cur, nxt);
  }
  // tokenizer FC stack: first FC reads the (position-major) activations directly
  float* tok = P->re_tok;
  float* tmp = P->re_tmp;
  float* hid = P->re_hid;
  {
    const sylph_ctx::Lin f0 = R.tok_fc[0];
    const void* x = cur;
    ops.push_back([=](hipStream_t s) { return launch_linear(xbf, x, npos * 256, S, f0.W, f0.b, f0.K, f0.O, tok, 256, 1, 0.f, s); });
    float* a = tok;
    float* b = tmp;
    for (size_t k = 1; k < R.tok_fc.size(); ++k) {
      const sylph_ctx::Lin f = R.tok_fc[k];
      ops.push_back([=](hipStream_t s) { return launch_linear(0, a, 256, S, f.W, f.b, f.K, f.O, b, 256, 1, 0.f, s); });
      std::swap(a, b);
    }
    tok = a;
    tmp = b;
  }
  for (auto& l : R.layers) {
    const sylph_ctx::Lin at = l.attn, l1 = l.l1, l2 = l.l2;
    const GNLayer n1 = l.n1, n2 = l.n2;
    float *x = tok, *t = tmp;
    ops.push_back([=](hipStream_t s) { return launch_linear(0, x, 256, S, at.W, at.b, 256, 256, t, 256, 0, 0.f, s); });
    ops.push_back([=](hipStream_t s) { return launch_add_layernorm(x, t, S, n1.gamma, n1.beta, s); });
    ops.push_back([=](hipStream_t s) { return launch_linear(0, x, 256, S, l1.W, l1.b, l1.K, l1.O, hid, l1.O, 1, 0.f, s); });
    ops.push_back([=](hipStream_t s) { return launch_linear(0, hid, l1.O, S, l2.W, l2.b, l2.K, l2.O, t, 256, 0, 0.f, s); });
    ops.push_back([=](hipStream_t s) { return launch_add_layernorm(x, t, S, n2.gamma, n2.beta, s); });
  }
  {
    float* cls = P->re_cls;
    float* x = tok;
    ops.push_back([=](hipStream_t s) { return launch_mean_tokens(x, S / PP->cur_shots, PP->cur_shots, cls, s); });
    const float prior = -logf((1.f - 0.01f) / 0.01f);  // ROIEncoder hard-codes prior_prob = 0.01 (roi_encoder.py:139-140), whatever MODEL.FCOS.PRIOR_PROB says
    for (int head = 0; head < 2; ++head) {
      const std::vector<sylph_ctx::Lin>& fcs = head == 0 ? R.wh : R.bh;
      const float* in = cls;
      float* h0 = P->re_h + (size_t)head * S * hdim;
      for (size_t k = 0; k < fcs.size(); ++k) {
        const sylph_ctx::Lin f = fcs[k];
        const bool last = k + 1 == fcs.size();
        const float add = (last && head == 1) ? prior : 0.f;
        const int off = head == 0 ? 0 : 256;
        if (last) {
          ops.push_back([=](hipStream_t s) {  // class k -> row k of the (classes, 257) output
            return launch_linear(0, in, f.K, S / PP->cur_shots, f.W, f.b, f.K, f.O, PP->cur_code_out + off, 257, 0, add, s);
          });
        } else {
          ops.push_back([=](hipStream_t s) { return launch_linear(0, in, f.K, S / PP->cur_shots, f.W, f.b, f.K, f.O, h0, f.O, 1, 0.f, s); });
          in = h0;
        }
      }
    }
  }
  P->support_built = true;
  return 0;
}

static int run_ops(sylph_ctx* c, const std::vector<OpFn>& ops, const char* what) {
  for (size_t i = 0; i < ops.size(); ++i) {
    const int r = ops[i](c->stream);
    if (r != 0) return fail(std::string(what) + ": op " + std::to_string(i) + " failed with " + std::to_string(r));
  }
  return 0;
}

// ================================================================================================
extern "C" {

void sylph_config_default(sylph_config* cfg) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->resnet_depth = 50; cfg->stride_in_1x1 = 1; cfg->num_cls_convs = 4; cfg->num_box_convs = 4;
  cfg->nlevels = 5;
  const int st[5] = {8, 16, 32, 64, 128};
  for (int i = 0; i < 5; ++i) cfg->strides[i] = st[i];
  cfg->pixel_mean[0] = 103.530f; cfg->pixel_mean[1] = 116.280f; cfg->pixel_mean[2] = 123.675f;
  cfg->pixel_std[0] = cfg->pixel_std[1] = cfg->pixel_std[2] = 1.f;
  cfg->size_divisibility = 32; cfg->use_scale = 1; cfg->cond_use_bias = 1;
  cfg->pre_nms_thresh = 0.05f; cfg->pre_nms_topk = 1000; cfg->nms_thresh = 0.6f; cfg->post_nms_topk = 100;
  cfg->thresh_with_ctr = 0; cfg->quality_mode = 0;
  cfg->cg_tower_layers = 2; cfg->cg_has_bias = 1; cfg->cg_bias_l2_norm = 0; cfg->cg_post_norm = 1;
  cfg->cg_conv_l2_norm = 1; cfg->cg_use_weight_scale = 1; cfg->prior_prob = 0.01f; cfg->cand_cap = 0;
  cfg->cg_type = 0; cfg->tok_num_conv = 2; cfg->tok_num_fc = 2; cfg->enc_layers = 2; cfg->head_num_fc = 2;
  cfg->head_fc_dim = 512;
  cfg->cg_meta_bias = 0;
  cfg->cg_has_weight = 0; cfg->cg_has_scale = 0;
}

const char* sylph_last_error(void) { return g_err.c_str(); }

int sylph_ctx_create(int device_id, int dtype, sylph_ctx** out) {
  if (!out) return fail("out is NULL");
  if (dtype != SYLPH_F32 && dtype != SYLPH_BF16) return fail("dtype must be SYLPH_F32 or SYLPH_BF16");
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (device_id < 0 || device_id >= n) return fail("no such HIP device: " + std::to_string(device_id));
  HIPCHK(hipSetDevice(device_id));
  sylph_ctx* c = new sylph_ctx();
  c->device = device_id;
  c->dt = dtype == SYLPH_BF16 ? DT_BF16 : DT_F32;
  sylph_config_default(&c->cfg);
  conv_set_nbuf(SYLPH_AB_ENV("SYLPH_CONV_NBUF", 1));  // A/B knob (-DSYLPH_ABLATE builds): LDS stages of the conv kernel
  if (const char* mp = getenv("SYLPH_MAX_PLANS")) c->max_plans = atoi(mp) > 1 ? (size_t)atoi(mp) : 2;
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) c->plan_byte_budget = (int64_t)(total_b / 10 * 6);  // 60 % of HBM for workspaces
    if (const char* pb = getenv("SYLPH_PLAN_BYTES_MB")) c->plan_byte_budget = (int64_t)atol(pb) << 20;
  }
  if (c->dalloc(&c->zeros, 256) != 0 || hipMemset(c->zeros, 0, 256) != hipSuccess) {
    delete c;
    return fail("cannot allocate the zero page");
  }
  *out = c;
  return 0;
}

void sylph_ctx_destroy(sylph_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto& kv : c->plans) free_plan(c, kv.second.get());
  c->plans.clear();
  for (void* p : c->allocs) (void)hipFree(p);
  delete c;
}

int sylph_set_stream(sylph_ctx* c, void* s) {
  c->stream = (hipStream_t)s;
  return 0;
}

int sylph_set_config(sylph_ctx* c, const sylph_config* cfg) {
  if (c->finalized) return fail("sylph_set_config must precede sylph_finalize_weights");
  if (cfg->nlevels != 5) return fail("only the 5-level FCOS pyramid (p3..p7) is supported");
  if (cfg->resnet_depth != 50 && cfg->resnet_depth != 101 && cfg->resnet_depth != 152)
    return fail("MODEL.RESNETS.DEPTH must be 50, 101 or 152");
  c->cfg = *cfg;
  return 0;
}

int sylph_load_weight(sylph_ctx* c, const char* name, const float* data, const int64_t* shape, int ndim) {
  if (c->finalized) return fail("weights already finalized");
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.data.assign(data, data + n);
  c->host_w[name] = std::move(t);
  return 0;
}

int sylph_finalize_weights(sylph_ctx* c) {
  if (c->finalized) return fail("weights already finalized");
  HIPCHK(hipSetDevice(c->device));
  const std::string bu = "backbone.bottom_up";
  if (has_prefix(c, bu + ".stem")) {
    const HostTensor* w = find_w(c, bu + ".stem.conv1.weight");
    const HostTensor *g = find_w(c, bu + ".stem.conv1.norm.weight"), *b = find_w(c, bu + ".stem.conv1.norm.bias");
    const HostTensor *rm = find_w(c, bu + ".stem.conv1.norm.running_mean"),
                     *rv = find_w(c, bu + ".stem.conv1.norm.running_var");
    if (!w || !g || !b || !rm || !rv) return fail("missing stem weights");
    if (w->shape[0] != 64 || w->shape[1] != 3 || w->shape[2] != 7 || w->shape[3] != 7) return fail("stem must be 64x3x7x7");
    {
      // Stem as an implicit GEMM (conv_igemm.hip, stem loader): K-slice s = RPS kernel rows of an
      // 8-pixel x 4-channel window; window pixel 0, channel 3 and kernel rows >= 7 carry zeros.
      const int BK = c->dt == DT_BF16 ? 64 : 32, RPS = BK / 32, NS = (7 + RPS - 1) / RPS;
      HostTensor hs;
      hs.shape = {64, BK, NS, 1};
      hs.data.assign((size_t)64 * BK * NS, 0.f);
      for (int n = 0; n < 64; ++n)
        for (int s = 0; s < NS; ++s)
          for (int e = 0; e < BK; ++e) {
            const int kh = s * RPS + e / 32, px = (e % 32) / 4, ch = e % 4, kw = px - 1;
            if (kh < 7 && kw >= 0 && kw < 7 && ch < 3)
              hs.data[((size_t)n * BK + e) * NS + s] = w->data[((n * 3 + ch) * 7 + kh) * 7 + kw];
          }
      RET(pack_conv(c, {&hs}, &c->stem));
      std::vector<float> sc(64), sh(64);
      for (int i = 0; i < 64; ++i) {
        sc[i] = g->data[i] * (1.0f / sqrtf(rv->data[i] + 1e-5f));
        sh[i] = b->data[i] - rm->data[i] * sc[i];
      }
      RET(upload_vec(c, &c->stem.scale, sc, c->stem.Cout_pad));
      RET(upload_vec(c, &c->stem.shift, sh, c->stem.Cout_pad));
      if (c->dt == DT_BF16) {  // dedicated stem kernel: [n][kh][8 px][4 ch], kernel column 7 / channel 3 zero
        std::vector<bf16_t> wp((size_t)64 * 224);
        for (int n = 0; n < 64; ++n)
          for (int kh = 0; kh < 7; ++kh)
            for (int px = 0; px < 8; ++px)
              for (int ch = 0; ch < 4; ++ch)
                wp[(size_t)n * 224 + kh * 32 + px * 4 + ch] =
                    (bf16_t)((px < 7 && ch < 3) ? w->data[((n * 3 + ch) * 7 + kh) * 7 + px] : 0.f);
        RET(upload(c, &c->stem_wp, wp.data(), wp.size() * sizeof(bf16_t)));
      }
    }
    const int nb50[4] = {3, 4, 6, 3}, nb101[4] = {3, 4, 23, 3}, nb152[4] = {3, 8, 36, 3};
    const int* nb = c->cfg.resnet_depth == 50 ? nb50 : (c->cfg.resnet_depth == 101 ? nb101 : nb152);
    c->stages.resize(4);
    for (int si = 0; si < 4; ++si) {
      c->stages[si].resize(nb[si]);
      for (int bi = 0; bi < nb[si]; ++bi) {
        const std::string q = bu + ".res" + std::to_string(si + 2) + "." + std::to_string(bi);
        auto& blk = c->stages[si][bi];
        RET(make_conv_bn(c, q + ".conv1", &blk.c1));
        RET(make_conv_bn(c, q + ".conv2", &blk.c2));
        RET(make_conv_bn(c, q + ".conv3", &blk.c3));
        blk.has_sc = bi == 0;
        if (blk.has_sc) RET(make_conv_bn(c, q + ".shortcut", &blk.sc));
        const char* fz = getenv("SYLPH_FUSE_SHORTCUT");
        if (blk.has_sc && !(fz && atoi(fz) == 0)) {
          // fold the two FrozenBN scales into the weights, sum the shifts (fp32 before the dtype cast)
          const HostTensor *w3 = find_w(c, q + ".conv3.weight"), *ws = find_w(c, q + ".shortcut.weight");
          const int co = (int)w3->shape[0];
          std::vector<float> s3(co), h3(co), ss(co), hs(co);
          for (int i = 0; i < co; ++i) {
            const float a3 = find_w(c, q + ".conv3.norm.weight")->data[i] *
                             (1.0f / sqrtf(find_w(c, q + ".conv3.norm.running_var")->data[i] + 1e-5f));
            s3[i] = a3;
            h3[i] = find_w(c, q + ".conv3.norm.bias")->data[i] - find_w(c, q + ".conv3.norm.running_mean")->data[i] * a3;
            const float as = find_w(c, q + ".shortcut.norm.weight")->data[i] *
                             (1.0f / sqrtf(find_w(c, q + ".shortcut.norm.running_var")->data[i] + 1e-5f));
            ss[i] = as;
            hs[i] = find_w(c, q + ".shortcut.norm.bias")->data[i] - find_w(c, q + ".shortcut.norm.running_mean")->data[i] * as;
          }
          RET(make_c3sc(c, *w3, s3.data(), h3.data(), *ws, ss.data(), hs.data(), &blk.c3sc));
          blk.fused_sc = true;
        }
      }
    }
    for (int k = 0; k < 3; ++k) {
      RET(make_conv_bias(c, {"backbone.fpn_lateral" + std::to_string(k + 3)}, &c->fpn_lat[k]));
      RET(make_conv_bias(c, {"backbone.fpn_output" + std::to_string(k + 3)}, &c->fpn_out[k]));
    }
    RET(make_conv_bias(c, {"backbone.top_block.p6"}, &c->p6));
    RET(make_conv_bias(c, {"backbone.top_block.p7"}, &c->p7));
    c->has_backbone = true;
  }
  const std::string hp = "proposal_generator.fcos_head";
  if (has_prefix(c, hp + ".cls_tower") || has_prefix(c, hp + ".bbox_tower")) {
    c->cls_tower.resize(c->cfg.num_cls_convs); c->cls_gn.resize(c->cfg.num_cls_convs);
    c->box_tower.resize(c->cfg.num_box_convs); c->box_gn.resize(c->cfg.num_box_convs);
    for (int i = 0; i < c->cfg.num_cls_convs; ++i) {
      RET(make_conv_bias(c, {hp + ".cls_tower." + std::to_string(3 * i)}, &c->cls_tower[i]));
      RET(make_gn(c, hp + ".cls_tower." + std::to_string(3 * i + 1), &c->cls_gn[i]));
    }
    for (int i = 0; i < c->cfg.num_box_convs; ++i) {
      RET(make_conv_bias(c, {hp + ".bbox_tower." + std::to_string(3 * i)}, &c->box_tower[i]));
      RET(make_gn(c, hp + ".bbox_tower." + std::to_string(3 * i + 1), &c->box_gn[i]));
    }
    const int pair_on = SYLPH_AB_ENV("SYLPH_PAIR_TOWERS", 0);  // A/B knob (-DSYLPH_ABLATE builds only)
    // Pairing (both towers as ONE grouped launch per layer) paid +2 % with the pre-halo kernel (the A tile was shared by
    // four N tiles); with halo tiles the separate towers are 1 % faster (1 666-1 672 vs 1 645-1 660 img/s), so it is opt-in.
    if (c->cfg.num_cls_convs == c->cfg.num_box_convs && c->cfg.num_cls_convs > 0 && pair_on == 1) {
      // run both towers as ONE launch per layer: outputs side by side ([rows][512] = cls | bbox)
      const int n = c->cfg.num_cls_convs;
      c->pair_tower.resize(n); c->pair_gn.resize(n);
      for (int i = 0; i < n; ++i) {
        RET(make_conv_bias(c, {hp + ".cls_tower." + std::to_string(3 * i), hp + ".bbox_tower." + std::to_string(3 * i)},
                           &c->pair_tower[i]));
        std::vector<float> ga, be;
        for (const char* t : {".cls_tower.", ".bbox_tower."}) {
          const HostTensor *g = find_w(c, hp + t + std::to_string(3 * i + 1) + ".weight"),
                           *b = find_w(c, hp + t + std::to_string(3 * i + 1) + ".bias");
          if (!g || !b || g->data.size() != 256) return fail("missing GroupNorm weights of the FCOS towers");
          ga.insert(ga.end(), g->data.begin(), g->data.end());
          be.insert(be.end(), b->data.begin(), b->data.end());
        }
        RET(upload_vec(c, &c->pair_gn[i].gamma, ga, 512));
        RET(upload_vec(c, &c->pair_gn[i].beta, be, 512));
      }
      c->paired = true;
    }
    RET(make_conv_bias(c, {hp + ".bbox_pred", hp + ".ctrness", hp + ".iou_overlap"}, &c->pred));
    if (find_w(c, hp + ".cls_logits.weight") && find_w(c, hp + ".cls_logits.bias")) {
      const HostTensor* w = find_w(c, hp + ".cls_logits.weight");
      if (w->shape.size() == 4 && w->shape[1] == 256 && w->shape[2] == w->shape[3] && (w->shape[2] == 1 || w->shape[2] == 3)) {
        RET(make_conv_bias(c, {hp + ".cls_logits"}, &c->cls_logits));
        c->has_cls_logits = true;
      }
    }
    c->pred_taps = nullptr;
    if (c->dt == DT_BF16 && c->pred.KH == 3 && c->pred.KW == 3 && c->pred.Cin == 256 && 3 * ((3 * c->pred.Cout + 3) & ~3) <= 64) {
      // the same weights stacked for the fused GroupNorm + prediction pass: row kh * sw + kw * Cout + n, sw = roundup4(3 * Cout)
      const int cp = c->pred.Cout, sw = (3 * cp + 3) & ~3;
      std::vector<uint16_t> tw((size_t)64 * 256, 0);
      int n0 = 0;
      for (const char* nm : {".bbox_pred", ".ctrness", ".iou_overlap"}) {
        const HostTensor* w = find_w(c, hp + nm + ".weight");
        if (!w) continue;
        const int co = (int)w->shape[0];
        for (int n = 0; n < co; ++n)
          for (int ci = 0; ci < 256; ++ci)
            for (int tap = 0; tap < 9; ++tap)
              tw[(size_t)((tap / 3) * sw + (tap % 3) * cp + n0 + n) * 256 + ci] = f2bf_host(w->data[((size_t)n * 256 + ci) * 9 + tap]);
        n0 += co;
      }
      RET(upload(c, &c->pred_taps, tw.data(), tw.size() * 2));
    }
    c->level_scales.assign(c->cfg.nlevels, 1.f);
    if (c->cfg.use_scale)
      for (int l = 0; l < c->cfg.nlevels; ++l) {
        const HostTensor* s = find_w(c, hp + ".scales." + std::to_string(l) + ".scale");
        if (!s) return fail("missing " + hp + ".scales." + std::to_string(l) + ".scale");
        c->level_scales[l] = s->data[0];
      }
    c->has_head = true;
  }
  const std::string cp = "code_generator.code_generator_head";
  if (has_prefix(c, cp)) {
    c->cg_tower.resize(c->cfg.cg_tower_layers); c->cg_gn.resize(c->cfg.cg_tower_layers);
    for (int i = 0; i < c->cfg.cg_tower_layers; ++i) {
      RET(make_conv_bias(c, {cp + ".support_set_shared_tower." + std::to_string(3 * i)}, &c->cg_tower[i]));
      RET(make_gn(c, cp + ".support_set_shared_tower." + std::to_string(3 * i + 1), &c->cg_gn[i]));
    }
    RET(make_conv_bias(c, {cp + ".support_set_cls_conv.0"}, &c->cg_cls));
    {
      std::vector<std::string> aux;
      if (c->cfg.cg_has_bias) { c->cg_ib = (int)aux.size(); aux.push_back(cp + ".support_set_cls_bias.0"); }
      if (c->cfg.cg_has_weight) { c->cg_iw = (int)aux.size(); aux.push_back(cp + ".support_set_cls_weight.0"); }
      if (c->cfg.cg_has_scale) { c->cg_is = (int)aux.size(); aux.push_back(cp + ".support_set_cls_scale.0"); }
      c->cg_naux = (int)aux.size();
      if (!aux.empty()) RET(make_conv_bias(c, aux, &c->cg_bias));
    }
    if (c->cfg.cg_post_norm) RET(make_gn(c, cp + ".post_norm", &c->cg_post));
    // conv_scale exists iff USE_WEIGHT_SCALE and (CONV_L2_NORM or POST_NORM)  (code_generator.py:372-374)
    c->cg_conv_scale = 1.f;
    if (c->cfg.cg_use_weight_scale && (c->cfg.cg_conv_l2_norm || c->cfg.cg_post_norm)) {
      const HostTensor* s = find_w(c, cp + ".conv_scale.scale");
      if (!s) return fail("missing " + cp + ".conv_scale.scale");
      c->cg_conv_scale = s->data[0];
    }
    c->cg_bias_scale = 1.f;
    if (c->cfg.cg_has_bias) {
      const HostTensor* s = find_w(c, cp + ".bias_scale.scale");
      if (!s) return fail("missing " + cp + ".bias_scale.scale");
      c->cg_bias_scale = s->data[0];
    }
    c->cg_bias_prior = -logf((1.f - c->cfg.prior_prob) / c->cfg.prior_prob);
    if (c->cfg.cg_meta_bias) {
      const HostTensor* bv = find_w(c, cp + ".bias_value");
      if (!bv || bv->data.empty()) return fail("META_BIAS is set but " + cp + ".bias_value is missing from the checkpoint");
      c->cg_bias_prior = bv->data[0];
    }
    c->has_codegen = true;
  }
  if (c->cfg.cg_type == 1 && has_prefix(c, "code_generator.box_pooler")) {
    const std::string rp = "code_generator";
    auto& R = c->re;
    RET(make_conv_bias(c, {rp + ".box_pooler.conv.0"}, &R.pool_conv));
    RET(make_gn(c, rp + ".box_pooler.conv.1", &R.pool_gn));
    const std::string cam = rp + ".box_pooler.context_attention_module";
    RET(upload_f32(c, &R.cam.l_w1, find_w(c, cam + ".local_att.0.weight"), cam, 64 * 256));
    RET(upload_f32(c, &R.cam.l_b1, find_w(c, cam + ".local_att.0.bias"), cam, 64));
    RET(upload_f32(c, &R.cam.l_g1, find_w(c, cam + ".local_att.1.weight"), cam, 64));
    RET(upload_f32(c, &R.cam.l_be1, find_w(c, cam + ".local_att.1.bias"), cam, 64));
    RET(upload_f32(c, &R.cam.l_w2, find_w(c, cam + ".local_att.3.weight"), cam, 256 * 64));
    RET(upload_f32(c, &R.cam.l_b2, find_w(c, cam + ".local_att.3.bias"), cam, 256));
    RET(upload_f32(c, &R.cam.l_g2, find_w(c, cam + ".local_att.4.weight"), cam, 256));
    RET(upload_f32(c, &R.cam.l_be2, find_w(c, cam + ".local_att.4.bias"), cam, 256));
    RET(upload_f32(c, &R.cam.g_w1, find_w(c, cam + ".global_att.1.weight"), cam, 64 * 256));
    RET(upload_f32(c, &R.cam.g_b1, find_w(c, cam + ".global_att.1.bias"), cam, 64));
    RET(upload_f32(c, &R.cam.g_g1, find_w(c, cam + ".global_att.2.weight"), cam, 64));
    RET(upload_f32(c, &R.cam.g_be1, find_w(c, cam + ".global_att.2.bias"), cam, 64));
    RET(upload_f32(c, &R.cam.g_w2, find_w(c, cam + ".global_att.4.weight"), cam, 256 * 64));
    RET(upload_f32(c, &R.cam.g_b2, find_w(c, cam + ".global_att.4.bias"), cam, 256));
    RET(upload_f32(c, &R.cam.g_g2, find_w(c, cam + ".global_att.5.weight"), cam, 256));
    RET(upload_f32(c, &R.cam.g_be2, find_w(c, cam + ".global_att.5.bias"), cam, 256));
    R.tok_conv.resize(c->cfg.tok_num_conv); R.tok_gn.resize(c->cfg.tok_num_conv);
    for (int k = 0; k < c->cfg.tok_num_conv; ++k) {
      const std::string q = rp + ".tokenizer.conv" + std::to_string(k + 1);
      const HostTensor* w = find_w(c, q + ".weight");
      if (!w) return fail("missing weights for " + q);
      if (find_w(c, q + ".bias")) return fail(q + ": a conv bias together with TOKENIZER.NORM is not supported");
      RET(pack_conv(c, {w}, &R.tok_conv[k]));
      RET(make_gn(c, q + ".norm", &R.tok_gn[k]));
    }
    if (c->cfg.tok_num_fc < 1) return fail("TOKENIZER.NUM_FC must be >= 1");
    R.tok_fc.resize(c->cfg.tok_num_fc);
    for (int k = 0; k < c->cfg.tok_num_fc; ++k) {
      const std::string q = rp + ".tokenizer.fc" + std::to_string(k + 1);
      if (k == 0) {
        // nn.Flatten order is (c, p); activations here are position-major (p, c): permute the columns once
        HostTensor* w = const_cast<HostTensor*>(find_w(c, q + ".weight"));
        if (!w || w->shape.size() != 2 || w->shape[1] != 256 * 49) return fail("tokenizer.fc1 must take 256*7*7 inputs");
        std::vector<float> perm(w->data.size());
        const int O = (int)w->shape[0];
        for (int o = 0; o < O; ++o)
          for (int ch = 0; ch < 256; ++ch)
            for (int pp = 0; pp < 49; ++pp) perm[(size_t)o * 12544 + pp * 256 + ch] = w->data[(size_t)o * 12544 + ch * 49 + pp];
        w->data.swap(perm);
      }
      RET(make_lin(c, q, &R.tok_fc[k]));
      if (R.tok_fc[k].O != 256) return fail("TOKENIZER.FC_DIM must be 256");
    }
    R.layers.resize(c->cfg.enc_layers);
    for (int l = 0; l < c->cfg.enc_layers; ++l) {
      const std::string q = rp + ".transformer_encoder.layers." + std::to_string(l);
      const HostTensor *ipw = find_w(c, q + ".self_attn.in_proj_weight"), *ipb = find_w(c, q + ".self_attn.in_proj_bias");
      const HostTensor *ow = find_w(c, q + ".self_attn.out_proj.weight"), *ob = find_w(c, q + ".self_attn.out_proj.bias");
      if (!ipw || !ipb || !ow || !ob || ipw->data.size() != 3 * 256 * 256) return fail("missing weights for " + q);
      // sequence length 1 => attention weights are 1: SA(x) = Wo (Wv x + bv) + bo, folded into one matrix
      HostTensor fw, fb;
      fw.shape = {256, 256}; fw.data.resize(256 * 256);
      fb.shape = {256}; fb.data.resize(256);
      const float* Wv = ipw->data.data() + 2 * 256 * 256;
      const float* bv = ipb->data.data() + 2 * 256;
      for (int i = 0; i < 256; ++i) {
        for (int k = 0; k < 256; ++k) {
          double a = 0.0;
          for (int j = 0; j < 256; ++j) a += (double)ow->data[i * 256 + j] * (double)Wv[j * 256 + k];
          fw.data[i * 256 + k] = (float)a;
        }
        double bb = ob->data[i];
        for (int j = 0; j < 256; ++j) bb += (double)ow->data[i * 256 + j] * (double)bv[j];
        fb.data[i] = (float)bb;
      }
      c->host_w[q + ".folded_attn.weight"] = fw;
      c->host_w[q + ".folded_attn.bias"] = fb;
      RET(make_lin(c, q + ".folded_attn", &R.layers[l].attn));
      RET(make_lin(c, q + ".linear1", &R.layers[l].l1));
      RET(make_lin(c, q + ".linear2", &R.layers[l].l2));
      RET(make_ln(c, q + ".norm1", &R.layers[l].n1));
      RET(make_ln(c, q + ".norm2", &R.layers[l].n2));
    }
    if (c->cfg.head_num_fc < 1 || c->cfg.head_num_fc > 2) return fail("HEAD.NUM_FC must be 1 or 2");
    R.wh.resize(c->cfg.head_num_fc); R.bh.resize(c->cfg.head_num_fc);
    for (int k = 0; k < c->cfg.head_num_fc; ++k) {
      RET(make_lin(c, rp + ".weight_head.fc" + std::to_string(k + 1), &R.wh[k]));
      RET(make_lin(c, rp + ".bias_head.fc" + std::to_string(k + 1), &R.bh[k]));
    }
    if (R.wh.back().O != 256 || R.bh.back().O != 1) return fail("HEAD.OUTPUT_DIM must be 256");
    c->has_roienc = true;
  }
  c->host_w.clear();
  c->finalized = true;
  return 0;
}

int sylph_preprocess(sylph_ctx* c, int B, const float* const* images, const int* hs, const int* ws, int* ph, int* pw) {
  if (!c->finalized) return fail("weights not finalized");
  if (B <= 0) return fail("empty batch");
  HIPCHK(hipSetDevice(c->device));
  int mh = 0, mw = 0;
  for (int b = 0; b < B; ++b) { mh = hs[b] > mh ? hs[b] : mh; mw = ws[b] > mw ? ws[b] : mw; }
  const int d = c->cfg.size_divisibility;
  if (d > 1) { mh = (mh + d - 1) / d * d; mw = (mw + d - 1) / d * d; }
  Plan* P = get_plan(c, B, mh, mw);
  OwnerScope own(c, P);
  BUILD(build_backbone(c, P), P);
  // the previous batch's H2D copy of the pinned descriptor table must have been consumed: wait for THAT copy only
  // (an event), not for the stream: the host stays free to enqueue the next step behind the running one
  if (P->img_desc_ev) HIPCHK(hipEventSynchronize(P->img_desc_ev));
  else HIPCHK(hipEventCreateWithFlags(&P->img_desc_ev, hipEventDisableTiming));
  for (int b = 0; b < B; ++b) {
    P->img_desc_host[b].ptr = images[b]; P->img_desc_host[b].h = hs[b]; P->img_desc_host[b].w = ws[b];
    P->img_h[b] = hs[b]; P->img_w[b] = ws[b];
  }
  HIPCHK(hipMemcpyAsync(P->img_desc_dev, P->img_desc_host, sizeof(ImageDesc) * B, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(P->img_desc_ev, c->stream));
  const char* fp = getenv("SYLPH_FUSE_PREPROCESS");  // read per call (tests compare the two paths in one process)
  P->raw_input = (!fp || atoi(fp) != 0) && P->stem_takes_raw;
  if (!P->raw_input)
    KCHK(launch_preprocess(c->dt, P->img_desc_dev, P->x0, B, mh, mw, c->cfg.pixel_mean, c->cfg.pixel_std, c->stream),
         "preprocess");
  c->cur = P;
  if (ph) *ph = mh;
  if (pw) *pw = mw;
  return 0;
}

int sylph_preprocess_u8(sylph_ctx* c, int B, const unsigned char* const* images, const int* hs, const int* ws, const int* nhs,
                        const int* nws, int rgb_input, int* ph, int* pw) {
  if (!c->finalized) return fail("weights not finalized");
  if (B <= 0) return fail("empty batch");
  HIPCHK(hipSetDevice(c->device));
  int mh = 0, mw = 0;
  for (int b = 0; b < B; ++b) {
    if (hs[b] <= 0 || ws[b] <= 0 || nhs[b] <= 0 || nws[b] <= 0) return fail("sylph_preprocess_u8: bad image size");
    mh = nhs[b] > mh ? nhs[b] : mh; mw = nws[b] > mw ? nws[b] : mw;
  }
  const int d = c->cfg.size_divisibility;
  if (d > 1) { mh = (mh + d - 1) / d * d; mw = (mw + d - 1) / d * d; }
  Plan* P = get_plan(c, B, mh, mw);
  OwnerScope own(c, P);
  BUILD(build_backbone(c, P), P);
  // resampling tables of every image (cached per (in, out) size pair), laid out back to back
  std::vector<std::shared_ptr<PilCoeffs>> hc((size_t)B), vc((size_t)B);
  size_t nint = 0;
  for (int b = 0; b < B; ++b) {
    for (int pass = 0; pass < 2; ++pass) {
      const std::pair<int, int> key = pass == 0 ? std::make_pair(ws[b], nws[b]) : std::make_pair(hs[b], nhs[b]);
      auto it = c->pil_cache.find(key);
      if (it == c->pil_cache.end()) {
        if (c->pil_cache.size() > 256) c->pil_cache.clear();
        it = c->pil_cache.emplace(key, pil_bilinear_coeffs(key.first, key.second)).first;
      }
      (pass == 0 ? hc : vc)[b] = it->second;
      nint += it->second->bounds.size() + it->second->kk.size();
    }
  }
  if (P->img_desc_ev) HIPCHK(hipEventSynchronize(P->img_desc_ev));
  else HIPCHK(hipEventCreateWithFlags(&P->img_desc_ev, hipEventDisableTiming));
  const size_t need = sizeof(ResizeDesc) * B + nint * sizeof(int);
  if (need > P->rz_tab_cap) {
    if (P->rz_host) (void)hipHostFree(P->rz_host);
    if (P->rz_desc_dev) c->dfree(P->rz_desc_dev);
    P->rz_host = nullptr; P->rz_desc_dev = nullptr; P->rz_tab_cap = 0;
    const size_t cap = need + need / 2;
    HIPCHK(hipHostMalloc((void**)&P->rz_host, cap));
    RET(c->dalloc((void**)&P->rz_desc_dev, cap));
    P->rz_tab_cap = cap;
  }
  ResizeDesc* dh = reinterpret_cast<ResizeDesc*>(P->rz_host);
  int* th = reinterpret_cast<int*>(P->rz_host + sizeof(ResizeDesc) * B);
  size_t off = 0;
  for (int b = 0; b < B; ++b) {
    ResizeDesc& r = dh[b];
    r.src = images[b]; r.h = hs[b]; r.w = ws[b]; r.new_h = nhs[b]; r.new_w = nws[b];
    r.ksh = hc[b]->ksize; r.ksv = vc[b]->ksize;
    r.hb_off = (int)off; memcpy(th + off, hc[b]->bounds.data(), hc[b]->bounds.size() * sizeof(int)); off += hc[b]->bounds.size();
    r.hk_off = (int)off; memcpy(th + off, hc[b]->kk.data(), hc[b]->kk.size() * sizeof(int)); off += hc[b]->kk.size();
    r.vb_off = (int)off; memcpy(th + off, vc[b]->bounds.data(), vc[b]->bounds.size() * sizeof(int)); off += vc[b]->bounds.size();
    r.vk_off = (int)off; memcpy(th + off, vc[b]->kk.data(), vc[b]->kk.size() * sizeof(int)); off += vc[b]->kk.size();
    P->img_h[b] = nhs[b]; P->img_w[b] = nws[b];
  }
  HIPCHK(hipMemcpyAsync(P->rz_desc_dev, P->rz_host, need, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(P->img_desc_ev, c->stream));
  const int* tab_dev = reinterpret_cast<const int*>(reinterpret_cast<const char*>(P->rz_desc_dev) + sizeof(ResizeDesc) * B);
  P->raw_input = false;  // this pipeline writes the normalised batch itself
  KCHK(launch_resize_preprocess(c->dt, P->rz_desc_dev, tab_dev, P->x0, B, mh, mw, c->cfg.pixel_mean, c->cfg.pixel_std, rgb_input,
                                c->stream), "resize_preprocess");
  c->cur = P;
  if (ph) *ph = mh;
  if (pw) *pw = mw;
  return 0;
}

int sylph_export_input(sylph_ctx* c, float* out) {
  Plan* P = c->cur;
  if (!P || !P->x0) return fail("sylph_preprocess must be called first");
  if (P->raw_input)  // the normalisation is fused into the stem kernel: x0 has not been written for this batch (raw_input stays set)
    KCHK(launch_preprocess(c->dt, P->img_desc_dev, P->x0, P->B, P->H, P->W, c->cfg.pixel_mean, c->cfg.pixel_std, c->stream), "preprocess");
  KCHK(launch_export_input(c->dt, P->x0, out, P->B, P->H, P->W, c->stream), "export_input");
  return 0;
}

int sylph_backbone_fpn(sylph_ctx* c) {
  if (!c->cur || !c->cur->backbone_built) return fail("sylph_preprocess must be called first");
  return run_ops(c, c->cur->backbone_ops, "backbone_fpn");
}

int sylph_import_pyramid(sylph_ctx* c, int B, int H, int W, const int* hs, const int* ws, const float* const* levels) {
  if (!c->finalized) return fail("weights not finalized");
  HIPCHK(hipSetDevice(c->device));
  Plan* P = get_plan(c, B, H, W);
  OwnerScope own(c, P);
  BUILD(ensure_pyramid(c, P), P);
  for (int b = 0; b < B; ++b) { P->img_h[b] = hs ? hs[b] : H; P->img_w[b] = ws ? ws[b] : W; }
  for (int l = 0; l < c->cfg.nlevels; ++l) {
    const int hw = P->hl[l] * P->wl[l];
    for (int b = 0; b < B; ++b)
      KCHK(launch_import_nchw(c->dt, levels[l] + (size_t)b * 256 * hw, P->F, 256, hw, b * P->Ltot + P->off[l], 256,
                              c->stream),
           "import_pyramid");
  }
  c->cur = P;
  return 0;
}

int sylph_export_pyramid(sylph_ctx* c, int level, float* out) {
  Plan* P = c->cur;
  if (!P || !P->F) return fail("no current batch");
  if (level < 0 || level >= c->cfg.nlevels) return fail("bad level");
  const int hw = P->hl[level] * P->wl[level];
  for (int b = 0; b < P->B; ++b)
    KCHK(launch_export_nchw(c->dt, P->F, out + (size_t)b * 256 * hw, 256, hw, b * P->Ltot + P->off[level], 256,
                            c->stream),
         "export_pyramid");
  return 0;
}

// logits / packed-code buffers of the current batch for N classes (grown on demand; the previous buffers are released)
static int ensure_logits(sylph_ctx* c, Plan* P, int N, bool allow_narrow = false) {
  const size_t rows = (size_t)P->B * P->Ltot;
  const int bn = N >= 128 ? 128 : (N > 32 ? 64 : 32);
  const int Npad = (N + bn - 1) / bn * bn;
  if (Npad > P->logits_cap_ld) {
    if (P->logits) c->dfree(P->logits);
    P->logits = nullptr; P->logits_cap_ld = 0;
    RET(c->dalloc((void**)&P->logits, rows * Npad * sizeof(float)));
    P->logits_cap_ld = Npad;
  }
  if (Npad > P->code_w_cap) {
    if (P->code_w) c->dfree(P->code_w);
    P->code_w = nullptr; P->code_w_cap = 0;
    RET(c->dalloc(&P->code_w, (size_t)Npad * 256 * c->esz()));
    if (P->code_wf) c->dfree(P->code_wf);
    P->code_wf = nullptr;
    RET(c->dalloc(&P->code_wf, (size_t)Npad * 256 * c->esz()));
    P->code_w_cap = Npad;
  }
  // row pitch of the logits: the padded class count, except for <= 8 classes on the fused GroupNorm + class-conditional conv path
  // (gn_logits_kernel stores any multiple of 4 columns): 8 floats per location instead of 32 -- the conv writes and the scan reads
  // a quarter of the bytes (a 5-way episode: 46 MB instead of 183 MB per 64 images)
  const bool narrow = allow_narrow && N <= 8 && c->dt == DT_BF16 && P->head_built && P->cls_coef && P->cls_ld == 256;
  P->logits_ld = narrow ? 8 : Npad;
  P->ncls = N;
  return 0;
}

int sylph_import_head(sylph_ctx* c, int N, int level, const float* logits, const float* reg, const float* ctr, const float* iou) {
  Plan* P = c->cur;
  if (!P) return fail("no current batch");
  if (N <= 0) return fail("class_code is empty");
  if (level < 0 || level >= c->cfg.nlevels) return fail("bad level");
  OwnerScope own(c, P);
  BUILD(build_head(c, P), P);
  if (!P->logits || N != P->ncls) RET(ensure_logits(c, P, N));
  P->scan_fused = false; P->logits_stale = false;
  const int hw = P->hl[level] * P->wl[level];
  for (int b = 0; b < P->B; ++b) {
    const int row0 = b * P->Ltot + P->off[level];
    if (logits) KCHK(launch_import_nchw(DT_F32, logits + (size_t)b * N * hw, P->logits, N, hw, row0, P->logits_ld, c->stream), "import logits");
    if (reg) KCHK(launch_import_nchw(DT_F32, reg + (size_t)b * 4 * hw, P->pred, 4, hw, row0, 8, c->stream), "import reg");
    if (ctr) KCHK(launch_import_nchw(DT_F32, ctr + (size_t)b * hw, P->pred + 4, 1, hw, row0, 8, c->stream), "import ctr");
    if (iou) KCHK(launch_import_nchw(DT_F32, iou + (size_t)b * hw, P->pred + 5, 1, hw, row0, 8, c->stream), "import iou");
  }
  return 0;
}

int sylph_roi_align(sylph_ctx* c, const float* boxes, float* out) {
  Plan* P = c->cur;
  if (!P || !P->F) return fail("no current batch");
  if (!boxes || !out) return fail("NULL argument");
  HIPCHK(hipSetDevice(c->device));
  const int S = P->B, L = c->cfg.nlevels;
  std::vector<LevelDesc> lv;
  for (int b = 0; b < S; ++b)
    for (int l = 0; l < L; ++l)
      lv.push_back(LevelDesc{b * P->Ltot + P->off[l], P->hl[l], P->wl[l], 1.0f / (float)c->cfg.strides[l]});
  sylph_ctx tmp;  // scratch allocations freed on return
  tmp.device = c->device; tmp.dt = c->dt; tmp.stream = c->stream; tmp.zeros = c->zeros;
  struct Guard { sylph_ctx* t; hipStream_t s; ~Guard() { (void)hipStreamSynchronize(s); for (void* p : t->allocs) (void)hipFree(p); } } guard{&tmp, c->stream};
  LevelDesc* lvd = nullptr;
  void* roi = nullptr;
  RET(upload(&tmp, (void**)&lvd, lv.data(), lv.size() * sizeof(LevelDesc)));
  RET(tmp.dalloc(&roi, (size_t)S * 49 * 256 * c->esz()));
  KCHK(launch_roi_align(c->dt, P->F, 256, lvd, L, boxes, S, 7, roi, c->stream), "roi_align");
  for (int s = 0; s < S; ++s)
    KCHK(launch_export_nchw(c->dt, roi, out + (size_t)s * 256 * 49, 256, 49, s * 49, 256, c->stream), "export roi");
  return 0;
}

// the class-conditional conv as its own launch(es): logits[rows][Npad] fp32 from the cls tower output, the packed codes and
// P->bias_pad (sylph_fcos_head; sylph_export_head after a fused many-way head)
static int run_cond_logits(sylph_ctx* c, Plan* P) {
  const int N = P->ncls, Npad = P->logits_ld;
  const int bn = N >= 128 ? 128 : (N > 32 ? 64 : 32);
  const size_t rows = (size_t)P->B * P->Ltot;
  const float* bias = P->has_bias ? P->bias_pad : nullptr;
  if (P->cls_coef) {
    if (bn == 32 && P->cls_ld == 256) {  // GroupNorm + ReLU + class-conditional conv in one HBM pass (head_fused.hip)
      const Plan* PP = P;
      KCHK(timed_op(c, "gn_logits_kernel", 2.0 * (double)rows * N * 256.0, c->stream, [=](hipStream_t st) {
             return launch_gn_logits(PP->cls_feat, 256, PP->cls_coef, PP->code_w, bias, N, PP->logits, Npad, PP->head_segs, PP->head_tiles32,
                                     PP->head_mtiles32, st);
           }), "gn_logits");
      return 0;
    }
    KCHK(P->cls_apply(c->stream), "gn_apply (cls tower, last layer)");
  }
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = P->cls_feat; a.wt = P->code_w; a.out = P->logits;
  a.shift = bias;
  a.zeros = c->zeros; a.tap_dy = 1;
  a.segs = P->head_segs;
  int BM = P->head_BM;
  if (bn == 32) { BM = 128; a.tiles = P->head_tiles32; a.n_mtiles = P->head_mtiles32; }
  else { a.tiles = P->head_tiles; a.n_mtiles = P->head_mtiles; }
  a.n_ntiles = Npad / bn;
  a.Cin = 256; a.Cout = N; a.KH = 1; a.KW = 1; a.stride = 1; a.pad = 0;
  a.in_ld = P->cls_ld; a.out_ld = Npad;
  KCHK(timed_conv(c, c->dt, true, a, BM, bn, 2.0 * (double)rows * N * 256.0, c->stream), "cond_cls_logits");
  return 0;
}

int sylph_fcos_head(sylph_ctx* c, const float* cls_conv, const float* cls_bias, int N) {
  Plan* P = c->cur;
  if (!P) return fail("no current batch");
  if (N <= 0) return fail("class_code is empty");
  if (!cls_conv) return fail("cls_conv is NULL");
  OwnerScope own(c, P);
  BUILD(build_head(c, P), P);
  const size_t rows = (size_t)P->B * P->Ltot;
  const int bn = N >= 128 ? 128 : (N > 32 ? 64 : 32);
  const int Npad = (N + bn - 1) / bn * bn;
  RET(ensure_logits(c, P, N, true));
  RET(run_ops(c, P->head_ops, "fcos_head"));
  KCHK(launch_pack_codes(c->dt, cls_conv, N, 256, Npad, P->code_w, c->stream), "pack_codes");
  // the biases, zero-padded to the packed code rows (device copy: the caller's buffer need not outlive this call)
  if (Npad > P->bias_pad_cap) {
    if (P->bias_pad) c->dfree(P->bias_pad);
    P->bias_pad = nullptr; P->bias_pad_cap = 0;
    RET(c->dalloc((void**)&P->bias_pad, (size_t)2 * Npad * sizeof(float)));
    P->bias_pad_cap = Npad;
  }
  P->has_bias = c->cfg.cond_use_bias && cls_bias;
  HIPCHK(hipMemsetAsync(P->bias_pad, 0, (size_t)Npad * sizeof(float), c->stream));
  if (P->has_bias) HIPCHK(hipMemcpyAsync(P->bias_pad, cls_bias, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  P->scan_fused = false; P->logits_stale = false;
  // Many-way episodes (bf16): conv + scan in one pass, the logits never reach HBM (detect.hip: logits_scan_kernel)
  static const int fuse_scan_on = getenv("SYLPH_FUSE_SCAN") ? atoi(getenv("SYLPH_FUSE_SCAN")) : 1;
  if (fuse_scan_on && c->dt == DT_BF16 && P->cls_coef && (bn != 32 || fuse_scan_on == 2) && P->cls_ld == 256 && N < 65536) {
    BUILD(build_decode(c, P), P);
    RET(ensure_cand_cap(c, P));
    const DecodeCfg d = decode_cfg(c, P, 0);
    float* bias_scan = P->bias_pad + P->bias_pad_cap;
    HIPCHK(hipMemcpyAsync(bias_scan, P->bias_pad, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    if (Npad > N) HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(bias_scan + N), (int)0xff800000u, (size_t)(Npad - N), c->stream));
    const Plan* PP = P;
    const int nseg = P->B * c->cfg.nlevels;
    KCHK(timed_op(c, "logits_scan_kernel", 2.0 * (double)rows * N * 256.0, c->stream, [=](hipStream_t st) {
           return launch_logits_scan(PP->cls_feat, 256, PP->cls_coef, PP->code_w, PP->code_wf, bias_scan, PP->head_segs, PP->head_tiles32,
                                     PP->head_mtiles32, PP->pred, 8, d, PP->dbuf, nseg, st);
         }), "logits_scan");
    P->scan_fused = true; P->logits_stale = true;
    return 0;
  }
  return run_cond_logits(c, P);
}

int sylph_fcos_head_pretrained(sylph_ctx* c, int* num_classes) {
  Plan* P = c->cur;
  if (!P) return fail("no current batch");
  if (!c->has_cls_logits) return fail("the checkpoint has no proposal_generator.fcos_head.cls_logits (1x1 or 3x3, 256 input channels)");
  OwnerScope own(c, P);
  BUILD(build_head(c, P), P);
  const int N = c->cls_logits.Cout;
  RET(ensure_logits(c, P, N));
  if (c->cls_logits.Cout_pad != P->logits_ld) return fail("internal: cls_logits padding");
  if (P->cls_logits_dst != P->logits) {  // (re)build the conv launch for this plan's buffers
    P->cls_logits_ops.clear();
    ConvOpts o; o.pad = c->cls_logits.KH / 2; o.out_f32 = true;
    RET(add_conv(c, P->cls_logits_ops, c->cls_logits, P->cls_feat, P->cls_ld, P->logits, P->logits_ld, pyramid_segs(c, P), o));
    P->cls_logits_dst = P->logits;
  }
  P->scan_fused = false; P->logits_stale = false;
  RET(run_ops(c, P->head_ops, "fcos_head"));
  if (P->cls_coef) KCHK(P->cls_apply(c->stream), "gn_apply (cls tower, last layer)");
  RET(run_ops(c, P->cls_logits_ops, "cls_logits"));
  if (num_classes) *num_classes = N;
  return 0;
}

int sylph_export_head(sylph_ctx* c, int level, float* logits, float* reg, float* ctr, float* iou) {
  Plan* P = c->cur;
  if (!P || !P->head_built || !P->logits) return fail("sylph_fcos_head must be called first");
  if (level < 0 || level >= c->cfg.nlevels) return fail("bad level");
  if (logits && P->logits_stale) {  // fused many-way head: the logits were never written
    OwnerScope own(c, P);
    RET(run_cond_logits(c, P));
    P->logits_stale = false;
  }
  const int hw = P->hl[level] * P->wl[level];
  for (int b = 0; b < P->B; ++b) {
    const int row0 = b * P->Ltot + P->off[level];
    if (logits)
      KCHK(launch_export_nchw_f32(P->logits, logits + (size_t)b * P->ncls * hw, P->ncls, hw, row0, P->logits_ld, 0,
                                  c->stream), "export logits");
    if (reg) KCHK(launch_export_nchw_f32(P->pred, reg + (size_t)b * 4 * hw, 4, hw, row0, 8, 0, c->stream), "export reg");
    if (ctr) KCHK(launch_export_nchw_f32(P->pred, ctr + (size_t)b * hw, 1, hw, row0, 8, 4, c->stream), "export ctr");
    if (iou) KCHK(launch_export_nchw_f32(P->pred, iou + (size_t)b * hw, 1, hw, row0, 8, 5, c->stream), "export iou");
  }
  return 0;
}

int sylph_decode_nms(sylph_ctx* c, const int* oh, const int* ow, int max_out, float* boxes, float* scores,
                     int* classes, int* levels, float* locations, int* cand, int* counts, int* status) {
  Plan* P = c->cur;
  if (!P || !P->head_built || !P->logits) return fail("sylph_fcos_head must be called first");
  if (max_out <= 0) return fail("max_out must be positive");
  OwnerScope own(c, P);
  BUILD(build_decode(c, P), P);
  RET(ensure_cand_cap(c, P));
  // img_out_host is rewritten below: wait only for the previous call's H2D copy of it (long finished in steady
  // state), not for the stream: the host must stay free to launch the next batch on another stream
  if (P->img_out_ev) HIPCHK(hipEventSynchronize(P->img_out_ev));
  else HIPCHK(hipEventCreateWithFlags(&P->img_out_ev, hipEventDisableTiming));
  for (int b = 0; b < P->B; ++b) {
    const int H = oh ? oh[b] : P->img_h[b], W = ow ? ow[b] : P->img_w[b];
    // detector_postprocess: python-double ratios cast to the fp32 tensor dtype
    P->img_out_host[b].sx = (float)((double)W / (double)P->img_w[b]);
    P->img_out_host[b].sy = (float)((double)H / (double)P->img_h[b]);
    P->img_out_host[b].out_w = (float)W;
    P->img_out_host[b].out_h = (float)H;
  }
  HIPCHK(hipMemcpyAsync(P->img_out_dev, P->img_out_host, sizeof(ImageOut) * P->B, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(P->img_out_ev, c->stream));
  const DecodeCfg d = decode_cfg(c, P, max_out);
  const int L = c->cfg.nlevels;
  int nwb = (L * c->cfg.pre_nms_topk + 63) / 64;
  if (nwb > P->pool_cap / 64) nwb = P->pool_cap / 64;
  KCHK(launch_decode(d, P->dsegs, P->B * L, P->hl[0] * P->wl[0], P->B, nwb, P->logits, P->pred, 8, P->dbuf,
                     P->img_out_dev, boxes, scores, classes, levels, locations, cand, counts, P->scan_fused, c->stream),
       "decode_nms");
  if (status) HIPCHK(hipMemcpyAsync(status, P->dbuf.status, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int sylph_codegen_classes(sylph_ctx* c, const float* boxes, int shots, float* codes_out) {
  Plan* P = c->cur;
  if (!P) return fail("no current batch");
  if (!boxes || !codes_out) return fail("NULL argument");
  if (shots < 1 || P->B % shots != 0) return fail("pooled_features.shape[0] " + std::to_string(P->B) + " Vs batch_size * num_shots: the batch is not a whole number of classes");
  if (shots > 64) return fail("codegen: " + std::to_string(shots) + " shots per class in one call; the shot reduction handles at most 64 (chunk the class and reduce the chunk codes, sylph_reduce_codes)");
  OwnerScope own(c, P);
  if (c->cfg.cg_type == 1) BUILD(build_support_roienc(c, P), P);
  else BUILD(build_support(c, P), P);
  P->cur_boxes = boxes;
  P->cur_code_out = codes_out;
  P->cur_shots = shots;
  return run_ops(c, P->support_ops, "codegen");
}

int sylph_codegen_weight_norm(sylph_ctx* c, float* out) {
  Plan* P = c->cur;
  if (!P || !P->support_built || !P->cg_wnorm) return fail("no code-generator pass on the current batch");
  if (!c->cfg.cg_has_scale) return fail("CODE_GENERATOR.SCALE_LAYER is empty: there is no cls_weight_norm");
  const int ncls = P->B / (P->cur_shots > 0 ? P->cur_shots : P->B);
  HIPCHK(hipMemcpyAsync(out, P->cg_wnorm, (size_t)ncls * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int sylph_codegen(sylph_ctx* c, const float* boxes, float* code_out) {
  if (!c->cur) return fail("no current batch");
  return sylph_codegen_classes(c, boxes, c->cur->B, code_out);
}

int sylph_normalize_codes(sylph_ctx* c, float* codes, int n, const float* weight_norm) {
  if (!c->has_codegen) return fail("code generator weights were not loaded");
  if (n <= 0) return 0;
  const float prior = c->cg_bias_prior;
  KCHK(launch_normalize_codes(codes, n, 256, c->cg_post.gamma, c->cg_post.beta, c->cfg.cg_post_norm,
                              c->cfg.cg_conv_l2_norm, c->cg_conv_scale, c->cg_bias_scale, prior, weight_norm, c->stream),
       "normalize_codes");
  return 0;
}

int sylph_reduce_codes(sylph_ctx* c, const float* rows, int n, int row_ld, float* out, int num_classes, int divide_by_acc) {
  if (!rows || !out) return fail("NULL argument");
  if (row_ld < 262) return fail("sylph_reduce_codes: rows must be at least 262 floats wide");
  if (num_classes <= 0 || n < 0) return fail("sylph_reduce_codes: bad sizes");
  HIPCHK(hipSetDevice(c->device));
  KCHK(launch_reduce_codes(rows, n, row_ld, out, num_classes, divide_by_acc, c->stream), "reduce_codes");
  return 0;
}

int sylph_conv2d(sylph_ctx* c, const float* x, int B, int C, int H, int W, const float* w_host, int Cout, int KH, int KW,
                 int stride, int pad, const float* scale_host, const float* shift_host, int relu, const float* residual,
                 float* y) {
  HIPCHK(hipSetDevice(c->device));
  const int bk = c->dt == DT_BF16 ? 64 : 32;
  if (C % bk != 0) return fail("sylph_conv2d: Cin must be a multiple of " + std::to_string(bk));
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  sylph_ctx tmp;  // scratch allocations freed on return
  tmp.device = c->device; tmp.dt = c->dt; tmp.stream = c->stream; tmp.zeros = c->zeros;
  struct Guard { sylph_ctx* t; hipStream_t s; ~Guard() { (void)hipStreamSynchronize(s); for (void* p : t->allocs) (void)hipFree(p); } } guard{&tmp, c->stream};
  HostTensor hw;
  hw.shape = {Cout, C, KH, KW};
  hw.data.assign(w_host, w_host + (size_t)Cout * C * KH * KW);
  ConvLayer L;
  RET(pack_conv(&tmp, {&hw}, &L));
  if (scale_host) RET(upload_vec(&tmp, &L.scale, std::vector<float>(scale_host, scale_host + Cout), L.Cout_pad));
  if (shift_host) RET(upload_vec(&tmp, &L.shift, std::vector<float>(shift_host, shift_host + Cout), L.Cout_pad));
  void *xin, *yout, *res = nullptr;
  RET(tmp.dalloc(&xin, (size_t)B * H * W * C * tmp.esz()));
  RET(tmp.dalloc(&yout, (size_t)B * Ho * Wo * Cout * tmp.esz()));
  for (int b = 0; b < B; ++b)
    KCHK(launch_import_nchw(c->dt, x + (size_t)b * C * H * W, xin, C, H * W, b * H * W, C, c->stream), "import");
  ConvOpts o; o.stride = stride; o.pad = pad; o.relu_nch = relu ? (1 << 30) : 0;
  if (residual) {
    RET(tmp.dalloc(&res, (size_t)B * Ho * Wo * Cout * tmp.esz()));
    for (int b = 0; b < B; ++b)
      KCHK(launch_import_nchw(c->dt, residual + (size_t)b * Cout * Ho * Wo, res, Cout, Ho * Wo, b * Ho * Wo, Cout,
                              c->stream), "import");
    o.res = res; o.res_ld = Cout; o.res_mode = 1;
  }
  std::vector<OpFn> ops;
  RET(add_conv(&tmp, ops, L, xin, C, yout, Cout, image_segs(B, H, W, Ho, Wo), o));
  RET(run_ops(c, ops, "conv2d"));
  for (int b = 0; b < B; ++b)
    KCHK(launch_export_nchw(c->dt, yout, y + (size_t)b * Cout * Ho * Wo, Cout, Ho * Wo, b * Ho * Wo, Cout, c->stream),
         "export");
  return 0;
}

int sylph_set_debug_taps(sylph_ctx* c, int on) {
  c->debug_taps = on != 0;
  return 0;
}

int sylph_export_stage(sylph_ctx* c, int stage, float* out) {
  Plan* P = c->cur;
  if (!P || !P->backbone_built) return fail("no backbone pass on the current batch");
  if (stage < 2 || stage > 5 || !P->stage_out[stage - 2]) return fail("bad stage");
  HIPCHK(hipSetDevice(c->device));
  const int si = stage - 2, C = 256 << si, hw = P->stage_h[si] * P->stage_w[si];
  for (int b = 0; b < P->B; ++b)
    KCHK(launch_export_nchw(c->dt, P->stage_out[si], out + (size_t)b * C * hw, C, hw, b * hw, C, c->stream), "export stage");
  return 0;
}

int sylph_export_tower(sylph_ctx* c, int tower, int layer, int level, float* y, float* coef) {
  Plan* P = c->cur;
  if (!P || !P->head_built) return fail("no head pass on the current batch");
  if (tower < 0 || tower > 1 || layer < 0 || layer >= (int)P->tap_out[tower].size()) return fail("bad tower / layer");
  if (level < 0 || level >= c->cfg.nlevels) return fail("bad level");
  if (!c->debug_taps && layer + 1 != (int)P->tap_out[tower].size()) return fail("intermediate tower layers need sylph_set_debug_taps(1) before the first head call");
  HIPCHK(hipSetDevice(c->device));
  const int hw = P->hl[level] * P->wl[level], L = c->cfg.nlevels;
  for (int b = 0; b < P->B; ++b) {
    if (y) KCHK(launch_export_nchw(c->dt, P->tap_out[tower][layer], y + (size_t)b * 256 * hw, 256, hw, b * P->Ltot + P->off[level], 256, c->stream), "export tower");
    if (coef) {
      if (!P->tap_coef[tower][layer]) return fail("this layer's GroupNorm was applied in place (no coefficient table)");
      HIPCHK(hipMemcpyAsync(coef + (size_t)b * 512, P->tap_coef[tower][layer] + (size_t)(b * L + level) * 256, 512 * sizeof(float),
                            hipMemcpyDeviceToDevice, c->stream));
    }
  }
  return 0;
}

int sylph_bottleneck(sylph_ctx* c, const float* x, int B, int Cin, int H, int W, int stride, int mid, int cout, const float* const* w_host,
                     const float* const* scale_host, const float* const* shift_host, float* y) {
  HIPCHK(hipSetDevice(c->device));
  const int bk = c->dt == DT_BF16 ? 64 : 32;
  if (Cin % bk != 0 || mid % bk != 0) return fail("sylph_bottleneck: channel counts must be multiples of " + std::to_string(bk));
  const bool has_sc = w_host[3] != nullptr;
  if (!has_sc && (Cin != cout || stride != 1)) return fail("sylph_bottleneck: an identity block needs Cin == cout and stride 1");
  sylph_ctx tmp;  // scratch allocations freed on return
  tmp.device = c->device; tmp.dt = c->dt; tmp.stream = c->stream; tmp.zeros = c->zeros; tmp.cfg = c->cfg;
  struct Guard { sylph_ctx* t; hipStream_t s; ~Guard() { (void)hipStreamSynchronize(s); for (void* p : t->allocs) (void)hipFree(p); } } guard{&tmp, c->stream};
  sylph_ctx::Block blk;
  const int cins[4] = {Cin, mid, mid, Cin}, couts[4] = {mid, mid, cout, cout}, ks[4] = {1, 3, 1, 1};
  ConvLayer* Ls[4] = {&blk.c1, &blk.c2, &blk.c3, &blk.sc};
  HostTensor hw[4];
  for (int i = 0; i < (has_sc ? 4 : 3); ++i) {
    hw[i].shape = {couts[i], cins[i], ks[i], ks[i]};
    hw[i].data.assign(w_host[i], w_host[i] + (size_t)couts[i] * cins[i] * ks[i] * ks[i]);
    RET(pack_conv(&tmp, {&hw[i]}, Ls[i]));
    RET(upload_vec(&tmp, &Ls[i]->scale, std::vector<float>(scale_host[i], scale_host[i] + couts[i]), Ls[i]->Cout_pad));
    RET(upload_vec(&tmp, &Ls[i]->shift, std::vector<float>(shift_host[i], shift_host[i] + couts[i]), Ls[i]->Cout_pad));
  }
  blk.has_sc = has_sc;
  const char* fz = getenv("SYLPH_FUSE_SHORTCUT");
  if (has_sc && !(fz && atoi(fz) == 0)) {
    RET(make_c3sc(&tmp, hw[2], scale_host[2], shift_host[2], hw[3], scale_host[3], shift_host[3], &blk.c3sc));
    blk.fused_sc = true;
  }
  const size_t e = tmp.esz();
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  void *xin, *yout, *t1, *t2, *sc, *trash = nullptr;
  RET(tmp.dalloc(&xin, (size_t)B * H * W * Cin * e));
  RET(tmp.dalloc(&yout, (size_t)B * Ho * Wo * cout * e));
  RET(tmp.dalloc(&t1, (size_t)B * H * W * mid * e));
  RET(tmp.dalloc(&t2, (size_t)B * Ho * Wo * mid * e));
  RET(tmp.dalloc(&sc, (size_t)B * Ho * Wo * cout * e));
  for (int b = 0; b < B; ++b)
    KCHK(launch_import_nchw(c->dt, x + (size_t)b * Cin * H * W, xin, Cin, H * W, b * H * W, Cin, c->stream), "import");
  std::vector<OpFn> ops;
  BkScratch scr{t1, t2, sc, &trash};
  RET(add_bottleneck(&tmp, ops, blk, B, xin, Cin, H, W, stride, mid, cout, yout, scr));
  RET(run_ops(c, ops, "bottleneck"));
  for (int b = 0; b < B; ++b)
    KCHK(launch_export_nchw(c->dt, yout, y + (size_t)b * cout * Ho * Wo, cout, Ho * Wo, b * Ho * Wo, cout, c->stream), "export");
  return 0;
}

int sylph_fpn_lateral(sylph_ctx* c, const float* x, int B, int C, int H, int W, const float* w_host, const float* bias_host, const float* top,
                      float* y) {
  HIPCHK(hipSetDevice(c->device));
  const int bk = c->dt == DT_BF16 ? 64 : 32;
  if (C % bk != 0) return fail("sylph_fpn_lateral: Cin must be a multiple of " + std::to_string(bk));
  if (top && ((H & 1) || (W & 1))) return fail("sylph_fpn_lateral: the top-down input is half the size: H and W must be even");
  sylph_ctx tmp;  // scratch allocations freed on return
  tmp.device = c->device; tmp.dt = c->dt; tmp.stream = c->stream; tmp.zeros = c->zeros; tmp.cfg = c->cfg;
  struct Guard { sylph_ctx* t; hipStream_t s; ~Guard() { (void)hipStreamSynchronize(s); for (void* p : t->allocs) (void)hipFree(p); } } guard{&tmp, c->stream};
  HostTensor hw;
  hw.shape = {256, C, 1, 1};
  hw.data.assign(w_host, w_host + (size_t)256 * C);
  ConvLayer L;
  RET(pack_conv(&tmp, {&hw}, &L));
  RET(upload_vec(&tmp, &L.shift, std::vector<float>(bias_host, bias_host + 256), L.Cout_pad));
  const size_t e = tmp.esz();
  void *xin, *yout, *tp = nullptr;
  RET(tmp.dalloc(&xin, (size_t)B * H * W * C * e));
  RET(tmp.dalloc(&yout, (size_t)B * H * W * 256 * e));
  for (int b = 0; b < B; ++b)
    KCHK(launch_import_nchw(c->dt, x + (size_t)b * C * H * W, xin, C, H * W, b * H * W, C, c->stream), "import");
  ConvOpts o;
  std::vector<SegDesc> segs = image_segs(B, H, W, H, W);
  if (top) {  // exactly the launch build_backbone makes for fpn_lateral3 / 4: residual = nearest 2x upsample of the level above
    const int h2 = H / 2, w2 = W / 2;
    RET(tmp.dalloc(&tp, (size_t)B * h2 * w2 * 256 * e));
    for (int b = 0; b < B; ++b)
      KCHK(launch_import_nchw(c->dt, top + (size_t)b * 256 * h2 * w2, tp, 256, h2 * w2, b * h2 * w2, 256, c->stream), "import");
    o.res = tp; o.res_ld = 256; o.res_mode = 2;
    segs = image_segs(B, H, W, H, W, h2, w2);
  }
  std::vector<OpFn> ops;
  RET(add_conv(&tmp, ops, L, xin, C, yout, 256, segs, o));
  RET(run_ops(c, ops, "fpn_lateral"));
  for (int b = 0; b < B; ++b)
    KCHK(launch_export_nchw(c->dt, yout, y + (size_t)b * 256 * H * W, 256, H * W, b * H * W, 256, c->stream), "export");
  return 0;
}

int sylph_group_norm(sylph_ctx* c, const float* x, int B, int H, int W, const float* gamma_host, const float* beta_host,
                     int relu, float* y) {
  HIPCHK(hipSetDevice(c->device));
  sylph_ctx tmp;
  tmp.device = c->device; tmp.dt = c->dt; tmp.stream = c->stream; tmp.zeros = c->zeros;
  struct Guard { sylph_ctx* t; hipStream_t s; ~Guard() { (void)hipStreamSynchronize(s); for (void* p : t->allocs) (void)hipFree(p); } } guard{&tmp, c->stream};
  const int HW = H * W;
  void* buf;
  RET(tmp.dalloc(&buf, (size_t)B * HW * 256 * tmp.esz()));
  for (int b = 0; b < B; ++b)
    KCHK(launch_import_nchw(c->dt, x + (size_t)b * 256 * HW, buf, 256, HW, b * HW, 256, c->stream), "import");
  std::vector<RowSeg> rs;
  for (int b = 0; b < B; ++b) rs.push_back(RowSeg{b * HW, HW});
  RowSeg* rsd;
  RET(upload(&tmp, (void**)&rsd, rs.data(), rs.size() * sizeof(RowSeg)));
  float *ga, *be, *partial;
  float2* stats;
  RET(upload_vec(&tmp, &ga, std::vector<float>(gamma_host, gamma_host + 256), 256));
  RET(upload_vec(&tmp, &be, std::vector<float>(beta_host, beta_host + 256), 256));
  const int max_chunks = (HW + GN_ROWS_PER_CHUNK - 1) / GN_ROWS_PER_CHUNK;
  RET(tmp.dalloc((void**)&partial, (size_t)B * max_chunks * 32 * 3 * 4));
  RET(tmp.dalloc((void**)&stats, (size_t)B * 32 * sizeof(float2)));
  KCHK(launch_groupnorm(c->dt, buf, rsd, B, HW, 256, ga, be, 1e-5f, relu, partial, stats, c->stream), "group_norm");
  for (int b = 0; b < B; ++b)
    KCHK(launch_export_nchw(c->dt, buf, y + (size_t)b * 256 * HW, 256, HW, b * HW, 256, c->stream), "export");
  return 0;
}

int sylph_stem_maxpool(sylph_ctx* c, const float* x, int B, int H, int W, const float* w_host, const float* scale_host,
                       const float* shift_host, float* stem_out, float* pool_out) {
  if (c->dt != DT_BF16) return fail("sylph_stem_maxpool: the dedicated stem kernels exist in bf16 mode only");
  HIPCHK(hipSetDevice(c->device));
  sylph_ctx tmp;  // scratch allocations freed on return
  tmp.device = c->device; tmp.dt = c->dt; tmp.stream = c->stream; tmp.zeros = c->zeros;
  struct Guard { sylph_ctx* t; hipStream_t s; ~Guard() { (void)hipStreamSynchronize(s); for (void* p : t->allocs) (void)hipFree(p); } } guard{&tmp, c->stream};
  const int H2 = (H - 1) / 2 + 1, W2 = (W - 1) / 2 + 1, H4 = (H2 - 1) / 2 + 1, W4 = (W2 - 1) / 2 + 1;
  std::vector<bf16_t> wp((size_t)64 * 224);  // [n][kh][8 px][4 ch], kernel column 7 / channel 3 zero (as sylph_finalize_weights)
  for (int n = 0; n < 64; ++n)
    for (int kh = 0; kh < 7; ++kh)
      for (int px = 0; px < 8; ++px)
        for (int ch = 0; ch < 4; ++ch)
          wp[(size_t)n * 224 + kh * 32 + px * 4 + ch] = (bf16_t)((px < 7 && ch < 3) ? w_host[((n * 3 + ch) * 7 + kh) * 7 + px] : 0.f);
  void *wpd, *x0, *so, *po;
  float *scd, *shd;
  ImageDesc* idd;
  RET(upload(&tmp, &wpd, wp.data(), wp.size() * sizeof(bf16_t)));
  RET(upload_vec(&tmp, &scd, std::vector<float>(scale_host, scale_host + 64), 64));
  RET(upload_vec(&tmp, &shd, std::vector<float>(shift_host, shift_host + 64), 64));
  std::vector<ImageDesc> id((size_t)B);
  for (int b = 0; b < B; ++b) { id[b].ptr = x + (size_t)b * 3 * H * W; id[b].h = H; id[b].w = W; }
  RET(upload(&tmp, (void**)&idd, id.data(), id.size() * sizeof(ImageDesc)));
  RET(tmp.dalloc(&x0, (size_t)B * H * W * 4 * 2));
  RET(tmp.dalloc(&so, (size_t)B * H2 * W2 * 64 * 2));
  RET(tmp.dalloc(&po, (size_t)B * H4 * W4 * 64 * 2));
  const float mean0[3] = {0.f, 0.f, 0.f}, std1[3] = {1.f, 1.f, 1.f};
  KCHK(launch_preprocess(c->dt, idd, x0, B, H, W, mean0, std1, c->stream), "preprocess");
  KCHK(launch_stem_conv(x0, wpd, scd, shd, so, B, H, W, H2, W2, c->stream), "stem_conv");
  static const int fuse_pool = getenv("SYLPH_FUSE_STEM_POOL") ? atoi(getenv("SYLPH_FUSE_STEM_POOL")) : 1;
  if (fuse_pool) {  // the product path: pool_out comes from the fused kernel, stem_out from the stand-alone stem kernel
    void* trash;
    RET(tmp.dalloc(&trash, (size_t)512 * 256 * 16));
    KCHK(launch_stem_pool(x0, wpd, scd, shd, po, trash, B, H, W, H2, W2, H4, W4, c->stream), "stem_pool");
  } else {
    KCHK(launch_maxpool(c->dt, so, po, B, H2, W2, 64, H4, W4, c->stream), "maxpool");
  }
  for (int b = 0; b < B; ++b) {
    if (stem_out) KCHK(launch_export_nchw(c->dt, so, stem_out + (size_t)b * 64 * H2 * W2, 64, H2 * W2, b * H2 * W2, 64, c->stream), "export");
    if (pool_out) KCHK(launch_export_nchw(c->dt, po, pool_out + (size_t)b * 64 * H4 * W4, 64, H4 * W4, b * H4 * W4, 64, c->stream), "export");
  }
  return 0;
}

int64_t sylph_device_bytes(sylph_ctx* c) { return c->bytes; }

int sylph_bench_conv(sylph_ctx* c, int B, int H, int W, int Cin, int Cout, int K, int stride, int pad, int has_res,
                     int relu, int with_gn, int iters, float* ms_out, double* flops_out) {
  HIPCHK(hipSetDevice(c->device));
  const int bk = c->dt == DT_BF16 ? 64 : 32;
  if (Cin % bk != 0) return fail("sylph_bench_conv: Cin must be a multiple of " + std::to_string(bk));
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  sylph_ctx tmp;
  tmp.device = c->device; tmp.dt = c->dt; tmp.stream = c->stream; tmp.zeros = c->zeros;
  struct Guard { sylph_ctx* t; hipStream_t s; ~Guard() { (void)hipStreamSynchronize(s); for (void* p : t->allocs) (void)hipFree(p); } } guard{&tmp, c->stream};
  HostTensor hw;
  hw.shape = {Cout, Cin, K, K};
  hw.data.resize((size_t)Cout * Cin * K * K);
  unsigned st = 12345u;
  const float wsc = 1.0f / sqrtf((float)(Cin * K * K));
  for (auto& v : hw.data) { st = st * 1664525u + 1013904223u; v = ((float)(st >> 8) * (2.0f / 16777216.0f) - 1.0f) * wsc; }
  ConvLayer L;
  RET(pack_conv(&tmp, {&hw}, &L));
  RET(upload_vec(&tmp, &L.scale, std::vector<float>((size_t)Cout, 1.0f), L.Cout_pad));
  RET(upload_vec(&tmp, &L.shift, std::vector<float>((size_t)Cout, 0.1f), L.Cout_pad));
  void *xin, *yout, *res = nullptr;
  const size_t nin = (size_t)B * H * W * Cin, nout = (size_t)B * Ho * Wo * Cout;
  RET(tmp.dalloc(&xin, nin * tmp.esz()));
  RET(tmp.dalloc(&yout, nout * tmp.esz()));
  KCHK(launch_fill_random(c->dt, xin, nin, 1u, c->stream), "fill");
  ConvOpts o; o.stride = stride; o.pad = pad; o.relu_nch = relu ? (1 << 30) : 0;
  if (has_res) {
    RET(tmp.dalloc(&res, nout * tmp.esz()));
    KCHK(launch_fill_random(c->dt, res, nout, 2u, c->stream), "fill");
    o.res = res; o.res_ld = Cout; o.res_mode = 1;
  }
  o.want_gn = with_gn & 1;
  if (with_gn & 2) {  // fused GroupNorm + ReLU of the input (conv_hpipe.hip): random (a, b) per (image, channel)
    float2* coef;
    RET(tmp.dalloc((void**)&coef, (size_t)B * Cin * sizeof(float2)));
    KCHK(launch_fill_random(DT_F32, coef, (size_t)B * Cin * 2, 3u, c->stream), "fill");
    o.gn_coef = coef; o.gn_relu = 1;
  }
  std::vector<OpFn> ops;
  RET(add_conv(&tmp, ops, L, xin, Cin, yout, Cout, image_segs(B, H, W, Ho, Wo), o));
  for (int i = 0; i < 2; ++i) RET(run_ops(c, ops, "bench_conv"));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, c->stream));
  for (int i = 0; i < iters; ++i) RET(run_ops(c, ops, "bench_conv"));
  HIPCHK(hipEventRecord(e1, c->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (ms_out) *ms_out = ms / (float)iters;
  if (flops_out) *flops_out = 2.0 * (double)B * Ho * Wo * Cout * K * K * Cin;
  return 0;
}

int sylph_profile_enable(sylph_ctx* c, int on) {
  c->prof = on != 0;
  return 0;
}

int sylph_profile_read(sylph_ctx* c, double* conv_ms, double* conv_flops, int64_t* conv_launches) {
  HIPCHK(hipStreamSynchronize(c->stream));
  double ms = 0.0, fl = 0.0;
  for (auto& r : c->prof_recs) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, r.a, r.b));
    ms += t;
    fl += r.flops;
    c->prof_free.push_back(std::make_pair(r.a, r.b));
  }
  if (conv_ms) *conv_ms = ms;
  if (conv_flops) *conv_flops = fl;
  if (conv_launches) *conv_launches = (int64_t)c->prof_recs.size();
  c->prof_recs.clear();
  return 0;
}

int sylph_profile_read_kernels(sylph_ctx* c, int max_kernels, char* names, double* ms, double* flops, int64_t* launches, int* n_out) {
  HIPCHK(hipStreamSynchronize(c->stream));
  std::vector<std::string> order;
  std::map<std::string, std::tuple<double, double, int64_t>> acc;
  for (auto& r : c->prof_recs) {
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, r.a, r.b));
    const std::string k = r.kern ? r.kern : "?";
    if (!acc.count(k)) order.push_back(k);
    auto& e = acc[k];
    std::get<0>(e) += t; std::get<1>(e) += r.flops; std::get<2>(e) += 1;
    c->prof_free.push_back(std::make_pair(r.a, r.b));
  }
  c->prof_recs.clear();
  int n = 0;
  for (auto& k : order) {
    if (n >= max_kernels) break;
    snprintf(names + (size_t)n * 64, 64, "%s", k.c_str());
    ms[n] = std::get<0>(acc[k]); flops[n] = std::get<1>(acc[k]); launches[n] = std::get<2>(acc[k]);
    ++n;
  }
  if (n_out) *n_out = n;
  return 0;
}

}  // extern "C"
