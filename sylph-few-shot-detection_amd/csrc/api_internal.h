// Internal declarations shared by the host-side translation units of libsylph_hip.so (api_*.hip): the context, the per-batch-shape
// execution plan, the packed layer records, the plan-building helpers.  Not part of the C ABI (include/sylph_hip.h).
#pragma once
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
namespace sylph_host {
int fail(const std::string& m);  // sets sylph_last_error() of this thread, returns 1
}
using sylph_host::fail;
// collective.hip
hipStream_t sylph_internal_stream(sylph_ctx* c);
int sylph_internal_fail(const std::string& m);
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
  std::vector<ConvLayer> cls_tower, box_tower, share_tower;  // share_tower: MODEL.FCOS.NUM_SHARE_CONVS layers in front of both
  std::vector<GNLayer> share_gn;
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
  // small batches: the two FCOS towers of a head pass run on two streams (fork / join events around the bbox tower, api_head.hip)
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int build_slot = 0;  // plan building: 1 while the ops being added will run on the side stream (split-K scratch slot, api_conv.hip)
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
  // split-K partial planes (api_conv.hip): ONE fp32 scratch per stream slot (0 = the context's stream, 1 = its side stream), sized to
  // the largest split layer of the plan and reused by every split layer of that stream -- stream order serialises conv -> finish ->
  // next conv.  The ops read the pointer through the slot at launch time (the buffer may grow while the plan is still being built).
  float* splitk_scratch[2] = {nullptr, nullptr};
  size_t splitk_bytes[2] = {0, 0};
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
  std::vector<ImageDesc> img_desc_last;  // what img_desc_dev holds (sylph_preprocess skips the H2D copy of an unchanged table)
  int img_desc_kind = 0;                 // 1: img_desc_dev was written by sylph_preprocess; anything else invalidates the cache
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
  bool cand_dirty = false;    // the last decode left the candidate counters non-zero (fused scan: see sylph_decode_nms)
  bool scan_fused = false;    // the candidate buffers were filled by logits_scan_kernel (many-way head): decode skips its scan
  bool logits_stale = false;  // ... and the logits buffer was not written: sylph_export_head runs the unfused conv first
  float* bias_pad = nullptr;  // fp32 class biases of the last sylph_fcos_head: [0, cap) zero-padded to the packed code rows; [cap, 2 cap) the
                              // same with -inf from class N on (logits_scan_kernel: padded classes never pass the threshold)
  int bias_pad_cap = 0;
  bool has_bias = false;
  ImageOut* img_out_dev = nullptr;
  ImageOut* img_out_host = nullptr;
  std::vector<ImageOut> img_out_last;  // what img_out_dev holds (the H2D copy is skipped when a call's scales equal it)
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

// allocations made inside the scope belong to plan P (nullptr: to the context, e.g. re-packed weights)
struct OwnerScope {
  sylph_ctx* c; Plan* prev;
  OwnerScope(sylph_ctx* c_, Plan* P) : c(c_), prev(c_->alloc_owner) { c->alloc_owner = P; }
  ~OwnerScope() { c->alloc_owner = prev; }
};

struct Geom {
  const SegDesc* segs;
  const int2* tiles;
  int n_mtiles;
  std::vector<int2> seg_tiles;  // per segment: {first tile, tile count}
  float* gn_partial = nullptr;  // per-tile GroupNorm partials written by the conv epilogue (want_gn)
};

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
  int segs_per_image = 1;  // consecutive segments that belong to one image (pyramid-wide launches: the FPN levels)
  int stream_slot = 0;     // 1: the op will run on the context's side stream (its split-K scratch must not be the main stream's)
};

struct BkScratch { void *t1, *t2, *sc; void** trash; };

namespace sylph_host {
std::shared_ptr<PilCoeffs> pil_bilinear_coeffs(int in_size, int out_size);
// api_core.hip
void free_plan(sylph_ctx* c, Plan* P);
int upload(sylph_ctx* c, void** dev, const void* host, size_t n);
void evict_plans(sylph_ctx* c, const Plan* keep);
void drop_plan(sylph_ctx* c, Plan* P);
Plan* get_plan(sylph_ctx* c, int B, int H, int W);
int run_ops(sylph_ctx* c, const std::vector<OpFn>& ops, const char* what);
// api_weights.hip
const HostTensor* find_w(sylph_ctx* c, const std::string& k);
int pack_conv(sylph_ctx* c, const std::vector<const HostTensor*>& ws, ConvLayer* L);
int upload_vec(sylph_ctx* c, float** dev, const std::vector<float>& v, int pad_to);
int make_conv_bn(sylph_ctx* c, const std::string& name, ConvLayer* L);
int make_conv_bias(sylph_ctx* c, const std::vector<std::string>& names, ConvLayer* L);
int make_c3sc(sylph_ctx* c, const HostTensor& w3, const float* s3, const float* h3, const HostTensor& ws, const float* ss, const float* hs, ConvLayer* L);
int make_gn(sylph_ctx* c, const std::string& name, GNLayer* G);
int upload_f32(sylph_ctx* c, const float** dev, const HostTensor* t, const std::string& what, size_t expect = 0);
int make_lin(sylph_ctx* c, const std::string& name, sylph_ctx::Lin* L);
int make_ln(sylph_ctx* c, const std::string& name, GNLayer* G);
bool has_prefix(sylph_ctx* c, const std::string& p);
// api_conv.hip
void level_dims(const sylph_config& cfg, int H, int W, int* hl, int* wl, int* off, int* Ltot);
int make_geom(sylph_ctx* c, const std::vector<SegDesc>& segs, int BM, Geom* g);
void pick_patch(int H, int W, int max_pos, int halo_rows, int xpad, int* ph_out, int* pw_out);
void set_patch(SegDesc* s, int ph, int pw, int xpad);
int make_geom_patch(sylph_ctx* c, std::vector<SegDesc> segs, int max_pos, int halo_rows, int xpad, bool pair, Geom* g, int group = 1);
long patch_count(const std::vector<SegDesc>& segs, int max_pos, int halo_rows, int xpad);
int timed_conv(sylph_ctx* c, DType dt, bool of32, const ConvArgs& a, int BM, int BN, double flops, hipStream_t s);
int timed_op(sylph_ctx* c, const char* kern, double flops, hipStream_t s, const std::function<int(hipStream_t)>& fn);
bool use_hpipe(sylph_ctx* c, const ConvLayer& L, const std::vector<SegDesc>& segs, const ConvOpts& o);
int add_conv(sylph_ctx* c, std::vector<OpFn>& ops, const ConvLayer& L, const void* in, int in_ld, void* out, int out_ld, const std::vector<SegDesc>& segs, const ConvOpts& o, Geom* geom_out = nullptr);
int add_conv_gn(sylph_ctx* c, std::vector<OpFn>& ops, const ConvLayer& L, const void* in, int in_ld, void* out, const std::vector<SegDesc>& segs, ConvOpts o, const GNLayer& G, int relu, const float2** coef_out = nullptr, OpFn* apply_out = nullptr);
std::vector<SegDesc> image_segs(int B, int Hin, int Win, int Hout, int Wout, int resH = 0, int resW = 0);
// api_backbone.hip
int ensure_pyramid(sylph_ctx* c, Plan* P);
int add_bottleneck(sylph_ctx* c, std::vector<OpFn>& ops, const sylph_ctx::Block& blk, int B, const void* X, int Cin, int Hin, int Win, int stride, int mid, int cout, void* Y, const BkScratch& scr);
int build_backbone(sylph_ctx* c, Plan* P);
// api_head.hip
std::vector<SegDesc> pyramid_segs(sylph_ctx* c, Plan* P);
int add_gn(sylph_ctx* c, Plan* P, std::vector<OpFn>& ops, void* x, const RowSeg* segs_dev, int nseg, int max_rows, const GNLayer& G, int relu);
int ensure_gn_ws(sylph_ctx* c, Plan* P, int nseg, int max_rows);
int build_head(sylph_ctx* c, Plan* P);
int want_cand_cap(const sylph_ctx* c, const Plan* P);
int build_decode(sylph_ctx* c, Plan* P);
int ensure_cand_cap(sylph_ctx* c, Plan* P);
DecodeCfg decode_cfg(const sylph_ctx* c, const Plan* P, int max_out);
int ensure_logits(sylph_ctx* c, Plan* P, int N, bool allow_narrow = false);
int run_cond_logits(sylph_ctx* c, Plan* P);
// api_codegen.hip
int build_support(sylph_ctx* c, Plan* P);
int build_support_roienc(sylph_ctx* c, Plan* P);
}  // namespace sylph_host
using namespace sylph_host;
