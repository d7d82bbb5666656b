// Launcher prototypes and small device-table structs shared by the translation units.
#pragma once
#include "common.h"

namespace sylph {

struct ImageDesc { const float* ptr; int h, w; };   // one (3,h,w) fp32 plane-major image on device
struct RowSeg { int row0, nrows; };                 // a run of rows (one GroupNorm sample)
// one uint8 HWC image of the fused resize input pipeline: source, target size, offsets of its tables in the int table buffer
struct ResizeDesc { const unsigned char* src; int h, w, new_h, new_w; int hb_off, hk_off, vb_off, vk_off, ksh, ksv; };

constexpr int GN_ROWS_PER_CHUNK = 256;
// a GroupNorm sample (= conv segment) whose statistics were left by the conv epilogue as per-M-tile partials
struct GnSeg { int row0, nrows, tile0, ntiles; };

// One (image, FPN level) slab of the head outputs, for decode.
struct DecodeSeg {
  int row0;        // first row in logits / pred buffers
  int nloc;        // H*W
  int W;           // feature width
  int stride;      // FPN stride
  int level;       // FPN level index (0..)
  int image;       // batch index
  unsigned loc_base;  // sum_{l'<l} nloc_l'; global candidate ordinal = (loc_base + loc) * N + cls
  int pad;
};

struct DecodeCfg {
  int num_classes;     // N
  int logits_ld;       // row stride of logits
  float pre_nms_thresh;
  int pre_nms_topk;
  float nms_thresh;
  int post_nms_topk;
  int thresh_with_ctr;
  int quality_mode;    // 0 ctrness, 1 iou, 2 sqrt(iou*ctrness)
  int cand_cap;        // per-(image,level) candidate capacity
  int pool_cap;        // per-image sorted pool capacity (power of two, >= levels*topk)
  int nlevels;
  int max_out;         // per-image output capacity
};

// pre-NMS top-k workspace (detect.hip)
constexpr int SEL_BINS = 4096;  // key >> 19 of a non-negative float: 8 exponent bits + the top 4 mantissa bits
constexpr int SEL_TIE = 4096;   // composites of the boundary bin kept for the exact selection
constexpr int SEL_STATE = 4;    // per segment: [0] boundary-bin candidates written, [1] boundary bin, [2] how many of it are kept, [3] its population
constexpr int SEL_WS = SEL_BINS + SEL_STATE;  // unsigned words of select workspace per segment (zeroed per call)

struct DecodeBuffers {
  // scan
  unsigned* cand_key;   // [nseg][cand_cap]   float bits of cls*quality
  unsigned* cand_idx;   // [nseg][cand_cap]   loc*N + cls
  unsigned* cand_count; // [nseg]
  // select (pre-NMS top-k)
  unsigned* sel_ws;             // [nseg][SEL_WS]   score histogram + state, zeroed per call
  unsigned long long* sel_tie;  // [nseg][SEL_TIE]  composites of the bin the k-th largest falls in
  // pool (per image)
  unsigned long long* pool_key;  // [B][pool_cap]  (sqrt-score bits << 32) | ~ordinal
  unsigned* pool_count;          // [B]
  // sorted candidates
  float* s_box;      // [B][pool_cap][4]
  float* s_score;    // [B][pool_cap]
  int* s_cls;        // [B][pool_cap]
  int* s_level;      // [B][pool_cap]
  float* s_loc;      // [B][pool_cap][2]
  unsigned* s_ord;   // [B][pool_cap]
  int* status;       // [2] [0] bit0: candidate overflow, bit1: output truncated (handed to the caller and cleared by nms_kernel); [1] nms blocks done
};

struct ImageOut { float sx, sy, out_w, out_h; };  // postprocess scale + clip box per image

// elementwise.hip
int launch_preprocess(DType dt, const ImageDesc* imgs_dev, void* out, int B, int H, int W, const float* mean,
                      const float* stdv, hipStream_t s);
int launch_resize_preprocess(DType dt, const ResizeDesc* descs_dev, const int* tab_dev, void* out, int B, int H, int W,
                             const float* mean, const float* stdv, int rgb_input, hipStream_t s);
int launch_export_input(DType dt, const void* x, float* out, int B, int H, int W, hipStream_t s);
int launch_maxpool(DType dt, const void* in, void* out, int B, int H, int W, int C, int Ho, int Wo, hipStream_t s);
int launch_groupnorm(DType dt, void* x, const RowSeg* segs_dev, int nseg, int max_rows, int ld, const float* gamma,
                     const float* beta, float eps, int relu, float* partial, float2* stats, hipStream_t s);
// stem_conv.hip
int launch_stem_conv(const void* x, const void* wp, const float* scale, const float* shift, void* out, int B, int H, int W, int H2,
                     int W2, hipStream_t s);
int launch_gn_logits(const void* x, int ld, const float2* coef, const void* w, const float* bias, int N, float* out, int out_ld,
                     const SegDesc* segs, const int2* tiles, int n_tiles, hipStream_t s);  // head_fused.hip (bf16, N <= 32)
int launch_gn_pred_taps(const void* x, int ld, const float2* coef, const void* w_taps, int cp, const float* bias, int relu_nch, int mul_nch,
                        float* planes_ws, size_t plane_rows, float* out, int out_ld, const SegDesc* segs, const int2* tiles, int n_tiles,
                        hipStream_t s);  // head_fused.hip: last bbox-tower GroupNorm + 3x3 prediction convs (bf16)
int launch_stem_pool(const void* x, const void* wp, const float* scale, const float* shift, void* out, void* trash, int B, int H, int W,
                     int H2, int W2, int H4, int W4, hipStream_t s);
constexpr int STEM_RAW_MAX_BATCH = 512;  // the image table of launch_stem_pool_raw lives in LDS
int launch_stem_pool_raw(const ImageDesc* imgs_dev, const float* mean, const float* stdv, const void* wp, const float* scale,
                         const float* shift, void* out, void* trash, int B, int H, int W, int H2, int W2, int H4, int W4, hipStream_t s);  // stem + max-pool fused (bf16); trash >= 512 x 256 x 16 B
int launch_gn_apply_partials(DType dt, void* x, int ld, int ngroups, const GnSeg* segs_dev, int nseg, int max_rows,
                             const float* partial, float2* stats_ws, const float* gamma, const float* beta, float eps, int relu,
                             hipStream_t s);
int launch_gn_finalize_coef(int ngroups, const GnSeg* segs_dev, int nseg, const float* partial, float2* stats_ws, const float* gamma,
                            const float* beta, float eps, float2* coef, hipStream_t s);
int launch_fill_random(DType dt, void* p, size_t n, unsigned seed, hipStream_t s);
struct CopySeg { int src_row0, dst_row0, nrows; };
int launch_relu_rows(DType dt, const void* src, void* dst, int ld, const CopySeg* segs_dev, int nseg, int max_rows,
                     hipStream_t s);
int launch_import_nchw(DType dt, const float* src, void* dst, int C, int HW, int row0, int ld, hipStream_t s);
int launch_export_nchw(DType dt, const void* src, float* dst, int C, int HW, int row0, int ld, hipStream_t s);
int launch_export_nchw_f32(const float* src, float* dst, int C, int HW, int row0, int ld, int ch0, hipStream_t s);
int launch_pack_codes(DType dt, const float* w, int N, int C, int Npad, void* out, const float* bias, float* bias_pad, float* bias_scan, hipStream_t s);

// detect.hip
// many-way class-conditional conv fused with the scan (detect.hip); x: raw cls-tower output, coef: its GroupNorm (a, b) per
// (segment, channel), w: packed codes [>= 32 * ceil(N/32)][256] bf16, wf_ws: as many bytes of workspace (the codes in MFMA
// fragment order), bias_scan: fp32 biases (zeros without a bias), -inf from class N up to the same row count
int launch_logits_scan(const void* x, int ld, const float2* coef, const void* w, void* wf_ws, const float* bias_scan,
                       const SegDesc* segs, const int2* tiles, int n_tiles, const float* pred, int pred_ld, const DecodeCfg& cfg,
                       const DecodeBuffers& buf, int nseg, hipStream_t s);
int launch_decode(const DecodeCfg& cfg, const DecodeSeg* segs_dev, int nseg, int max_nloc, int B, int nw_bound,
                  const float* logits, const float* pred, int pred_ld, const DecodeBuffers& buf,
                  const ImageOut* img_out_dev, float* out_boxes, float* out_scores, int* out_classes,
                  int* out_levels, float* out_locations, int* out_cand, int* out_counts, int* status_out, bool candidates_ready, hipStream_t s);

// codegen.hip
struct LevelDesc { int row0; int H, W; float scale; };  // per (image, level): rows of the feature pyramid
int launch_roi_align(DType dt, const void* feats, int ld, const LevelDesc* lv_dev, int nlevels, const float* boxes_dev,
                     int S, int out_size, void* out, hipStream_t s);
int launch_codegen_tail(const float* conv_out, int conv_ld, const float* aux_out, int aux_ld, int ib, int iw, int is, int ncls, int S, int npos,
                        int C, int bias_l2_norm, float* code_out, float* wnorm_out, hipStream_t s);
int launch_normalize_codes(float* codes, int ncodes, int C, const float* gn_gamma, const float* gn_beta, int post_norm,
                           int l2_norm, float conv_scale, float bias_scale, float bias_prior, const float* weight_norm,
                           hipStream_t s);
int launch_reduce_codes(const float* rows, int n, int ld, float* out, int num_classes, int divide_by_acc, hipStream_t s);

// roi_encoder.hip
struct MsCamWeights {  // fp32 device pointers: conv1x1 256->64, GN(32,64), conv1x1 64->256, GN(32,256); local / global
  const float *l_w1, *l_b1, *l_g1, *l_be1, *l_w2, *l_b2, *l_g2, *l_be2;
  const float *g_w1, *g_b1, *g_g1, *g_be1, *g_w2, *g_b2, *g_g2, *g_be2;
};
int launch_adaptive_context(DType dt, const void* feats, int ld, const LevelDesc* lv_dev, int nlevels, int S,
                            int out_size, float* ctx, hipStream_t s);
int launch_mscam(DType dt, const float* ctx, void* x, int S, const MsCamWeights& w, hipStream_t s);
int launch_linear(int x_is_bf16, const void* x, int ldx, int S, const float* W, const float* b, int K, int O, float* y,
                  int ldy, int relu, float add, hipStream_t s);
int launch_add_layernorm(float* x, const float* r, int S, const float* gamma, const float* beta, hipStream_t s);
int launch_mean_tokens(const float* x, int n_classes, int S, float* out, hipStream_t s);

}  // namespace sylph
