// Host side of libsylph_hip.so, unit "head": FCOS towers, class-conditional conv, decode + NMS (sylph_fcos_head*, sylph_decode_nms, head import / export).
// No torch types, no CPU compute fallback: every stage is a HIP kernel from this directory.
#include "api_internal.h"

namespace sylph_host {

std::vector<SegDesc> pyramid_segs(sylph_ctx* c, Plan* P) {
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

int add_gn(sylph_ctx* c, Plan* P, std::vector<OpFn>& ops, void* x, const RowSeg* segs_dev, int nseg,
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

int ensure_gn_ws(sylph_ctx* c, Plan* P, int nseg, int max_rows) {
  if (P->gn_partial) return 0;
  const int max_chunks = (max_rows + GN_ROWS_PER_CHUNK - 1) / GN_ROWS_PER_CHUNK;
  RET(c->dalloc((void**)&P->gn_partial, (size_t)nseg * max_chunks * 32 * 3 * sizeof(float)));
  RET(c->dalloc((void**)&P->gn_stats, (size_t)nseg * 32 * sizeof(float2)));
  return 0;
}

int build_head(sylph_ctx* c, Plan* P) {
  if (P->head_built) return 0;
  if (!c->has_head) return fail("FCOS head weights were not loaded");
  c->build_slot = 0;  // (an earlier build that failed half-way may have left it set)
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
  const bool tower_gn = c->cfg.tower_norm == 0;  // MODEL.FCOS.NORM "GN"; otherwise "none": conv + bias + ReLU layers
  const void* tower_in = P->F;                    // what the cls / bbox towers read: the pyramid, or the shared tower's output
  auto tower = [&](int which, const std::vector<ConvLayer>& convs, const std::vector<GNLayer>& gns, void* b0, void* b1,
                   void** last, const float2** coef_last, OpFn* apply_last) -> int {
    const bool defer_last = coef_last != nullptr;
    const void* in = tower_in;
    void* out = b0;
    if (!tower_gn) {  // no norm layer: the ReLU is the conv epilogue's
      for (size_t i = 0; i < convs.size(); ++i) {
        ConvOpts o; o.pad = 1; o.segs_per_image = c->cfg.nlevels; o.relu_nch = 1 << 30;
        RET(add_conv(c, ops, convs[i], in, 256, out, 256, segs, o));
        if (which < 2) { P->tap_out[which].push_back(out); P->tap_coef[which].push_back(nullptr); }
        in = out;
        out = (out == b0) ? b1 : b0;
        if (c->debug_taps && i + 1 < convs.size()) RET(c->dalloc(&out, rows * 256 * e));
      }
      *last = const_cast<void*>(in);
      return 0;
    }
    // GroupNorm + ReLU of layers 0 .. n-2 are applied by the NEXT layer's conv to its input halo in LDS (conv_hpipe.hip):
    // no separate streaming pass over those tensors.  The last layer keeps its apply pass (its readers are the
    // prediction convs and the class-conditional 1x1 conv).
    static const int gn_fuse_on = getenv("SYLPH_GN_FUSE") ? atoi(getenv("SYLPH_GN_FUSE")) : 1;
    ConvOpts probe; probe.pad = 1;
    const bool fuse = gn_fuse_on && convs.size() > 1 && convs[0].Cin <= 512 && use_hpipe(c, convs[1], segs, probe);
    const float2* coef_prev = nullptr;
    for (size_t i = 0; i < convs.size(); ++i) {
      ConvOpts o; o.pad = 1; o.segs_per_image = c->cfg.nlevels;
      if (coef_prev) { o.gn_coef = coef_prev; o.gn_relu = 1; }
      const float2* coef = nullptr;
      const bool is_last = i + 1 == convs.size();
      const bool defer = (fuse && !is_last) || (is_last && defer_last);
      OpFn apply;
      RET(add_conv_gn(c, ops, convs[i], in, 256, out, segs, o, gns[i], 1, defer ? &coef : nullptr, (is_last && defer_last) ? &apply : nullptr));
      if (is_last && defer_last) { *coef_last = coef; *apply_last = apply; }
      coef_prev = coef;
      if (which < 2) { P->tap_out[which].push_back(out); P->tap_coef[which].push_back(coef); }
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
  bool box_defer = false, two_streams = false;
  size_t side_from = 0;  // two streams: ops[side_from ..] (bbox tower + prediction pass) go to the side stream
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
        ConvOpts o; o.pad = 1; o.segs_per_image = c->cfg.nlevels;
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
    const bool defer = gn_logits_on && c->dt == DT_BF16 && tower_gn;
    OpFn cls_apply;
    if (!c->share_tower.empty()) {
      // MODEL.FCOS.NUM_SHARE_CONVS (fcos.py:397,626): a shared tower in front of both; its last norm is applied in place (two readers)
      void *s0 = nullptr, *s1 = nullptr, *share_out = nullptr;
      RET(c->dalloc(&s0, rows * 256 * e));
      RET(c->dalloc(&s1, rows * 256 * e));
      RET(tower(2, c->share_tower, c->share_gn, s0, s1, &share_out, nullptr, nullptr));
      tower_in = share_out;
    }
    // Small batches (SylphPredictor and the reference's query loop run batch 1, meta_learn_evaluation.py:421-426, predictor.py:248-274):
    // a tower layer is a launch of a few hundred blocks whose K loop is latency-bound, and the two towers are independent chains of
    // four such launches -> the bbox tower (+ its prediction pass) runs on a second stream between a fork and a join event, the cls
    // tower stays on the caller's stream.  Large batches fill the chip for many rounds per launch: one stream (measured equal).
    static const int two_on = getenv("SYLPH_HEAD_STREAMS") ? atoi(getenv("SYLPH_HEAD_STREAMS")) : 1;
    // Round 6: up to 32 full-size images (was 8): the two towers' launches of one layer share the last partial round of blocks -- batch 12
    // 1 737 -> 1 806 img/s, batch 16 / 24 / 32 +1 %, 48 ... 192 equal (profiles/r6_small_batch.md)
    two_streams = two_on == 2 || (two_on == 1 && rows <= (size_t)32 * 22400);
    if (two_streams && !c->side_stream) {
      HIPCHK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    if (two_streams)
      ops.push_back([c](hipStream_t s) {
        if (hipEventRecord(c->ev_fork, s) != hipSuccess || hipStreamWaitEvent(c->side_stream, c->ev_fork, 0) != hipSuccess) return -101;
        return 0;
      });
    RET(tower(0, c->cls_tower, c->cls_gn, P->tA, P->tB, &cls_feat, (defer && !c->cls_tower.empty()) ? &P->cls_coef : nullptr, &cls_apply));
    P->cls_apply = cls_apply;
    box_defer = defer && c->pred_taps && !c->box_tower.empty();
    side_from = ops.size();
    // from here to the join the ops run on the side stream: a split-K conv among them (towers without GroupNorm, the prediction conv)
    // must take the side stream's partial-plane scratch, not the one the cls tower is using at the same time
    c->build_slot = two_streams ? 1 : 0;
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
    ConvOpts op; op.pad = 1; op.segs_per_image = c->cfg.nlevels; op.relu_nch = 4; op.mul_nch = 4; op.out_f32 = true;
    RET(add_conv(c, ops, c->pred, box_feat, feat_ld, P->pred, 8, segs, op));
  }
  c->build_slot = 0;
  if (two_streams) {
    for (size_t k = side_from; k < ops.size(); ++k) {
      const OpFn inner = ops[k];
      ops[k] = [c, inner](hipStream_t) { return inner(c->side_stream); };
    }
    ops.push_back([c](hipStream_t s) {
      if (hipEventRecord(c->ev_join, c->side_stream) != hipSuccess || hipStreamWaitEvent(s, c->ev_join, 0) != hipSuccess) return -102;
      return 0;
    });
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
int want_cand_cap(const sylph_ctx* c, const Plan* P) {
  if (c->cfg.cand_cap > 0) return c->cfg.cand_cap;
  const long all = (long)P->hl[0] * P->wl[0] * (long)(P->ncls > 0 ? P->ncls : 1);
  long w = all <= 262144 ? all : (all / 8 > 262144 ? all / 8 : 262144);
  if (w < 4096) w = 4096;
  if (w > (1L << 22)) w = 1L << 22;
  return (int)w;
}

int build_decode(sylph_ctx* c, Plan* P) {
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
  RET(c->dalloc((void**)&d.status, 8));
  // zero once: every decode leaves these zero again (nms_kernel)
  HIPCHK(hipMemsetAsync(d.cand_count, 0, (size_t)nseg * 4, c->stream));
  HIPCHK(hipMemsetAsync(d.sel_ws, 0, (size_t)nseg * SEL_WS * 4, c->stream));
  HIPCHK(hipMemsetAsync(d.pool_count, 0, (size_t)B * 4, c->stream));
  HIPCHK(hipMemsetAsync(d.status, 0, 8, c->stream));
  RET(c->dalloc((void**)&P->img_out_dev, sizeof(ImageOut) * B));
  HIPCHK(hipHostMalloc((void**)&P->img_out_host, sizeof(ImageOut) * B));
  P->decode_built = true;
  return 0;
}

// more classes than when the decode buffers were built: grow the candidate buffers
int ensure_cand_cap(sylph_ctx* c, Plan* P) {
  if (want_cand_cap(c, P) <= P->cand_cap) return 0;
  const int nseg = P->B * c->cfg.nlevels;
  P->cand_cap = want_cand_cap(c, P);
  c->dfree(P->dbuf.cand_key); c->dfree(P->dbuf.cand_idx);
  P->dbuf.cand_key = nullptr; P->dbuf.cand_idx = nullptr;
  RET(c->dalloc((void**)&P->dbuf.cand_key, (size_t)nseg * P->cand_cap * 4));
  RET(c->dalloc((void**)&P->dbuf.cand_idx, (size_t)nseg * P->cand_cap * 4));
  return 0;
}

DecodeCfg decode_cfg(const sylph_ctx* c, const Plan* P, int max_out) {
  DecodeCfg d;
  d.num_classes = P->ncls; d.logits_ld = P->logits_ld; d.pre_nms_thresh = c->cfg.pre_nms_thresh;
  d.pre_nms_topk = c->cfg.pre_nms_topk; d.nms_thresh = c->cfg.nms_thresh; d.post_nms_topk = c->cfg.post_nms_topk;
  d.thresh_with_ctr = c->cfg.thresh_with_ctr; d.quality_mode = c->cfg.quality_mode; d.cand_cap = P->cand_cap;
  d.pool_cap = P->pool_cap; d.nlevels = c->cfg.nlevels; d.max_out = max_out;
  return d;
}

// logits / packed-code buffers of the current batch for N classes (grown on demand; the previous buffers are released)
int ensure_logits(sylph_ctx* c, Plan* P, int N, bool allow_narrow) {
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

// the class-conditional conv as its own launch(es): logits[rows][Npad] fp32 from the cls tower output, the packed codes and
// P->bias_pad (sylph_fcos_head; sylph_export_head after a fused many-way head)
int run_cond_logits(sylph_ctx* c, Plan* P) {
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

}  // namespace sylph_host

extern "C" {

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
  // the biases, zero-padded to the packed code rows (device copy: the caller's buffer need not outlive this call)
  if (Npad > P->bias_pad_cap) {
    if (P->bias_pad) c->dfree(P->bias_pad);
    P->bias_pad = nullptr; P->bias_pad_cap = 0;
    RET(c->dalloc((void**)&P->bias_pad, (size_t)2 * Npad * sizeof(float)));
    P->bias_pad_cap = Npad;
  }
  P->has_bias = c->cfg.cond_use_bias && cls_bias;
  // one launch: packed codes + zero-padded biases + the -inf padded copy the fused scan reads
  // (in FRONT of the towers: it depends on the caller's codes only, and at small batches the main stream waits for the bbox tower on the
  // side stream at the end of the head ops anyway -- behind them it was 5 us of the step's serial tail)
  KCHK(launch_pack_codes(c->dt, cls_conv, N, 256, Npad, P->code_w, P->has_bias ? cls_bias : nullptr, P->bias_pad, P->bias_pad + P->bias_pad_cap, c->stream),
       "pack_codes");
  RET(run_ops(c, P->head_ops, "fcos_head"));
  P->scan_fused = false; P->logits_stale = false;
  // Many-way episodes (bf16): conv + scan in one pass, the logits never reach HBM (detect.hip: logits_scan_kernel)
  static const int fuse_scan_on = getenv("SYLPH_FUSE_SCAN") ? atoi(getenv("SYLPH_FUSE_SCAN")) : 1;
  if (fuse_scan_on && c->dt == DT_BF16 && P->cls_coef && (bn != 32 || fuse_scan_on == 2) && P->cls_ld == 256 && N < 65536) {
    BUILD(build_decode(c, P), P);
    RET(ensure_cand_cap(c, P));
    const DecodeCfg d = decode_cfg(c, P, 0);
    float* bias_scan = P->bias_pad + P->bias_pad_cap;  // written by the pack_codes launch above
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
    ConvOpts o; o.pad = c->cls_logits.KH / 2; o.segs_per_image = c->cfg.nlevels; o.out_f32 = true;
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
  // postprocess scales of this call; the H2D copy is skipped when they equal what the device table already holds (every step of a
  // steady query stream).  Otherwise img_out_host is rewritten: wait only for the previous H2D copy of it, not for the stream
  std::vector<ImageOut> io((size_t)P->B);
  for (int b = 0; b < P->B; ++b) {
    const int H = oh ? oh[b] : P->img_h[b], W = ow ? ow[b] : P->img_w[b];
    // detector_postprocess: python-double ratios cast to the fp32 tensor dtype
    io[b].sx = (float)((double)W / (double)P->img_w[b]);
    io[b].sy = (float)((double)H / (double)P->img_h[b]);
    io[b].out_w = (float)W;
    io[b].out_h = (float)H;
  }
  if (!P->img_out_ev || P->img_out_last.size() != io.size() || memcmp(P->img_out_last.data(), io.data(), io.size() * sizeof(ImageOut)) != 0) {
    if (P->img_out_ev) HIPCHK(hipEventSynchronize(P->img_out_ev));
    else HIPCHK(hipEventCreateWithFlags(&P->img_out_ev, hipEventDisableTiming));
    memcpy(P->img_out_host, io.data(), io.size() * sizeof(ImageOut));
    HIPCHK(hipMemcpyAsync(P->img_out_dev, P->img_out_host, sizeof(ImageOut) * P->B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(P->img_out_ev, c->stream));
    P->img_out_last = io;
  }
  const DecodeCfg d = decode_cfg(c, P, max_out);
  const int L = c->cfg.nlevels;
  // the candidate counters are left zero by every decode that ran its own scan; after a fused many-way step (whose launcher clears them
  // itself and whose candidates stay valid for a repeated decode) the plain scan starts from a cleared table again
  if (!P->scan_fused && P->cand_dirty) HIPCHK(hipMemsetAsync(P->dbuf.cand_count, 0, (size_t)P->B * L * 4, c->stream));
  P->cand_dirty = P->scan_fused;
  int nwb = (L * c->cfg.pre_nms_topk + 63) / 64;
  if (nwb > P->pool_cap / 64) nwb = P->pool_cap / 64;
  KCHK(launch_decode(d, P->dsegs, P->B * L, P->hl[0] * P->wl[0], P->B, nwb, P->logits, P->pred, 8, P->dbuf,
                     P->img_out_dev, boxes, scores, classes, levels, locations, cand, counts, status, P->scan_fused, c->stream),
       "decode_nms");
  return 0;
}

}  // extern "C"
