// Host side of libsylph_hip.so, unit "codegen": support path: ROIAlign, code generators, code normalisation / reduction (sylph_codegen*, sylph_normalize_codes, sylph_reduce_codes).
// No torch types, no CPU compute fallback: every stage is a HIP kernel from this directory.
#include "api_internal.h"

namespace sylph_host {

int build_support(sylph_ctx* c, Plan* P) {
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
    // CODE_GENERATOR.TOWER_LAYERS[i] = [norm, activation] (code_generator.py:648-688): conv3x3 + bias, GroupNorm(32) if "GN", ReLU if "ReLU"
    const bool gn = (c->cfg.cg_tower_gn_mask >> i) & 1, relu = (c->cfg.cg_tower_relu_mask >> i) & 1;
    ConvOpts o; o.pad = 1;
    if (gn) {
      RET(add_conv_gn(c, ops, c->cg_tower[i], in, 256, out, segs, o, c->cg_gn[i], relu ? 1 : 0));
    } else {
      if (relu) o.relu_nch = 1 << 30;
      RET(add_conv(c, ops, c->cg_tower[i], in, 256, out, 256, segs, o));
    }
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

int build_support_roienc(sylph_ctx* c, Plan* P) {
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

}  // namespace sylph_host

extern "C" {

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

}  // extern "C"
