// Host side of libsylph_hip.so, unit "weights": checkpoint tensors -> packed device weights (sylph_load_weight / sylph_finalize_weights).
// No torch types, no CPU compute fallback: every stage is a HIP kernel from this directory.
#include "api_internal.h"

namespace sylph_host {

const HostTensor* find_w(sylph_ctx* c, const std::string& k) {
  auto it = c->host_w.find(k);
  return it == c->host_w.end() ? nullptr : &it->second;
}

// pack (Cout,Cin,KH,KW) fp32 -> [Cout_pad][KH][KW][Cin] compute dtype; several tensors may be stacked on Cout
int pack_conv(sylph_ctx* c, const std::vector<const HostTensor*>& ws, ConvLayer* L) {
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
  } else if (c->dt == DT_F32S) {
    // split-bf16 parity mode (conv_igemm.hip MmaSplit): every 32-element K-slice of a row becomes [32 bf16 hi | 32 bf16 lo], the same
    // 128 bytes as its fp32 form; hi = bf16(w), lo = bf16(w - hi), both round-to-nearest-even
    if (K % 32 != 0) return fail("split-bf16 mode: K = KH * KW * Cin must be a multiple of 32");
    std::vector<uint16_t> h(packed.size() * 2);
    for (size_t i = 0; i < packed.size(); ++i) {
      const uint16_t hi = f2bf_host(packed[i]);
      uint32_t hb = (uint32_t)hi << 16;
      float hf;
      memcpy(&hf, &hb, 4);
      const size_t sl = i / 32, e = i % 32;
      h[sl * 64 + e] = hi;
      h[sl * 64 + 32 + e] = f2bf_host(packed[i] - hf);
    }
    RET(upload(c, &L->w, h.data(), h.size() * 2));
  } else {
    RET(upload(c, &L->w, packed.data(), packed.size() * 4));
  }
  return 0;
}

int upload_vec(sylph_ctx* c, float** dev, const std::vector<float>& v, int pad_to) {
  std::vector<float> t(v);
  t.resize((size_t)pad_to, 0.f);
  return upload(c, (void**)dev, t.data(), t.size() * 4);
}

// conv + FrozenBN folded into per-channel scale/shift (detectron2 FrozenBatchNorm2d, eps 1e-5)
int make_conv_bn(sylph_ctx* c, const std::string& name, ConvLayer* L) {
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

int make_conv_bias(sylph_ctx* c, const std::vector<std::string>& names, ConvLayer* L) {
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
int make_c3sc(sylph_ctx* c, const HostTensor& w3, const float* s3, const float* h3, const HostTensor& ws, const float* ss,
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

int make_gn(sylph_ctx* c, const std::string& name, GNLayer* G) {
  const HostTensor *g = find_w(c, name + ".weight"), *b = find_w(c, name + ".bias");
  if (!g || !b) return fail("missing weights for " + name);
  if (g->data.size() != 256) return fail("GroupNorm layers must have 256 channels: " + name);
  RET(upload_vec(c, &G->gamma, g->data, 256));
  RET(upload_vec(c, &G->beta, b->data, 256));
  return 0;
}

int upload_f32(sylph_ctx* c, const float** dev, const HostTensor* t, const std::string& what, size_t expect) {
  if (!t) return fail("missing weights for " + what);
  if (expect && t->data.size() != expect) return fail("unexpected size for " + what);
  void* d;
  RET(upload(c, &d, t->data.data(), t->data.size() * 4));
  *dev = (const float*)d;
  return 0;
}

int make_lin(sylph_ctx* c, const std::string& name, sylph_ctx::Lin* L) {
  const HostTensor *w = find_w(c, name + ".weight"), *b = find_w(c, name + ".bias");
  if (!w || !b || w->shape.size() != 2) return fail("missing weights for " + name);
  L->O = (int)w->shape[0]; L->K = (int)w->shape[1];
  RET(upload(c, (void**)&L->W, w->data.data(), w->data.size() * 4));
  RET(upload(c, (void**)&L->b, b->data.data(), b->data.size() * 4));
  return 0;
}

int make_ln(sylph_ctx* c, const std::string& name, GNLayer* G) {
  const HostTensor *g = find_w(c, name + ".weight"), *b = find_w(c, name + ".bias");
  if (!g || !b || g->data.size() != 256) return fail("missing weights for " + name);
  RET(upload_vec(c, &G->gamma, g->data, 256));
  RET(upload_vec(c, &G->beta, b->data, 256));
  return 0;
}

bool has_prefix(sylph_ctx* c, const std::string& p) {
  auto it = c->host_w.lower_bound(p);
  return it != c->host_w.end() && it->first.compare(0, p.size(), p) == 0;
}

}  // namespace sylph_host

extern "C" {

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
    c->share_tower.resize(c->cfg.num_share_convs); c->share_gn.resize(c->cfg.num_share_convs);
    // nn.Sequential indices of a tower (fcos.py:72-122): conv 3 i, norm 3 i + 1, ReLU 3 i + 2 -- without a norm layer conv 2 i, ReLU 2 i + 1
    const bool gn = c->cfg.tower_norm == 0;
    const int step = gn ? 3 : 2;
    struct { const char* name; std::vector<ConvLayer>* convs; std::vector<GNLayer>* gns; int n; } towers[3] = {
        {".cls_tower.", &c->cls_tower, &c->cls_gn, c->cfg.num_cls_convs}, {".bbox_tower.", &c->box_tower, &c->box_gn, c->cfg.num_box_convs},
        {".share_tower.", &c->share_tower, &c->share_gn, c->cfg.num_share_convs}};
    for (auto& t : towers)
      for (int i = 0; i < t.n; ++i) {
        RET(make_conv_bias(c, {hp + t.name + std::to_string(step * i)}, &(*t.convs)[i]));
        if (gn) RET(make_gn(c, hp + t.name + std::to_string(step * i + 1), &(*t.gns)[i]));
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
    int seq = 0;  // nn.Sequential index: conv, then the norm / activation modules that exist (code_generator.py:648-688)
    for (int i = 0; i < c->cfg.cg_tower_layers; ++i) {
      RET(make_conv_bias(c, {cp + ".support_set_shared_tower." + std::to_string(seq)}, &c->cg_tower[i]));
      ++seq;
      if ((c->cfg.cg_tower_gn_mask >> i) & 1) {
        RET(make_gn(c, cp + ".support_set_shared_tower." + std::to_string(seq), &c->cg_gn[i]));
        ++seq;
      }
      if ((c->cfg.cg_tower_relu_mask >> i) & 1) ++seq;
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

}  // extern "C"
