// Host side of libsylph_hip.so, unit "parity": single-kernel / single-block parity and micro-benchmark entries used by tests/ and tools/.
// No torch types, no CPU compute fallback: every stage is a HIP kernel from this directory.
#include "api_internal.h"

extern "C" {

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
  tmp.prof = c->prof;  // per-launch HIP-event timing (tools/bench_bottleneck.py): the records move to the caller's context below
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
  for (auto& r : tmp.prof_recs) c->prof_recs.push_back(r);
  tmp.prof_recs.clear();
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

}  // extern "C"
