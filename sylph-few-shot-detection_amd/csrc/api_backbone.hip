// Host side of libsylph_hip.so, unit "backbone": input pipeline + ResNet-FPN stage (sylph_preprocess*, sylph_backbone_fpn, pyramid import / export).
// No torch types, no CPU compute fallback: every stage is a HIP kernel from this directory.
#include "api_internal.h"

namespace sylph_host {

std::shared_ptr<PilCoeffs> pil_bilinear_coeffs(int in_size, int out_size) {
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

int ensure_pyramid(sylph_ctx* c, Plan* P) {
  if (!P->F) RET(c->dalloc(&P->F, (size_t)P->B * P->Ltot * 256 * c->esz()));
  return 0;
}

// One ResNet bottleneck block (detectron2 BottleneckBlock: 1x1 -> 3x3 -> 1x1, FrozenBN folded, residual / projection shortcut)
// appended to `ops`: X [B][Hin*Win][Cin] -> Y [B][Ho*Wo][cout].  t1 / t2 / sc are scratch activations of the stage.
// Shared by build_backbone and the single-block parity entry sylph_bottleneck, so both run the same kernels.

int add_bottleneck(sylph_ctx* c, std::vector<OpFn>& ops, const sylph_ctx::Block& blk, int B, const void* X, int Cin, int Hin, int Win,
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
  if ((fuse_id || fuse_pr) && (size_t)Hin * Win * 512 < ((size_t)1 << 32) && (size_t)B * Hin * Win < ((size_t)1 << 31)) {  // 32-bit byte offsets inside ONE image (64-bit image base)
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
    // (A 64-position variant with a double-buffered halo was measured in round 3: 1.56 vs 1.31 ms per launch -- 2.2 x as many tiles pay
    // the per-tile fixed costs; it left the tree in round 5, see bottleneck.hip.)
    const int bk_small = 0;
    pick_patch(Hin, Win, 128, 184, 2, &ph, &pw);
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
  // res3 conv2 (3x3, 128 -> 128, stride 1), bf16: weights in registers, LDS holds only the activation halo (conv_rw3.hip)
  static const int rw3_on = getenv("SYLPH_CONV_RW3") ? atoi(getenv("SYLPH_CONV_RW3")) : 1;
  const bool rw3 = rw3_on && dt == DT_BF16 && mid == 128 && s3 == 1 && blk.c2.Cin == 128 && blk.c2.Cout_pad == 128 && blk.c2.KH == 3 && blk.c2.KW == 3 &&
                   blk.c2.scale && blk.c2.shift && (size_t)B * H1 * W1 * 256 < ((size_t)1 << 31) &&  // (2 GiB buffer descriptors)
                   (rw3_on == 2 || (size_t)B * H1 * W1 >= (size_t)256 * 120);
  int ph = 0, pw = 0;
  if (rw3) pick_patch(H1, W1, 128, 184, 2, &ph, &pw);  // 100 x 168 -> 10 x 12 patches (halo 12 x 14 = 168 rows)
  if (rw3 && conv_rw3_patch_ok(ph, pw)) {
    BottleneckArgs ba;
    memset(&ba, 0, sizeof(ba));
    ba.x = t1; ba.y = t2;
    ba.w2 = (const __bf16*)blk.c2.w; ba.s2 = blk.c2.scale; ba.b2 = blk.c2.shift;
    std::vector<SegDesc> sg = image_segs(B, H1, W1, H1, W1);
    std::vector<BkTile> bt;
    for (size_t si2 = 0; si2 < sg.size(); ++si2)
      for (int yy = 0; yy < H1; yy += ph)
        for (int xx = 0; xx < W1; xx += pw)
          bt.push_back(BkTile{sg[si2].in_row0, H1, W1, (yy << 16) | xx, ph, pw, (65536u + pw - 1) / pw, (65536u + pw + 2 - 1) / (pw + 2)});
    void* btd = nullptr;
    RET(upload(c, &btd, bt.data(), bt.size() * sizeof(BkTile)));
    ba.bk = (const BkTile*)btd;
    ba.n_tiles = (int)bt.size();
    const double fl = 2.0 * (double)B * H1 * W1 * 128.0 * 1152.0;
    ops.push_back([=](hipStream_t s) { return timed_op(c, "conv_rw3_kernel", fl, s, [=](hipStream_t st) { return launch_conv_rw3(ba, st); }); });
  } else {
    ConvOpts o2; o2.stride = s3; o2.pad = 1; o2.relu_nch = 1 << 30;
    RET(add_conv(c, ops, blk.c2, t1, mid, t2, mid, image_segs(B, H1, W1, Ho, Wo), o2));
  }
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

int build_backbone(sylph_ctx* c, Plan* P) {
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
  // Small batches: after lateral5 the FPN is two independent chains of small launches -- {output5, P6, relu, P7} and {lateral4, output4,
  // lateral3, output3} -- so the first one runs on the context's side stream between a fork and a join (as the bbox tower does,
  // api_head.hip); large batches keep one stream.
  static const int fpn_two_on = getenv("SYLPH_HEAD_STREAMS") ? atoi(getenv("SYLPH_HEAD_STREAMS")) : 1;
  const bool fpn_two = fpn_two_on == 2 || (fpn_two_on == 1 && (size_t)B * P->Ltot <= (size_t)32 * 22400);  // (round 6: 32 images, as the head)
  if (fpn_two && !c->side_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  }
  std::vector<OpFn> side_ops;
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
    if (k == 2 && fpn_two)  // fork right behind lateral5: the side stream continues from here
      ops.push_back([c](hipStream_t s) {
        if (hipEventRecord(c->ev_fork, s) != hipSuccess || hipStreamWaitEvent(c->side_stream, c->ev_fork, 0) != hipSuccess) return -101;
        return 0;
      });
    std::vector<SegDesc> so = image_segs(B, h, w, h, w);
    for (int b = 0; b < B; ++b) so[b].out_row0 = b * P->Ltot + P->off[k];
    ConvOpts oo; oo.pad = 1; oo.stream_slot = (k == 2 && fpn_two) ? 1 : 0;
    RET(add_conv(c, (k == 2 && fpn_two) ? side_ops : ops, c->fpn_out[k], lat[k], 256, P->F, 256, so, oo));
  }
  std::vector<OpFn>& top_ops = fpn_two ? side_ops : ops;  // P6 / P7 hang off output5
  for (int k = 3; k < c->cfg.nlevels && k < 5; ++k) {
    std::vector<SegDesc> sg = image_segs(B, P->hl[k - 1], P->wl[k - 1], P->hl[k], P->wl[k]);
    for (int b = 0; b < B; ++b) {
      sg[b].in_row0 = b * P->Ltot + P->off[k - 1];
      sg[b].out_row0 = b * P->Ltot + P->off[k];
    }
    ConvOpts op; op.stride = 2; op.pad = 1; op.stream_slot = fpn_two ? 1 : 0;
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
      top_ops.push_back([=](hipStream_t s) { return launch_relu_rows(dt, F, p6r, 256, csd, B, n6, s); });
      src = p6r;
    }
    RET(add_conv(c, top_ops, k == 3 ? c->p6 : c->p7, src, 256, P->F, 256, sg, op));
  }
  if (fpn_two) {
    for (const OpFn& inner : side_ops) ops.push_back([c, inner](hipStream_t) { return inner(c->side_stream); });
    ops.push_back([c](hipStream_t s) {
      if (hipEventRecord(c->ev_join, c->side_stream) != hipSuccess || hipStreamWaitEvent(s, c->ev_join, 0) != hipSuccess) return -102;
      return 0;
    });
  }
  P->backbone_built = true;
  return 0;
}

}  // namespace sylph_host

extern "C" {

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
  // image table of this call; the H2D copy is skipped when the device table already holds it (a caller that reuses its input buffers:
  // every step of a steady query stream).  Otherwise the previous batch's H2D copy of the pinned table must have been consumed: wait for
  // THAT copy only (an event), not for the stream: the host stays free to enqueue the next step behind the running one
  std::vector<ImageDesc> tab((size_t)B);
  for (int b = 0; b < B; ++b) {
    tab[b].ptr = images[b]; tab[b].h = hs[b]; tab[b].w = ws[b];
    P->img_h[b] = hs[b]; P->img_w[b] = ws[b];
  }
  if (!P->img_desc_ev || P->img_desc_kind != 1 || P->img_desc_last.size() != tab.size() ||
      memcmp(P->img_desc_last.data(), tab.data(), tab.size() * sizeof(ImageDesc)) != 0) {
    if (P->img_desc_ev) HIPCHK(hipEventSynchronize(P->img_desc_ev));
    else HIPCHK(hipEventCreateWithFlags(&P->img_desc_ev, hipEventDisableTiming));
    memcpy(P->img_desc_host, tab.data(), tab.size() * sizeof(ImageDesc));
    HIPCHK(hipMemcpyAsync(P->img_desc_dev, P->img_desc_host, sizeof(ImageDesc) * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(P->img_desc_ev, c->stream));
    P->img_desc_last = tab;
    P->img_desc_kind = 1;
  }
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

}  // extern "C"
