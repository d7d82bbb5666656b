// Host side of libsylph_hip.so, unit "conv": plan-building blocks shared by every stage: tile / patch geometry and the conv launch selection (add_conv).
// No torch types, no CPU compute fallback: every stage is a HIP kernel from this directory.
#include "api_internal.h"

namespace sylph_host {

// ------------------------------------------------------------------------------------------------
void level_dims(const sylph_config& cfg, int H, int W, int* hl, int* wl, int* off, int* Ltot) {
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

int make_geom(sylph_ctx* c, const std::vector<SegDesc>& segs, int BM, Geom* g) {
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
void pick_patch(int H, int W, int max_pos, int halo_rows, int xpad, int* ph_out, int* pw_out) {
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

void set_patch(SegDesc* s, int ph, int pw, int xpad) {
  s->ph = ph; s->pw = pw; s->hpitch = pw + xpad;
  s->inv_pw = (65536u + pw - 1) / pw;
  s->inv_hw2 = (65536u + s->hpitch - 1) / s->hpitch;
}

// 3x3 s1 p1 halo modes: M tiles are ph x pw patches of one segment, tile.y = (row << 16) | col.  `pair` (conv_hpipe.hip works on two
// patches per block): the patch list of every IMAGE (`group` consecutive segments) is padded to an even length with an empty patch, so a
// pair never straddles two images, and the first tile of a pair carries the pair's index INSIDE its image in tile.x >> 20 (the kernel's
// K-walk rotation key; tile.x & 0xfffff = segment): an image's results do not depend on where in the batch it sits.  Images that would
// grow by more than 1/8 (one-patch ROI maps) keep the flat pairing and the global pair index as the key.
int make_geom_patch(sylph_ctx* c, std::vector<SegDesc> segs, int max_pos, int halo_rows, int xpad, bool pair, Geom* g, int group) {
  std::vector<int2> tiles;
  static const int fixed = SYLPH_AB_ENV("SYLPH_CONV_PATCH_8X16", 0);  // A/B knob (-DSYLPH_ABLATE builds): the round-1 geometry
  if (group < 1 || segs.size() % (size_t)group != 0) group = 1;
  size_t g0 = 0;  // first tile of the current image
  bool per_image = pair && segs.size() < (1u << 20);
  for (size_t s = 0; s < segs.size(); ++s) {
    int ph, pw;
    if (fixed) { ph = max_pos / 16; pw = 16; }
    else pick_patch(segs[s].out_H, segs[s].out_W, max_pos, halo_rows, xpad, &ph, &pw);
    set_patch(&segs[s], ph, pw, xpad);
    const int t0 = (int)tiles.size();
    for (int y = 0; y < segs[s].out_H; y += ph)
      for (int x = 0; x < segs[s].out_W; x += pw) tiles.push_back(make_int2((int)s, (y << 16) | x));
    g->seg_tiles.push_back(make_int2(t0, (int)tiles.size() - t0));
    if (per_image && (s + 1) % (size_t)group == 0) {
      const size_t n = tiles.size() - g0;
      if ((n & 1) && n < 8) per_image = false;  // (every image has the same shape: decided on the first one)
      if (per_image) {
        if (n & 1) tiles.push_back(make_int2(0, 0x7fff << 16));  // origin below every map: nothing loaded, nothing stored
        for (size_t t = g0; t < tiles.size(); t += 2) tiles[t].x |= (int)((((t - g0) >> 1) & 0x7ff) << 20);
      }
      g0 = tiles.size();
    }
  }
  if (pair && (tiles.size() & 1)) tiles.push_back(make_int2(0, 0x7fff << 16));
  if (pair && !per_image)  // flat pairing (tiny maps): the rotation key is the global pair index, as in rounds 2-5
    for (size_t t = 0; t < tiles.size(); t += 2) tiles[t].x |= (int)(((t >> 1) & 0x7ff) << 20);
  g->n_mtiles = (int)tiles.size();
  void *ds = nullptr, *dtl = nullptr;
  RET(upload(c, &ds, segs.data(), segs.size() * sizeof(SegDesc)));
  RET(upload(c, &dtl, tiles.data(), tiles.size() * sizeof(int2)));
  g->segs = (const SegDesc*)ds;
  g->tiles = (const int2*)dtl;
  return 0;
}

long patch_count(const std::vector<SegDesc>& segs, int max_pos, int halo_rows, int xpad) {
  long n = 0;
  for (auto& sg : segs) {
    int ph, pw;
    pick_patch(sg.out_H, sg.out_W, max_pos, halo_rows, xpad, &ph, &pw);
    n += (long)((sg.out_H + ph - 1) / ph) * ((sg.out_W + pw - 1) / pw);
  }
  return n;
}

// launch the conv kernel, optionally bracketed by HIP events on the same stream
int timed_conv(sylph_ctx* c, DType dt, bool of32, const ConvArgs& a, int BM, int BN, double flops,
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
int timed_op(sylph_ctx* c, const char* kern, double flops, hipStream_t s, const std::function<int(hipStream_t)>& fn) {
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


// Will add_conv route this 3x3 layer to the deep-pipelined halo kernel (conv_hpipe.hip)?  MFMA-bound 3x3 stride-1 layers
// with Cout % 256 == 0 (FCOS towers, FPN outputs, res4/res5 conv2) when its rounds fill the chip: at least two rounds over
// the 256 CUs and >= 80 % of the last one used.

bool use_hpipe(sylph_ctx* c, const ConvLayer& L, const std::vector<SegDesc>& segs, const ConvOpts& o) {
  static const int hp_on = getenv("SYLPH_CONV_HPIPE") ? atoi(getenv("SYLPH_CONV_HPIPE")) : 1;
  const int cout_l = o.cout_override >= 0 ? o.cout_override : L.Cout;
  const bool k3s1 = L.KH == 3 && L.KW == 3 && o.stride == 1 && o.pad == 1 && !o.stem && !o.in2;
  if (!(hp_on && k3s1 && c->dt == DT_BF16 && !o.out_f32 && !o.res && o.res_mode == 0 && o.mul_nch == 0 && o.cout_override < 0 &&
        (o.relu_nch == 0 || o.relu_nch >= cout_l) && L.Cin % 32 == 0 && L.Cout % 256 == 0 && L.Cout == L.Cout_pad))
    return false;
  const long blocks = (patch_count(segs, 128, 256, 4) + 1) / 2 * (L.Cout / 256), rounds = (blocks + 255) / 256;
  if (hp_on == 2) return true;
  if (blocks >= 512) return blocks * 10 >= rounds * 256 * 8;
  // Two rounds with a nearly empty second one (backbone conv2 of res4 at 16 images: 280 blocks, 111 us = two block times) lose to
  // conv_igemm's halo tiles (1120 blocks of 128 x 128, 97 us; profiles/r6_small_batch.md); the pyramid-wide head launches keep the
  // kernel (their alternative adds the GroupNorm apply passes the fused halo transform saves)
  if (o.segs_per_image == 1 && blocks > 256 && blocks < 256 + 64) return false;
  // Small batches (round 4): a launch of fewer than 512 blocks is at most two rounds, i.e. it costs one or two block times (~63 us for
  // K = 2304) whatever its fill; the alternative -- conv_igemm on 64-row tiles -- walks the same K as a latency-bound chain (~54 us) and,
  // for a tower layer, adds the GroupNorm finalize + apply launches the fused halo transform makes unnecessary (~40 us at batch 1).
  // Measured at 800x1333 (tools/sweep_small_batches.py): >= 90 blocks (the FCOS towers from batch 1 on: 94 blocks each, both towers in
  // one round on two streams) 540 -> 581 img/s at batch 1, 806 -> 914 at batch 2, 1 227 -> 1 280 at batch 4; with the 70-block FPN
  // P3 output conv of batch 1 included (threshold 64) 574.
  return blocks >= 90;
}

int add_conv(sylph_ctx* c, std::vector<OpFn>& ops, const ConvLayer& L, const void* in, int in_ld, void* out,
                    int out_ld, const std::vector<SegDesc>& segs, const ConvOpts& o, Geom* geom_out) {
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
  bool pw = false, spw = false;
  int pw_bm = 0, pw_bn = 0;
  if (pw_on && !hpipe && !halo && c->dt == DT_BF16 && !o.out_f32 && L.KH == 1 && L.KW == 1 && o.pad == 0 && !o.stem && o.group_cout == 0 &&
      o.mul_nch == 0 && !o.want_gn && !o.gn_coef && o.cout_override < 0 && L.Cout == L.Cout_pad &&
      (o.relu_nch == 0 || o.relu_nch >= L.Cout) && L.Cin % 32 == 0 && L.Cin >= 128 && (!o.in2 || o.Cin2 % 32 == 0) &&
      conv_pw_tile(L.Cout, L.Cin, o.res_mode != 0, &pw_bm, &pw_bn)) {
    const int bn = pw_bn, bm = pw_bm;
    const long tiles = ((rows + bm - 1) / bm) * (L.Cout / bn);
    // 32-bit byte offsets into the activation buffers
    // (round 6: the two INPUTS are addressed per segment -- a 64-bit base per image from the tile descriptor + 32-bit offsets inside it --
    // so only one image has to fit; the residual / output side still offsets the whole tensor: 17 MB per 800 x 1333 image at most)
    long in_rows = 0, in_rows_all = 0, in2_rows = 0, res_rows = 0, out_rows = 0;
    for (auto& sg : segs) {
      in_rows = std::max(in_rows, (long)sg.in_H * sg.in_W);
      in_rows_all = std::max(in_rows_all, (long)sg.in_row0 + (long)sg.in_H * sg.in_W);  // conv_spw offsets its (narrow) input as a whole
      in2_rows = std::max(in2_rows, (long)sg.out_H * o.stride2 * sg.in2_W);
      res_rows = std::max(res_rows, (long)sg.res_row0 + (long)sg.res_H * sg.res_W);
      out_rows = std::max(out_rows, (long)sg.out_row0 + (long)sg.out_H * sg.out_W);
    }
    const bool fits = in_rows * in_ld * 2 < (1L << 32) && (!o.in2 || in2_rows * o.in2_ld * 2 < (1L << 32)) &&
                      (!o.res || res_rows * o.res_ld * 2 < (1L << 32));
    (void)out_rows;  // (the stores already use 64-bit addresses)
    // Where it pays (in-situ timeline at B = 64, profiles/r3_*): every pointwise layer without a same-geometry residual -- bottleneck
    // conv1 (-13 ... -20 %), conv3 + projection as one GEMM (-14 ... -20 %), FPN laterals incl. the top-down add (-10 ... -15 %) --
    // except the res3-shaped identity conv1 (N = 128, stride 1: already at 5 TB/s in conv_igemm).  With a residual tile to fetch the
    // two kernels are equal (res4 / res5) or conv_igemm's five co-resident blocks win (res3, K = 128): those stay there.
    // round 5: a same-geometry residual (conv3 of the identity blocks) takes the streaming variant (conv_spw.hip) on the same operands
    static const int spw_on = getenv("SYLPH_CONV_SPW") ? atoi(getenv("SYLPH_CONV_SPW")) : 1;
    // round 6: two more layers whose weights fit the registers take it -- conv3 + projection shortcut of the first res3 block (K = 128 + 256
    // from two inputs) and FPN lateral3 (K 512, top-down add): 1 127 -> 1 051 us and 820 -> 722 us at 120 images, equal at 16, slower
    // at 8 (87.8 vs 77.8 us: 1 050 M tiles over 128 block slots per N tile) -> from 2 048 M tiles on.  (The strided conv1 of the first res4
    // block, K 512 without a residual, runs on the kernel too -- SYLPH_CONV_SPW=2 -- but never faster than conv_pw: 208 vs 205 us at 120
    // images, 48 vs 40 at 16.)
    const long mtiles = (rows + bm - 1) / bm;
    const bool spw_shape = L.Cout % 256 == 0 && L.Cout <= 2048 && bm == 128 && bn == 256;
    const bool spw_r1 = o.res_mode == 1 && o.res && (L.Cin == 128 || L.Cin == 256 || L.Cin == 512) && o.stride == 1 && !o.in2 && (o.res_ld & 7) == 0 &&
                        (spw_on == 2 || mtiles >= 512);  // a block owns whole M tiles: at least two per CU
    const bool spw_r2 = o.res_mode == 2 && o.res && L.Cin == 512 && o.stride == 1 && !o.in2 && o.relu_nch == 0 && (o.res_ld & 7) == 0 &&
                        (spw_on == 2 || mtiles >= 2048);
    const bool spw_r0 = o.res_mode == 0 && !o.res && o.relu_nch > 0 &&
                        ((!o.in2 && L.Cin == 512 && o.stride == 2 && spw_on == 2) ||
                         (o.in2 && L.Cin == 384 && o.Cin2 == 256 && o.stride == 1 && (o.in2_ld & 7) == 0 && (spw_on == 2 || mtiles >= 2048)));
    spw = spw_on && spw_shape && (spw_r1 || spw_r2 || spw_r0);
    const bool pays = (o.res_mode != 1 || spw) && (L.Cout % 256 == 0 || o.stride != 1);
    pw = fits && (pw_on == 2 || spw || (tiles >= 256 && pays));
    if (!pw) spw = false;
    if (pw) { BM = bm; BN = bn; }
  }
  Geom g;
  if (hpipe) RET(make_geom_patch(c, segs, 128, 256, 4, true, &g, o.segs_per_image));
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
    if (it != c->pw_weights.end() && !spw && !it->second.first) {  // first packed for conv_spw (table only): add the stage images
      OwnerScope ctx_owned(c, nullptr);
      RET(c->dalloc(&it->second.first, (size_t)L.Cout * L.Cin * 2));
      KCHK(launch_pw_pack_weights(L.w, it->second.first, L.Cout, L.Cin, BN, c->stream), "pw_pack_weights");
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (it == c->pw_weights.end()) {
      void* wp = nullptr;
      float* tb = nullptr;
      OwnerScope ctx_owned(c, nullptr);
      RET(c->dalloc((void**)&tb, (size_t)2 * L.Cout * sizeof(float)));
      if (!spw) {  // conv_spw reads the conv_igemm layout itself (weights in registers): no stage-image copy (7 MB of dead memory)
        RET(c->dalloc(&wp, (size_t)L.Cout * L.Cin * 2));
        KCHK(launch_pw_pack_weights(L.w, wp, L.Cout, L.Cin, BN, c->stream), "pw_pack_weights");
      }
      KCHK(launch_pw_pack_table(L.scale, L.shift, tb, L.Cout, BN, c->stream), "pw_pack_table");
      HIPCHK(hipStreamSynchronize(c->stream));
      it = c->pw_weights.emplace(L.w, std::make_pair(wp, tb)).first;
    }
    if (!spw) a.wt = it->second.first;
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
  if (!hpipe && !pw) {  // conv_igemm indexes its inputs with 31-bit ELEMENT offsets from the tensor base: refuse what would overflow them
    long in_end = 0, in2_end = 0;
    for (auto& sg : segs) {
      in_end = std::max(in_end, ((long)sg.in_row0 + (long)sg.in_H * sg.in_W) * in_ld);
      if (o.in2) in2_end = std::max(in2_end, ((long)sg.in2_row0 + (long)sg.out_H * o.stride2 * sg.in2_W) * o.in2_ld);
    }
    if (in_end >= (1L << 31) || in2_end >= (1L << 31))
      return fail("batch too large for this layer's kernel: an input of " + std::to_string(std::max(in_end, in2_end)) +
                  " elements exceeds conv_igemm's 31-bit offsets (split the batch)");
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
  static const int nbuf2_on = getenv("SYLPH_CONV_NBUF2") ? atoi(getenv("SYLPH_CONV_NBUF2")) : 1;
  // 64-row tiles are what conv_pick_tile gives a launch of fewer than 1024 128-row blocks, i.e. one that cannot hide the round trip of
  // a K-slice behind co-resident blocks: those walk K through TWO LDS stages (the next slice's loads under the current MFMAs).
  // Measured (tools/sweep_small_batches.py): batch 1 574 -> 606 img/s, batch 2 908 -> 931, batch 8 1 643 -> 1 665.
  // Round 6: THREE stages (two slices in flight, 72 KiB: two blocks per CU) for launches of at most 400 tiles -- about one block per CU,
  // where nothing else hides a slice's round trip: batch 1 622 -> 652 img/s, batch 2 944 -> 962, batch 4-16 +0.3 ... 0.8 %
  // (profiles/r6_small_batch.md; thresholds 320 / 400 / 512 equal, 256 half the gain, 600 -5 % at batch 4: the launches of 400-600
  // tiles want three blocks per CU).  Four stages (96 KiB, one block per CU) add nothing on top.  SYLPH_CONV_NBUF3_MAX=0: two stages.
  static const int nbuf3_max = getenv("SYLPH_CONV_NBUF3_MAX") ? atoi(getenv("SYLPH_CONV_NBUF3_MAX")) : 400;
  if (nbuf2_on && dt == DT_BF16 && !hpipe && !halo && !pw && BM == 64)
    a.nbuf2 = (long)g.n_mtiles * (L.Cout_pad / BN) <= nbuf3_max ? 3 : 2;  // (deep-K launches of up to 640 tiles on three stages: batch 4 / 8 -3 %)
  // ---- split K (small batches: SylphPredictor / the reference's batch-1 query loop, predictor.py:248-274) ------------------------
  // A launch with fewer tiles than CUs walks its whole K range as ONE latency-bound chain per block (load slice -> wait -> MFMA, no
  // co-resident blocks to hide it) while most of the chip idles: res5 conv2 of one 800x1333 image is 96 blocks x 72 slices = 82 us
  // for 5 GFLOP.  Such launches are split along K into grid.y ranges that write fp32 partial planes; a finish pass adds the planes
  // in plane order (deterministic) and applies the epilogue.  Same rounding points as the unsplit kernel, fp32 summation order differs.
  static const int split_on = getenv("SYLPH_SPLIT_K") ? atoi(getenv("SYLPH_SPLIT_K")) : 1;
  // Round 6: with three LDS stages in the small tiles (above) the un-split K walk of a launch of ~100-400 tiles is as fast as split + finish
  // and saves the finish launch: only launches of at most 64 tiles (FPN P6 / P7, the small heads) are still split.  Rounds 4-5 split up to
  // 384 tiles (>= 32 slices) / 768 (>= 64 slices).  Batch 1 / 2 / 4 / 8: 655 / 985 / 1 333 / 1 701 -> 676 / 1 031 / 1 354 / 1 705 img/s together
  // with the 64 x 64 tiles of conv_pick_tile (profiles/r6_small_batch.md); 13 of the 17 finish launches of a batch-1 step are gone.
  static const int split_t1 = getenv("SYLPH_SPLIT_T1") ? atoi(getenv("SYLPH_SPLIT_T1")) : 64;
  static const int split_t2 = getenv("SYLPH_SPLIT_T2") ? atoi(getenv("SYLPH_SPLIT_T2")) : 64;
  const int nk_slices = L.KH * L.KW * (L.Cin / 64);
  const long tiles_all = (long)g.n_mtiles * (L.Cout_pad / BN);
  if (split_on && dt == DT_BF16 && !hpipe && !halo && !pw && !of32 && !o.in2 && !o.stem && !o.want_gn && !o.gn_coef && o.group_cout == 0 &&
      o.mul_nch == 0 && o.cout_override < 0 && L.Cout == L.Cout_pad && (o.relu_nch == 0 || o.relu_nch >= L.Cout) && o.res_mode != 2 &&
      (o.res_mode == 0 || (o.res_ld & 7) == 0) && L.Cin % 64 == 0 && ((nk_slices >= 32 && tiles_all <= split_t1) || (nk_slices >= 64 && tiles_all < split_t2) || (split_on == 2 && nk_slices >= 8))) {
    // Where it pays (B = 1 timeline, profiles/r4_timeline_B1.txt): deep K (>= 32 slices: the 3x3 convs of res4 / res5 / FPN P5..P7, the
    // 2048-channel 1x1s) on at most 1.5 tiles per CU.  Every extra launch costs ~9 us of dispatch latency at batch 1 and the fp32
    // planes are 2 x ks times the bf16 output: shallow-K or many-row layers (res3, the 1x1s of res4) lose, so they are not split.
    int ks = (int)((512 + tiles_all - 1) / tiles_all);  // two blocks per CU (rounds 4-5: 768; 512 is equal at batch 1 / 4 / 8 and +2 % at batch 2)
    if (ks > nk_slices / 4) ks = nk_slices / 4;          // at least four slices per range
    if (ks > 8) ks = 8;
    const long plane_bytes = rows * (long)L.Cout * 4;
    if (split_on != 2 && plane_bytes * ks > (24L << 20)) ks = (int)((24L << 20) / plane_bytes);  // partial planes: at most 24 MB
    if (ks >= 2) {
      // compact row numbering of the partial planes: segment after segment
      std::vector<SegDesc> ps = segs;
      std::vector<SplitSeg> ss;
      int r0 = 0, max_rows = 0;
      for (size_t i = 0; i < ps.size(); ++i) {
        const int n = ps[i].out_H * ps[i].out_W;
        ss.push_back(SplitSeg{r0, segs[i].out_row0, segs[i].res_row0, n});
        ps[i].out_row0 = r0;
        r0 += n;
        max_rows = n > max_rows ? n : max_rows;
      }
      Geom gs;
      RET(make_geom(c, ps, BM, &gs));
      const size_t plane = (size_t)r0 * L.Cout;
      // partial planes: the plan's scratch of this op's stream, grown to the largest split layer (a buffer per layer was
      // 100-200 MB per cached batch-1 shape for scratch that is live for one launch pair); no owning plan (parity entries): own buffer
      float** slot = nullptr;
      float* own = nullptr;
      if (Plan* PO = c->alloc_owner) {
        const int si = (o.stream_slot || c->build_slot) ? 1 : 0;
        const size_t need = plane * ks * sizeof(float);
        if (PO->splitk_bytes[si] < need) {
          if (PO->splitk_scratch[si]) {  // ops built so far may be in flight on it: dfree drains the stream first
            for (size_t i = 0; i < PO->allocs.size(); ++i)
              if (PO->allocs[i] == PO->splitk_scratch[si]) { PO->allocs[i] = PO->allocs.back(); PO->allocs.pop_back(); break; }
            PO->bytes -= (int64_t)PO->splitk_bytes[si];
            c->dfree(PO->splitk_scratch[si]);
            PO->splitk_scratch[si] = nullptr; PO->splitk_bytes[si] = 0;
          }
          RET(c->dalloc((void**)&PO->splitk_scratch[si], need));
          PO->splitk_bytes[si] = need;
        }
        slot = &PO->splitk_scratch[si];
      } else {
        RET(c->dalloc((void**)&own, plane * ks * sizeof(float)));
      }
      SplitSeg* ssd = nullptr;
      RET(upload(c, (void**)&ssd, ss.data(), ss.size() * sizeof(SplitSeg)));
      ConvArgs b = a;
      b.segs = gs.segs; b.tiles = gs.tiles; b.n_mtiles = gs.n_mtiles;
      b.out = nullptr; b.out_ld = L.Cout; b.res = nullptr; b.res_mode = 0; b.scale = nullptr; b.shift = nullptr; b.relu_nch = 0;
      b.ksplit = ks; b.split_stride = (long long)plane;
      if (b.nbuf2) b.nbuf2 = 2;  // ks K ranges per tile, two blocks per CU: the two-stage ring (three stages measured equal)
      const float *scl = L.scale, *shf = L.shift;
      const void* resp = o.res_mode == 1 ? o.res : nullptr;
      const int res_ld = o.res_ld, relu_nch = o.relu_nch, Cout = L.Cout, nseg = (int)ss.size();
      ops.push_back([=](hipStream_t s) {
        return timed_op(c, "conv_igemm_kernel", flops, s, [=](hipStream_t st) {
          float* partial = slot ? *slot : own;
          ConvArgs bb = b;
          bb.out = partial;
          const int rc = launch_conv(dt, true, bb, BM, BN, st);
          if (rc != 0) return rc;
          return launch_splitk_finish(partial, ks, plane, Cout, Cout, ssd, nseg, max_rows, scl, shf, resp, res_ld, relu_nch, out, out_ld, st);
        });
      });
      return 0;
    }
  }
  if (pw && spw) {
    if (!conv_spw_ok(dt, of32, a)) return fail("internal: conv_spw selected for a layer it cannot run");
    ops.push_back([a, c, flops](hipStream_t s) { return timed_op(c, "conv_spw_kernel", flops, s, [=](hipStream_t st) { return launch_conv_spw(a, st); }); });
    return 0;
  }
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
int add_conv_gn(sylph_ctx* c, std::vector<OpFn>& ops, const ConvLayer& L, const void* in, int in_ld, void* out,
                       const std::vector<SegDesc>& segs, ConvOpts o, const GNLayer& G, int relu, const float2** coef_out,
                       OpFn* apply_out) {
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

std::vector<SegDesc> image_segs(int B, int Hin, int Win, int Hout, int Wout, int resH, int resW) {
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

}  // namespace sylph_host

