// Host side of libsylph_hip.so, unit "core": context life cycle, device allocations owned by plans, the plan cache, configuration, per-launch profiling.
// No torch types, no CPU compute fallback: every stage is a HIP kernel from this directory.
#include "api_internal.h"

namespace sylph_host {

thread_local std::string g_err;
int fail(const std::string& m) {
  g_err = m;
  return 1;
}

void free_plan(sylph_ctx* c, Plan* P) {
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
int upload(sylph_ctx* c, void** dev, const void* host, size_t n) {
  RET(c->dalloc(dev, n));
  HIPCHK(hipMemcpy(*dev, host, n, hipMemcpyHostToDevice));
  return 0;
}

void evict_plans(sylph_ctx* c, const Plan* keep) {
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

void drop_plan(sylph_ctx* c, Plan* P) {  // a plan whose build failed half way: release it so that a retry starts clean
  for (auto it = c->plans.begin(); it != c->plans.end(); ++it)
    if (it->second.get() == P) { free_plan(c, P); c->plans.erase(it); return; }
}

Plan* get_plan(sylph_ctx* c, int B, int H, int W) {
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

int run_ops(sylph_ctx* c, const std::vector<OpFn>& ops, const char* what) {
  for (size_t i = 0; i < ops.size(); ++i) {
    const int r = ops[i](c->stream);
    if (r != 0) return fail(std::string(what) + ": op " + std::to_string(i) + " failed with " + std::to_string(r));
  }
  return 0;
}

// ================================================================================================

}  // namespace sylph_host

int sylph_internal_fail(const std::string& m) { return fail(m); }

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
  cfg->num_share_convs = 0; cfg->tower_norm = 0;
  cfg->cg_tower_gn_mask = 0x3fffffff; cfg->cg_tower_relu_mask = 0x3fffffff;  // every TOWER_LAYERS entry is ["GN", "ReLU"]
}

const char* sylph_last_error(void) { return g_err.c_str(); }

int sylph_ctx_create(int device_id, int dtype, sylph_ctx** out) {
  if (!out) return fail("out is NULL");
  if (dtype != SYLPH_F32 && dtype != SYLPH_BF16 && dtype != SYLPH_F32S) return fail("dtype must be SYLPH_F32, SYLPH_BF16 or SYLPH_F32S");
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (device_id < 0 || device_id >= n) return fail("no such HIP device: " + std::to_string(device_id));
  HIPCHK(hipSetDevice(device_id));
  sylph_ctx* c = new sylph_ctx();
  c->device = device_id;
  c->dt = dtype == SYLPH_BF16 ? DT_BF16 : (dtype == SYLPH_F32S ? DT_F32S : DT_F32);
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
  if (c && c->side_stream) {
    (void)hipStreamSynchronize(c->side_stream);
    (void)hipStreamDestroy(c->side_stream);
    (void)hipEventDestroy(c->ev_fork);
    (void)hipEventDestroy(c->ev_join);
    c->side_stream = nullptr;
  }
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

int64_t sylph_device_bytes(sylph_ctx* c) { return c->bytes; }

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
