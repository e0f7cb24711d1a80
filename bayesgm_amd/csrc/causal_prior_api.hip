// causal_prior_api.hip -- the CausalBGM sampling kernels with a per-row conditional latent prior (IdentifiableCausalBGM,
// models/causalbgm/identifiable.py:195-211 update_latent_variable_sgd, :521-555 get_log_posterior, :557-614
// metropolis_hastings_sampler): the PRIOR = 1 instantiations of causal_kernels.h, kept in their own translation unit.
// Selected by bgm_causal_set_prior (include/bgm_hip.h); the default path (causal_api.hip) is untouched.
#include <algorithm>
#include <cmath>
#include <string>

#include "bgm_host.h"
#include "prior_kernels.h"

static constexpr int PR_WAVES = 8;
#define BGM_PRIOR_VARIANTS(X) X(1, 3, 13) X(1, 3, 7) X(1, 3, 2) X(2, 1, 10) X(2, 1, 7) X(2, 1, 2)

extern "C" int bgm_causal_set_prior(bgm_handle *h, const int32_t *seg_dev, const float *tab_dev, int32_t n_segments) {
  if (!h || !h->configured) { bgm_set_error("bgm_causal_set_prior: handle not configured"); return BGM_E_STATE; }
  if ((seg_dev == nullptr) != (tab_dev == nullptr) || (seg_dev && n_segments <= 0)) { bgm_set_error("bgm_causal_set_prior: seg_dev and tab_dev go together"); return BGM_E_INVALID; }
  h->prior_seg = seg_dev; h->prior_tab = tab_dev; h->prior_segments = seg_dev ? n_segments : 0;
  return BGM_OK;
}

template <class K>
static int pr_set_lds(K kernel, int bytes) {
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return BGM_OK;
}

int bgm_causal_prior_logpost(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, float *out,
                             int grid, hipStream_t stream) {
  int rc;
  const int lds = h->meta.total * 4;
#define X(KT1_, KSL1_, NTL_)                                                                                     \
  if (h->KT1 == KT1_ && h->KSL1 == KSL1_ && h->NTL == NTL_) {                                                    \
    auto k = causal_logpost_kernel<KT1_, KSL1_, NTL_, 1, PR_WAVES, 1>;                                           \
    rc = pr_set_lds(k, lds);                                                                                     \
    if (rc) return rc;                                                                                           \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * PR_WAVES), lds, stream, (const float *)h->sblob_dev, h->meta, x, y, v, z, \
                       (long long)n, out, (const int *)h->prior_seg, h->prior_tab);                              \
    BGM_HIP_CHECK(hipGetLastError());                                                                            \
    return BGM_OK;                                                                                               \
  }
  BGM_PRIOR_VARIANTS(X)
#undef X
  bgm_set_error("conditional prior: no compiled kernel variant for this shape");
  return BGM_E_UNSUPPORTED;
}

template <int EFFECT>
static int pr_launch_mh(bgm_handle *h, const CausalMhKArgs &ka, int grid, int lds, hipStream_t stream) {
  int rc;
#define X(KT1_, KSL1_, NTL_)                                                                   \
  if (h->KT1 == KT1_ && h->KSL1 == KSL1_ && h->NTL == NTL_) {                                  \
    auto k = causal_mh_kernel<KT1_, KSL1_, NTL_, 1, PR_WAVES, EFFECT, 1>;                      \
    rc = pr_set_lds(k, lds);                                                                   \
    if (rc) return rc;                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * PR_WAVES), lds, stream, ka);                   \
    BGM_HIP_CHECK(hipGetLastError());                                                          \
    return BGM_OK;                                                                             \
  }
  BGM_PRIOR_VARIANTS(X)
#undef X
  bgm_set_error("conditional prior: no compiled MH kernel variant for this shape");
  return BGM_E_UNSUPPORTED;
}

int bgm_causal_prior_mh_launch(bgm_handle *h, const CausalMhKArgs &a, int effect, int grid, int lds, hipStream_t stream) {
  CausalMhKArgs ka = a;
  ka.seg = (const int *)h->prior_seg;
  ka.prior_tab = h->prior_tab;
  if (effect == 3) return pr_launch_mh<3>(h, ka, grid, lds, stream);         // event form of the retained phase (causal_event_api.hip)
  if (effect == BGM_EFFECT_ADRF) return pr_launch_mh<1>(h, ka, grid, lds, stream);
  if (effect == BGM_EFFECT_ITE) return pr_launch_mh<2>(h, ka, grid, lds, stream);
  return pr_launch_mh<0>(h, ka, grid, lds, stream);
}


// ---- the prior network itself (prior_kernels.h)
static int prior_net_of(const bgm_prior_config *cfg, int rows, PriorNet &n, int &lds_floats, const char *who) {
  if (!cfg || cfg->n_layers < 1 || cfg->n_layers > PRIOR_MAX_LAYERS) { bgm_set_error(std::string(who) + ": prior net needs 1.." + std::to_string(PRIOR_MAX_LAYERS) + " dense layers"); return BGM_E_INVALID; }
  n = PriorNet{};
  n.n_layers = cfg->n_layers;
  int off = 0, a = 0, wmax = 0;
  for (int l = 0; l <= cfg->n_layers; ++l) {
    if (cfg->dims[l] < 1) { bgm_set_error(std::string(who) + ": bad layer width"); return BGM_E_INVALID; }
    n.dims[l] = cfg->dims[l];
    if (l) wmax = std::max(wmax, cfg->dims[l]);
  }
  for (int l = 0; l < cfg->n_layers; ++l) {
    n.w_off[l] = off; off += n.dims[l] * n.dims[l + 1];
    n.b_off[l] = off; off += n.dims[l + 1];
    n.a_off[l] = a; a += rows * n.dims[l + 1];
  }
  n.a_off[cfg->n_layers] = a;
  n.n_params = off; n.wmax = wmax;
  lds_floats = a;
  return BGM_OK;
}

extern "C" int bgm_prior_n_params(const bgm_prior_config *cfg, int64_t *count) {
  PriorNet n; int lf;
  int rc = prior_net_of(cfg, 1, n, lf, "bgm_prior_n_params");
  if (rc) return rc;
  if (!count) { bgm_set_error("bgm_prior_n_params: NULL count"); return BGM_E_INVALID; }
  *count = n.n_params;
  return BGM_OK;
}

extern "C" int bgm_prior_table(bgm_handle *h, const bgm_prior_config *cfg, const float *theta_dev, float *table_dev, void *stream_) {
  if (!h || !h->configured || !theta_dev || !table_dev) { bgm_set_error("bgm_prior_table: bad argument"); return BGM_E_INVALID; }
  PriorNet n; int lf;
  int rc = prior_net_of(cfg, cfg ? cfg->dims[0] : 0, n, lf, "bgm_prior_table");
  if (rc) return rc;
  if (n.dims[n.n_layers] != h->q + 1) { bgm_set_error("bgm_prior_table: the prior net must end in q + 1 outputs"); return BGM_E_INVALID; }
  if ((size_t)lf * 4 > 150 * 1024) { bgm_set_error("bgm_prior_table: prior net too wide for one workgroup's LDS"); return BGM_E_UNSUPPORTED; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(prior_table_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lf * 4));
  hipLaunchKernelGGL(prior_table_kernel, dim3(1), dim3(PRIOR_THREADS), lf * 4, (hipStream_t)stream_, n, theta_dev, table_dev, h->q);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

static int prior_step_impl(bgm_handle *h, const bgm_prior_config *cfg, float *theta_dev, float *m_dev, float *v_dev, const int32_t *seg_dev,
                           float *data_z_dev, const int32_t *idx_dev, int32_t batch, int32_t batch_global, const float *dz_dev, float lr_z,
                           float lr_prior, int64_t t_z, int64_t t_prior, float *grad_dev, int apply, float *out_dev, void *stream_, const char *who) {
  if (!h || !h->configured || !theta_dev || (apply && (!m_dev || !v_dev)) || !seg_dev || !data_z_dev || !idx_dev || !dz_dev || batch < 1 ||
      batch_global < batch || t_z < 1 || (apply && t_prior < 1) || (!apply && !grad_dev)) {
    bgm_set_error(std::string(who) + ": bad argument"); return BGM_E_INVALID;
  }
  PriorNet n; int lf;
  int rc = prior_net_of(cfg, batch, n, lf, who);
  if (rc) return rc;
  const int q = h->q;
  if (n.dims[n.n_layers] != q + 1) { bgm_set_error(std::string(who) + ": the prior net must end in q + 1 outputs"); return BGM_E_INVALID; }
  const size_t bytes = ((size_t)lf + (size_t)batch * q + 2 * (size_t)batch * n.wmax + 3 * (size_t)batch) * 4;
  if (bytes > 150 * 1024) { bgm_set_error(std::string(who) + ": minibatch x prior-net widths exceed one workgroup's LDS"); return BGM_E_UNSUPPORTED; }
  auto lr_t = [](double lr, double t) { return (float)(lr * std::sqrt(1.0 - std::pow(0.99, t)) / (1.0 - std::pow(0.9, t))); };
  PriorStepArgs a{};
  a.net = n; a.theta = theta_dev; a.m = m_dev; a.v = v_dev; a.seg = seg_dev; a.data_z = data_z_dev; a.idx = idx_dev; a.dz = dz_dev;
  a.B = batch; a.q = q; a.lr_t_z = lr_t(lr_z, (double)t_z); a.lr_t_p = apply ? lr_t(lr_prior, (double)t_prior) : 0.0f;
  a.b1 = 0.9f; a.b2 = 0.99f; a.eps = 1e-7f;        // tf.keras.optimizers.Adam(lr, beta_1=0.9, beta_2=0.99), identifiable.py:88-95
  a.out = out_dev; a.inv_B = 1.0f / (float)batch_global; a.grad = grad_dev; a.apply = apply;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(prior_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  hipLaunchKernelGGL(prior_step_kernel, dim3(1), dim3(PRIOR_THREADS), bytes, (hipStream_t)stream_, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_prior_step(bgm_handle *h, const bgm_prior_config *cfg, float *theta_dev, float *m_dev, float *v_dev, const int32_t *seg_dev,
                              float *data_z_dev, const int32_t *idx_dev, int32_t batch, const float *dz_dev, float lr_z, float lr_prior,
                              int64_t t_z, int64_t t_prior, float *out_dev, void *stream_) {
  return prior_step_impl(h, cfg, theta_dev, m_dev, v_dev, seg_dev, data_z_dev, idx_dev, batch, batch, dz_dev, lr_z, lr_prior, t_z, t_prior, nullptr, 1,
                         out_dev, stream_, "bgm_prior_step");
}

extern "C" int bgm_prior_grad(bgm_handle *h, const bgm_prior_config *cfg, const float *theta_dev, const int32_t *seg_dev, float *data_z_dev,
                              const int32_t *idx_dev, int32_t batch, int32_t batch_global, const float *dz_dev, float lr_z, int64_t t_z,
                              float *grad_dev, float *out_dev, void *stream_) {
  return prior_step_impl(h, cfg, const_cast<float *>(theta_dev), nullptr, nullptr, seg_dev, data_z_dev, idx_dev, batch, batch_global, dz_dev, lr_z, 0.0f,
                         t_z, 0, grad_dev, 0, out_dev, stream_, "bgm_prior_grad");
}

extern "C" int bgm_prior_apply(bgm_handle *h, const bgm_prior_config *cfg, float *theta_dev, float *m_dev, float *v_dev, const float *grad_dev,
                               float lr_prior, int64_t t_prior, void *stream_) {
  if (!h || !theta_dev || !m_dev || !v_dev || !grad_dev || t_prior < 1) { bgm_set_error("bgm_prior_apply: bad argument"); return BGM_E_INVALID; }
  int64_t count = 0;
  int rc = bgm_prior_n_params(cfg, &count);
  if (rc) return rc;
  const float lr_t = (float)((double)lr_prior * std::sqrt(1.0 - std::pow(0.99, (double)t_prior)) / (1.0 - std::pow(0.9, (double)t_prior)));
  BGM_HIP_CHECK(hipSetDevice(h->device));
  hipLaunchKernelGGL(prior_adam_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, theta_dev, m_dev, v_dev, grad_dev, (int)count,
                     lr_t, 0.9f, 0.99f, 1e-7f);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}
