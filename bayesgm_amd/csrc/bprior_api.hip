// bprior_api.hip -- C ABI of the Bayesian conditional-prior network of IdentifiableCausalBGM(use_bnn=True) (bprior_kernels.h).
// replaces: src/bayesgm/models/causalbgm/identifiable.py:66-67 (prior_net as a BayesianFullyConnectedNet), :195-226 (the joint latent /
// prior-net step with kl_weight * sum(prior_net.losses)), :541-551 (prior_net(data_u) inside get_log_posterior).
#include <algorithm>
#include <cmath>
#include <string>

#include "bgm_host.h"
#include "bnn_kernels.h"
#include "bnn_state.h"
#include "bprior_kernels.h"

static constexpr float BP_B1 = 0.9f, BP_B2 = 0.99f, BP_ADAM_EPS = 1e-7f;       // prior_optimizer / posterior_optimizer (:95-99)

static int bprior_net_of(const bgm_prior_config *cfg, int norm_mode, BPriorNet &n, const char *who) {
  if (!cfg || cfg->n_layers < 1 || cfg->n_layers > BPRIOR_MAX_LAYERS) { bgm_set_error(std::string(who) + ": prior net needs 1..4 dense layers"); return BGM_E_INVALID; }
  if (norm_mode != 0 && norm_mode != 1) { bgm_set_error(std::string(who) + ": norm_mode must be 0 (batch statistics) or 1 (fixed)"); return BGM_E_INVALID; }
  n = BPriorNet{};
  n.n_layers = cfg->n_layers; n.norm_mode = norm_mode;
  for (int l = 0; l <= cfg->n_layers; ++l) {
    if (cfg->dims[l] < 1 || cfg->dims[l] > 1024) { bgm_set_error(std::string(who) + ": layer widths must be in [1, 1024]"); return BGM_E_INVALID; }
    n.dims[l] = cfg->dims[l];
    n.wmax = std::max(n.wmax, cfg->dims[l]);
  }
  int off = 0, w = 0;
  n.gamma_off = off; off += n.dims[0];
  n.beta_off = off; off += n.dims[0];
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.dims[l] * n.dims[l + 1];
    n.loc_off[l] = off; off += cnt;
    n.rho_off[l] = off; off += cnt;
    n.bias_off[l] = off; off += n.dims[l + 1];
    n.n_kernel += cnt;
    n.sin_w[l] = w; w += (n.dims[l] + 31) / 32;
    n.sout_w[l] = w; w += (n.dims[l + 1] + 31) / 32;
  }
  n.words = (w + 3) / 4 * 4;
  n.n_params = off;
  return BGM_OK;
}

extern "C" int bgm_bprior_n_params(const bgm_prior_config *cfg, int64_t *count) {
  BPriorNet n;
  int rc = bprior_net_of(cfg, 1, n, "bgm_bprior_n_params");
  if (rc) return rc;
  if (!count) { bgm_set_error("bgm_bprior_n_params: count == NULL"); return BGM_E_INVALID; }
  *count = n.n_params;
  return BGM_OK;
}

static double adam_lr(double lr, long long t) { return lr * std::sqrt(1.0 - std::pow((double)BP_B2, (double)t)) / (1.0 - std::pow((double)BP_B1, (double)t)); }

extern "C" int bgm_bprior_step(bgm_handle *h, const bgm_prior_config *cfg, int32_t norm_mode, float kl_weight, float *theta_dev, float *m_dev,
                               float *v_dev, const int32_t *seg_dev, float *data_z_dev, const int32_t *idx_dev, int32_t batch, int32_t batch_global,
                               int32_t row0, const float *dz_dev, float lr_z, float lr_prior, int64_t t_z, int64_t t_prior, uint64_t seed,
                               uint32_t stream_id, float *grad_dev, int32_t apply, float *out_dev, void *stream_) {
  if (!h || !h->configured && !h->bnn_state) { bgm_set_error("bgm_bprior_step: handle has no model"); return BGM_E_STATE; }
  BPriorNet n;
  int rc = bprior_net_of(cfg, norm_mode, n, "bgm_bprior_step");
  if (rc) return rc;
  const BnnState *s = static_cast<const BnnState *>(h->bnn_state);
  const int q = n.dims[n.n_layers] - 1;
  if (s && s->q != q) { bgm_set_error("bgm_bprior_step: the prior net's output width must be sum(z_dims) + 1"); return BGM_E_INVALID; }
  if (!theta_dev || !seg_dev || !data_z_dev || !idx_dev || !dz_dev || batch < 1 || batch_global < batch || t_z < 1 || t_prior < 1 || (apply && (!m_dev || !v_dev)) ||
      (!apply && !grad_dev)) {
    bgm_set_error("bgm_bprior_step: bad argument"); return BGM_E_INVALID;
  }
  if (norm_mode == 0 && batch < 2) { bgm_set_error("bgm_bprior_step: batch statistics need two rows"); return BGM_E_INVALID; }
  int tot = 0;
  for (int l = 0; l <= n.n_layers; ++l) tot += batch * n.dims[l];
  const size_t lds = sizeof(float) * ((size_t)tot + (size_t)batch * n.dims[0] + 2 * (size_t)n.n_kernel + (size_t)batch * q + 2 * (size_t)batch * n.wmax + 3 * (size_t)batch +
                                      2 * (size_t)n.dims[0] + (size_t)batch * n.words + (size_t)batch);
  if (lds > 150 * 1024) { bgm_set_error("bgm_bprior_step: minibatch x prior-net widths exceed the one-workgroup LDS budget"); return BGM_E_UNSUPPORTED; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bprior_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  BPriorStepArgs a{};
  a.net = n; a.theta = theta_dev; a.m = m_dev; a.v = v_dev; a.seg = seg_dev; a.data_z = data_z_dev; a.idx = idx_dev; a.dz = dz_dev;
  a.B = batch; a.q = q; a.lr_t_z = (float)adam_lr(lr_z, t_z); a.lr_t_p = (float)adam_lr(lr_prior, t_prior);
  a.b1 = BP_B1; a.b2 = BP_B2; a.eps = BP_ADAM_EPS; a.kl_weight = kl_weight;
  a.k0 = (unsigned)seed; a.k1 = (unsigned)(seed >> 32); a.stream = stream_id; a.row0 = (unsigned)row0;
  a.out = out_dev; a.inv_B = 1.0f / (float)batch_global; a.grad = grad_dev; a.apply = apply;
  hipLaunchKernelGGL(bprior_step_kernel, dim3(1), dim3(BPRIOR_THREADS), lds, stream, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bprior_apply(bgm_handle *h, const bgm_prior_config *cfg, float kl_weight, float *theta_dev, float *m_dev, float *v_dev,
                                const float *grad_dev, float lr_prior, int64_t t_prior, void *stream_) {
  if (!h) { bgm_set_error("bgm_bprior_apply: null handle"); return BGM_E_INVALID; }
  BPriorNet n;
  int rc = bprior_net_of(cfg, 1, n, "bgm_bprior_apply");
  if (rc) return rc;
  if (!theta_dev || !m_dev || !v_dev || !grad_dev || t_prior < 1) { bgm_set_error("bgm_bprior_apply: bad argument"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  hipLaunchKernelGGL(bprior_adam_kernel, dim3((n.n_params + 255) / 256), dim3(256), 0, (hipStream_t)stream_, n, theta_dev, m_dev, v_dev, grad_dev,
                     (float)adam_lr(lr_prior, t_prior), BP_B1, BP_B2, BP_ADAM_EPS, kl_weight);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

// Sampling calls of the session made afterwards (bgm_bnn_logpost, bgm_bnn_mh_run) use the conditional prior N(mu(u), sigma^2(u) I) with
// (mu, sigma^2) from ONE noisy call of this net per log-posterior evaluation; theta_dev == NULL clears it.
extern "C" int bgm_bnn_set_prior(bgm_handle *h, const bgm_prior_config *cfg, const float *theta_dev, const int32_t *seg_dev) {
  if (!h || !h->bnn_state) { bgm_set_error("bgm_bnn_set_prior: no Bayesian-network session (bgm_bnn_begin)"); return BGM_E_STATE; }
  BnnState *s = static_cast<BnnState *>(h->bnn_state);
  if (!theta_dev) { s->bp_on = false; s->bp_theta = nullptr; s->bp_seg = nullptr; return BGM_OK; }
  if (!seg_dev) { bgm_set_error("bgm_bnn_set_prior: seg_dev == NULL"); return BGM_E_INVALID; }
  // the prior net's input BatchNormalization follows the session's mode: inference mode (fixed mean 0 / variance 1), or the statistics of
  // the block of rows of the call (the reference as written; the nets g, h, f then run on the any-width path's statistics passes)
  BPriorNet n;
  int rc = bprior_net_of(cfg, s->cfg.norm_mode == 1 ? 1 : 0, n, "bgm_bnn_set_prior");
  if (rc) return rc;
  if (n.norm_mode == 0 && n.dims[0] > 64) { bgm_set_error("bgm_bnn_set_prior: batch statistics hold at most 64 segments"); return BGM_E_UNSUPPORTED; }
  s->bp_hist_n = -1;
  if (n.dims[n.n_layers] != s->q + 1) { bgm_set_error("bgm_bnn_set_prior: the prior net's output width must be sum(z_dims) + 1"); return BGM_E_INVALID; }
  s->bp_net = n; s->bp_theta = theta_dev; s->bp_seg = seg_dev; s->bp_on = true;
  return BGM_OK;
}

// rows_out [n_states][n][q + 2] for the calls stream0 .. stream0 + n_states - 1 of the blocks of this sampling call (bnf_api.hip)
int bprior_rows(bgm_handle *h, BnnState *s, long long n, int bs, int block0, uint64_t seed, uint32_t stream0, int n_states, hipStream_t stream,
                long long rib0) {
  const BPriorNet &net = s->bp_net;
  const int q = s->q, n_blocks = (int)((n + bs - 1) / bs);
  const size_t need = (size_t)n_states * (size_t)n * (size_t)(q + 2);
  if (s->bp_rows_cap < need) {
    if (s->bp_rows) BGM_HIP_CHECK(hipFree(s->bp_rows));
    s->bp_rows = nullptr; s->bp_rows_cap = 0;
    BGM_HIP_CHECK(hipMalloc((void **)&s->bp_rows, need * sizeof(float)));
    s->bp_rows_cap = need;
  }
  int tot = 0;
  for (int l = 0; l <= net.n_layers; ++l) tot += BPRIOR_ROWS_CHUNK * net.dims[l];
  const size_t lds = sizeof(float) * ((size_t)tot + (size_t)net.n_kernel + (size_t)BPRIOR_ROWS_CHUNK * net.words + BPRIOR_ROWS_CHUNK);
  if (lds > 150 * 1024) { bgm_set_error("conditional prior: the prior net's widths exceed the LDS budget of the sampling-side kernel"); return BGM_E_UNSUPPORTED; }
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bprior_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (net.norm_mode == 0) {
    if (rib0 != 0) { bgm_set_error("conditional prior: a share of one block needs the inference-mode normalisation"); return BGM_E_UNSUPPORTED; }
    if (s->bp_hist_n != n || s->bp_hist_bs != bs) {      // the segments of the panel do not change between the calls of a run
      const size_t cnt = (size_t)n_blocks * (size_t)net.dims[0];
      if (s->bp_hist_cap < cnt) {
        if (s->bp_hist) BGM_HIP_CHECK(hipFree(s->bp_hist));
        s->bp_hist = nullptr; s->bp_hist_cap = 0;
        BGM_HIP_CHECK(hipMalloc((void **)&s->bp_hist, cnt * sizeof(int)));
        s->bp_hist_cap = cnt;
      }
      hipLaunchKernelGGL(bprior_hist_kernel, dim3(n_blocks), dim3(256), 0, stream, s->bp_seg, n, bs, net.dims[0], s->bp_hist);
      BGM_HIP_CHECK(hipGetLastError());
      s->bp_hist_n = n; s->bp_hist_bs = bs;
    }
  }
  BPriorRowsArgs a{};
  a.hist = s->bp_hist;
  a.net = net; a.theta = s->bp_theta; a.seg = s->bp_seg; a.n = n; a.bs = bs; a.n_blocks = n_blocks; a.block0 = block0; a.q = q;
  a.parts = std::max(1, std::min((bs + BPRIOR_ROWS_CHUNK - 1) / BPRIOR_ROWS_CHUNK, std::max(1, 2048 / std::max(1, n_blocks * n_states))));
  a.k0 = (unsigned)seed; a.k1 = (unsigned)(seed >> 32); a.stream0 = stream0; a.rows_out = s->bp_rows;
  a.rib0 = (unsigned)rib0;      // (a rank's share of one block: the sign words are keyed by the position in the block)
  hipLaunchKernelGGL(bprior_rows_kernel, dim3(n_blocks * a.parts, n_states), dim3(BPRIOR_THREADS), lds, stream, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}
