// bgm_egm_api.hip -- C-ABI entry points of the BGM EGM warm start (include/bgm_hip.h, BGM EGM section).
#include <algorithm>
#include <cmath>
#include <cstring>

#include "bgm_host.h"
#include "bgm_state.h"
#include "bgm_egm_kernels.h"

static constexpr float BEGM_B1 = 0.5f, BEGM_B2 = 0.9f, BEGM_ADAM_EPS = 1e-7f;   // bgm/base.py:82-85, Keras epsilon

struct BgmEgmState {
  bgm_bgm_egm_config cfg{};
  BgmEgmArgs base{};
  size_t n_g = 0, n_e = 0, n_gen = 0, n_dz = 0, n_dx = 0, n_disc = 0, ws_floats = 0, enc_ws_per_block = 0;
  long long t_g = 0, t_d = 0;
  float *dev = nullptr;
  float *enc_ws = nullptr;
  int enc_blocks = 0;
};

static BgmEgmState *best(bgm_handle *h) { return static_cast<BgmEgmState *>(h->bgm_egm_state); }

void bgm_bgm_egm_free_state(bgm_handle *h) {
  if (!h->bgm_egm_state) return;
  BgmEgmState *s = best(h);
  if (s->dev) hipFree(s->dev);
  if (s->enc_ws) hipFree(s->enc_ws);
  delete s;
  h->bgm_egm_state = nullptr;
}

static int fill_disc(EgmDisc &d, int in_dim, int n_hidden, const int32_t *units) {
  d.n_hidden = n_hidden;
  d.dims[0] = in_dim;
  for (int l = 0; l < n_hidden; ++l) d.dims[l + 1] = units[l];
  d.dims[n_hidden + 1] = 1;
  int o = 0;
  for (int l = 0; l <= n_hidden; ++l) { d.w[l] = o; o += d.dims[l] * d.dims[l + 1]; }
  for (int l = 0; l <= n_hidden; ++l) { d.b[l] = o; o += d.dims[l + 1]; }
  for (int l = 0; l < n_hidden; ++l) { d.gamma[l] = o; o += d.dims[l + 1]; }
  for (int l = 0; l < n_hidden; ++l) { d.beta[l] = o; o += d.dims[l + 1]; }
  d.n_params = o;
  egm_finish_disc(d);
  d.fixed_norm = 0;
  return o;
}

extern "C" int bgm_bgm_egm_begin(bgm_handle *h, const bgm_bgm_egm_config *cfg, const float *theta_e_host, int64_t count_e,
                                 const float *theta_dz_host, int64_t count_dz, const float *theta_dx_host, int64_t count_dx,
                                 void *stream_) {
  (void)stream_;
  if (!h || !h->bgm_state || !bst(h)->configured || !bst(h)->set) { bgm_set_error("bgm_bgm_egm_begin: configure the BGM model and set its weights first"); return BGM_E_STATE; }
  if (!cfg || !theta_e_host || !theta_dz_host || !theta_dx_host) { bgm_set_error("bgm_bgm_egm_begin: NULL argument"); return BGM_E_INVALID; }
  if (cfg->batch_size < 2 || cfg->batch_size > 4096) { bgm_set_error("bgm_bgm_egm_begin: batch_size must be in [2, 4096]"); return BGM_E_INVALID; }
  if (cfg->n_hidden_e < 1 || cfg->n_hidden_e + 1 > EGM_MAX_LAYERS || cfg->n_hidden_dz < 1 || cfg->n_hidden_dz + 1 > EGM_MAX_LAYERS ||
      cfg->n_hidden_dx < 1 || cfg->n_hidden_dx + 1 > EGM_MAX_LAYERS) { bgm_set_error("bgm_bgm_egm_begin: bad layer counts"); return BGM_E_INVALID; }
  BgmState *bs = bst(h);
  const int q = bs->cfg.z_dim, p = bs->cfg.x_dim, NH = bs->cfg.n_hidden_g;
  if (NH + 1 > EGM_MAX_LAYERS) { bgm_set_error("bgm_bgm_egm_begin: too many generator layers"); return BGM_E_UNSUPPORTED; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (bs->fit_active) { bgm_set_error("bgm_bgm_egm_begin: a fit session is active"); return BGM_E_STATE; }
  bgm_bgm_egm_free_state(h);
  BgmEgmState *s = new BgmEgmState();
  h->bgm_egm_state = s;
  s->cfg = *cfg;
  BgmEgmArgs &a = s->base;
  // generator
  EgmVarNet &g = a.g;
  g.q = q; g.p = p; g.hlast = bs->cfg.g_units[NH - 1];
  g.mlp.n_layers = NH + 1;
  g.mlp.dims[0] = q;
  for (int i = 0; i < NH; ++i) g.mlp.dims[i + 1] = bs->cfg.g_units[i];
  g.mlp.dims[NH + 1] = p;
  g.mlp.off = 4 * q;
  egm_finish_mlp(g.mlp);
  int o = 4 * q;
  for (int l = 0; l <= NH; ++l) o += g.mlp.dims[l] * g.mlp.dims[l + 1] + g.mlp.dims[l + 1];
  g.wvar = o; o += g.hlast * p;
  g.bvar = o; o += p;
  s->n_g = (size_t)o;
  if (s->n_g != bs->theta.size()) { bgm_bgm_egm_free_state(h); bgm_set_error("bgm_bgm_egm_begin: internal parameter count mismatch"); return BGM_E_INVALID; }
  // encoder
  EgmMlp &e = a.e;
  e.n_layers = cfg->n_hidden_e + 1;
  e.dims[0] = p;
  for (int i = 0; i < cfg->n_hidden_e; ++i) e.dims[i + 1] = cfg->e_units[i];
  e.dims[cfg->n_hidden_e + 1] = q;
  e.off = (int)s->n_g;
  egm_finish_mlp(e);
  size_t ne = 0;
  for (int l = 0; l < e.n_layers; ++l) ne += (size_t)e.dims[l] * e.dims[l + 1] + e.dims[l + 1];
  s->n_e = ne;
  s->n_gen = s->n_g + s->n_e;
  s->n_dz = (size_t)fill_disc(a.dz, q, cfg->n_hidden_dz, cfg->dz_units);
  s->n_dx = (size_t)fill_disc(a.dx, p, cfg->n_hidden_dx, cfg->dx_units);
  a.dz.fixed_norm = a.dx.fixed_norm = h->disc_norm;
  s->n_disc = s->n_dz + s->n_dx;
  if ((int64_t)s->n_e != count_e || (int64_t)s->n_dz != count_dz || (int64_t)s->n_dx != count_dx) {
    const std::string msg = "bgm_bgm_egm_begin: expected " + std::to_string(s->n_e) + " / " + std::to_string(s->n_dz) + " / " +
                            std::to_string(s->n_dx) + " encoder / dz / dx parameters";
    bgm_bgm_egm_free_state(h);
    bgm_set_error(msg);
    return BGM_E_INVALID;
  }
  int wmax = std::max(q, p);
  size_t widths = 0;
  auto scan = [&](const int *dims, int n_layers) { size_t t = 0; for (int l = 0; l <= n_layers; ++l) { wmax = std::max(wmax, dims[l]); t += dims[l] + 4; } return t; };
  const size_t wg = scan(g.mlp.dims, g.mlp.n_layers), we = scan(e.dims, e.n_layers);
  const size_t wdz = scan(a.dz.dims, a.dz.n_hidden + 1), wdx = scan(a.dx.dims, a.dx.n_hidden + 1);
  const int B = cfg->batch_size;
  a.B = B; a.wmax = wmax; a.n_gen = (int)s->n_gen; a.n_disc = (int)s->n_disc;
  a.gamma = cfg->gamma; a.alpha = cfg->alpha;
  // workspace bound (floats per row): generator caches x2 (+ zhat, zn, s_raw), encoder caches x2, data-sized temporaries,
  // one discriminator cache, the gradient-penalty scratch of the wider discriminator, scratch rows
  widths = 2 * (wg + 3 * (size_t)q + p) + 2 * we + 10 * (size_t)p + 3 * std::max(wdz, wdx) + 8 * std::max(wdz, wdx) + 12 * (size_t)wmax + 64;
  s->ws_floats = (size_t)B * widths + s->n_disc + 8192;
  const size_t total = 4 * s->n_gen + 4 * s->n_disc + s->ws_floats + 64;
  BGM_HIP_CHECK(hipMalloc(&s->dev, total * sizeof(float)));
  BGM_HIP_CHECK(hipMemset(s->dev, 0, total * sizeof(float)));
  float *pp = s->dev;
  auto take = [&](size_t n) { float *r = pp; pp += (n + 3) / 4 * 4; return r; };
  a.theta_g = take(s->n_gen); a.m_g = take(s->n_gen); a.v_g = take(s->n_gen); a.grad_g = take(s->n_gen);
  a.theta_d = take(s->n_disc); a.m_d = take(s->n_disc); a.v_d = take(s->n_disc); a.grad_d = take(s->n_disc);
  a.ws = take(s->ws_floats);
  BGM_HIP_CHECK(hipMemcpy(a.theta_g, bs->theta.data(), s->n_g * sizeof(float), hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMemcpy(a.theta_g + s->n_g, theta_e_host, s->n_e * sizeof(float), hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMemcpy(a.theta_d, theta_dz_host, s->n_dz * sizeof(float), hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMemcpy(a.theta_d + s->n_dz, theta_dx_host, s->n_dx * sizeof(float), hipMemcpyHostToDevice));
  // encoder workspace for whole-panel passes
  s->enc_blocks = h->n_cus;
  s->enc_ws_per_block = (size_t)64 * (we + 8) + 64;
  BGM_HIP_CHECK(hipMalloc(&s->enc_ws, s->enc_ws_per_block * s->enc_blocks * sizeof(float)));
  return BGM_OK;
}

static EgmAdam begm_adam(float lr, long long t) {
  EgmAdam ad;
  ad.b1 = BEGM_B1; ad.b2 = BEGM_B2; ad.eps = BEGM_ADAM_EPS;
  ad.lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow((double)BEGM_B2, (double)t)) / (1.0 - std::pow((double)BEGM_B1, (double)t)));
  return ad;
}
static constexpr int BEGM_LDS = 64 * (int)sizeof(float);

extern "C" int bgm_bgm_egm_disc_step(bgm_handle *h, const float *z_dev, const float *x_dev, const float *noise_dev, float eps_z,
                                     float eps_x, int32_t apply, float *out_dev, void *stream_) {
  if (!h || !h->bgm_egm_state) { bgm_set_error("bgm_bgm_egm_disc_step: call bgm_bgm_egm_begin first"); return BGM_E_STATE; }
  if (!z_dev || !x_dev || !noise_dev) { bgm_set_error("bgm_bgm_egm_disc_step: NULL pointer"); return BGM_E_INVALID; }
  BgmEgmState *s = best(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BgmEgmArgs a = s->base;
  a.z = z_dev; a.x = x_dev; a.n1 = noise_dev; a.n2 = nullptr; a.eps_z = eps_z; a.eps_x = eps_x; a.out = out_dev; a.apply = apply ? 1 : 0;
  if (apply) s->t_d += 1;
  a.adam = begm_adam(s->cfg.lr, std::max<long long>(1, s->t_d));
  hipLaunchKernelGGL(bgm_egm_disc_step_kernel, dim3(1), dim3(EGM_THREADS), BEGM_LDS, (hipStream_t)stream_, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bgm_egm_gen_step(bgm_handle *h, const float *z_dev, const float *x_dev, const float *noise1_dev,
                                    const float *noise2_dev, int32_t apply, float *out_dev, void *stream_) {
  if (!h || !h->bgm_egm_state) { bgm_set_error("bgm_bgm_egm_gen_step: call bgm_bgm_egm_begin first"); return BGM_E_STATE; }
  if (!z_dev || !x_dev || !noise1_dev || !noise2_dev) { bgm_set_error("bgm_bgm_egm_gen_step: NULL pointer"); return BGM_E_INVALID; }
  BgmEgmState *s = best(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BgmEgmArgs a = s->base;
  a.z = z_dev; a.x = x_dev; a.n1 = noise1_dev; a.n2 = noise2_dev; a.eps_z = a.eps_x = 0.0f; a.out = out_dev; a.apply = apply ? 1 : 0;
  if (apply) s->t_g += 1;
  a.adam = begm_adam(s->cfg.lr, std::max<long long>(1, s->t_g));
  hipLaunchKernelGGL(bgm_egm_gen_step_kernel, dim3(1), dim3(EGM_THREADS), BEGM_LDS, (hipStream_t)stream_, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

static int begm_region(BgmEgmState *s, int32_t what, float **ptr, size_t *n) {
  switch (what) {
    case 0: *ptr = s->base.theta_g; *n = s->n_gen; return BGM_OK;
    case 1: *ptr = s->base.theta_d; *n = s->n_disc; return BGM_OK;
    case 2: *ptr = s->base.grad_g; *n = s->n_gen; return BGM_OK;
    case 3: *ptr = s->base.grad_d; *n = s->n_disc; return BGM_OK;
    default: bgm_set_error("bgm_bgm_egm_read/write: what must be 0..3"); return BGM_E_INVALID;
  }
}

extern "C" int bgm_bgm_egm_read(bgm_handle *h, int32_t what, float *host, int64_t count, void *stream_) {
  if (!h || !h->bgm_egm_state || !host) { bgm_set_error("bgm_bgm_egm_read: no session / NULL"); return BGM_E_STATE; }
  BgmEgmState *s = best(h);
  float *src; size_t n;
  int rc = begm_region(s, what, &src, &n);
  if (rc) return rc;
  if ((size_t)count != n) { bgm_set_error("bgm_bgm_egm_read: expected " + std::to_string(n) + " floats"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  BGM_HIP_CHECK(hipMemcpy(host, src, n * sizeof(float), hipMemcpyDeviceToHost));
  return BGM_OK;
}

extern "C" int bgm_bgm_egm_write(bgm_handle *h, int32_t what, const float *host, int64_t count, void *stream_) {
  if (!h || !h->bgm_egm_state || !host) { bgm_set_error("bgm_bgm_egm_write: no session / NULL"); return BGM_E_STATE; }
  BgmEgmState *s = best(h);
  float *dst; size_t n;
  int rc = begm_region(s, what, &dst, &n);
  if (rc) return rc;
  if (what > 1 || (size_t)count != n) { bgm_set_error("bgm_bgm_egm_write: parameters only, expected " + std::to_string(n) + " floats"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  BGM_HIP_CHECK(hipMemcpy(dst, host, n * sizeof(float), hipMemcpyHostToDevice));
  return BGM_OK;
}

extern "C" int bgm_bgm_egm_encode(bgm_handle *h, const float *x_dev, int64_t n, float *z_dev, void *stream_) {
  if (!h || !h->bgm_egm_state) { bgm_set_error("bgm_bgm_egm_encode: no session"); return BGM_E_STATE; }
  if (n <= 0) return BGM_OK;
  if (!x_dev || !z_dev) { bgm_set_error("bgm_bgm_egm_encode: NULL pointer"); return BGM_E_INVALID; }
  BgmEgmState *s = best(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  const int B = 64;
  const int grid = (int)std::max<long long>(1, std::min<long long>((n + B - 1) / B, s->enc_blocks));
  hipLaunchKernelGGL(bgm_egm_encode_kernel, dim3(grid), dim3(EGM_THREADS), BEGM_LDS, (hipStream_t)stream_, s->base.e, s->base.theta_g, x_dev,
                     (long long)n, z_dev, s->enc_ws, (long long)s->enc_ws_per_block, B);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bgm_egm_sync(bgm_handle *h, void *stream_) {
  if (!h || !h->bgm_egm_state) { bgm_set_error("bgm_bgm_egm_sync: no session"); return BGM_E_STATE; }
  BgmEgmState *s = best(h);
  BgmState *bs = bst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  BGM_HIP_CHECK(hipMemcpy(bs->theta.data(), s->base.theta_g, s->n_g * sizeof(float), hipMemcpyDeviceToHost));
  bs->set = true; bs->blob_valid = false; bs->gx_valid = false;
  return BGM_OK;
}

extern "C" int bgm_bgm_egm_end(bgm_handle *h, void *stream_) {
  if (!h) return BGM_E_INVALID;
  if (!h->bgm_egm_state) return BGM_OK;
  int rc = bgm_bgm_egm_sync(h, stream_);
  bgm_bgm_egm_free_state(h);
  return rc;
}
