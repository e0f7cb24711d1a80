// bgmb_api.hip -- C ABI of BGM with the Bayesian generator (use_bnn=True): session, minibatch steps, log posterior, HMC, decode.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

#include "bgm_host.h"
#include "bgmb_kernels.h"
#include "bgmb_state.h"
#include "gx_flipout.h"

static BgmbState *vst(bgm_handle *h) { return static_cast<BgmbState *>(h->bvn_state); }
static void gxf_free(BgmbState *s);

void bgm_bvn_free_state(bgm_handle *h) {
  if (!h->bvn_state) return;
  BgmbState *s = vst(h);
  if (s->dev) hipFree(s->dev);
  if (s->big_dev) hipFree(s->big_dev);
  if (s->dw_dev) hipFree(s->dw_dev);
  gxf_free(s);
  bgmf_free(s);
  bgm_bvn_egm_free(s->egm);
  delete s;
  h->bvn_state = nullptr;
}

int bgmb_fill(const bgm_bvn_config *cfg, BnnNet &n) {
  if (!cfg) return BGM_E_INVALID;
  const int T = cfg->n_hidden_g;
  if (cfg->x_dim < 1 || cfg->z_dim < 1 || T < 1 || T + 2 > BNN_MAX_LAYERS) return BGM_E_INVALID;
  std::memset(&n, 0, sizeof(n));
  n.n_layers = T + 2;
  n.dims[0] = cfg->z_dim;
  for (int i = 0; i < T; ++i) { if (cfg->g_units[i] < 1) return BGM_E_INVALID; n.dims[i + 1] = cfg->g_units[i]; }
  n.dims[T + 1] = cfg->x_dim;
  n.dims[T + 2] = cfg->x_dim;
  n.off = 0; n.net_id = 0; n.bn_fixed = 0; n.heads = 1; n.mv = 1;
  n.prior_iv = 100.0f; n.prior_logs = logf(0.1f); n.bias_prior = 1;
  bnn_finish_net(n);
  return BGM_OK;
}

extern "C" int bgm_bvn_layout(const bgm_bvn_config *cfg, int64_t *n_params) {
  BnnNet n;
  if (!n_params || bgmb_fill(cfg, n)) { bgm_set_error("bgm_bvn_layout: bad configuration"); return BGM_E_INVALID; }
  *n_params = n.n_params;
  return BGM_OK;
}

extern "C" int bgm_bvn_begin(bgm_handle *h, const bgm_bvn_config *cfg, const float *theta_host, int64_t count, void *stream_) {
  (void)stream_;
  if (!h || !cfg || !theta_host) { bgm_set_error("bgm_bvn_begin: NULL argument"); return BGM_E_INVALID; }
  if (cfg->max_batch < 2 || cfg->max_batch > BGMB_MAX_BATCH) { bgm_set_error("bgm_bvn_begin: max_batch must be in [2, 4096]"); return BGM_E_UNSUPPORTED; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  bgm_bvn_free_state(h);
  BgmbState *s = new BgmbState();
  h->bvn_state = s;
  s->cfg = *cfg;
  if (bgmb_fill(cfg, s->net)) { bgm_bvn_free_state(h); bgm_set_error("bgm_bvn_begin: bad configuration"); return BGM_E_INVALID; }
  s->n_params = s->net.n_params;
  if (count != s->n_params) { bgm_bvn_free_state(h); bgm_set_error("bgm_bvn_begin: wrong parameter count"); return BGM_E_INVALID; }
  s->q = cfg->z_dim; s->p = cfg->x_dim;
  int wmax = std::max(2 * s->p, s->q);
  for (int i = 0; i <= s->net.n_layers; ++i) wmax = std::max(wmax, s->net.dims[i]);
  s->wmax = wmax;
  const int B = cfg->max_batch;
  s->ws_floats = bgmb_ws_floats(B, s->q, s->p, wmax) + 2 * bnn_cache_floats(s->net, B) + 256;      // (the second cache: the kept upstream gradients of a wide theta step)
  const size_t np = ((size_t)s->n_params + 63) & ~(size_t)63;
  const size_t total = 4 * np + s->ws_floats + 64 + 64;      // (+ 64: out, + 64: the KL partial sums of a wide theta step)
  BGM_HIP_CHECK(hipMalloc((void **)&s->dev, sizeof(float) * total));
  BGM_HIP_CHECK(hipMemset(s->dev, 0, sizeof(float) * total));
  s->theta_dev = s->dev; s->m_dev = s->dev + np; s->v_dev = s->dev + 2 * np; s->grad_dev = s->dev + 3 * np;
  s->ws_dev = s->dev + 4 * np;
  s->out_dev = s->ws_dev + s->ws_floats;
  BGM_HIP_CHECK(hipMemcpy(s->theta_dev, theta_host, sizeof(float) * count, hipMemcpyHostToDevice));
  s->t_theta = 0; s->t_z = 0;
  return BGM_OK;
}

static int bvn_need(bgm_handle *h, const char *who) {
  if (!h || !h->bvn_state) { bgm_set_error(std::string(who) + ": no session (bgm_bvn_begin)"); return BGM_E_STATE; }
  return BGM_OK;
}

static float *bvn_what(BgmbState *s, int what) {
  return what == 0 ? s->theta_dev : what == 1 ? s->grad_dev : what == 2 ? s->m_dev : what == 3 ? s->v_dev : nullptr;
}

extern "C" int bgm_bvn_read(bgm_handle *h, int32_t what, float *host, int64_t count, void *stream_) {
  int rc = bvn_need(h, "bgm_bvn_read");
  if (rc) return rc;
  BgmbState *s = vst(h);
  float *src = bvn_what(s, what);
  if (!src || !host || count != s->n_params) { bgm_set_error("bgm_bvn_read: bad argument"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  BGM_HIP_CHECK(hipMemcpy(host, src, sizeof(float) * count, hipMemcpyDeviceToHost));
  return BGM_OK;
}

extern "C" int bgm_bvn_write(bgm_handle *h, int32_t what, const float *host, int64_t count, void *stream_) {
  int rc = bvn_need(h, "bgm_bvn_write");
  if (rc) return rc;
  BgmbState *s = vst(h);
  float *dst = bvn_what(s, what);
  if (!dst || !host || count != s->n_params) { bgm_set_error("bgm_bvn_write: bad argument"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  BGM_HIP_CHECK(hipMemcpy(dst, host, sizeof(float) * count, hipMemcpyHostToDevice));
  return BGM_OK;
}

static float bvn_lr_t(float lr, long long t_) {
  const double t = (double)t_;
  return (float)((double)lr * std::sqrt(1.0 - std::pow((double)BGMB_ADAM_B2, t)) / (1.0 - std::pow((double)BGMB_ADAM_B1, t)));
}

static int bvn_step_args(bgm_handle *h, const char *who, BgmbArgs &a, const float *x_dev, float *data_z_dev, const int32_t *idx_dev,
                         int32_t batch, int32_t batch_global, uint64_t seed, uint32_t stream_id, float *out_dev) {
  int rc = bvn_need(h, who);
  if (rc) return rc;
  BgmbState *s = vst(h);
  if (!x_dev || !data_z_dev || !idx_dev || batch < 2 || batch > s->cfg.max_batch || batch_global < batch) {
    bgm_set_error(std::string(who) + ": bad argument"); return BGM_E_INVALID;
  }
  std::memset(&a, 0, sizeof(a));
  a.net = s->net;
  a.theta = s->theta_dev; a.m = s->m_dev; a.v = s->v_dev; a.grad = s->grad_dev;
  a.B = batch; a.q = s->q; a.p = s->p; a.wmax = s->wmax; a.kl_weight = s->cfg.kl_weight;
  a.data_z = data_z_dev; a.idx = idx_dev; a.x_ = x_dev;
  a.k0 = (uint32_t)(seed & 0xFFFFFFFFull); a.k1 = (uint32_t)(seed >> 32); a.stream = stream_id;
  a.inv_B = 1.0f / (float)batch_global;
  a.ws = s->ws_dev; a.out = out_dev;
  return BGM_OK;
}

extern "C" int bgm_bvn_theta_step(bgm_handle *h, const float *x_dev, float *data_z_dev, const int32_t *idx_dev, int32_t batch,
                                  int32_t batch_global, float lr, uint64_t seed, uint32_t stream_id, int32_t apply, float *out_dev,
                                  void *stream_) {
  BgmbArgs a;
  int rc = bvn_step_args(h, "bgm_bvn_theta_step", a, x_dev, data_z_dev, idx_dev, batch, batch_global, seed, stream_id, out_dev);
  if (rc) return rc;
  BgmbState *s = vst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  a.apply = apply ? 1 : 0;
  a.kl_weight *= (float)batch / (float)batch_global;      // data parallel: the ranks' gradients are summed, the KL term counts once
  if (apply) { s->t_theta += 1; a.adam = BnnAdam{bvn_lr_t(lr, s->t_theta), BGMB_ADAM_B1, BGMB_ADAM_B2, BGMB_ADAM_EPS}; }
  // one workgroup walks the call; the elementwise parts (eps / dW, KL terms, Adam) and the parameter-gradient tiles run over the chip
  // around it -- same arithmetic per element / tile.  BGM_BNN_STEP_ONE_LAUNCH: everything inside the one workgroup.
  static const bool one_launch = std::getenv("BGM_BNN_STEP_ONE_LAUNCH") != nullptr;
  a.wide = one_launch ? 0 : 1;
  a.kl_part = s->out_dev + 64;
  hipStream_t st = (hipStream_t)stream_;
  if (a.wide) hipLaunchKernelGGL(bgmb_noise_kernel, dim3(16), dim3(BNN_THREADS), 0, st, a);
  hipLaunchKernelGGL(bgmb_theta_step_kernel, dim3(1), dim3(BNN_THREADS), 0, st, a);
  if (a.wide) {
    hipLaunchKernelGGL(bgmb_dw_kernel, dim3(16), dim3(BNN_THREADS), 0, st, a);
    hipLaunchKernelGGL(bgmb_kl_kernel, dim3(BNN_KL_PARTS), dim3(256), 0, st, a);
    hipLaunchKernelGGL(bgmb_kl_finish_kernel, dim3(1), dim3(64), 0, st, a);
    if (a.apply)
      hipLaunchKernelGGL(bnn_adam_kernel, dim3((s->n_params + 255) / 256), dim3(256), 0, st, s->theta_dev, s->m_dev, s->v_dev, s->grad_dev, s->n_params, a.adam);
  }
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bvn_grad_exchange(bgm_handle *h, float *buf_dev, int32_t to_session, void *stream_) {
  int rc = bvn_need(h, "bgm_bvn_grad_exchange");
  if (rc) return rc;
  if (!buf_dev) { bgm_set_error("bgm_bvn_grad_exchange: NULL buffer"); return BGM_E_INVALID; }
  BgmbState *s = vst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipMemcpyAsync(to_session ? s->grad_dev : buf_dev, to_session ? buf_dev : s->grad_dev, sizeof(float) * s->n_params,
                               hipMemcpyDeviceToDevice, (hipStream_t)stream_));
  return BGM_OK;
}

extern "C" int bgm_bvn_theta_apply(bgm_handle *h, float lr, void *stream_) {
  int rc = bvn_need(h, "bgm_bvn_theta_apply");
  if (rc) return rc;
  BgmbState *s = vst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  s->t_theta += 1;
  const BnnAdam ad{bvn_lr_t(lr, s->t_theta), BGMB_ADAM_B1, BGMB_ADAM_B2, BGMB_ADAM_EPS};
  hipLaunchKernelGGL(bnn_adam_kernel, dim3((s->n_params + 255) / 256), dim3(256), 0, (hipStream_t)stream_, s->theta_dev, s->m_dev,
                     s->v_dev, s->grad_dev, s->n_params, ad);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bvn_z_step(bgm_handle *h, const float *x_dev, float *data_z_dev, const int32_t *idx_dev, int32_t batch,
                              int32_t batch_global, float lr_z, uint64_t seed, uint32_t stream_id, float *out_dev, void *stream_) {
  BgmbArgs a;
  int rc = bvn_step_args(h, "bgm_bvn_z_step", a, x_dev, data_z_dev, idx_dev, batch, batch_global, seed, stream_id, out_dev);
  if (rc) return rc;
  BgmbState *s = vst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  s->t_z += 1;
  a.z_lr_t = bvn_lr_t(lr_z, s->t_z); a.z_b1 = BGMB_ADAM_B1; a.z_b2 = BGMB_ADAM_B2; a.z_eps = BGMB_ADAM_EPS;
  a.wide = std::getenv("BGM_BNN_STEP_ONE_LAUNCH") ? 0 : 1;
  if (a.wide) hipLaunchKernelGGL(bgmb_noise_kernel, dim3(16), dim3(BNN_THREADS), 0, (hipStream_t)stream_, a);
  hipLaunchKernelGGL(bgmb_z_step_kernel, dim3(1), dim3(BNN_THREADS), 0, (hipStream_t)stream_, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

// ----------------------------------------------------------------------------------------- large batches
static int bvn_big_ws(bgm_handle *h, BgmbState *s, long long tiles, float *&ws, long long &stride) {
  BnnNet n = s->net;
  n.bn_fixed = 2;
  stride = (long long)((bgmb_tile_floats(n, s->q, s->p, s->wmax) + 63) & ~(size_t)63);
  const size_t need = (size_t)stride * (size_t)tiles;
  if (need > s->big_cap) {
    if (s->big_dev) { BGM_HIP_CHECK(hipDeviceSynchronize()); hipFree(s->big_dev); s->big_dev = nullptr; s->big_cap = 0; }
    BGM_HIP_CHECK(hipMalloc((void **)&s->big_dev, sizeof(float) * need));
    s->big_cap = need;
  }
  (void)h;
  ws = s->big_dev;
  return BGM_OK;
}

// perturbations of `slots` generator calls: streams stream0 + s * step for s < n_seq, then (optionally) stream_extra
static long long bvn_dw_stride(const BgmbState *s) { return ((long long)s->net.eoff[s->net.n_layers] + 63) & ~63LL; }
static int bvn_noise(bgm_handle *h, BgmbState *s, uint64_t seed, uint32_t stream0, uint32_t step, int n_seq, bool extra,
                     uint32_t stream_extra, hipStream_t stream) {
  const long long stride = bvn_dw_stride(s);
  const int slots = n_seq + (extra ? 1 : 0);
  const size_t need = (size_t)stride * (size_t)slots;
  if (need > s->dw_cap) {
    if (s->dw_dev) { BGM_HIP_CHECK(hipDeviceSynchronize()); hipFree(s->dw_dev); s->dw_dev = nullptr; s->dw_cap = 0; }
    BGM_HIP_CHECK(hipMalloc((void **)&s->dw_dev, sizeof(float) * need));
    s->dw_cap = need;
  }
  BgmbNoiseArgs a;
  std::memset(&a, 0, sizeof(a));
  a.net = s->net; a.theta = s->theta_dev; a.dw = s->dw_dev; a.stride = stride;
  a.k0 = (uint32_t)(seed & 0xFFFFFFFFull); a.k1 = (uint32_t)(seed >> 32);
  a.stream0 = stream0; a.stream_step = step; a.n_seq = n_seq; a.stream_extra = stream_extra;
  const int groups = (int)((stride / 4 + 255) / 256);
  hipLaunchKernelGGL(bgmb_noise_kernel, dim3((unsigned)std::max(1, std::min(groups, 64)), (unsigned)slots), dim3(256), 0, stream, a);
  BGM_HIP_CHECK(hipGetLastError());
  (void)h;
  return BGM_OK;
}

// rows per workgroup: whole 64-row tiles once they fill the chip, otherwise 16-row multiples so that every CU gets work
static int bvn_rt(const bgm_handle *h, long long n) {
  const long long per_cu = (n + h->n_cus - 1) / h->n_cus;
  return (int)std::min<long long>(BGMB_RT, std::max<long long>(16, (per_cu + 15) / 16 * 16));
}

static void bvn_big_base(BgmbState *s, BgmbBigArgs &a, uint64_t seed) {
  std::memset(&a, 0, sizeof(a));
  a.net = s->net;
  a.net.bn_fixed = 2;
  a.theta = s->theta_dev;
  a.q = s->q; a.p = s->p; a.wmax = s->wmax;
  a.k0 = (uint32_t)(seed & 0xFFFFFFFFull); a.k1 = (uint32_t)(seed >> 32);
}

extern "C" int bgm_bvn_logpost(bgm_handle *h, const float *z_dev, const float *x_dev, int64_t n, int64_t row_base, uint64_t seed,
                               uint32_t stream_id, float *out_dev, float *grad_dev, void *stream_) {
  int rc = bvn_need(h, "bgm_bvn_logpost");
  if (rc) return rc;
  if (n == 0) return BGM_OK;
  if (!z_dev || !x_dev || !out_dev || n < 0) { bgm_set_error("bgm_bvn_logpost: bad argument"); return BGM_E_INVALID; }
  BgmbState *s = vst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  const int rt = bvn_rt(h, n);
  const long long n_tiles = (n + rt - 1) / rt;
  const long long tiles = std::min<long long>(n_tiles, 8LL * h->n_cus);       // workgroups (one workspace slice each)
  BgmbBigArgs a;
  bvn_big_base(s, a, seed);
  a.rt = rt; a.n_tiles = n_tiles;
  rc = bvn_big_ws(h, s, tiles, a.ws, a.ws_stride);
  if (rc) return rc;
  a.x = x_dev; a.n = n; a.row_base = row_base;
  a.state = const_cast<float *>(z_dev); a.logp = out_dev; a.grad = grad_dev;
  a.stream = stream_id;
  rc = bvn_noise(h, s, seed, stream_id, 0u, 1, false, 0u, (hipStream_t)stream_);
  if (rc) return rc;
  a.dw = s->dw_dev; a.dw_stride = bvn_dw_stride(s);
  hipLaunchKernelGGL(bgmb_logpost_kernel, dim3((unsigned)tiles), dim3(BNN_THREADS), 0, (hipStream_t)stream_, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// frozen-noise HMC on 32-row LDS tiles (gx_flipout.h): plan + packs
// ---------------------------------------------------------------------------------------------------------------------------
struct GxfState {
  GxfModel f{};
  GxfPackArgs pk{};
  float *pack = nullptr, *packT = nullptr, *dw = nullptr, *dwT = nullptr;
  int lds_bytes = 0;
};
static void gxf_free(BgmbState *s) {
  GxfState *g = static_cast<GxfState *>(s->gxf);
  if (!g) return;
  for (void *p : {(void *)g->pack, (void *)g->packT, (void *)g->dw, (void *)g->dwT}) if (p) hipFree(p);
  delete g;
  s->gxf = nullptr;
}
// 1: this generator is outside the LDS tiles (too wide / too deep): the workspace kernel serves it
static int gxf_session(BgmbState *s, GxfState *&g) {
  g = static_cast<GxfState *>(s->gxf);
  if (g) return BGM_OK;
  const BnnNet &n = s->net;
  const int NF = n.n_layers, T = NF - 2, q = s->q, p = s->p;          // Flipout layers: T trunk layers, mean head, variance head
  if (T < 1 || T + 1 > GX_MAXL) return 1;
  GxfState *st = new GxfState();
  GxBgmModel &m = st->f.m;
  const int Pp = gx_pad32(p);
  m.q = q; m.p = p; m.Pp = Pp;
  GxNet &G = m.g;
  G.L = T + 1;
  G.dim[0] = q; G.pad[0] = gx_pad32(q);
  for (int l = 0; l < T; ++l) { G.dim[l + 1] = n.lout[l]; G.pad[l + 1] = gx_pad32(n.lout[l]); }
  G.dim[T + 1] = 2 * Pp; G.pad[T + 1] = 2 * Pp;
  size_t off = 0, offT = 0;
  int wmax = 32, mb = 0, w = 0;
  for (int l = 0; l <= T; ++l) {
    const int Kp = G.pad[l], Np = G.pad[l + 1];
    G.w[l] = (int)off; off += (size_t)Kp * Np;
    G.b[l] = (int)off; off += Np;
    G.wt[l] = (int)offT; offT += (size_t)Np * Kp;
    wmax = std::max(wmax, Kp);
    if (l < T) { m.moff[l] = mb; mb += GX_ROWS * (Np >> 1); }
  }
  for (int l = 0; l < NF; ++l) {        // oracle/bnn.py sign_layout over the (in, out) pairs trunk ..., mean, var
    st->f.sin_w[l] = w; w += (n.lin[l] + 31) / 32;
    st->f.sout_w[l] = w; w += (n.lout[l] + 31) / 32;
    st->pk.lin[l] = n.lin[l]; st->pk.lout[l] = n.lout[l]; st->pk.woff[l] = n.woff[l]; st->pk.eoff[l] = n.eoff[l];
  }
  st->f.swords = (w + 3) / 4 * 4;
  m.mask_bytes = mb;
  m.ld = gx_ld(wmax);
  m.ch = std::min(Pp, (m.ld - 8) / 32 * 32);
  st->lds_bytes = gx_bgm_lds_bytes(m.ld, q, mb) + GX_ROWS * st->f.swords * 4;
  if (st->lds_bytes > 160 * 1024 || off >= (1u << 30)) { delete st; return 1; }
  for (float **p_ : {&st->pack, &st->dw}) {
    if (hipMalloc((void **)p_, sizeof(float) * off) != hipSuccess || hipMemset(*p_, 0, sizeof(float) * off) != hipSuccess) { s->gxf = st; gxf_free(s); bgm_set_error("frozen-noise HMC: device allocation failed"); return BGM_E_HIP; }
  }
  for (float **p_ : {&st->packT, &st->dwT}) {
    if (hipMalloc((void **)p_, sizeof(float) * offT) != hipSuccess || hipMemset(*p_, 0, sizeof(float) * offT) != hipSuccess) { s->gxf = st; gxf_free(s); bgm_set_error("frozen-noise HMC: device allocation failed"); return BGM_E_HIP; }
  }
  m.pack = st->pack; m.packT = st->packT; st->f.dw = st->dw; st->f.dwT = st->dwT;
  st->pk.g = G; st->pk.n_flip = NF; st->pk.Pp = Pp;
  st->pk.pack = st->pack; st->pk.packT = st->packT; st->pk.dw = st->dw; st->pk.dwT = st->dwT;
  s->gxf = st;
  g = st;
  return BGM_OK;
}

extern "C" int bgm_bvn_hmc_run(bgm_handle *h, const bgm_hmc_args *g, void *stream_) {
  int rc = bvn_need(h, "bgm_bvn_hmc_run");
  if (rc) return rc;
  if (g && g->n == 0) return BGM_OK;
  if (!g || !g->x_dev || !g->state_dev || !g->logp_dev || !g->grad_dev || !g->step_dev || g->n < 0 || g->n_iters < 0 ||
      g->n_leapfrog < 1) { bgm_set_error("bgm_bvn_hmc_run: bad argument"); return BGM_E_INVALID; }
  if (g->n_iters == 0 && !g->init) return BGM_OK;
  BgmbState *s = vst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  static const bool no_gxf = std::getenv("BGM_BVN_NO_TILES") != nullptr;      // dev A/B: the workspace kernel also for frozen noise
  if (s->cfg.hmc_frozen_noise && !no_gxf) {
    // frozen noise (the shipped default): one perturbation and one sign string per row for the whole run -> 32-chain LDS tiles with
    // the posterior means and the perturbation as two padded packs in L2 (gx_flipout.h)
    // ... or, for the reference's generator shape (hidden layers of 64 units), register-chained 16-chain row tiles with the posterior
    // means LDS-resident and the perturbation streamed through an LDS stage (bgmf_kernels.h)
    GxfState *gx;
    rc = gxf_session(s, gx);
    if (rc < 0) return rc;
    if (rc == 0) {
      hipStream_t st = (hipStream_t)stream_;
      rc = bvn_noise(h, s, g->seed, 0u, 0u, 1, false, 0u, st);              // the run's one perturbation (generator call 0), as the workspace kernel draws it
      if (rc) return rc;
      rc = bgmf_hmc_try(h, s, g, st);
      if (rc <= 0) return rc;
      GxfPackArgs pk = gx->pk;
      pk.theta = s->theta_dev; pk.dwc = s->dw_dev;
      hipLaunchKernelGGL(gxf_pack_kernel, dim3(64, pk.n_flip), dim3(256), 0, st, pk);       // packs of the CURRENT parameters (they may have been trained since)
      GxfHmcArgs k{};
      k.f = gx->f; k.f.m.bnp = s->theta_dev + s->net.off;
      k.x = g->x_dev; k.n = g->n; k.row_base = g->row_base; k.state = g->state_dev; k.logp = g->logp_dev; k.grad = g->grad_dev;
      k.init = g->init; k.it_begin = g->it_begin; k.n_iters = g->n_iters; k.burn_in = g->burn_in; k.n_leapfrog = g->n_leapfrog; k.step = g->step_dev;
      k.k0 = (uint32_t)(g->seed & 0xFFFFFFFFull); k.k1 = (uint32_t)(g->seed >> 32); k.acc_prob_sum = g->acc_prob_sum_dev; k.acc_count = g->acc_count_dev; k.draws = g->draws_dev;
      BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gxf_bgm_hmc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, gx->lds_bytes));
      const long long tiles32 = (g->n + GX_ROWS - 1) / GX_ROWS;
      const int occ = std::max(1, std::min(4, (160 * 1024) / gx->lds_bytes));
      hipLaunchKernelGGL(gxf_bgm_hmc_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(tiles32, (long long)h->n_cus * occ))), dim3(GX_THREADS),
                         gx->lds_bytes, st, k);
      BGM_HIP_CHECK(hipGetLastError());
      return BGM_OK;
    }
  }
  const int rt = bvn_rt(h, g->n);
  const long long n_tiles = (g->n + rt - 1) / rt;
  const long long tiles = std::min<long long>(n_tiles, 8LL * h->n_cus);       // workgroups (one workspace slice each)
  BgmbBigArgs a;
  bvn_big_base(s, a, g->seed);
  a.rt = rt; a.n_tiles = n_tiles;
  rc = bvn_big_ws(h, s, tiles, a.ws, a.ws_stride);
  if (rc) return rc;
  a.x = g->x_dev; a.n = g->n; a.row_base = g->row_base;
  a.state = g->state_dev; a.logp = g->logp_dev; a.grad = g->grad_dev;
  a.burn_in = g->burn_in; a.n_leapfrog = g->n_leapfrog;
  a.step = g->step_dev;
  a.frozen = s->cfg.hmc_frozen_noise ? 1 : 0;
  a.acc_prob_sum = g->acc_prob_sum_dev; a.acc_count = g->acc_count_dev; a.draws = g->draws_dev;
  a.dw_stride = bvn_dw_stride(s);
  // The perturbations of a launch's generator calls are produced once for all workgroups (bgmb_noise_kernel); a launch covers
  // as many transitions as fit a 64 MB perturbation buffer.
  const int L = g->n_leapfrog;
  const long long per_it = a.dw_stride * (long long)L * (long long)sizeof(float);
  const char *budget_env = std::getenv("BGM_BVN_NOISE_BYTES");        // test hook: force several launches per segment
  const long long budget = budget_env ? std::max<long long>(1, std::atoll(budget_env)) : (64LL << 20);
  const int chunk = a.frozen ? g->n_iters : (int)std::max<long long>(1, budget / std::max<long long>(1, per_it));
  int it = g->it_begin, left = g->n_iters;
  bool first = true;
  do {
    const int c = std::min(left, std::max(chunk, 1));
    a.init = (first && g->init) ? 1 : 0;
    a.it_begin = it; a.n_iters = c;
    if (a.frozen) rc = bvn_noise(h, s, g->seed, 0u, 0u, 1, false, 0u, (hipStream_t)stream_);
    else rc = bvn_noise(h, s, g->seed, 1u + (uint32_t)it * (uint32_t)L, 1u, c * L, a.init != 0, 0u, (hipStream_t)stream_);
    if (rc) return rc;
    a.dw = s->dw_dev;
    rc = 1;
    if (!a.frozen) {       // fresh noise on the row-tile chains (bgmf_kernels.h) for the reference's generator shape, else the workspace kernel
      rc = bgmf_hmc_fresh(h, s, g, it, c, a.init, a.dw_stride, (hipStream_t)stream_);
      if (rc < 0) return rc;
    }
    if (rc == 1) {
      hipLaunchKernelGGL(bgmb_hmc_kernel, dim3((unsigned)tiles), dim3(BNN_THREADS), 0, (hipStream_t)stream_, a);
      BGM_HIP_CHECK(hipGetLastError());
    }
    it += c; left -= c; first = false;
  } while (left > 0);
  return BGM_OK;
}

extern "C" int bgm_bvn_set_precision(bgm_handle *h, int32_t mode) {
  int rc = bvn_need(h, "bgm_bvn_set_precision");
  if (rc) return rc;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  return bgmf_set_precision(vst(h), mode);
}

extern "C" int bgm_bvn_decode(bgm_handle *h, const float *draws_dev, int64_t n, int64_t row_base, int32_t n_draws, int32_t burn_in,
                              uint64_t seed, uint32_t stream_id, uint32_t sign_stride, uint32_t sign_off, const int32_t *slot_dev,
                              int32_t k_slots, float *cells_dev, float *full_dev, float *var_full_dev, int32_t add_noise,
                              void *stream_) {
  int rc = bvn_need(h, "bgm_bvn_decode");
  if (rc) return rc;
  const long long total = (long long)n * n_draws;
  if (total == 0) return BGM_OK;
  if (!draws_dev || n < 0 || n_draws < 0 || (cells_dev && (!slot_dev || k_slots < 1))) {
    bgm_set_error("bgm_bvn_decode: bad argument"); return BGM_E_INVALID;
  }
  if ((long long)n_draws * (long long)sign_stride + (long long)sign_off + n >= (1LL << 32)) {
    bgm_set_error("bgm_bvn_decode: sign row ids must stay below 2^32 (split the rows)"); return BGM_E_UNSUPPORTED;
  }
  BgmbState *s = vst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  const long long tiles = std::min<long long>(((n + BGMB_RT - 1) / BGMB_RT) * (long long)n_draws, 4LL * h->n_cus);
  BgmbDecodeArgs a;
  std::memset(&a, 0, sizeof(a));
  a.net = s->net; a.net.bn_fixed = 2;
  a.theta = s->theta_dev; a.q = s->q; a.p = s->p;
  rc = bvn_big_ws(h, s, tiles, a.ws, a.ws_stride);
  if (rc) return rc;
  a.draws = draws_dev; a.n = n; a.row_base = row_base; a.n_draws = n_draws; a.burn_in = burn_in;
  a.k0 = a.x0 = (uint32_t)(seed & 0xFFFFFFFFull); a.k1 = a.x1 = (uint32_t)(seed >> 32); a.stream = stream_id;
  a.sign_stride = sign_stride; a.sign_off = sign_off;
  rc = bvn_noise(h, s, seed, stream_id, 0u, 1, false, 0u, (hipStream_t)stream_);
  if (rc) return rc;
  a.dw = s->dw_dev;
  a.slot = slot_dev; a.k_slots = k_slots; a.cells = cells_dev; a.full = full_dev; a.var_full = var_full_dev; a.add_noise = add_noise;
  hipLaunchKernelGGL(bgmb_decode_kernel, dim3((unsigned)tiles), dim3(BNN_THREADS), 0, (hipStream_t)stream_, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bvn_end(bgm_handle *h, void *stream_) {
  if (!h) { bgm_set_error("bgm_bvn_end: NULL handle"); return BGM_E_INVALID; }
  if (!h->bvn_state) return BGM_OK;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  bgm_bvn_free_state(h);
  return BGM_OK;
}
