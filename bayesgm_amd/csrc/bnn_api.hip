// bnn_api.hip -- C ABI of the Bayesian-network (use_bnn=True) CausalBGM path: session, minibatch steps.
#include <cmath>
#include <cstring>
#include <vector>

#include "bgm_host.h"
#include "comm_host.h"
#include <cstdlib>

#include "bnn_kernels.h"
#include "bnn_state.h"
#include "egm_chain_bnn.h"

// ---- iterative-update steps as row-tile chains (egm_chain_bnn.h) when the shapes are the compiled ones
struct BnnFitChain {
  int ntl = 0, n_tiles = 0;
  int t0 = 1;               // latent input tiles of g (q <= 16 t0); two tiles: the masked 13-tile kernels, B = 32
  bool pad = false;         // the 13-tile kernels on a narrower p + 1 (B = 32 only)
  EcbTab *tab_theta = nullptr, *tab_z = nullptr;
  int *tiles_theta = nullptr;
  float *ws = nullptr;
  float *ws_z = nullptr;     // the latent phase's own workspace: it may run beside the next minibatch's theta phase (bgm_bnn_fit_epoch)
  // bgm_bnn_fit_epoch, device-side ordering (fit_sync.h) of the next launches: the gradient-tile kernel (waits for the previous latent
  // phase, counts the Adam step done), the latent phase's noise kernel (waits for the Adam step), its row-update kernel (counts it done)
  FitSync sync_dw{}, sync_zn{}, sync_zr{};
  // bgm_bnn_fit_epoch, noise off the critical path (EcbRider / EcbAhead, egm_chain_bnn.h): both workspaces exist twice (a phase reads one
  // while riders fill the other); `rider` / `ahead` are what the next theta launch carries, `theta_prepared` / `z_prepared`: the next
  // theta / latent phase finds its noise in its workspace (no noise launch)
  float *ws_ring[2] = {nullptr, nullptr}, *ws_z_ring[2] = {nullptr, nullptr};
  size_t ws_floats = 0;
  int kl_cnt_off = 0;
  int dw_t[4] = {0, 0, 0, 0}, dw_z[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};      // EcbAhead: perturbation offsets of each net's calls
  EcbRider rider{};
  EcbAhead ahead{};
  bool riders_on = false, theta_prepared = false, z_prepared = false;
};
template <bool KL>
static __global__ __launch_bounds__(BNN_THREADS) void bnn_fit_noise_kernel(BnnArgs a, const EcbTab *tab, float *ws, FitSync sy) {
  __shared__ float red[16];
  fit_sync_wait(sy);
  ecb_gen_noise<BnnArgs>(a, *tab, ws);
  if (blockIdx.x == 0 && threadIdx.x < (KL ? 6 : 1) && a.out) a.out[threadIdx.x] = 0.0f;      // the chain workgroups accumulate into it
  if (KL) {     // theta step: the value of the KL term, summed by the chain kernel (one call per net there)
    const int c = blockIdx.x / ECB_NOISE_PARTS, part = blockIdx.x % ECB_NOISE_PARTS;
    ecb_kl_partial(a.theta, a.net[tab->c[c].net], ws + tab->klp + blockIdx.x, part, ECB_NOISE_PARTS, red);
  }
}
// rd.kl_cnt != NULL: workgroups blockIdx.y >= 3 are riders (ecb_rider), two per grid row
template <int NTL, int NB, bool PAD = false, int T0 = 1, bool WS = false>
static __global__ __launch_bounds__(BNN_THREADS) void bnn_theta_chain_kernel(BnnArgs a, const EcbTab *tab, float *ws, EcbRider rd) {
  extern __shared__ __attribute__((aligned(16))) float bnn_chain_lds[];
  if (blockIdx.y >= 3) {
    ecb_rider<BnnArgs>(a, *tab, ws, rd, 2 * ((int)blockIdx.y - 3) + (int)blockIdx.x, bnn_chain_lds);
    return;
  }
  ecb_theta_chain<BnnArgs, 4, NTL, 4, 2, 1, NB, PAD, T0, WS>(a, *tab, ws, bnn_chain_lds, rd.kl_cnt);
}
template <int NB>
static __global__ __launch_bounds__(BNN_THREADS) void bnn_theta_dw_kernel(BnnArgs a, const EcbTab *tab, const int *tiles, const float *ws, FitSync sy,
                                                                       EcbAhead ah) {
  ecb_theta_dw<BnnArgs, NB>(a, *tab, tiles, ws, ah, &sy);      // (waits inside, in front of its parameter stores)
  fit_sync_done(sy);
}
template <int NTL, int NB, bool PAD = false, int T0 = 1, bool WS = false>
static __global__ __launch_bounds__(BNN_THREADS) void bnn_z_chain_kernel(BnnArgs a, const EcbTab *tab, float *ws, FitSync sy, EcbZRows zr) {
  extern __shared__ __attribute__((aligned(16))) float bnn_chain_lds[];
  fit_sync_wait(sy);              // (noise prepared ahead: the wait for the Adam step the noise launch otherwise carries)
  ecb_z_chain<BnnArgs, 4, NTL, 4, 2, 1, NB, PAD, T0, WS>(a, *tab, ws, bnn_chain_lds, zr);
  if (zr.zm) fit_sync_done(sy);   // (rows updated here: the latent phase of this minibatch is complete)
}
static void bnn_chain_free(BnnState *s) {
  BnnFitChain *c = static_cast<BnnFitChain *>(s->chain);
  if (!c) return;
  for (void *p : {(void *)c->tab_theta, (void *)c->tab_z, (void *)c->tiles_theta, (void *)c->ws_ring[0], (void *)c->ws_z_ring[0], (void *)c->ws_ring[1],
                  (void *)c->ws_z_ring[1]})
    if (p) hipFree(p);
  delete c;
  s->chain = nullptr;
}
static int bnn_chain_setup(BnnState *s) {
  if (std::getenv("BGM_BNN_NO_CHAIN") || !ecb_shapes_ok(s->net, s->q, s->p, false)) return BGM_OK;
  const int ntl_need = (s->p + 1 + 15) / 16, B = 32;
  const int t0 = s->q <= 16 ? 1 : 2;
  const int ntl = ((ntl_need == 13 || ntl_need == 7) && t0 == 1) ? ntl_need : 13;
  BnnFitChain *c = new BnnFitChain();
  s->chain = c;
  c->ntl = ntl;
  c->t0 = t0;
  c->pad = ntl != ntl_need || t0 == 2;
  EcbTab tt{}, tz{};
  std::vector<int> tiles, none;
  const int tnet[3] = {BNN_G, BNN_H, BNN_F}, tso[3] = {0, 0, 0}, trained[3] = {BNN_G, BNN_H, BNN_F};
  size_t w1 = ecb_build_tab(s->net, tnet, tso, 3, B, ntl, tt, tiles, trained, 3);
  tt.klp = (int)w1; w1 += 3 * ECB_NOISE_PARTS + 16;
  const int znet[6] = {BNN_G, BNN_G, BNN_H, BNN_H, BNN_F, BNN_F}, zso[6] = {0, 1, 0, 1, 0, 1};
  const size_t w2 = ecb_build_tab(s->net, znet, zso, 6, B, ntl, tz, none, trained, 0);
  c->n_tiles = tt.n_tiles;
  const size_t wsf = std::max(w1, w2) + 64;
  for (int r = 0; r < 2; ++r) {
    BGM_HIP_CHECK(hipMalloc((void **)&c->ws_ring[r], sizeof(float) * wsf));
    BGM_HIP_CHECK(hipMemset(c->ws_ring[r], 0, sizeof(float) * wsf));
    BGM_HIP_CHECK(hipMalloc((void **)&c->ws_z_ring[r], sizeof(float) * wsf));
    BGM_HIP_CHECK(hipMemset(c->ws_z_ring[r], 0, sizeof(float) * wsf));
  }
  c->ws = c->ws_ring[0]; c->ws_z = c->ws_z_ring[0];
  c->ws_floats = wsf;
  for (int k = 0; k < 4; ++k) {
    if (tt.net_ncalls[k] > 0) c->dw_t[k] = tt.c[tt.net_calls[k][0]].dW;
    for (int cc = 0; cc < tz.net_ncalls[k] && cc < 2; ++cc) c->dw_z[k][cc] = tz.c[tz.net_calls[k][cc]].dW;
  }
  c->kl_cnt_off = tt.klp + 3 * ECB_NOISE_PARTS + 4;          // (a word of the 16 spare ones behind the KL partial sums)
  BGM_HIP_CHECK(hipMalloc((void **)&c->tab_theta, sizeof(EcbTab)));
  BGM_HIP_CHECK(hipMalloc((void **)&c->tab_z, sizeof(EcbTab)));
  BGM_HIP_CHECK(hipMemcpy(c->tab_theta, &tt, sizeof(EcbTab), hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMemcpy(c->tab_z, &tz, sizeof(EcbTab), hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMalloc((void **)&c->tiles_theta, sizeof(int) * tiles.size()));
  BGM_HIP_CHECK(hipMemcpy(c->tiles_theta, tiles.data(), sizeof(int) * tiles.size(), hipMemcpyHostToDevice));
  return BGM_OK;
}

static BnnState *bst(bgm_handle *h) { return static_cast<BnnState *>(h->bnn_state); }

void bgm_bnn_free_state(bgm_handle *h) {
  if (!h->bnn_state) return;
  BnnState *s = bst(h);
  if (s->dev) hipFree(s->dev);
  if (s->kl_part_dev) hipFree(s->kl_part_dev);
  if (s->tlast_dev) hipFree(s->tlast_dev);
  bnn_free_sampler(s);
  bgm_bnn_egm_free(s->egm);
  bnn_chain_free(s);
  delete s;
  h->bnn_state = nullptr;
}

static int bnn_fill(const bgm_bnn_config *cfg, BnnNet net[4], int64_t offsets[5]) {
  if (!cfg) return BGM_E_INVALID;
  const int q = cfg->z_dims[0] + cfg->z_dims[1] + cfg->z_dims[2] + cfg->z_dims[3], p = cfg->v_dim;
  if (q < 1 || p < 1 || cfg->z_dims[0] < 0 || cfg->z_dims[1] < 0 || cfg->z_dims[2] < 0 || cfg->z_dims[3] < 0) return BGM_E_INVALID;
  const int in[4] = {q, p, cfg->z_dims[0] + cfg->z_dims[1] + 1, cfg->z_dims[0] + cfg->z_dims[2]};
  const int out[4] = {p + 1, q, 2, 2};
  int off = 0;
  for (int k = 0; k < 4; ++k) {
    const int nh = cfg->n_hidden[k];
    if (nh < 1 || nh + 1 > BNN_MAX_LAYERS || in[k] < 1) return BGM_E_INVALID;
    BnnNet &n = net[k];
    std::memset(&n, 0, sizeof(n));
    n.n_layers = nh + 1;
    n.dims[0] = in[k];
    for (int i = 0; i < nh; ++i) { if (cfg->units[k][i] < 1) return BGM_E_INVALID; n.dims[i + 1] = cfg->units[k][i]; }
    n.dims[nh + 1] = out[k];
    n.off = off;
    n.net_id = k;
    n.bn_fixed = cfg->norm_mode == 1 ? 1 : 0;
    bnn_finish_net(n);
    offsets[k] = off;
    off += n.n_params;
  }
  offsets[4] = off;
  return BGM_OK;
}

extern "C" int bgm_bnn_layout(const bgm_bnn_config *cfg, int64_t offsets[5]) {
  BnnNet net[4];
  if (!offsets || bnn_fill(cfg, net, offsets)) { bgm_set_error("bgm_bnn_layout: bad configuration"); return BGM_E_INVALID; }
  return BGM_OK;
}

extern "C" int bgm_bnn_begin(bgm_handle *h, const bgm_bnn_config *cfg, const float *theta_host, int64_t count, void *stream_) {
  (void)stream_;
  if (!h) { bgm_set_error("bgm_bnn_begin: NULL handle"); return BGM_E_INVALID; }
  if (!cfg || !theta_host) { bgm_set_error("bgm_bnn_begin: NULL argument"); return BGM_E_INVALID; }
  if (cfg->norm_mode != 0 && cfg->norm_mode != 1) { bgm_set_error("bgm_bnn_begin: norm_mode must be 0 or 1"); return BGM_E_INVALID; }
  if (cfg->max_batch < 2 || cfg->max_batch > BNN_MAX_BATCH) { bgm_set_error("bgm_bnn_begin: max_batch must be in [2, 4096]"); return BGM_E_UNSUPPORTED; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  bgm_bnn_free_state(h);
  BnnState *s = new BnnState();
  h->bnn_state = s;
  s->cfg = *cfg;
  int64_t offs[5];
  if (bnn_fill(cfg, s->net, offs)) { bgm_bnn_free_state(h); bgm_set_error("bgm_bnn_begin: bad configuration"); return BGM_E_INVALID; }
  s->n_params = (int)offs[4];
  if (count != offs[4]) { bgm_bnn_free_state(h); bgm_set_error("bgm_bnn_begin: wrong parameter count"); return BGM_E_INVALID; }
  s->q = s->net[BNN_G].dims[0];
  s->p = cfg->v_dim;
  int wmax = 0;
  long long cache_max = 0;
  const int B = cfg->max_batch;
  for (int k = 0; k < 4; ++k) {
    const BnnNet &n = s->net[k];
    for (int i = 0; i <= n.n_layers; ++i) wmax = std::max(wmax, n.dims[i]);
    long long c = (long long)B * n.dims[0] + n.dims[0] + 2LL * B * n.hoff[n.n_layers + 1] + 2LL * n.eoff[n.n_layers] + (long long)B * n.swords + 64;
    cache_max = std::max(cache_max, c);
  }
  s->wmax = wmax;
  const long long gather = (long long)B * (s->q + s->p + 2 + s->net[BNN_F].dims[0] + s->net[BNN_H].dims[0]) + 64;
  s->ws_stride = (gather + 6LL * B * wmax + 64 + 2 * cache_max + 63) & ~63LL;      // one slice per workgroup (net)
  s->ws_floats = (size_t)(9 * s->ws_stride);      // three for the theta step's nets, six for the latent step's (net, call) pairs (the two steps overlap in bgm_bnn_fit_epoch)
  const size_t np = ((size_t)s->n_params + 63) & ~(size_t)63;
  const size_t total = 4 * np + s->ws_floats + 64 + 8 * ((size_t)B * s->q + 64);
  BGM_HIP_CHECK(hipMalloc((void **)&s->dev, sizeof(float) * total));
  BGM_HIP_CHECK(hipMemset(s->dev, 0, sizeof(float) * total));
  s->theta_dev = s->dev; s->m_dev = s->dev + np; s->v_dev = s->dev + 2 * np; s->grad_dev = s->dev + 3 * np;
  s->ws_dev = s->dev + 4 * np;
  s->out_dev = s->ws_dev + s->ws_floats;
  s->dz_dev = s->out_dev + 64;
  s->dz_part_dev = s->dz_dev + (size_t)B * s->q + 64;      // [6][B x q] + 3 loss partials
  BGM_HIP_CHECK(hipMemcpy(s->theta_dev, theta_host, sizeof(float) * count, hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMalloc((void **)&s->kl_part_dev, sizeof(float) * 3 * BNN_KL_PARTS));
  BGM_HIP_CHECK(hipMemset(s->kl_part_dev, 0, sizeof(float) * 3 * BNN_KL_PARTS));
  s->t_theta = 0; s->t_z = 0;
  return bnn_chain_setup(s);
}

static int bnn_need(bgm_handle *h, const char *who) {
  if (!h || !h->bnn_state) { bgm_set_error(std::string(who) + ": no session (bgm_bnn_begin)"); return BGM_E_STATE; }
  return BGM_OK;
}

static float *bnn_what(BnnState *s, int what) {
  return what == 0 ? s->theta_dev : what == 1 ? s->grad_dev : what == 2 ? s->m_dev : what == 3 ? s->v_dev : nullptr;
}

extern "C" int bgm_bnn_read(bgm_handle *h, int32_t what, float *host, int64_t count, void *stream_) {
  int rc = bnn_need(h, "bgm_bnn_read");
  if (rc) return rc;
  BnnState *s = bst(h);
  float *src = bnn_what(s, what);
  if (!src || !host || count != s->n_params) { bgm_set_error("bgm_bnn_read: bad argument"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  BGM_HIP_CHECK(hipMemcpy(host, src, sizeof(float) * count, hipMemcpyDeviceToHost));
  return BGM_OK;
}

extern "C" int bgm_bnn_write(bgm_handle *h, int32_t what, const float *host, int64_t count, void *stream_) {
  int rc = bnn_need(h, "bgm_bnn_write");
  if (rc) return rc;
  BnnState *s = bst(h);
  float *dst = bnn_what(s, what);
  if (!dst || !host || count != s->n_params) { bgm_set_error("bgm_bnn_write: bad argument"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  BGM_HIP_CHECK(hipMemcpy(dst, host, sizeof(float) * count, hipMemcpyHostToDevice));
  s->packed_valid = false; s->bnf_valid = false;
  return BGM_OK;
}

extern "C" int bgm_bnn_grad_dev(bgm_handle *h, float **grad_dev, int64_t *count) {
  int rc = bnn_need(h, "bgm_bnn_grad_dev");
  if (rc) return rc;
  if (grad_dev) *grad_dev = bst(h)->grad_dev;
  if (count) *count = bst(h)->n_params;
  return BGM_OK;
}

extern "C" int bgm_bnn_grad_exchange(bgm_handle *h, float *buf_dev, int32_t to_session, void *stream_) {
  int rc = bnn_need(h, "bgm_bnn_grad_exchange");
  if (rc) return rc;
  if (!buf_dev) { bgm_set_error("bgm_bnn_grad_exchange: NULL buffer"); return BGM_E_INVALID; }
  BnnState *s = bst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipMemcpyAsync(to_session ? s->grad_dev : buf_dev, to_session ? buf_dev : s->grad_dev, sizeof(float) * s->n_params,
                               hipMemcpyDeviceToDevice, (hipStream_t)stream_));
  return BGM_OK;
}

static float adam_lr_t(float lr, long long t_) {
  const double t = (double)t_;
  return (float)((double)lr * std::sqrt(1.0 - std::pow((double)BNN_ADAM_B2, t)) / (1.0 - std::pow((double)BNN_ADAM_B1, t)));
}

static void bnn_base_args(BnnState *s, BnnArgs &a, int batch, int batch_global, uint64_t seed, uint32_t stream_id) {
  for (int k = 0; k < 4; ++k) a.net[k] = s->net[k];
  a.theta = s->theta_dev; a.m = s->m_dev; a.v = s->v_dev; a.grad = s->grad_dev;
  a.B = batch; a.q = s->q; a.p = s->p;
  a.z0 = s->cfg.z_dims[0]; a.z1 = s->cfg.z_dims[1]; a.z2 = s->cfg.z_dims[2];
  a.binary = s->cfg.binary_treatment; a.wmax = s->wmax; a.kl_weight = s->cfg.kl_weight;
  a.k0 = (uint32_t)(seed & 0xFFFFFFFFull); a.k1 = (uint32_t)(seed >> 32); a.stream = stream_id;
  a.inv_B = 1.0f / (float)(batch_global > 0 ? batch_global : batch);
  a.sig2[0] = s->cfg.sigma_v > 0.0f ? s->cfg.sigma_v * s->cfg.sigma_v : 0.0f;
  a.sig2[1] = s->cfg.sigma_x > 0.0f ? s->cfg.sigma_x * s->cfg.sigma_x : 0.0f;
  a.sig2[2] = s->cfg.sigma_y > 0.0f ? s->cfg.sigma_y * s->cfg.sigma_y : 0.0f;
  // data parallel: every rank adds its share of the KL term, the all-reduce (sum) restores kl_weight * KL
  if (batch_global > batch) a.kl_weight = s->cfg.kl_weight * (float)batch / (float)batch_global;
  a.ws = s->ws_dev; a.ws_stride = s->ws_stride;
  a.dz_part = s->dz_part_dev; a.loss_part = s->dz_part_dev + 6 * (size_t)s->cfg.max_batch * s->q;
}

#ifdef BNN_PROF
// development: every 100 launches of bnn_theta_step_kernel, where the profiled workgroup's cycles went (bnn_kernels.h BNN_T)
static void bnn_prof_report(hipStream_t st) {
  static int calls = 0;
  if (++calls % 100) return;
  unsigned long long acc[8], zero[8] = {0}, sp[8];
  hipStreamSynchronize(st);
  hipMemcpyFromSymbol(acc, HIP_SYMBOL(bnn_prof_acc), sizeof(acc));
  hipMemcpyToSymbol(HIP_SYMBOL(bnn_prof_acc), zero, sizeof(zero));
  hipMemcpyFromSymbol(sp, HIP_SYMBOL(bnn_prof_span), sizeof(sp));
  double tot = 0; for (int i = 0; i < 7; ++i) tot += (double)acc[i];
  fprintf(stderr, "BNN_PROF last call, workgroups g / h / f on the 100 MHz counter: start +%.1f +%.1f +%.1f us, end +%.1f +%.1f +%.1f us after g's start\n",
          0.0, ((double)sp[2] - (double)sp[0]) / 100.0, ((double)sp[4] - (double)sp[0]) / 100.0, ((double)sp[1] - (double)sp[0]) / 100.0,
          ((double)sp[3] - (double)sp[0]) / 100.0, ((double)sp[5] - (double)sp[0]) / 100.0);
  fprintf(stderr, "BNN_PROF shader clock during the kernel: %.2f GHz (%.1f us per call on the 100 MHz counter)\n", tot / ((double)acc[7] * 10.0), (double)acc[7] / 100.0 / 100.0);
  fprintf(stderr, "BNN_PROF theta step, workgroup %d (100 calls, %.0f cycles each): gather %.3f noise %.3f forward %.3f loss %.3f backward %.3f KL %.3f Adam %.3f\n",
          BNN_PROF_WG, tot / 100, acc[0] / tot, acc[1] / tot, acc[2] / tot, acc[3] / tot, acc[4] / tot, acc[5] / tot, acc[6] / tot);
}
#else
static void bnn_prof_report(hipStream_t) {}
#endif

// The launches of one theta step.  parts & 1: noise + forward / backward chains (read the parameters); parts & 2: the gradient tiles
// and, with a.apply, the Adam step on them (write the parameters).  bgm_bnn_fit_epoch puts a stream dependency between the two.
static void bnn_theta_launch(BnnState *s, BnnArgs &a, int batch, int parts, hipStream_t st) {
  BnnFitChain *fc = static_cast<BnnFitChain *>(s->chain);
  if (fc && (batch == 32 || (batch == 16 && !fc->pad))) {
    if (parts & 1) {
    if (!fc->theta_prepared)
    hipLaunchKernelGGL(bnn_fit_noise_kernel<true>, dim3(3 * ECB_NOISE_PARTS), dim3(BNN_THREADS), 0, st, a, fc->tab_theta, fc->ws, FitSync{});
    auto kc = fc->t0 == 2 ? bnn_theta_chain_kernel<13, 2, true, 2> : fc->pad ? bnn_theta_chain_kernel<13, 2, true>
              : batch == 32 ? (fc->ntl == 13 ? bnn_theta_chain_kernel<13, 2> : bnn_theta_chain_kernel<7, 2>)
                          : (fc->ntl == 13 ? bnn_theta_chain_kernel<13, 1> : bnn_theta_chain_kernel<7, 1>);
    static const bool one_wg = std::getenv("BGM_FIT_ONE_WG") != nullptr;
    if (batch == 32 && !fc->pad && fc->t0 == 1 && !one_wg) {      // row tiles and networks on their own workgroups (see ecb_theta_chain)
      static const bool no_ws = std::getenv("BGM_FIT_NO_WORKERS") != nullptr;
      if (no_ws) {
        auto ks = fc->ntl == 13 ? bnn_theta_chain_kernel<13, 1> : bnn_theta_chain_kernel<7, 1>;
        hipLaunchKernelGGL(ks, dim3(2, 3), dim3(BNN_THREADS), 64 * sizeof(float), st, a, fc->tab_theta, fc->ws, EcbRider{});
      } else {      // g's last layer over the idle waves of its workgroup (ecb_theta_chain<WS>); bgm_bnn_fit_epoch: + the rider workgroups
        auto ks = fc->ntl == 13 ? bnn_theta_chain_kernel<13, 1, false, 1, true> : bnn_theta_chain_kernel<7, 1, false, 1, true>;
        hipLaunchKernelGGL(ks, dim3(2, fc->riders_on ? 3 + ECB_RIDERS / 2 : 3), dim3(BNN_THREADS), ECB_WS_LDS_FLOATS * sizeof(float), st, a, fc->tab_theta,
                           fc->ws, fc->riders_on ? fc->rider : EcbRider{});
      }
    } else
      hipLaunchKernelGGL(kc, dim3(1), dim3(BNN_THREADS), 64 * sizeof(float), st, a, fc->tab_theta, fc->ws, EcbRider{});
    }
    if (parts & 2) {
    auto kd = batch == 32 ? bnn_theta_dw_kernel<2> : bnn_theta_dw_kernel<1>;
    hipLaunchKernelGGL(kd, dim3((fc->n_tiles + ECH_WAVES - 1) / ECH_WAVES + 1), dim3(BNN_THREADS), 0, st, a, fc->tab_theta, fc->tiles_theta, fc->ws, fc->sync_dw,
                       fc->riders_on ? fc->ahead : EcbAhead{});
    }
  } else {
    // general widths: one workgroup per net walks the layer products; the elementwise parts (the call's eps / dW, the KL terms, the Adam
    // step: half of the step's time at [256] x 3 when the one workgroup did them too) run as their own launches over the chip.
    // parts & 1 reads the parameters (noise, step kernel, gradient tiles), parts & 2 writes them (KL gradients + Adam in one launch).
    // BGM_BNN_STEP_ONE_LAUNCH: the one-launch phase machine, at the point of the parameter write.
    static const bool one_launch = std::getenv("BGM_BNN_STEP_ONE_LAUNCH") != nullptr;
    a.wide = (!one_launch && s->kl_part_dev) ? 1 : 0;
    a.kl_part = s->kl_part_dev;
    if (!a.wide) {
      if (parts & 2) { hipLaunchKernelGGL(bnn_theta_step_kernel, dim3(3), dim3(BNN_THREADS), 0, st, a); bnn_prof_report(st); }
      return;
    }
    if (parts & 1) {
      hipLaunchKernelGGL(bnn_step_noise_kernel, dim3(BNN_NOISE_PARTS, 3, 1), dim3(BNN_THREADS), 0, st, a, 4, 0);
      static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(bnn_theta_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)BNN_R32_LDS_BYTES) == hipSuccess;
      (void)lds_ok;
      hipLaunchKernelGGL(bnn_theta_step_kernel, dim3(3), dim3(BNN_THREADS), BNN_R32_LDS_BYTES, st, a);
      hipLaunchKernelGGL(bnn_dw_kernel, dim3(BNN_DW_PARTS, 3), dim3(BNN_THREADS), 0, st, a);      // the parameter-gradient tiles of all layers
    }
    if (parts & 2) {      // KL gradients and (apply) Adam of every parameter: one launch; the KL terms join the reported losses
      hipLaunchKernelGGL(bnn_kl_adam_kernel, dim3(BNN_KL_PARTS, 3), dim3(256), 0, st, a);
      hipLaunchKernelGGL(bnn_kl_finish_kernel, dim3(1), dim3(64), 0, st, a);
    }
    if (parts & 1) bnn_prof_report(st);
  }
}

extern "C" int bgm_bnn_theta_step(bgm_handle *h, const float *data_z, const int32_t *idx, const float *x, const float *y,
                                  const float *v, int32_t batch, int32_t batch_global, float lr_theta, uint64_t seed,
                                  uint32_t stream_id, int32_t apply, float *out, void *stream_) {
  int rc = bnn_need(h, "bgm_bnn_theta_step");
  if (rc) return rc;
  BnnState *s = bst(h);
  if (!data_z || !idx || !x || !y || !v) { bgm_set_error("bgm_bnn_theta_step: NULL argument"); return BGM_E_INVALID; }
  if (batch < 2 || batch > s->cfg.max_batch) { bgm_set_error("bgm_bnn_theta_step: batch outside [2, max_batch]"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BnnArgs a{};
  bnn_base_args(s, a, batch, batch_global, seed, stream_id);
  a.data_z = data_z; a.idx = idx; a.x_ = x; a.y_ = y; a.v_ = v;
  a.apply = apply; a.out = out;
  if (apply) { s->t_theta += 1; a.adam = BnnAdam{adam_lr_t(lr_theta, s->t_theta), BNN_ADAM_B1, BNN_ADAM_B2, BNN_ADAM_EPS}; }
  bnn_theta_launch(s, a, batch, 3, (hipStream_t)stream_);
  BGM_HIP_CHECK(hipGetLastError());
  if (apply) { s->packed_valid = false; s->bnf_valid = false; }
  return BGM_OK;
}

extern "C" int bgm_bnn_theta_apply(bgm_handle *h, float lr_theta, void *stream_) {
  int rc = bnn_need(h, "bgm_bnn_theta_apply");
  if (rc) return rc;
  BnnState *s = bst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  s->t_theta += 1;
  const BnnAdam ad{adam_lr_t(lr_theta, s->t_theta), BNN_ADAM_B1, BNN_ADAM_B2, BNN_ADAM_EPS};
  for (int k : {BNN_G, BNN_H, BNN_F}) {   // the encoder is not trained by the iterative updates
    const BnnNet &n = s->net[k];
    hipLaunchKernelGGL(bnn_adam_kernel, dim3((n.n_params + 255) / 256), dim3(256), 0, (hipStream_t)stream_, s->theta_dev + n.off,
                       s->m_dev + n.off, s->v_dev + n.off, s->grad_dev + n.off, n.n_params, ad);
  }
  BGM_HIP_CHECK(hipGetLastError());
  s->packed_valid = false; s->bnf_valid = false;
  return BGM_OK;
}

extern "C" int bgm_bnn_z_step(bgm_handle *h, const float *x, const float *y, const float *v, float *data_z, float *zm, float *zv,
                              const int32_t *idx, int64_t n_rows, int32_t batch, int32_t batch_global, float lr_z, int32_t lazy,
                              uint64_t seed, uint32_t stream_id, float *out, float *dz_out, void *stream_) {
  int rc = bnn_need(h, "bgm_bnn_z_step");
  if (rc) return rc;
  BnnState *s = bst(h);
  if (!data_z || !idx || !x || !y || !v) { bgm_set_error("bgm_bnn_z_step: NULL argument"); return BGM_E_INVALID; }
  if (batch < 2 || batch > s->cfg.max_batch) { bgm_set_error("bgm_bnn_z_step: batch outside [2, max_batch]"); return BGM_E_INVALID; }
  if (!dz_out && (!zm || !zv)) { bgm_set_error("bgm_bnn_z_step: NULL Adam slots"); return BGM_E_INVALID; }
  if (!dz_out) {
    if (lazy < 0 || lazy > 2) { bgm_set_error("bgm_bnn_z_step: lazy must be 0 (dense), 1 (batch rows only) or 2 (replay)"); return BGM_E_INVALID; }
    if (lazy == 2 && (!s->tlast_dev || s->tlast_rows != n_rows || s->z_synced != s->t_z + 1)) {
      bgm_set_error("bgm_bnn_z_step: lazy = 2 needs bgm_bnn_z_sync on this minibatch's rows first (before its theta steps)");
      return BGM_E_STATE;
    }
    if (lazy != 2 && s->tlast_dev && s->z_synced != -2) {
      bgm_set_error("bgm_bnn_z_step: rows have pending replay steps; flush with bgm_bnn_z_sync(idx = NULL) before changing mode");
      return BGM_E_STATE;
    }
  }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BnnArgs a{};
  bnn_base_args(s, a, batch, batch_global, seed, stream_id);
  a.data_z = data_z; a.idx = idx; a.x_ = x; a.y_ = y; a.v_ = v;
  a.out = out; a.dz = dz_out ? dz_out : s->dz_dev;
  BnnFitChain *fc = static_cast<BnnFitChain *>(s->chain);
  bool fused_rows = false;
  if (fc && (batch == 32 || (batch == 16 && !fc->pad))) {
    FitSync zsy = fc->z_prepared ? fc->sync_zn : FitSync{};      // (prepared noise: the chain kernel itself waits for the Adam step)
    // ... and (one row tile per workgroup, batch rows only) applies the rows' Adam step in its epilogue and counts the phase done
    fused_rows = fc->z_prepared && !dz_out && lazy != 0 && batch == 32 && !fc->pad && fc->t0 == 1;
    EcbZRows zr{};
    if (fused_rows) {
      zr = EcbZRows{data_z, zm, zv, adam_lr_t(lr_z, s->t_z + 1), BNN_ADAM_B1, BNN_ADAM_B2, BNN_ADAM_EPS, lazy == 2 ? s->tlast_dev : nullptr, (int)(s->t_z + 1)};
      zsy.done_ctr = fc->sync_zr.done_ctr;
    }
    if (!fc->z_prepared)
    hipLaunchKernelGGL(bnn_fit_noise_kernel<false>, dim3(6 * ECB_NOISE_PARTS), dim3(BNN_THREADS), 0, stream, a, fc->tab_z, fc->ws_z, fc->sync_zn);
    auto kc = fc->t0 == 2 ? bnn_z_chain_kernel<13, 2, true, 2> : fc->pad ? bnn_z_chain_kernel<13, 2, true>
              : batch == 32 ? (fc->ntl == 13 ? bnn_z_chain_kernel<13, 2> : bnn_z_chain_kernel<7, 2>)
                          : (fc->ntl == 13 ? bnn_z_chain_kernel<13, 1> : bnn_z_chain_kernel<7, 1>);
    static const bool one_wg = std::getenv("BGM_FIT_ONE_WG") != nullptr;
    const size_t lds_z = (32 + 2 * batch + 4 * 16 * fc->t0 * batch) * sizeof(float);
    if (batch == 32 && !fc->pad && fc->t0 == 1 && !one_wg) {      // one row tile per workgroup (see ecb_z_chain)
      static const bool no_ws = std::getenv("BGM_FIT_NO_WORKERS") != nullptr;
      if (no_ws) {
        auto ks = fc->ntl == 13 ? bnn_z_chain_kernel<13, 1> : bnn_z_chain_kernel<7, 1>;
        hipLaunchKernelGGL(ks, dim3(2), dim3(BNN_THREADS), lds_z, stream, a, fc->tab_z, fc->ws_z, zsy, zr);
      } else {      // the mean call's last layer over the idle waves, the variance-head call on its one column (ecb_z_chain<WS>)
        auto ks = fc->ntl == 13 ? bnn_z_chain_kernel<13, 1, false, 1, true> : bnn_z_chain_kernel<7, 1, false, 1, true>;
        const size_t lds_ws = (32 + 2 * 16 + 4 * 16 * 16 + 2 * 1024 + 64 + 4 * 1024 + 8) * sizeof(float);
        hipLaunchKernelGGL(ks, dim3(2), dim3(BNN_THREADS), std::max(lds_z, lds_ws), stream, a, fc->tab_z, fc->ws_z, zsy, zr);
      }
    } else
      hipLaunchKernelGGL(kc, dim3(1), dim3(BNN_THREADS), lds_z, stream, a, fc->tab_z, fc->ws_z, zsy, EcbZRows{});
  } else {
    static const bool one_launch = std::getenv("BGM_BNN_STEP_ONE_LAUNCH") != nullptr;
    a.wide = one_launch ? 0 : 1;      // the two calls' eps / dW over the chip (bnn_step_noise_kernel), the sign words in the step kernel
    a.ws = s->ws_dev + 3 * s->ws_stride;      // (its own slices: the theta step of the next minibatch may be running beside it)
    if (a.wide) {      // the two noise calls of a net on workgroups of their own, forward and backward as two launches
      hipLaunchKernelGGL(bnn_step_noise_kernel, dim3(BNN_NOISE_PARTS, 3, 2), dim3(BNN_THREADS), 0, stream, a, 6, 1);
      static const bool lds_ok_f = hipFuncSetAttribute(reinterpret_cast<const void *>(bnn_z_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)BNN_R32_LDS_BYTES) == hipSuccess;
      (void)lds_ok_f;
      hipLaunchKernelGGL(bnn_z_fwd_kernel, dim3(6), dim3(BNN_THREADS), BNN_R32_LDS_BYTES, stream, a);
      static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(bnn_z_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)BNN_R32_LDS_BYTES) == hipSuccess;
      (void)lds_ok;
      hipLaunchKernelGGL(bnn_z_bwd_kernel, dim3(6), dim3(BNN_THREADS), BNN_R32_LDS_BYTES, stream, a);
      hipLaunchKernelGGL(bnn_z_combine6_kernel, dim3((batch * s->q + 255) / 256), dim3(256), 0, stream, a.dz_part, a.loss_part, a.dz, out, batch * s->q,
                         a.data_z, a.idx, s->q, a.inv_B);
    } else {
      hipLaunchKernelGGL(bnn_z_grad_kernel, dim3(3), dim3(BNN_THREADS), 0, stream, a);
      hipLaunchKernelGGL(bnn_z_combine_kernel, dim3((batch * s->q + 255) / 256), dim3(256), 0, stream, a.dz_part, a.loss_part, a.dz, out, batch * s->q);
    }
  }
  BGM_HIP_CHECK(hipGetLastError());
  if (dz_out) return BGM_OK;    // gradient only (parity tests)
  s->t_z += 1;
  const float lr_t = adam_lr_t(lr_z, s->t_z);
  const int q = s->q;
  const long long n = (long long)n_rows * q;
  const int nb = batch * q;
  if (fused_rows) {
    // (the chain kernel has applied the step)
  } else if (lazy == 2) {
    hipLaunchKernelGGL(bnn_z_rows_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, data_z, zm, zv, a.dz, idx, batch, q, lr_t,
                       BNN_ADAM_B1, BNN_ADAM_B2, BNN_ADAM_EPS, 1, s->tlast_dev, (int)s->t_z, fc ? fc->sync_zr : FitSync{});
  } else if (lazy) {
    hipLaunchKernelGGL(bnn_z_rows_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, data_z, zm, zv, a.dz, idx, batch, q, lr_t,
                       BNN_ADAM_B1, BNN_ADAM_B2, BNN_ADAM_EPS, 1, nullptr, 0, fc ? fc->sync_zr : FitSync{});
  } else {
    hipLaunchKernelGGL(bnn_z_decay_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, zm, zv, n, BNN_ADAM_B1, BNN_ADAM_B2);
    hipLaunchKernelGGL(bnn_z_rows_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, data_z, zm, zv, a.dz, idx, batch, q, lr_t,
                       BNN_ADAM_B1, BNN_ADAM_B2, BNN_ADAM_EPS, 0);
    hipLaunchKernelGGL(bnn_z_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, data_z, zm, zv, n, lr_t, BNN_ADAM_EPS);
  }
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

static int bnn_z_sync_impl(bgm_handle *h, float *data_z, float *zm, float *zv, const int32_t *idx, int64_t n_rows, int32_t batch, float lr_z,
                           void *stream_, int t_to_ofs, bool mark);
extern "C" int bgm_bnn_z_sync(bgm_handle *h, float *data_z, float *zm, float *zv, const int32_t *idx, int64_t n_rows, int32_t batch,
                              float lr_z, void *stream_) {
  return bnn_z_sync_impl(h, data_z, zm, zv, idx, n_rows, batch, lr_z, stream_, 0, true);
}
// t_to_ofs: the rows are brought to step t_z + t_to_ofs (a minibatch that many latent steps ahead, bgm_bnn_fit_epoch);
// mark = false: the caller guarantees the latent step of these rows follows (bnn_z_rows_kernel stamps them)
static int bnn_z_sync_impl(bgm_handle *h, float *data_z, float *zm, float *zv, const int32_t *idx, int64_t n_rows, int32_t batch, float lr_z,
                           void *stream_, int t_to_ofs, bool mark) {
  int rc = bnn_need(h, "bgm_bnn_z_sync");
  if (rc) return rc;
  BnnState *s = bst(h);
  if (!data_z || !zm || !zv || n_rows < 1 || (idx && batch < 1)) { bgm_set_error("bgm_bnn_z_sync: bad argument"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (s->tlast_dev && s->tlast_rows != n_rows) {
    // another latent table (a second fit() on a panel of another size, another shard): legal once the previous table is flushed
    if (s->z_synced != -2) { bgm_set_error("bgm_bnn_z_sync: n_rows differs from the table the replay state belongs to, and that table has pending steps (flush it with idx = NULL first)"); return BGM_E_STATE; }
    BGM_HIP_CHECK(hipFree(s->tlast_dev));
    s->tlast_dev = nullptr; s->tlast_rows = 0;
  }
  if (!s->tlast_dev) {                               // every row is current at the step the mode is entered
    BGM_HIP_CHECK(hipMalloc(&s->tlast_dev, sizeof(int) * n_rows));
    s->tlast_rows = n_rows;
    hipLaunchKernelGGL(fit_fill_int_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream, s->tlast_dev, (long long)n_rows, (int)s->t_z);
  }
  const long long n_sel = idx ? batch : n_rows;
  const long long threads = n_sel * s->q * 16;
  hipLaunchKernelGGL(fit_adam_z_replay_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, data_z, zm, zv,
                     s->tlast_dev, s->q, idx, n_sel, (int)s->t_z + t_to_ofs, lr_z, BNN_ADAM_B1, BNN_ADAM_B2, BNN_ADAM_EPS);
  if (idx && !mark) {
    if (t_to_ofs == 0) s->z_synced = s->t_z + 1;
  } else if (!idx) {
    hipLaunchKernelGGL(fit_fill_int_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream, s->tlast_dev, (long long)n_rows, (int)s->t_z);
    s->z_synced = -2;
  } else {
    hipLaunchKernelGGL(fit_mark_rows_kernel, dim3((unsigned)((n_sel + 255) / 256)), dim3(256), 0, stream, s->tlast_dev, idx, n_sel, (int)s->t_z);
    s->z_synced = s->t_z + 1;
  }
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// A list of minibatches with the loop inside the library (single process).  replaces: the loop body causalbgm/base.py:490-505 with
// use_bnn for the minibatches perm[0 .. n_use) taken `batch` rows at a time: bgm_bnn_z_sync (lazy = 2), bgm_bnn_theta_step(apply = 1),
// bgm_bnn_z_step in that order, noise streams stream_id0 + 3 k (theta step) and + 3 k + 1 (latent step) for the k-th minibatch
// of at least two rows (a one-row tail is skipped, as the host loop does).  *n_done = minibatches run.
// On the row-tile chains with lazy = 1 / 2 the latent phase of minibatch k runs on a second stream beside the noise / forward /
// backward chains of minibatch k + 1 (both read the parameters of step k + 1; disjoint rows of the latent table); the gradient-tile
// kernel of minibatch k + 1, which applies Adam, waits for it.  Results are those of the sequential order.
// ---------------------------------------------------------------------------------------------------------------------------
// a device-side wait of the previous bgm_bnn_fit_epoch call gave up (fit_sync.h): that call's results are void
static int bnn_epoch_check(bgm_handle *h, hipStream_t stream) {
  if (!h->epoch_ctr) return BGM_OK;
  unsigned c[4];
  BGM_HIP_CHECK(hipMemcpyAsync(c, h->epoch_ctr, sizeof(c), hipMemcpyDeviceToHost, stream));
  BGM_HIP_CHECK(hipStreamSynchronize(stream));
  if (!c[2]) return BGM_OK;
  BGM_HIP_CHECK(hipMemset(h->epoch_ctr, 0, sizeof(c)));
  h->epoch_theta_done = h->epoch_z_done = 0;
  bgm_set_error("bgm_bnn_fit_epoch: a device-side ordering wait timed out in the previous call (its updates are unordered); "
                "BGM_FIT_NO_FLAGS=1 orders the two streams with HIP events instead");
  return BGM_E_HIP;
}

extern "C" int bgm_bnn_fit_epoch(bgm_handle *h, const float *x, const float *y, const float *v, float *data_z, float *zm, float *zv,
                                 const int32_t *perm, int64_t n_rows, int64_t n_use, int32_t batch, float lr_theta, float lr_z, int32_t lazy,
                                 uint64_t seed, uint32_t stream_id0, float *out_t, float *out_z, int32_t *n_done, void *stream_) {
  int rc = bnn_need(h, "bgm_bnn_fit_epoch");
  if (rc) return rc;
  BnnState *s = bst(h);
  if (!x || !y || !v || !data_z || !zm || !zv || !perm || n_use < 1 || batch < 2 || batch > s->cfg.max_batch) { bgm_set_error("bgm_bnn_fit_epoch: bad argument"); return BGM_E_INVALID; }
  hipStream_t sA = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BnnFitChain *fc = static_cast<BnnFitChain *>(s->chain);
  static const bool no_overlap = std::getenv("BGM_FIT_NO_OVERLAP") != nullptr;
  const bool chains = fc && (batch == 32 || (batch == 16 && !fc->pad));
  // (the general-width step kernels overlap as well -- latent step k beside the reading part of theta step k + 1, own workspace slices --
  // ordered by HIP events; the device-side counters belong to the row-tile chains)
  static const bool one_launch = std::getenv("BGM_BNN_STEP_ONE_LAUNCH") != nullptr;
  const bool overlap = !no_overlap && lazy != 0 && (chains || (!one_launch && s->kl_part_dev));
  if (overlap && !h->epoch_stream) {
    BGM_HIP_CHECK(hipStreamCreateWithFlags(&h->epoch_stream, hipStreamNonBlocking));
    for (int k = 0; k < 4; ++k) {
      BGM_HIP_CHECK(hipEventCreateWithFlags(&h->epoch_ev_t[k], hipEventDisableTiming));
      BGM_HIP_CHECK(hipEventCreateWithFlags(&h->epoch_ev_z[k], hipEventDisableTiming));
    }
  }
  hipStream_t sB = overlap ? h->epoch_stream : sA;
  long long k = 0;
  // Ordering of the two streams by counters in device memory the kernels wait on / advance themselves (fit_sync.h) instead of HIP
  // events (4-6 us of command-processor time per record / wait pair in the stream it sits in); the replay of a minibatch's pending
  // latent steps runs two minibatches ahead on the second stream (in front of the latent phase k - 2, whose completion the
  // gradient-tile kernel k - 1 waits for), and the latent step itself stamps the rows (no mark launch).  BGM_FIT_NO_FLAGS: events.
  static const bool no_flags = std::getenv("BGM_FIT_NO_FLAGS") != nullptr;
  bool flags = overlap && !no_flags && chains;
  if (flags) {
    if (!h->epoch_ctr) {
      BGM_HIP_CHECK(hipMalloc((void **)&h->epoch_ctr, sizeof(unsigned) * 8));
      BGM_HIP_CHECK(hipMemsetAsync(h->epoch_ctr, 0, sizeof(unsigned) * 8, sA));
      h->epoch_theta_done = h->epoch_z_done = 0;
    } else if ((rc = bnn_epoch_check(h, sA))) return rc;
    if (!h->epoch_flags_ok || h->epoch_probe_stream != (void *)sA) {      // once per handle and caller stream: do the two streams run side by side (fit_sync.h)?
      int ok = 0;
      BGM_HIP_CHECK(fit_sync_probe(sA, sB, h->epoch_ctr + 4, &ok));
      h->epoch_flags_ok = ok ? 1 : -1;
      h->epoch_probe_stream = (void *)sA;
    }
    if (h->epoch_flags_ok < 0) flags = false;      // (a profiler serialising kernels, one hardware queue): HIP events
  }
  int *err = flags ? (int *)(h->epoch_ctr + 2) : nullptr;
  static const bool no_ahead = std::getenv("BGM_BNN_NO_AHEAD") != nullptr;
  static const bool one_wg_env = std::getenv("BGM_FIT_ONE_WG") != nullptr, no_ws_env = std::getenv("BGM_FIT_NO_WORKERS") != nullptr;
  const bool ahead_ok = flags && !no_ahead && fc && batch == 32 && !fc->pad && fc->t0 == 1 && !one_wg_env && !no_ws_env && lazy != 0;
  if (fc) { fc->theta_prepared = fc->z_prepared = fc->riders_on = false; }
  const int q = s->q;
  const unsigned dw_blocks = fc ? (unsigned)((fc->n_tiles + ECH_WAVES - 1) / ECH_WAVES + 1) : 0, zr_blocks = (unsigned)((batch * q + 255) / 256);
  auto rows_of = [&](int64_t i) { return (int)std::max<int64_t>(0, std::min<int64_t>(batch, n_use - i)); };
  const long long tz0 = s->t_z;
  long long replayed_to = 0;                 // minibatches (ordinals) [k, replayed_to) have been replayed ahead but not stepped
  auto replay_ahead = [&](int64_t i, int ofs, hipStream_t st) -> int {      // minibatch at perm + i, `ofs` latent steps from now
    const int bj = rows_of(i);
    if (bj < 2) return BGM_OK;
    const int rc_ = bnn_z_sync_impl(h, data_z, zm, zv, perm + i, n_rows, bj, lr_z, st, ofs, false);
    if (!rc_) replayed_to = std::max<long long>(replayed_to, i / batch + 1);
    return rc_;
  };
  // an error exit: nothing stays in flight on the private stream, and the rows replayed ahead without their latent step count as
  // current to the step they were brought to (minibatch j -> tz0 + j), so that the caller's flush does not replay them twice
  auto fail = [&](int rc_) -> int {
    if (fc) fc->theta_prepared = fc->z_prepared = fc->riders_on = false;
    if (overlap) hipStreamSynchronize(sB);
    if (lazy == 2 && flags && s->tlast_dev)
      for (long long j = k; j < replayed_to; ++j) {
        const int bj = rows_of(j * (int64_t)batch);
        if (bj >= 2) hipLaunchKernelGGL(fit_mark_rows_kernel, dim3((unsigned)((bj + 255) / 256)), dim3(256), 0, sA, s->tlast_dev,
                                        perm + j * (int64_t)batch, (long long)bj, (int)(tz0 + j));
      }
    if (n_done) *n_done = (int32_t)k;
    return rc_;
  };
  for (int64_t i = 0; i < n_use; i += batch) {
    const int32_t *idx = perm + i;
    const int b = rows_of(i);
    if (b < 2) continue;                     // batch statistics need two rows
    const bool ov = flags && b == batch;     // (a short last minibatch runs on the phase machine: in stream order, on the caller's stream)
    const uint32_t s0 = stream_id0 + (uint32_t)(3 * k);
    if (lazy == 2 && !flags && (rc = bnn_z_sync_impl(h, data_z, zm, zv, idx, n_rows, b, lr_z, sA, 0, !overlap))) return fail(rc);
    if (lazy == 2 && flags && k == 0 && ((rc = replay_ahead(i, 0, sA)) || (rc = replay_ahead(i + batch, 1, sA)))) return fail(rc);
    if (overlap && k == 0) {                 // the second stream starts behind everything queued on the caller's so far
      BGM_HIP_CHECK(hipEventRecord(h->epoch_ev_t[1], sA));
      BGM_HIP_CHECK(hipStreamWaitEvent(sB, h->epoch_ev_t[1], 0));
    }
    if (flags && !ov && k > 0) {             // the short tail: everything of the earlier minibatches first
      BGM_HIP_CHECK(hipEventRecord(h->epoch_ev_z[0], sB));
      BGM_HIP_CHECK(hipStreamWaitEvent(sA, h->epoch_ev_z[0], 0));
    }
    BnnArgs a{};
    bnn_base_args(s, a, b, 0, seed, s0);
    a.data_z = data_z; a.idx = idx; a.x_ = x; a.y_ = y; a.v_ = v; a.apply = 1; a.out = out_t;
    s->t_theta += 1;
    a.adam = BnnAdam{adam_lr_t(lr_theta, s->t_theta), BNN_ADAM_B1, BNN_ADAM_B2, BNN_ADAM_EPS};
    // noise ahead (EcbRider): this minibatch's theta launch prepares the noise of its own latent phase and of the next theta phase
    const bool ride = ov && ahead_ok;
    if (ride) {
      const bool next_full = i + batch < n_use && rows_of(i + batch) == batch;
      fc->ws = fc->ws_ring[k & 1]; fc->ws_z = fc->ws_z_ring[k & 1];
      float *ws_next = next_full ? fc->ws_ring[(k + 1) & 1] : nullptr;
      unsigned *kl_cnt = reinterpret_cast<unsigned *>(fc->ws + fc->kl_cnt_off);
      fc->rider = EcbRider{ws_next, stream_id0 + (uint32_t)(3 * (k + 1)), fc->tab_z, fc->ws_z, s0 + 1, kl_cnt};
      fc->ahead = EcbAhead{ws_next, fc->tab_z, fc->ws_z, {0, 0, 0, 0}, {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, next_full ? out_t : nullptr, out_z, kl_cnt};
      for (int kk = 0; kk < 4; ++kk) { fc->ahead.dw_t[kk] = fc->dw_t[kk]; fc->ahead.dw_z[kk][0] = fc->dw_z[kk][0]; fc->ahead.dw_z[kk][1] = fc->dw_z[kk][1]; }
      fc->riders_on = true;
    }
    bnn_theta_launch(s, a, b, 1, sA);
    if (fc) fc->theta_prepared = false;
    if (ov) fc->sync_dw = FitSync{k > 0 ? h->epoch_ctr + 1 : nullptr, h->epoch_z_done, h->epoch_ctr, err};
    else if (overlap && !flags && k > 0) BGM_HIP_CHECK(hipStreamWaitEvent(sA, h->epoch_ev_z[(k - 1) & 1], 0));
    bnn_theta_launch(s, a, b, 2, sA);
    BGM_HIP_CHECK(hipGetLastError());
    s->packed_valid = false; s->bnf_valid = false;
    if (ride) {
      fc->riders_on = false;
      fc->theta_prepared = fc->ahead.ws_next != nullptr;      // (the next minibatch's theta launch)
      fc->z_prepared = true;
    }
    if (ov) {
      fc->sync_dw = FitSync{};
      h->epoch_theta_done += dw_blocks;
      fc->sync_zn = FitSync{h->epoch_ctr, h->epoch_theta_done, nullptr, err};
      fc->sync_zr = FitSync{nullptr, 0, h->epoch_ctr + 1, err};
      if (lazy == 2 && (rc = replay_ahead(i + 2 * (int64_t)batch, 2, sB))) return fail(rc);      // (in front of this latent phase: its count covers it)
      s->z_synced = s->t_z + 1;
    } else if (overlap && !flags) {
      BGM_HIP_CHECK(hipEventRecord(h->epoch_ev_t[k & 1], sA));
      BGM_HIP_CHECK(hipStreamWaitEvent(sB, h->epoch_ev_t[k & 1], 0));
    }
    hipStream_t sz = (flags && !ov) ? sA : sB;
    if (flags && !ov) s->z_synced = s->t_z + 1;
    rc = bgm_bnn_z_step(h, x, y, v, data_z, zm, zv, idx, n_rows, b, 0, lr_z, lazy, seed, s0 + 1, out_z, nullptr, sz);
    if (fc) fc->z_prepared = false;
    if (ov) { fc->sync_zn = FitSync{}; fc->sync_zr = FitSync{}; if (!rc) h->epoch_z_done += ride ? 2u : zr_blocks; }      // (ride: counted by the chain kernel's two workgroups)
    if (rc) return fail(rc);
    if (overlap && !flags) BGM_HIP_CHECK(hipEventRecord(h->epoch_ev_z[k & 1], sB));
    ++k;
  }
  if (overlap && k > 0) {
    BGM_HIP_CHECK(hipEventRecord(h->epoch_ev_z[0], sB));
    BGM_HIP_CHECK(hipStreamWaitEvent(sA, h->epoch_ev_z[0], 0));
  }
  if (n_done) *n_done = (int32_t)k;
  // a device-side wait that gave up voids THIS call: reported now, before the caller evaluates or checkpoints the state
  return flags ? bnn_epoch_check(h, sA) : BGM_OK;
}

// This rank's share of a data-parallel epoch (models/causalbgm_bnn.py under torch.distributed): per minibatch the steps of the host loop
// -- replay of the rows' pending latent steps, theta gradient (apply = 0, scaled 1 / (b * world)), the sum of the session's fused
// g|h|f gradient over the ranks (ncclAllReduce on `stream`, in place in the session's buffer: no exchange copies), the Adam step, the
// latent step -- issued from C++ on one stream.  replaces: the loop body causalbgm/base.py:490-505 with use_bnn, sharded by rows.
extern "C" int bgm_bnn_fit_epoch_dp(bgm_handle *h, const float *x, const float *y, const float *v, float *data_z, float *zm, float *zv,
                                    const int32_t *perm, int64_t n_rows, int64_t n_use, int32_t batch, float lr_theta, float lr_z,
                                    int32_t lazy, uint64_t seed, uint32_t stream_id0, float *out_t, float *out_z, int32_t *n_done,
                                    void *comm, void *stream_) {
  int rc = bnn_need(h, "bgm_bnn_fit_epoch_dp");
  if (rc) return rc;
  BnnState *s = bst(h);
  if (!x || !y || !v || !data_z || !zm || !zv || !perm || n_use < 1 || batch < 2 || batch > s->cfg.max_batch) { bgm_set_error("bgm_bnn_fit_epoch_dp: bad argument"); return BGM_E_INVALID; }
  if (!comm) { bgm_set_error("bgm_bnn_fit_epoch_dp: NULL communicator (bgm_comm_create)"); return BGM_E_INVALID; }
  int world = 0, rank = 0;
  if ((rc = bgm_comm_world(comm, &world, &rank))) return rc;
  hipStream_t st = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (BnnFitChain *fc = static_cast<BnnFitChain *>(s->chain)) fc->theta_prepared = fc->z_prepared = fc->riders_on = false;
  long long k = 0;
  for (int64_t i = 0; i < n_use; i += batch) {
    const int32_t *idx = perm + i;
    const int b = (int)std::min<int64_t>(batch, n_use - i);
    if (b < 2) continue;                     // batch statistics need two rows (the same decision on every rank)
    const int bg = world > 1 ? b * world : 0;
    const uint32_t s0 = stream_id0 + (uint32_t)(3 * k);
    if (lazy == 2 && (rc = bgm_bnn_z_sync(h, data_z, zm, zv, idx, n_rows, b, lr_z, st))) break;
    if ((rc = bgm_bnn_theta_step(h, data_z, idx, x, y, v, b, bg, lr_theta, seed, s0, 0, out_t, st))) break;
    if ((rc = bgm_comm_enqueue_all_reduce(comm, s->grad_dev, s->n_params, st))) break;
    if ((rc = bgm_bnn_theta_apply(h, lr_theta, st))) break;
    if ((rc = bgm_bnn_z_step(h, x, y, v, data_z, zm, zv, idx, n_rows, b, bg, lr_z, lazy, seed, s0 + 1, out_z, nullptr, st))) break;
    ++k;
  }
  if (n_done) *n_done = (int32_t)k;
  return rc;
}

extern "C" int bgm_bnn_end(bgm_handle *h, void *stream_) {
  if (!h) return BGM_E_INVALID;
  if (!h->bnn_state) return BGM_OK;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  const int rc_epoch = bnn_epoch_check(h, (hipStream_t)stream_);
  bgm_bnn_free_state(h);
  return rc_epoch;
}
