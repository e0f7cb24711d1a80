// bnn_egm_api.hip -- C ABI of the EGM warm start with Bayesian networks (a sub-session of bgm_bnn_begin).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bgm_host.h"
#include "bnn_egm_kernels.h"
#include "bnn_state.h"

static BnnState *bst(bgm_handle *h) { return static_cast<BnnState *>(h->bnn_state); }

struct BnnEgmState {
  bgm_egm_config cfg{};
  BnnEgmArgs base{};
  size_t n_dz = 0, ws_floats = 0;
  int lds_bytes = 0;
  long long t_g = 0, t_d = 0;
  float *dev = nullptr;      // m | v (EGM Adam slots of the Bayesian nets) | theta_d | m_d | v_d | grad_d | ws
  // generator step as row-tile chains (egm_chain_bnn.h)
  int chain_gen_lds = 0, chain_ntl = 0, n_tiles = 0;
  int chain_t0 = 1;            // latent input tiles of the chain kernels (q <= 16 t0; two tiles: B = 32, 13-tile kernels)
  bool chain_pad = false;
  EcbCall disc_call{};         // noise of the discriminator step's encoder call (workspace offsets)
  EcbTab *tab_dev = nullptr;
  int *tiles_dev = nullptr;
  float *thetaT_dev = nullptr;
};

void bgm_bnn_egm_free(void *p) {
  BnnEgmState *e = static_cast<BnnEgmState *>(p);
  if (!e) return;
  if (e->dev) hipFree(e->dev);
  if (e->tab_dev) hipFree(e->tab_dev);
  if (e->tiles_dev) hipFree(e->tiles_dev);
  if (e->thetaT_dev) hipFree(e->thetaT_dev);
  delete e;
}

extern "C" int bgm_bnn_egm_begin(bgm_handle *h, const bgm_egm_config *cfg, const float *theta_dz_host, int64_t count, void *stream_) {
  (void)stream_;
  if (!h || !h->bnn_state) { bgm_set_error("bgm_bnn_egm_begin: no session (bgm_bnn_begin)"); return BGM_E_STATE; }
  if (!cfg || !theta_dz_host) { bgm_set_error("bgm_bnn_egm_begin: NULL argument"); return BGM_E_INVALID; }
  BnnState *s = bst(h);
  if (cfg->batch_size < 2 || cfg->batch_size > s->cfg.max_batch) { bgm_set_error("bgm_bnn_egm_begin: batch_size outside [2, max_batch]"); return BGM_E_INVALID; }
  if (cfg->n_hidden_dz < 1 || cfg->n_hidden_dz + 1 > EGM_MAX_LAYERS) { bgm_set_error("bgm_bnn_egm_begin: bad dz_units"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  bgm_bnn_egm_free(s->egm); s->egm = nullptr;
  BnnEgmState *e = new BnnEgmState();
  s->egm = e;
  e->cfg = *cfg;
  BnnEgmArgs &a = e->base;
  for (int k = 0; k < 4; ++k) a.net[k] = s->net[k];
  EgmDisc &d = a.dz;
  const int L = cfg->n_hidden_dz;
  d.n_hidden = L;
  d.dims[0] = s->q;
  for (int l = 0; l < L; ++l) d.dims[l + 1] = cfg->dz_units[l];
  d.dims[L + 1] = 1;
  int o = 0;
  for (int l = 0; l <= L; ++l) { d.w[l] = o; o += d.dims[l] * d.dims[l + 1]; }
  for (int l = 0; l <= L; ++l) { d.b[l] = o; o += d.dims[l + 1]; }
  for (int l = 0; l < L; ++l) { d.gamma[l] = o; o += d.dims[l + 1]; }
  for (int l = 0; l < L; ++l) { d.beta[l] = o; o += d.dims[l + 1]; }
  d.n_params = o;
  egm_finish_disc(d);
  d.fixed_norm = h->disc_norm;
  e->n_dz = (size_t)o;
  if ((int64_t)o != count) {
    bgm_bnn_egm_free(e); s->egm = nullptr;
    bgm_set_error("bgm_bnn_egm_begin: expected " + std::to_string(o) + " discriminator parameters, got " + std::to_string(count));
    return BGM_E_INVALID;
  }
  const int B = cfg->batch_size;
  int wmax = s->wmax, dmax = 0, dsum = 0, dall = 0;
  for (int l = 0; l <= L; ++l) { dmax = std::max(dmax, d.dims[l]); dall += d.dims[l]; wmax = std::max(wmax, d.dims[l]); }
  for (int l = 1; l <= L; ++l) dsum += d.dims[l];
  a.dmax = dmax; a.wmax = wmax;
  const size_t cache = ((size_t)(2 * B + 1) * dsum + B + 3) / 4 * 4;
  const size_t gp = ((size_t)B * dall + (size_t)(5 * B + 1) * dsum + 3) / 4 * 4 + 3 * (size_t)B * dmax;
  const size_t arena = cache + std::max(cache + 2 * (size_t)B * dmax, gp);
  a.disc_lds = (64 + arena) * sizeof(float) <= 160 * 1024 ? 1 : 0;
  e->lds_bytes = (int)((64 + (a.disc_lds ? arena : 0)) * sizeof(float));
  a.n_gen = s->n_params; a.B = B; a.q = s->q; a.p = s->p;
  a.z0 = s->cfg.z_dims[0]; a.z1 = s->cfg.z_dims[1]; a.z2 = s->cfg.z_dims[2];
  a.binary = s->cfg.binary_treatment; a.use_z_rec = cfg->use_z_rec;
  // workspace: nine Flipout call caches (g x 3, e x 2, f x 2, h x 2) + discriminator caches + scratch rows
  auto cache_floats = [&](const BnnNet &n) {
    return (size_t)B * n.dims[0] + n.dims[0] + 2 * (size_t)B * n.hoff[n.n_layers + 1] + 2 * (size_t)n.eoff[n.n_layers] + (size_t)B * n.swords + 64;
  };
  size_t gen_ws = 0;
  EcbTab tab{};
  std::vector<int> tiles;
  {
    const int ntl_need = (s->p + 1 + 15) / 16;
    const int t0 = s->q <= 16 ? 1 : 2;
    e->chain_t0 = t0;
    const int ntl = ((ntl_need == 13 || ntl_need == 7) && t0 == 1) ? ntl_need : 13;       // narrower outputs: the 13-tile kernels with masked columns (B = 32)
    bool chain = d.fixed_norm && L == 3 && d.dims[0] <= 32 && (d.dims[1] + 15) / 16 == 4 && (d.dims[2] + 15) / 16 == 2 && d.dims[3] >= 1 &&
                 d.dims[3] <= 16 && (B == 16 || B == 32) && ntl_need <= 13 && (ntl == ntl_need || B == 32) && s->q <= 32 && (t0 == 1 || B == 32) &&
                 !std::getenv("BGM_EGM_NO_CHAIN_GEN");
    for (int k = 0; k < 4; ++k) chain = chain && s->net[k].bn_fixed == 1 && !s->net[k].heads && !s->net[k].mv;
    const BnnNet &G = s->net[BNN_G], &E = s->net[BNN_E];
    chain = chain && G.n_layers >= 3 && E.n_layers >= 3 && G.dims[0] == s->q && E.dims[E.n_layers] == s->q && G.dims[G.n_layers] == s->p + 1 &&
            E.dims[0] == s->p;
    for (int l = 1; l < G.n_layers; ++l) chain = chain && G.dims[l] == 64;
    for (int l = 1; l < E.n_layers; ++l) chain = chain && E.dims[l] == 64;
    for (int k : {BNN_F, BNN_H}) {
      const BnnNet &m = s->net[k];
      chain = chain && m.n_layers == 4 && m.dims[0] <= 16 && m.dims[1] == 64 && m.dims[2] == 32 && m.dims[3] >= 1 && m.dims[3] <= 16 &&
              m.dims[4] >= 1 && m.dims[4] <= 16;
    }
    if (chain) {
      const int call_net[ECB_CALLS] = {BNN_G, BNN_G, BNN_E, BNN_E, BNN_G, BNN_F, BNN_F, BNN_H, BNN_H};
      const int call_soff[ECB_CALLS] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
      const int all_nets[4] = {BNN_G, BNN_E, BNN_F, BNN_H};
      gen_ws = ecb_build_tab(s->net, call_net, call_soff, ECB_CALLS, B, ntl, tab, tiles, all_nets, 4);
      tab.n_warm = s->n_params;
      e->chain_ntl = ntl;
      e->chain_pad = ntl != ntl_need || t0 == 2;
      e->n_tiles = tab.n_tiles;
      e->chain_gen_lds = (int)(sizeof(float) * (size_t)(t0 == 1 ? ecb_lds_floats<4, 2, 1>(d, B) : ecb_lds_floats<4, 2, 1, 2>(d, B)));
    }
  }
  {   // noise block of the discriminator step's encoder call, behind the generator chain's region
    const BnnNet &E = s->net[BNN_E];
    size_t off = (gen_ws + 3) / 4 * 4;
    e->disc_call.net = BNN_E;
    e->disc_call.dW = (int)off; off += ((size_t)E.eoff[E.n_layers] + 16 + 3) / 4 * 4;
    e->disc_call.sg = (int)off; off += ((size_t)B * E.swords + 3) / 4 * 4;
    e->disc_call.xh = (int)off; off += (size_t)B * EchDims<4, 2, 1, 2>::SW;      // two latent tiles: the tail's fourth stash (EchDiscIo::gstash)
    gen_ws = off;
  }
  size_t keep = 0;      // the kept upstream gradients of the nine calls (wide generator step: G / GS of bnn_bwd)
  for (int k : {BNN_G, BNN_G, BNN_G, BNN_E, BNN_E, BNN_F, BNN_F, BNN_H, BNN_H}) keep += 2 * ((size_t)B * s->net[k].hoff[s->net[k].n_layers + 1] + 4);
  e->ws_floats = keep + gen_ws + 3 * cache_floats(s->net[BNN_G]) + 2 * cache_floats(s->net[BNN_E]) + 2 * cache_floats(s->net[BNN_F]) +
                 2 * cache_floats(s->net[BNN_H]) + (size_t)B * (4 * (size_t)s->p + 32 * (size_t)wmax + 256) + 4 * cache + e->n_dz + arena + 8192;
  const size_t np = ((size_t)s->n_params + 63) & ~(size_t)63, nd = (e->n_dz + 63) & ~(size_t)63;
  const size_t total = 2 * np + 4 * nd + e->ws_floats + 64;
  BGM_HIP_CHECK(hipMalloc((void **)&e->dev, total * sizeof(float)));
  BGM_HIP_CHECK(hipMemset(e->dev, 0, total * sizeof(float)));
  a.theta = s->theta_dev; a.grad = s->grad_dev;
  BGM_HIP_CHECK(hipMemset(s->grad_dev, 0, sizeof(float) * (size_t)s->n_params));   // entries no step writes stay zero (bgm_bnn_egm_apply runs Adam over all of them)
  a.m = e->dev; a.v = e->dev + np;
  a.theta_d = e->dev + 2 * np; a.m_d = a.theta_d + nd; a.v_d = a.m_d + nd; a.grad_d = a.v_d + nd;
  a.ws = a.grad_d + nd;
  BGM_HIP_CHECK(hipMemcpy(a.theta_d, theta_dz_host, e->n_dz * sizeof(float), hipMemcpyHostToDevice));
  if (e->chain_gen_lds > 0) {
    // (no transposed mirror: the backward products read the canonical arrays K-contiguously, egm_chain_gen.h)
    BGM_HIP_CHECK(hipMalloc((void **)&e->tab_dev, sizeof(EcbTab)));
    BGM_HIP_CHECK(hipMemcpy(e->tab_dev, &tab, sizeof(EcbTab), hipMemcpyHostToDevice));
    BGM_HIP_CHECK(hipMalloc((void **)&e->tiles_dev, sizeof(int) * tiles.size()));
    BGM_HIP_CHECK(hipMemcpy(e->tiles_dev, tiles.data(), sizeof(int) * tiles.size(), hipMemcpyHostToDevice));
  }
  return BGM_OK;
}

static EgmAdam bnn_egm_adam(float lr, long long t) {
  EgmAdam ad;
  ad.b1 = BNN_ADAM_B1; ad.b2 = BNN_ADAM_B2; ad.eps = BNN_ADAM_EPS;
  ad.lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow((double)BNN_ADAM_B2, (double)t)) / (1.0 - std::pow((double)BNN_ADAM_B1, (double)t)));
  return ad;
}

static int bnn_egm_need(bgm_handle *h, const char *who, BnnState *&s, BnnEgmState *&e) {
  if (!h || !h->bnn_state || !bst(h)->egm) { bgm_set_error(std::string(who) + ": call bgm_bnn_egm_begin first"); return BGM_E_STATE; }
  s = bst(h);
  e = static_cast<BnnEgmState *>(s->egm);
  return BGM_OK;
}

extern "C" int bgm_bnn_egm_disc_step(bgm_handle *h, const float *z_dev, const int32_t *idx_dev, const float *v_dev, float eps,
                                     uint64_t seed, uint32_t stream_id, int32_t apply, float *out_dev, void *stream_) {
  BnnState *s; BnnEgmState *e;
  int rc = bnn_egm_need(h, "bgm_bnn_egm_disc_step", s, e);
  if (rc) return rc;
  if (!z_dev || !idx_dev || !v_dev) { bgm_set_error("bgm_bnn_egm_disc_step: NULL pointer"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BnnEgmArgs a = e->base;
  a.z = z_dev; a.idx = idx_dev; a.v_ = v_dev; a.eps = eps; a.out = out_dev; a.apply = apply ? 1 : 0;
  a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.stream = stream_id;
  if (apply) e->t_d += 1;
  a.adam = bnn_egm_adam(e->cfg.lr, std::max<long long>(1, e->t_d));
  {   // discriminator passes as register-chained row tiles when the shapes are the compiled ones (egm_chain.h)
    const EgmDisc &d = a.dz;
    const int t0 = a.q <= 16 ? 1 : 2;
    const bool chain = d.fixed_norm && d.n_hidden == 3 && d.dims[0] <= 32 && (t0 == 1 || a.B == 32) && (d.dims[1] + 15) / 16 == 4 && (d.dims[2] + 15) / 16 == 2 &&
                       d.dims[3] >= 1 && d.dims[3] <= 16 && (a.B == 16 || a.B == 32) && !std::getenv("BGM_EGM_NO_CHAIN");
    const size_t bytes = !chain ? 0 : sizeof(float) * (size_t)(t0 == 1 ? ech_disc_lds_floats<4, 2, 1>(d, a.B) : ech_disc_lds_floats<4, 2, 1, 2>(d, a.B));
    if (chain && bytes <= 160 * 1024) {
      // the Flipout encoder as a row-tile chain when its shape is a compiled one
      const BnnNet &E = a.net[BNN_E];
      bool ech = E.bn_fixed == 1 && !E.heads && E.n_layers >= 2 && E.dims[E.n_layers] == a.q && a.q <= 32 && !std::getenv("BGM_EGM_NO_CHAIN_BNN");
      for (int l = 1; l < E.n_layers; ++l) ech = ech && E.dims[l] == 64;
      int ntl = ech ? (a.p + 15) / 16 : 0;
      if (ntl != 13 && ntl != 7 && ntl > 0 && ntl < 13) ntl = 13;      // the 13-tile encoder chain masks the inputs beyond p
      if (t0 == 2 && ntl == 7) ntl = 13;
      if (ntl > 13) ntl = 0;
      auto kc = t0 == 2 ? (ntl == 13 ? bnn_egm_disc_chain_kernel<13, 4, 2, 1, 2, 2> : bnn_egm_disc_chain_kernel<0, 4, 2, 1, 2, 2>) :
                a.B == 32 ? (ntl == 13 ? bnn_egm_disc_chain_kernel<13, 4, 2, 1, 2> : ntl == 7 ? bnn_egm_disc_chain_kernel<7, 4, 2, 1, 2> : bnn_egm_disc_chain_kernel<0, 4, 2, 1, 2>)
                          : (ntl == 13 ? bnn_egm_disc_chain_kernel<13, 4, 2, 1, 1> : ntl == 7 ? bnn_egm_disc_chain_kernel<7, 4, 2, 1, 1> : bnn_egm_disc_chain_kernel<0, 4, 2, 1, 1>);
      BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      if (ntl == 13 || ntl == 7) {
        hipLaunchKernelGGL(bnn_egm_disc_noise_kernel, dim3(ECB_NOISE_PARTS), dim3(EGM_THREADS), 0, (hipStream_t)stream_, a, e->disc_call);
        BGM_HIP_CHECK(hipGetLastError());
      } else if (!std::getenv("BGM_BNN_STEP_ONE_LAUNCH")) {      // general encoder: its call's eps / dW over the chip
        a.wide = 1;
        hipLaunchKernelGGL(bnn_egm_disc_noise_wide_kernel, dim3(16), dim3(EGM_THREADS), 0, (hipStream_t)stream_, a, 0);
      }
      hipLaunchKernelGGL(kc, dim3(1), dim3(EGM_THREADS), bytes, (hipStream_t)stream_, a, e->disc_call);
      BGM_HIP_CHECK(hipGetLastError());
      return BGM_OK;
    }
  }
  auto k = bnn_egm_disc_step_kernel;
  if (!std::getenv("BGM_BNN_STEP_ONE_LAUNCH")) {
    a.wide = 1;
    hipLaunchKernelGGL(bnn_egm_disc_noise_wide_kernel, dim3(16), dim3(EGM_THREADS), 0, (hipStream_t)stream_, a, 1);
  }
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes));
  hipLaunchKernelGGL(k, dim3(1), dim3(EGM_THREADS), e->lds_bytes, (hipStream_t)stream_, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

// the generator chain's instantiations live in bnn_egm_gen_chain_{a,b,c}.hip (compile time)
int bnn_egm_gen_chain_launch_a(const BnnEgmArgs &a, int nb, int lds, const EcbTab *tab, float *thetaT, hipStream_t stream);
int bnn_egm_gen_chain_launch_b(const BnnEgmArgs &a, int nb, int lds, const EcbTab *tab, float *thetaT, hipStream_t stream);
int bnn_egm_gen_chain_launch_c(const BnnEgmArgs &a, int nb, int lds, const EcbTab *tab, float *thetaT, hipStream_t stream);
int bnn_egm_gen_chain_launch_d(const BnnEgmArgs &a, int nb, int lds, const EcbTab *tab, float *thetaT, hipStream_t stream);

extern "C" int bgm_bnn_egm_gen_step(bgm_handle *h, const float *z_dev, const int32_t *idx_dev, const float *v_dev, const float *x_dev,
                                    const float *y_dev, uint64_t seed, uint32_t stream_id, int32_t apply, float *out_dev, void *stream_) {
  BnnState *s; BnnEgmState *e;
  int rc = bnn_egm_need(h, "bgm_bnn_egm_gen_step", s, e);
  if (rc) return rc;
  if (!z_dev || !idx_dev || !v_dev || !x_dev || !y_dev) { bgm_set_error("bgm_bnn_egm_gen_step: NULL pointer"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BnnEgmArgs a = e->base;
  a.z = z_dev; a.idx = idx_dev; a.v_ = v_dev; a.x_ = x_dev; a.y_ = y_dev; a.out = out_dev; a.apply = apply ? 1 : 0;
  a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.stream = stream_id;
  if (apply) { e->t_g += 1; s->packed_valid = false; s->bnf_valid = false; }
  a.adam = bnn_egm_adam(e->cfg.lr, std::max<long long>(1, e->t_g));
  if (e->chain_gen_lds > 0) {
    hipLaunchKernelGGL(bnn_egm_gen_noise_kernel, dim3(ECB_CALLS * ECB_NOISE_PARTS), dim3(EGM_THREADS), 0, (hipStream_t)stream_, a, e->tab_dev);
    BGM_HIP_CHECK(hipGetLastError());
    const int nb = a.B / 16;
    rc = e->chain_t0 == 2 ? bnn_egm_gen_chain_launch_d(a, nb, e->chain_gen_lds, e->tab_dev, e->thetaT_dev, (hipStream_t)stream_)
         : e->chain_pad ? bnn_egm_gen_chain_launch_c(a, nb, e->chain_gen_lds, e->tab_dev, e->thetaT_dev, (hipStream_t)stream_)
         : e->chain_ntl == 13 ? bnn_egm_gen_chain_launch_a(a, nb, e->chain_gen_lds, e->tab_dev, e->thetaT_dev, (hipStream_t)stream_)
                              : bnn_egm_gen_chain_launch_b(a, nb, e->chain_gen_lds, e->tab_dev, e->thetaT_dev, (hipStream_t)stream_);
    if (rc) return rc;
    auto kd = a.B == 32 ? bnn_egm_gen_dw_kernel<2> : bnn_egm_gen_dw_kernel<1>;
    hipLaunchKernelGGL(kd, dim3((e->n_tiles + ECH_WAVES - 1) / ECH_WAVES + 1), dim3(EGM_THREADS), 0, (hipStream_t)stream_, a, e->tab_dev, e->tiles_dev,
                       e->thetaT_dev);
    BGM_HIP_CHECK(hipGetLastError());
    return BGM_OK;
  }
  // general widths: one workgroup walks the nine calls and their backward passes; the calls' eps / dW and the Adam step (elementwise over
  // all parameters of g, e, f, h) run as launches over the chip around it.  BGM_BNN_STEP_ONE_LAUNCH: everything inside the one workgroup.
  static const bool one_launch = std::getenv("BGM_BNN_STEP_ONE_LAUNCH") != nullptr;
  a.wide = one_launch ? 0 : 1;
  auto k = bnn_egm_gen_step_kernel;
  const int lds = 64 * (int)sizeof(float);
  if (a.wide) hipLaunchKernelGGL(bnn_egm_gen_noise_wide_kernel, dim3(16, 9), dim3(EGM_THREADS), 0, (hipStream_t)stream_, a);
  hipLaunchKernelGGL(k, dim3(1), dim3(EGM_THREADS), lds, (hipStream_t)stream_, a);
  if (a.wide) hipLaunchKernelGGL(bnn_egm_gen_dw_wide_kernel, dim3(16, 4), dim3(EGM_THREADS), 0, (hipStream_t)stream_, a);      // the gradient tiles of all layers
  if (a.wide && a.apply)
    hipLaunchKernelGGL(egm_dp_adam_kernel, dim3((unsigned)((a.n_gen + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, a.theta, a.m, a.v, a.grad, a.n_gen,
                       a.adam, (float *)nullptr, (const int *)nullptr);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bnn_egm_set_share(bgm_handle *h, int32_t row0) {
  BnnState *s; BnnEgmState *e;
  int rc = bnn_egm_need(h, "bgm_bnn_egm_set_share", s, e);
  if (rc) return rc;
  if (row0 < 0) { bgm_set_error("bgm_bnn_egm_set_share: row0 < 0"); return BGM_E_INVALID; }
  e->base.row0 = (uint32_t)row0;
  return BGM_OK;
}

extern "C" int bgm_bnn_egm_grad(bgm_handle *h, int32_t which, float scale, float *grad_dev, int64_t count, void *stream_) {
  BnnState *s; BnnEgmState *e;
  int rc = bnn_egm_need(h, "bgm_bnn_egm_grad", s, e);
  if (rc) return rc;
  const size_t n = which == 0 ? (size_t)s->n_params : e->n_dz;
  if ((which != 0 && which != 1) || !grad_dev || (size_t)count != n) {
    bgm_set_error("bgm_bnn_egm_grad: which must be 0 (g|e|f|h: " + std::to_string(s->n_params) + " floats) or 1 (discriminator: " + std::to_string(e->n_dz) + ")");
    return BGM_E_INVALID;
  }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  hipLaunchKernelGGL(egm_dp_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                     which == 0 ? e->base.grad : e->base.grad_d, grad_dev, (int)n, scale);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bnn_egm_apply(bgm_handle *h, int32_t which, const float *grad_dev, int64_t count, void *stream_) {
  BnnState *s; BnnEgmState *e;
  int rc = bnn_egm_need(h, "bgm_bnn_egm_apply", s, e);
  if (rc) return rc;
  const size_t n = which == 0 ? (size_t)s->n_params : e->n_dz;
  if ((which != 0 && which != 1) || !grad_dev || (size_t)count != n) { bgm_set_error("bgm_bnn_egm_apply: bad which / count"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  const BnnEgmArgs &a = e->base;
  const EgmAdam ad = bnn_egm_adam(e->cfg.lr, which == 0 ? ++e->t_g : ++e->t_d);
  if (which == 0) { s->packed_valid = false; s->bnf_valid = false; }
  hipLaunchKernelGGL(egm_dp_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, which == 0 ? a.theta : a.theta_d,
                     which == 0 ? a.m : a.m_d, which == 0 ? a.v : a.v_d, grad_dev, (int)n, ad, (float *)nullptr, (const int *)nullptr);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bnn_egm_read(bgm_handle *h, int32_t what, float *host, int64_t count, void *stream_) {
  BnnState *s; BnnEgmState *e;
  int rc = bnn_egm_need(h, "bgm_bnn_egm_read", s, e);
  if (rc) return rc;
  const float *src = what == 1 ? e->base.theta_d : what == 3 ? e->base.grad_d : nullptr;
  if (!src || !host || (size_t)count != e->n_dz) { bgm_set_error("bgm_bnn_egm_read: what must be 1 (parameters) or 3 (gradient) of the discriminator"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  BGM_HIP_CHECK(hipMemcpy(host, src, e->n_dz * sizeof(float), hipMemcpyDeviceToHost));
  return BGM_OK;
}

extern "C" int bgm_bnn_egm_end(bgm_handle *h, void *stream_) {
  if (!h || !h->bnn_state) return BGM_OK;
  BnnState *s = bst(h);
  if (!s->egm) return BGM_OK;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  bgm_bnn_egm_free(s->egm);
  s->egm = nullptr;
  return BGM_OK;
}
