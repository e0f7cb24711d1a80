// egm_api.hip -- C-ABI entry points of the CausalBGM EGM warm start (include/bgm_hip.h, EGM section).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bgm_host.h"
#include "egm_kernels.h"
#include "egm_chain.h"
#include "egm_chain_gen.h"

static constexpr float EGM_B1 = 0.9f, EGM_B2 = 0.99f, EGM_ADAM_EPS = 1e-7f;   // causalbgm/base.py:86-87, Keras epsilon

struct EgmState {
  bgm_egm_config cfg{};
  EgmArgs base{};              // network descriptions + device pointers shared by both kernels
  size_t n_gen = 0, n_dz = 0, ws_floats = 0;
  int lds_bytes = 0;
  int chain_t0 = 1;            // latent input tiles of the chain kernels (q <= 16 t0)
  int chain_disc_lds = 0;      // > 0: the discriminator step runs as register-chained row tiles (egm_chain.h) with this much LDS
  int chain_gen_lds = 0;       // > 0: likewise the generator step (egm_chain_gen.h)
  int chain_ntl = 0;           // its compiled output-tile count of g
  bool chain_pad = false;      // ... used for a narrower p + 1 (masked columns)
  EcgTab gen_tab{};
  std::vector<int> gen_tiles;
  float *thetaT_dev = nullptr;
  int *tiles_dev = nullptr;
  int *mirror_dev = nullptr;   // [n_gen] index of a parameter's copy in thetaT_dev, -1 for biases (data-parallel Adam, bgm_causal_egm_apply)
  long long t_g = 0, t_d = 0;  // Adam iteration counters of g_pre_optimizer / d_pre_optimizer
  float *dev = nullptr;        // one allocation: theta_g|m_g|v_g|grad_g|theta_d|m_d|v_d|grad_d|ws
};

static EgmState *est(bgm_handle *h) { return static_cast<EgmState *>(h->egm_state); }

void bgm_egm_free_state(bgm_handle *h) {
  if (!h->egm_state) return;
  EgmState *s = est(h);
  if (s->dev) hipFree(s->dev);
  if (s->thetaT_dev) hipFree(s->thetaT_dev);
  if (s->tiles_dev) hipFree(s->tiles_dev);
  if (s->mirror_dev) hipFree(s->mirror_dev);
  delete s;
  h->egm_state = nullptr;
}

static bool fill_mlp(const HostNet &n, EgmMlp &m, int off) {
  const int L = (int)n.dims.size() - 1;
  if (L < 1 || L > EGM_MAX_LAYERS) return false;
  m.n_layers = L;
  for (int i = 0; i <= L; ++i) m.dims[i] = n.dims[i];
  m.off = off;
  egm_finish_mlp(m);
  return true;
}

extern "C" int bgm_causal_egm_begin(bgm_handle *h, const bgm_egm_config *cfg, const float *theta_dz_host, int64_t count,
                                    void *stream_) {
  (void)stream_;
  if (!h || !h->configured) { bgm_set_error("bgm_causal_egm_begin: handle not configured"); return BGM_E_STATE; }
  if (!cfg || !theta_dz_host) { bgm_set_error("bgm_causal_egm_begin: NULL argument"); return BGM_E_INVALID; }
  for (int k = 0; k < 4; ++k)
    if (!h->nets[k].set) { bgm_set_error("bgm_causal_egm_begin: install g, f, h and e with bgm_causal_set_weights first"); return BGM_E_STATE; }
  if (cfg->batch_size < 2 || cfg->batch_size > 4096) { bgm_set_error("bgm_causal_egm_begin: batch_size must be in [2, 4096]"); return BGM_E_INVALID; }
  if (cfg->n_hidden_dz < 1 || cfg->n_hidden_dz + 1 > EGM_MAX_LAYERS) { bgm_set_error("bgm_causal_egm_begin: bad dz_units"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  bgm_egm_free_state(h);
  EgmState *s = new EgmState();
  h->egm_state = s;
  s->cfg = *cfg;
  EgmArgs &a = s->base;
  const HostNet &G = h->nets[BGM_NET_G], &E = h->nets[BGM_NET_E], &F = h->nets[BGM_NET_F], &H = h->nets[BGM_NET_H];
  int off = 0;
  bool ok = fill_mlp(G, a.g, off); off += (int)G.count();
  ok = ok && fill_mlp(E, a.e, off); off += (int)E.count();
  ok = ok && fill_mlp(F, a.f, off); off += (int)F.count();
  ok = ok && fill_mlp(H, a.h, off); off += (int)H.count();
  if (!ok) { bgm_egm_free_state(h); bgm_set_error("bgm_causal_egm_begin: too many layers"); return BGM_E_UNSUPPORTED; }
  s->n_gen = (size_t)off;
  // discriminator layout: W0..WL | b0..bL | gamma0..gamma(L-1) | beta0..beta(L-1)
  EgmDisc &d = a.dz;
  const int L = cfg->n_hidden_dz;
  d.n_hidden = L;
  d.dims[0] = h->q;
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
  s->n_dz = (size_t)o;
  if ((int64_t)o != count) {
    bgm_egm_free_state(h);
    bgm_set_error("bgm_causal_egm_begin: expected " + std::to_string(o) + " discriminator parameters, got " + std::to_string(count));
    return BGM_E_INVALID;
  }
  // widest layer / largest staged matrix
  int wmax = std::max(h->q, h->p + 1);
  size_t stage = 0;
  auto scan = [&](const int *dims, int n_layers) {
    for (int l = 0; l < n_layers; ++l) {
      wmax = std::max(wmax, std::max(dims[l], dims[l + 1]));
      stage = std::max(stage, (size_t)dims[l] * (size_t)(dims[l + 1] | 1));
    }
  };
  scan(a.g.dims, a.g.n_layers); scan(a.e.dims, a.e.n_layers); scan(a.f.dims, a.f.n_layers); scan(a.h.dims, a.h.n_layers);
  scan(d.dims, L + 1);
  (void)stage;
  const int B = cfg->batch_size;
  int dmax = 0, dsum = 0, dall = 0;
  for (int l = 0; l <= L; ++l) { dmax = std::max(dmax, d.dims[l]); dall += d.dims[l]; }
  for (int l = 1; l <= L; ++l) dsum += d.dims[l];
  a.dmax = dmax;
  // discriminator working set of the disc step (egm_disc_step_kernel): cache A | max(cache B + da + du, GP scratch)
  const size_t cache = ((size_t)(2 * B + 1) * dsum + B + 3) / 4 * 4;
  const size_t gp = ((size_t)B * dall + (size_t)(5 * B + 1) * dsum + 3) / 4 * 4 + 3 * (size_t)B * dmax;
  const size_t arena = cache + std::max(cache + 2 * (size_t)B * dmax, gp);
  a.disc_lds = (64 + arena) * sizeof(float) <= 160 * 1024 ? 1 : 0;
  s->lds_bytes = (int)((64 + (a.disc_lds ? arena : 0)) * sizeof(float));
  {   // register-chained discriminator step: fixed normalisation, the default layer shapes, one or two 16-row tiles
    // latent width: one input tile (q <= 16), or two (q <= 32: B = 32 only, the fourth pass's stash in global scratch)
    const int t0 = h->q <= 16 ? 1 : 2;
    bool chain = d.fixed_norm && L == 3 && d.dims[0] <= 32 && (t0 == 1 || B == 32) && (d.dims[1] + 15) / 16 == 4 && (d.dims[2] + 15) / 16 == 2 &&
                 d.dims[3] >= 1 && d.dims[3] <= 16 && (B == 16 || B == 32) && a.e.n_layers >= 2 && a.e.dims[a.e.n_layers] == h->q && h->q <= 32;
    for (int l = 1; l < a.e.n_layers; ++l) chain = chain && a.e.dims[l] == 64;
    const size_t bytes = !chain ? 0 : sizeof(float) * (size_t)(t0 == 1 ? ech_disc_lds_floats<4, 2, 1>(d, B) : ech_disc_lds_floats<4, 2, 1, 2>(d, B));
    s->chain_t0 = t0;
    if (std::getenv("BGM_EGM_NO_CHAIN")) chain = false;   // A/B switch for measurements
    s->chain_disc_lds = (chain && bytes <= 160 * 1024) ? (int)bytes : 0;
  }
  size_t gen_stash = 0;
  {   // register-chained generator step: same discriminator / encoder shapes, g with 64-wide hidden layers and a compiled output
      // extent, the head networks f, h with the 64-32-8 default
    // compiled output extents of g: 13 and 7 tiles exactly; a narrower p + 1 runs the 13-tile kernel with masked columns (B = 32)
    const int ntl_need = (h->p + 1 + 15) / 16;
    const int ntl = ((ntl_need == 13 || ntl_need == 7) && s->chain_t0 == 1) ? ntl_need : 13;      // two latent tiles: the masked 13-tile variant only
    bool chain = s->chain_disc_lds > 0 && ntl_need <= 13 && (ntl == ntl_need || B == 32) && a.g.n_layers >= 2 && a.g.dims[0] == h->q;
    for (int l = 1; l < a.g.n_layers; ++l) chain = chain && a.g.dims[l] == 64;
    for (const EgmMlp *m : {&a.f, &a.h})
      chain = chain && m->n_layers == 4 && m->dims[0] <= 16 && m->dims[1] == 64 && m->dims[2] == 32 && m->dims[3] >= 1 && m->dims[3] <= 16 &&
              m->dims[4] >= 1 && m->dims[4] <= 16;
    if (std::getenv("BGM_EGM_NO_CHAIN_GEN")) chain = false;
    if (chain) {
      EcgTab &T = s->gen_tab;
      size_t off = 0;
      auto take = [&](size_t n) { const size_t r = off; off += (n + 3) / 4 * 4; return (int)r; };
      auto tiles_of = [](int n) { return (n + 15) / 16; };
      const EgmMlp *nets[6] = {&a.g, &a.e, &a.e, &a.g, &a.f, &a.h};
      int xw[6][EGM_MAX_LAYERS], dw[6][EGM_MAX_LAYERS];
      for (int ps = 0; ps < 6; ++ps) {
        const EgmMlp &m = *nets[ps];
        for (int l = 0; l < m.n_layers; ++l) {
          xw[ps][l] = 16 * tiles_of(m.dims[l]);
          dw[ps][l] = 16 * tiles_of(m.dims[l + 1]);
        }
        if (nets[ps] == &a.e) xw[ps][0] = 16 * ntl;                   // the encoder's input tiles are the generator's output tiles
        if (nets[ps] == &a.g) dw[ps][m.n_layers - 1] = 16 * ntl;
        for (int l = 0; l < m.n_layers; ++l) { T.x[ps][l] = take((size_t)B * xw[ps][l]); T.d[ps][l] = take((size_t)B * dw[ps][l]); }
      }
      gen_stash = off;
      // weight-gradient tiles: (layer, 16 input features, 16 output features), the passes that feed the layer
      auto add = [&](const EgmMlp &m, int p0, int p1) {
        for (int l = 0; l < m.n_layers; ++l) {
          const int ni = m.dims[l], no = m.dims[l + 1];
          for (int u = 0; u < tiles_of(ni); ++u)
            for (int v = 0; v < tiles_of(no); ++v) {
              int e[ECG_TILE_INTS] = {T.x[p0][l], p1 >= 0 ? T.x[p1][l] : -1, T.d[p0][l], p1 >= 0 ? T.d[p1][l] : -1, xw[p0][l], dw[p0][l], u, v,
                                      m.woff[l], ni, no, u == 0 ? m.woff[l] + ni * no : -1, 0, 0, 0, 0};
              s->gen_tiles.insert(s->gen_tiles.end(), e, e + ECG_TILE_INTS);
            }
        }
      };
      add(a.g, ECG_PASS_G1, ECG_PASS_G2); add(a.e, ECG_PASS_E2, ECG_PASS_E1); add(a.f, ECG_PASS_F, -1); add(a.h, ECG_PASS_H, -1);
      T.n_tiles = (int)(s->gen_tiles.size() / ECG_TILE_INTS);
      T.n_warm = (int)s->n_gen;
      s->chain_ntl = ntl;
      s->chain_pad = ntl != ntl_need || s->chain_t0 == 2;
      s->chain_gen_lds = (int)(sizeof(float) * (size_t)(s->chain_t0 == 1 ? ecg_lds_floats<4, 2, 1>(d, B) : ecg_lds_floats<4, 2, 1, 2>(d, B)));
    }
  }
  a.n_gen = (int)s->n_gen; a.B = B; a.q = h->q; a.p = h->p; a.wmax = wmax;
  a.z0 = h->cfg.z_dims[0]; a.z1 = h->cfg.z_dims[1]; a.z2 = h->cfg.z_dims[2];
  a.binary = h->cfg.binary_treatment; a.use_z_rec = cfg->use_z_rec;
  // workspace bound: every buffer of either kernel is [B x width <= wmax]; count them generously
  size_t acts = 0;
  auto widths = [&](const int *dims, int n_layers) { size_t t = 0; for (int l = 0; l <= n_layers; ++l) t += dims[l] + 4; return t; };
  acts += 2 * widths(a.g.dims, a.g.n_layers) + 2 * widths(a.e.dims, a.e.n_layers) + widths(a.f.dims, a.f.n_layers) +
          widths(a.h.dims, a.h.n_layers) + 3 * 3 * widths(d.dims, L + 1) + 12 * widths(d.dims, L + 1);
  s->ws_floats = std::max((size_t)B * (acts + 16 * (size_t)wmax + 4 * (size_t)h->p + 64) + s->n_dz + arena + 4096, gen_stash + 64);
  const size_t total = 4 * s->n_gen + 4 * s->n_dz + s->ws_floats + 64;
  BGM_HIP_CHECK(hipMalloc(&s->dev, total * sizeof(float)));
  BGM_HIP_CHECK(hipMemset(s->dev, 0, total * sizeof(float)));
  float *p = s->dev;
  auto take = [&](size_t n) { float *r = p; p += (n + 3) / 4 * 4; return r; };
  a.theta_g = take(s->n_gen); a.m_g = take(s->n_gen); a.v_g = take(s->n_gen); a.grad_g = take(s->n_gen);
  a.theta_d = take(s->n_dz); a.m_d = take(s->n_dz); a.v_d = take(s->n_dz); a.grad_d = take(s->n_dz);
  a.ws = take(s->ws_floats);
  std::vector<float> tg;
  tg.reserve(s->n_gen);
  for (const HostNet *n : {&G, &E, &F, &H}) tg.insert(tg.end(), n->theta.begin(), n->theta.end());
  BGM_HIP_CHECK(hipMemcpy(a.theta_g, tg.data(), tg.size() * sizeof(float), hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMemcpy(a.theta_d, theta_dz_host, s->n_dz * sizeof(float), hipMemcpyHostToDevice));
  if (s->chain_gen_lds > 0) {
    // transposed mirror of the weight matrices (the backward chains read W^T rows contiguously); kept current by the kernel's Adam
    std::vector<float> tT(s->n_gen, 0.0f);
    std::vector<int> mdst(s->n_gen, -1);
    auto mirror = [&](const EgmMlp &m) {
      for (int l = 0; l < m.n_layers; ++l) {
        const int ni = m.dims[l], no = m.dims[l + 1];
        for (int f = 0; f < ni; ++f)
          for (int o = 0; o < no; ++o) {
            tT[m.woff[l] + (size_t)o * ni + f] = tg[m.woff[l] + (size_t)f * no + o];
            mdst[m.woff[l] + (size_t)f * no + o] = m.woff[l] + o * ni + f;
          }
      }
    };
    mirror(a.g); mirror(a.e); mirror(a.f); mirror(a.h);
    BGM_HIP_CHECK(hipMalloc(&s->mirror_dev, sizeof(int) * s->n_gen));
    BGM_HIP_CHECK(hipMemcpy(s->mirror_dev, mdst.data(), sizeof(int) * s->n_gen, hipMemcpyHostToDevice));
    BGM_HIP_CHECK(hipMalloc(&s->thetaT_dev, sizeof(float) * (s->n_gen + 64)));     // K-contiguous tile loads read up to 15 floats past a row
    BGM_HIP_CHECK(hipMemset(s->thetaT_dev, 0, sizeof(float) * (s->n_gen + 64)));
    BGM_HIP_CHECK(hipMemcpy(s->thetaT_dev, tT.data(), sizeof(float) * s->n_gen, hipMemcpyHostToDevice));
    BGM_HIP_CHECK(hipMalloc(&s->tiles_dev, sizeof(int) * s->gen_tiles.size()));
    BGM_HIP_CHECK(hipMemcpy(s->tiles_dev, s->gen_tiles.data(), sizeof(int) * s->gen_tiles.size(), hipMemcpyHostToDevice));
    s->gen_tab.thetaT = s->thetaT_dev;
    s->gen_tab.tiles = s->tiles_dev;
  }
  return BGM_OK;
}

#ifdef EGM_PHASE_CLOCK
#include <cstdio>
static unsigned long long *egm_stamp_buf() {
  static unsigned long long *buf = nullptr;
  if (!buf) hipMalloc(&buf, sizeof(unsigned long long) * 8192);
  return buf;
}
static void egm_stamp_report(const char *what, int *calls) {
  if (++*calls != 40) return;
  hipDeviceSynchronize();
  std::vector<unsigned long long> st(8192);
  hipMemcpy(st.data(), egm_stamp_buf(), sizeof(unsigned long long) * 8192, hipMemcpyDeviceToHost);
  std::fprintf(stderr, "EGM_PHASE %s\n", what);
  for (int i = 1; i < 4096 && st[2 * i]; ++i)
    std::fprintf(stderr, "EGM_PHASE %s %3d line %4llu cycles %8llu\n", what, i, st[2 * i], st[2 * i + 1] - st[2 * i - 1]);
}
#endif

#ifdef EGM_PHASE_CLOCK
static void egm_chain_report(const char *what, int *calls) {
  if (++*calls != 40) return;
  hipDeviceSynchronize();
  std::vector<unsigned long long> st(8192);
  hipMemcpy(st.data(), egm_stamp_buf(), sizeof(unsigned long long) * 8192, hipMemcpyDeviceToHost);
  for (int w = 0; w < 8; ++w) {
    std::fprintf(stderr, "ECH_PHASE %s wave %d:", what, w);
    for (int k = 1; k < 16; ++k) if (st[4096 + w * 16 + k] && st[4096 + w * 16 + k - 1]) std::fprintf(stderr, " %d:%llu", k, st[4096 + w * 16 + k] - st[4096 + w * 16 + k - 1]);
    if (st[4096 + w * 16 + 8]) std::fprintf(stderr, "  enc_start-k0:%llu", st[4096 + w * 16 + 8] - st[4096 + w * 16]);
    std::fprintf(stderr, "\n");
  }
}
#endif

static EgmAdam egm_adam_coeffs(float lr, long long t) {
  EgmAdam ad;
  ad.b1 = EGM_B1; ad.b2 = EGM_B2; ad.eps = EGM_ADAM_EPS;
  ad.lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow((double)EGM_B2, (double)t)) / (1.0 - std::pow((double)EGM_B1, (double)t)));
  return ad;
}

extern "C" int bgm_causal_egm_disc_step(bgm_handle *h, const float *z_dev, const int32_t *idx_dev, const float *v_dev, float eps,
                                        int32_t apply, float *out_dev, void *stream_) {
  if (!h || !h->egm_state) { bgm_set_error("bgm_causal_egm_disc_step: call bgm_causal_egm_begin first"); return BGM_E_STATE; }
  if (!z_dev || !idx_dev || !v_dev) { bgm_set_error("bgm_causal_egm_disc_step: NULL pointer"); return BGM_E_INVALID; }
  EgmState *s = est(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  EgmArgs a = s->base;
  a.z = z_dev; a.idx = idx_dev; a.v = v_dev; a.x = nullptr; a.y = nullptr; a.eps = eps; a.out = out_dev; a.apply = apply ? 1 : 0;
  if (apply) s->t_d += 1;
  a.adam = egm_adam_coeffs(s->cfg.lr, std::max<long long>(1, s->t_d));
  if (s->chain_disc_lds > 0) {
    const int kt0 = (a.p + 15) / 16;     // compiled first-layer extents: p in (192, 208] and (96, 112]; any other p takes the streaming variant
    auto kc = s->chain_t0 == 2 ? egm_disc_chain_kernel<4, 0, 4, 2, 1, 2, 2> :
              a.B == 32 ? (kt0 == 13 ? egm_disc_chain_kernel<4, 13, 4, 2, 1, 2> : kt0 == 7 ? egm_disc_chain_kernel<4, 7, 4, 2, 1, 2> : egm_disc_chain_kernel<4, 0, 4, 2, 1, 2>)
                        : egm_disc_chain_kernel<4, 0, 4, 2, 1, 1>;
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, s->chain_disc_lds));
#ifdef EGM_PHASE_CLOCK
    a.stamps = egm_stamp_buf();
    hipMemsetAsync(a.stamps, 0, sizeof(unsigned long long) * 8192, (hipStream_t)stream_);
#endif
    hipLaunchKernelGGL(kc, dim3(1), dim3(ECH_THREADS), s->chain_disc_lds, (hipStream_t)stream_, a);
    BGM_HIP_CHECK(hipGetLastError());
#ifdef EGM_PHASE_CLOCK
    { static int calls = 0; egm_chain_report("disc", &calls); }
#endif
    return BGM_OK;
  }
  auto k = egm_disc_step_kernel;
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes));
#ifdef EGM_PHASE_CLOCK
  a.stamps = egm_stamp_buf();
  hipMemsetAsync(a.stamps, 0, sizeof(unsigned long long) * 8192, (hipStream_t)stream_);
#endif
  hipLaunchKernelGGL(k, dim3(1), dim3(EGM_THREADS), s->lds_bytes, (hipStream_t)stream_, a);
  BGM_HIP_CHECK(hipGetLastError());
#ifdef EGM_PHASE_CLOCK
  { static int calls = 0; egm_stamp_report("disc", &calls); }
#endif
  return BGM_OK;
}

extern "C" int bgm_causal_egm_gen_step(bgm_handle *h, const float *z_dev, const int32_t *idx_dev, const float *v_dev,
                                       const float *x_dev, const float *y_dev, int32_t apply, float *out_dev, void *stream_) {
  if (!h || !h->egm_state) { bgm_set_error("bgm_causal_egm_gen_step: call bgm_causal_egm_begin first"); return BGM_E_STATE; }
  if (!z_dev || !idx_dev || !v_dev || !x_dev || !y_dev) { bgm_set_error("bgm_causal_egm_gen_step: NULL pointer"); return BGM_E_INVALID; }
  EgmState *s = est(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  EgmArgs a = s->base;
  a.z = z_dev; a.idx = idx_dev; a.v = v_dev; a.x = x_dev; a.y = y_dev; a.eps = 0.0f; a.out = out_dev; a.apply = apply ? 1 : 0;
  if (apply) s->t_g += 1;
  a.adam = egm_adam_coeffs(s->cfg.lr, std::max<long long>(1, s->t_g));
  if (s->chain_gen_lds > 0) {
    auto kc = s->chain_t0 == 2 ? egm_gen_chain_kernel<4, 13, 4, 2, 1, 2, true, 2>
              : s->chain_pad ? egm_gen_chain_kernel<4, 13, 4, 2, 1, 2, true>
              : a.B == 32 ? (s->chain_ntl == 13 ? egm_gen_chain_kernel<4, 13, 4, 2, 1, 2> : egm_gen_chain_kernel<4, 7, 4, 2, 1, 2>)
                        : (s->chain_ntl == 13 ? egm_gen_chain_kernel<4, 13, 4, 2, 1, 1> : egm_gen_chain_kernel<4, 7, 4, 2, 1, 1>);
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, s->chain_gen_lds));
#ifdef EGM_PHASE_CLOCK
    a.stamps = egm_stamp_buf();
    hipMemsetAsync(a.stamps, 0, sizeof(unsigned long long) * 8192, (hipStream_t)stream_);
#endif
    hipLaunchKernelGGL(kc, dim3(1), dim3(ECH_THREADS), s->chain_gen_lds, (hipStream_t)stream_, a, s->gen_tab);
    BGM_HIP_CHECK(hipGetLastError());
    auto kd = a.B == 32 ? egm_gen_dw_kernel<2> : egm_gen_dw_kernel<1>;
    hipLaunchKernelGGL(kd, dim3((s->gen_tab.n_tiles + ECH_WAVES - 1) / ECH_WAVES), dim3(ECH_THREADS), 0, (hipStream_t)stream_, a, s->gen_tab);
    BGM_HIP_CHECK(hipGetLastError());
#ifdef EGM_PHASE_CLOCK
    { static int calls = 0; egm_chain_report("gen", &calls); }
#endif
    return BGM_OK;
  }
  auto k = egm_gen_step_kernel;
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes));
#ifdef EGM_PHASE_CLOCK
  a.stamps = egm_stamp_buf();
  hipMemsetAsync(a.stamps, 0, sizeof(unsigned long long) * 8192, (hipStream_t)stream_);
#endif
  hipLaunchKernelGGL(k, dim3(1), dim3(EGM_THREADS), s->lds_bytes, (hipStream_t)stream_, a);
  BGM_HIP_CHECK(hipGetLastError());
#ifdef EGM_PHASE_CLOCK
  { static int calls = 0; egm_stamp_report("gen", &calls); }
#endif
  return BGM_OK;
}

extern "C" int bgm_causal_egm_grad(bgm_handle *h, int32_t which, float scale, float *grad_dev, int64_t count, void *stream_) {
  if (!h || !h->egm_state) { bgm_set_error("bgm_causal_egm_grad: call bgm_causal_egm_begin first"); return BGM_E_STATE; }
  EgmState *s = est(h);
  const size_t n = which == 0 ? s->n_gen : s->n_dz;
  if ((which != 0 && which != 1) || !grad_dev || (size_t)count != n) {
    bgm_set_error("bgm_causal_egm_grad: which must be 0 (g|e|f|h: " + std::to_string(s->n_gen) + " floats) or 1 (discriminator: " + std::to_string(s->n_dz) + ")");
    return BGM_E_INVALID;
  }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  hipLaunchKernelGGL(egm_dp_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                     which == 0 ? s->base.grad_g : s->base.grad_d, grad_dev, (int)n, scale);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_causal_egm_apply(bgm_handle *h, int32_t which, const float *grad_dev, int64_t count, void *stream_) {
  if (!h || !h->egm_state) { bgm_set_error("bgm_causal_egm_apply: call bgm_causal_egm_begin first"); return BGM_E_STATE; }
  EgmState *s = est(h);
  const size_t n = which == 0 ? s->n_gen : s->n_dz;
  if ((which != 0 && which != 1) || !grad_dev || (size_t)count != n) { bgm_set_error("bgm_causal_egm_apply: bad which / count"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  const EgmArgs &a = s->base;
  const EgmAdam ad = egm_adam_coeffs(s->cfg.lr, which == 0 ? ++s->t_g : ++s->t_d);
  if (which == 0)
    hipLaunchKernelGGL(egm_dp_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, a.theta_g, a.m_g, a.v_g, grad_dev, (int)n, ad,
                       s->thetaT_dev, (const int *)s->mirror_dev);
  else
    hipLaunchKernelGGL(egm_dp_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, a.theta_d, a.m_d, a.v_d, grad_dev, (int)n, ad,
                       (float *)nullptr, (const int *)nullptr);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_causal_egm_read(bgm_handle *h, int32_t what, float *host, int64_t count, void *stream_) {
  if (!h || !h->egm_state || !host) { bgm_set_error("bgm_causal_egm_read: no session / NULL"); return BGM_E_STATE; }
  EgmState *s = est(h);
  const float *src = nullptr;
  size_t n = 0;
  switch (what) {
    case 0: src = s->base.theta_g; n = s->n_gen; break;
    case 1: src = s->base.theta_d; n = s->n_dz; break;
    case 2: src = s->base.grad_g; n = s->n_gen; break;
    case 3: src = s->base.grad_d; n = s->n_dz; break;
    default: bgm_set_error("bgm_causal_egm_read: what must be 0..3"); return BGM_E_INVALID;
  }
  if ((size_t)count != n) { bgm_set_error("bgm_causal_egm_read: expected " + std::to_string(n) + " floats"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  BGM_HIP_CHECK(hipMemcpy(host, src, n * sizeof(float), hipMemcpyDeviceToHost));
  return BGM_OK;
}

extern "C" int bgm_causal_egm_sync(bgm_handle *h, void *stream_) {
  if (!h || !h->egm_state) { bgm_set_error("bgm_causal_egm_sync: no session"); return BGM_E_STATE; }
  EgmState *s = est(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  std::vector<float> tg(s->n_gen);
  BGM_HIP_CHECK(hipMemcpy(tg.data(), s->base.theta_g, s->n_gen * sizeof(float), hipMemcpyDeviceToHost));
  size_t off = 0;
  for (int id : {BGM_NET_G, BGM_NET_E, BGM_NET_F, BGM_NET_H}) {
    HostNet &n = h->nets[id];
    std::memcpy(n.theta.data(), tg.data() + off, n.count() * sizeof(float));
    off += n.count();
  }
  h->blob_valid = false; h->bx_valid = false; h->eblob_valid = false;
  h->det_valid = false;   // the general sampling path's packed copy follows the refreshed networks as well
  h->gx_valid = false;
  return BGM_OK;
}

extern "C" int bgm_causal_egm_end(bgm_handle *h, void *stream_) {
  if (!h) return BGM_E_INVALID;
  if (!h->egm_state) return BGM_OK;
  int rc = bgm_causal_egm_sync(h, stream_);
  bgm_egm_free_state(h);
  return rc;
}
