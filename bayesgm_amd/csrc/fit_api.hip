// fit_api.hip -- C-ABI entry points of the CausalBGM iterative-update step functions.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "bgm_host.h"
#include "comm_host.h"
#include "bnf_det_host.h"
#include "gx_host.h"
#include "fit_kernels.h"
#include "fit_chain.h"

#include <cstdlib>
#include <vector>

// ---- row-tile-chain step kernels (fit_chain.h) for the reference batch sizes
struct FitChainState {
  FitChainArgs base{};
  int ntl = 0;
  int t0 = 1;            // latent input tiles of g (q <= 16 t0); two tiles run the padded B = 32 variant only
  bool pad = false;      // the 13-tile kernels on a narrower p + 1 (B = 32 only)
  float *thetaT = nullptr, *ws = nullptr;
  float *theta_x[3] = {}, *thetaT_x[3] = {};        // further parameter buffers + their transposed mirrors (bgm_causal_fit_epoch)
  const float *theta_use = nullptr, *thetaT_use = nullptr;   // when set: the buffers the next chain launch reads
  float *ws_z = nullptr;   // the latent phase's own stash, so that it may overlap the next minibatch's theta phase (bgm_causal_fit_epoch)
  int *tiles = nullptr, *mirror_dst = nullptr;
  FitSync sync_t{}, sync_z{};     // device-side ordering of the next theta-phase / latent-phase launch (bgm_causal_fit_epoch)
  const int *rp_idx = nullptr, *rp_tlast = nullptr; int rp_n = 0, rp_t_to = 0; float rp_lr = 0.0f;   // rider of the next latent-phase launch
  int last_z_blocks = 0;          // workgroups of the last latent-phase launch (what its done counter advanced by)
  FitAdamTheta fused{};    // fused.on: the next theta-phase launch applies the Adam step in its gradient-tile kernel (bgm_causal_fit_epoch)
};
static void fit_chain_free(bgm_handle *h) {
  FitChainState *c = static_cast<FitChainState *>(h->fit_chain);
  if (!c) return;
  for (void *p : {(void *)c->thetaT, (void *)c->ws, (void *)c->ws_z, (void *)c->tiles, (void *)c->mirror_dst, (void *)c->theta_x[0], (void *)c->thetaT_x[0],
                  (void *)c->theta_x[1], (void *)c->thetaT_x[1], (void *)c->theta_x[2], (void *)c->thetaT_x[2]})
    if (p) hipFree(p);
  delete c;
  h->fit_chain = nullptr;
}
static bool fit_fill_mlp(const HostNet &n, EgmMlp &m, int off) {
  const int L = (int)n.dims.size() - 1;
  if (L < 1 || L > EGM_MAX_LAYERS) return false;
  m.n_layers = L;
  for (int i = 0; i <= L; ++i) m.dims[i] = n.dims[i];
  m.off = off;
  egm_finish_mlp(m);
  return true;
}
// called at the end of bgm_causal_fit_begin (theta_dev holds [g | f | h])
static int fit_chain_setup(bgm_handle *h, const std::vector<float> &theta) {
  const HostNet &G = h->nets[BGM_NET_G], &F = h->nets[BGM_NET_F], &H = h->nets[BGM_NET_H];
  const int ng = (int)G.count(), nf = (int)F.count(), np = h->n_params, q = h->q, p = h->p;
  const int ntl_need = (p + 1 + 15) / 16;
  const int t0 = q <= 16 ? 1 : 2;
  const int ntl = ((ntl_need == 13 || ntl_need == 7) && t0 == 1) ? ntl_need : 13;
  FitChainArgs a{};
  bool ok = !std::getenv("BGM_FIT_NO_CHAIN") && q <= 32 && ntl_need <= 13 && fit_fill_mlp(G, a.g, 0) && fit_fill_mlp(F, a.f, ng) &&
            fit_fill_mlp(H, a.h, ng + nf);
  ok = ok && a.g.n_layers >= 3 && a.g.dims[0] == q && a.g.dims[a.g.n_layers] == p + 1;
  for (int l = 1; ok && l < a.g.n_layers; ++l) ok = a.g.dims[l] == 64;
  for (const EgmMlp *m : {&a.f, &a.h})
    ok = ok && m->n_layers == 4 && m->dims[0] <= 16 && m->dims[1] == 64 && m->dims[2] == 32 && m->dims[3] >= 1 && m->dims[3] <= 16 && m->dims[4] == 2;
  if (!ok) return BGM_OK;
  FitChainState *c = new FitChainState();
  h->fit_chain = c;
  c->ntl = ntl;
  c->t0 = t0;
  c->pad = ntl != ntl_need || t0 == 2;
  const int B = 32;
  auto tl = [](int n) { return (n + 15) / 16; };
  size_t off = 0;
  auto take = [&](size_t n) { const size_t r = off; off += (n + 3) / 4 * 4; return (int)r; };
  const EgmMlp *nets[3] = {&a.g, &a.f, &a.h};
  int xw[3][EGM_MAX_LAYERS], dw[3][EGM_MAX_LAYERS];
  std::vector<int> tiles, mdst(np, -1);
  std::vector<float> tT((size_t)np + 64, 0.0f);
  for (int k = 0; k < 3; ++k) {
    const EgmMlp &m = *nets[k];
    for (int l = 0; l < m.n_layers; ++l) {
      xw[k][l] = 16 * tl(m.dims[l]); dw[k][l] = 16 * tl(m.dims[l + 1]);
      if (k == 0 && l == m.n_layers - 1) dw[k][l] = 16 * ntl;
      a.xo[k][l] = take((size_t)B * xw[k][l]); a.dofs[k][l] = take((size_t)B * dw[k][l]);
    }
    for (int l = 0; l < m.n_layers; ++l) {
      const int ni = m.dims[l], no = m.dims[l + 1];
      for (int f = 0; f < ni; ++f)
        for (int o = 0; o < no; ++o) {
          mdst[m.woff[l] + f * no + o] = m.woff[l] + o * ni + f;
          tT[m.woff[l] + (size_t)o * ni + f] = theta[m.woff[l] + (size_t)f * no + o];
        }
      for (int u = 0; u < tl(ni); ++u)
        for (int v = 0; v < tl(no); ++v) {
          int e[ECG_TILE_INTS] = {a.xo[k][l], -1, a.dofs[k][l], -1, xw[k][l], dw[k][l], u, v, m.woff[l], ni, no, u == 0 ? m.woff[l] + ni * no : -1, 0, 0, 0, 0};
          tiles.insert(tiles.end(), e, e + ECG_TILE_INTS);
        }
    }
  }
  a.n_tiles = (int)(tiles.size() / ECG_TILE_INTS);
  BGM_HIP_CHECK(hipMalloc((void **)&c->ws, sizeof(float) * (off + 64)));
  BGM_HIP_CHECK(hipMemset(c->ws, 0, sizeof(float) * (off + 64)));
  BGM_HIP_CHECK(hipMalloc((void **)&c->ws_z, sizeof(float) * (off + 64)));
  BGM_HIP_CHECK(hipMemset(c->ws_z, 0, sizeof(float) * (off + 64)));
  BGM_HIP_CHECK(hipMalloc((void **)&c->thetaT, sizeof(float) * tT.size()));
  BGM_HIP_CHECK(hipMemcpy(c->thetaT, tT.data(), sizeof(float) * tT.size(), hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMalloc((void **)&c->tiles, sizeof(int) * tiles.size()));
  BGM_HIP_CHECK(hipMemcpy(c->tiles, tiles.data(), sizeof(int) * tiles.size(), hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMalloc((void **)&c->mirror_dst, sizeof(int) * np));
  BGM_HIP_CHECK(hipMemcpy(c->mirror_dst, mdst.data(), sizeof(int) * np, hipMemcpyHostToDevice));
  a.theta = h->theta_dev; a.thetaT = c->thetaT; a.ws = c->ws; a.tiles = c->tiles; a.n_warm = np;
  a.q = q; a.p = p; a.z0 = h->cfg.z_dims[0]; a.z1 = h->cfg.z_dims[1]; a.z2 = h->cfg.z_dims[2];
  a.binary = h->cfg.binary_treatment;
  a.sig2_v = h->meta.sig2_v; a.sig2_x = h->meta.sig2_x; a.sig2_y = h->meta.sig2_y;
  c->base = a;
  return BGM_OK;
}
// one launch of the chains; Z_MODE 0 also the gradient tiles into `grad`
static void fit_chain_launch(FitChainState *c, FitChainArgs &a, int batch, int z_mode, hipStream_t stream) {
  if (c->theta_use) { a.theta = c->theta_use; a.thetaT = c->thetaT_use; }
  a.ad = z_mode ? FitAdamTheta{} : c->fused;
  a.sy = z_mode ? c->sync_z : c->sync_t;
  static const bool one_wg_ = std::getenv("BGM_FIT_ONE_WG") != nullptr;
  a.n_valid = batch;                                 // rows of this minibatch; the tile rows behind them are masked
  const int nb = batch <= 16 ? 1 : 2;                // row tiles (the padded / two-k-tile instantiations are compiled for two only)
  // latent phase: workgroups of the chains, then the riders that replay a later minibatch's rows (bgm_causal_fit_epoch)
  const int nchain = (c->t0 != 2 && !c->pad && nb == 2 && !one_wg_ && (c->ntl == 13 || c->ntl == 7)) ? 2 : 1;
  const int rpb = (z_mode && c->rp_n > 0 && a.zm) ? (int)(((long long)c->rp_n * a.q * 16 + ECH_THREADS - 1) / ECH_THREADS) : 0;
  a.rp_first = nchain; a.rp_n = rpb ? c->rp_n : 0; a.rp_idx = c->rp_idx; a.rp_t_to = c->rp_t_to; a.rp_lr = c->rp_lr; a.rp_tlast = c->rp_tlast;
  if (z_mode) c->last_z_blocks = nchain + rpb;
#define FC(NTL_, NB_) \
  if (c->ntl == NTL_ && nb == NB_) { \
    if (z_mode) hipLaunchKernelGGL((fit_chain_kernel<4, NTL_, 4, 2, 1, NB_, 1>), dim3(nchain + rpb), dim3(ECH_THREADS), 0, stream, a); \
    else { \
      hipLaunchKernelGGL((fit_chain_kernel<4, NTL_, 4, 2, 1, NB_, 0>), dim3(1, one_wg_ ? 1 : 3), dim3(ECH_THREADS), 0, stream, a); \
      hipLaunchKernelGGL(fit_chain_dw_kernel<NB_>, dim3((a.n_tiles + ECH_WAVES - 1) / ECH_WAVES), dim3(ECH_THREADS), 0, stream, a); \
    } \
  }
  if (c->t0 == 2) {
    if (z_mode) hipLaunchKernelGGL((fit_chain_kernel<4, 13, 4, 2, 1, 2, 1, true, 2>), dim3(nchain + rpb), dim3(ECH_THREADS), 0, stream, a);
    else {
      hipLaunchKernelGGL((fit_chain_kernel<4, 13, 4, 2, 1, 2, 0, true, 2>), dim3(1), dim3(ECH_THREADS), 0, stream, a);
      hipLaunchKernelGGL(fit_chain_dw_kernel<2>, dim3((a.n_tiles + ECH_WAVES - 1) / ECH_WAVES), dim3(ECH_THREADS), 0, stream, a);
    }
    return;
  }
  if (c->pad) {
    if (z_mode) hipLaunchKernelGGL((fit_chain_kernel<4, 13, 4, 2, 1, 2, 1, true>), dim3(nchain + rpb), dim3(ECH_THREADS), 0, stream, a);
    else {
      hipLaunchKernelGGL((fit_chain_kernel<4, 13, 4, 2, 1, 2, 0, true>), dim3(1), dim3(ECH_THREADS), 0, stream, a);
      hipLaunchKernelGGL(fit_chain_dw_kernel<2>, dim3((a.n_tiles + ECH_WAVES - 1) / ECH_WAVES), dim3(ECH_THREADS), 0, stream, a);
    }
    return;
  }
  // 17..32 rows: the two row tiles on two workgroups -- on one CU the six chain waves share one address unit for their weight
  // streams (83.0 -> 77.4 us per minibatch at N = 1e6) -- and, in the theta phase (no cross-network term), one network per workgroup
  // as well (-> 75.8 us); the gradient tiles run over both row tiles as before.  BGM_FIT_ONE_WG=1: everything in one workgroup (dev A/B)
  static const bool one_wg = std::getenv("BGM_FIT_ONE_WG") != nullptr;
  static const bool no_ws = std::getenv("BGM_FIT_NO_WORKERS") != nullptr;      // dev A/B: g's last layer on the chain wave alone
  if (nb == 2 && !one_wg && (c->ntl == 13 || c->ntl == 7)) {
#define FS(NTL_) \
    if (c->ntl == NTL_) { \
      if (z_mode) { \
        if (no_ws) hipLaunchKernelGGL((fit_chain_kernel<4, NTL_, 4, 2, 1, 1, 1>), dim3(nchain + rpb), dim3(ECH_THREADS), 0, stream, a); \
        else hipLaunchKernelGGL((fit_chain_kernel<4, NTL_, 4, 2, 1, 1, 1, false, 1, true>), dim3(nchain + rpb), dim3(ECH_THREADS), 0, stream, a); \
      } else { \
        if (no_ws) hipLaunchKernelGGL((fit_chain_kernel<4, NTL_, 4, 2, 1, 1, 0>), dim3(2, 3), dim3(ECH_THREADS), 0, stream, a); \
        else hipLaunchKernelGGL((fit_chain_kernel<4, NTL_, 4, 2, 1, 1, 0, false, 1, true>), dim3(2, 3), dim3(ECH_THREADS), 0, stream, a); \
        hipLaunchKernelGGL(fit_chain_dw_kernel<2>, dim3((a.n_tiles + ECH_WAVES - 1) / ECH_WAVES), dim3(ECH_THREADS), 0, stream, a); \
      } \
    }
    FS(13) FS(7)
#undef FS
    return;
  }
  if (nb == 1 && !one_wg && !no_ws && (c->ntl == 13 || c->ntl == 7)) {      // <= 16 rows (a rank's share under data parallelism): the same worker split
#define FW(NTL_) \
    if (c->ntl == NTL_) { \
      if (z_mode) hipLaunchKernelGGL((fit_chain_kernel<4, NTL_, 4, 2, 1, 1, 1, false, 1, true>), dim3(nchain + rpb), dim3(ECH_THREADS), 0, stream, a); \
      else { \
        hipLaunchKernelGGL((fit_chain_kernel<4, NTL_, 4, 2, 1, 1, 0, false, 1, true>), dim3(1, 3), dim3(ECH_THREADS), 0, stream, a); \
        hipLaunchKernelGGL(fit_chain_dw_kernel<1>, dim3((a.n_tiles + ECH_WAVES - 1) / ECH_WAVES), dim3(ECH_THREADS), 0, stream, a); \
      } \
    }
    FW(13) FW(7)
#undef FW
    return;
  }
  FC(13, 2) FC(13, 1) FC(7, 2) FC(7, 1)
#undef FC
#ifdef FITC_CLOCK
  {
    static int calls = 0;
    if (++calls == 300 || calls == 301) {
      hipStreamSynchronize(stream);
      unsigned long long t[64];
      hipMemcpyFromSymbol(t, HIP_SYMBOL(g_fitc), sizeof(t));
      fprintf(stderr, "[FITC z_mode=%d] in %llu | L0 %llu | hidden fwd %llu (", z_mode, t[1] - t[0], t[2] - t[1], t[3] - t[2]);
      for (int l = 1; l <= 4; ++l) fprintf(stderr, "k%llu+e%llu ", t[30 + 2 * l] - (l == 1 ? t[2] : t[31 + 2 * (l - 1)]), t[31 + 2 * l] - t[30 + 2 * l]);
      fprintf(stderr, ") | last %llu | loss %llu | bwd last %llu | hidden bwd %llu (", t[4] - t[3], t[6] - t[5], t[7] - t[6], t[8] - t[7]);
      for (int l = 4; l >= 1; --l) fprintf(stderr, "k%llu+e%llu ", t[40 + 2 * l] - (l == 4 ? t[7] : t[41 + 2 * (l + 1)]), t[41 + 2 * l] - t[40 + 2 * l]);
      fprintf(stderr, ") | tail %llu | end %llu | total %llu\n", t[9] - t[8], t[10] - t[9], t[10] - t[0]);
    }
  }
#endif
}

static constexpr int FIT_WAVES = 8;
static constexpr float ADAM_B1 = 0.9f, ADAM_B2 = 0.99f, ADAM_EPS = 1e-7f;  // causalbgm/base.py:90-93

// Transposed ("backward") packing: a forward layer W [n_in x n_out] becomes the layer
// Wt [rows = out features (KT_b tiles)] -> [cols = in features (NT_b tiles)];  colmap(c) gives the
// canonical input row of W for packed output column c (or -1).
template <class ColMap>
static void pack_layer_t(std::vector<float> &blob, int off, const float *W, int n_in, int n_out, int KT_b,
                         int NT_b, ColMap colmap) {
  const int K_ROWS = 16 * KT_b;
  for (int T0 = 0; T0 < NT_b;) {
    const int GS = group_size(NT_b - T0);
    float *base = blob.data() + off + K_ROWS * 16 * T0;
    for (int rho = 0; rho < K_ROWS; ++rho)   // rho = output feature of the forward layer
      for (int j = 0; j < 16; ++j)
        for (int u = 0; u < GS; ++u) {
          const int src = colmap(16 * (T0 + u) + j);
          float val = 0.0f;
          if (src >= 0 && src < n_in && rho < n_out) val = W[(size_t)src * n_out + rho];
          base[(rho * 16 + j) * GS + u] = val;
        }
    T0 += GS;
  }
}

static void fit_layout_backward(bgm_handle *h, FitMeta &bm) {
  const int KT1 = h->KT1, NTL = h->NTL;
  int off = 0;
  auto take = [&](int n) { int o = off; off += (n + 3) / 4 * 4; return o; };
  bm.w1g = take(64 * 16 * KT1); bm.w1f = take(64 * 16 * KT1); bm.w1h = take(64 * 16 * KT1);
  bm.wg = take(h->meta.n_gh * 4096);
  bm.wgl = take(16 * NTL * 64);
  bm.wf2 = take(32 * 64); bm.wf3 = take(16 * 32); bm.wf4 = take(16 * 16);
  bm.wh2 = take(32 * 64); bm.wh3 = take(16 * 32); bm.wh4 = take(16 * 16);
  bm.total = off;
}

static void causal_pack_backward(bgm_handle *h, const HostNet &G, const HostNet &F, const HostNet &H,
                                 const FitMeta &bm, std::vector<float> &blob) {
  const int q = h->q, p = h->p, KT1 = h->KT1, NTL = h->NTL;
  const int z0 = h->cfg.z_dims[0], z1 = h->cfg.z_dims[1], z2 = h->cfg.z_dims[2];
  blob.assign(bm.total, 0.0f);
  auto ident = [](int c) { return c; };
  // first layers: packed output column c <-> extended input feature l1_feature(c)
  pack_layer_t(blob, bm.w1g, G.W(0), q, 64, 4, KT1, [&](int c) { int f = l1_feature(c); return f < q ? f : -1; });
  pack_layer_t(blob, bm.w1f, F.W(0), z0 + z1 + 1, 64, 4, KT1, [&](int c) {
    int f = l1_feature(c);
    if (f < z0 + z1) return f;
    if (f == q) return z0 + z1;
    return -1;
  });
  pack_layer_t(blob, bm.w1h, H.W(0), z0 + z2, 64, 4, KT1, [&](int c) {
    int f = l1_feature(c);
    if (f < z0) return f;
    if (f >= z0 + z1 && f < z0 + z1 + z2) return z0 + (f - z0 - z1);
    return -1;
  });
  for (int l = 0; l < h->meta.n_gh; ++l) pack_layer_t(blob, bm.wg + l * 4096, G.W(1 + l), 64, 64, 4, 4, ident);
  const int LG = (int)G.dims.size() - 2;
  {
    std::vector<float> Wp, bp;
    bgm_g_last_padded(G.W(LG), G.b(LG), p, NTL, Wp, bp);
    pack_layer_t(blob, bm.wgl, Wp.data(), 64, 16 * NTL, NTL, 4, ident);
  }
  pack_layer_t(blob, bm.wf2, F.W(1), 64, 32, 2, 4, ident);
  pack_layer_t(blob, bm.wf3, F.W(2), 32, 8, 1, 2, ident);
  pack_layer_t(blob, bm.wf4, F.W(3), 8, 2, 1, 1, ident);
  pack_layer_t(blob, bm.wh2, H.W(1), 64, 32, 2, 4, ident);
  pack_layer_t(blob, bm.wh3, H.W(2), 32, 8, 1, 2, ident);
  pack_layer_t(blob, bm.wh4, H.W(3), 8, 2, 1, 1, ident);
}

static void fit_free(bgm_handle *h) {
  for (void *p : {(void *)h->theta_dev, (void *)h->m1_dev, (void *)h->m2_dev, (void *)h->bblob_dev, (void *)h->ws_dev,
                  (void *)h->partial_dev, (void *)h->tables_dev, (void *)h->pos_dev, (void *)h->tlast_dev})
    if (p) hipFree(p);
  h->tlast_dev = nullptr;
  h->z_synced = -1;
  if (h->epoch_grad) { hipFree(h->epoch_grad); h->epoch_grad = nullptr; h->epoch_grad_n = 0; }
  h->theta_dev = h->m1_dev = h->m2_dev = h->bblob_dev = h->ws_dev = h->partial_dev = nullptr;
  h->tables_dev = h->pos_dev = nullptr;
  h->fit_active = false;
  fit_chain_free(h);
  gx_fit_end(h);
}

static HostNet iota_net(const HostNet &n, int base) {
  HostNet r;
  r.dims = n.dims;
  r.theta.resize(n.count());
  for (size_t i = 0; i < r.theta.size(); ++i) r.theta[i] = (float)(base + (int)i + 1);  // exact below 2^24
  r.set = true;
  return r;
}

extern "C" int bgm_causal_fit_n_params(bgm_handle *h, int64_t *n_params) {
  if (!h || !h->configured || !n_params) { bgm_set_error("bgm_causal_fit_n_params: bad argument"); return BGM_E_INVALID; }
  *n_params = (int64_t)(h->nets[BGM_NET_G].count() + h->nets[BGM_NET_F].count() + h->nets[BGM_NET_H].count());
  return BGM_OK;
}

extern "C" int bgm_causal_fit_begin(bgm_handle *h, int64_t n_rows, int32_t max_batch, void *stream_) {
  if (!h || !h->configured) { bgm_set_error("bgm_causal_fit_begin: handle not configured"); return BGM_E_STATE; }
  if (n_rows <= 0 || max_batch <= 0) { bgm_set_error("bgm_causal_fit_begin: n_rows / max_batch must be positive"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  // forward blob + meta from the host weights.  A model that no LDS-resident shape holds (sum(z_dims) > 19, ...) is fitted by the
  // row-tile chains alone: they read the canonical parameters in place (fit_chain.h) and need neither blob.
  const bool gx = gx_wanted(h);     // hidden widths / depths outside the compiled families: the general-width engine (gx_api.hip)
  int rc = (gx || bnf_det_wanted(h)) ? BGM_E_UNSUPPORTED : bgm_causal_build_blob(h, stream);
  const bool chain_only = rc == BGM_E_UNSUPPORTED;
  if (rc && !chain_only) return rc;
  fit_free(h);
  const HostNet &G = h->nets[BGM_NET_G], &F = h->nets[BGM_NET_F], &H = h->nets[BGM_NET_H];
  const int ng = (int)G.count(), nf = (int)F.count(), nh = (int)H.count();
  const int np = ng + nf + nh;
  if (np >= (1 << 24)) { bgm_set_error("bgm_causal_fit_begin: too many parameters"); return BGM_E_UNSUPPORTED; }
  h->n_params = np;
  const int KT1 = h->KT1, NTL = h->NTL, n_gh = h->meta.n_gh, q = h->q;
  // ---- canonical theta on the device
  std::vector<float> theta(np);
  std::copy(G.theta.begin(), G.theta.end(), theta.begin());
  std::copy(F.theta.begin(), F.theta.end(), theta.begin() + ng);
  std::copy(H.theta.begin(), H.theta.end(), theta.begin() + ng + nf);
  BGM_HIP_CHECK(hipMalloc(&h->theta_dev, sizeof(float) * np));
  BGM_HIP_CHECK(hipMalloc(&h->m1_dev, sizeof(float) * np));
  BGM_HIP_CHECK(hipMalloc(&h->m2_dev, sizeof(float) * np));
  BGM_HIP_CHECK(hipMemcpy(h->theta_dev, theta.data(), sizeof(float) * np, hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMemset(h->m1_dev, 0, sizeof(float) * np));
  BGM_HIP_CHECK(hipMemset(h->m2_dev, 0, sizeof(float) * np));
  h->t_theta = 0; h->t_z = 0;
  if (gx) {
    rc = gx_fit_begin(h, n_rows, max_batch, stream);
    if (rc) { fit_free(h); return rc; }
    h->fit_active = true;
    return BGM_OK;
  }
  if (chain_only) {
    // default hidden widths, but no LDS-resident blob holds the model: the row-tile chains where they apply (minibatches of at most 32
    // rows, v_dim <= 207), the general-width engine otherwise (any minibatch size, any v_dim)
    if (max_batch <= 32) {
      auto s2 = [](float s) { return s > 0.0f ? s * s : -1.0f; };
      h->meta.sig2_v = s2(h->cfg.sigma_v); h->meta.sig2_x = s2(h->cfg.sigma_x); h->meta.sig2_y = s2(h->cfg.sigma_y);
      std::vector<int> tables(4 * (size_t)np, -1);          // no blob positions to refresh after an Adam step
      BGM_HIP_CHECK(hipMalloc(&h->tables_dev, sizeof(int) * tables.size()));
      BGM_HIP_CHECK(hipMemcpy(h->tables_dev, tables.data(), sizeof(int) * tables.size(), hipMemcpyHostToDevice));
      std::memset(&h->fit_ws, 0, sizeof(h->fit_ws));
      h->fit_ws.B = 32; h->fit_ws.dz = 0; h->fit_ws.total = 32 * (long long)h->q;
      BGM_HIP_CHECK(hipMalloc(&h->ws_dev, sizeof(float) * h->fit_ws.total));
      BGM_HIP_CHECK(hipMemset(h->ws_dev, 0, sizeof(float) * h->fit_ws.total));
      h->fit_bcap = 32;
      h->fit_rows = n_rows;
      BGM_HIP_CHECK(hipMalloc(&h->pos_dev, sizeof(int) * 2 * n_rows));
      BGM_HIP_CHECK(hipMemset(h->pos_dev, 0xFF, sizeof(int) * 2 * n_rows));
      BGM_HIP_CHECK(hipDeviceSynchronize());
      h->fit_active = true;
      rc = fit_chain_setup(h, theta);
      if (rc) { fit_free(h); return rc; }
      if (h->fit_chain) return BGM_OK;
      for (void *p : {(void *)h->tables_dev, (void *)h->ws_dev, (void *)h->pos_dev}) if (p) hipFree(p);
      h->tables_dev = nullptr; h->ws_dev = nullptr; h->pos_dev = nullptr; h->fit_active = false;
    }
    rc = gx_fit_begin(h, n_rows, max_batch, stream);
    if (rc) { fit_free(h); return rc; }
    h->fit_active = true;
    return BGM_OK;
  }
  // ---- transposed blob
  fit_layout_backward(h, h->fit_meta);
  if ((size_t)h->fit_meta.total * 4 > 160 * 1024) { bgm_set_error("transposed weights do not fit in LDS"); return BGM_E_UNSUPPORTED; }
  std::vector<float> bblob;
  causal_pack_backward(h, G, F, H, h->fit_meta, bblob);
  BGM_HIP_CHECK(hipMalloc(&h->bblob_dev, sizeof(float) * bblob.size()));
  BGM_HIP_CHECK(hipMemcpy(h->bblob_dev, bblob.data(), sizeof(float) * bblob.size(), hipMemcpyHostToDevice));
  // ---- parameter -> blob position tables, derived by packing iota nets through the same packers
  HostNet Gi = iota_net(G, 0), Fi = iota_net(F, ng), Hi = iota_net(H, ng + nf);
  std::vector<float> fidx, bidx;
  CausalMeta keep = h->meta;
  rc = causal_pack_forward(h, Gi, Fi, Hi, fidx);
  h->meta = keep;
  if (rc) return rc;
  causal_pack_backward(h, Gi, Fi, Hi, h->fit_meta, bidx);
  std::vector<int> tables(4 * (size_t)np, -1);
  int *fwd_dst = tables.data(), *fwd_dst2 = fwd_dst + np, *bwd_dst = fwd_dst2 + np, *grad_src = bwd_dst + np;
  for (size_t d = 0; d < fidx.size(); ++d) {
    const int c = (int)fidx[d] - 1;
    if (c < 0) continue;
    if (fwd_dst[c] < 0) fwd_dst[c] = (int)d; else fwd_dst2[c] = (int)d;
  }
  for (size_t d = 0; d < bidx.size(); ++d) {
    const int c = (int)bidx[d] - 1;
    if (c >= 0) bwd_dst[c] = (int)d;
  }
  // ---- workspace layout for up to max_batch rows
  const int B = (max_batch + 15) / 16 * 16;
  h->fit_bcap = B;
  FitWs &w = h->fit_ws;
  std::memset(&w, 0, sizeof(w));
  w.B = B;
  long long off = 0;
  auto wtake = [&](long long n) { long long o = off; off += (n + 3) / 4 * 4; return o; };
  w.zin = wtake((long long)B * 16 * KT1);
  w.ag = wtake((long long)(n_gh + 1) * B * 64);
  w.outg = wtake((long long)B * 16 * NTL);
  w.af1 = wtake((long long)B * 64); w.af2 = wtake((long long)B * 32); w.af3 = wtake((long long)B * 16); w.outf = wtake((long long)B * 16);
  w.ah1 = wtake((long long)B * 64); w.ah2 = wtake((long long)B * 32); w.ah3 = wtake((long long)B * 16); w.outh = wtake((long long)B * 16);
  w.dg = wtake((long long)(n_gh + 1) * B * 64);
  w.dgl = wtake((long long)B * 16 * NTL);
  w.df1 = wtake((long long)B * 64); w.df2 = wtake((long long)B * 32); w.df3 = wtake((long long)B * 16); w.df4 = wtake((long long)B * 16);
  w.dh1 = wtake((long long)B * 64); w.dh2 = wtake((long long)B * 32); w.dh3 = wtake((long long)B * 16); w.dh4 = wtake((long long)B * 16);
  w.dz = wtake((long long)B * q);
  w.total = off;
  BGM_HIP_CHECK(hipMalloc(&h->ws_dev, sizeof(float) * w.total));
  BGM_HIP_CHECK(hipMemset(h->ws_dev, 0, sizeof(float) * w.total));
  // ---- weight-gradient GEMM layers + gradient source table
  DwArgs &dw = h->dw;
  std::memset(&dw, 0, sizeof(dw));
  int nl = 0, poff = 0;
  auto add = [&](long long a_off, long long d_off, int K, int N) {
    DwLayer &L = dw.layer[nl++];
    L.a_off = a_off; L.d_off = d_off; L.K = K; L.N = N; L.out_off = poff;
    poff += K * N + N;
    return nl - 1;
  };
  const int z0 = h->cfg.z_dims[0], z1 = h->cfg.z_dims[1];
  // maps canonical (layer, in, out) -> partial offset
  int out_slot_from = -1, out_slot_to = -1;   // one output column may live at another padded position (g's variance column)
  auto fill = [&](const HostNet &net, int base, int layer, int li, auto in_map) {
    const DwLayer &L = dw.layer[li];
    const int n_in = net.dims[layer], n_out = net.dims[layer + 1];
    const int wbase = base + (int)(net.W(layer) - net.theta.data());
    auto om = [&](int o) { return o == out_slot_from ? out_slot_to : o; };
    for (int i = 0; i < n_in; ++i)
      for (int o = 0; o < n_out; ++o) grad_src[wbase + i * n_out + o] = L.out_off + in_map(i) * L.N + om(o);
    const int bbase = base + (int)(net.b(layer) - net.theta.data());
    for (int o = 0; o < n_out; ++o) grad_src[bbase + o] = L.out_off + L.K * L.N + om(o);
  };
  auto idm = [](int i) { return i; };
  {
    int li = add(w.zin, w.dg, 16 * KT1, 64);
    fill(G, 0, 0, li, idm);
    for (int l = 1; l <= n_gh; ++l) {
      li = add(w.ag + (long long)(l - 1) * B * 64, w.dg + (long long)l * B * 64, 64, 64);
      fill(G, 0, l, li, idm);
    }
    li = add(w.ag + (long long)n_gh * B * 64, w.dgl, 64, 16 * NTL);
    out_slot_from = h->p; out_slot_to = bgm_sig_slot(h->p, NTL);
    fill(G, 0, n_gh + 1, li, idm);
    out_slot_from = out_slot_to = -1;
    li = add(w.zin, w.df1, 16 * KT1, 64);
    fill(F, ng, 0, li, [&](int i) { return i < z0 + z1 ? i : q; });
    li = add(w.af1, w.df2, 64, 32); fill(F, ng, 1, li, idm);
    li = add(w.af2, w.df3, 32, 16); fill(F, ng, 2, li, idm);
    li = add(w.af3, w.df4, 16, 16); fill(F, ng, 3, li, idm);
    li = add(w.zin, w.dh1, 16 * KT1, 64);
    fill(H, ng + nf, 0, li, [&](int i) { return i < z0 ? i : z0 + z1 + (i - z0); });
    li = add(w.ah1, w.dh2, 64, 32); fill(H, ng + nf, 1, li, idm);
    li = add(w.ah2, w.dh3, 32, 16); fill(H, ng + nf, 2, li, idm);
    li = add(w.ah3, w.dh4, 16, 16); fill(H, ng + nf, 3, li, idm);
  }
  if (nl > BGM_MAX_DW_LAYERS) { bgm_set_error("too many layers"); return BGM_E_UNSUPPORTED; }
  dw.n_layers = nl;
  dw.partial_stride = (poff + 3) / 4 * 4;
  h->rows_per_slice = 256;
  h->n_slices_cap = (B + h->rows_per_slice - 1) / h->rows_per_slice;
  BGM_HIP_CHECK(hipMalloc(&h->partial_dev, sizeof(float) * dw.partial_stride * h->n_slices_cap));
  BGM_HIP_CHECK(hipMalloc(&h->tables_dev, sizeof(int) * tables.size()));
  BGM_HIP_CHECK(hipMemcpy(h->tables_dev, tables.data(), sizeof(int) * tables.size(), hipMemcpyHostToDevice));
  h->fit_rows = n_rows;
  BGM_HIP_CHECK(hipMalloc(&h->pos_dev, sizeof(int) * 2 * n_rows));      // [position in the batch | step stamp]
  BGM_HIP_CHECK(hipMemset(h->pos_dev, 0xFF, sizeof(int) * 2 * n_rows));
  BGM_HIP_CHECK(hipDeviceSynchronize());
  h->fit_active = true;
  return fit_chain_setup(h, theta);
}

#define BGM_FIT_VARIANTS(X) X(1, 3, 13) X(1, 3, 7) X(1, 3, 2) X(2, 1, 10) X(2, 1, 7) X(2, 1, 2)

static int fit_grid(const bgm_handle *h, int B) {
  const int tiles = (B + 15) / 16;
  return std::max(1, std::min((tiles + FIT_WAVES - 1) / FIT_WAVES, h->n_cus));
}

static int launch_fwd_bwd(bgm_handle *h, FitKArgs &ka, hipStream_t stream) {
  const int grid = fit_grid(h, ka.B);
#define X(KT1_, KSL1_, NTL_)                                                                           \
  if (h->KT1 == KT1_ && h->KSL1 == KSL1_ && h->NTL == NTL_) {                                          \
    auto kf = fit_fwd_kernel<KT1_, KSL1_, NTL_, FIT_WAVES>;                                            \
    auto kb = fit_bwd_kernel<KT1_, KSL1_, NTL_, FIT_WAVES>;                                            \
    const int lds_f = h->meta.total * 4, lds_b = h->fit_meta.total * 4;                                \
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, lds_f)); \
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b)); \
    ka.blob = h->blob_dev;                                                                             \
    hipLaunchKernelGGL(kf, dim3(grid), dim3(64 * FIT_WAVES), lds_f, stream, ka);                       \
    BGM_HIP_CHECK(hipGetLastError());                                                                  \
    ka.blob = h->bblob_dev;                                                                            \
    ka.loss = nullptr;                                                                                 \
    hipLaunchKernelGGL(kb, dim3(grid), dim3(64 * FIT_WAVES), lds_b, stream, ka);                       \
    BGM_HIP_CHECK(hipGetLastError());                                                                  \
    return BGM_OK;                                                                                     \
  }
  BGM_FIT_VARIANTS(X)
#undef X
  bgm_set_error("no compiled fit kernel variant for this shape");
  return BGM_E_UNSUPPORTED;
}

static int fit_check(bgm_handle *h, const void *x, const void *y, const void *v, const void *z, int batch, int bg, const char *who) {
  if (!h || !h->fit_active) { bgm_set_error(std::string(who) + ": call bgm_causal_fit_begin first"); return BGM_E_STATE; }
  if (!x || !y || !v || !z) { bgm_set_error(std::string(who) + ": NULL data pointer"); return BGM_E_INVALID; }
  if (batch <= 0 || batch > h->fit_bcap || bg < batch) { bgm_set_error(std::string(who) + ": batch out of range (max_batch of fit_begin)"); return BGM_E_INVALID; }
  return BGM_OK;
}

extern "C" int bgm_causal_fit_theta_grad(bgm_handle *h, const float *x, const float *y, const float *v,
                                         const float *data_z, const int32_t *idx, int64_t row_lo, int32_t batch,
                                         int32_t batch_global, float *grad, double *loss, void *stream_) {
  int rc = fit_check(h, x, y, v, data_z, batch, batch_global, "bgm_causal_fit_theta_grad");
  if (rc) return rc;
  if (!grad) { bgm_set_error("bgm_causal_fit_theta_grad: grad_dev is NULL"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (FitChainState *fc = static_cast<FitChainState *>(h->fit_chain); fc && batch <= 32) {
    FitChainArgs ca = fc->base;
    ca.x = x; ca.y = y; ca.v = v; ca.data_z = data_z; ca.idx = idx; ca.row_lo = row_lo; ca.inv_B = 1.0f / (float)batch_global;
    ca.loss = loss; ca.grad = grad;
    fit_chain_launch(fc, ca, batch, 0, stream);
    BGM_HIP_CHECK(hipGetLastError());
    return BGM_OK;
  }
  FitKArgs ka{};
  ka.m = h->meta; ka.bm = h->fit_meta; ka.ws = h->fit_ws; ka.wsp = h->ws_dev;
  ka.x = x; ka.y = y; ka.v = v; ka.data_z = data_z; ka.idx = idx; ka.row_lo = row_lo; ka.B = batch;
  ka.inv_B = 1.0f / (float)batch_global; ka.z_mode = 0; ka.loss = loss;
  rc = gx_fit_active(h) ? gx_fit_grads(h, x, y, v, data_z, idx, row_lo, batch, batch_global, 0, grad, loss, stream) : launch_fwd_bwd(h, ka, stream);
  if (rc) return rc;
  DwArgs dw = h->dw;
  dw.ws = h->ws_dev; dw.partial = h->partial_dev; dw.B = batch; dw.rows_per_slice = h->rows_per_slice;
  const int n_slices = (batch + h->rows_per_slice - 1) / h->rows_per_slice;
  hipLaunchKernelGGL(fit_dw_kernel, dim3(n_slices, dw.n_layers, fit_dw_chunks(dw)), dim3(256), 0, stream, dw);
  BGM_HIP_CHECK(hipGetLastError());
  const int np = h->n_params;
  hipLaunchKernelGGL(fit_grad_reduce_kernel, dim3((np + 255) / 256), dim3(256), 0, stream, h->partial_dev,
                     dw.partial_stride, n_slices, h->tables_dev + 3 * (size_t)np, np, grad);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_causal_fit_theta_apply(bgm_handle *h, const float *grad, float lr_theta, void *stream_) {
  if (!h || !h->fit_active) { bgm_set_error("bgm_causal_fit_theta_apply: call bgm_causal_fit_begin first"); return BGM_E_STATE; }
  if (!grad) { bgm_set_error("bgm_causal_fit_theta_apply: grad_dev is NULL"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  h->t_theta += 1;
  const double t = (double)h->t_theta;
  const float lr_t = (float)((double)lr_theta * std::sqrt(1.0 - std::pow((double)ADAM_B2, t)) / (1.0 - std::pow((double)ADAM_B1, t)));
  const int np = h->n_params;
  const int *tb = h->tables_dev;
  hipLaunchKernelGGL(fit_adam_theta_kernel, dim3((np + 255) / 256), dim3(256), 0, (hipStream_t)stream_, h->theta_dev,
                     h->m1_dev, h->m2_dev, grad, np, lr_t, ADAM_B1, ADAM_B2, ADAM_EPS, gx_fit_active(h) ? gx_pack(h) : h->blob_dev,
                     gx_fit_active(h) ? gx_packT(h) : h->bblob_dev, tb,
                     tb + np, tb + 2 * (size_t)np, h->fit_chain ? static_cast<FitChainState *>(h->fit_chain)->thetaT : nullptr,
                     h->fit_chain ? static_cast<FitChainState *>(h->fit_chain)->mirror_dst : nullptr);
  BGM_HIP_CHECK(hipGetLastError());
  h->sblob_valid = false;   // the sampling copy (evaluate between epochs, predict after the fit) follows the new parameters
  h->det_valid = false;
  return BGM_OK;
}

extern "C" int bgm_causal_fit_z_step(bgm_handle *h, const float *x, const float *y, const float *v, float *data_z,
                                     float *zm, float *zv, const int32_t *idx, int64_t row_lo, int32_t batch,
                                     int32_t batch_global, float lr_z, int32_t lazy, double *loss, void *stream_) {
  int rc = fit_check(h, x, y, v, data_z, batch, batch_global, "bgm_causal_fit_z_step");
  if (rc) return rc;
  if (!zm || !zv) { bgm_set_error("bgm_causal_fit_z_step: NULL Adam slots"); return BGM_E_INVALID; }
  if (!idx) { bgm_set_error("bgm_causal_fit_z_step: idx_dev is required"); return BGM_E_INVALID; }
  if (lazy < 0 || lazy > 2) { bgm_set_error("bgm_causal_fit_z_step: lazy must be 0 (dense), 1 (batch rows only) or 2 (replay)"); return BGM_E_INVALID; }
  if (lazy == 2 && (!h->tlast_dev || h->z_synced != h->t_z + 1)) {
    bgm_set_error("bgm_causal_fit_z_step: lazy = 2 needs bgm_causal_fit_z_sync on this minibatch's rows first (before its theta phase)");
    return BGM_E_STATE;
  }
  if (lazy != 2 && h->tlast_dev && h->z_synced != -2) {
    bgm_set_error("bgm_causal_fit_z_step: rows have pending replay steps; flush with bgm_causal_fit_z_sync(idx = NULL) before changing mode");
    return BGM_E_STATE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  FitKArgs ka{};
  ka.m = h->meta; ka.bm = h->fit_meta; ka.ws = h->fit_ws; ka.wsp = h->ws_dev;
  ka.x = x; ka.y = y; ka.v = v; ka.data_z = data_z; ka.idx = idx; ka.row_lo = row_lo; ka.B = batch;
  ka.inv_B = 1.0f / (float)batch_global; ka.z_mode = 1; ka.loss = loss;
  bool pos_set = false, adam_fused = false;
  if (FitChainState *fc = static_cast<FitChainState *>(h->fit_chain); fc && batch <= 32) {
    FitChainArgs ca = fc->base;
    ca.x = x; ca.y = y; ca.v = v; ca.data_z = data_z; ca.idx = idx; ca.row_lo = row_lo; ca.inv_B = ka.inv_B;
    ca.loss = loss; ca.dz = h->ws_dev + h->fit_ws.dz; ca.ws = fc->ws_z;
    if (!lazy) { ca.pos = h->pos_dev; ca.pos_n = h->fit_rows; ca.epoch = (int)((h->t_z + 1) & 0x3FFFFFFF); pos_set = true; }
    else {                      // the batch rows' Adam step rides on the chain kernel's epilogue
      const double t1 = (double)(h->t_z + 1);
      ca.zm = zm; ca.zv = zv; ca.z_out = data_z; ca.t_last = lazy == 2 ? h->tlast_dev : nullptr; ca.t_now = (int)(h->t_z + 1);
      ca.lr_t = (float)((double)lr_z * std::sqrt(1.0 - std::pow((double)ADAM_B2, t1)) / (1.0 - std::pow((double)ADAM_B1, t1)));
      ca.b1 = ADAM_B1; ca.b2 = ADAM_B2; ca.eps = ADAM_EPS;
      adam_fused = true;
    }
    fit_chain_launch(fc, ca, batch, 1, stream);
    rc = BGM_OK;
  } else if (gx_fit_active(h)) rc = gx_fit_grads(h, x, y, v, data_z, idx, row_lo, batch, batch_global, 1, nullptr, loss, stream);
  else rc = launch_fwd_bwd(h, ka, stream);
  if (rc) return rc;
  h->t_z += 1;
  if (adam_fused) { BGM_HIP_CHECK(hipGetLastError()); return BGM_OK; }
  const double t = (double)h->t_z;
  const float lr_t = (float)((double)lr_z * std::sqrt(1.0 - std::pow((double)ADAM_B2, t)) / (1.0 - std::pow((double)ADAM_B1, t)));
  const int q = h->q;
  const float *dz = h->ws_dev + h->fit_ws.dz;
  if (lazy == 2) {
    const long long n = (long long)batch * q;
    hipLaunchKernelGGL(fit_adam_z_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, data_z, zm, zv, dz,
                       h->tlast_dev, h->fit_rows, q, lr_t, ADAM_B1, ADAM_B2, ADAM_EPS, 2, idx, batch, (int)h->t_z);
  } else if (lazy) {
    const long long n = (long long)batch * q;
    hipLaunchKernelGGL(fit_adam_z_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, data_z, zm, zv, dz,
                       h->pos_dev, h->fit_rows, q, lr_t, ADAM_B1, ADAM_B2, ADAM_EPS, 1, idx, batch, 0);
  } else {
    const int epoch = (int)(h->t_z & 0x3FFFFFFF);      // stamps of earlier steps never match (2^30 steps before a wrap)
    if (!pos_set)      // (the chain kernel of the latent phase has written the map already)
      hipLaunchKernelGGL(fit_set_pos_kernel, dim3((batch + 255) / 256), dim3(256), 0, stream, h->pos_dev, h->fit_rows, idx, batch, epoch);
    const long long n = h->fit_rows * q;
    hipLaunchKernelGGL(fit_adam_z_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, data_z, zm, zv, dz,
                       h->pos_dev, h->fit_rows, q, lr_t, ADAM_B1, ADAM_B2, ADAM_EPS, 0, idx, batch, epoch);
  }
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

// the replay launches: rows idx[0 .. n_sel) (NULL: all rows) of (z, zm, zv) brought from their t_last to step t_to
static void fit_mark_rows(bgm_handle *h, const int32_t *idx, int n_sel, long long t_to, hipStream_t stream) {      // replayed rows are current to t_to
  hipLaunchKernelGGL(fit_mark_rows_kernel, dim3((unsigned)((n_sel + 255) / 256)), dim3(256), 0, stream, h->tlast_dev, idx, (long long)n_sel, (int)t_to);
}
// mark = false: the caller guarantees the latent step of these rows follows (its Adam epilogue stamps them with t_to + 1)
static int fit_z_sync_rows(bgm_handle *h, float *data_z, float *zm, float *zv, const int32_t *idx, long long n_sel, int t_to, float lr_z,
                           hipStream_t stream, bool mark = true) {
  const long long threads = n_sel * h->q * 16;
  hipLaunchKernelGGL(fit_adam_z_replay_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, data_z, zm, zv,
                     h->tlast_dev, h->q, idx, n_sel, t_to, lr_z, ADAM_B1, ADAM_B2, ADAM_EPS);
  if (!mark) { BGM_HIP_CHECK(hipGetLastError()); return BGM_OK; }
  if (!idx) hipLaunchKernelGGL(fit_fill_int_kernel, dim3((unsigned)((n_sel + 255) / 256)), dim3(256), 0, stream, h->tlast_dev, n_sel, t_to);
  else hipLaunchKernelGGL(fit_mark_rows_kernel, dim3((unsigned)((n_sel + 255) / 256)), dim3(256), 0, stream, h->tlast_dev, idx, n_sel, t_to);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

static int fit_z_sync_impl(bgm_handle *h, float *data_z, float *zm, float *zv, const int32_t *idx, int32_t batch, float lr_z, void *stream_,
                           bool mark);
extern "C" int bgm_causal_fit_z_sync(bgm_handle *h, float *data_z, float *zm, float *zv, const int32_t *idx, int32_t batch, float lr_z,
                                     void *stream_) {
  return fit_z_sync_impl(h, data_z, zm, zv, idx, batch, lr_z, stream_, true);
}
static int fit_z_sync_impl(bgm_handle *h, float *data_z, float *zm, float *zv, const int32_t *idx, int32_t batch, float lr_z, void *stream_,
                           bool mark) {
  if (!h || !h->fit_active) { bgm_set_error("bgm_causal_fit_z_sync: call bgm_causal_fit_begin first"); return BGM_E_STATE; }
  if (!data_z || !zm || !zv || (idx && batch < 1)) { bgm_set_error("bgm_causal_fit_z_sync: bad argument"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  const long long n_rows = h->fit_rows;
  if (!h->tlast_dev) {                               // every row is current at the step the mode is entered
    BGM_HIP_CHECK(hipMalloc(&h->tlast_dev, sizeof(int) * n_rows));
    hipLaunchKernelGGL(fit_fill_int_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream, h->tlast_dev, n_rows, (int)h->t_z);
  }
  int rc = fit_z_sync_rows(h, data_z, zm, zv, idx, idx ? batch : n_rows, (int)h->t_z, lr_z, stream, mark || !idx);
  if (rc) return rc;
  h->z_synced = idx ? h->t_z + 1 : -2;               // -2: flushed, any mode may follow
  return BGM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// One pass over a list of minibatches with the loop inside the library (single process: no all-reduce between the phases).
// replaces: the loop body causalbgm/base.py:490-505 for every minibatch of an epoch -- the same updates, in the same order, as the
// host loop bgm_causal_fit_z_sync / _theta_grad / _theta_apply / _z_step; results are those of the sequential order bit for bit.
// With the latent optimizer in batch-rows or replay mode (lazy = 1 / 2) and the row-tile chains:
//   * the latent phase of minibatch k runs on a second stream beside the theta phase of minibatch k + 1: the two touch disjoint rows of
//     the latent table (the minibatches of one call must be disjoint -- a permutation); the parameters live in a ring of D buffers
//     (step k's Adam reads one and writes the next), so the update k + 1 never waits for the latent phase k;
//   * the Adam step rides on the gradient-tile kernel (one expression, fit_adam_theta_one), the replay of a minibatch's pending latent
//     steps rides D minibatches ahead on the latent phase's kernel (rider workgroups), no separate row-mark launch: three launches
//     per minibatch (theta chains, gradient tiles + Adam, latent chains + Adam-on-Z + replay);
//   * the two streams are ordered by counters in device memory the kernels themselves wait on and advance (FitSync), not by HIP
//     events: a record / wait pair costs the stream it sits in 4-6 us of command-processor time per minibatch.
// N = 1e6, B = 32, replayed latent Adam, per minibatch: 69.5 us in one stream; 51.5 with two streams, two buffers and events (round-4
// start); 50.0 with the fused Adam; 46.9 with the replay ahead; 35.9 with device-side ordering = the parameter stream's own chain
// (theta chains 26.5 + gradient tiles / Adam 8).  Measured and dropped: a single parameter buffer (56.5 us); ring depth 2 / 3 / 4 under
// events (46.8 / 45.6 / 45.8 us: the events, not the dependency distance, were the cost); a release fence by every wave of the
// gradient-tile kernel before its count (39.2 us: 272 L2 write-back sweeps; one per workgroup is enough).
// ---------------------------------------------------------------------------------------------------------------------------
// a device-side wait of the previous bgm_causal_fit_epoch call gave up (FitSync): that call's results are void
static int fit_epoch_check(bgm_handle *h, hipStream_t stream) {
  if (!h->epoch_ctr) return BGM_OK;
  unsigned c[4];
  BGM_HIP_CHECK(hipMemcpyAsync(c, h->epoch_ctr, sizeof(c), hipMemcpyDeviceToHost, stream));
  BGM_HIP_CHECK(hipStreamSynchronize(stream));
  if (!c[2]) return BGM_OK;
  BGM_HIP_CHECK(hipMemset(h->epoch_ctr, 0, sizeof(c)));
  h->epoch_theta_done = h->epoch_z_done = 0;
  bgm_set_error("bgm_causal_fit_epoch: a device-side ordering wait timed out in the previous call (its updates are unordered); "
                "BGM_FIT_NO_FLAGS=1 orders the two streams with HIP events instead");
  return BGM_E_HIP;
}

// comm != NULL (bgm_causal_fit_epoch_dp): this rank's share of a data-parallel epoch -- every rank calls with its own rows and the
// same n_use / batch; a minibatch is `batch` local rows of a GLOBAL minibatch of batch * world rows (gradients scaled 1 / (b * world)),
// and the fused g|f|h gradient is summed over the ranks (ncclAllReduce on the parameter stream) between the gradient tiles and the Adam
// step, so all ranks take identical steps.  The Adam step is then its own launch (it cannot ride on the tiles: the sum sits between),
// the streams are ordered by HIP events, the latent phase of minibatch k still runs beside the theta phase of k + 1.
static int fit_epoch_impl(bgm_handle *h, const float *x, const float *y, const float *v, float *data_z, float *zm, float *zv,
                          const int32_t *perm, int64_t n_use, int32_t batch, float lr_theta, float lr_z, int32_t lazy,
                          double *loss, double *loss_z, void *comm, int world, void *stream_);

extern "C" int bgm_causal_fit_epoch(bgm_handle *h, const float *x, const float *y, const float *v, float *data_z, float *zm, float *zv,
                                    const int32_t *perm, int64_t n_use, int32_t batch, float lr_theta, float lr_z, int32_t lazy,
                                    double *loss, double *loss_z, void *stream_) {
  return fit_epoch_impl(h, x, y, v, data_z, zm, zv, perm, n_use, batch, lr_theta, lr_z, lazy, loss, loss_z, nullptr, 1, stream_);
}

extern "C" int bgm_causal_fit_epoch_dp(bgm_handle *h, const float *x, const float *y, const float *v, float *data_z, float *zm, float *zv,
                                       const int32_t *perm, int64_t n_use, int32_t batch, float lr_theta, float lr_z, int32_t lazy,
                                       double *loss, double *loss_z, void *comm, void *stream_) {
  if (!comm) { bgm_set_error("bgm_causal_fit_epoch_dp: NULL communicator (bgm_comm_create)"); return BGM_E_INVALID; }
  int world = 0, rank = 0;
  int rc = bgm_comm_world(comm, &world, &rank);
  if (rc) return rc;
  return fit_epoch_impl(h, x, y, v, data_z, zm, zv, perm, n_use, batch, lr_theta, lr_z, lazy, loss, loss_z, comm, world, stream_);
}

static int fit_epoch_impl(bgm_handle *h, const float *x, const float *y, const float *v, float *data_z, float *zm, float *zv,
                          const int32_t *perm, int64_t n_use, int32_t batch, float lr_theta, float lr_z, int32_t lazy,
                          double *loss, double *loss_z, void *comm, int world, void *stream_) {
  if (!h || !h->fit_active) { bgm_set_error("bgm_causal_fit_epoch: call bgm_causal_fit_begin first"); return BGM_E_STATE; }
  if (!perm || n_use < 1 || batch < 1) { bgm_set_error("bgm_causal_fit_epoch: bad minibatch list"); return BGM_E_INVALID; }
  hipStream_t sA = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (!h->epoch_grad || h->epoch_grad_n < h->n_params) {
    if (h->epoch_grad) BGM_HIP_CHECK(hipFree(h->epoch_grad));
    BGM_HIP_CHECK(hipMalloc(&h->epoch_grad, sizeof(float) * h->n_params));
    h->epoch_grad_n = h->n_params;
  }
  static const bool no_overlap = std::getenv("BGM_FIT_NO_OVERLAP") != nullptr;       // dev A/B
  const bool overlap = !no_overlap && lazy != 0 && h->fit_chain != nullptr && batch <= 32;
  if (overlap && !h->epoch_stream) {
    BGM_HIP_CHECK(hipStreamCreateWithFlags(&h->epoch_stream, hipStreamNonBlocking));
    for (int k = 0; k < 4; ++k) {
      BGM_HIP_CHECK(hipEventCreateWithFlags(&h->epoch_ev_t[k], hipEventDisableTiming));
      BGM_HIP_CHECK(hipEventCreateWithFlags(&h->epoch_ev_z[k], hipEventDisableTiming));
    }
  }
  hipStream_t sB = overlap ? h->epoch_stream : sA;
  int rc = BGM_OK;
  long long k = 0;
  // D parameter buffers in a ring: step k's Adam reads buffer k % D and writes the next one, so the latent phase of minibatch k (reading
  // the new buffer) holds up nothing before the parameter update of minibatch k + D - 1, which overwrites the buffer the latent phase
  // k - 1 read.  With D = 2 the cycle  latent phase k -> (event) -> theta phase k + 2 -> (event) -> latent phase k + 2  carries two
  // cross-stream hops of ~10 us per two minibatches (46.3 us per minibatch measured = 36 + 10); D = 4 spreads them over four.
  static const int depth_env = std::getenv("BGM_FIT_EPOCH_DEPTH") ? std::atoi(std::getenv("BGM_FIT_EPOCH_DEPTH")) : 4;
  const int D = overlap ? std::min(4, std::max(2, depth_env)) : 1;
  FitChainState *fc = static_cast<FitChainState *>(h->fit_chain);
  const int np = h->n_params;
  float *tb[4] = {h->theta_dev, nullptr, nullptr, nullptr}, *tTb[4] = {overlap ? fc->thetaT : nullptr, nullptr, nullptr, nullptr};
  for (int d = 1; d < D; ++d) {
    if (!fc->theta_x[d - 1]) {
      BGM_HIP_CHECK(hipMalloc((void **)&fc->theta_x[d - 1], sizeof(float) * np));
      BGM_HIP_CHECK(hipMalloc((void **)&fc->thetaT_x[d - 1], sizeof(float) * ((size_t)np + 64)));
    }
    tb[d] = fc->theta_x[d - 1]; tTb[d] = fc->thetaT_x[d - 1];
    BGM_HIP_CHECK(hipMemcpyAsync(tTb[d], fc->thetaT, sizeof(float) * ((size_t)np + 64), hipMemcpyDeviceToDevice, sA));   // (rows no parameter maps to)
  }
  int cur = 0;
  static const bool no_fuse = std::getenv("BGM_FIT_NO_FUSED_ADAM") != nullptr;       // dev A/B: separate Adam launch + explicit row marks
  // chained: the machinery of the row-tile chains (replay ahead, device-side ordering, rows stamped by the latent step) is on;
  // fuse: the Adam step rides on the gradient-tile kernel -- not under data parallelism, where the all-reduce sits between the tiles
  // and Adam: there the step is its own launch that waits / counts like the tiles would (fit_adam_theta_sync_kernel)
  const bool chained = overlap && !no_fuse, fuse = chained && !comm;
  // Replay mode: the pending zero-gradient steps of minibatch j's rows are replayed D minibatches ahead, on the second stream behind
  // the latent phase j - D -- off the parameter stream's critical path (replay + chains + gradient tiles), and covered by the event the
  // theta phase j waits for anyway.  Legal because the minibatches of one call are disjoint: nothing touches those rows in between,
  // and the step a replay runs to (the latent step count when minibatch j starts) is known in advance.
  static const bool no_ahead = std::getenv("BGM_FIT_NO_REPLAY_AHEAD") != nullptr;    // dev A/B
  const bool ahead = chained && lazy == 2 && !no_ahead;
  const long long tz0 = h->t_z, n_mb = (n_use + batch - 1) / batch;
  // Ordering between the two streams without events (each record / wait pair costs the stream it sits in 4-6 us of command-processor
  // time: 45.0 us per minibatch with them, 34.7 with the dependencies dropped -- unsafe, measured for the bound): the gradient-tile
  // kernel counts its workgroups into a device counter the latent phase's kernel spins on at entry, and the latent kernel's workgroups
  // (chains + replay riders) count into a second one the theta phase D minibatches later waits for (fit_types.h FitSync).
  static const bool no_flags = std::getenv("BGM_FIT_NO_FLAGS") != nullptr;           // dev A/B: HIP events
  bool flags = chained && !no_flags && (lazy != 2 || ahead);
  unsigned zt[4] = {0, 0, 0, 0};   // the latent counter's value once minibatch (slot)'s kernel is done
  if (flags) {
    if (!h->epoch_ctr) {
      BGM_HIP_CHECK(hipMalloc((void **)&h->epoch_ctr, sizeof(unsigned) * 8));
      BGM_HIP_CHECK(hipMemsetAsync(h->epoch_ctr, 0, sizeof(unsigned) * 8, sA));
      h->epoch_theta_done = h->epoch_z_done = 0;
    } else if ((rc = fit_epoch_check(h, sA))) return rc;
    if (!h->epoch_flags_ok || h->epoch_probe_stream != (void *)sA) {      // once per handle and caller stream: do the two streams run side by side (fit_sync.h)?
      int ok = 0;
      BGM_HIP_CHECK(fit_sync_probe(sA, sB, h->epoch_ctr + 4, &ok));
      h->epoch_flags_ok = ok ? 1 : -1;
      h->epoch_probe_stream = (void *)sA;
    }
    if (h->epoch_flags_ok < 0) flags = false;      // (a profiler serialising kernels, one hardware queue): HIP events
  }
  long long replayed = 0;          // minibatches [k, replayed) have been replayed but not stepped
  auto replay = [&](long long j, hipStream_t st) -> int {
    if (j >= n_mb) return BGM_OK;
    const int bj = (int)std::min<int64_t>(batch, n_use - j * batch);
    replayed = j + 1;
    return fit_z_sync_rows(h, data_z, zm, zv, perm + j * batch, bj, (int)(tz0 + j), lr_z, st, false);
  };
  auto mark_pending = [&](hipStream_t st) {      // a failure: the rows replayed ahead count as current to the step they were brought to
    if (lazy != 2 || !chained) return;
    for (long long j = k; j < std::max(replayed, k + 1) && j < n_mb; ++j)
      fit_mark_rows(h, perm + j * batch, (int)std::min<int64_t>(batch, n_use - j * batch), tz0 + j, st);
  };
  for (int64_t i = 0; i < n_use; i += batch, ++k) {
    const int32_t *idx = perm + i;
    const int b = (int)std::min<int64_t>(batch, n_use - i);
    // (the replayed rows are stamped by the latent step below; only a failure in between needs the explicit mark)
    if (lazy == 2 && (!ahead || k == 0) && (rc = fit_z_sync_impl(h, data_z, zm, zv, idx, b, lr_z, sA, !chained))) break;
    if (ahead && k == 0) {
      for (int d = 1; d < D && !rc; ++d) rc = replay(d, sA);
      if (rc) break;
    }
    if (overlap && k == 0) {       // the second stream starts behind everything queued on the caller's so far (incl. the first replay)
      BGM_HIP_CHECK(hipEventRecord(h->epoch_ev_t[1], sA));
      BGM_HIP_CHECK(hipStreamWaitEvent(sB, h->epoch_ev_t[1], 0));
    }
    if (overlap) { fc->theta_use = tb[cur]; fc->thetaT_use = tTb[cur]; }
    const int *tbl = h->tables_dev;
    const double t = (double)(h->t_theta + 1);
    const float lr_t = (float)((double)lr_theta * std::sqrt(1.0 - std::pow((double)ADAM_B2, t)) / (1.0 - std::pow((double)ADAM_B1, t)));
    if (fuse) {                    // the Adam step rides on the gradient-tile kernel: it writes the other buffer, which the latent
      if (flags) {               // ... phase k - D (same slot) has read: waited for in the chain kernel itself (and the rows' replay with it)
        fc->sync_t = FitSync{k >= D ? h->epoch_ctr + 1 : nullptr, zt[k % D], h->epoch_ctr, (int *)(h->epoch_ctr + 2)};
      } else if (k >= D) BGM_HIP_CHECK(hipStreamWaitEvent(sA, h->epoch_ev_z[k % D], 0));
      FitAdamTheta &ad = fc->fused;
      ad.on = 1; ad.lr_t = lr_t; ad.b1 = ADAM_B1; ad.b2 = ADAM_B2; ad.eps = ADAM_EPS;
      ad.m1 = h->m1_dev; ad.m2 = h->m2_dev; ad.theta_out = tb[(cur + 1) % D];
      ad.fwd_blob = h->blob_dev; ad.bwd_blob = h->bblob_dev; ad.mirror = tTb[(cur + 1) % D];
      ad.fwd_dst = tbl; ad.fwd_dst2 = tbl + np; ad.bwd_dst = tbl + 2 * (size_t)np; ad.mirror_dst = fc->mirror_dst;
    } else if (chained && flags) {   // (data parallel) the chains wait as above -- their rows were replayed by the riders of the latent
      // phase k - D, whose buffer the Adam launch below overwrites -- but nothing of this launch counts the step done: Adam does
      fc->sync_t = FitSync{k >= D ? h->epoch_ctr + 1 : nullptr, zt[k % D], nullptr, (int *)(h->epoch_ctr + 2)};
    } else if (chained && k >= D) BGM_HIP_CHECK(hipStreamWaitEvent(sA, h->epoch_ev_z[k % D], 0));      // (the same under HIP events)
    rc = bgm_causal_fit_theta_grad(h, x, y, v, data_z, idx, 0, b, b * world, h->epoch_grad, loss, sA);
    if (fuse) fc->fused.on = 0;
    if (chained) fc->sync_t = FitSync{};
    if (!rc && comm) rc = bgm_comm_enqueue_all_reduce(comm, h->epoch_grad, np, sA);      // C1: the fused g|f|h gradient, summed over the ranks
    if (rc) { hipStreamSynchronize(sB); mark_pending(sA); break; }
    if (flags && fuse) h->epoch_theta_done += (unsigned)((fc->base.n_tiles + ECH_WAVES - 1) / ECH_WAVES);      // (the gradient-tile kernel's workgroups)
    if (overlap) {
      if (!fuse && flags) {       // (data parallel) the step as its own launch, ordered on the device like the fused tiles
        const FitSync sy{nullptr, 0u, h->epoch_ctr, (int *)(h->epoch_ctr + 2)};      // (the chains of this minibatch have waited already)
        const int ab = (np + 1023) / 1024;
        hipLaunchKernelGGL(fit_adam_theta_sync_kernel, dim3(ab), dim3(1024), 0, sA, tb[cur], h->m1_dev, h->m2_dev, h->epoch_grad, np, lr_t,
                           ADAM_B1, ADAM_B2, ADAM_EPS, h->blob_dev, h->bblob_dev, tbl, tbl + np, tbl + 2 * (size_t)np, tTb[(cur + 1) % D], fc->mirror_dst,
                           tb[(cur + 1) % D], sy);
        BGM_HIP_CHECK(hipGetLastError());
        h->epoch_theta_done += (unsigned)ab;
      } else if (!fuse) {
        if (k >= D && !chained) BGM_HIP_CHECK(hipStreamWaitEvent(sA, h->epoch_ev_z[k % D], 0));       // latent phase k - D (same slot): it read the buffer written now
        hipLaunchKernelGGL(fit_adam_theta_kernel, dim3((np + 255) / 256), dim3(256), 0, sA, tb[cur], h->m1_dev, h->m2_dev, h->epoch_grad, np, lr_t,
                           ADAM_B1, ADAM_B2, ADAM_EPS, h->blob_dev, h->bblob_dev, tbl, tbl + np, tbl + 2 * (size_t)np, tTb[(cur + 1) % D], fc->mirror_dst,
                           tb[(cur + 1) % D]);
        BGM_HIP_CHECK(hipGetLastError());
      }
      h->t_theta += 1;
      h->sblob_valid = false; h->det_valid = false;
      cur = (cur + 1) % D;
      fc->theta_use = tb[cur]; fc->thetaT_use = tTb[cur];
      if (!flags) {
        BGM_HIP_CHECK(hipEventRecord(h->epoch_ev_t[k % D], sA));
        BGM_HIP_CHECK(hipStreamWaitEvent(sB, h->epoch_ev_t[k % D], 0));
      }
    } else if ((rc = bgm_causal_fit_theta_apply(h, h->epoch_grad, lr_theta, sA))) break;
    if (ahead) h->z_synced = h->t_z + 1;       // (minibatch k's rows were replayed to step tz0 + k = t_z)
    if (flags) {                               // the latent phase waits for this minibatch's Adam step in the kernel; the replay rides along
      fc->sync_z = FitSync{h->epoch_ctr, h->epoch_theta_done, h->epoch_ctr + 1, (int *)(h->epoch_ctr + 2)};
      if (ahead && k + D < n_mb) {
        fc->rp_idx = perm + (k + D) * batch; fc->rp_n = (int)std::min<int64_t>(batch, n_use - (k + D) * batch);
        fc->rp_t_to = (int)(tz0 + k + D); fc->rp_lr = lr_z; fc->rp_tlast = h->tlast_dev;
        replayed = k + D + 1;
      }
    }
    rc = bgm_causal_fit_z_step(h, x, y, v, data_z, zm, zv, idx, 0, b, b * world, lr_z, lazy, loss_z, sB);
    if (flags) {
      fc->sync_z = FitSync{}; fc->rp_n = 0;
      if (!rc) { h->epoch_z_done += (unsigned)fc->last_z_blocks; zt[k % D] = h->epoch_z_done; }
    }
    if (rc) { mark_pending(sB); break; }
    if (ahead && !flags && (rc = replay(k + D, sB))) { ++k; mark_pending(sB); break; }
    if (overlap && !flags) BGM_HIP_CHECK(hipEventRecord(h->epoch_ev_z[k % D], sB));
  }
  if (overlap) {
    fc->theta_use = nullptr; fc->thetaT_use = nullptr;
    if (k > 0) {      // join: whatever follows on the caller's stream sees the last latent phase ...
      BGM_HIP_CHECK(hipEventRecord(h->epoch_ev_z[0], sB));
      BGM_HIP_CHECK(hipStreamWaitEvent(sA, h->epoch_ev_z[0], 0));
    }
    if (cur != 0) {   // ... and the parameters in the session's own buffers
      BGM_HIP_CHECK(hipMemcpyAsync(h->theta_dev, tb[cur], sizeof(float) * np, hipMemcpyDeviceToDevice, sA));
      BGM_HIP_CHECK(hipMemcpyAsync(fc->thetaT, tTb[cur], sizeof(float) * ((size_t)np + 64), hipMemcpyDeviceToDevice, sA));
    }
  }
  // a device-side wait that gave up voids THIS call: say so now, before the caller evaluates or checkpoints the state (one small read
  // per epoch behind the join above)
  if (flags && !rc) rc = fit_epoch_check(h, sA);
  return rc;
}

extern "C" int bgm_causal_get_weights(bgm_handle *h, int net_id, float *theta_host, int64_t count, void *stream_) {
  if (!h || !h->configured) { bgm_set_error("bgm_causal_get_weights: handle not configured"); return BGM_E_STATE; }
  if (net_id < 0 || net_id > 3 || !theta_host) { bgm_set_error("bgm_causal_get_weights: bad argument"); return BGM_E_INVALID; }
  HostNet &n = h->nets[net_id];
  if ((size_t)count != n.count()) { bgm_set_error("bgm_causal_get_weights: wrong count"); return BGM_E_INVALID; }
  if (h->fit_active && net_id != BGM_NET_E) {
    BGM_HIP_CHECK(hipSetDevice(h->device));
    BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
    size_t off = 0;
    if (net_id == BGM_NET_F) off = h->nets[BGM_NET_G].count();
    if (net_id == BGM_NET_H) off = h->nets[BGM_NET_G].count() + h->nets[BGM_NET_F].count();
    BGM_HIP_CHECK(hipMemcpy(n.theta.data(), h->theta_dev + off, sizeof(float) * count, hipMemcpyDeviceToHost));
  }
  std::memcpy(theta_host, n.theta.data(), sizeof(float) * count);
  return BGM_OK;
}

extern "C" int bgm_causal_fit_z_grad(bgm_handle *h, const float *x, const float *y, const float *v, const float *data_z,
                                     const int32_t *idx, int64_t row_lo, int32_t batch, int32_t batch_global, float *dz_out,
                                     double *loss, void *stream_) {
  int rc = fit_check(h, x, y, v, data_z, batch, batch_global, "bgm_causal_fit_z_grad");
  if (rc) return rc;
  if (!dz_out) { bgm_set_error("bgm_causal_fit_z_grad: dz_out_dev is NULL"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  FitKArgs ka{};
  ka.m = h->meta; ka.bm = h->fit_meta; ka.ws = h->fit_ws; ka.wsp = h->ws_dev;
  ka.x = x; ka.y = y; ka.v = v; ka.data_z = data_z; ka.idx = idx; ka.row_lo = row_lo; ka.B = batch;
  ka.inv_B = 1.0f / (float)batch_global; ka.z_mode = 1; ka.loss = loss;
  if (FitChainState *fc = static_cast<FitChainState *>(h->fit_chain); fc && batch <= 32) {
    FitChainArgs ca = fc->base;
    ca.x = x; ca.y = y; ca.v = v; ca.data_z = data_z; ca.idx = idx; ca.row_lo = row_lo; ca.inv_B = ka.inv_B;
    ca.loss = loss; ca.dz = h->ws_dev + h->fit_ws.dz; ca.ws = fc->ws_z;
    fit_chain_launch(fc, ca, batch, 1, stream);
    rc = BGM_OK;
  } else if (gx_fit_active(h)) rc = gx_fit_grads(h, x, y, v, data_z, idx, row_lo, batch, batch_global, 1, nullptr, loss, stream);
  else rc = launch_fwd_bwd(h, ka, stream);
  if (rc) return rc;
  BGM_HIP_CHECK(hipMemcpyAsync(dz_out, h->ws_dev + h->fit_ws.dz, sizeof(float) * (size_t)batch * h->q, hipMemcpyDeviceToDevice, stream));
  return BGM_OK;
}

extern "C" int bgm_causal_fit_state(bgm_handle *h, int32_t write, float *m_host, float *v_host, int64_t count, int64_t *steps,
                                    void *stream_) {
  if (!h || !h->fit_active) { bgm_set_error("bgm_causal_fit_state: call bgm_causal_fit_begin first"); return BGM_E_STATE; }
  if (!m_host || !v_host || !steps || count != (int64_t)h->n_params) { bgm_set_error("bgm_causal_fit_state: bad argument (count must be n_params)"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  const size_t bytes = sizeof(float) * (size_t)count;
  if (write) {
    BGM_HIP_CHECK(hipMemcpy(h->m1_dev, m_host, bytes, hipMemcpyHostToDevice));
    BGM_HIP_CHECK(hipMemcpy(h->m2_dev, v_host, bytes, hipMemcpyHostToDevice));
    h->t_theta = steps[0]; h->t_z = steps[1];
    if (h->tlast_dev) { hipFree(h->tlast_dev); h->tlast_dev = nullptr; h->z_synced = -1; }   // rows count as current at the restored step
  } else {
    BGM_HIP_CHECK(hipMemcpy(m_host, h->m1_dev, bytes, hipMemcpyDeviceToHost));
    BGM_HIP_CHECK(hipMemcpy(v_host, h->m2_dev, bytes, hipMemcpyDeviceToHost));
    steps[0] = h->t_theta; steps[1] = h->t_z;
  }
  return BGM_OK;
}

extern "C" int bgm_causal_fit_end(bgm_handle *h, void *stream_) {
  if (!h) return BGM_E_INVALID;
  if (!h->fit_active) return BGM_OK;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  const int rc_epoch = fit_epoch_check(h, (hipStream_t)stream_);      // (reported after the session is closed)
  size_t off = 0;
  for (int id : {BGM_NET_G, BGM_NET_F, BGM_NET_H}) {   // theta order g | f | h
    HostNet &n = h->nets[id];
    BGM_HIP_CHECK(hipMemcpy(n.theta.data(), h->theta_dev + off, sizeof(float) * n.count(), hipMemcpyDeviceToHost));
    off += n.count();
  }
  h->bx_valid = false;   // the split-precision sampling blob is packed from the host copies refreshed above
  h->gx_valid = false;   // ... and so are the general-width packs, unless this session stepped on them (then they are re-uploaded once: harmless)
  fit_free(h);   // forward blob on the device is already current (blob_valid stays true)
  return rc_epoch;
}

// ===========================================================================================
// BGM.fit step functions (bgm/base.py:145-187, 399-413) -- see bgm_fit_kernels.h
// ===========================================================================================
#include "bgm_state.h"
#include "gx_bgm_host.h"

static constexpr int BGM_FIT_WAVES = 8;
static constexpr int BGM_SPLIT_MAX_WG = 8;      // workgroups of a small-minibatch pass (bgm_fit_kernels.h BgmFitSplit)

void bgm_bgm_fit_free(bgm_handle *h) {
  if (!h->bgm_state) return;
  BgmState *s = static_cast<BgmState *>(h->bgm_state);
  for (void *p : {(void *)s->theta_dev, (void *)s->m1_dev, (void *)s->m2_dev, (void *)s->tblob_dev, (void *)s->ws_dev,
                  (void *)s->partial_dev, (void *)s->bn_dev, (void *)s->tables_dev, (void *)s->split_part_dev,
                  (void *)s->epoch_grad_dev})
    if (p) hipFree(p);
  s->theta_dev = s->m1_dev = s->m2_dev = s->tblob_dev = s->ws_dev = s->partial_dev = s->bn_dev = s->split_part_dev = s->epoch_grad_dev = nullptr;
  s->tables_dev = nullptr;
  s->fit_active = false; s->gx_fit = false;
}

// training blob: same layout as the inference blob but WITHOUT folding the BatchNorm into layer 1
static void bgm_pack_training(const BgmState *s, const std::vector<float> &theta, std::vector<float> &blob) {
  const BgmMeta &m = s->tmeta;
  const int q = m.q, p = m.p, KTQ = s->KTQ, NTX = m.ntx;
  blob.assign(m.total, 0.0f);
  size_t o = 4 * (size_t)q;
  std::vector<float> W1(theta.begin() + o, theta.begin() + o + (size_t)q * 64); o += (size_t)q * 64;
  pack17(blob, m.w1, W1, q, 64, 16 * KTQ, 4, [&](int slot) { int f = l1_feature(slot); return f < q ? f : -1; });
  for (int k = 0; k < 64; ++k) blob[m.b1 + k] = theta[o + k];
  o += 64;
  auto ident = [](int r) { return r; };
  for (int l = 0; l < m.n_hh; ++l) {
    std::vector<float> W(theta.begin() + o, theta.begin() + o + 4096); o += 4096;
    pack17(blob, m.wh + l * 4 * 64 * 17, W, 64, 64, 64, 4, ident);
    for (int k = 0; k < 64; ++k) blob[m.bh + l * 64 + k] = theta[o + k];
    o += 64;
  }
  for (int head = 0; head < 2; ++head) {
    std::vector<float> W(theta.begin() + o, theta.begin() + o + (size_t)64 * p); o += (size_t)64 * p;
    pack17_heads(blob, m.whd, W.data(), p, NTX, head);
    for (int k = 0; k < p; ++k) blob[m.bhd + head * 16 * NTX + k] = theta[o + k];
    o += p;
  }
}

extern "C" int bgm_bgm_fit_n_params(bgm_handle *h, int64_t *n) {
  if (!h || !h->bgm_state || !bst(h)->configured || !n) { bgm_set_error("bgm_bgm_fit_n_params: bad argument"); return BGM_E_INVALID; }
  *n = (int64_t)bst(h)->theta.size();
  return BGM_OK;
}

extern "C" int bgm_bgm_fit_begin(bgm_handle *h, int64_t n_rows, int32_t max_batch, void *stream_) {
  (void)stream_;
  if (!h || !h->bgm_state || !bst(h)->configured || !bst(h)->set) { bgm_set_error("bgm_bgm_fit_begin: configure and set weights first"); return BGM_E_STATE; }
  if (n_rows <= 0 || max_batch <= 0) { bgm_set_error("bgm_bgm_fit_begin: n_rows / max_batch must be positive"); return BGM_E_INVALID; }
  BgmState *s = bst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  bgm_bgm_fit_free(h);
  if (gxb_wanted(s)) {         // trunk widths / depth outside the compiled blob kernels: the general-width engine (gx_bgm_api.hip)
    int rc = gxb_fit_begin(h, s, n_rows, max_batch, (hipStream_t)stream_);
    if (rc) bgm_bgm_fit_free(h);
    return rc;
  }
  const int q = s->cfg.z_dim, p = s->cfg.x_dim, NH = s->cfg.n_hidden_g;
  const int KTQ = (q + 15) / 16, NTX = (p + 15) / 16, KQ = 16 * KTQ;
  BgmMeta &m = s->tmeta;
  int ntx_variant = 0;
  s->fit_lds_bytes = bgm_layout(q, p, NH, m, ntx_variant);
  if (s->fit_lds_bytes < 0) { bgm_set_error("BGM generator does not fit the LDS layout (x_dim too large)"); return BGM_E_UNSUPPORTED; }
  s->KTQ = KTQ; s->fit_NTX = ntx_variant; s->NH = NH;      // (fit_NTX: the posterior kernels keep their own variant, bgm_api.hip)
  const int np = (int)s->theta.size();
  if (np >= (1 << 24)) { bgm_set_error("too many parameters"); return BGM_E_UNSUPPORTED; }
  s->n_params = np;
  BGM_HIP_CHECK(hipMalloc(&s->theta_dev, sizeof(float) * np));
  BGM_HIP_CHECK(hipMalloc(&s->m1_dev, sizeof(float) * np));
  BGM_HIP_CHECK(hipMalloc(&s->m2_dev, sizeof(float) * np));
  BGM_HIP_CHECK(hipMemcpy(s->theta_dev, s->theta.data(), sizeof(float) * np, hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMemset(s->m1_dev, 0, sizeof(float) * np));
  BGM_HIP_CHECK(hipMemset(s->m2_dev, 0, sizeof(float) * np));
  s->t_theta = 0; s->t_z = 0; s->batch_global = 0;
  std::vector<float> blob;
  bgm_pack_training(s, s->theta, blob);
  BGM_HIP_CHECK(hipMalloc(&s->tblob_dev, sizeof(float) * blob.size()));
  BGM_HIP_CHECK(hipMemcpy(s->tblob_dev, blob.data(), sizeof(float) * blob.size(), hipMemcpyHostToDevice));
  // canonical parameter -> training-blob position (pack an iota vector through the same packer)
  std::vector<float> iota(np);
  for (int i = 0; i < np; ++i) iota[i] = (float)(i + 1);
  std::vector<float> bidx;
  bgm_pack_training(s, iota, bidx);
  std::vector<int> tables(4 * (size_t)np, -1);
  int *dst = tables.data(), *grad_src = dst + 3 * (size_t)np;
  for (size_t d = 0; d < bidx.size(); ++d) { const int c = (int)bidx[d] - 1; if (c >= 0) dst[c] = (int)d; }
  // workspace
  const int B = (max_batch + 15) / 16 * 16;
  s->fit_bcap = B;
  BgmFitWs &w = s->fit_ws;
  std::memset(&w, 0, sizeof(w));
  w.B = B;
  long long woff = 0;
  auto wtake = [&](long long n) { long long o = woff; woff += (n + 3) / 4 * 4; return o; };
  w.zn = wtake((long long)B * KQ); w.zhat = wtake((long long)B * KQ);
  w.act = wtake((long long)NH * B * 64);
  w.omean = wtake((long long)B * 16 * NTX); w.osraw = wtake((long long)B * 16 * NTX);
  w.dact = wtake((long long)NH * B * 64);
  w.dmean = wtake((long long)B * 16 * NTX); w.dsraw = wtake((long long)B * 16 * NTX);
  w.dzn = wtake((long long)B * KQ); w.dz = wtake((long long)B * q);
  w.total = woff;
  BGM_HIP_CHECK(hipMalloc(&s->ws_dev, sizeof(float) * w.total));
  BGM_HIP_CHECK(hipMemset(s->ws_dev, 0, sizeof(float) * w.total));
  BGM_HIP_CHECK(hipMalloc(&s->bn_dev, sizeof(float) * 4 * KQ));
  // dW layers + gradient source table (theta order: gamma,beta,mmean,mvar | trunk W,b.. | mean W,b | var W,b)
  DwArgs &dw = s->dw;
  std::memset(&dw, 0, sizeof(dw));
  int nl = 0, poff = 0;
  size_t o = 4 * (size_t)q;
  auto add = [&](long long a_off, long long d_off, int K, int N, int n_in, int n_out) {
    DwLayer &L = dw.layer[nl++];
    L.a_off = a_off; L.d_off = d_off; L.K = K; L.N = N; L.out_off = poff;
    for (int i = 0; i < n_in; ++i) for (int k = 0; k < n_out; ++k) grad_src[o + (size_t)i * n_out + k] = poff + i * N + k;
    o += (size_t)n_in * n_out;
    for (int k = 0; k < n_out; ++k) grad_src[o + k] = poff + K * N + k;
    o += n_out;
    poff += K * N + N;
  };
  add(w.zn, w.dact, KQ, 64, q, 64);
  for (int l = 1; l < NH; ++l) add(w.act + (long long)(l - 1) * B * 64, w.dact + (long long)l * B * 64, 64, 64, 64, 64);
  add(w.act + (long long)(NH - 1) * B * 64, w.dmean, 64, 16 * NTX, 64, p);
  add(w.act + (long long)(NH - 1) * B * 64, w.dsraw, 64, 16 * NTX, 64, p);
  dw.n_layers = nl;
  dw.partial_stride = (poff + 3) / 4 * 4;
  s->n_slices_cap = (B + s->rows_per_slice - 1) / s->rows_per_slice;
  BGM_HIP_CHECK(hipMalloc(&s->partial_dev, sizeof(float) * dw.partial_stride * s->n_slices_cap));
  BGM_HIP_CHECK(hipMalloc(&s->split_part_dev, sizeof(float) * ((size_t)BGM_SPLIT_MAX_WG * BGM_FIT_S * 32 * 64 + 4)));
  BGM_HIP_CHECK(hipMemset(s->split_part_dev, 0, sizeof(float) * ((size_t)BGM_SPLIT_MAX_WG * BGM_FIT_S * 32 * 64 + 4)));
  BGM_HIP_CHECK(hipMalloc(&s->tables_dev, sizeof(int) * tables.size()));
  BGM_HIP_CHECK(hipMemcpy(s->tables_dev, tables.data(), sizeof(int) * tables.size(), hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipDeviceSynchronize());
  s->fit_active = true;
  s->blob_valid = false;   // the inference blob must be rebuilt from the trained parameters
  return BGM_OK;
}

#define BGM_BGM_FIT_VARIANTS(X) X(1, 2, 5) X(1, 7, 5) X(1, 0, 5) X(1, 2, 3) X(1, 7, 3) X(1, 0, 3)

static int bgm_fit_fwd_bwd(bgm_handle *h, BgmState *s, const float *x, const float *data_z, const int32_t *idx, int batch,
                           double *loss, int update_moving, hipStream_t stream) {
  const int q = s->cfg.z_dim, KQ = 16 * s->KTQ;
  const int tiles = (batch + 15) / 16;
  // <= 32 rows (the reference's batch_size): head tiles dealt over the waves of a few workgroups instead of one wave per row tile,
  // the batch statistics formed at the head of the forward kernel
  static const bool no_split = std::getenv("BGM_BGM_FIT_NO_SPLIT") != nullptr;       // dev A/B
  const bool split = tiles <= 2 && !no_split && !s->gx_fit;
  if (!split) {
    hipLaunchKernelGGL(bgm_bn_stats_kernel, dim3(1), dim3(256), 0, stream, data_z, idx, batch, q, KQ, s->theta_dev, s->bn_dev,
                       s->theta_dev + 2 * q, update_moving);
    BGM_HIP_CHECK(hipGetLastError());
  }
  if (s->gx_fit) return gxb_fit_fwd_bwd(h, s, x, data_z, idx, batch, loss, stream);
  BgmFitKArgs ka{};
  ka.blob = s->tblob_dev; ka.m = s->tmeta; ka.ws = s->fit_ws; ka.wsp = s->ws_dev; ka.x = x; ka.data_z = data_z;
  ka.idx = idx; ka.B = batch; ka.inv_B = 1.0f / (float)(s->batch_global > 0 ? s->batch_global : batch); ka.bn = s->bn_dev; ka.loss = loss;
  int grid = std::max(1, std::min((tiles + BGM_FIT_WAVES - 1) / BGM_FIT_WAVES, h->n_cus));
  int lds = s->fit_lds_bytes;
  if (split) {
    const int ntx = s->tmeta.ntx, rounds = (ntx + BGM_FIT_S - 1) / BGM_FIT_S;
    ka.split = 1;
    ka.part = s->split_part_dev;
    ka.part_ctr = reinterpret_cast<unsigned *>(s->split_part_dev + (size_t)BGM_SPLIT_MAX_WG * BGM_FIT_S * 32 * 64);
    ka.bn_w = s->bn_dev; ka.bn_theta = s->theta_dev; ka.bn_moving = s->theta_dev + 2 * q; ka.bn_update = update_moving;
    grid = std::max(1, std::min(BGM_SPLIT_MAX_WG, s->fit_NTX == 0 ? rounds : (rounds + 1) / 2));
    if (s->fit_NTX == 0) lds += (BGM_FIT_S - 2) * BGM_PAIR * (int)sizeof(float);      // the stage holds BGM_FIT_S tile pairs
  }
#define X(KTQ_, NTX_, NH_)                                                                                          \
  if (s->KTQ == KTQ_ && s->fit_NTX == NTX_ && s->NH == NH_) {                                                           \
    auto kf = bgm_fit_fwd_kernel<KTQ_, NTX_, NH_, BGM_FIT_WAVES>;                                                   \
    auto kb = bgm_fit_bwd_kernel<KTQ_, NTX_, NH_, BGM_FIT_WAVES>;                                                   \
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL(kf, dim3(grid), dim3(64 * BGM_FIT_WAVES), lds, stream, ka);                                  \
    BGM_HIP_CHECK(hipGetLastError());                                                                               \
    hipLaunchKernelGGL(kb, dim3(grid), dim3(64 * BGM_FIT_WAVES), lds, stream, ka);                                  \
    BGM_HIP_CHECK(hipGetLastError());                                                                               \
    return BGM_OK;                                                                                                  \
  }
  BGM_BGM_FIT_VARIANTS(X)
#undef X
  bgm_set_error("no compiled BGM fit kernel variant for this shape");
  return BGM_E_UNSUPPORTED;
}

static int bgm_fit_check(bgm_handle *h, const void *x, const void *z, const void *idx, int batch, const char *who) {
  if (!h || !h->bgm_state || !bst(h)->fit_active) { bgm_set_error(std::string(who) + ": call bgm_bgm_fit_begin first"); return BGM_E_STATE; }
  if (!x || !z || !idx) { bgm_set_error(std::string(who) + ": NULL pointer"); return BGM_E_INVALID; }
  if (batch <= 0 || batch > bst(h)->fit_bcap) { bgm_set_error(std::string(who) + ": batch out of range"); return BGM_E_INVALID; }
  return BGM_OK;
}

extern "C" int bgm_bgm_fit_set_global_batch(bgm_handle *h, int32_t batch_global) {
  if (!h || !h->bgm_state || !bst(h)->fit_active) { bgm_set_error("bgm_bgm_fit_set_global_batch: call bgm_bgm_fit_begin first"); return BGM_E_STATE; }
  if (batch_global < 0) { bgm_set_error("bgm_bgm_fit_set_global_batch: negative batch"); return BGM_E_INVALID; }
  bst(h)->batch_global = batch_global;
  return BGM_OK;
}

extern "C" int bgm_bgm_fit_theta_grad(bgm_handle *h, const float *x, const float *data_z, const int32_t *idx,
                                      int32_t batch, float *grad, double *loss, void *stream_) {
  int rc = bgm_fit_check(h, x, data_z, idx, batch, "bgm_bgm_fit_theta_grad");
  if (rc) return rc;
  if (!grad) { bgm_set_error("bgm_bgm_fit_theta_grad: grad_dev is NULL"); return BGM_E_INVALID; }
  BgmState *s = bst(h);
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  rc = bgm_fit_fwd_bwd(h, s, x, data_z, idx, batch, loss, 1, stream);
  if (rc) return rc;
  DwArgs dw = s->dw;
  dw.ws = s->ws_dev; dw.partial = s->partial_dev; dw.B = batch; dw.rows_per_slice = s->rows_per_slice;
  const int n_slices = (batch + s->rows_per_slice - 1) / s->rows_per_slice;
  hipLaunchKernelGGL(fit_dw_kernel, dim3(n_slices, dw.n_layers, fit_dw_chunks(dw)), dim3(256), 0, stream, dw);
  BGM_HIP_CHECK(hipGetLastError());
  const int np = s->n_params, q = s->cfg.z_dim;
  hipLaunchKernelGGL(fit_grad_reduce_kernel, dim3((np + 255) / 256), dim3(256), 0, stream, s->partial_dev, dw.partial_stride,
                     n_slices, s->tables_dev + 3 * (size_t)np, np, grad);
  BGM_HIP_CHECK(hipGetLastError());
  // d gamma, d beta of the input BatchNorm -> grad[0..2q)
  hipLaunchKernelGGL(bgm_bn_bwd_kernel, dim3(1), dim3(256), 0, stream, s->ws_dev, s->fit_ws, s->bn_dev, batch, q, 16 * s->KTQ,
                     1.0f / (float)(s->batch_global > 0 ? s->batch_global : batch), data_z, idx, grad, (float *)nullptr, 0);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bgm_fit_theta_apply(bgm_handle *h, const float *grad, float lr_theta, void *stream_) {
  if (!h || !h->bgm_state || !bst(h)->fit_active) { bgm_set_error("bgm_bgm_fit_theta_apply: call bgm_bgm_fit_begin first"); return BGM_E_STATE; }
  if (!grad) { bgm_set_error("bgm_bgm_fit_theta_apply: grad_dev is NULL"); return BGM_E_INVALID; }
  BgmState *s = bst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  s->t_theta += 1;
  const double t = (double)s->t_theta;
  const float lr_t = (float)((double)lr_theta * std::sqrt(1.0 - std::pow((double)ADAM_B2, t)) / (1.0 - std::pow((double)ADAM_B1, t)));
  const int np = s->n_params;
  const int *tb = s->tables_dev;
  // moving mean/var sit in theta but receive a zero gradient (m = v = 0 -> no Adam movement)
  hipLaunchKernelGGL(fit_adam_theta_kernel, dim3((np + 255) / 256), dim3(256), 0, (hipStream_t)stream_, s->theta_dev, s->m1_dev,
                     s->m2_dev, grad, np, lr_t, ADAM_B1, ADAM_B2, ADAM_EPS, s->gx_fit ? gxb_pack(s) : s->tblob_dev, s->gx_fit ? gxb_packT(s) : s->tblob_dev, tb,
                     tb + np, tb + 2 * (size_t)np);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bgm_fit_z_step(bgm_handle *h, const float *x, float *data_z, const int32_t *idx, int32_t batch,
                                  float lr_z, double *loss, void *stream_) {
  int rc = bgm_fit_check(h, x, data_z, idx, batch, "bgm_bgm_fit_z_step");
  if (rc) return rc;
  BgmState *s = bst(h);
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  // g_net(data_z) is called with training=True again (bgm/base.py:172): batch statistics, moving stats updated
  rc = bgm_fit_fwd_bwd(h, s, x, data_z, idx, batch, loss ? loss + 2 : nullptr, 1, stream);
  if (rc) return rc;
  const int q = s->cfg.z_dim;
  float *dz = s->ws_dev + s->fit_ws.dz;
  s->t_z += 1;
  const double t = (double)s->t_z;
  const float lr_t = (float)((double)lr_z * std::sqrt(1.0 - std::pow((double)ADAM_B2, t)) / (1.0 - std::pow((double)ADAM_B1, t)));
  // batch-norm backward -> d loss / d z (+ the prior term), and the fresh-slot Adam step on the batch rows in the same launch
  hipLaunchKernelGGL(bgm_bn_bwd_kernel, dim3(1), dim3(256), 0, stream, s->ws_dev, s->fit_ws, s->bn_dev, batch, q, 16 * s->KTQ,
                     1.0f / (float)(s->batch_global > 0 ? s->batch_global : batch), data_z, idx, (float *)nullptr, dz, 1, data_z, lr_t,
                     ADAM_B1, ADAM_B2, ADAM_EPS);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The minibatch loop of BGM.fit inside the library (single process).  replaces: the loop body bgm/base.py:399-413 for the minibatches
// perm[k * batch .. (k + 1) * batch), k = 0 .. n_steps - 1: bgm_bgm_fit_theta_grad, _theta_apply, _z_step in that order (the same
// launches, issued from C++; results identical to the per-minibatch calls).
// ---------------------------------------------------------------------------------------------------------------------------
extern "C" int bgm_bgm_fit_epoch(bgm_handle *h, const float *x, float *data_z, const int32_t *perm, int64_t n_steps, int32_t batch,
                                 float lr_theta, float lr_z, double *loss, void *stream_) {
  int rc = bgm_fit_check(h, x, data_z, perm, batch, "bgm_bgm_fit_epoch");
  if (rc) return rc;
  if (n_steps < 0) { bgm_set_error("bgm_bgm_fit_epoch: negative step count"); return BGM_E_INVALID; }
  BgmState *s = bst(h);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (!s->epoch_grad_dev) BGM_HIP_CHECK(hipMalloc(&s->epoch_grad_dev, sizeof(float) * s->n_params));
  for (int64_t k = 0; k < n_steps; ++k) {
    const int32_t *idx = perm + k * batch;
    if ((rc = bgm_bgm_fit_theta_grad(h, x, data_z, idx, batch, s->epoch_grad_dev, loss, stream_))) return rc;
    if ((rc = bgm_bgm_fit_theta_apply(h, s->epoch_grad_dev, lr_theta, stream_))) return rc;
    if ((rc = bgm_bgm_fit_z_step(h, x, data_z, idx, batch, lr_z, loss, stream_))) return rc;
  }
  return BGM_OK;
}

extern "C" int bgm_bgm_get_weights(bgm_handle *h, float *theta_host, int64_t count, void *stream_) {
  if (!h || !h->bgm_state || !bst(h)->configured) { bgm_set_error("bgm_bgm_get_weights: not configured"); return BGM_E_STATE; }
  BgmState *s = bst(h);
  if (!theta_host || (size_t)count != s->theta.size()) { bgm_set_error("bgm_bgm_get_weights: wrong count"); return BGM_E_INVALID; }
  if (s->fit_active) {
    BGM_HIP_CHECK(hipSetDevice(h->device));
    BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
    BGM_HIP_CHECK(hipMemcpy(s->theta.data(), s->theta_dev, sizeof(float) * count, hipMemcpyDeviceToHost));
    s->blob_valid = false;
  }
  std::memcpy(theta_host, s->theta.data(), sizeof(float) * count);
  return BGM_OK;
}

extern "C" int bgm_bgm_fit_end(bgm_handle *h, void *stream_) {
  if (!h || !h->bgm_state) return BGM_E_INVALID;
  BgmState *s = bst(h);
  if (!s->fit_active) return BGM_OK;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  BGM_HIP_CHECK(hipMemcpy(s->theta.data(), s->theta_dev, sizeof(float) * s->theta.size(), hipMemcpyDeviceToHost));
  bgm_bgm_fit_free(h);
  s->blob_valid = false;   // inference blob (BN folded with the new moving statistics) is rebuilt on next use
  s->gx_valid = false;
  return BGM_OK;
}
