// gx_bgm_api.hip -- host side of the general-width engine for BGM (gx_bgm_kernels.h): generators g_net = BaseVariationalNet with
// arbitrary params['g_units'] / ['z_dim'] (bgm/base.py:59-80 forwards any list to networks/base.py:53-117; the reference's own
// integration test builds g_units (8, 8), z_dim 3, x_dim 8: r-package/bayesgm/tests/testthat/test-bgm.R:31-36), entered from
// bgm_bgm_logpost / _hmc_run / _predict_draws (bgm_api.hip) and bgm_bgm_fit_* (fit_api.hip) whenever the trunk is not [64] x 3 or
// [64] x 5 with z_dim <= 16 -- the shapes the dual-access-blob kernels of bgm_kernels.h are compiled for.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bgm_state.h"
#include "gx_bgm_host.h"
#include "gx_bgm_kernels.h"

namespace {

struct GxbState {
  GxBgmModel m{};
  float *pack = nullptr, *packT = nullptr, *bn4 = nullptr;
  size_t pack_floats = 0, packT_floats = 0;
  std::vector<int> fwd_map, bwd_map;       // canonical parameter -> pack / packT position (-1: none)
  int lds_bytes = 0, lds_fit = 0;
  long long act[GX_MAXL]{}, dy[GX_MAXL]{};
};
GxbState *gxs(const BgmState *s) { return static_cast<GxbState *>(s->gx); }

template <class K>
int set_lds(K kernel, int bytes) {
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return BGM_OK;
}

int session(bgm_handle *h, BgmState *s, GxbState *&g, hipStream_t stream) {
  if (!s->set) { bgm_set_error("BGM weights not set"); return BGM_E_STATE; }
  g = gxs(s);
  const int q = s->cfg.z_dim, p = s->cfg.x_dim, NH = s->cfg.n_hidden_g;
  if (!g) {
    if (NH + 1 > GX_MAXL) { bgm_set_error("general-width engine: too many layers"); return BGM_E_UNSUPPORTED; }
    GxbState *n = new GxbState();
    GxBgmModel &m = n->m;
    const int Pp = gx_pad32(p);
    m.q = q; m.p = p; m.Pp = Pp;
    GxNet &G = m.g;
    G.L = NH + 1;
    G.dim[0] = q; G.pad[0] = gx_pad32(q);
    for (int i = 0; i < NH; ++i) { G.dim[i + 1] = s->cfg.g_units[i]; G.pad[i + 1] = gx_pad32(s->cfg.g_units[i]); }
    G.dim[NH + 1] = 2 * Pp; G.pad[NH + 1] = 2 * Pp;
    n->fwd_map.assign(s->theta.size(), -1); n->bwd_map.assign(s->theta.size(), -1);
    size_t off = 0, offT = 0, c = 4 * (size_t)q;
    int wmax = 32, mb = 0;
    for (int l = 0; l <= NH; ++l) {
      const int Kp = G.pad[l], Np = G.pad[l + 1];
      G.w[l] = (int)off; off += (size_t)Kp * Np;
      G.b[l] = (int)off; off += Np;
      G.wt[l] = (int)offT; offT += (size_t)Np * Kp;
      wmax = std::max(wmax, Kp);
      if (l < NH) {
        const int K = G.dim[l], N = G.dim[l + 1];
        for (int i = 0; i < K; ++i) for (int o = 0; o < N; ++o) { n->fwd_map[c] = G.w[l] + i * Np + o; n->bwd_map[c] = G.wt[l] + o * Kp + i; ++c; }
        for (int o = 0; o < N; ++o) n->fwd_map[c++] = G.b[l] + o;
        m.moff[l] = mb; mb += GX_ROWS * (Np >> 1);
      } else {
        const int K = G.dim[l];
        for (int head = 0; head < 2; ++head) {       // mean_layer, then var_layer (bgm_bgm_set_weights order)
          for (int i = 0; i < K; ++i) for (int o = 0; o < p; ++o) { n->fwd_map[c] = G.w[l] + i * Np + head * Pp + o; n->bwd_map[c] = G.wt[l] + (head * Pp + o) * Kp + i; ++c; }
          for (int o = 0; o < p; ++o) n->fwd_map[c++] = G.b[l] + head * Pp + o;
        }
      }
    }
    m.mask_bytes = mb;
    m.ld = gx_ld(wmax);
    if (const char *e = std::getenv("BGM_GX_LD")) m.ld = std::max(m.ld, gx_ld(std::atoi(e)));      // dev: wider head chunks
    m.ch = std::min(Pp, (m.ld - 8) / 32 * 32);
    n->lds_bytes = gx_bgm_lds_bytes(m.ld, q, mb);
    n->lds_fit = 4 * (2 * GX_ROWS * m.ld + 64);
    if (n->lds_bytes > 160 * 1024 || off >= (1u << 30)) { delete n; bgm_set_error("general-width engine (BGM): trunk too wide for the 32-row LDS tiles (hidden widths up to ~280)"); return BGM_E_UNSUPPORTED; }
    n->pack_floats = off; n->packT_floats = offT;
    if (hipMalloc((void **)&n->pack, sizeof(float) * off) != hipSuccess || hipMalloc((void **)&n->packT, sizeof(float) * offT) != hipSuccess ||
        hipMalloc((void **)&n->bn4, sizeof(float) * 4 * q) != hipSuccess) {
      delete n; bgm_set_error("general-width engine (BGM): device allocation failed"); return BGM_E_HIP;
    }
    m.pack = n->pack; m.packT = n->packT;
    s->gx = n; s->gx_valid = false;
    g = n;
  }
  if (!s->gx_valid) {
    std::vector<float> pk(g->pack_floats, 0.0f), pt(g->packT_floats, 0.0f);
    for (size_t c = 0; c < s->theta.size(); ++c) {
      if (g->fwd_map[c] >= 0) pk[g->fwd_map[c]] = s->theta[c];
      if (g->bwd_map[c] >= 0) pt[g->bwd_map[c]] = s->theta[c];
    }
    BGM_HIP_CHECK(hipStreamSynchronize(stream));
    BGM_HIP_CHECK(hipMemcpy(g->pack, pk.data(), sizeof(float) * pk.size(), hipMemcpyHostToDevice));
    BGM_HIP_CHECK(hipMemcpy(g->packT, pt.data(), sizeof(float) * pt.size(), hipMemcpyHostToDevice));
    BGM_HIP_CHECK(hipMemcpy(g->bn4, s->theta.data(), sizeof(float) * 4 * q, hipMemcpyHostToDevice));
    s->gx_valid = true;
  }
  g->m.bnp = (s->fit_active && s->gx_fit) ? s->theta_dev : g->bn4;      // a fit session keeps gamma | beta | moving statistics on the device
  return BGM_OK;
}

// persistent workgroups: as many per CU as the LDS holds (up to four: 16 waves per CU hide each other's L2 latencies)
int grid_for(const bgm_handle *h, long long tiles, int lds_bytes) {
  const int occ = std::max(1, std::min(4, (160 * 1024) / std::max(lds_bytes, 1)));
  return (int)std::max<long long>(1, std::min<long long>(tiles, (long long)h->n_cus * occ));
}

}  // namespace

bool gxb_wanted(const BgmState *s) {
  const char *e = std::getenv("BGM_FORCE_GX");
  if (e && e[0] == '1') return true;
  const bgm_bgm_config &c = s->cfg;
  for (int i = 0; i < c.n_hidden_g; ++i) if (c.g_units[i] != 64) return true;
  return !(c.n_hidden_g == 3 || c.n_hidden_g == 5) || c.z_dim > 16;
}

void gxb_free(BgmState *s) {
  GxbState *g = gxs(s);
  if (!g) return;
  for (void *p : {(void *)g->pack, (void *)g->packT, (void *)g->bn4}) if (p) hipFree(p);
  delete g;
  s->gx = nullptr; s->gx_valid = false; s->gx_fit = false;
}

int gxb_logpost(bgm_handle *h, BgmState *s, const float *z, const float *x, int64_t n, float *out, float *grad, hipStream_t stream) {
  GxbState *g;
  int rc = session(h, s, g, stream);
  if (rc) return rc;
  rc = set_lds(gx_bgm_logpost_kernel, g->lds_bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(gx_bgm_logpost_kernel, dim3(grid_for(h, (n + GX_ROWS - 1) / GX_ROWS, g->lds_bytes)), dim3(GX_THREADS), g->lds_bytes, stream, g->m, z, x,
                     (long long)n, out, grad);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

int gxb_hmc_run(bgm_handle *h, BgmState *s, const bgm_hmc_args *a, hipStream_t stream) {
  GxbState *g;
  int rc = session(h, s, g, stream);
  if (rc) return rc;
  GxHmcArgs k{};
  k.m = g->m; k.x = a->x_dev; k.n = a->n; k.row_base = a->row_base; k.state = a->state_dev; k.logp = a->logp_dev; k.grad = a->grad_dev;
  k.init = a->init; k.it_begin = a->it_begin; k.n_iters = a->n_iters; k.burn_in = a->burn_in; k.n_leapfrog = a->n_leapfrog; k.step = a->step_dev;
  k.k0 = (unsigned)(a->seed & 0xFFFFFFFFull); k.k1 = (unsigned)(a->seed >> 32);
  k.acc_prob_sum = a->acc_prob_sum_dev; k.acc_count = a->acc_count_dev; k.draws = a->draws_dev;
  rc = set_lds(gx_bgm_hmc_kernel, g->lds_bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(gx_bgm_hmc_kernel, dim3(grid_for(h, (a->n + GX_ROWS - 1) / GX_ROWS, g->lds_bytes)), dim3(GX_THREADS), g->lds_bytes, stream, k);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

int gxb_predict_draws(bgm_handle *h, BgmState *s, const float *draws, int64_t n, int64_t row_base, int32_t n_draws, int32_t burn_in,
                      uint64_t seed, const int32_t *slot, int32_t k_slots, float *cells, float *full, float *var_full, int32_t add_noise,
                      hipStream_t stream) {
  GxbState *g;
  int rc = session(h, s, g, stream);
  if (rc) return rc;
  GxPredArgs k{};
  k.m = g->m; k.draws = draws; k.n = n; k.row_base = row_base; k.n_draws = n_draws; k.burn_in = burn_in; k.k_slots = k_slots; k.slot = slot;
  k.cells = cells; k.full = full; k.var_full = var_full; k.add_noise = add_noise;
  k.k0 = (unsigned)(seed & 0xFFFFFFFFull); k.k1 = (unsigned)(seed >> 32);
  rc = set_lds(gx_bgm_predict_kernel, g->lds_bytes);
  if (rc) return rc;
  const long long work = ((n + GX_ROWS - 1) / GX_ROWS) * (long long)n_draws;
  hipLaunchKernelGGL(gx_bgm_predict_kernel, dim3(grid_for(h, work, g->lds_bytes)), dim3(GX_THREADS), g->lds_bytes, stream, k);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// fit session: everything bgm_bgm_fit_begin sets up for the blob kernels, for the general-width engine
// ---------------------------------------------------------------------------------------------------------------------------
int gxb_fit_begin(bgm_handle *h, BgmState *s, int64_t n_rows, int32_t max_batch, hipStream_t stream) {
  (void)n_rows;
  GxbState *g;
  int rc = session(h, s, g, stream);
  if (rc) return rc;
  const GxBgmModel &m = g->m;
  const int q = m.q, p = m.p, np = (int)s->theta.size(), KQ = m.g.pad[0], T = m.g.L - 1, Pp = m.Pp;
  if (np >= (1 << 24)) { bgm_set_error("too many parameters"); return BGM_E_UNSUPPORTED; }
  s->n_params = np;
  BGM_HIP_CHECK(hipMalloc(&s->theta_dev, sizeof(float) * np));
  BGM_HIP_CHECK(hipMalloc(&s->m1_dev, sizeof(float) * np));
  BGM_HIP_CHECK(hipMalloc(&s->m2_dev, sizeof(float) * np));
  BGM_HIP_CHECK(hipMemcpy(s->theta_dev, s->theta.data(), sizeof(float) * np, hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMemset(s->m1_dev, 0, sizeof(float) * np));
  BGM_HIP_CHECK(hipMemset(s->m2_dev, 0, sizeof(float) * np));
  s->t_theta = 0; s->t_z = 0; s->batch_global = 0;
  s->KTQ = KQ / 16;                           // the shared BatchNorm kernels take the row stride 16 * KTQ
  const int B = (max_batch + GX_ROWS - 1) / GX_ROWS * GX_ROWS;
  s->fit_bcap = B;
  std::vector<int> tables(4 * (size_t)np, -1);
  int *fwd_dst = tables.data(), *bwd_dst = fwd_dst + 2 * (size_t)np, *grad_src = bwd_dst + np;
  for (int c = 0; c < np; ++c) { fwd_dst[c] = g->fwd_map[c]; bwd_dst[c] = g->bwd_map[c]; }
  BgmFitWs &w = s->fit_ws;
  std::memset(&w, 0, sizeof(w));
  w.B = B;
  long long off = 0;
  auto take = [&](long long n) { long long o = off; off += (n + 31) / 32 * 32; return o; };
  DwArgs &dw = s->dw;
  std::memset(&dw, 0, sizeof(dw));
  int nl = 0, poff = 0;
  size_t c = 4 * (size_t)q;
  for (int l = 0; l <= T; ++l) {
    g->act[l] = take((long long)B * m.g.pad[l]);
    g->dy[l] = take((long long)B * m.g.pad[l + 1]);
    DwLayer &L = dw.layer[nl++];
    L.a_off = g->act[l]; L.d_off = g->dy[l]; L.K = m.g.pad[l]; L.N = m.g.pad[l + 1]; L.out_off = poff;
    if (l < T) {
      for (int i = 0; i < m.g.dim[l]; ++i) for (int o = 0; o < m.g.dim[l + 1]; ++o) grad_src[c++] = poff + i * L.N + o;
      for (int o = 0; o < m.g.dim[l + 1]; ++o) grad_src[c++] = poff + L.K * L.N + o;
    } else {
      for (int head = 0; head < 2; ++head) {
        for (int i = 0; i < m.g.dim[l]; ++i) for (int o = 0; o < p; ++o) grad_src[c++] = poff + i * L.N + head * Pp + o;
        for (int o = 0; o < p; ++o) grad_src[c++] = poff + L.K * L.N + head * Pp + o;
      }
    }
    poff += L.K * L.N + L.N;
  }
  w.zn = g->act[0];
  w.zhat = take((long long)B * KQ); w.dzn = take((long long)B * KQ); w.dz = take((long long)B * q);
  w.total = off;
  dw.n_layers = nl;
  dw.partial_stride = (poff + 3) / 4 * 4;
  s->n_slices_cap = (B + s->rows_per_slice - 1) / s->rows_per_slice;
  BGM_HIP_CHECK(hipMalloc(&s->ws_dev, sizeof(float) * off));
  BGM_HIP_CHECK(hipMemset(s->ws_dev, 0, sizeof(float) * off));
  BGM_HIP_CHECK(hipMalloc(&s->bn_dev, sizeof(float) * 4 * KQ));
  BGM_HIP_CHECK(hipMalloc(&s->partial_dev, sizeof(float) * dw.partial_stride * s->n_slices_cap));
  BGM_HIP_CHECK(hipMalloc(&s->tables_dev, sizeof(int) * tables.size()));
  BGM_HIP_CHECK(hipMemcpy(s->tables_dev, tables.data(), sizeof(int) * tables.size(), hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipDeviceSynchronize());
  s->fit_active = true; s->gx_fit = true;
  s->blob_valid = false;
  return BGM_OK;
}

float *gxb_pack(BgmState *s) { return gxs(s) ? gxs(s)->pack : nullptr; }
float *gxb_packT(BgmState *s) { return gxs(s) ? gxs(s)->packT : nullptr; }

// forward + backward of one minibatch (after bgm_bn_stats_kernel has filled s->bn_dev): layer inputs / pre-activation gradients and
// d loss / d zn in the workspace, for fit_dw_kernel and bgm_bn_bwd_kernel
int gxb_fit_fwd_bwd(bgm_handle *h, BgmState *s, const float *x, const float *data_z, const int32_t *idx, int batch, double *loss,
                    hipStream_t stream) {
  GxbState *g = gxs(s);
  if (!g || !s->gx_fit) { bgm_set_error("general-width engine (BGM): no fit session"); return BGM_E_STATE; }
  GxBgmFitArgs a{};
  a.m = g->m; a.m.bnp = s->theta_dev;
  for (int l = 0; l < GX_MAXL; ++l) { a.act[l] = g->act[l]; a.dy[l] = g->dy[l]; }
  a.zhat = s->fit_ws.zhat; a.dzn = s->fit_ws.dzn; a.ws = s->ws_dev; a.x = x; a.data_z = data_z; a.idx = idx; a.B = batch;
  a.inv_B = 1.0f / (float)(s->batch_global > 0 ? s->batch_global : batch); a.bn = s->bn_dev; a.loss = loss;
  int rc = set_lds(gx_bgm_fit_kernel, g->lds_fit);
  if (rc) return rc;
  const int tiles = (batch + GX_ROWS - 1) / GX_ROWS;
  hipLaunchKernelGGL(gx_bgm_fit_kernel, dim3(std::min(tiles, 2 * h->n_cus)), dim3(GX_THREADS), g->lds_fit, stream, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}
