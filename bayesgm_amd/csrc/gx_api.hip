// gx_api.hip -- host side of the general-width engine (gx_device.h, gx_causal_kernels.h, gx_fit_kernels.h): CausalBGM models with
// arbitrary params['g_units' | 'e_units' | 'f_units' | 'h_units'] (causalbgm/base.py:64-81 forwards any list to
// BaseFullyConnectedNet, networks/base.py:7-28), entered from bgm_causal_logpost / _mh_run / _effects / _evaluate / _encode /
// _fit_* (causal_api.hip, aux_kernels.hip, fit_api.hip) whenever the hidden layers are not the reference defaults the other
// kernel families are compiled for.  Also the fit path of default-width models that the row-tile chains do not hold
// (v_dim > 207, minibatches > 32 rows of a chain-only shape) and the encoder beyond v_dim = 208.
//
// Device state: ONE padded copy of the Keras-order parameters of g, f, h, e (every layer [K][N] with K, N rounded up to multiples of
// 32, zero filled) and one transposed copy for the backward products.  During a fit session the Adam kernel scatters each updated
// parameter into both copies through index tables (fit_adam_theta_kernel), so sampling between epochs always sees the current nets.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bgm_host.h"
#include "gx_fit_kernels.h"
#include "gw_kernels.h"
#include "gx_host.h"
#include "bnf_det_host.h"

namespace {

struct GxState {
  GxCausalModel m{};
  float *pack = nullptr, *packT = nullptr;
  float *packF = nullptr;      // fragment pack of the row-tile-per-wave kernels (gx_device.h, FRAG): pack with every weight matrix in fragment order
  unsigned *packx = nullptr;   // split-precision pack of g, f, h (gx_dense_x3): hi / lo fp16 fragments, GxNet::wx offsets in dwords
  size_t packx_dwords = 0;
  bool x3_ok = false;          // every layer input of g, f, h within GX_X3_MAXKB K blocks
  bool x3_k64 = false;         // ... within 2 (the kernels with a fourth wave per SIMD)
  size_t pack_floats = 0, packT_floats = 0;
  std::vector<int> fwd_map[4], bwd_map[4];     // canonical parameter of net (G, F, H, E) -> position in pack / packT (-1: none)
  int ld_enc = 0, kc = 0;
  int lds_bytes = 0, lds_enc = 0, lds_fit = 0, occ = 1;
  // row-tile-per-wave sampling kernels (gw_kernels.h): used when a wave's LDS region stays small enough for >= 8 waves per CU
  bool gw = false;
  int gw_db = 1, gw_lds = 0, gw_occ = 1, gw_ld = 0, gw_ldf = 0;
  bool fit = false;
  long long fit_dzp = 0, fit_cnt = 0;
  int fit_dzp_rows = 0;
  GxFitNet wg{}, wf{}, wh{};
};

GxState *gst(const bgm_handle *h) { return static_cast<GxState *>(h->gx_state); }

bool force_gx() { const char *e = std::getenv("BGM_FORCE_GX"); return e && e[0] == '1'; }

template <class K>
int set_lds(K kernel, int bytes) {
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return BGM_OK;
}

void plan_net(const HostNet &n, GxNet &g, size_t &off, size_t &offT, std::vector<int> &fwd, std::vector<int> &bwd) {
  g.L = (int)n.dims.size() - 1;
  for (int l = 0; l <= g.L; ++l) { g.dim[l] = n.dims[l]; g.pad[l] = gx_pad32(n.dims[l]); }
  fwd.assign(n.count(), -1); bwd.assign(n.count(), -1);
  size_t c = 0;
  for (int l = 0; l < g.L; ++l) {
    const int K = g.dim[l], N = g.dim[l + 1], Kp = g.pad[l], Np = g.pad[l + 1];
    g.w[l] = (int)off; off += (size_t)Kp * Np;
    g.b[l] = (int)off; off += Np;
    g.wt[l] = (int)offT; offT += (size_t)Np * Kp;
    for (int i = 0; i < K; ++i)
      for (int o = 0; o < N; ++o) { fwd[c] = g.w[l] + i * Np + o; bwd[c] = g.wt[l] + o * Kp + i; ++c; }
    for (int o = 0; o < N; ++o) fwd[c++] = g.b[l] + o;
  }
}

int gx_session(bgm_handle *h, GxState *&st, hipStream_t stream, bool need_ghf = true) {
  const HostNet *nets[4] = {&h->nets[BGM_NET_G], &h->nets[BGM_NET_F], &h->nets[BGM_NET_H], &h->nets[BGM_NET_E]};
  if (need_ghf && (!nets[0]->set || !nets[1]->set || !nets[2]->set)) { bgm_set_error("weights of g, f, h must be set (bgm_causal_set_weights)"); return BGM_E_STATE; }
  st = gst(h);
  if (!st) {
    for (const HostNet *n : nets)
      if ((int)n->dims.size() - 1 > GX_MAXL) { bgm_set_error("general-width engine: too many layers"); return BGM_E_UNSUPPORTED; }
    GxState *s = new GxState();
    GxCausalModel &m = s->m;
    size_t off = 0, offT = 0;
    GxNet *gn[4] = {&m.g, &m.f, &m.h, &m.e};
    for (int k = 0; k < 4; ++k) plan_net(*nets[k], *gn[k], off, offT, s->fwd_map[k], s->bwd_map[k]);
    if (off >= (1u << 30) || offT >= (1u << 30)) { delete s; bgm_set_error("general-width engine: networks too large for 32-bit offsets"); return BGM_E_UNSUPPORTED; }
    s->pack_floats = off; s->packT_floats = offT;
    {   // split-precision pack: per layer of g, f, h (column groups of 32) x (K blocks of 32) x 64 lanes x 16 dwords
      size_t offx = 0;
      bool k64 = true;
      s->x3_ok = true;
      for (int k = 0; k < 3; ++k)
        for (int l = 0; l < gn[k]->L; ++l) {
          gn[k]->wx[l] = (int)offx;
          offx += (size_t)(gn[k]->pad[l + 1] / 32) * (size_t)(gn[k]->pad[l] / 32) * 64 * 16;
          if (gn[k]->pad[l] / 32 > GX_X3_MAXKB) s->x3_ok = false;
          if (gn[k]->pad[l] / 32 > 2) k64 = false;
        }
      s->packx_dwords = offx;
      s->x3_k64 = k64 && std::getenv("BGM_GW_X3_NO_K64") == nullptr;
      if (offx >= (1u << 30)) s->x3_ok = false;
    }
    m.q = h->q; m.p = h->p; m.z0 = h->cfg.z_dims[0]; m.z1 = h->cfg.z_dims[1]; m.z2 = h->cfg.z_dims[2]; m.binary = h->cfg.binary_treatment ? 1 : 0;
    auto s2 = [](float s_) { return s_ > 0.0f ? s_ * s_ : -1.0f; };
    m.sig2_v = s2(h->cfg.sigma_v); m.sig2_x = s2(h->cfg.sigma_x); m.sig2_y = s2(h->cfg.sigma_y);
    // widest activation that lives in LDS: every layer input of g, f, h and the (2-wide) outputs of f, h -- not g's p-wide output
    int wmax = 32;
    for (int l = 0; l < m.g.L; ++l) wmax = std::max(wmax, m.g.pad[l]);
    for (int l = 0; l <= m.f.L; ++l) wmax = std::max(wmax, m.f.pad[l]);
    for (int l = 0; l <= m.h.L; ++l) wmax = std::max(wmax, m.h.pad[l]);
    m.ld = gx_ld(wmax);
    m.ncg = m.g.pad[m.g.L] / 32;
    {   // doses per pass of the effect routine (more rows per barrier and per weight fetch for the small outcome net): as many as leave
        // two workgroups per CU (measured, profiles/r04_gx_cost.txt: the kept phase gains 10-25 %, the transitions lose ~7 % at 2 instead of 4)
      int wf = 32;
      for (int l = 0; l <= m.f.L; ++l) wf = std::max(wf, m.f.pad[l]);
      m.ldf = gx_ld(wf);
      auto occ_of = [](int bytes) { return std::max(1, std::min(4, (160 * 1024) / std::max(bytes, 1))); };
      const int occ1 = occ_of(4 * gx_causal_lds_floats(m.ld, m.q, m.ncg, m.ldf, 1));
      m.db = 1;
      static const bool no_db = std::getenv("BGM_GX_NO_DOSE_BATCH") != nullptr;
      for (int cand : {4, 3, 2})
        if (!no_db && 4 * gx_causal_lds_floats(m.ld, m.q, m.ncg, m.ldf, cand) <= 160 * 1024 && occ_of(4 * gx_causal_lds_floats(m.ld, m.q, m.ncg, m.ldf, cand)) >= std::min(occ1, 2)) { m.db = cand; break; }
    }
    if (const char *f_ = std::getenv("BGM_GX_DB")) {      // dev: force the dose batch
        const int want = std::max(1, std::min(GX_MAXDB, std::atoi(f_)));
        if (4 * gx_causal_lds_floats(m.ld, m.q, m.ncg, m.ldf, want) <= 160 * 1024) m.db = want;
      }
      s->lds_bytes = 4 * gx_causal_lds_floats(m.ld, m.q, m.ncg, m.ldf, m.db);
    {   // gw: log posterior / MH / effects with one row tile per wave; the mapping needs a wave's region within 24 KB (hidden layers up to ~128 wide)
      static const bool no_gw = std::getenv("BGM_NO_GW") != nullptr;      // dev A/B
      int wf = 32;
      for (int l = 0; l <= m.f.L; ++l) wf = std::max(wf, m.f.pad[l]);
      int wg_ = 32;
      for (int l = 0; l < m.g.L; ++l) wg_ = std::max(wg_, m.g.pad[l]);
      for (int l = 0; l <= m.f.L; ++l) wg_ = std::max(wg_, m.f.pad[l]);
      for (int l = 0; l <= m.h.L; ++l) wg_ = std::max(wg_, m.h.pad[l]);
      const int ldf = gw_ld(wf);
      s->gw_ld = gw_ld(wg_); s->gw_ldf = ldf;
      s->gw = !no_gw && 4 * gw_wave_floats(s->gw_ld, m.q, ldf, 1) <= 24 * 1024;
      s->gw_db = 1;      // doses per pass of the effect routine: one (measured: 2 doses at 8 waves per CU cost 10 % against 1 at 12)
      if (const char *f_ = std::getenv("BGM_GW_DB")) s->gw_db = std::max(1, std::min(GX_MAXDB, std::atoi(f_)));
      s->gw_lds = 4 * GW_WAVES * gw_wave_floats(s->gw_ld, m.q, ldf, s->gw_db);
      if (s->gw_lds > 160 * 1024) s->gw = false;
      s->gw_occ = std::max(1, std::min(4, (160 * 1024) / std::max(s->gw_lds, 1)));
      if (const char *o_ = std::getenv("BGM_GW_OCC")) s->gw_occ = std::max(1, std::min(std::atoi(o_), (160 * 1024) / std::max(s->gw_lds, 1)));
    }
    s->lds_fit = 4 * gx_fit_lds_floats(m.ld, m.q);
    int wenc = 32;
    for (int l = 1; l <= m.e.L; ++l) wenc = std::max(wenc, m.e.pad[l]);
    s->kc = std::min(m.e.pad[0], std::max(wenc, 256));       // columns of V staged per chunk
    s->ld_enc = gx_ld(std::max(wenc, s->kc));
    s->lds_enc = 4 * 2 * GX_ROWS * s->ld_enc;
    if (std::max(s->lds_bytes, std::max(s->lds_fit, s->lds_enc)) > 160 * 1024) {
      delete s; bgm_set_error("general-width engine: a hidden layer is too wide for the 32-row LDS tiles (hidden widths up to ~550)"); return BGM_E_UNSUPPORTED;
    }
    s->occ = std::max(1, std::min(4, (160 * 1024) / std::max(s->lds_bytes, 1)));      // up to 16 waves per CU hide each other's L2 latencies
    if (const char *o_ = std::getenv("BGM_GX_OCC")) s->occ = std::max(1, std::min(std::atoi(o_), (160 * 1024) / std::max(s->lds_bytes, 1)));      // dev: workgroups per CU
    if (hipMalloc((void **)&s->pack, sizeof(float) * std::max<size_t>(off, 1)) != hipSuccess ||
        hipMalloc((void **)&s->packT, sizeof(float) * std::max<size_t>(offT, 1)) != hipSuccess ||
        hipMalloc((void **)&s->packF, sizeof(float) * std::max<size_t>(off, 1)) != hipSuccess) {
      if (s->pack) hipFree(s->pack);
      if (s->packT) hipFree(s->packT);
      delete s; bgm_set_error("general-width engine: device allocation failed"); return BGM_E_HIP;
    }
    m.pack = s->pack;
    h->gx_state = s; h->gx_valid = false;
    st = s;
  }
  st->m.prior_seg = h->prior_seg; st->m.prior_tab = h->prior_tab;
  if (!h->gx_valid) {
    std::vector<float> pk(st->pack_floats, 0.0f), pt(st->packT_floats, 0.0f);
    for (int k = 0; k < 4; ++k) {
      const HostNet &n = *nets[k];
      if (!n.set) continue;
      for (size_t c = 0; c < n.theta.size(); ++c) {
        if (st->fwd_map[k][c] >= 0) pk[st->fwd_map[k][c]] = n.theta[c];
        if (st->bwd_map[k][c] >= 0) pt[st->bwd_map[k][c]] = n.theta[c];
      }
    }
    BGM_HIP_CHECK(hipStreamSynchronize(stream));
    BGM_HIP_CHECK(hipMemcpy(st->pack, pk.data(), sizeof(float) * pk.size(), hipMemcpyHostToDevice));
    BGM_HIP_CHECK(hipMemcpy(st->packT, pt.data(), sizeof(float) * pt.size(), hipMemcpyHostToDevice));
    if (st->gw) {      // fragment pack: biases where the padded pack has them, W_l [Kp][Np] re-ordered per (column group, K block, lane)
      std::vector<float> pf(pk);
      const GxNet *gn[3] = {&st->m.g, &st->m.f, &st->m.h};
      for (const GxNet *net : gn)
        for (int l = 0; l < net->L; ++l) {
          const int Np = net->pad[l + 1], KB = gw_k16(*net, l) / 16;      // (rows beyond the true input width are zero in the padded pack)
          const float *src = pk.data() + net->w[l];
          float *dst = pf.data() + net->w[l];
          for (int cg = 0; cg < Np / 32; ++cg)
            for (int kb = 0; kb < KB; ++kb)
              for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                  const int j = lane & 15, g = lane >> 4, sidx = i >> 1, c = i & 1;
                  dst[((size_t)(cg * KB + kb) * 64 + lane) * 8 + i] = src[(size_t)(16 * kb + 4 * g + sidx) * Np + 32 * cg + 2 * j + c];
                }
        }
      BGM_HIP_CHECK(hipMemcpy(st->packF, pf.data(), sizeof(float) * pf.size(), hipMemcpyHostToDevice));
    }
    if (st->gw && st->x3_ok && h->precision == 2) {      // split pack (built when the mode is on: bgm_causal_set_precision invalidates the packs)
      std::vector<unsigned> px(st->packx_dwords, 0u);
      auto half_bits = [](float x) { const _Float16 hx = (_Float16)x; unsigned short b; std::memcpy(&b, &hx, 2); return b; };
      const GxNet *gn[3] = {&st->m.g, &st->m.f, &st->m.h};
      for (const GxNet *net : gn)
        for (int l = 0; l < net->L; ++l) {
          const int Np = net->pad[l + 1], KB = net->pad[l] / 32;
          const float *src = pk.data() + net->w[l];      // [Kp][Np], zeros beyond the true widths
          unsigned *dst = px.data() + net->wx[l];
          for (int cg = 0; cg < Np / 32; ++cg)
            for (int kb = 0; kb < KB; ++kb)
              for (int lane = 0; lane < 64; ++lane) {
                const int j = lane & 15, g = lane >> 4;
                unsigned *d = dst + ((size_t)(cg * KB + kb) * 64 + lane) * 16;
                for (int c = 0; c < 2; ++c)
                  for (int i = 0; i < 8; i += 2) {
                    unsigned hi2 = 0, lo2 = 0;
                    for (int e = 0; e < 2; ++e) {
                      const float w = src[(size_t)(32 * kb + 8 * g + i + e) * Np + 32 * cg + 2 * j + c];
                      const _Float16 hh = (_Float16)w;
                      hi2 |= (unsigned)half_bits((float)hh) << (16 * e);
                      lo2 |= (unsigned)half_bits(w - (float)hh) << (16 * e);
                    }
                    d[4 * c + i / 2] = hi2; d[8 + 4 * c + i / 2] = lo2;
                  }
              }
        }
      if (!st->packx) BGM_HIP_CHECK(hipMalloc((void **)&st->packx, sizeof(unsigned) * std::max<size_t>(st->packx_dwords, 1)));
      BGM_HIP_CHECK(hipMemcpy(st->packx, px.data(), sizeof(unsigned) * px.size(), hipMemcpyHostToDevice));
    }
    h->gx_valid = true;
  }
  return BGM_OK;
}

int grid_for(const bgm_handle *h, const GxState *s, int64_t n) {
  const int64_t tiles = (n + GX_ROWS - 1) / GX_ROWS;
  return (int)std::max<int64_t>(1, std::min<int64_t>(tiles, (int64_t)h->n_cus * s->occ));
}

// the model as the row-tile-per-wave kernels read it (their own dose batch) and their launch grid (workgroups of GW_WAVES row tiles)
GxCausalModel gw_model(const GxState *s) { GxCausalModel w = s->m; w.db = s->gw_db; w.pack = s->packF; w.packx = s->packx; w.ld = s->gw_ld; w.ldf = s->gw_ldf; return w; }
// Split precision (bgm_causal_set_precision) on this engine: "f16x3" on the row-tile-per-wave kernels.  0: fp32; 4 / 2: split (kernel variant, gw_kernels.h); < 0: refused.
int gx_x3(const bgm_handle *h, const GxState *s, const char *who) {
  if (h->precision == 0) return 0;
  if (h->precision == 2 && s->gw && !s->fit && s->x3_ok && s->packx) return s->x3_k64 ? 2 : 4;
  bgm_set_error(std::string(who) + ": split precision outside the default shapes exists as 'f16x3' for hidden widths up to 128 (the row-tile-per-wave "
                "kernels of the general-width engine), outside a fit session");
  return -1;
}
int gw_grid(const bgm_handle *h, const GxState *s, int64_t n) {
  const int64_t wgs = ((n + GW_ROWS - 1) / GW_ROWS + GW_WAVES - 1) / GW_WAVES;
  return (int)std::max<int64_t>(1, std::min<int64_t>(wgs, (int64_t)h->n_cus * s->gw_occ));
}

// (inside an open gx fit session the padded pack is kept current on the device by the Adam kernel; the fragment pack is host-built)
bool use_gw(const GxState *s) { return s->gw && !s->fit; }

bool default_units(const int32_t *u, int n, bool fh) {
  if (fh) return n == 3 && u[0] == 64 && u[1] == 32 && u[2] == 8;
  for (int i = 0; i < n; ++i) if (u[i] != 64) return false;
  return true;
}

}  // namespace

bool gx_wanted(const bgm_handle *h) {
  if (force_gx()) return true;
  const bgm_causal_config &c = h->cfg;
  if (!(default_units(c.g_units, c.n_hidden_g, false) && default_units(c.f_units, c.n_hidden_f, true) && default_units(c.h_units, c.n_hidden_h, true))) return true;
  // default widths outside the LDS-resident shapes WITH a conditional latent prior (IdentifiableCausalBGM): the streamed-fragment
  // kernels (bnf_det_api.hip) carry no prior table, this engine does
  return h->prior_seg != nullptr && bnf_det_wanted(h);
}
bool gx_enc_wanted(const bgm_handle *h) {
  if (force_gx()) return true;
  const bgm_causal_config &c = h->cfg;
  return !default_units(c.e_units, c.n_hidden_e, false) || h->p > 208 || h->q > 32;
}

bool gx_row_tile_per_wave(bgm_handle *h) {
  GxState *s = gst(h);
  if (!s) { gx_slots(h, 16); s = gst(h); }
  return s && use_gw(s);
}

int gx_slots(bgm_handle *h, int64_t n) {
  GxState *s = gst(h);
  if (!s) {      // plan without touching the device contents: the slot count only needs the LDS budget
    hipStream_t st = nullptr;
    if (gx_session(h, s, st, false) != BGM_OK) return h->n_cus;
    h->gx_valid = false;
  }
  if (use_gw(s)) return gw_grid(h, s, n) * GW_WAVES;      // one slot per wave
  return grid_for(h, s, n);
}

int gx_logpost(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, float *out, hipStream_t stream) {
  GxState *s;
  int rc = gx_session(h, s, stream);
  if (rc) return rc;
  const int x3 = gx_x3(h, s, "bgm_causal_logpost");
  if (x3 < 0) return BGM_E_UNSUPPORTED;
  if (use_gw(s)) {
    auto kern = x3 == 4 ? gw_causal_logpost_kernel<4> : x3 == 2 ? gw_causal_logpost_kernel<2> : gw_causal_logpost_kernel<0>;
    rc = set_lds(kern, s->gw_lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(gw_grid(h, s, n)), dim3(GW_THREADS), s->gw_lds, stream, gw_model(s), x, y, v, z, (long long)n, out);
    BGM_HIP_CHECK(hipGetLastError());
    return BGM_OK;
  }
  rc = set_lds(gx_causal_logpost_kernel, s->lds_bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(gx_causal_logpost_kernel, dim3(grid_for(h, s, n)), dim3(GX_THREADS), s->lds_bytes, stream, s->m, x, y, v, z, (long long)n, out);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

int gx_mh_run(bgm_handle *h, const bgm_mh_args *a, hipStream_t stream) {
  GxState *s;
  int rc = gx_session(h, s, stream);
  if (rc) return rc;
  GxMhArgs k{};
  k.m = s->m; k.x = a->x_dev; k.y = a->y_dev; k.v = a->v_dev; k.n = a->n; k.row_base = a->row_base;
  k.state = a->state_dev; k.logp = a->logp_dev; k.init = a->init; k.it_begin = a->it_begin; k.n_iters = a->n_iters; k.burn_in = a->burn_in;
  k.q_sd = a->q_sd; k.k0 = (unsigned)(a->seed & 0xFFFFFFFFull); k.k1 = (unsigned)(a->seed >> 32);
  k.acc_count = a->acc_count_dev; k.draws = a->draws_dev;
  k.e.n_keep = a->n_keep; k.e.sample_y = a->sample_y; k.e.n_doses = a->n_doses; k.e.x_values = a->x_values_dev; k.e.adrf_slot = nullptr;
  k.e.ite = a->ite_dev; k.e.k0 = k.k0; k.e.k1 = k.k1;
  k.adrf_partial = a->adrf_partial_dev;
  const bool gw = use_gw(s);
  const int x3 = gx_x3(h, s, "bgm_causal_mh_run");
  if (x3 < 0) return BGM_E_UNSUPPORTED;
  if (gw) k.m = gw_model(s);
  const int grid = gw ? gw_grid(h, s, a->n) : grid_for(h, s, a->n);
  const int it_end = a->it_begin + a->n_iters;
  if (a->effect != BGM_EFFECT_NONE && it_end > a->burn_in) {      // outcome-net cache of the retained iterations (bgm_causal_set_outcome_cache)
    const size_t need = (size_t)grid * (size_t)(a->effect == BGM_EFFECT_ITE ? 2 : a->n_doses) * (gw ? GW_WAVES * GW_ROWS : GX_ROWS) * 2;
    if (h->eff_cache_cap < need) {
      if (h->eff_cache) BGM_HIP_CHECK(hipFree(h->eff_cache));
      BGM_HIP_CHECK(hipMalloc(&h->eff_cache, need * sizeof(float)));
      h->eff_cache_cap = need;
    }
    if (!h->eff_stats_dev) {
      BGM_HIP_CHECK(hipMalloc(&h->eff_stats_dev, 2 * sizeof(unsigned long long)));
      BGM_HIP_CHECK(hipMemsetAsync(h->eff_stats_dev, 0, 2 * sizeof(unsigned long long), stream));
    }
    k.e.cache = reinterpret_cast<float2 *>(h->eff_cache);
    k.e.eff_skip = h->outcome_cache ? 1 : 0;
    k.e.stats = h->eff_stats_dev;
    h->eff_total += (unsigned long long)((a->n + 15) / 16) * (unsigned long long)(it_end - std::max(a->burn_in, a->it_begin));
  }
  const int lds_mh = gw ? s->gw_lds : s->lds_bytes, threads_mh = gw ? GW_THREADS : GX_THREADS;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->timing) { BGM_HIP_CHECK(hipEventCreate(&e0)); BGM_HIP_CHECK(hipEventCreate(&e1)); BGM_HIP_CHECK(hipEventRecord(e0, stream)); }
  auto launch = [&](auto kern) {
    int r = set_lds(kern, lds_mh);
    if (r) return r;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads_mh), lds_mh, stream, k);
    BGM_HIP_CHECK(hipGetLastError());
    return (int)BGM_OK;
  };
  if (gw && x3 == 4) {
    if (a->effect == BGM_EFFECT_ADRF) rc = launch(gw_causal_mh_kernel<1, 4>);
    else if (a->effect == BGM_EFFECT_ITE) rc = launch(gw_causal_mh_kernel<2, 4>);
    else rc = launch(gw_causal_mh_kernel<0, 4>);
  } else if (gw && x3 == 2) {
    if (a->effect == BGM_EFFECT_ADRF) rc = launch(gw_causal_mh_kernel<1, 2>);
    else if (a->effect == BGM_EFFECT_ITE) rc = launch(gw_causal_mh_kernel<2, 2>);
    else rc = launch(gw_causal_mh_kernel<0, 2>);
  } else if (gw) {
    if (a->effect == BGM_EFFECT_ADRF) rc = launch(gw_causal_mh_kernel<1>);
    else if (a->effect == BGM_EFFECT_ITE) rc = launch(gw_causal_mh_kernel<2>);
    else rc = launch(gw_causal_mh_kernel<0>);
  } else if (a->effect == BGM_EFFECT_ADRF) rc = launch(gx_causal_mh_kernel<1>);
  else if (a->effect == BGM_EFFECT_ITE) rc = launch(gx_causal_mh_kernel<2>);
  else rc = launch(gx_causal_mh_kernel<0>);
  if (rc) return rc;
  if (h->timing) { BGM_HIP_CHECK(hipEventRecord(e1, stream)); h->events.push_back({e0, e1, a->effect}); }
#ifdef GX_PHASE_CLOCK
  {
    unsigned long long c[16], z[16] = {0};
    hipStreamSynchronize(stream);
    hipMemcpyFromSymbol(c, HIP_SYMBOL(gx_phase_clk), sizeof(c));
    hipMemcpyToSymbol(HIP_SYMBOL(gx_phase_clk), z, sizeof(z));
    static const char *nm[9] = {"proposal", "stage-g", "g-hidden", "g-last", "f", "h", "assemble", "accept", "effects"};
    double tot = 0; for (int i = 0; i < 9; ++i) tot += (double)c[i];
    fprintf(stderr, "[GX_PHASE_CLOCK] cycles per workgroup and iteration (grid %d, %d iterations):", grid, a->n_iters);
    for (int i = 0; i < 9; ++i) fprintf(stderr, " %s %.0f (%.1f%%)", nm[i], (double)c[i] / grid / std::max(1, a->n_iters), 100.0 * c[i] / std::max(1.0, tot));
    fprintf(stderr, "\n");
  }
#endif
  return BGM_OK;
}

int gx_evaluate(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, const float *x_values,
                int32_t n_doses, double *sums, float *adrf_partial, float *ite, hipStream_t stream) {
  GxState *s;
  int rc = gx_session(h, s, stream);
  if (rc) return rc;
  GxEvalArgs k{};
  k.m = s->m; k.m.prior_seg = nullptr; k.m.prior_tab = nullptr;
  k.x = x; k.y = y; k.v = v; k.z = z; k.n = n; k.x_values = x_values; k.n_doses = n_doses; k.sums = sums; k.adrf_partial = adrf_partial; k.ite = ite;
  rc = set_lds(gx_causal_eval_kernel, s->lds_bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(gx_causal_eval_kernel, dim3(grid_for(h, s, n)), dim3(GX_THREADS), s->lds_bytes, stream, k);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

int gx_effects(bgm_handle *h, const float *draws, int64_t n, int64_t row_base, int32_t n_keep, int32_t burn_in, uint64_t seed,
               int32_t sample_y, const float *x_values, int32_t n_doses, float *adrf_partial, float *ite, hipStream_t stream) {
  GxState *s;
  int rc = gx_session(h, s, stream);
  if (rc) return rc;
  GxEffKArgs k{};
  k.m = s->m; k.draws = draws; k.n = n; k.row_base = row_base; k.burn_in = burn_in;
  k.e.n_keep = n_keep; k.e.sample_y = sample_y; k.e.n_doses = n_doses; k.e.x_values = x_values; k.e.ite = ite;
  k.e.k0 = (unsigned)(seed & 0xFFFFFFFFull); k.e.k1 = (unsigned)(seed >> 32);
  k.adrf_partial = adrf_partial;
  const bool binary = h->cfg.binary_treatment != 0;
  const int x3 = (h->precision == 2 && s->gw && !s->fit && s->x3_ok && s->packx) ? (s->x3_k64 ? 2 : 4) : 0;      // (stand-alone effects: split precision where it exists, else fp32 as on the resident kernels)
  if (use_gw(s)) {
    k.m = gw_model(s);
    const int gg = gw_grid(h, s, n);
    auto go = [&](auto kern) -> int {
      int r = set_lds(kern, s->gw_lds);
      if (r) return r;
      hipLaunchKernelGGL(kern, dim3(gg), dim3(GW_THREADS), s->gw_lds, stream, k);
      return BGM_OK;
    };
    if (binary) rc = x3 == 4 ? go(gw_causal_effects_kernel<2, 4>) : x3 == 2 ? go(gw_causal_effects_kernel<2, 2>) : go(gw_causal_effects_kernel<2>);
    else rc = x3 == 4 ? go(gw_causal_effects_kernel<1, 4>) : x3 == 2 ? go(gw_causal_effects_kernel<1, 2>) : go(gw_causal_effects_kernel<1>);
    if (rc) return rc;
    BGM_HIP_CHECK(hipGetLastError());
    return BGM_OK;
  }
  const int grid = grid_for(h, s, n);
  if (binary) {
    rc = set_lds(gx_causal_effects_kernel<2>, s->lds_bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(gx_causal_effects_kernel<2>, dim3(grid), dim3(GX_THREADS), s->lds_bytes, stream, k);
  } else {
    rc = set_lds(gx_causal_effects_kernel<1>, s->lds_bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(gx_causal_effects_kernel<1>, dim3(grid), dim3(GX_THREADS), s->lds_bytes, stream, k);
  }
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

int gx_encode(bgm_handle *h, const float *v, int64_t n, float *z, hipStream_t stream) {
  if (!h->nets[BGM_NET_E].set) { bgm_set_error("bgm_causal_encode: encoder weights not set"); return BGM_E_STATE; }
  GxState *s;
  int rc = gx_session(h, s, stream, false);
  if (rc) return rc;
  GxEncArgs k{};
  k.e = s->m.e; k.pack = s->pack; k.p = h->p; k.q = h->q; k.ld = s->ld_enc; k.kc = s->kc; k.v = v; k.n = n; k.z = z;
  rc = set_lds(gx_encode_kernel, s->lds_enc);
  if (rc) return rc;
  const int64_t tiles = (n + GX_ROWS - 1) / GX_ROWS;
  const int occ = std::max(1, std::min(4, (160 * 1024) / s->lds_enc));
  hipLaunchKernelGGL(gx_encode_kernel, dim3((int)std::max<int64_t>(1, std::min<int64_t>(tiles, (int64_t)h->n_cus * occ))), dim3(GX_THREADS), s->lds_enc, stream, k);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// fit session
// ---------------------------------------------------------------------------------------------------------------------------
bool gx_fit_active(const bgm_handle *h) { return gst(h) && gst(h)->fit; }
float *gx_pack(bgm_handle *h) { return gst(h) ? gst(h)->pack : nullptr; }
float *gx_packT(bgm_handle *h) { return gst(h) ? gst(h)->packT : nullptr; }
void gx_fit_end(bgm_handle *h) { if (gst(h)) gst(h)->fit = false; }

// Called by bgm_causal_fit_begin once theta_dev / m1_dev / m2_dev (g | f | h, Keras order) are on the device.  Fills the handle's
// workspace, weight-gradient layer list (fit_dw_kernel), gradient source table and the Adam scatter tables for the padded packs.
int gx_fit_begin(bgm_handle *h, int64_t n_rows, int32_t max_batch, hipStream_t stream) {
  GxState *s;
  int rc = gx_session(h, s, stream);
  if (rc) return rc;
  const HostNet &G = h->nets[BGM_NET_G], &F = h->nets[BGM_NET_F], &H = h->nets[BGM_NET_H];
  const int ng = (int)G.count(), nf = (int)F.count(), np = h->n_params;
  const int B = (max_batch + GX_ROWS - 1) / GX_ROWS * GX_ROWS;
  const GxCausalModel &m = s->m;
  // ---- workspace: per net and layer the layer input and the pre-activation gradient, [B][padded width] each
  long long off = 0;
  auto take = [&](long long n) { long long o = off; off += (n + 31) / 32 * 32; return o; };
  DwArgs &dw = h->dw;
  std::memset(&dw, 0, sizeof(dw));
  int nl = 0, poff = 0;
  std::vector<int> tables(4 * (size_t)np, -1);
  int *fwd_dst = tables.data(), *bwd_dst = fwd_dst + 2 * (size_t)np, *grad_src = bwd_dst + np;
  const GxNet *gn[3] = {&m.g, &m.f, &m.h};
  GxFitNet *fw[3] = {&s->wg, &s->wf, &s->wh};
  const HostNet *hn[3] = {&G, &F, &H};
  const int base[3] = {0, ng, ng + nf};
  for (int k = 0; k < 3; ++k) {
    const GxNet &n = *gn[k];
    if (nl + n.L > BGM_MAX_DW_LAYERS) { bgm_set_error("bgm_causal_fit_begin: too many layers"); return BGM_E_UNSUPPORTED; }
    size_t c = base[k];
    for (int l = 0; l < n.L; ++l) {
      fw[k]->act[l] = take((long long)B * n.pad[l]);
      fw[k]->dy[l] = take((long long)B * n.pad[l + 1]);
      DwLayer &L = dw.layer[nl++];
      L.a_off = fw[k]->act[l]; L.d_off = fw[k]->dy[l]; L.K = n.pad[l]; L.N = n.pad[l + 1]; L.out_off = poff;
      for (int i = 0; i < n.dim[l]; ++i)
        for (int o = 0; o < n.dim[l + 1]; ++o) grad_src[c++] = poff + i * L.N + o;
      for (int o = 0; o < n.dim[l + 1]; ++o) grad_src[c++] = poff + L.K * L.N + o;
      poff += L.K * L.N + L.N;
    }
    for (size_t i = 0; i < hn[k]->count(); ++i) { fwd_dst[base[k] + i] = s->fwd_map[k][i]; bwd_dst[base[k] + i] = s->bwd_map[k][i]; }
  }
  dw.n_layers = nl;
  dw.partial_stride = (poff + 3) / 4 * 4;
  std::memset(&h->fit_ws, 0, sizeof(h->fit_ws));
  h->fit_ws.B = B;
  h->fit_ws.dz = take((long long)B * h->q);
  s->fit_dzp = take(3LL * B * h->q);                 // the three networks' shares of d loss / dz (gx_causal_fit_kernel)
  s->fit_cnt = take(B / GX_ROWS + 1);                // arrival counters per 32-row tile (zeroed with the workspace, self-resetting)
  s->fit_dzp_rows = B;
  h->fit_ws.total = off;
  h->fit_bcap = B;
  h->rows_per_slice = 256;
  h->n_slices_cap = (B + h->rows_per_slice - 1) / h->rows_per_slice;
  BGM_HIP_CHECK(hipMalloc(&h->ws_dev, sizeof(float) * off));
  BGM_HIP_CHECK(hipMemset(h->ws_dev, 0, sizeof(float) * off));
  BGM_HIP_CHECK(hipMalloc(&h->partial_dev, sizeof(float) * dw.partial_stride * h->n_slices_cap));
  BGM_HIP_CHECK(hipMalloc(&h->tables_dev, sizeof(int) * tables.size()));
  BGM_HIP_CHECK(hipMemcpy(h->tables_dev, tables.data(), sizeof(int) * tables.size(), hipMemcpyHostToDevice));
  h->fit_rows = n_rows;
  BGM_HIP_CHECK(hipMalloc(&h->pos_dev, sizeof(int) * 2 * n_rows));
  BGM_HIP_CHECK(hipMemset(h->pos_dev, 0xFF, sizeof(int) * 2 * n_rows));
  BGM_HIP_CHECK(hipDeviceSynchronize());
  s->fit = true;
  return BGM_OK;
}

// forward + backward of one local minibatch: z_mode 0 leaves every layer's input and pre-activation gradient in the workspace (the
// caller runs fit_dw_kernel / fit_grad_reduce_kernel on them), z_mode 1 leaves d loss / d z [batch x q] at ws + fit_ws.dz.
int gx_fit_grads(bgm_handle *h, const float *x, const float *y, const float *v, const float *data_z, const int32_t *idx, int64_t row_lo,
                 int32_t batch, int32_t batch_global, int z_mode, float *grad, double *loss, hipStream_t stream) {
  (void)grad;
  GxState *s = gst(h);
  if (!s || !s->fit) { bgm_set_error("general-width engine: no fit session"); return BGM_E_STATE; }
  GxFitArgs a{};
  a.m = s->m; a.m.prior_seg = nullptr; a.m.prior_tab = nullptr;
  a.packT = s->packT; a.wg = s->wg; a.wf = s->wf; a.wh = s->wh; a.ws = h->ws_dev; a.dz_off = h->fit_ws.dz;
  a.dzp_off = s->fit_dzp; a.cnt_off = s->fit_cnt; a.dzp_rows = s->fit_dzp_rows;
  a.x = x; a.y = y; a.v = v; a.data_z = data_z; a.idx = idx; a.row_lo = row_lo; a.B = batch; a.inv_B = 1.0f / (float)batch_global;
  a.z_mode = z_mode; a.loss = loss;
  int rc = set_lds(gx_causal_fit_kernel, s->lds_fit);
  if (rc) return rc;
  const int tiles = (batch + GX_ROWS - 1) / GX_ROWS;
  hipLaunchKernelGGL(gx_causal_fit_kernel, dim3(std::min(tiles, h->n_cus * 2), 3), dim3(GX_THREADS), s->lds_fit, stream, a);      // y: g, f, h
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

void gx_free(bgm_handle *h) {
  GxState *s = gst(h);
  if (!s) return;
  if (s->pack) hipFree(s->pack);
  if (s->packT) hipFree(s->packT);
  if (s->packF) hipFree(s->packF);
  if (s->packx) hipFree(s->packx);
  delete s;
  h->gx_state = nullptr; h->gx_valid = false;
}
