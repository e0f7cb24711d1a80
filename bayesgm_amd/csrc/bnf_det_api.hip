// bnf_det_api.hip -- the general-shape sampling path of the DETERMINISTIC CausalBGM (params['use_bnn'] = False): the persistent
// streamed-fragment kernels of bnf_kernels.h instantiated without the Flipout half (DET), entered from bgm_causal_logpost /
// bgm_causal_mh_run / bgm_causal_effects / bgm_causal_evaluate (causal_api.hip) whenever no LDS-resident compiled shape of
// causal_kernels.h contains the model: sum(z_dims) up to 31 at any data width -- the weights of g's last layer [64 x (p + 1)] are
// streamed from L2 one output tile ahead when they do not fit the LDS next to the rest (p > 207: "wide").
//   replaces: get_log_posterior causalbgm/base.py:765-817, metropolis_hastings_sampler :820-904, infer_from_latent_posterior :671-763,
//   evaluate :534-570 for the shapes cli/cli.py:40-60 and configs/*.yaml admit and the resident kernels do not.
// Hidden widths stay the reference defaults (g [64] x 5, f / h [64, 32, 8]); BGM_FORCE_GENERAL=1 sends every shape here (tests).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bgm_host.h"
#include "bnf_build.h"
#include "bnf_det_host.h"
#include "bnf_kernels.h"

namespace {

constexpr int DR = 2, DW_ = 8;      // row tiles per wave, waves per workgroup (162 VGPRs, no spills)
using MhFn = void (*)(BnfMhArgs);
using EffFn = void (*)(BnfEffArgs);

template <int KS, int MODE>
MhFn det_fn_ks(bool wide) { return wide ? bnf_mh_kernel<KS, DR, DW_, MODE, true, true> : bnf_mh_kernel<KS, DR, DW_, MODE, true, false>; }
template <int MODE>
MhFn det_fn(int KS, bool wide) {
  switch (KS) {
    case 3: return det_fn_ks<3, MODE>(wide);
    case 4: return det_fn_ks<4, MODE>(wide);
    case 5: return det_fn_ks<5, MODE>(wide);
    case 6: return det_fn_ks<6, MODE>(wide);
    default: return det_fn_ks<8, MODE>(wide);
  }
}
EffFn det_eff_fn(int KSF) {
  switch (KSF) {
    case 1: return bnf_effects_kernel<1, DR, DW_, true>;
    case 2: return bnf_effects_kernel<2, DR, DW_, true>;
    case 3: return bnf_effects_kernel<3, DR, DW_, true>;
    case 4: return bnf_effects_kernel<4, DR, DW_, true>;
    default: return bnf_effects_kernel<8, DR, DW_, true>;
  }
}

template <class K>
int set_lds(K kernel, int bytes) {
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return BGM_OK;
}

BnfNetSrc src_of(const HostNet &n, int base, int net_id) {
  BnfNetSrc s;
  s.n_layers = (int)n.dims.size() - 1; s.net_id = net_id;
  for (size_t l = 0; l < n.dims.size() && l < 9; ++l) s.dims[l] = n.dims[l];
  for (int l = 0; l < s.n_layers && l < 8; ++l) {
    s.woff[l] = base + (int)(n.W(l) - n.theta.data()); s.roff[l] = -1; s.boff[l] = base + (int)(n.b(l) - n.theta.data());
  }
  s.gamma = s.beta = -2;
  return s;
}

int grid_for(const bgm_handle *h) { return std::max(8, (h->n_cus / 8) * 8); }

// session: tables + packed blobs of the handle's current weights; 1 when the model is outside this path as well
int det_session(bgm_handle *h, BnfState *&st, hipStream_t stream) {
  const HostNet &G = h->nets[BGM_NET_G], &F = h->nets[BGM_NET_F], &H = h->nets[BGM_NET_H];
  if (!G.set || !F.set || !H.set) { bgm_set_error("weights of g, f, h must be set (bgm_causal_set_weights)"); return BGM_E_STATE; }
  st = static_cast<BnfState *>(h->det_state);
  const int ng = (int)G.count(), nf = (int)F.count(), nh = (int)H.count();
  if (!st) {
    if (G.dims.size() > 9 || F.dims.size() > 9 || H.dims.size() > 9) return 1;
    BnfSrc S;
    S.g = src_of(G, 0, 0); S.f = src_of(F, ng, 2); S.h = src_of(H, ng + nf, 3);
    S.q = h->q; S.p = h->p; S.z0 = h->cfg.z_dims[0]; S.z1 = h->cfg.z_dims[1]; S.z2 = h->cfg.z_dims[2]; S.binary = h->cfg.binary_treatment ? 1 : 0;
    S.det = true;
    BnfState *n = new BnfState();
    BnfTabs tb;
    if (!bnf_build(S, *n, tb)) { delete n; return 1; }
    if (!bnf_upload(n, tb) || hipMalloc((void **)&n->theta_dev, sizeof(float) * (ng + nf + nh)) != hipSuccess) {
      bnf_release(n); bgm_set_error("general sampling path: device allocation failed"); return BGM_E_HIP;
    }
    h->det_state = n; h->det_valid = false;
    st = n;
  }
  if (!h->det_valid && h->fit_active && h->theta_dev && h->n_params == ng + nf + nh) {
    // a fit session is open: the parameters live on the device in the same g | f | h order (bgm_causal_fit_begin)
    BGM_HIP_CHECK(hipMemcpyAsync(st->theta_dev, h->theta_dev, sizeof(float) * (ng + nf + nh), hipMemcpyDeviceToDevice, stream));
    int rc = bnf_pack(st, st->theta_dev, stream);
    if (rc) return rc;
    h->det_valid = true;
  }
  if (!h->det_valid) {
    std::vector<float> theta((size_t)ng + nf + nh);
    std::copy(G.theta.begin(), G.theta.end(), theta.begin());
    std::copy(F.theta.begin(), F.theta.end(), theta.begin() + ng);
    std::copy(H.theta.begin(), H.theta.end(), theta.begin() + ng + nf);
    BGM_HIP_CHECK(hipStreamSynchronize(stream));
    BGM_HIP_CHECK(hipMemcpy(st->theta_dev, theta.data(), sizeof(float) * theta.size(), hipMemcpyHostToDevice));
    int rc = bnf_pack(st, st->theta_dev, stream);
    if (rc) return rc;
    h->det_valid = true;
  }
  return BGM_OK;
}

void fill_common(const bgm_handle *h, const BnfState *st, BnfMhArgs &a, const float *x, const float *y, const float *v, float *z, long long n) {
  a.pl = st->P; a.blob = st->blob_dev; a.dw = nullptr;
  a.x = x; a.y = y; a.v = v; a.z = z; a.n = n;
  a.bs = (int)n; a.n_blocks = 1; a.block0 = 0;
  a.groups_per_block = (int)(((n + 15) / 16 + DR - 1) / DR); a.n_items = a.groups_per_block; a.n_states = 1;
  a.queue = st->queue_dev;
  auto s2 = [](float s) { return s > 0.0f ? s * s : -1.0f; };
  a.sig2_v = s2(h->cfg.sigma_v); a.sig2_x = s2(h->cfg.sigma_x); a.sig2_y = s2(h->cfg.sigma_y);
}

int launch_mh(bgm_handle *h, BnfState *st, MhFn fn, const BnfMhArgs &a, hipStream_t stream) {
  BGM_HIP_CHECK(hipMemsetAsync(st->queue_dev, 0, 8 * sizeof(unsigned), stream));
  hipLaunchKernelGGL(fn, dim3(grid_for(h)), dim3(64 * DW_), st->lds_mh, stream, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

// the outcome net at the doses on one state per row -> per-workgroup float partial sums (adrf) or per-row differences (ite)
int launch_effects(bgm_handle *h, BnfState *st, const float *z, long long n, long long row_base, int n_doses, const float *xvals, uint64_t seed,
                   uint32_t it_noise, int sample_y, float *fsum, long long slot_stride, long long dose_stride, float *ite, long long ite_stride,
                   hipStream_t stream) {
  BnfEffArgs ea{};
  ea.pl = st->P; ea.eblob = st->eblob_dev; ea.dw = nullptr; ea.sgf = nullptr; ea.z = z; ea.n = n; ea.row_base = row_base;
  ea.bs = (int)n; ea.n_blocks = 1; ea.block0 = 0;
  ea.groups_per_block = (int)(((n + 15) / 16 + DR - 1) / DR); ea.n_items = ea.groups_per_block; ea.n_doses = n_doses;
  ea.xvals = xvals; ea.k0 = (uint32_t)seed; ea.k1 = (uint32_t)(seed >> 32); ea.sample_y = sample_y; ea.it_noise = it_noise;
  ea.sum_out = nullptr; ea.sum_stride = dose_stride; ea.fsum_out = fsum; ea.fsum_slot_stride = slot_stride;
  ea.ite_out = ite; ea.ite_stride = ite_stride; ea.queue = st->queue_dev + 8;
  const float sy = h->cfg.sigma_y;
  ea.sig2_y = sy > 0.0f ? sy * sy : -1.0f;
  EffFn fn = det_eff_fn(st->KSFc);
  int rc = set_lds(fn, st->lds_eff);
  if (rc) return rc;
  BGM_HIP_CHECK(hipMemsetAsync(st->queue_dev + 8, 0, 8 * sizeof(unsigned), stream));
  hipLaunchKernelGGL(fn, dim3(grid_for(h)), dim3(64 * DW_), st->lds_eff, stream, ea);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

}  // namespace

void bnf_det_free(bgm_handle *h) {
  bnf_release(static_cast<BnfState *>(h->det_state));
  h->det_state = nullptr;
  h->det_valid = false;
}

// true when the model has no LDS-resident compiled shape (or the general path is forced) -- the caller then takes this path
bool bnf_det_wanted(const bgm_handle *h) {
  static const bool force = std::getenv("BGM_FORCE_GENERAL") != nullptr;
  if (force) return true;
  int KT1, KSL1, NTL;
  if (!bgm_causal_shape(h->q + 1, h->p + 1, KT1, KSL1, NTL)) return true;
  return false;
}

int bnf_det_slots(bgm_handle *h) { return grid_for(h); }

int bnf_det_logpost(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, float *out, hipStream_t stream) {
  BnfState *st;
  int rc = det_session(h, st, stream);
  if (rc) { if (rc == 1) { bgm_set_error("no kernel for this model: hidden widths must be g [64] x 5, f / h [64, 32, 8]; sum(z_dims) <= 31"); return BGM_E_UNSUPPORTED; } return rc; }
  BnfMhArgs a{};
  fill_common(h, st, a, x, y, v, const_cast<float *>(z), n);
  a.mode = 0; a.out = out;
  MhFn fn = det_fn<0>(st->KSc, st->wide);
  rc = set_lds(fn, st->lds_mh);
  if (rc) return rc;
  return launch_mh(h, st, fn, a, stream);
}

int bnf_det_evaluate(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, const float *x_values,
                     int32_t n_doses, double *sums, float *adrf_partial, float *ite, hipStream_t stream) {
  BnfState *st;
  int rc = det_session(h, st, stream);
  if (rc) { if (rc == 1) { bgm_set_error("no kernel for this model: hidden widths must be g [64] x 5, f / h [64, 32, 8]; sum(z_dims) <= 31"); return BGM_E_UNSUPPORTED; } return rc; }
  const bool binary = h->cfg.binary_treatment != 0;
  if (!binary && n_doses > BNF_MAX_DOSES) { bgm_set_error("general sampling path: at most 256 doses per call"); return BGM_E_UNSUPPORTED; }
  BnfMhArgs a{};
  fill_common(h, st, a, x, y, v, const_cast<float *>(z), n);
  a.mode = 2; a.sums = sums;
  MhFn fn = det_fn<2>(st->KSc, st->wide);
  rc = set_lds(fn, st->lds_mh);
  if (rc) return rc;
  rc = launch_mh(h, st, fn, a, stream);
  if (rc) return rc;
  // plug-in effects: the outcome net at the doses, no outcome noise
  if (binary) return launch_effects(h, st, z, n, 0, 2, st->pair_dev, 0, 0, 0, nullptr, 0, 0, ite, 1, stream);
  return launch_effects(h, st, z, n, 0, n_doses, x_values, 0, 0, 0, adrf_partial, n_doses, 1, nullptr, 0, stream);
}

int bnf_det_effects(bgm_handle *h, const float *draws, int64_t n, int64_t row_base, int32_t n_keep, int32_t burn_in, uint64_t seed,
                    int32_t sample_y, const float *x_values, int32_t n_doses, float *adrf_partial, float *ite, hipStream_t stream) {
  BnfState *st;
  int rc = det_session(h, st, stream);
  if (rc) { if (rc == 1) { bgm_set_error("no kernel for this model: hidden widths must be g [64] x 5, f / h [64, 32, 8]; sum(z_dims) <= 31"); return BGM_E_UNSUPPORTED; } return rc; }
  const bool binary = h->cfg.binary_treatment != 0;
  const int nd = binary ? 2 : n_doses;
  if (nd > BNF_MAX_DOSES) { bgm_set_error("general sampling path: at most 256 doses per call"); return BGM_E_UNSUPPORTED; }
  const int q = h->q;
  for (int d = 0; d < n_keep; ++d) {
    const float *z = draws + (long long)d * n * q;
    // adrf_partial [n_slots][n_keep][n_doses]; ite [n][n_keep]
    rc = binary ? launch_effects(h, st, z, n, row_base, 2, st->pair_dev, seed, (uint32_t)(burn_in + d), sample_y, nullptr, 0, 0, ite + d, n_keep, stream)
                : launch_effects(h, st, z, n, row_base, nd, x_values, seed, (uint32_t)(burn_in + d), sample_y, adrf_partial + (long long)d * nd,
                                 (long long)nd * n_keep, 1, nullptr, 0, stream);
    if (rc) return rc;
  }
  return BGM_OK;
}

int bnf_det_mh_run(bgm_handle *h, const bgm_mh_args *g, hipStream_t stream) {
  BnfState *st;
  int rc = det_session(h, st, stream);
  if (rc) { if (rc == 1) { bgm_set_error("no kernel for this model: hidden widths must be g [64] x 5, f / h [64, 32, 8]; sum(z_dims) <= 31"); return BGM_E_UNSUPPORTED; } return rc; }
  const long long n = g->n;
  const int q = h->q;
  const int n_doses = g->effect == BGM_EFFECT_ADRF ? g->n_doses : (g->effect == BGM_EFFECT_ITE ? 2 : 0);
  if (n_doses > BNF_MAX_DOSES) { bgm_set_error("general sampling path: at most 256 doses per call"); return BGM_E_UNSUPPORTED; }
  BnfMhArgs a{};
  fill_common(h, st, a, g->x_dev, g->y_dev, g->v_dev, g->state_dev, n);
  a.row_base = g->row_base; a.mode = 1; a.q_sd = g->q_sd; a.q_sd_blocks = nullptr;
  a.k0 = (uint32_t)g->seed; a.k1 = (uint32_t)(g->seed >> 32); a.lp_cache = g->logp_dev;
  MhFn fn = det_fn<1>(st->KSc, st->wide);
  rc = set_lds(fn, st->lds_mh);
  if (rc) return rc;
  if (g->acc_count_dev) BGM_HIP_CHECK(hipMemsetAsync(g->acc_count_dev + g->it_begin, 0, sizeof(unsigned) * g->n_iters, stream));
  for (int i = 0; i < g->n_iters; ++i) {
    const int it = g->it_begin + i;
    a.it = it; a.init = (i == 0 && g->init) ? 1 : 0;
    a.acc_count = g->acc_count_dev ? g->acc_count_dev + it : nullptr;      // accepted proposals of iteration `it`
    rc = launch_mh(h, st, fn, a, stream);
    if (rc) return rc;
    const int d = it - g->burn_in;
    if (d >= 0 && d < g->n_keep) {
      if (g->draws_dev)
        BGM_HIP_CHECK(hipMemcpyAsync(g->draws_dev + (long long)d * n * q, g->state_dev, sizeof(float) * n * q, hipMemcpyDeviceToDevice, stream));
      if (g->effect == BGM_EFFECT_ADRF)
        rc = launch_effects(h, st, g->state_dev, n, g->row_base, n_doses, g->x_values_dev, g->seed, (uint32_t)it, g->sample_y,
                            g->adrf_partial_dev + (long long)d * n_doses, (long long)n_doses * g->n_keep, 1, nullptr, 0, stream);
      else if (g->effect == BGM_EFFECT_ITE)
        rc = launch_effects(h, st, g->state_dev, n, g->row_base, 2, st->pair_dev, g->seed, (uint32_t)it, g->sample_y, nullptr, 0, 0,
                            g->ite_dev + d, g->n_keep, stream);
      if (rc) return rc;
    }
  }
  return BGM_OK;
}
