// bnf_api.hip -- host side of the fixed-normalisation Bayesian-network sampling path (bnf_kernels.h): shape check, fragment
// index tables, buffers, launch sequences.  Entered from bgm_bnn_logpost / bgm_bnn_mh_run / bgm_bnn_effects (bnn_sample_api.hip)
// when the session's nets have the default shapes and params['bnn_norm'] = "fixed"; everything else keeps the batch-statistics
// kernels of bnn_sample_kernels.h.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bgm_host.h"
#include "bnf_host.h"
#include "bnf_build.h"
#include "bnf_kernels.h"
#include "bnn_state.h"
#include "bnx_host.h"

namespace {

BnfNetSrc src_of(const BnnNet &b) {
  BnfNetSrc n;
  n.n_layers = b.n_layers; n.net_id = b.net_id;
  for (int l = 0; l <= b.n_layers && l < 9; ++l) n.dims[l] = b.dims[l];
  for (int l = 0; l < b.n_layers && l < 8; ++l) {
    const int cnt = b.dims[l] * b.dims[l + 1];
    n.woff[l] = b.woff[l]; n.roff[l] = b.woff[l] + cnt; n.boff[l] = b.woff[l] + 2 * cnt;
  }
  n.gamma = b.off; n.beta = b.off + b.dims[0];
  return n;
}
// false when the session is outside this path (inference-mode normalisation, default widths, q <= 31, 4 <= p <= 207)
bool bnf_build_bnn(const BnnState *s, BnfState &st, BnfTabs &tb) {
  for (int k : {BNN_G, BNN_H, BNN_F}) {
    const BnnNet &b = s->net[k];
    if (b.bn_fixed != 1 || b.heads || b.mv || b.n_layers > 8) return false;
  }
  BnfSrc S;
  S.g = src_of(s->net[BNN_G]); S.h = src_of(s->net[BNN_H]); S.f = src_of(s->net[BNN_F]);
  S.q = s->q; S.p = s->p; S.z0 = s->cfg.z_dims[0]; S.z1 = s->cfg.z_dims[1]; S.z2 = s->cfg.z_dims[2]; S.binary = s->cfg.binary_treatment;
  S.det = false;
  return bnf_build(S, st, tb);
}

// ---- kernel variants: (R row tiles per wave, WAVES per workgroup).  The shipped one is (BNF_R, BNF_W); a development build
// (-D BNF_ALL_CFGS) also carries the others for the bench shape (KS = 3) and picks by the environment variable BGM_BNF_CFG=r<R>w<W>.
#ifndef BNF_R
#define BNF_R 2
#define BNF_W 8
#endif
#ifndef BNF_ER
#define BNF_ER 2
#define BNF_EW 8
#endif
struct BnfCfg { int R, W; };
BnfCfg env_cfg(const char *name, BnfCfg d) {
#ifdef BNF_ALL_CFGS
  if (const char *e = std::getenv(name)) {
    int r = 0, w = 0;
    if (std::sscanf(e, "r%dw%d", &r, &w) == 2) d = BnfCfg{r, w};
  }
#endif
  (void)name;
  return d;
}
using MhFn = void (*)(BnfMhArgs);
using EffFn = void (*)(BnfEffArgs);
template <int KS, int MODE>
MhFn mh_fn_ks(BnfCfg &c) {
#ifdef BNF_ALL_CFGS
  if constexpr (KS == 3) {
    if (c.R == 1 && c.W == 8) return bnf_mh_kernel<KS, 1, 8, MODE>;
    if (c.R == 2 && c.W == 8) return bnf_mh_kernel<KS, 2, 8, MODE>;
    if (c.R == 2 && c.W == 4) return bnf_mh_kernel<KS, 2, 4, MODE>;
    if (c.R == 3 && c.W == 4) return bnf_mh_kernel<KS, 3, 4, MODE>;
    if (c.R == 1 && c.W == 12) return bnf_mh_kernel<KS, 1, 12, MODE>;
  }
#endif
  c = BnfCfg{BNF_R, BNF_W};
  return bnf_mh_kernel<KS, BNF_R, BNF_W, MODE>;
}
template <int KS>
EffFn eff_fn_ks(BnfCfg &c) {
#ifdef BNF_ALL_CFGS
  if constexpr (KS == 1) {
    if (c.R == 1 && c.W == 8) return bnf_effects_kernel<KS, 1, 8>;
    if (c.R == 2 && c.W == 8) return bnf_effects_kernel<KS, 2, 8>;
    if (c.R == 2 && c.W == 4) return bnf_effects_kernel<KS, 2, 4>;
    if (c.R == 4 && c.W == 4) return bnf_effects_kernel<KS, 4, 4>;
    if (c.R == 1 && c.W == 12) return bnf_effects_kernel<KS, 1, 12>;
    if (c.R == 1 && c.W == 16) return bnf_effects_kernel<KS, 1, 16>;
    if (c.R == 2 && c.W == 12) return bnf_effects_kernel<KS, 2, 12>;
  }
#endif
  c = BnfCfg{BNF_ER, BNF_EW};
  return bnf_effects_kernel<KS, BNF_ER, BNF_EW>;
}
// c: in = the requested variant, out = the one returned
template <int MODE>
MhFn mh_fn(int KS, BnfCfg &c) {
  c = env_cfg("BGM_BNF_CFG", BnfCfg{BNF_R, BNF_W});
  switch (KS) {
    case 3: return mh_fn_ks<3, MODE>(c);
    case 4: return mh_fn_ks<4, MODE>(c);
    case 5: return mh_fn_ks<5, MODE>(c);
    case 6: return mh_fn_ks<6, MODE>(c);
    default: return mh_fn_ks<8, MODE>(c);
  }
}
EffFn eff_fn(int KSF, BnfCfg &c) {
  c = env_cfg("BGM_BNF_ECFG", BnfCfg{BNF_ER, BNF_EW});
  switch (KSF) {
    case 1: return eff_fn_ks<1>(c);
    case 2: return eff_fn_ks<2>(c);
    case 3: return eff_fn_ks<3>(c);
    case 4: return eff_fn_ks<4>(c);
    default: return eff_fn_ks<8>(c);
  }
}

int bnf_session(bgm_handle *h, BnnState *s, BnfState *&st, hipStream_t stream) {
  st = static_cast<BnfState *>(s->bnf);
  if (!st) {
    if (std::getenv("BGM_BNF_OFF")) return 1;
    BnfState *n = new BnfState();
    BnfTabs tb;
    if (!bnf_build_bnn(s, *n, tb)) { delete n; s->bnf_unsupported = true; return 1; }
    if (!bnf_upload(n, tb)) { bnf_release(n); bgm_set_error("bnf: device allocation failed"); return BGM_E_HIP; }
    s->bnf = n; s->bnf_valid = false;
    st = n;
  }
  if (!s->bnf_valid) {
    int rc = bnf_pack(st, s->theta_dev, stream);
    if (rc) return rc;
    s->bnf_valid = true;
  }
  if (s->precision == 2) {      // split precision: the blobs in the fragment encoding of bnx_kernels.h
    int rc = bnx_prepare(st, stream);
    if (rc) return rc;
  }
  (void)h;
  return BGM_OK;
}

// zero: the padding positions of the perturbation sets must be zero and the noise kernel writes real elements only; the set
// boundaries move with (n_blocks, n_doses), so the used part is cleared at every call (tens of MB, once per call)
int grow(void **p, size_t *cap, size_t bytes, hipStream_t stream, bool zero) {
  if (bytes > *cap) {
    BGM_HIP_CHECK(hipStreamSynchronize(stream));
    if (*p) BGM_HIP_CHECK(hipFree(*p));
    *p = nullptr; *cap = 0;
    BGM_HIP_CHECK(hipMalloc(p, bytes));
    *cap = bytes;
  }
  if (zero && bytes) BGM_HIP_CHECK(hipMemsetAsync(*p, 0, bytes, stream));
  return BGM_OK;
}

struct SgLayout { uint4 *g; uint32_t *gout; uint4 *h, *f; size_t bytes; };
SgLayout sg_layout(void *base, long long n, int states_ghf, int states_f_only) {
  SgLayout L{};
  char *p = (char *)base;
  const size_t n16 = (size_t)n * 16;
  L.g = (uint4 *)p; p += (size_t)states_ghf * BNF_NG_G * n16;
  L.gout = (uint32_t *)p; p += (size_t)states_ghf * n * BNF_GOUT * 4;
  L.h = (uint4 *)p; p += (size_t)states_ghf * BNF_NG_H * n16;
  L.f = (uint4 *)p; p += (size_t)std::max(states_ghf, states_f_only) * BNF_NG_H * n16;
  L.bytes = (size_t)(p - (char *)base);
  return L;
}

template <class K>
int set_lds(K kernel, int bytes) {
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return BGM_OK;
}

// effects = true: sets of the outcome net in the effects layout
void launch_noise(const BnfState *st, bool effects, float *dw, long long set_floats, int n_blocks, int n_states, uint64_t seed, uint32_t stream0,
                  int block0, hipStream_t stream, bool x3) {
  BnfNoiseArgs na{};
  const int n_lay = effects ? 4 : 14;
  for (int i = 0; i < n_lay; ++i) na.lay[i] = effects ? st->lay_e[i] : st->lay[i];
  na.n_lay = n_lay; na.n_calls = effects ? st->n_calls_e : st->n_calls; na.npos = effects ? st->npos_e_dev : st->npos_dev;
  na.sf = effects ? st->esf_dev : st->sf_dev; na.dw = dw; na.set_floats = set_floats;
  na.n_states = n_states; na.k0 = (uint32_t)seed; na.k1 = (uint32_t)(seed >> 32); na.stream0 = stream0; na.block0 = block0;
  if (x3) bnx_launch_noise(na, effects ? st->posx_e_dev : st->posx_dev, n_blocks * n_states, stream);
  else hipLaunchKernelGGL(bnf_noise_kernel, dim3((na.n_calls + 1023) / 1024, n_blocks * n_states), dim3(256), 0, stream, na);
}

void launch_signs(const SgLayout &L, long long n, int bs, int block0, int n_states, int nets, uint64_t seed, uint32_t stream0, unsigned *queue,
                  hipStream_t stream, long long rib0 = 0) {
  BnfSignArgs sa{};
  sa.rib0 = rib0;
  sa.g = L.g; sa.gout = L.gout; sa.h = L.h; sa.f = L.f; sa.n = n; sa.bs = bs; sa.block0 = block0; sa.n_states = n_states; sa.nets = nets;
  sa.k0 = (uint32_t)seed; sa.k1 = (uint32_t)(seed >> 32); sa.stream0 = stream0; sa.queue = queue;
  hipLaunchKernelGGL(bnf_signs_kernel, dim3((unsigned)((n + 255) / 256), n_states), dim3(256), 0, stream, sa);
}

int grid_for(const bgm_handle *h) { return std::max(8, (h->n_cus / 8) * 8); }

}  // namespace

void bnf_free(void *state) { bnf_release(static_cast<BnfState *>(state)); }

extern "C" int bgm_bnn_set_precision(bgm_handle *h, int32_t mode) {
  if (!h || !h->bnn_state) { bgm_set_error("bgm_bnn_set_precision: no Bayesian-network session (bgm_bnn_begin)"); return BGM_E_STATE; }
  if (mode != 0 && mode != 2) {
    bgm_set_error("bgm_bnn_set_precision: mode must be 0 (fp32) or 2 (f16x3); the Bayesian sampling kernels have no bf16 form");
    return mode == 1 ? BGM_E_UNSUPPORTED : BGM_E_INVALID;
  }
  BnnState *s = static_cast<BnnState *>(h->bnn_state);
  if (mode == 2) {      // the split-precision kernels exist for the sessions bnf_kernels.h serves: say so now, not at the first sampling call
    BnfState probe; BnfTabs tb;
    if (s->bnf_unsupported || (!s->bnf && !bnf_build_bnn(s, probe, tb))) {
      bgm_set_error("bgm_bnn_set_precision: split precision exists on the default-shape sampling kernels with inference-mode input normalisation "
                    "(bnn_norm='fixed', g_units [64]x5, f_units / h_units [64,32,8], sum(z_dims) <= 31, v_dim <= 207)");
      return BGM_E_UNSUPPORTED;
    }
  }
  s->precision = mode;
  return BGM_OK;
}

// returns 1 when the session is outside this path (the caller continues with the batch-statistics kernels), 0 when handled,
// a negative BGM_E_* code on failure
int bnf_logpost(bgm_handle *h, BnnState *s, const float *x, const float *y, const float *v, const float *z, int64_t n, int32_t block_rows,
                int32_t block0, uint64_t seed, uint32_t stream_id, float *out, hipStream_t stream) {
  if (s->bnf_unsupported) return 1;
  BnfState *st;
  int rc = bnf_session(h, s, st, stream);
  if (rc) return rc;
  const BnfPlan &P = st->P;
  BnfCfg c;
  const bool x3 = s->precision == 2;
  MhFn fn = x3 ? bnx_mh_fn(st->KSc, 0, &c.R, &c.W) : mh_fn<0>(st->KSc, c);
  const int n_blocks = (int)((n + block_rows - 1) / block_rows);
  rc = grow((void **)&st->dw_dev, &st->dw_cap, sizeof(float) * (size_t)n_blocks * P.set_floats, stream, true);
  if (rc) return rc;
  SgLayout L = sg_layout(nullptr, n, 1, 0);
  rc = grow(&st->sg_dev, &st->sg_cap, L.bytes, stream, false);
  if (rc) return rc;
  L = sg_layout(st->sg_dev, n, 1, 0);
  launch_noise(st, false, st->dw_dev, P.set_floats, n_blocks, 1, seed, stream_id, block0, stream, x3);
  launch_signs(L, n, block_rows, block0, 1, 7, seed, stream_id, st->queue_dev, stream);
  BnfMhArgs a{};
  a.pl = P; a.blob = x3 ? st->blobx_dev : st->blob_dev; a.dw = st->dw_dev;
  a.sig2_v = s->cfg.sigma_v > 0.0f ? s->cfg.sigma_v * s->cfg.sigma_v : 0.0f;      // fixed params['sigma_*'] (0: the variance heads)
  a.sig2_x = s->cfg.sigma_x > 0.0f ? s->cfg.sigma_x * s->cfg.sigma_x : 0.0f;
  a.sig2_y = s->cfg.sigma_y > 0.0f ? s->cfg.sigma_y * s->cfg.sigma_y : 0.0f;
  a.sg.g = L.g; a.sg.gout = L.gout; a.sg.h = L.h; a.sg.f = L.f;
  a.x = x; a.y = y; a.v = v; a.z = const_cast<float *>(z); a.n = n; a.row_base = 0;
  a.bs = block_rows; a.n_blocks = n_blocks; a.block0 = block0;
  a.groups_per_block = ((block_rows + 15) / 16 + c.R - 1) / c.R; a.n_items = n_blocks * a.groups_per_block; a.n_states = 1;
  a.mode = 0; a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.out = out; a.queue = st->queue_dev;
  if (s->bp_on) {      // conditional prior: one noisy call of the prior net for this evaluation (bprior_api.hip)
    if ((rc = bprior_rows(h, s, n, block_rows, block0, seed, stream_id, 1, stream))) return rc;
    a.prior = s->bp_rows; a.prior_stride = 0;
  }
  rc = set_lds(fn, st->lds_mh);
  if (rc) return rc;
  hipLaunchKernelGGL(fn, dim3(grid_for(h)), dim3(64 * c.W), st->lds_mh, stream, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

namespace {
// effects of ONE draw (state z) -> adrf column / ite column d
int effects_of(bgm_handle *h, BnfState *st, const float *z, long long n, int bs, int n_blocks, int block0, long long row_base, int n_doses,
               const float *xvals, uint64_t seed, uint32_t stream0, uint32_t it_noise, int sample_y, double *sum_out, long long sum_stride,
               float *ite_out, long long ite_stride, float *dw_eff, const SgLayout &L, hipStream_t stream, bool x3, long long rib0 = 0) {
  const BnfPlan &P = st->P;
  BnfCfg c;
  EffFn fn = x3 ? bnx_eff_fn(st->KSFc, &c.R, &c.W) : eff_fn(st->KSFc, c);
  const long long eset = (long long)P.e_frags * 256;
  launch_noise(st, true, dw_eff, eset, n_blocks, n_doses, seed, stream0, block0, stream, x3);
  launch_signs(L, n, bs, block0, n_doses, 4, seed, stream0, st->queue_dev + 8, stream, rib0);
  BnfEffArgs ea{};
  ea.pl = P; ea.eblob = x3 ? st->eblobx_dev : st->eblob_dev; ea.dw = dw_eff; ea.sgf = L.f; ea.z = z; ea.n = n; ea.row_base = row_base;
  { const BnnState *bs_ = static_cast<const BnnState *>(h->bnn_state); ea.sig2_y = (bs_ && bs_->cfg.sigma_y > 0.0f) ? bs_->cfg.sigma_y * bs_->cfg.sigma_y : 0.0f; }
  ea.bs = bs; ea.n_blocks = n_blocks; ea.block0 = block0;
  ea.groups_per_block = ((bs + 15) / 16 + c.R - 1) / c.R; ea.n_items = n_blocks * ea.groups_per_block; ea.n_doses = n_doses;
  ea.xvals = xvals; ea.k0 = (uint32_t)seed; ea.k1 = (uint32_t)(seed >> 32); ea.sample_y = sample_y; ea.it_noise = it_noise;
  ea.sum_out = sum_out; ea.sum_stride = sum_stride; ea.ite_out = ite_out; ea.ite_stride = ite_stride; ea.queue = st->queue_dev + 8;
  int rc = set_lds(fn, st->lds_eff);
  if (rc) return rc;
  hipLaunchKernelGGL(fn, dim3(grid_for(h)), dim3(64 * c.W), st->lds_eff, stream, ea);
  return BGM_OK;
}
}  // namespace

int bnf_mh_run(bgm_handle *h, BnnState *s, const bgm_bnn_mh_args *g, hipStream_t stream) {
  if (s->bnf_unsupported) return 1;
  const int n_doses = g->effect == 1 ? g->n_doses : (g->effect == 2 ? 2 : 0);
  if (n_doses > BNF_MAX_DOSES) return 1;
  BnfState *st;
  int rc = bnf_session(h, s, st, stream);
  if (rc) return rc;
  const BnfPlan &P = st->P;
  BnfCfg c;
  const bool x3 = s->precision == 2;
  // three-launch iteration (proposal / both evaluations as (item, state) units / accept step): the split-precision path always, fp32 when
  // BGM_BNF_SPLIT=1 (development A/B)
  static const bool split32 = std::getenv("BGM_BNF_SPLIT") != nullptr;
  const bool split = x3 || split32;
  MhFn fn = x3 ? bnx_mh_fn(st->KSc, 1, &c.R, &c.W) : (split32 ? mh_fn<3>(st->KSc, c) : mh_fn<1>(st->KSc, c));
  const long long n = g->n;
  const int bs = g->block_rows, n_blocks = (int)((n + bs - 1) / bs), q = s->q;
  // A rank's share of ONE block (row_base inside the block block0): the Flipout sign words of the samplers and of the prior net are keyed
  // by (block, position in the block), so the position of the first row travels with the call -- results do not depend on how the block's
  // rows are split over ranks (bgm_bnn_mh_args::block_row0; 0 for calls that start a block).
  const long long rib0 = g->block_row0;
  if (rib0 < 0 || rib0 >= bs || (rib0 > 0 && rib0 + n > bs)) { bgm_set_error("bgm_bnn_mh_run: block_row0 must lie inside the block, and a call that starts inside a block must end inside it"); return BGM_E_INVALID; }
  const long long eset = (long long)P.e_frags * 256;
  const size_t dw_mh = (size_t)n_blocks * 2 * P.set_floats, dw_eff = (size_t)n_blocks * n_doses * eset;
  rc = grow((void **)&st->dw_dev, &st->dw_cap, sizeof(float) * (dw_mh + dw_eff), stream, true);
  if (rc) return rc;
  SgLayout L = sg_layout(nullptr, n, 2, n_doses);
  rc = grow(&st->sg_dev, &st->sg_cap, L.bytes, stream, false);
  if (rc) return rc;
  L = sg_layout(st->sg_dev, n, 2, n_doses);
  float *dw_eff_dev = st->dw_dev + dw_mh;
  BnfMhArgs a{};
  a.pl = P; a.blob = x3 ? st->blobx_dev : st->blob_dev; a.dw = st->dw_dev;
  a.sig2_v = s->cfg.sigma_v > 0.0f ? s->cfg.sigma_v * s->cfg.sigma_v : 0.0f;      // fixed params['sigma_*'] (0: the variance heads)
  a.sig2_x = s->cfg.sigma_x > 0.0f ? s->cfg.sigma_x * s->cfg.sigma_x : 0.0f;
  a.sig2_y = s->cfg.sigma_y > 0.0f ? s->cfg.sigma_y * s->cfg.sigma_y : 0.0f;
  a.sg.g = L.g; a.sg.gout = L.gout; a.sg.h = L.h; a.sg.f = L.f;
  a.x = g->x_dev; a.y = g->y_dev; a.v = g->v_dev; a.z = g->state_dev; a.n = n; a.row_base = g->row_base;
  a.bs = bs; a.n_blocks = n_blocks; a.block0 = g->block0;
  a.groups_per_block = ((bs + 15) / 16 + c.R - 1) / c.R; a.n_items = n_blocks * a.groups_per_block; a.n_states = 2;
  a.mode = 1; a.q_sd = g->q_sd; a.q_sd_blocks = g->q_sd_blocks_dev;
  a.k0 = (uint32_t)g->seed; a.k1 = (uint32_t)(g->seed >> 32); a.acc_count = g->acc_count_dev; a.queue = st->queue_dev;
  rc = set_lds(fn, st->lds_mh);
  if (rc) return rc;
  const int grid = grid_for(h);
#ifdef BNF_PROF
  static unsigned long long *prof_dev = nullptr;
  if (!prof_dev) hipMalloc((void **)&prof_dev, 256);
  hipMemsetAsync(prof_dev, 0, 256, stream);
  a.prof = prof_dev;
#endif
  if (split) {      // proposals [n x q] | the two log posteriors [2][n] of the three-launch iteration
    rc = grow((void **)&st->mh_dev, &st->mh_cap, sizeof(float) * (size_t)n * (q + 2), stream, false);
    if (rc) return rc;
    a.zprop = st->mh_dev; a.out = st->mh_dev + (size_t)n * q; a.mode = 3;
  }
  for (int i = 0; i < g->n_iters; ++i) {
    const int it = g->it_begin + i;
    launch_noise(st, false, st->dw_dev, P.set_floats, n_blocks, 2, g->seed, 2u * (uint32_t)it, g->block0, stream, x3);
    launch_signs(L, n, bs, g->block0, 2, 7, g->seed, 2u * (uint32_t)it, st->queue_dev, stream, rib0);
    a.it = it; a.init = (i == 0 && g->init) ? 1 : 0;
    if (s->bp_on) {    // conditional prior: the two evaluations' own calls of the prior net (streams 2 it, 2 it + 1)
      if ((rc = bprior_rows(h, s, n, bs, g->block0, g->seed, 2u * (uint32_t)it, 2, stream, rib0))) return rc;
      a.prior = s->bp_rows; a.prior_stride = n * (long long)(q + 2);
    }
    a.acc_blocks = g->acc_blocks_dev ? g->acc_blocks_dev + (long long)i * n_blocks : nullptr;
    if (split) {      // proposal, both evaluations as independent (item, state) units, accept step (bnx_kernels.h)
      BnxMhStep ms{};
      ms.z = g->state_dev; ms.zprop = st->mh_dev; ms.lp = a.out; ms.n = n; ms.row_base = g->row_base; ms.q = q; ms.bs = bs; ms.it = it; ms.init = a.init;
      ms.q_sd = g->q_sd; ms.q_sd_blocks = g->q_sd_blocks_dev; ms.k0 = a.k0; ms.k1 = a.k1; ms.acc_count = g->acc_count_dev; ms.acc_blocks = a.acc_blocks;
      bnx_launch_propose(ms, stream);
      hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * c.W), st->lds_mh, stream, a);
      bnx_launch_accept(ms, stream);
    } else
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * c.W), st->lds_mh, stream, a);
    const int d = it - g->burn_in;
    if (d >= 0 && d < g->n_keep) {
      if (g->draws_dev)
        BGM_HIP_CHECK(hipMemcpyAsync(g->draws_dev + (long long)d * n * q, g->state_dev, sizeof(float) * n * q, hipMemcpyDeviceToDevice, stream));
      if (g->effect) {
        rc = effects_of(h, st, g->state_dev, n, bs, n_blocks, g->block0, g->row_base, n_doses, g->effect == 1 ? g->x_values_dev : st->pair_dev,
                        g->seed, 0x40000000u + (uint32_t)d * (uint32_t)n_doses, (uint32_t)it, g->sample_y,
                        g->effect == 1 ? g->adrf_sum_dev + d : nullptr, g->n_keep, g->effect == 2 ? g->ite_dev + d : nullptr, g->n_keep,
                        dw_eff_dev, L, stream, x3, rib0);
        if (rc) return rc;
      }
    }
  }
#ifdef BNF_PROF
  {
    unsigned long long hp[32];
    hipStreamSynchronize(stream);
    hipMemcpy(hp, prof_dev, sizeof(hp), hipMemcpyDeviceToHost);
    const double d = (double)grid * std::max(1, g->n_iters);
    fprintf(stderr, "[BNF_PROF] cycles per workgroup-launch (wave 0, R=%d W=%d): prologue %.0f g-first %.0f g-hidden %.0f g-last %.0f h %.0f f %.0f rest %.0f\n",
            c.R, c.W, hp[0] / d, hp[1] / d, hp[2] / d, hp[3] / d, hp[4] / d, hp[5] / d, hp[6] / d);
    fprintf(stderr, "[BNF_PROF] extra stamp 7 (split precision: hidden-layer epilogues; 2 = their matrix products): %.0f\n", hp[28] / d);
    const double dw = d * c.W;
    fprintf(stderr, "[BNF_PROF] wave busy cycles per launch: mean %.0f, max (over all launches) %llu; by wave index:", hp[7] / dw, hp[8]);
    for (int w = 0; w < c.W; ++w) fprintf(stderr, " %.0f", hp[9 + w] / d);
    fprintf(stderr, "\n");
  }
#endif
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

int bnf_effects(bgm_handle *h, BnnState *s, const float *draws, int64_t n, int32_t block_rows, int32_t block0, int64_t row_base, int32_t n_keep,
                int32_t it0, uint64_t seed, int32_t effect, int32_t sample_y, const float *x_values, int32_t n_doses, double *adrf_sum,
                float *ite, hipStream_t stream) {
  if (s->bnf_unsupported) return 1;
  const int nd = effect == 1 ? n_doses : 2;
  if (nd > BNF_MAX_DOSES) return 1;
  BnfState *st;
  int rc = bnf_session(h, s, st, stream);
  if (rc) return rc;
  const BnfPlan &P = st->P;
  const int bs = block_rows, n_blocks = (int)((n + bs - 1) / bs), q = s->q;
  const long long eset = (long long)P.e_frags * 256;
  rc = grow((void **)&st->dw_dev, &st->dw_cap, sizeof(float) * (size_t)n_blocks * nd * eset, stream, true);
  if (rc) return rc;
  SgLayout L = sg_layout(nullptr, n, 0, nd);
  rc = grow(&st->sg_dev, &st->sg_cap, L.bytes, stream, false);
  if (rc) return rc;
  L = sg_layout(st->sg_dev, n, 0, nd);
  for (int d = 0; d < n_keep; ++d) {
    rc = effects_of(h, st, draws + (long long)d * n * q, n, bs, n_blocks, block0, row_base, nd, effect == 1 ? x_values : st->pair_dev, seed,
                    0x40000000u + (uint32_t)d * (uint32_t)nd, (uint32_t)(it0 + d), sample_y, effect == 1 ? adrf_sum + d : nullptr, n_keep,
                    effect == 2 ? ite + d : nullptr, n_keep, st->dw_dev, L, stream, s->precision == 2);
    if (rc) return rc;
  }
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}
