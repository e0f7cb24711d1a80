// bnn_state.h -- host-side session state of the Bayesian-network path (bnn_api.hip, bnn_sample_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/bgm_hip.h"
#include "bnn_kernels.h"

#define BNN_ADAM_B1 0.9f
#define BNN_ADAM_B2 0.99f
#define BNN_ADAM_EPS 1e-7f

#include "bprior_types.h"
struct BnnState {
  bgm_bnn_config cfg{};
  BnnNet net[4];
  int n_params = 0, q = 0, p = 0, wmax = 0;
  float *dev = nullptr;        // one allocation: theta | m | v | grad | workspace | out | dz
  float *theta_dev = nullptr, *m_dev = nullptr, *v_dev = nullptr, *grad_dev = nullptr, *ws_dev = nullptr, *out_dev = nullptr,
        *dz_dev = nullptr, *dz_part_dev = nullptr;
  long long ws_stride = 0;
  size_t ws_floats = 0;
  float *kl_part_dev = nullptr;      // partial KL sums of the general theta step (bnn_kl_adam_kernel)
  long long t_theta = 0, t_z = 0;
  int *tlast_dev = nullptr;    // replay mode of the latent Adam (bgm_bnn_z_sync): step each row's (z, m, v) are current to
  long long tlast_rows = 0, z_synced = -1;
  // large-batch (sampling / evaluation) side: packed kernels and per-call perturbations, rebuilt when theta changes
  bool packed_valid = false;
  float *samp_dev = nullptr;
  size_t samp_cap = 0;
  void *bnf = nullptr;         // BnfState (bnf_api.hip): the fixed-normalisation sampling path's tables, packed blob, buffers
  bool bnf_valid = false, bnf_unsupported = false;
  int precision = 0;           // arithmetic of the sampling calls: 0 fp32, 2 split precision "f16 x 3" (bgm_bnn_set_precision; bnx_kernels.h)
  void *egm = nullptr;         // BnnEgmState (bnn_egm_api.hip)
  void *chain = nullptr;       // BnnFitChain (bnn_api.hip): tables / workspace of the row-tile-chain step kernels, or NULL
  void *bnw = nullptr;         // BnwState (bnw_api.hip): buffers of the any-width sampling path
  // conditional latent prior of IdentifiableCausalBGM(use_bnn=True) for the sampling calls (bprior_api.hip, bgm_bnn_set_prior)
  bool bp_on = false;
  BPriorNet bp_net{};
  const float *bp_theta = nullptr;
  const int *bp_seg = nullptr;
  float *bp_rows = nullptr;    // [n_states][n][q + 2] of the current call
  size_t bp_rows_cap = 0;
  int *bp_hist = nullptr;      // batch statistics of the prior net's one-hot input: rows per (block, segment) of the panel bp_seg [n_blocks][k]
  size_t bp_hist_cap = 0;
  long long bp_hist_n = -1;    // (n, block rows) the counts were made for; -1: none (bgm_bnn_set_prior invalidates)
  int bp_hist_bs = 0;
};
int bprior_rows(bgm_handle *h, BnnState *s, long long n, int bs, int block0, uint64_t seed, uint32_t stream0, int n_states, hipStream_t stream,
                long long rib0 = 0);

void bgm_bnn_egm_free(void *egm_state);
void bnf_free(void *state);
void bnw_free(void *state);
int bnw_logpost(bgm_handle *h, BnnState *s, const float *x, const float *y, const float *v, const float *z, int64_t n, int32_t bs, int32_t block0,
                uint64_t seed, uint32_t stream_id, float *out, hipStream_t stream);
int bnw_mh_run(bgm_handle *h, BnnState *s, const bgm_bnn_mh_args *g, hipStream_t stream);
int bnw_effects(bgm_handle *h, BnnState *s, const float *draws, int64_t n, int32_t bs, int32_t block0, int64_t row_base, int32_t n_keep, int32_t it0,
                uint64_t seed, int32_t effect, int32_t sample_y, const float *x_values, int32_t n_doses, double *adrf_sum, float *ite,
                hipStream_t stream);
int bnw_evaluate(bgm_handle *h, BnnState *s, const float *x, const float *y, const float *v, float *z, int32_t encode, int64_t n, const float *x_values,
                 int32_t n_doses, uint64_t seed, uint32_t stream_id, double *sums, double *dose_sums, float *ite, hipStream_t stream);

inline void bnn_free_sampler(BnnState *s) {
  if (s->samp_dev) hipFree(s->samp_dev);
  s->samp_dev = nullptr;
  s->samp_cap = 0;
  s->packed_valid = false;
  if (s->bnf) bnf_free(s->bnf);
  s->bnf = nullptr;
  if (s->bnw) bnw_free(s->bnw);
  s->bnw = nullptr;
  s->bnf_valid = false;
  if (s->bp_rows) hipFree(s->bp_rows);
  s->bp_rows = nullptr; s->bp_rows_cap = 0;
  if (s->bp_hist) hipFree(s->bp_hist);
  s->bp_hist = nullptr; s->bp_hist_cap = 0; s->bp_hist_n = -1;
}
