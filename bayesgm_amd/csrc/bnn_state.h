// bnn_state.h -- host-side session state of the Bayesian-network path (bnn_api.hip, bnn_sample_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/bgm_hip.h"
#include "bnn_kernels.h"

#define BNN_ADAM_B1 0.9f
#define BNN_ADAM_B2 0.99f
#define BNN_ADAM_EPS 1e-7f

struct BnnState {
  bgm_bnn_config cfg{};
  BnnNet net[4];
  int n_params = 0, q = 0, p = 0, wmax = 0;
  float *dev = nullptr;        // one allocation: theta | m | v | grad | workspace | out | dz
  float *theta_dev = nullptr, *m_dev = nullptr, *v_dev = nullptr, *grad_dev = nullptr, *ws_dev = nullptr, *out_dev = nullptr,
        *dz_dev = nullptr, *dz_part_dev = nullptr;
  long long ws_stride = 0;
  size_t ws_floats = 0;
  long long t_theta = 0, t_z = 0;
  int *tlast_dev = nullptr;    // replay mode of the latent Adam (bgm_bnn_z_sync): step each row's (z, m, v) are current to
  long long tlast_rows = 0, z_synced = -1;
  // large-batch (sampling / evaluation) side: packed kernels and per-call perturbations, rebuilt when theta changes
  bool packed_valid = false;
  float *samp_dev = nullptr;
  size_t samp_cap = 0;
  void *bnf = nullptr;         // BnfState (bnf_api.hip): the fixed-normalisation sampling path's tables, packed blob, buffers
  bool bnf_valid = false, bnf_unsupported = false;
  void *egm = nullptr;         // BnnEgmState (bnn_egm_api.hip)
  void *chain = nullptr;       // BnnFitChain (bnn_api.hip): tables / workspace of the row-tile-chain step kernels, or NULL
};

void bgm_bnn_egm_free(void *egm_state);
void bnf_free(void *state);

inline void bnn_free_sampler(BnnState *s) {
  if (s->samp_dev) hipFree(s->samp_dev);
  s->samp_dev = nullptr;
  s->samp_cap = 0;
  s->packed_valid = false;
  if (s->bnf) bnf_free(s->bnf);
  s->bnf = nullptr;
  s->bnf_valid = false;
}
