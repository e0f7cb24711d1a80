// bgmb_state.h -- host-side session state of BGM with the Bayesian generator (bgmb_api.hip, bgmb_egm_api.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/bgm_hip.h"
#include "bnn_kernels.h"

#define BGMB_ADAM_B1 0.9f
#define BGMB_ADAM_B2 0.99f
#define BGMB_ADAM_EPS 1e-7f

struct BgmbState {
  bgm_bvn_config cfg{};
  BnnNet net{};
  int n_params = 0, q = 0, p = 0, wmax = 0;
  float *dev = nullptr;        // one allocation: theta | m | v | grad | workspace | out
  float *theta_dev = nullptr, *m_dev = nullptr, *v_dev = nullptr, *grad_dev = nullptr, *ws_dev = nullptr, *out_dev = nullptr;
  size_t ws_floats = 0;
  long long t_theta = 0, t_z = 0;
  float *big_dev = nullptr;    // row-tile workspace of the large-batch kernels (grown on demand)
  size_t big_cap = 0;
  float *dw_dev = nullptr;     // perturbations of the generator calls of one large-batch launch (bgmb_noise_kernel)
  size_t dw_cap = 0;
  void *egm = nullptr;         // BgmbEgmState (bgmb_egm_api.hip)
  void *gxf = nullptr;         // GxfState (bgmb_api.hip): packs of the LDS-tiled frozen-noise HMC kernel (gx_flipout.h)
  void *bgmf = nullptr;        // BgmfState (bgmf_api.hip): blob of the register-chained frozen-noise HMC kernel (bgmf_kernels.h)
  int precision = 0;           // 0 fp32 | 2 f16x3 (bgm_bvn_set_precision; frozen-noise HMC on bgmfx_hmc_kernel)
};

int bgmb_fill(const bgm_bvn_config *cfg, BnnNet &n);
void bgm_bvn_egm_free(void *egm_state);
void bgmf_free(BgmbState *s);
int bgmf_set_precision(BgmbState *s, int mode);
int bgmf_hmc_try(bgm_handle *h, BgmbState *s, const bgm_hmc_args *g, hipStream_t st);     // 0 launched, 1 not its shape, < 0 error
int bgmf_hmc_fresh(bgm_handle *h, BgmbState *s, const bgm_hmc_args *g, int it_begin, int n_iters, int init, long long dw_stride, hipStream_t st);
