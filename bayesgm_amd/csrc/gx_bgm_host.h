// gx_bgm_host.h -- the general-width engine for BGM (gx_bgm_api.hip) as seen from bgm_api.hip / fit_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bgm_hip.h"

struct bgm_handle;
struct BgmState;

bool gxb_wanted(const BgmState *s);     // trunk not [64] x 3 / [64] x 5, or z_dim > 16 (or BGM_FORCE_GX=1)
void gxb_free(BgmState *s);
int gxb_logpost(bgm_handle *h, BgmState *s, const float *z, const float *x, int64_t n, float *out, float *grad, hipStream_t stream);
int gxb_hmc_run(bgm_handle *h, BgmState *s, const bgm_hmc_args *a, hipStream_t stream);
int gxb_predict_draws(bgm_handle *h, BgmState *s, const float *draws, int64_t n, int64_t row_base, int32_t n_draws, int32_t burn_in,
                      uint64_t seed, const int32_t *slot, int32_t k_slots, float *cells, float *full, float *var_full, int32_t add_noise,
                      hipStream_t stream);
int gxb_fit_begin(bgm_handle *h, BgmState *s, int64_t n_rows, int32_t max_batch, hipStream_t stream);
int gxb_fit_fwd_bwd(bgm_handle *h, BgmState *s, const float *x, const float *data_z, const int32_t *idx, int batch, double *loss,
                    hipStream_t stream);
float *gxb_pack(BgmState *s);
float *gxb_packT(BgmState *s);
