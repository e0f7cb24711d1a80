// bnf_det_host.h -- the general-shape sampling path of the deterministic CausalBGM (bnf_det_api.hip), entered from causal_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bgm_hip.h"

struct bgm_handle;

bool bnf_det_wanted(const bgm_handle *h);      // no LDS-resident compiled shape contains the model (or BGM_FORCE_GENERAL is set)
int bnf_det_slots(bgm_handle *h);              // leading dimension of adrf_partial: one slot per workgroup
int bnf_det_logpost(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, float *out, hipStream_t stream);
int bnf_det_mh_run(bgm_handle *h, const bgm_mh_args *a, hipStream_t stream);
int bnf_det_evaluate(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, const float *x_values,
                     int32_t n_doses, double *sums, float *adrf_partial, float *ite, hipStream_t stream);
int bnf_det_effects(bgm_handle *h, const float *draws, int64_t n, int64_t row_base, int32_t n_keep, int32_t burn_in, uint64_t seed,
                    int32_t sample_y, const float *x_values, int32_t n_doses, float *adrf_partial, float *ite, hipStream_t stream);
void bnf_det_free(bgm_handle *h);
