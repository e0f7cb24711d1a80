// bnf_host.h -- entry points of the fixed-normalisation Bayesian-network sampling path (bnf_api.hip), called from
// bnn_sample_api.hip.  Return 0: handled; 1: the session is outside this path (default net shapes, bnn_norm = "fixed",
// q <= 31, p <= 207, blob <= 160 KB) and the caller continues with the batch-statistics kernels; < 0: BGM_E_* failure.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bgm_hip.h"

struct BnnState;
struct bgm_handle;

int bnf_logpost(bgm_handle *h, BnnState *s, const float *x, const float *y, const float *v, const float *z, int64_t n, int32_t block_rows,
                int32_t block0, uint64_t seed, uint32_t stream_id, float *out, hipStream_t stream);
int bnf_mh_run(bgm_handle *h, BnnState *s, const bgm_bnn_mh_args *g, hipStream_t stream);
int bnf_effects(bgm_handle *h, BnnState *s, const float *draws, int64_t n, int32_t block_rows, int32_t block0, int64_t row_base, int32_t n_keep,
                int32_t it0, uint64_t seed, int32_t effect, int32_t sample_y, const float *x_values, int32_t n_doses, double *adrf_sum,
                float *ite, hipStream_t stream);
void bnf_free(void *state);
