// gx_host.h -- the general-width engine (gx_api.hip, gx_device.h) as seen from the C-ABI translation units: CausalBGM models whose
// hidden widths / depths are not the ones the resident, streamed-fragment and row-tile-chain kernel families are compiled for.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bgm_hip.h"

struct bgm_handle;

bool gx_wanted(const bgm_handle *h);        // g / f / h are not g [64] x k, f / h [64, 32, 8]  (or BGM_FORCE_GX=1)
bool gx_enc_wanted(const bgm_handle *h);    // e is not [64] x k with v_dim <= 208 (or BGM_FORCE_GX=1)
int gx_slots(bgm_handle *h, int64_t n);     // leading dimension of adrf_partial for n rows: one slot per workgroup (per wave: row-tile-per-wave kernels)
bool gx_row_tile_per_wave(bgm_handle *h);   // the sampling calls of this model run on gw_kernels.h (hidden layers up to ~128 wide)
int gx_logpost(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, float *out, hipStream_t stream);
int gx_mh_run(bgm_handle *h, const bgm_mh_args *a, hipStream_t stream);
int gx_evaluate(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, const float *x_values,
                int32_t n_doses, double *sums, float *adrf_partial, float *ite, hipStream_t stream);
int gx_effects(bgm_handle *h, const float *draws, int64_t n, int64_t row_base, int32_t n_keep, int32_t burn_in, uint64_t seed,
               int32_t sample_y, const float *x_values, int32_t n_doses, float *adrf_partial, float *ite, hipStream_t stream);
int gx_encode(bgm_handle *h, const float *v, int64_t n, float *z, hipStream_t stream);
// fit session on the general engine: called by bgm_causal_fit_begin after the canonical parameters / Adam slots are on the device
int gx_fit_begin(bgm_handle *h, int64_t n_rows, int32_t max_batch, hipStream_t stream);
bool gx_fit_active(const bgm_handle *h);
int gx_fit_grads(bgm_handle *h, const float *x, const float *y, const float *v, const float *data_z, const int32_t *idx, int64_t row_lo,
                 int32_t batch, int32_t batch_global, int z_mode, float *grad, double *loss, hipStream_t stream);
float *gx_pack(bgm_handle *h);              // padded forward / transposed packs (targets of the Adam kernel's scatter tables)
float *gx_packT(bgm_handle *h);
void gx_fit_end(bgm_handle *h);
void gx_free(bgm_handle *h);
