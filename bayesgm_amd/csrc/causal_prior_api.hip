// causal_prior_api.hip -- the CausalBGM sampling kernels with a per-row conditional latent prior (IdentifiableCausalBGM,
// models/causalbgm/identifiable.py:195-211 update_latent_variable_sgd, :521-555 get_log_posterior, :557-614
// metropolis_hastings_sampler): the PRIOR = 1 instantiations of causal_kernels.h, kept in their own translation unit.
// Selected by bgm_causal_set_prior (include/bgm_hip.h); the default path (causal_api.hip) is untouched.
#include <string>

#include "bgm_host.h"

static constexpr int PR_WAVES = 8;
#define BGM_PRIOR_VARIANTS(X) X(1, 3, 13) X(1, 3, 7) X(1, 3, 2) X(2, 1, 10) X(2, 1, 7) X(2, 1, 2)

extern "C" int bgm_causal_set_prior(bgm_handle *h, const int32_t *seg_dev, const float *tab_dev, int32_t n_segments) {
  if (!h || !h->configured) { bgm_set_error("bgm_causal_set_prior: handle not configured"); return BGM_E_STATE; }
  if ((seg_dev == nullptr) != (tab_dev == nullptr) || (seg_dev && n_segments <= 0)) { bgm_set_error("bgm_causal_set_prior: seg_dev and tab_dev go together"); return BGM_E_INVALID; }
  h->prior_seg = seg_dev; h->prior_tab = tab_dev; h->prior_segments = seg_dev ? n_segments : 0;
  return BGM_OK;
}

template <class K>
static int pr_set_lds(K kernel, int bytes) {
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return BGM_OK;
}

int bgm_causal_prior_logpost(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, float *out,
                             int grid, hipStream_t stream) {
  int rc;
  const int lds = h->meta.total * 4;
#define X(KT1_, KSL1_, NTL_)                                                                                     \
  if (h->KT1 == KT1_ && h->KSL1 == KSL1_ && h->NTL == NTL_) {                                                    \
    auto k = causal_logpost_kernel<KT1_, KSL1_, NTL_, 1, PR_WAVES, 1>;                                           \
    rc = pr_set_lds(k, lds);                                                                                     \
    if (rc) return rc;                                                                                           \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * PR_WAVES), lds, stream, (const float *)h->sblob_dev, h->meta, x, y, v, z, \
                       (long long)n, out, (const int *)h->prior_seg, h->prior_tab);                              \
    BGM_HIP_CHECK(hipGetLastError());                                                                            \
    return BGM_OK;                                                                                               \
  }
  BGM_PRIOR_VARIANTS(X)
#undef X
  bgm_set_error("conditional prior: no compiled kernel variant for this shape");
  return BGM_E_UNSUPPORTED;
}

template <int EFFECT>
static int pr_launch_mh(bgm_handle *h, const CausalMhKArgs &ka, int grid, int lds, hipStream_t stream) {
  int rc;
#define X(KT1_, KSL1_, NTL_)                                                                   \
  if (h->KT1 == KT1_ && h->KSL1 == KSL1_ && h->NTL == NTL_) {                                  \
    auto k = causal_mh_kernel<KT1_, KSL1_, NTL_, 1, PR_WAVES, EFFECT, 1>;                      \
    rc = pr_set_lds(k, lds);                                                                   \
    if (rc) return rc;                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * PR_WAVES), lds, stream, ka);                   \
    BGM_HIP_CHECK(hipGetLastError());                                                          \
    return BGM_OK;                                                                             \
  }
  BGM_PRIOR_VARIANTS(X)
#undef X
  bgm_set_error("conditional prior: no compiled MH kernel variant for this shape");
  return BGM_E_UNSUPPORTED;
}

int bgm_causal_prior_mh_launch(bgm_handle *h, const CausalMhKArgs &a, int effect, int grid, int lds, hipStream_t stream) {
  CausalMhKArgs ka = a;
  ka.seg = (const int *)h->prior_seg;
  ka.prior_tab = h->prior_tab;
  if (effect == BGM_EFFECT_ADRF) return pr_launch_mh<1>(h, ka, grid, lds, stream);
  if (effect == BGM_EFFECT_ITE) return pr_launch_mh<2>(h, ka, grid, lds, stream);
  return pr_launch_mh<0>(h, ka, grid, lds, stream);
}
