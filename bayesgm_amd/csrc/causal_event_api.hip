// causal_event_api.hip -- host side of the event form of CausalBGM.predict's retained phase (causal_event_kernels.h):
// the EFFECT = 3 instantiations of causal_mh_kernel (transitions + event append), the dense outcome-net tiles and the spread pass,
// the per-segment buffers, and the segment length that fits the memory budget.  Called from bgm_causal_mh_run (causal_api.hip).
// replaces: the retained iterations of metropolis_hastings_sampler (causalbgm/base.py:860-899) + infer_from_latent_posterior (:671-763).
#include <algorithm>
#include <cstdlib>
#include <string>

#include "bgm_host.h"
#include "causal_event_kernels.h"

#ifndef BGM_MH_R
#define BGM_MH_R 1
#endif
#ifndef BGM_MH_WAVES
#define BGM_MH_WAVES 8
#endif
static constexpr int EV_MH_R = BGM_MH_R, EV_MH_WAVES = BGM_MH_WAVES, EV_SPREAD_WAVES = 4, EV_SPREAD_LDS_FLOATS = 8192;
#define BGM_CAUSAL_VARIANTS(X) X(1, 3, 13) X(1, 3, 7) X(1, 3, 2) X(2, 1, 10) X(2, 1, 7) X(2, 1, 2)

// served by the event form: dose-response sums on the LDS-resident kernels (fp32 or split precision, with either prior),
// doses in registers
bool bgm_causal_event_wanted(const bgm_handle *h, int effect, int n_doses) {
  static const bool off = std::getenv("BGM_NO_EVENT_SPLIT") != nullptr;      // dev A/B
  if (off || EV_MH_R != 1 || h->outcome_cache != 2) return false;
  if (effect == BGM_EFFECT_ITE) return h->cfg.binary_treatment != 0;         // the two arms of a binary treatment (round 6)
  return effect == BGM_EFFECT_ADRF && n_doses >= 1 && (n_doses + 3) / 4 <= EV_NCMAX;
}

template <class T>
static int ev_reserve(T *&ptr, size_t &cap, size_t need) {
  if (cap >= need) return BGM_OK;
  if (ptr) BGM_HIP_CHECK(hipFree(ptr));
  ptr = nullptr; cap = 0;
  if (hipMalloc((void **)&ptr, need * sizeof(T)) != hipSuccess) {      // out of memory: the caller falls back to the per-wave cache
    (void)hipGetLastError();
    ptr = nullptr;
    return 1;
  }
  cap = need;
  return BGM_OK;
}

// Segment length: a slot's region must hold the worst case (every chain of every one of its tiles moves at every iteration), so a
// segment of S iterations needs  n_slots * tiles_per_slot * 16 * S  events of  4 q + 4 + 32 n_calls  bytes; S is the largest length
// that fits the budget (bgm_causal_set_event_budget, else BGM_EVENT_BUDGET_MB, else 8 GiB and never more than half of the device
// memory that is free right now), down to S = 1.  Typical use touches a fraction `acceptance rate` of it.
// Returns BGM_OK, a negative error, or +1: the buffers of even a one-iteration segment do not fit the budget / the device
// (the caller then runs the retained phase on the fused kernel with the per-wave cache, mode 1).
int bgm_causal_event_plan(bgm_handle *h, long long n, int n_slots, int n_doses, int n_iters, int *seg_len, long long *ev_cap) {
  const long long n_tiles = (n + 15) / 16, tps = (n_tiles + n_slots - 1) / n_slots;
  const int n_calls = (n_doses + 3) / 4;
  long long budget = h->ev_budget_bytes;
  if (budget <= 0) {
    const char *e = std::getenv("BGM_EVENT_BUDGET_MB");
    budget = (e ? std::max(1ll, std::atoll(e)) : 8192ll) << 20;
    if (!e) {      // the default never asks for more than half of what the device has free (buffers already held count as free)
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t held = (h->ev_z_cap * (size_t)1 + h->ev_meta_cap + h->ev_out_cap) * 4;
        budget = std::min<long long>(budget, (long long)((free_b + held) / 2));
      } else (void)hipGetLastError();
    }
  }
  // bytes per retained iteration of a segment: state + meta word + (mean, sd) pairs of every event
  const long long per_iter = (long long)n_slots * tps * 16 * (4ll * h->q + 4 + 32ll * n_calls);
  // what does not scale with S: the carried pairs (two buffers) and the tile table
  const long long fixed = 2ll * n_tiles * n_calls * 64 * 2 * 4 + n_tiles * 2 * 4 + (long long)n_slots * 4;
  long long S = (budget - fixed) / std::max(1ll, per_iter);
  if (S < 1) return 1;
  S = std::min<long long>(S, n_iters);
  S = std::min<long long>(S, 128);      // (a longer segment saves nothing: three launches and one re-read of the panel per segment are ~1e-4 of its
                                         //  cost by then -- and a small panel would otherwise reserve the whole 8 GiB: hipMalloc of gigabytes is not free)
  S = std::min<long long>(S, (1ll << 27) / std::max(1ll, tps * 16));      // event indices of a slot stay far inside 32 bits
  S = std::min<long long>(S, std::max(1, EV_SPREAD_LDS_FLOATS / std::max(1, n_doses)));      // the spread pass keeps [S][n_doses] sums in LDS
  S = std::max(1ll, S);
  const long long cap = tps * 16 * S;
  const size_t ev_total = (size_t)n_slots * (size_t)cap;
  int rc;
  if ((rc = ev_reserve(h->ev_z, h->ev_z_cap, ev_total * (size_t)h->q))) return rc;
  if ((rc = ev_reserve(h->ev_meta, h->ev_meta_cap, ev_total))) return rc;
  if ((rc = ev_reserve(h->ev_tile, h->ev_tile_cap, (size_t)n_tiles * 2))) return rc;
  if ((rc = ev_reserve(h->ev_slot_cnt, h->ev_slot_cap, (size_t)n_slots))) return rc;
  if ((rc = ev_reserve(h->ev_out, h->ev_out_cap, ev_total / 16 * (size_t)n_calls * 64 * 2))) return rc;
  if ((rc = ev_reserve(h->ev_carry, h->ev_carry_cap, 2 * (size_t)n_tiles * (size_t)n_calls * 64 * 2))) return rc;      // two buffers, alternating
  *seg_len = (int)S; *ev_cap = cap;
  return BGM_OK;
}

void bgm_causal_event_free(bgm_handle *h) {
  for (void *p : {(void *)h->ev_z, (void *)h->ev_meta, (void *)h->ev_tile, (void *)h->ev_slot_cnt, (void *)h->ev_out, (void *)h->ev_carry})
    if (p) hipFree(p);
  h->ev_z = nullptr; h->ev_meta = nullptr; h->ev_tile = nullptr; h->ev_slot_cnt = nullptr; h->ev_out = nullptr; h->ev_carry = nullptr;
  h->ev_z_cap = h->ev_meta_cap = h->ev_tile_cap = h->ev_slot_cap = h->ev_out_cap = h->ev_carry_cap = 0;
}

template <class K>
static int ev_set_lds(K kernel, int bytes) {
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return BGM_OK;
}

// transitions of one segment; ka carries the segment (it_begin, n_iters, ev_first) and everything bgm_causal_mh_run filled in
int bgm_causal_event_mh_launch(bgm_handle *h, CausalMhKArgs &ka, int grid, int lds, hipStream_t stream) {
  ka.ev_z = h->ev_z; ka.ev_meta = h->ev_meta; ka.tile_ev = h->ev_tile; ka.slot_cnt = h->ev_slot_cnt;
  // split precision: the transitions run on the bf16x3 / f16x3 kernel; the events' outcome-net tiles and the spread pass are the fp32 ones
  if (h->precision != 0) return bgm_causal_bx3_mh_launch(h, ka, 3, grid, stream);
  // conditional latent prior (IdentifiableCausalBGM): the PRIOR = 1 instantiation; the outcome net does not see the prior
  if (h->prior_seg) return bgm_causal_prior_mh_launch(h, ka, 3, grid, lds, stream);
  int rc;
#define X(KT1_, KSL1_, NTL_)                                                                   \
  if (h->KT1 == KT1_ && h->KSL1 == KSL1_ && h->NTL == NTL_) {                                  \
    auto k = causal_mh_kernel<KT1_, KSL1_, NTL_, EV_MH_R, EV_MH_WAVES, 3>;                     \
    if ((rc = ev_set_lds(k, lds))) return rc;                                                  \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * EV_MH_WAVES), lds, stream, ka);                \
    BGM_HIP_CHECK(hipGetLastError());                                                          \
    return BGM_OK;                                                                             \
  }
  BGM_CAUSAL_VARIANTS(X)
#undef X
  bgm_set_error("no compiled MH kernel variant for this shape (event form)");
  return BGM_E_UNSUPPORTED;
}

// outcome net on the segment's events, then the spread over its retained iterations
int bgm_causal_event_finish(bgm_handle *h, const CausalMhKArgs &ka, int grid, int first, hipStream_t stream, int effect) {
  CausalEventFArgs fa{};
  fa.blob = ka.blob; fa.ev_z = h->ev_z; fa.slot_cnt = h->ev_slot_cnt; fa.ev_cap = ka.ev_cap; fa.n_doses = ka.n_doses;
  fa.x_values = ka.x_values; fa.ev_out = reinterpret_cast<float2 *>(h->ev_out); fa.eff_stats = ka.eff_stats; fa.m = ka.m;
  int rc = BGM_E_UNSUPPORTED;
  bool done = false;
  constexpr int FW = 4, WPS = 2;       // waves per workgroup, waves per sampler slot
  if (effect == BGM_EFFECT_ITE) {      // binary treatment: the two arms per event, then one thread per chain (causal_event_kernels.h)
#define X(KT1_, KSL1_)                                                                         \
    if (!done && h->KT1 == KT1_ && h->KSL1 == KSL1_) {                                         \
      const int lds = 4 * (16 * KT1_ * 64 + 64 + 64 * 32 + 32 + 32 * 16 + 16 + 16 * 16 + 16 + 64); \
      auto k = causal_event_f_ite_kernel<KT1_, KSL1_, FW, WPS>;                                \
      if ((rc = ev_set_lds(k, lds))) return rc;                                                \
      hipLaunchKernelGGL(k, dim3(grid * EV_MH_WAVES * WPS / FW), dim3(64 * FW), lds, stream, fa); \
      BGM_HIP_CHECK(hipGetLastError());                                                        \
      done = true;                                                                             \
    }
    X(1, 3) X(2, 1)
#undef X
    if (!done) { bgm_set_error("no compiled outcome-net kernel for this shape (event form, binary treatment)"); return BGM_E_UNSUPPORTED; }
    CausalEventIteArgs ia{};
    ia.n = ka.n; ia.row_base = ka.row_base; ia.it_begin = ka.it_begin; ia.n_iters = ka.n_iters; ia.burn_in = ka.burn_in; ia.n_keep = ka.n_keep;
    ia.sample_y = ka.sample_y; ia.n_slots = grid * EV_MH_WAVES; ia.k0 = ka.k0; ia.k1 = ka.k1;
    ia.ev_meta = h->ev_meta; ia.tile_ev = h->ev_tile; ia.ev_cap = ka.ev_cap; ia.ev_out = reinterpret_cast<const float4 *>(h->ev_out);
    const size_t carry_floats = (size_t)((ka.n + 15) / 16) * 64 * 2;      // (one Philox call's worth per tile: >= 16 bytes per chain)
    h->ev_carry_flip = first ? 0 : (h->ev_carry_flip ^ 1);
    ia.carry_in = reinterpret_cast<const float4 *>(h->ev_carry + (h->ev_carry_flip ^ 1) * carry_floats);
    ia.carry_out = reinterpret_cast<float4 *>(h->ev_carry + h->ev_carry_flip * carry_floats);
    ia.first = first; ia.ite = ka.ite;
    hipLaunchKernelGGL(causal_event_ite_kernel, dim3((unsigned)((ka.n + 255) / 256)), dim3(256), 0, stream, ia);
    BGM_HIP_CHECK(hipGetLastError());
    return BGM_OK;
  }
#define X(KT1_, KSL1_)                                                                         \
  if (!done && h->KT1 == KT1_ && h->KSL1 == KSL1_) {                                           \
    const int lds = 4 * (16 * KT1_ * 64 + 64 + 64 * 32 + 32 + 32 * 16 + 16 + 16 * 16 + 16 + 64); \
    auto k = causal_event_f_kernel<KT1_, KSL1_, FW, WPS>;                                      \
    if ((rc = ev_set_lds(k, lds))) return rc;                                                  \
    hipLaunchKernelGGL(k, dim3(grid * EV_MH_WAVES * WPS / FW), dim3(64 * FW), lds, stream, fa); \
    BGM_HIP_CHECK(hipGetLastError());                                                          \
    done = true;                                                                               \
  }
  X(1, 3) X(2, 1)
#undef X
  if (!done) { bgm_set_error("no compiled outcome-net kernel for this shape (event form)"); return BGM_E_UNSUPPORTED; }
  CausalEventSpreadArgs sa{};
  sa.n = ka.n; sa.row_base = ka.row_base; sa.it_begin = ka.it_begin; sa.n_iters = ka.n_iters; sa.burn_in = ka.burn_in; sa.n_keep = ka.n_keep;
  sa.sample_y = ka.sample_y; sa.n_doses = ka.n_doses; sa.k0 = ka.k0; sa.k1 = ka.k1;
  sa.ev_meta = h->ev_meta; sa.tile_ev = h->ev_tile; sa.ev_cap = ka.ev_cap; sa.ev_out = reinterpret_cast<const float2 *>(h->ev_out);
  const size_t carry_floats = (size_t)((ka.n + 15) / 16) * (size_t)((ka.n_doses + 3) / 4) * 64 * 2;
  h->ev_carry_flip = first ? 0 : (h->ev_carry_flip ^ 1);
  sa.carry_in = reinterpret_cast<const float2 *>(h->ev_carry + (h->ev_carry_flip ^ 1) * carry_floats);
  sa.carry_out = reinterpret_cast<float2 *>(h->ev_carry + h->ev_carry_flip * carry_floats);
  sa.first = first; sa.adrf_partial = ka.adrf_partial;
  // one block of EV_SPREAD_WAVES waves per sampler slot
  hipLaunchKernelGGL(causal_event_spread_kernel<EV_SPREAD_WAVES>, dim3(grid * EV_MH_WAVES), dim3(64 * EV_SPREAD_WAVES),
                     sizeof(float) * (size_t)ka.n_iters * (size_t)ka.n_doses, stream, sa);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_causal_set_event_budget(bgm_handle *h, int64_t bytes) {
  if (!h || bytes < 0) { bgm_set_error("bgm_causal_set_event_budget: bad argument"); return BGM_E_INVALID; }
  h->ev_budget_bytes = bytes;
  return BGM_OK;
}
