// bgm_host.h -- host-side state shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/bgm_hip.h"
#include "causal_kernels.h"
#include "fit_types.h"

void bgm_set_error(const std::string &msg);

#define BGM_HIP_CHECK(expr)                                                                  \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      bgm_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                      \
      return BGM_E_HIP;                                                                      \
    }                                                                                        \
  } while (0)

struct HostNet {
  std::vector<int> dims;    // [in, h1, ..., out]
  std::vector<float> theta; // Keras order: W0 [in x out], b0, W1, b1, ...
  bool set = false;
  const float *W(int l) const {
    size_t off = 0;
    for (int i = 0; i < l; ++i) off += (size_t)dims[i] * dims[i + 1] + dims[i + 1];
    return theta.data() + off;
  }
  const float *b(int l) const { return W(l) + (size_t)dims[l] * dims[l + 1]; }
  size_t count() const {
    size_t c = 0;
    for (size_t i = 0; i + 1 < dims.size(); ++i) c += (size_t)dims[i] * dims[i + 1] + dims[i + 1];
    return c;
  }
};

struct bgm_handle {
  int device = 0;
  int n_cus = 256;
  int disc_norm = 0;   // Discriminator BatchNormalization: 0 batch statistics, 1 inference mode (bgm_set_disc_norm)
  bool configured = false;
  bgm_causal_config cfg{};
  int q = 0, p = 0;
  HostNet nets[4];
  // MH / log-posterior blob (g, f, h)
  CausalMeta meta{};
  int KT1 = 0, KSL1 = 0, NTL = 0;  // required kernel variant
  float *blob_dev = nullptr;
  size_t blob_cap = 0;
  bool blob_valid = false;
  // sampling copy of the blob: the weights of every layer that consumes a LeakyReLU activation carry the factor 0.6 of the
  // one-instruction activation lrelu_s (bgm_device.h); rebuilt from blob_dev on the device whenever that one changed
  float *sblob_dev = nullptr;
  size_t sblob_cap = 0;
  bool sblob_valid = false;
  // split-precision (bf16 x 3) sampling blob (causal_bx3_api.hip); precision: 0 fp32 (default), 1 bf16x3, 2 f16x3 (bgm_causal_set_precision)
  int precision = 0;
  void *bx_blob_dev = nullptr;
  size_t bx_cap = 0;
  // per-wave-slot (mean, sd) of the outcome net at every dose, kept between the retained iterations of an ADRF launch (causal_kernels.h)
  float *eff_cache = nullptr;
  size_t eff_cache_cap = 0;
  int outcome_cache = 2;                          // bgm_causal_set_outcome_cache: 0 off, 1 per wave of 16 chains, 2 per chain (event form where it exists, else 1)
  unsigned long long *eff_stats_dev = nullptr;    // [0]: retained tile-iterations served from the cache
  unsigned long long eff_total = 0;               // retained tile-iterations launched since the last reset
  unsigned long long ev_total = 0;                // event form: retained chain-iterations launched since the last reset
  // event form of the retained phase (causal_event_api.hip; outcome_cache == 2): per-slot event regions of one segment of retained
  // iterations, the outcome net's (mean, sd) per event, and the chains' current pairs carried between the segments of a call
  float *ev_z = nullptr; unsigned *ev_meta = nullptr; int *ev_tile = nullptr, *ev_slot_cnt = nullptr; float *ev_out = nullptr, *ev_carry = nullptr;
  size_t ev_z_cap = 0, ev_meta_cap = 0, ev_tile_cap = 0, ev_slot_cap = 0, ev_out_cap = 0, ev_carry_cap = 0;
  int ev_carry_flip = 0;
  long long ev_budget_bytes = 0;                  // bgm_causal_set_event_budget (0: BGM_EVENT_BUDGET_MB, else min(8 GiB, half of the free device memory))
  int ev_fallbacks = 0;                           // retained phases that ran on the per-wave cache because no segment fitted the budget / the device
  bool bx_valid = false;
  alignas(8) unsigned char bx_meta_store[192];
  // per-row conditional latent prior of the sampling kernels (causal_prior_api.hip, bgm_causal_set_prior); NULL = standard normal
  const int32_t *prior_seg = nullptr;
  const float *prior_tab = nullptr;
  int prior_segments = 0;
  // encoder blob
  float *eblob_dev = nullptr;
  size_t eblob_cap = 0;
  bool eblob_valid = false;
  // fit state (device)
  bool fit_active = false;
  void *fit_chain = nullptr;     // FitChainState (fit_api.hip): row-tile-chain step kernels at B = 16 / 32, or NULL
  long long fit_rows = 0;
  int fit_bcap = 0, n_params = 0, n_slices_cap = 0, rows_per_slice = 0;
  long long t_theta = 0, t_z = 0;
  float *theta_dev = nullptr, *m1_dev = nullptr, *m2_dev = nullptr, *bblob_dev = nullptr, *ws_dev = nullptr,
        *partial_dev = nullptr;
  int *tables_dev = nullptr;  // fwd_dst | fwd_dst2 | bwd_dst | grad_src, n_params each
  int *pos_dev = nullptr;
  int *tlast_dev = nullptr;   // replay mode of the latent Adam (bgm_causal_fit_z_sync): step each row's (z, m, v) are current to
  long long z_synced = -1;    // the minibatch step whose rows bgm_causal_fit_z_sync has brought up to date
  FitMeta fit_meta{};
  FitWs fit_ws{};
  DwArgs dw{};
  unsigned *acc_scratch = nullptr;   // [n_slots x n_iters] per-launch acceptance counters
  size_t acc_scratch_cap = 0;
  void *det_state = nullptr;  // BnfState (bnf_det_api.hip): general-shape sampling path of the deterministic nets
  bool det_valid = false;
  // bgm_causal_fit_epoch: gradient buffer, second stream and the events that order the two phases
  float *epoch_grad = nullptr; int epoch_grad_n = 0;
  hipStream_t epoch_stream = nullptr;
  unsigned *epoch_ctr = nullptr;      // device: [0] gradient-tile workgroups done, [1] latent-phase workgroups done, [2] a wait gave up (fit_types.h FitSync)
  int epoch_flags_ok = 0;             // 0: not probed yet, 1: the two streams run side by side (device-side ordering is safe), -1: they do not
  void *epoch_probe_stream = nullptr; // the caller stream that probe was made with (a different one is probed again)
  unsigned epoch_theta_done = 0, epoch_z_done = 0;      // the counters' values once everything issued so far is done
  hipEvent_t epoch_ev_t[4] = {}, epoch_ev_z[4] = {}, epoch_ev_s = nullptr;      // one pair per minibatch in flight (BGM_EPOCH_DEPTH_MAX)
  void *gx_state = nullptr;   // GxState (gx_api.hip): general-width engine (hidden widths / depths outside the compiled families)
  bool gx_valid = false;      // its padded packs hold the handle's current g, f, h, e
  void *bgm_state = nullptr;  // BgmState (bgm_api.hip)
  void *egm_state = nullptr;  // EgmState (egm_api.hip)
  void *bgm_egm_state = nullptr;  // BgmEgmState (bgm_egm_api.hip)
  void *bnn_state = nullptr;      // BnnState (bnn_api.hip)
  void *bvn_state = nullptr;      // BgmbState (bgmb_api.hip)
  // timing
  bool timing = false;
  struct Ev { hipEvent_t a, b; int kind; };
  std::vector<Ev> events;
  long long timed_launches[3] = {0, 0, 0};
  double timed_ms[3] = {0.0, 0.0, 0.0};
};

// Compiled kernel shapes.  A model runs on the smallest compiled shape that contains it; the extra K rows / output
// tiles are zero weights (and zero-padded data), so results are unchanged and only some MFMAs are wasted.
//   first layers (extended input z, x: q + 1 features):  (KT1, KSL1) = (1, 3) for q + 1 <= 12, (2, 1) for q + 1 <= 20
//   g's last layer (p + 1 outputs):  NTL in {2, 7, 13} 16-wide tiles ({2, 7, 10} with KT1 = 2: LDS budget)
// Returns false when no compiled shape contains the model.
static inline bool bgm_causal_shape(int q1, int p1, int &KT1, int &KSL1, int &NTL) {
  if (q1 <= 12) { KT1 = 1; KSL1 = 3; }
  else if (q1 <= 20) { KT1 = 2; KSL1 = 1; }
  else return false;
  const int need = (p1 + 15) / 16;
  const int big = (KT1 == 1) ? 13 : 10;
  NTL = need <= 2 ? 2 : need <= 7 ? 7 : need <= big ? big : -1;
  return NTL > 0;
}
static inline int bgm_enc_in_tiles(int p) { const int t = (p + 15) / 16; return t <= 2 ? 2 : t <= 7 ? 7 : t <= 13 ? 13 : -1; }

// g's last layer [64 x (p + 1)] laid out for a compiled shape with NTL output tiles: the mean columns keep their natural
// positions, the variance column (feature p) moves to element p % 16 of the LAST tile (where the sampling kernels look
// for it at compile time), everything else is zero.  Identity when the shape is exact.
static inline int bgm_sig_slot(int p, int NTL) { return 16 * (NTL - 1) + p % 16; }
static inline void bgm_g_last_padded(const float *W, const float *b, int p, int NTL, std::vector<float> &Wp, std::vector<float> &bp) {
  const int N = 16 * NTL, slot = bgm_sig_slot(p, NTL);
  Wp.assign((size_t)64 * N, 0.0f);
  bp.assign(N, 0.0f);
  for (int i = 0; i < 64; ++i) {
    for (int k = 0; k < p; ++k) Wp[(size_t)i * N + k] = W[(size_t)i * (p + 1) + k];
    Wp[(size_t)i * N + slot] = W[(size_t)i * (p + 1) + p];
  }
  for (int k = 0; k < p; ++k) bp[k] = b[k];
  bp[slot] = b[p];
}

int bgm_causal_build_blob(bgm_handle *h, hipStream_t stream);
int bgm_causal_sampling_blob(bgm_handle *h, hipStream_t stream);   // build_blob + the scaled sampling copy (sblob_dev)
void bgm_bgm_free_state(bgm_handle *h);
void bgm_egm_free_state(bgm_handle *h);
void bgm_bgm_egm_free_state(bgm_handle *h);
void bgm_bnn_free_state(bgm_handle *h);
void bgm_bvn_free_state(bgm_handle *h);
int causal_pack_forward(bgm_handle *h, const HostNet &G, const HostNet &F, const HostNet &H, std::vector<float> &blob);
// conditional-prior sampling path (causal_prior_api.hip)
int bgm_causal_prior_logpost(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, float *out, int grid,
                             hipStream_t stream);
int bgm_causal_prior_mh_launch(bgm_handle *h, const CausalMhKArgs &a, int effect, int grid, int lds, hipStream_t stream);
// split-precision sampling path (causal_bx3_api.hip)
int bgm_causal_bx3_blob(bgm_handle *h, hipStream_t stream);
int bgm_causal_bx3_logpost(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, float *out, int grid,
                           hipStream_t stream);
int bgm_causal_bx3_mh_launch(bgm_handle *h, const CausalMhKArgs &a, int effect, int grid, hipStream_t stream);

// ---- packing into MFMA fragment order (layout documented in bgm_device.h)
// rowmap(rho) -> source input-feature row of W (or -1 for a zero row)
template <class RowMap>
inline void pack_layer(std::vector<float> &blob, int off, const float *W, int n_in, int n_out, int KT,
                       int NT, RowMap rowmap) {
  const int K_ROWS = 16 * KT;
  for (int T0 = 0; T0 < NT;) {
    const int GS = group_size(NT - T0);
    float *base = blob.data() + off + K_ROWS * 16 * T0;
    for (int rho = 0; rho < K_ROWS; ++rho) {
      const int src = rowmap(rho);
      for (int j = 0; j < 16; ++j)
        for (int u = 0; u < GS; ++u) {
          const int o = 16 * (T0 + u) + j;
          float val = 0.0f;
          if (src >= 0 && src < n_in && o < n_out) val = W[(size_t)src * n_out + o];
          base[(rho * 16 + j) * GS + u] = val;
        }
    }
    T0 += GS;
  }
}
inline void pack_bias(std::vector<float> &blob, int off, const float *b, int n_out, int NT) {
  for (int i = 0; i < 16 * NT; ++i) blob[off + i] = i < n_out ? b[i] : 0.0f;
}
// packed row rho = 16 t + 4 g + r  <->  extended input feature 16 t + 4 r + g (first layer only)
static inline int l1_feature(int rho) {
  const int t = rho >> 4, g = (rho >> 2) & 3, r = rho & 3;
  return 16 * t + 4 * r + g;
}

// causal_event_api.hip: the retained phase as transitions + events + dense outcome-net tiles + spread (causal_event_kernels.h)
struct CausalMhKArgs;
bool bgm_causal_event_wanted(const bgm_handle *h, int effect, int n_doses);
int bgm_causal_event_plan(bgm_handle *h, long long n, int n_slots, int n_doses, int n_iters, int *seg_len, long long *ev_cap);
int bgm_causal_event_mh_launch(bgm_handle *h, CausalMhKArgs &ka, int grid, int lds, hipStream_t stream);
int bgm_causal_event_finish(bgm_handle *h, const CausalMhKArgs &ka, int grid, int first, hipStream_t stream, int effect);
void bgm_causal_event_free(bgm_handle *h);
