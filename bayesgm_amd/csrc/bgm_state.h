// bgm_state.h -- host-side state of the BGM path shared by bgm_api.hip and fit_api.hip.
#pragma once
#include <vector>

#include "bgm_host.h"
#include "bgm_kernels.h"
#include "bgm_fit_kernels.h"

struct BgmState {
  bgm_bgm_config cfg{};
  bool configured = false, set = false;
  std::vector<float> theta;
  BgmMeta meta{};
  int KTQ = 0, NTX = 0, NH = 0;
  float *blob_dev = nullptr;
  size_t blob_cap = 0;
  bool blob_valid = false;
  // fit session (device)
  bool fit_active = false;
  int fit_bcap = 0, n_params = 0, rows_per_slice = 256, n_slices_cap = 0;
  long long t_theta = 0, t_z = 0;
  float *theta_dev = nullptr, *m1_dev = nullptr, *m2_dev = nullptr, *tblob_dev = nullptr, *ws_dev = nullptr,
        *partial_dev = nullptr, *bn_dev = nullptr;
  int *tables_dev = nullptr;   // dst | dst2(-1) | bwd(-1) | grad_src
  BgmMeta tmeta{};             // training blob layout (same offsets as meta)
  BgmFitWs fit_ws{};
  DwArgs dw{};
};

static inline BgmState *bst(bgm_handle *h) {
  if (!h->bgm_state) h->bgm_state = new BgmState();
  return static_cast<BgmState *>(h->bgm_state);
}

// dual-access pack: [out tile][in slot (K_ROWS)][17]; slotmap(slot) -> source input row or -1
template <class SlotMap>
inline void pack17(std::vector<float> &blob, int off, const std::vector<float> &W, int n_in, int n_out, int K_ROWS,
                   int NT, SlotMap slotmap) {
  for (int t = 0; t < NT; ++t)
    for (int rho = 0; rho < K_ROWS; ++rho) {
      const int src = slotmap(rho);
      for (int j = 0; j < 16; ++j) {
        const int o = 16 * t + j;
        float v = 0.0f;
        if (src >= 0 && src < n_in && o < n_out) v = W[(size_t)src * n_out + o];
        blob[off + (t * K_ROWS + rho) * 17 + j] = v;
      }
    }
}


void bgm_bgm_fit_free(bgm_handle *h);
