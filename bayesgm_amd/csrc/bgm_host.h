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
  // encoder blob
  float *eblob_dev = nullptr;
  size_t eblob_cap = 0;
  bool eblob_valid = false;
  // fit state (device)
  bool fit_active = false;
  long long fit_rows = 0;
  int fit_bcap = 0, n_params = 0, n_slices_cap = 0, rows_per_slice = 0;
  long long t_theta = 0, t_z = 0;
  float *theta_dev = nullptr, *m1_dev = nullptr, *m2_dev = nullptr, *bblob_dev = nullptr, *ws_dev = nullptr,
        *partial_dev = nullptr;
  int *tables_dev = nullptr;  // fwd_dst | fwd_dst2 | bwd_dst | grad_src, n_params each
  int *pos_dev = nullptr;
  FitMeta fit_meta{};
  FitWs fit_ws{};
  DwArgs dw{};
  unsigned *acc_scratch = nullptr;   // [n_slots x n_iters] per-launch acceptance counters
  size_t acc_scratch_cap = 0;
  void *bgm_state = nullptr;  // BgmState (bgm_api.hip)
  void *egm_state = nullptr;  // EgmState (egm_api.hip)
  void *bgm_egm_state = nullptr;  // BgmEgmState (bgm_egm_api.hip)
  // timing
  bool timing = false;
  struct Ev { hipEvent_t a, b; int kind; };
  std::vector<Ev> events;
  long long timed_launches[3] = {0, 0, 0};
  double timed_ms[3] = {0.0, 0.0, 0.0};
};

int bgm_causal_build_blob(bgm_handle *h, hipStream_t stream);
void bgm_bgm_free_state(bgm_handle *h);
void bgm_egm_free_state(bgm_handle *h);
void bgm_bgm_egm_free_state(bgm_handle *h);
int causal_pack_forward(bgm_handle *h, const HostNet &G, const HostNet &F, const HostNet &H, std::vector<float> &blob);

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


