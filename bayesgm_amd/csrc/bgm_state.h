// bgm_state.h -- host-side state of the BGM path shared by bgm_api.hip and fit_api.hip.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bgm_host.h"
#include "bgm_kernels.h"
#include "bgm_fit_kernels.h"

struct BgmState {
  bgm_bgm_config cfg{};
  bool configured = false, set = false;
  std::vector<float> theta;
  BgmMeta meta{};
  int KTQ = 0, NTX = 0, NH = 0;   // NTX = 0: wide variant (head weights streamed, any x_dim)
  int fit_NTX = 0;                // the minibatch kernels' variant (its own field: a split-precision posterior blob is always the streamed variant)
  int lds_bytes = 0;              // dynamic LDS of the posterior kernels
  int fit_lds_bytes = 0;
  float *blob_dev = nullptr;
  size_t blob_cap = 0;
  bool blob_valid = false;
  // split precision (bgm_bgm_set_precision; bgm_kernels.h "Split precision"): 0 fp32 (default), 2 f16x3 -- the whole generator as one
  // linear stream of 32 KiB steps of fp16 fragments [L1 | hidden forward | head steps | hidden backward | L1], the biases as the only
  // LDS-resident data
  int precision = 0;
  unsigned char *sx3_dev = nullptr;
  size_t sx3_cap = 0;
  float *sx3_bias_dev = nullptr;
  size_t sx3_bias_cap = 0;
  BgmMeta sx3_meta{};
  int lds_bytes_sx3 = 0;
  // fit session (device)
  bool fit_active = false;
  int fit_bcap = 0, n_params = 0, rows_per_slice = 256, n_slices_cap = 0;
  int batch_global = 0;   // > 0: losses are means over this many rows (data-parallel fit); 0: the local batch
  long long t_theta = 0, t_z = 0;
  float *theta_dev = nullptr, *m1_dev = nullptr, *m2_dev = nullptr, *tblob_dev = nullptr, *ws_dev = nullptr,
        *partial_dev = nullptr, *bn_dev = nullptr;
  float *epoch_grad_dev = nullptr;      // bgm_bgm_fit_epoch: the gradient between _theta_grad and _theta_apply
  float *split_part_dev = nullptr;      // small-minibatch passes (BgmFitKArgs::part): [8 * BGM_FIT_S][32][64] floats + one counter word
  int *tables_dev = nullptr;   // dst | dst2(-1) | bwd(-1) | grad_src
  BgmMeta tmeta{};             // training blob layout (same offsets as meta)
  BgmFitWs fit_ws{};
  DwArgs dw{};
  // general-width engine (gx_bgm_api.hip): trunk widths other than [64] x {3, 5}, z_dim > 16
  void *gx = nullptr;
  bool gx_valid = false, gx_fit = false;
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


// head weights, pair-contiguous: [tx][head][64 in][17]
inline void pack17_heads(std::vector<float> &blob, int off, const float *W, int p, int ntx, int head) {
  for (int tx = 0; tx < ntx; ++tx)
    for (int rho = 0; rho < 64; ++rho)
      for (int j = 0; j < 16; ++j) {
        const int o = 16 * tx + j;
        blob[off + tx * BGM_PAIR + head * 64 * 17 + rho * 17 + j] = (o < p) ? W[(size_t)rho * p + o] : 0.0f;
      }
}

// Blob layout shared by the inference and the training blob.  Chooses the LDS-resident variant when the
// whole generator fits the 160 KiB LDS AND a kernel is compiled for ceil(p/16); otherwise the wide variant.
// Returns the dynamic LDS bytes, or -1 if not even the trunk + stage fits.
inline int bgm_layout(int q, int p, int NH, BgmMeta &m, int &ntx_variant, bool force_wide = false) {
  const int KTQ = (q + 15) / 16, NTX = (p + 15) / 16;
  m = BgmMeta{};
  m.q = q; m.p = p; m.n_hh = NH - 1; m.ntx = NTX;
  int off = 0;
  auto take = [&](int n) { int o = off; off += (n + 3) / 4 * 4; return o; };
  m.w1 = take(4 * 16 * KTQ * 17); m.b1 = take(64);
  m.wh = take(m.n_hh * 4 * 64 * 17); m.bh = take(m.n_hh * 64);
  m.bhd = take(2 * 16 * NTX);
  m.whd = take(NTX * BGM_PAIR);
  m.total = off;
  const char *force = std::getenv("BGM_FORCE_WIDE");   // test hook: run the wide variant on small shapes too
  const bool resident = (size_t)m.total * 4 <= 160 * 1024 && (NTX == 2 || NTX == 7) && !(force && force[0] == '1') && !force_wide;
  if (resident) {
    ntx_variant = NTX; m.lds_resident = m.total; m.stage = 0;
    return m.total * 4;
  }
  ntx_variant = 0; m.lds_resident = m.whd; m.stage = m.whd;
  const size_t bytes = ((size_t)m.whd + 2 * BGM_PAIR) * 4;
  return bytes <= 160 * 1024 ? (int)bytes : -1;
}

// fp32 -> fp16 bits, round to nearest even, clamped to the largest finite value (host-side packing of the split-precision fragments)
inline unsigned short bgm_f2h(float f) {
  unsigned x;
  std::memcpy(&x, &f, 4);
  const unsigned sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x477ff000u) return (unsigned short)(sign | 0x7bffu);          // >= 65520 (or inf / nan): 65504
  if (x < 0x33000001u) return (unsigned short)sign;                       // < 2^-25: zero
  if (x < 0x38800000u) {                                                  // subnormal half
    const int e = (int)(x >> 23);                                         // biased fp32 exponent, 102 .. 112
    const unsigned m = (x & 0x7fffffu) | 0x800000u;
    const int sh = 126 - e;                                               // 14 .. 24
    unsigned r = m >> sh;
    const unsigned rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
    if (rem > half || (rem == half && (r & 1u))) ++r;
    return (unsigned short)(sign | r);
  }
  unsigned r = ((x - 0x38000000u) >> 13);
  const unsigned rem = x & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
  return (unsigned short)(sign | r);
}
inline float bgm_h2f(unsigned short h) {
  const unsigned sign = (unsigned)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  float f;
  if (e == 0) f = std::ldexp((float)m, -24);
  else { const unsigned x = ((e + 112u) << 23) | (m << 13); std::memcpy(&f, &x, 4); }
  if (sign) f = -f;
  return f;
}

void bgm_bgm_fit_free(bgm_handle *h);
