// causal_bx3_api.hip -- host side of the split-precision (bf16 x 3 / f16 x 3) CausalBGM sampling kernels (causal_bx3_kernels.h,
// compiled once per 16-bit operand format: namespaces bxb / bxh): packing of the Keras-order weights into hi / lo MFMA fragments,
// launchers, and the precision switch
// bgm_causal_set_precision (include/bgm_hip.h).  The fp32 path (causal_api.hip) is the default and is untouched.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "bgm_host.h"
#define BX_NS bxb
#define BX_F16 0
#include "causal_bx3_kernels.h"
#undef BX_NS
#undef BX_F16
#define BX_NS bxh
#define BX_F16 1
#include "causal_bx3_kernels.h"
#undef BX_NS
#undef BX_F16

static constexpr int BX_WAVES = 8;
#define BGM_BX3_VARIANTS(X) X(1, 13) X(1, 7) X(1, 2) X(2, 10) X(2, 7) X(2, 2)

// round-to-nearest-even fp32 -> bf16 (finite inputs)
static inline uint16_t bx_bf16_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bx_bf16_value(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// the same pair for fp16 (round to nearest even; |w| beyond 65504 saturates instead of overflowing to infinity)
static inline uint16_t bx_f16_bits(float f) {
  f = f > 65504.0f ? 65504.0f : (f < -65504.0f ? -65504.0f : f);
  const _Float16 h = (_Float16)f;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}
static inline float bx_f16_value(uint16_t b) {
  _Float16 h;
  std::memcpy(&h, &b, 2);
  return (float)h;
}
static inline uint16_t bx_bits(float f, bool f16) { return f16 ? bx_f16_bits(f) : bx_bf16_bits(f); }
static inline float bx_value(uint16_t b, bool f16) { return f16 ? bx_f16_value(b) : bx_bf16_value(b); }

// One layer's A fragments.  W: [n_in x n_out] row-major (Keras); feat(KT tile, lane group g, r) -> source input row of W
// (or -1 for a zero row); `col(o)` -> source output column of padded position o (or -1); `scale` multiplies every weight.
template <class Feat, class Col>
static void bx_pack_layer(std::vector<unsigned char> &blob, bool f16, int off, const float *W, int n_out_src, int KT, int NT, float scale,
                          Feat feat, Col col) {
  const int NK32 = KT / 2, K16 = KT & 1;
  const int tile_bytes = NK32 * 2048 + K16 * 1024;
  auto put = [&](size_t byte_off, float w) {   // hi at byte_off, lo at the caller-supplied distance
    const uint16_t hi = bx_bits(w, f16);
    std::memcpy(blob.data() + byte_off, &hi, 2);
    return hi;
  };
  for (int mt = 0; mt < NT; ++mt)
    for (int lane = 0; lane < 64; ++lane) {
      const int i = lane & 15, gk = lane >> 4;
      const int o = col(16 * mt + i);
      for (int T = 0; T < NK32; ++T)
        for (int u = 0; u < 8; ++u) {
          const int src = feat(2 * T + (u >> 2), gk, u & 3);
          const float w = (src >= 0 && o >= 0) ? W[(size_t)src * n_out_src + o] * scale : 0.0f;
          const size_t at = (size_t)off + (size_t)mt * tile_bytes + (size_t)T * 2048 + (size_t)lane * 16 + (size_t)u * 2;
          const uint16_t hi = put(at, w);
          const uint16_t lo = bx_bits(w - bx_value(hi, f16), f16);
          std::memcpy(blob.data() + at + 1024, &lo, 2);
        }
      if (K16)
        for (int u = 0; u < 4; ++u) {
          const int src = feat(KT - 1, gk, u);
          const float w = (src >= 0 && o >= 0) ? W[(size_t)src * n_out_src + o] * scale : 0.0f;
          const size_t at = (size_t)off + (size_t)mt * tile_bytes + (size_t)NK32 * 2048 + (size_t)lane * 8 + (size_t)u * 2;
          const uint16_t hi = put(at, w);
          const uint16_t lo = bx_bits(w - bx_value(hi, f16), f16);
          std::memcpy(blob.data() + at + 512, &lo, 2);
        }
    }
}

template <int KT1, int NTL>
static void bx_fill(bgm_handle *h, std::vector<unsigned char> &blob, BxMeta &m) {
  const bool f16 = h->precision == 2;
  using L = BxLayout<KT1, NTL>;
  const HostNet &G = h->nets[BGM_NET_G], &F = h->nets[BGM_NET_F], &H = h->nets[BGM_NET_H];
  const int q = h->q, p = h->p;
  const int z0 = h->cfg.z_dims[0], z1 = h->cfg.z_dims[1], z2 = h->cfg.z_dims[2];
  m.total_bytes = L::total(m.n_gh);
  blob.assign((size_t)m.total_bytes, 0);
  float *bf = reinterpret_cast<float *>(blob.data());
  auto ident_col = [](int o) { return o; };
  // first layers: input tile t, lane group g, register r  <->  extended input feature 16 t + 4 r + g of [z, x, 0 ...]
  auto l1 = [](int t, int g, int r) { return 16 * t + 4 * r + g; };
  // hidden layers: accumulator order, feature 16 t + 4 g + r
  auto acc_feat = [](int t, int g, int r) { return 16 * t + 4 * g + r; };
  bx_pack_layer(blob, f16, L::w1g, G.W(0), 64, KT1, 4, 1.0f, [&](int t, int g, int r) { const int f = l1(t, g, r); return f < q ? f : -1; }, ident_col);
  bx_pack_layer(blob, f16, L::w1f, F.W(0), 64, KT1, 4, 1.0f, [&](int t, int g, int r) {
    const int f = l1(t, g, r);
    if (f < z0 + z1) return f;
    if (f == q) return z0 + z1;          // treatment column
    return -1;
  }, ident_col);
  bx_pack_layer(blob, f16, L::w1h, H.W(0), 64, KT1, 4, 1.0f, [&](int t, int g, int r) {
    const int f = l1(t, g, r);
    if (f < z0) return f;
    if (f >= z0 + z1 && f < z0 + z1 + z2) return z0 + (f - z0 - z1);
    return -1;
  }, ident_col);
  for (int i = 0; i < 64; ++i) { bf[L::b1g + i] = G.b(0)[i]; bf[L::b1f + i] = F.b(0)[i]; bf[L::b1h + i] = H.b(0)[i]; }
  const float S = BGM_LRS_W;      // weights behind a one-instruction LeakyReLU (lrelu_s) carry the factor 0.6
  for (int l = 0; l < m.n_gh; ++l) {
    bx_pack_layer(blob, f16, L::wg(m.n_gh) + l * bx_layer_bytes(4, 4), G.W(1 + l), 64, 4, 4, S, acc_feat, ident_col);
    for (int i = 0; i < 64; ++i) bf[L::bg + 64 * l + i] = G.b(1 + l)[i];
  }
  {
    const int LG = (int)G.dims.size() - 2;
    std::vector<float> Wp, bp;
    bgm_g_last_padded(G.W(LG), G.b(LG), p, NTL, Wp, bp);
    bx_pack_layer(blob, f16, L::wgl, Wp.data(), 16 * NTL, 4, NTL, S, acc_feat, ident_col);
    for (int i = 0; i < 16 * NTL; ++i) bf[L::bgl + i] = bp[i];
  }
  auto rows_lt = [&](int n_in) { return [=](int t, int g, int r) { const int f = 16 * t + 4 * g + r; return f < n_in ? f : -1; }; };
  auto cols_lt = [&](int n_out) { return [=](int o) { return o < n_out ? o : -1; }; };
  auto rep2 = [](int o) { return (o & 3) < 2 ? (o & 3) : -1; };       // (mu, s) replicated at positions 4 g' + {0, 1}
  const HostNet *nets[2] = {&F, &H};
  const int w2[2] = {L::wf2, L::wh2}, w3[2] = {L::wf3, L::wh3}, w4[2] = {L::wf4, L::wh4};
  const int b2[2] = {L::bf2, L::bh2}, b3[2] = {L::bf3, L::bh3}, b4[2] = {L::bf4, L::bh4};
  for (int k = 0; k < 2; ++k) {
    const HostNet &N = *nets[k];
    bx_pack_layer(blob, f16, w2[k], N.W(1), 32, 4, 2, S, rows_lt(64), cols_lt(32));
    bx_pack_layer(blob, f16, w3[k], N.W(2), 8, 2, 1, S, rows_lt(32), cols_lt(8));
    bx_pack_layer(blob, f16, w4[k], N.W(3), 2, 1, 1, S, rows_lt(8), rep2);
    for (int i = 0; i < 32; ++i) bf[b2[k] + i] = N.b(1)[i];
    for (int i = 0; i < 8; ++i) bf[b3[k] + i] = N.b(2)[i];
    for (int i = 0; i < 16; ++i) bf[b4[k] + i] = (i & 3) < 2 ? N.b(3)[i & 3] : 0.0f;
  }
  for (int o = 0; o < 64; ++o) bf[L::wxf + o] = F.W(0)[(size_t)(z0 + z1) * 64 + o];
}

static int bx_pack(bgm_handle *h, std::vector<unsigned char> &blob, BxMeta &m) {
  const HostNet &G = h->nets[BGM_NET_G], &F = h->nets[BGM_NET_F], &H = h->nets[BGM_NET_H];
  const int q = h->q, p = h->p;
  int KT1, KSL1, NTL;
  if (!bgm_causal_shape(q + 1, p + 1, KT1, KSL1, NTL)) { bgm_set_error("bf16x3: no compiled kernel shape contains this model"); return BGM_E_UNSUPPORTED; }
  if (F.dims.size() != 5 || H.dims.size() != 5 || F.dims[1] != 64 || F.dims[2] != 32 || F.dims[3] != 8 || H.dims[1] != 64 ||
      H.dims[2] != 32 || H.dims[3] != 8) { bgm_set_error("bf16x3: f_units / h_units must be [64, 32, 8]"); return BGM_E_UNSUPPORTED; }
  for (size_t l = 1; l + 1 < G.dims.size(); ++l)
    if (G.dims[l] != 64) { bgm_set_error("bf16x3: g_units must be 64 wide"); return BGM_E_UNSUPPORTED; }
  std::memset(&m, 0, sizeof(m));
  m.q = q; m.p = p; m.binary = h->cfg.binary_treatment ? 1 : 0; m.sig_pc = p % 16;
  auto s2 = [](float s) { return s > 0.0f ? s * s : -1.0f; };
  m.sig2_v = s2(h->cfg.sigma_v); m.sig2_x = s2(h->cfg.sigma_x); m.sig2_y = s2(h->cfg.sigma_y);
  m.n_gh = h->cfg.n_hidden_g - 1;
  bool done = false;
#define X(KT1_, NTL_) if (KT1 == KT1_ && NTL == NTL_) { bx_fill<KT1_, NTL_>(h, blob, m); done = true; }
  BGM_BX3_VARIANTS(X)
#undef X
  if (!done) { bgm_set_error("bf16x3: no compiled kernel variant for this shape"); return BGM_E_UNSUPPORTED; }
  if (m.total_bytes + 64 > 160 * 1024) {
    bgm_set_error("bf16x3: model does not fit the 160 KiB LDS-resident layout (" + std::to_string(m.total_bytes) + " B)");
    return BGM_E_UNSUPPORTED;
  }
  h->KT1 = KT1; h->KSL1 = KSL1; h->NTL = NTL;
  return BGM_OK;
}

int bgm_causal_bx3_blob(bgm_handle *h, hipStream_t stream) {
  if (h->fit_active) { bgm_set_error("bf16x3 sampling inside an open fit session is not supported (close it with bgm_causal_fit_end)"); return BGM_E_STATE; }
  if (h->bx_valid) return BGM_OK;
  for (int id : {BGM_NET_G, BGM_NET_F, BGM_NET_H})
    if (!h->nets[id].set) { bgm_set_error("weights of g/f/h not all set"); return BGM_E_STATE; }
  std::vector<unsigned char> blob;
  BxMeta m;
  int rc = bx_pack(h, blob, m);
  if (rc) return rc;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (h->bx_cap < blob.size()) {
    if (h->bx_blob_dev) BGM_HIP_CHECK(hipFree(h->bx_blob_dev));
    BGM_HIP_CHECK(hipMalloc(&h->bx_blob_dev, blob.size()));
    h->bx_cap = blob.size();
  }
  BGM_HIP_CHECK(hipMemcpyAsync(h->bx_blob_dev, blob.data(), blob.size(), hipMemcpyHostToDevice, stream));
  BGM_HIP_CHECK(hipStreamSynchronize(stream));
  static_assert(sizeof(BxMeta) <= sizeof(h->bx_meta_store), "bx_meta_store too small");
  std::memcpy(h->bx_meta_store, &m, sizeof(m));
  h->bx_valid = true;
  return BGM_OK;
}

extern "C" int bgm_causal_set_precision(bgm_handle *h, int32_t mode) {
  if (!h || mode < 0 || mode > 2) { bgm_set_error("bgm_causal_set_precision: mode must be 0 (fp32), 1 (bf16x3) or 2 (f16x3)"); return BGM_E_INVALID; }
  if (mode != h->precision) { h->bx_valid = false; h->gx_valid = false; }      // the packed blob is per operand format; the general-width engine builds its split pack when the mode is on
  h->precision = mode;
  return BGM_OK;
}

template <class K>
static int bx_set_lds(K kernel, int bytes) {
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return BGM_OK;
}

int bgm_causal_bx3_logpost(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n, float *out,
                           int grid, hipStream_t stream) {
  int rc = bgm_causal_bx3_blob(h, stream);
  if (rc) return rc;
  BxMeta m;
  std::memcpy(&m, h->bx_meta_store, sizeof(m));
  const int lds = m.total_bytes;
#define X(KT1_, NTL_)                                                                                              \
  if (h->KT1 == KT1_ && h->NTL == NTL_) {                                                                          \
    auto k = h->precision == 2 ? bxh::causal_logpost_bx3_kernel<KT1_, NTL_, BX_WAVES> : bxb::causal_logpost_bx3_kernel<KT1_, NTL_, BX_WAVES>; \
    rc = bx_set_lds(k, lds);                                                                                       \
    if (rc) return rc;                                                                                             \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * BX_WAVES), lds, stream, (const unsigned char *)h->bx_blob_dev, m, x, y, v, z, \
                       (long long)n, out, (const int *)h->prior_seg, (const float *)h->prior_tab);                 \
    BGM_HIP_CHECK(hipGetLastError());                                                                              \
    return BGM_OK;                                                                                                 \
  }
  BGM_BX3_VARIANTS(X)
#undef X
  bgm_set_error("bf16x3: no compiled kernel variant for this shape");
  return BGM_E_UNSUPPORTED;
}

template <int EFFECT>
static int bx_launch_mh(bgm_handle *h, const CausalBxKArgs &ka, int grid, int lds, hipStream_t stream) {
  int rc;
#define X(KT1_, NTL_)                                                                          \
  if (h->KT1 == KT1_ && h->NTL == NTL_) {                                                      \
    auto k = h->precision == 2 ? bxh::causal_mh_bx3_kernel<KT1_, NTL_, BX_WAVES, EFFECT> : bxb::causal_mh_bx3_kernel<KT1_, NTL_, BX_WAVES, EFFECT>; \
    rc = bx_set_lds(k, lds);                                                                   \
    if (rc) return rc;                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * BX_WAVES), lds, stream, ka);                   \
    BGM_HIP_CHECK(hipGetLastError());                                                          \
    return BGM_OK;                                                                             \
  }
  BGM_BX3_VARIANTS(X)
#undef X
  bgm_set_error("bf16x3: no compiled MH kernel variant for this shape");
  return BGM_E_UNSUPPORTED;
}

// one launch of the split-precision MH kernel with the fp32 kernel's argument block (effect: 0 none, 1 ADRF, 2 ITE)
int bgm_causal_bx3_mh_launch(bgm_handle *h, const CausalMhKArgs &a, int effect, int grid, hipStream_t stream) {
  int rc = bgm_causal_bx3_blob(h, stream);
  if (rc) return rc;
  CausalBxKArgs ka{};
  ka.a = a;
  ka.a.seg = (const int *)h->prior_seg;            // conditional prior (bgm_causal_set_prior) or NULL
  ka.a.prior_tab = h->prior_tab;
  ka.bblob = (const unsigned char *)h->bx_blob_dev;
  std::memcpy(&ka.bx, h->bx_meta_store, sizeof(BxMeta));
  const int lds = ka.bx.total_bytes + 64;
  if (effect == 3) return bx_launch_mh<3>(h, ka, grid, lds, stream);        // event form of the retained phase (causal_event_api.hip)
  if (effect == BGM_EFFECT_ADRF) return bx_launch_mh<1>(h, ka, grid, lds, stream);
  if (effect == BGM_EFFECT_ITE) return bx_launch_mh<2>(h, ka, grid, lds, stream);
  return bx_launch_mh<0>(h, ka, grid, lds, stream);
}
