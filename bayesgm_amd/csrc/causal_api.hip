// causal_api.hip -- C-ABI entry points of the CausalBGM posterior-sampling path
// (declared in include/bgm_hip.h) + host-side packing of the Keras-order weights
// into the LDS fragment order of causal_kernels.h.
#include <algorithm>
#include <string>
#include <vector>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>

#include "bgm_host.h"
#include "bnf_det_host.h"
#include "gx_host.h"

static thread_local std::string g_err;
void bgm_set_error(const std::string &msg) { g_err = msg; }

extern "C" const char *bgm_last_error(void) { return g_err.c_str(); }
extern "C" const char *bgm_version(void) { return "bayesgm_amd-hip 0.1 (gfx950)"; }

extern "C" int bgm_create(bgm_handle **out, int device) {
  if (!out) { bgm_set_error("bgm_create: out == NULL"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  BGM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
    bgm_set_error(std::string("bgm_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    return BGM_E_UNSUPPORTED;
  }
  bgm_handle *h = new bgm_handle();
  h->device = device;
  h->n_cus = prop.multiProcessorCount;
  *out = h;
  return BGM_OK;
}

extern "C" int bgm_set_disc_norm(bgm_handle *h, int32_t mode) {
  if (!h || (mode != 0 && mode != 1)) { bgm_set_error("bgm_set_disc_norm: mode must be 0 (batch statistics) or 1 (inference mode)"); return BGM_E_INVALID; }
  h->disc_norm = mode;
  return BGM_OK;
}

extern "C" int bgm_destroy(bgm_handle *h) {
  if (!h) return BGM_OK;
  hipSetDevice(h->device);
  if (h->blob_dev) hipFree(h->blob_dev);
  if (h->eblob_dev) hipFree(h->eblob_dev);
  if (h->sblob_dev) hipFree(h->sblob_dev);
  if (h->bx_blob_dev) hipFree(h->bx_blob_dev);
  if (h->acc_scratch) hipFree(h->acc_scratch);
  if (h->eff_cache) hipFree(h->eff_cache);
  if (h->eff_stats_dev) hipFree(h->eff_stats_dev);
  bgm_causal_event_free(h);
  bnf_det_free(h);
  bgm_causal_fit_end(h, nullptr);
  gx_free(h);
  if (h->epoch_ctr) { hipFree(h->epoch_ctr); h->epoch_ctr = nullptr; }
  if (h->epoch_stream) {
    hipStreamDestroy(h->epoch_stream);
    for (int k = 0; k < 4; ++k) { if (h->epoch_ev_t[k]) hipEventDestroy(h->epoch_ev_t[k]); if (h->epoch_ev_z[k]) hipEventDestroy(h->epoch_ev_z[k]); }
    if (h->epoch_ev_s) hipEventDestroy(h->epoch_ev_s);
  }
  bgm_bgm_free_state(h);
  bgm_egm_free_state(h);
  bgm_bgm_egm_free_state(h);
  bgm_bnn_free_state(h);
  bgm_bvn_free_state(h);
  for (auto &e : h->events) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  delete h;
  return BGM_OK;
}

// ---------------------------------------------------------------------------
// configuration
// ---------------------------------------------------------------------------
static bool units_are(const int32_t *u, int n, std::initializer_list<int> want) {
  if (n != (int)want.size()) return false;
  int i = 0;
  for (int w : want) if (u[i++] != w) return false;
  return true;
}

extern "C" int bgm_causal_configure(bgm_handle *h, const bgm_causal_config *cfg) {
  if (!h || !cfg) { bgm_set_error("bgm_causal_configure: NULL argument"); return BGM_E_INVALID; }
  const int q = cfg->z_dims[0] + cfg->z_dims[1] + cfg->z_dims[2] + cfg->z_dims[3];
  const int p = cfg->v_dim;
  if (p < 1 || q < 1 || cfg->z_dims[0] < 0 || cfg->z_dims[1] < 0 || cfg->z_dims[2] < 0 || cfg->z_dims[3] < 0) {
    bgm_set_error("bgm_causal_configure: bad v_dim / z_dims"); return BGM_E_INVALID;
  }
  for (int n : {cfg->n_hidden_g, cfg->n_hidden_f, cfg->n_hidden_h, cfg->n_hidden_e})
    if (n < 1 || n > BGM_MAX_LAYERS) { bgm_set_error("bgm_causal_configure: hidden layer count out of range"); return BGM_E_INVALID; }
  // Any hidden widths / depths (networks/base.py:7-28 takes any nb_units): the reference defaults (g_units [64]*k, f_units = h_units =
  // [64,32,8]; configs/*.yaml, cli/cli.py) run on the specialised kernel families, every other shape on the general-width engine
  // (gx_api.hip).
  for (const int32_t *u : {cfg->g_units, cfg->f_units, cfg->h_units, cfg->e_units})
    for (int i = 0; i < BGM_MAX_LAYERS; ++i)
      if (u[i] < 0 || u[i] > 4096) { bgm_set_error("bgm_causal_configure: hidden widths must be in [1, 4096]"); return BGM_E_INVALID; }
  for (int i = 0; i < cfg->n_hidden_g; ++i) if (cfg->g_units[i] < 1) { bgm_set_error("bgm_causal_configure: g_units must be positive"); return BGM_E_INVALID; }
  for (int i = 0; i < cfg->n_hidden_f; ++i) if (cfg->f_units[i] < 1) { bgm_set_error("bgm_causal_configure: f_units must be positive"); return BGM_E_INVALID; }
  for (int i = 0; i < cfg->n_hidden_h; ++i) if (cfg->h_units[i] < 1) { bgm_set_error("bgm_causal_configure: h_units must be positive"); return BGM_E_INVALID; }
  for (int i = 0; i < cfg->n_hidden_e; ++i) if (cfg->e_units[i] < 1) { bgm_set_error("bgm_causal_configure: e_units must be positive"); return BGM_E_INVALID; }
  h->cfg = *cfg;
  if (q + 1 > 32 && !gx_wanted(h)) { bgm_set_error("bgm_causal_configure: sum(z_dims) > 31 with the default hidden widths is not compiled"); h->configured = false; return BGM_E_UNSUPPORTED; }
  h->cfg = *cfg;
  h->q = q;
  h->p = p;
  auto mk = [](HostNet &n, int in, const int32_t *units, int nh, int out) {
    n.dims.clear();
    n.dims.push_back(in);
    for (int i = 0; i < nh; ++i) n.dims.push_back(units[i]);
    n.dims.push_back(out);
    n.theta.assign(n.count(), 0.0f);
    n.set = false;
  };
  mk(h->nets[BGM_NET_G], q, cfg->g_units, cfg->n_hidden_g, p + 1);
  mk(h->nets[BGM_NET_F], cfg->z_dims[0] + cfg->z_dims[1] + 1, cfg->f_units, cfg->n_hidden_f, 2);
  mk(h->nets[BGM_NET_H], cfg->z_dims[0] + cfg->z_dims[2], cfg->h_units, cfg->n_hidden_h, 2);
  mk(h->nets[BGM_NET_E], p, cfg->e_units, cfg->n_hidden_e, q);
  bnf_det_free(h);
  gx_free(h);
  h->configured = true;
  h->blob_valid = false;
  h->bx_valid = false;
  h->eblob_valid = false;
  return BGM_OK;
}

extern "C" int bgm_causal_set_weights(bgm_handle *h, int net_id, const float *theta_host, int64_t count,
                                      void *stream) {
  (void)stream;
  if (!h || !h->configured) { bgm_set_error("bgm_causal_set_weights: handle not configured"); return BGM_E_STATE; }
  if (net_id < 0 || net_id > 3 || !theta_host) { bgm_set_error("bgm_causal_set_weights: bad net_id / NULL"); return BGM_E_INVALID; }
  HostNet &n = h->nets[net_id];
  if ((size_t)count != n.count()) {
    bgm_set_error("bgm_causal_set_weights: expected " + std::to_string(n.count()) + " floats, got " + std::to_string(count));
    return BGM_E_INVALID;
  }
  std::memcpy(n.theta.data(), theta_host, sizeof(float) * count);
  n.set = true;
  if (net_id == BGM_NET_E) h->eblob_valid = false; else { h->blob_valid = false; h->bx_valid = false; h->det_valid = false; }
  h->gx_valid = false;
  return BGM_OK;
}

// ---------------------------------------------------------------------------
// packing (layout documented in bgm_device.h)
// ---------------------------------------------------------------------------
// Lay out and fill the forward (LDS) blob from three HostNets (g, f, h).  Also used with
// "iota" nets to derive the canonical-parameter -> blob-position tables of the fit path.
int causal_pack_forward(bgm_handle *h, const HostNet &G, const HostNet &F, const HostNet &H,
                        std::vector<float> &blob) {
  const int q = h->q, p = h->p;
  const int z0 = h->cfg.z_dims[0], z1 = h->cfg.z_dims[1], z2 = h->cfg.z_dims[2];
  const int q1 = q + 1;
  int KT1, KSL1, NTL;
  if (!bgm_causal_shape(q1, p + 1, KT1, KSL1, NTL)) {
    bgm_set_error("no compiled kernel shape contains this model (needs sum(z_dims) <= 19 and v_dim <= 207, or <= 159 when sum(z_dims) > 11)");
    return BGM_E_UNSUPPORTED;
  }
  h->KT1 = KT1; h->KSL1 = KSL1; h->NTL = NTL;
  CausalMeta &m = h->meta;
  std::memset(&m, 0, sizeof(m));
  m.q = q; m.p = p; m.binary = h->cfg.binary_treatment ? 1 : 0;
  m.sig_pc = p % 16; m.sig_slot = bgm_sig_slot(p, NTL);
  auto s2 = [](float s) { return s > 0.0f ? s * s : -1.0f; };
  m.sig2_v = s2(h->cfg.sigma_v); m.sig2_x = s2(h->cfg.sigma_x); m.sig2_y = s2(h->cfg.sigma_y);
  m.n_gh = h->cfg.n_hidden_g - 1;
  int off = 0;
  auto take = [&](int n) { int o = off; off += (n + 3) / 4 * 4; return o; };
  m.w1g = take(16 * KT1 * 64); m.w1f = take(16 * KT1 * 64); m.w1h = take(16 * KT1 * 64);
  m.b1g = take(64); m.b1f = take(64); m.b1h = take(64);
  m.wg = take(m.n_gh * 4096); m.bg = take(m.n_gh * 64);
  m.wgl = take(64 * 16 * NTL); m.bgl = take(16 * NTL);
  m.wf2 = take(64 * 32); m.bf2 = take(32); m.wf3 = take(32 * 16); m.bf3 = take(16); m.wf4 = take(16 * 16); m.bf4 = take(16);
  m.wh2 = take(64 * 32); m.bh2 = take(32); m.wh3 = take(32 * 16); m.bh3 = take(16); m.wh4 = take(16 * 16); m.bh4 = take(16);
  m.wxf = take(64);
  m.total = off;
  if ((size_t)m.total * 4 > 160 * 1024) {
    bgm_set_error("model does not fit the 160 KiB LDS-resident layout (" + std::to_string(m.total * 4) + " B); p <= 207 with default widths");
    return BGM_E_UNSUPPORTED;
  }
  blob.assign(m.total, 0.0f);
  auto ident = [](int rho) { return rho; };
  pack_layer(blob, m.w1g, G.W(0), q, 64, KT1, 4, [&](int rho) { int f = l1_feature(rho); return f < q ? f : -1; });
  pack_layer(blob, m.w1f, F.W(0), z0 + z1 + 1, 64, KT1, 4, [&](int rho) {
    int f = l1_feature(rho);
    if (f < z0 + z1) return f;
    if (f == q) return z0 + z1;  // treatment column
    return -1;
  });
  pack_layer(blob, m.w1h, H.W(0), z0 + z2, 64, KT1, 4, [&](int rho) {
    int f = l1_feature(rho);
    if (f < z0) return f;
    if (f >= z0 + z1 && f < z0 + z1 + z2) return z0 + (f - z0 - z1);
    return -1;
  });
  pack_bias(blob, m.b1g, G.b(0), 64, 4); pack_bias(blob, m.b1f, F.b(0), 64, 4); pack_bias(blob, m.b1h, H.b(0), 64, 4);
  for (int l = 0; l < m.n_gh; ++l) {
    pack_layer(blob, m.wg + l * 4096, G.W(1 + l), 64, 64, 4, 4, ident);
    pack_bias(blob, m.bg + l * 64, G.b(1 + l), 64, 4);
  }
  const int LG = (int)G.dims.size() - 2;  // index of last layer
  {
    std::vector<float> Wp, bp;
    bgm_g_last_padded(G.W(LG), G.b(LG), p, NTL, Wp, bp);
    pack_layer(blob, m.wgl, Wp.data(), 64, 16 * NTL, 4, NTL, ident);
    pack_bias(blob, m.bgl, bp.data(), 16 * NTL, NTL);
  }
  pack_layer(blob, m.wf2, F.W(1), 64, 32, 4, 2, ident); pack_bias(blob, m.bf2, F.b(1), 32, 2);
  pack_layer(blob, m.wf3, F.W(2), 32, 8, 2, 1, ident); pack_bias(blob, m.bf3, F.b(2), 8, 1);
  pack_layer(blob, m.wf4, F.W(3), 8, 2, 1, 1, ident); pack_bias(blob, m.bf4, F.b(3), 2, 1);
  pack_layer(blob, m.wh2, H.W(1), 64, 32, 4, 2, ident); pack_bias(blob, m.bh2, H.b(1), 32, 2);
  pack_layer(blob, m.wh3, H.W(2), 32, 8, 2, 1, ident); pack_bias(blob, m.bh3, H.b(2), 8, 1);
  pack_layer(blob, m.wh4, H.W(3), 8, 2, 1, 1, ident); pack_bias(blob, m.bh4, H.b(3), 2, 1);
  for (int o = 0; o < 64; ++o) blob[m.wxf + o] = F.W(0)[(size_t)(z0 + z1) * 64 + o];
  return BGM_OK;
}

// sampling copy of the forward blob: dst = src with the weights of the layers behind a LeakyReLU scaled by 0.6, and the tails of
// the two small nets (f, h: ... -> 8 -> 2) re-laid out for the sampling kernels (causal_kernels.h, above causal_effects):
//   layer 3 [32 x 8]: output column f moves to tile position pos(f) = 4 (f >> 1) + (f & 1) (weights and bias), so that the 8
//                     activations sit in accumulator registers 0, 1 of every lane group;
//   layer 4 [8 x 2] : input row f moves to K row pos(f) (K-steps 0 and 1 only), and the two output columns (mu, s) are
//                     replicated at positions 4 g' + {0, 1} (weights and bias): every lane group receives (mu, s).
static __device__ __forceinline__ int tail_pos_to_feature(int pos) { return ((pos & 3) < 2) ? 2 * (pos >> 2) + (pos & 1) : -1; }
static __global__ void causal_scale_blob_kernel(const float *src, float *dst, CausalMeta m, int NTL) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.total) return;
  auto in = [&](int off, int n) { return i >= off && i < off + n; };
  const int w3[2] = {m.wf3, m.wh3}, b3[2] = {m.bf3, m.bh3}, w4[2] = {m.wf4, m.wh4}, b4[2] = {m.bf4, m.bh4};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (in(w3[k], 32 * 16)) {            // [K row rho][tile position]
      const int rho = (i - w3[k]) >> 4, f = tail_pos_to_feature((i - w3[k]) & 15);
      dst[i] = f >= 0 ? src[w3[k] + rho * 16 + f] * BGM_LRS_W : 0.0f;
      return;
    }
    if (in(b3[k], 16)) {
      const int f = tail_pos_to_feature(i - b3[k]);
      dst[i] = f >= 0 ? src[b3[k] + f] : 0.0f;
      return;
    }
    if (in(w4[k], 16 * 16)) {
      const int f = tail_pos_to_feature((i - w4[k]) >> 4), c = (i - w4[k]) & 3;
      dst[i] = (f >= 0 && c < 2) ? src[w4[k] + f * 16 + c] * BGM_LRS_W : 0.0f;
      return;
    }
    if (in(b4[k], 16)) {
      const int c = (i - b4[k]) & 3;
      dst[i] = c < 2 ? src[b4[k] + c] : 0.0f;
      return;
    }
  }
  const bool scaled = in(m.wg, m.n_gh * 4096) || in(m.wgl, 64 * 16 * NTL) || in(m.wf2, 64 * 32) || in(m.wh2, 64 * 32);
  dst[i] = scaled ? src[i] * BGM_LRS_W : src[i];
}

int bgm_causal_sampling_blob(bgm_handle *h, hipStream_t stream) {
  if (!h->blob_valid) h->sblob_valid = false;
  int rc = bgm_causal_build_blob(h, stream);
  if (rc) return rc;
  if (h->sblob_valid) return BGM_OK;
  const size_t n = (size_t)h->meta.total;
  if (h->sblob_cap < n) {
    if (h->sblob_dev) BGM_HIP_CHECK(hipFree(h->sblob_dev));
    BGM_HIP_CHECK(hipMalloc(&h->sblob_dev, n * sizeof(float)));
    h->sblob_cap = n;
  }
  hipLaunchKernelGGL(causal_scale_blob_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, h->blob_dev, h->sblob_dev,
                     h->meta, h->NTL);
  BGM_HIP_CHECK(hipGetLastError());
  h->sblob_valid = true;
  return BGM_OK;
}

int bgm_causal_build_blob(bgm_handle *h, hipStream_t stream) {
  if (h->blob_valid) return BGM_OK;
  h->sblob_valid = false;
  for (int id : {BGM_NET_G, BGM_NET_F, BGM_NET_H})
    if (!h->nets[id].set) { bgm_set_error("weights of g/f/h not all set"); return BGM_E_STATE; }
  std::vector<float> blob;
  int rc = causal_pack_forward(h, h->nets[BGM_NET_G], h->nets[BGM_NET_F], h->nets[BGM_NET_H], blob);
  if (rc) return rc;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (h->blob_cap < blob.size()) {
    if (h->blob_dev) BGM_HIP_CHECK(hipFree(h->blob_dev));
    BGM_HIP_CHECK(hipMalloc(&h->blob_dev, blob.size() * sizeof(float)));
    h->blob_cap = blob.size();
  }
  BGM_HIP_CHECK(hipMemcpyAsync(h->blob_dev, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  BGM_HIP_CHECK(hipStreamSynchronize(stream));  // blob is a stack-local staging buffer
  h->blob_valid = true;
  return BGM_OK;
}

// ---------------------------------------------------------------------------
// kernel variants.  (KT1, KSL1, NTL): first-layer K tiling and number of
// 16-wide output tiles of g's last layer.
//   (1,3,13): z_dims [1,1,1,7], p = 200   (configs/Sim_Hirano_Imbens.yaml)
//   (2,1, 7): z_dims [3,3,6,6], p = 100   (cli/cli.py defaults)
//   (1,3, 2): z_dims [1,1,1,7], p <= 31   (small panels / tests)
//   (2,1, 2): z_dims [3,3,6,6], p <= 31
// ---------------------------------------------------------------------------
#ifndef BGM_MH_R
#define BGM_MH_R 1
#endif
#ifndef BGM_MH_WAVES
#define BGM_MH_WAVES 8
#endif
static constexpr int MH_R = BGM_MH_R, MH_WAVES = BGM_MH_WAVES;

#define BGM_CAUSAL_VARIANTS(X) X(1, 3, 13) X(1, 3, 7) X(1, 3, 2) X(2, 1, 10) X(2, 1, 7) X(2, 1, 2)

template <class K>
static int set_lds(K kernel, int bytes) {
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return BGM_OK;
}

static int mh_grid(const bgm_handle *h, int64_t n) {
  const int64_t tiles = (n + 16 * MH_R - 1) / (16 * MH_R);
  const int64_t blocks = (tiles + MH_WAVES - 1) / MH_WAVES;
  return (int)std::max<int64_t>(1, std::min<int64_t>(blocks, h->n_cus));
}

extern "C" int bgm_causal_mh_slots(bgm_handle *h, int64_t n, int32_t *n_slots) {
  if (!h || !n_slots) { bgm_set_error("bgm_causal_mh_slots: NULL"); return BGM_E_INVALID; }
  if (h->configured && gx_wanted(h)) { *n_slots = gx_slots(h, n); return BGM_OK; }            // general-width engine: one slot per workgroup
  if (h->configured && bnf_det_wanted(h)) { *n_slots = bnf_det_slots(h); return BGM_OK; }      // general path: one slot per workgroup
  *n_slots = mh_grid(h, n) * MH_WAVES;
  return BGM_OK;
}

extern "C" int bgm_causal_logpost(bgm_handle *h, const float *x, const float *y, const float *v, const float *z,
                                  int64_t n, float *out, void *stream_) {
  if (!h || !h->configured) { bgm_set_error("bgm_causal_logpost: handle not configured"); return BGM_E_STATE; }
  if (n <= 0) return BGM_OK;
  if (!x || !y || !v || !z || !out) { bgm_set_error("bgm_causal_logpost: NULL pointer"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (gx_wanted(h)) {          // hidden widths / depths outside the compiled families: the general-width engine (gx_api.hip)
    return gx_logpost(h, x, y, v, z, n, out, stream);      // (split precision: 'f16x3' on its row-tile-per-wave kernels, else it says so)
  }
  if (bnf_det_wanted(h)) {     // no LDS-resident compiled shape holds the model: the streamed-fragment kernels (bnf_det_api.hip)
    if (h->prior_seg || h->precision != 0) { bgm_set_error("bgm_causal_logpost: the conditional prior / split precision exist for the LDS-resident shapes only"); return BGM_E_UNSUPPORTED; }
    return bnf_det_logpost(h, x, y, v, z, n, out, stream);
  }
  int rc = bgm_causal_sampling_blob(h, stream);
  if (rc) return rc;
  const int grid = mh_grid(h, n);
  if (h->precision != 0) return bgm_causal_bx3_logpost(h, x, y, v, z, n, out, grid, stream);      // (carries the conditional prior, if one is set)
  if (h->prior_seg) return bgm_causal_prior_logpost(h, x, y, v, z, n, out, grid, stream);
  const int lds = h->meta.total * 4;
#define X(KT1_, KSL1_, NTL_)                                                                   \
  if (h->KT1 == KT1_ && h->KSL1 == KSL1_ && h->NTL == NTL_) {                                  \
    auto k = causal_logpost_kernel<KT1_, KSL1_, NTL_, MH_R, MH_WAVES>;                         \
    rc = set_lds(k, lds);                                                                      \
    if (rc) return rc;                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * MH_WAVES), lds, stream, h->sblob_dev, h->meta,  \
                       x, y, v, z, (long long)n, out, (const int *)nullptr, (const float *)nullptr); \
    BGM_HIP_CHECK(hipGetLastError());                                                          \
    return BGM_OK;                                                                             \
  }
  BGM_CAUSAL_VARIANTS(X)
#undef X
  bgm_set_error("no compiled kernel variant for (KT1,KSL1,NTL)=(" + std::to_string(h->KT1) + "," +
                std::to_string(h->KSL1) + "," + std::to_string(h->NTL) + ")");
  return BGM_E_UNSUPPORTED;
}

template <int EFFECT>
static int launch_mh(bgm_handle *h, const CausalMhKArgs &ka, int grid, int lds, hipStream_t stream) {
  int rc;
#define X(KT1_, KSL1_, NTL_)                                                                   \
  if (h->KT1 == KT1_ && h->KSL1 == KSL1_ && h->NTL == NTL_) {                                  \
    auto k = causal_mh_kernel<KT1_, KSL1_, NTL_, MH_R, MH_WAVES, EFFECT>;                      \
    rc = set_lds(k, lds);                                                                      \
    if (rc) return rc;                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * MH_WAVES), lds, stream, ka);                   \
    BGM_HIP_CHECK(hipGetLastError());                                                          \
    return BGM_OK;                                                                             \
  }
  BGM_CAUSAL_VARIANTS(X)
#undef X
  bgm_set_error("no compiled MH kernel variant for (KT1,KSL1,NTL)=(" + std::to_string(h->KT1) + "," +
                std::to_string(h->KSL1) + "," + std::to_string(h->NTL) + ")");
  return BGM_E_UNSUPPORTED;
}

extern "C" int bgm_causal_mh_run(bgm_handle *h, const bgm_mh_args *a, void *stream_) {
  if (!h || !h->configured) { bgm_set_error("bgm_causal_mh_run: handle not configured"); return BGM_E_STATE; }
  if (!a) { bgm_set_error("bgm_causal_mh_run: NULL args"); return BGM_E_INVALID; }
  if (a->n <= 0 || a->n_iters <= 0) return BGM_OK;
  if (!a->x_dev || !a->y_dev || !a->v_dev || !a->state_dev || !a->logp_dev) { bgm_set_error("bgm_causal_mh_run: NULL data pointer"); return BGM_E_INVALID; }
  if (a->row_base + a->n > 0xFFFFFFFFll) { bgm_set_error("bgm_causal_mh_run: row index exceeds the 32-bit RNG counter"); return BGM_E_INVALID; }
  if (a->effect == BGM_EFFECT_ADRF && (!a->x_values_dev || a->n_doses <= 0 || !a->adrf_partial_dev)) { bgm_set_error("bgm_causal_mh_run: ADRF effect needs x_values / adrf_partial"); return BGM_E_INVALID; }
  if (a->effect == BGM_EFFECT_ITE && !a->ite_dev) { bgm_set_error("bgm_causal_mh_run: ITE effect needs ite_dev"); return BGM_E_INVALID; }
  if (a->effect == BGM_EFFECT_ADRF && h->cfg.binary_treatment) { bgm_set_error("bgm_causal_mh_run: ADRF effect on a binary-treatment model"); return BGM_E_INVALID; }
  const int it_end = a->it_begin + a->n_iters;
  if ((a->effect != BGM_EFFECT_NONE || a->draws_dev) && it_end - a->burn_in > a->n_keep) { bgm_set_error("bgm_causal_mh_run: iterations beyond burn_in + n_keep"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (gx_wanted(h)) {
    return gx_mh_run(h, a, stream);
  }
  if (bnf_det_wanted(h)) {
    if (h->prior_seg || h->precision != 0) { bgm_set_error("bgm_causal_mh_run: the conditional prior / split precision exist for the LDS-resident shapes only"); return BGM_E_UNSUPPORTED; }
    return bnf_det_mh_run(h, a, stream);
  }
  int rc = bgm_causal_sampling_blob(h, stream);
  if (rc) return rc;
  const int grid = mh_grid(h, a->n);
  const int lds = h->meta.total * 4 + 64;   // + per-wave progress counters

  CausalMhKArgs ka{};
  ka.blob = h->sblob_dev; ka.x = a->x_dev; ka.y = a->y_dev; ka.v = a->v_dev;
  ka.n = a->n; ka.row_base = a->row_base; ka.state = a->state_dev; ka.logp = a->logp_dev;
  ka.burn_in = a->burn_in; ka.q_sd = a->q_sd;
  ka.k0 = (unsigned)(a->seed & 0xFFFFFFFFull); ka.k1 = (unsigned)(a->seed >> 32);
  ka.acc_count = a->acc_count_dev; ka.draws = a->draws_dev; ka.n_keep = a->n_keep;
  ka.sample_y = a->sample_y; ka.n_doses = a->n_doses; ka.x_values = a->x_values_dev;
  ka.adrf_partial = a->adrf_partial_dev; ka.ite = a->ite_dev; ka.clk = (unsigned long long *)a->clock_dev; ka.m = h->meta;

  // Split the segment at burn_in: the burn-in part runs the pure-transition kernel.
  struct Seg { int begin, n, effect, init, ev; };       // ev: 0 fused kernels; 1 / 2 = first / later segment of the event form
  std::vector<Seg> segs;
  const int split = std::min(std::max(a->burn_in, a->it_begin), it_end);
  if (split > a->it_begin) segs.push_back({a->it_begin, split - a->it_begin, BGM_EFFECT_NONE, a->init, 0});
  const int n_slots = grid * MH_WAVES;
  // the retained iterations: the fused sampler + outcome-net kernel, or (outcome cache mode 2, causal_event_api.hip) segments of
  // transitions that append events, each followed by the outcome net on dense event tiles and the spread over the segment's draws
  bool ev_form = it_end > split && bgm_causal_event_wanted(h, a->effect, a->n_doses);
  int S = 0;
  long long cap = 0;
  const int ev_doses = a->effect == BGM_EFFECT_ITE ? 2 : a->n_doses;      // (binary treatment: the two arms)
  if (ev_form && (rc = bgm_causal_event_plan(h, a->n, n_slots, ev_doses, it_end - split, &S, &cap))) {
    if (rc < 0) return rc;
    // the event buffers do not fit the budget / the device (rc > 0): release what was reserved and run the retained phase on the
    // fused kernel with the per-wave cache (mode 1) -- same sums to the last bit, no failure where predict ran before
    bgm_causal_event_free(h);
    ev_form = false;
    ++h->ev_fallbacks;
  }
  if (ev_form) {
    ka.ev_cap = cap;
    for (int b = split; b < it_end; b += S)
      segs.push_back({b, std::min(S, it_end - b), a->effect, (segs.empty() && b == split) ? a->init : 0, b == split ? 1 : 2});
    h->ev_total += (unsigned long long)a->n * (unsigned long long)(it_end - split);
  } else if (it_end > split) segs.push_back({split, it_end - split, a->effect, segs.empty() ? a->init : 0, 0});
  const int nseg = (int)segs.size();
  if (a->effect != BGM_EFFECT_NONE && it_end > split) {
    if (a->effect == BGM_EFFECT_ADRF) {      // (mean, sd) of the outcome net per wave slot, pass and lane: causal_effects_cached
      const size_t need = (size_t)n_slots * (size_t)((a->n_doses + 3) / 4) * 64 * 2;
      if (h->eff_cache_cap < need) {
        if (h->eff_cache) BGM_HIP_CHECK(hipFree(h->eff_cache));
        BGM_HIP_CHECK(hipMalloc(&h->eff_cache, need * sizeof(float)));
        h->eff_cache_cap = need;
      }
      ka.eff_cache = h->eff_cache;
    }                                        // (binary treatment: the two arms' pairs stay in registers, causal_ite_cached)
    ka.eff_skip = h->outcome_cache ? 1 : 0;
    if (!h->eff_stats_dev) {
      BGM_HIP_CHECK(hipMalloc(&h->eff_stats_dev, 2 * sizeof(unsigned long long)));
      BGM_HIP_CHECK(hipMemsetAsync(h->eff_stats_dev, 0, 2 * sizeof(unsigned long long), stream));
    }
    ka.eff_stats = h->eff_stats_dev;
    if (!ev_form) h->eff_total += (unsigned long long)((a->n + 15) / 16) * (unsigned long long)(it_end - split);      // retained tile-iterations launched
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  for (int s = 0; s < nseg; ++s) {
    ka.it_begin = segs[s].begin; ka.n_iters = segs[s].n; ka.init = segs[s].init;
    if (a->acc_count_dev) {   // slot-private counters for this launch, reduced into acc_count_dev afterwards
      const size_t need = (size_t)n_slots * segs[s].n;
      if (h->acc_scratch_cap < need) {
        if (h->acc_scratch) BGM_HIP_CHECK(hipFree(h->acc_scratch));
        BGM_HIP_CHECK(hipMalloc(&h->acc_scratch, need * sizeof(unsigned)));
        h->acc_scratch_cap = need;
      }
      BGM_HIP_CHECK(hipMemsetAsync(h->acc_scratch, 0, need * sizeof(unsigned), stream));
      ka.acc_count = h->acc_scratch;
    }
    // (the segments of the event form are timed as ONE interval: the retained phase)
    const bool t_open = h->timing && segs[s].ev != 2, t_close = h->timing && (segs[s].ev == 0 || s + 1 == nseg);
    if (t_open) {
      BGM_HIP_CHECK(hipEventCreate(&e0)); BGM_HIP_CHECK(hipEventCreate(&e1));
      BGM_HIP_CHECK(hipEventRecord(e0, stream));
    }
    if (segs[s].ev) {
      ka.ev_first = segs[s].ev == 1 ? 1 : 0;
      if ((rc = bgm_causal_event_mh_launch(h, ka, grid, lds, stream))) return rc;
      rc = bgm_causal_event_finish(h, ka, grid, ka.ev_first, stream, a->effect);
    } else if (h->precision != 0) rc = bgm_causal_bx3_mh_launch(h, ka, segs[s].effect, grid, stream);      // (carries the conditional prior)
    else if (h->prior_seg) rc = bgm_causal_prior_mh_launch(h, ka, segs[s].effect, grid, lds, stream);
    else if (segs[s].effect == BGM_EFFECT_ADRF) rc = launch_mh<1>(h, ka, grid, lds, stream);
    else if (segs[s].effect == BGM_EFFECT_ITE) rc = launch_mh<2>(h, ka, grid, lds, stream);
    else rc = launch_mh<0>(h, ka, grid, lds, stream);
    if (rc) return rc;
    if (a->acc_count_dev) {
      hipLaunchKernelGGL(acc_reduce_kernel, dim3((segs[s].n + 63) / 64), dim3(1024), 0, stream, h->acc_scratch, n_slots,
                         segs[s].n, a->acc_count_dev + segs[s].begin);
      BGM_HIP_CHECK(hipGetLastError());
    }
    if (t_close) {
      BGM_HIP_CHECK(hipEventRecord(e1, stream));
      h->events.push_back({e0, e1, segs[s].effect});
    }
  }
  return BGM_OK;
}

template <int EFFECT>
static int launch_eval(bgm_handle *h, const CausalEvalKArgs &ka, int grid, int lds, hipStream_t stream) {
  int rc;
#define X(KT1_, KSL1_, NTL_)                                                                   \
  if (h->KT1 == KT1_ && h->KSL1 == KSL1_ && h->NTL == NTL_) {                                  \
    auto k = causal_eval_kernel<KT1_, KSL1_, NTL_, MH_WAVES, EFFECT>;                          \
    rc = set_lds(k, lds);                                                                      \
    if (rc) return rc;                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * MH_WAVES), lds, stream, ka);                   \
    BGM_HIP_CHECK(hipGetLastError());                                                          \
    return BGM_OK;                                                                             \
  }
  BGM_CAUSAL_VARIANTS(X)
#undef X
  bgm_set_error("no compiled evaluate kernel variant for this shape");
  return BGM_E_UNSUPPORTED;
}

extern "C" int bgm_causal_evaluate(bgm_handle *h, const float *x, const float *y, const float *v, const float *z,
                                   int64_t n, const float *x_values, int32_t n_doses, double *sums,
                                   float *adrf_partial, float *ite, void *stream_) {
  if (!h || !h->configured) { bgm_set_error("bgm_causal_evaluate: handle not configured"); return BGM_E_STATE; }
  if (n <= 0) return BGM_OK;
  if (!x || !y || !v || !z || !sums) { bgm_set_error("bgm_causal_evaluate: NULL pointer"); return BGM_E_INVALID; }
  const bool binary = h->cfg.binary_treatment != 0;
  if (binary && !ite) { bgm_set_error("bgm_causal_evaluate: ite_dev required for binary treatment"); return BGM_E_INVALID; }
  if (!binary && (!x_values || n_doses <= 0 || !adrf_partial)) { bgm_set_error("bgm_causal_evaluate: x_values / adrf_partial required"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (gx_wanted(h)) return gx_evaluate(h, x, y, v, z, n, x_values, n_doses, sums, adrf_partial, ite, stream);
  if (bnf_det_wanted(h)) return bnf_det_evaluate(h, x, y, v, z, n, x_values, n_doses, sums, adrf_partial, ite, stream);
  int rc = bgm_causal_sampling_blob(h, stream);
  if (rc) return rc;
  CausalEvalKArgs ka{};
  ka.blob = h->sblob_dev; ka.x = x; ka.y = y; ka.v = v; ka.z = z; ka.n = n; ka.sums = sums;
  ka.x_values = x_values; ka.n_doses = binary ? 2 : n_doses; ka.adrf_partial = adrf_partial; ka.ite = ite; ka.m = h->meta;
  const int64_t tiles = (n + 15) / 16;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((tiles + MH_WAVES - 1) / MH_WAVES, h->n_cus));
  const int lds = h->meta.total * 4;
  return binary ? launch_eval<2>(h, ka, grid, lds, stream) : launch_eval<1>(h, ka, grid, lds, stream);
}

template <int EFFECT>
static int launch_effects(bgm_handle *h, const CausalEffKArgs &ka, int grid, int lds, hipStream_t stream) {
  int rc;
#define X(KT1_, KSL1_, NTL_)                                                                   \
  if (h->KT1 == KT1_ && h->KSL1 == KSL1_ && h->NTL == NTL_) {                                  \
    auto k = causal_effects_kernel<KT1_, KSL1_, MH_WAVES, EFFECT>;                             \
    rc = set_lds(k, lds);                                                                      \
    if (rc) return rc;                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * MH_WAVES), lds, stream, ka);                   \
    BGM_HIP_CHECK(hipGetLastError());                                                          \
    return BGM_OK;                                                                             \
  }
  BGM_CAUSAL_VARIANTS(X)
#undef X
  bgm_set_error("no compiled effects kernel variant for this shape");
  return BGM_E_UNSUPPORTED;
}

extern "C" int bgm_causal_effects(bgm_handle *h, const float *x, const float *draws, int64_t n, int64_t row_base, int32_t n_keep,
                                  int32_t burn_in, uint64_t seed, int32_t sample_y, const float *x_values, int32_t n_doses,
                                  float *adrf_partial, float *ite, void *stream_) {
  if (!h || !h->configured) { bgm_set_error("bgm_causal_effects: handle not configured"); return BGM_E_STATE; }
  if (n <= 0 || n_keep <= 0) return BGM_OK;
  if (!x || !draws) { bgm_set_error("bgm_causal_effects: NULL pointer"); return BGM_E_INVALID; }
  const bool binary = h->cfg.binary_treatment != 0;
  if (binary && !ite) { bgm_set_error("bgm_causal_effects: ite_dev required for binary treatment"); return BGM_E_INVALID; }
  if (!binary && (!x_values || n_doses <= 0 || !adrf_partial)) { bgm_set_error("bgm_causal_effects: x_values / adrf_partial required"); return BGM_E_INVALID; }
  if (row_base + n > 0xFFFFFFFFll) { bgm_set_error("bgm_causal_effects: row index exceeds the 32-bit RNG counter"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (gx_wanted(h)) return gx_effects(h, draws, n, row_base, n_keep, burn_in, seed, sample_y, x_values, n_doses, adrf_partial, ite, stream);
  if (bnf_det_wanted(h)) return bnf_det_effects(h, draws, n, row_base, n_keep, burn_in, seed, sample_y, x_values, n_doses, adrf_partial, ite, stream);
  int rc = bgm_causal_sampling_blob(h, stream);
  if (rc) return rc;
  CausalEffKArgs ka{};
  ka.blob = h->sblob_dev; ka.x = x; ka.draws = draws; ka.n = n; ka.row_base = row_base; ka.n_keep = n_keep; ka.burn_in = burn_in;
  ka.sample_y = sample_y; ka.n_doses = binary ? 2 : n_doses; ka.x_values = x_values; ka.adrf_partial = adrf_partial; ka.ite = ite;
  ka.k0 = (unsigned)(seed & 0xFFFFFFFFull); ka.k1 = (unsigned)(seed >> 32); ka.m = h->meta;
  const int64_t tiles = (n + 15) / 16;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((tiles + MH_WAVES - 1) / MH_WAVES, h->n_cus));
  const int lds = h->meta.total * 4 + 64;
  return binary ? launch_effects<2>(h, ka, grid, lds, stream) : launch_effects<1>(h, ka, grid, lds, stream);
}

extern "C" int bgm_causal_evaluate_slots(bgm_handle *h, int64_t n, int32_t *n_slots) {
  if (!h || !n_slots) { bgm_set_error("bgm_causal_evaluate_slots: NULL"); return BGM_E_INVALID; }
  if (h->configured && gx_wanted(h)) { *n_slots = gx_slots(h, n); return BGM_OK; }
  if (h->configured && bnf_det_wanted(h)) { *n_slots = bnf_det_slots(h); return BGM_OK; }
  const int64_t tiles = (n + 15) / 16;
  *n_slots = (int)std::max<int64_t>(1, std::min<int64_t>((tiles + MH_WAVES - 1) / MH_WAVES, h->n_cus)) * MH_WAVES;
  return BGM_OK;
}

extern "C" int bgm_causal_set_outcome_cache(bgm_handle *h, int32_t on) {
  if (!h || on < 0 || on > 2) { bgm_set_error("bgm_causal_set_outcome_cache: mode must be 0 (off), 1 (per wave) or 2 (per chain: event form)"); return BGM_E_INVALID; }
  h->outcome_cache = on;
  return BGM_OK;
}

extern "C" int bgm_causal_outcome_cache_stats(bgm_handle *h, int64_t *out2, int32_t reset) {
  if (!h || !out2) { bgm_set_error("bgm_causal_outcome_cache_stats: bad argument"); return BGM_E_INVALID; }
  out2[0] = out2[1] = 0;
  if (!h->eff_stats_dev) return BGM_OK;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  BGM_HIP_CHECK(hipDeviceSynchronize());
  unsigned long long v[2];
  BGM_HIP_CHECK(hipMemcpy(v, h->eff_stats_dev, sizeof(v), hipMemcpyDeviceToHost));
  // wave cache: retained tile-iterations served / launched; event form: retained chain-iterations that needed no outcome-net
  // evaluation (all of them minus the events) / retained chain-iterations
  out2[0] = (int64_t)v[0] + (int64_t)(h->ev_total - std::min<unsigned long long>(v[1], h->ev_total)); out2[1] = (int64_t)(h->eff_total + h->ev_total);
  if (reset) { BGM_HIP_CHECK(hipMemset(h->eff_stats_dev, 0, sizeof(v))); h->eff_total = 0; h->ev_total = 0; }
  return BGM_OK;
}

extern "C" int bgm_timing_enable(bgm_handle *h, int enable) {
  if (!h) return BGM_E_INVALID;
  h->timing = enable != 0;
  return BGM_OK;
}

extern "C" int bgm_timing_read(bgm_handle *h, int kind, int64_t *n_launches, double *total_ms, int reset) {
  if (!h || kind < -1 || kind > 2) { bgm_set_error("bgm_timing_read: bad argument"); return BGM_E_INVALID; }
  for (auto &e : h->events) {
    BGM_HIP_CHECK(hipEventSynchronize(e.b));
    float ms = 0.0f;
    BGM_HIP_CHECK(hipEventElapsedTime(&ms, e.a, e.b));
    h->timed_ms[e.kind] += ms;
    h->timed_launches[e.kind] += 1;
    hipEventDestroy(e.a); hipEventDestroy(e.b);
  }
  h->events.clear();
  long long n = 0; double ms = 0.0;
  for (int k = 0; k < 3; ++k) if (kind < 0 || kind == k) { n += h->timed_launches[k]; ms += h->timed_ms[k]; }
  if (n_launches) *n_launches = n;
  if (total_ms) *total_ms = ms;
  if (reset) for (int k = 0; k < 3; ++k) if (kind < 0 || kind == k) { h->timed_launches[k] = 0; h->timed_ms[k] = 0.0; }
  return BGM_OK;
}

extern "C" int bgm_causal_describe(bgm_handle *h, int32_t batch, char *out, int32_t cap) {
  if (!h || !h->configured || !out || cap < 1) { bgm_set_error("bgm_causal_describe: bad argument"); return BGM_E_INVALID; }
  std::string s = "sampler=";
  s += gx_wanted(h) ? (gx_row_tile_per_wave(h) ? "gx_causal_mh_kernel -> gw_causal_mh_kernel (general-width engine, one 16-row tile per wave: no barriers inside a transition, padded weights streamed from L2)"
                                                : "gx_causal_mh_kernel (general-width engine: 32-row LDS activation tiles, padded weights streamed from L2)") : bnf_det_wanted(h) ? "bnf_mh_kernel<DET> / bnf_effects_kernel<DET> (general shapes: persistent workgroups, weights streamed from L2)"
                         : "causal_mh_kernel (weights LDS-resident, one launch per rank shard)";
  if (h->fit_active) {
    s += "; fit=";
    if (gx_fit_active(h)) s += "gx_causal_fit_kernel / fit_dw_kernel (general-width engine)";
    else if (h->fit_chain && batch <= 32) s += "fit_chain_kernel (register-chained row tiles, rows masked to the " + std::to_string(batch) + "-row local minibatch)";
    else s += "fit_fwd_kernel / fit_bwd_kernel / fit_dw_kernel (LDS-blob phase kernels)";
  }
  std::snprintf(out, (size_t)cap, "%s", s.c_str());
  return BGM_OK;
}

extern "C" int bgm_causal_mh_info(bgm_handle *h, int64_t n, bgm_mh_info *info) {
  if (!h || !h->configured || !info) { bgm_set_error("bgm_causal_mh_info: bad argument"); return BGM_E_INVALID; }
  if (gx_wanted(h)) {
    double macs = 0.0, pmacs = 0.0;
    for (int id : {BGM_NET_G, BGM_NET_F, BGM_NET_H}) {
      const HostNet &nn = h->nets[id];
      for (size_t l = 0; l + 1 < nn.dims.size(); ++l) {
        macs += (double)nn.dims[l] * nn.dims[l + 1];
        pmacs += (double)((nn.dims[l] + 31) / 32 * 32) * ((nn.dims[l + 1] + 31) / 32 * 32);
      }
    }
    info->rows_per_wave = 8; info->waves_per_block = 4; info->grid_blocks = gx_row_tile_per_wave(h) ? gx_slots(h, n) / 4 : gx_slots(h, n);
    if (gx_row_tile_per_wave(h)) info->rows_per_wave = 16;
    info->mfma_per_transition_per_wave = (int)(pmacs * 32.0 / 1024.0 / 4.0);   // issued 16x16x4 MFMAs per 32-row tile and transition, per wave
    info->lds_bytes = 0;
    info->flop_per_row_transition = 2.0 * macs;
    return BGM_OK;
  }
  const int q1 = h->q + 1;
  int KT1, KSL1, NTL;
  if (!bgm_causal_shape(q1, h->p + 1, KT1, KSL1, NTL)) { bgm_set_error("bgm_causal_mh_info: no compiled kernel shape contains this model"); return BGM_E_UNSUPPORTED; }
  const int ks1 = 4 * (KT1 - 1) + KSL1;
  const int n_gh = h->cfg.n_hidden_g - 1;
  info->rows_per_wave = 16 * MH_R;
  info->waves_per_block = MH_WAVES;
  info->grid_blocks = mh_grid(h, n);
  // issued MFMAs per 16 rows, times R row groups
  const int per16 = 3 * ks1 * 4 + n_gh * 64 + 16 * NTL + 2 * (32 + 8 + 4);
  info->mfma_per_transition_per_wave = per16 * MH_R;
  info->lds_bytes = h->blob_valid ? h->meta.total * 4 : 0;
  double macs = 0.0;
  for (int id : {BGM_NET_G, BGM_NET_F, BGM_NET_H}) {
    const HostNet &nn = h->nets[id];
    for (size_t l = 0; l + 1 < nn.dims.size(); ++l) macs += (double)nn.dims[l] * nn.dims[l + 1];
  }
  info->flop_per_row_transition = 2.0 * macs;
  return BGM_OK;
}
