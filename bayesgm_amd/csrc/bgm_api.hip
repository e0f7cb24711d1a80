// bgm_api.hip -- C-ABI entry points of the BGM posterior path (include/bgm_hip.h, BGM section).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "bgm_host.h"
#include "bgm_kernels.h"
#include "bgm_state.h"
#include "gx_bgm_host.h"

static constexpr int BGM_WAVES = 8;
#ifndef BGM_WAVES_WIDE_HMC
#define BGM_WAVES_WIDE_HMC 12   // 148 VGPRs -> 3 waves/SIMD; measured 89 vs 86 (8) vs 86 (16) TF at p=500
#endif
#ifndef BGM_SX3_WAVES_DEFAULT
#define BGM_SX3_WAVES_DEFAULT 12      // ms per transition at N = 2e5, p = 500: 3.43 (8 waves, no spill) / 3.19 (12 waves, 168 registers); two 6-wave workgroups per CU: 4.0; two / three / four units per stream step: 3.12 / 3.24 / 3.34
#endif
static constexpr float BN_EPS_F = 1e-3f;   // keras BatchNormalization default epsilon

void bgm_bgm_free_state(bgm_handle *h) {
  if (!h->bgm_state) return;
  BgmState *s = static_cast<BgmState *>(h->bgm_state);
  bgm_bgm_fit_free(h);
  gxb_free(s);
  if (s->blob_dev) hipFree(s->blob_dev);
  if (s->sx3_dev) hipFree(s->sx3_dev);
  if (s->sx3_bias_dev) hipFree(s->sx3_bias_dev);
  delete s;
  h->bgm_state = nullptr;
}


static size_t bgm_count(const bgm_bgm_config &c) {
  size_t n = 4 * (size_t)c.z_dim;
  int in = c.z_dim;
  for (int i = 0; i < c.n_hidden_g; ++i) { n += (size_t)in * c.g_units[i] + c.g_units[i]; in = c.g_units[i]; }
  n += 2 * ((size_t)in * c.x_dim + c.x_dim);
  return n;
}

extern "C" int bgm_bgm_configure(bgm_handle *h, const bgm_bgm_config *cfg) {
  if (!h || !cfg) { bgm_set_error("bgm_bgm_configure: NULL argument"); return BGM_E_INVALID; }
  if (cfg->x_dim < 1 || cfg->z_dim < 1 || cfg->n_hidden_g < 1 || cfg->n_hidden_g > BGM_MAX_LAYERS) {
    bgm_set_error("bgm_bgm_configure: bad dimensions"); return BGM_E_INVALID;
  }
  // any trunk (networks/base.py:53-117 takes any nb_units): [64] x 3 / [64] x 5 at z_dim <= 16 run on the dual-access-blob kernels,
  // every other shape on the general-width engine (gx_bgm_api.hip)
  for (int i = 0; i < cfg->n_hidden_g; ++i)
    if (cfg->g_units[i] < 1 || cfg->g_units[i] > 4096) { bgm_set_error("bgm_bgm_configure: g_units must be in [1, 4096]"); return BGM_E_INVALID; }
  BgmState *s = bst(h);
  gxb_free(s);
  s->cfg = *cfg;
  s->configured = true; s->set = false; s->blob_valid = false;
  s->theta.assign(bgm_count(*cfg), 0.0f);
  return BGM_OK;
}

extern "C" int bgm_bgm_set_weights(bgm_handle *h, const float *theta, int64_t count, void *stream) {
  (void)stream;
  if (!h || !h->bgm_state || !bst(h)->configured) { bgm_set_error("bgm_bgm_set_weights: not configured"); return BGM_E_STATE; }
  BgmState *s = bst(h);
  if (!theta || (size_t)count != s->theta.size()) {
    bgm_set_error("bgm_bgm_set_weights: expected " + std::to_string(s->theta.size()) + " floats"); return BGM_E_INVALID;
  }
  std::memcpy(s->theta.data(), theta, sizeof(float) * count);
  s->set = true; s->blob_valid = false; s->gx_valid = false;
  return BGM_OK;
}

static int bgm_build_blob(bgm_handle *h, hipStream_t stream) {
  BgmState *s = bst(h);
  if (s->blob_valid) return BGM_OK;
  if (!s->set) { bgm_set_error("BGM weights not set"); return BGM_E_STATE; }
  const int q = s->cfg.z_dim, p = s->cfg.x_dim, NH = s->cfg.n_hidden_g;
  const int KTQ = (q + 15) / 16;
  BgmMeta &m = s->meta;
  int ntx_variant = 0;
  const int lds_bytes = bgm_layout(q, p, NH, m, ntx_variant, s->precision != 0);      // (split-precision heads: the streamed variant)
  if (lds_bytes < 0) {
    bgm_set_error("BGM generator: trunk + head biases + stage exceed the 160 KiB LDS (x_dim too large)");
    return BGM_E_UNSUPPORTED;
  }
  s->KTQ = KTQ; s->NTX = ntx_variant; s->NH = NH; s->lds_bytes = lds_bytes;
  const int NTX = m.ntx;
  std::vector<float> blob(m.total, 0.0f);
  const float *th = s->theta.data();
  const float *gamma = th, *beta = th + q, *mmean = th + 2 * q, *mvar = th + 3 * q;
  size_t o = 4 * (size_t)q;
  // first Dense with the inference-mode BatchNorm folded in:  zn = z*scale + shift
  std::vector<float> scale(q), shift(q);
  for (int i = 0; i < q; ++i) { scale[i] = gamma[i] / std::sqrt(mvar[i] + BN_EPS_F); shift[i] = beta[i] - mmean[i] * scale[i]; }
  std::vector<float> W1(th + o, th + o + (size_t)q * 64); o += (size_t)q * 64;
  std::vector<float> b1(th + o, th + o + 64); o += 64;
  for (int k = 0; k < 64; ++k) { double acc = b1[k]; for (int i = 0; i < q; ++i) acc += (double)shift[i] * W1[(size_t)i * 64 + k]; b1[k] = (float)acc; }
  for (int i = 0; i < q; ++i) for (int k = 0; k < 64; ++k) W1[(size_t)i * 64 + k] *= scale[i];
  pack17(blob, m.w1, W1, q, 64, 16 * KTQ, 4, [&](int slot) { int f = l1_feature(slot); return f < q ? f : -1; });
  for (int k = 0; k < 64; ++k) blob[m.b1 + k] = b1[k];
  auto ident = [](int r) { return r; };
  for (int l = 0; l < m.n_hh; ++l) {
    std::vector<float> W(th + o, th + o + 4096); o += 4096;
    pack17(blob, m.wh + l * 4 * 64 * 17, W, 64, 64, 64, 4, ident);
    for (int k = 0; k < 64; ++k) blob[m.bh + l * 64 + k] = th[o + k];
    o += 64;
  }
  for (int head = 0; head < 2; ++head) {   // mean, var
    std::vector<float> W(th + o, th + o + (size_t)64 * p); o += (size_t)64 * p;
    pack17_heads(blob, m.whd, W.data(), p, NTX, head);
    for (int k = 0; k < p; ++k) blob[m.bhd + head * 16 * NTX + k] = th[o + k];
    o += p;
  }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (s->blob_cap < blob.size()) {
    if (s->blob_dev) BGM_HIP_CHECK(hipFree(s->blob_dev));
    BGM_HIP_CHECK(hipMalloc(&s->blob_dev, blob.size() * sizeof(float)));
    s->blob_cap = blob.size();
  }
  BGM_HIP_CHECK(hipMemcpyAsync(s->blob_dev, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  std::vector<unsigned short> hx3;
  if (s->precision != 0) {
    // split-precision head fragments (bgm_kernels.h): per 16-feature block 16 fragments of 64 lanes x 8 halves --
    //   forward f = 2 (2 head + b) + (lo): lane (i, gA), slot u <-> W_head[unit 16 (2 b + (u >> 2)) + 4 gA + (u & 3)][column 16 tx + i]
    //   backward f = 8 + 2 ti + (lo):     lane (i, gA), slot u <-> (u < 4 ? W_mean : W_var)[unit 16 ti + i][column 16 tx + 4 gA + (u & 3)]
    const float *Wm = th + (o - 2 * ((size_t)64 * p + p)), *Wv = Wm + (size_t)64 * p + p;
    hx3.assign((size_t)((NTX + BGM_X3_STEP - 1) / BGM_X3_STEP * BGM_X3_STEP) * (BGM_X3_BLOCK_BYTES / 2), 0);      // (whole steps of the stream; padding = zero fragments)
    auto put = [&](size_t base, float w) {      // hi at base, lo one fragment (512 halves) further
      const unsigned short hi = bgm_f2h(w);
      hx3[base] = hi;
      hx3[base + 512] = bgm_f2h(w - bgm_h2f(hi));
    };
    for (int tx = 0; tx < NTX; ++tx) {
      const size_t blk = (size_t)tx * (BGM_X3_BLOCK_BYTES / 2);
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, gA = lane >> 4;
        for (int u = 0; u < 8; ++u) {
          for (int head = 0; head < 2; ++head)
            for (int b = 0; b < 2; ++b) {
              const int unit = 16 * (2 * b + (u >> 2)) + 4 * gA + (u & 3), col = 16 * tx + i;
              const float w = col < p ? (head ? Wv : Wm)[(size_t)unit * p + col] : 0.0f;
              put(blk + (size_t)(2 * (2 * head + b)) * 512 + (size_t)lane * 8 + u, w);
            }
          for (int ti = 0; ti < 4; ++ti) {
            const int unit = 16 * ti + i, col = 16 * tx + 4 * gA + (u & 3);
            const float w = col < p ? (u < 4 ? Wm : Wv)[(size_t)unit * p + col] : 0.0f;
            put(blk + (size_t)(8 + 2 * ti) * 512 + (size_t)lane * 8 + u, w);
          }
        }
      }
    }
    // ---- the stream: steps of BGM_X3_STEP units of 16 KiB, the head blocks above in the middle; unit layouts (64 lanes x 8 halves per fragment, hi at an
    //      even fragment index, lo behind it):
    //   L1      f = 2 mt (+1): lane (i, gA), slot u < 4 <-> W1'[latent 4 u + gA][unit 16 mt + i] (BatchNorm folded in, as in the fp32 blob), u >= 4: 0;
    //           f = 8 + 2 b (+1), the transposed copy: lane (i, gA), slot u <-> W1'[latent 4 (i % 4) + i / 4][unit 16 (2 b + (u >> 2)) + 4 gA + (u & 3)]
    //   hidden  forward  f = 2 (2 mt + b) (+1): lane (i, gA), slot u <-> W[unit 16 (2 b + (u >> 2)) + 4 gA + (u & 3)][unit 16 mt + i]
    //           backward f = 2 (2 ti + b) (+1): lane (i, gA), slot u <-> W[unit 16 ti + i][unit 16 (2 b + (u >> 2)) + 4 gA + (u & 3)]
    //   heads   the 16 KiB blocks above, BGM_X3_STEP per step
    if (KTQ != 1) { bgm_set_error("BGM generator: split precision serves z_dim <= 16"); return BGM_E_UNSUPPORTED; }
    {
      const int S = (NTX + BGM_X3_STEP - 1) / BGM_X3_STEP, U = BGM_X3_STEP, FT = (NH + U - 1) / U, n_steps = 2 * FT + S;
      // trunk layer l forward: unit l % U of step l / U; backward unit i = NH - 1 - l (hidden NH - 1 first, L1 last): unit i % U of step FT + S + i / U
      auto fpos = [&](int l) { return std::pair<size_t, int>((size_t)(l / U), (l % U) * 16); };
      auto bpos = [&](int l) { const int i = NH - 1 - l; return std::pair<size_t, int>((size_t)(FT + S + i / U), (i % U) * 16); };
      const size_t step_h = (size_t)BGM_X3_STEP * (BGM_X3_BLOCK_BYTES / 2);      // halves per step
      std::vector<unsigned short> sx((size_t)n_steps * step_h, 0);
      auto put2 = [&](size_t step, int frag, int lane, int u, float w) {
        const size_t base = step * step_h + (size_t)frag * 512 + (size_t)lane * 8 + u;
        const unsigned short hi = bgm_f2h(w);
        sx[base] = hi;
        sx[base + 512] = bgm_f2h(w - bgm_h2f(hi));
      };
      const float *Wh = th + 4 * (size_t)q + (size_t)q * 64 + 64;      // hidden layer l (1-based) at Wh + (l - 1) * (4096 + 64)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, gA = lane >> 4;
        for (int u = 0; u < 8; ++u) {
          const int ku = 4 * gA + (u & 3);      // + 16 (2 b + (u >> 2))
          for (int first = 0; first < 2; ++first) {      // L1 sits at its forward position and at its backward position (the last unit)
            const auto pos = first ? fpos(0) : bpos(0);
            for (int mt = 0; mt < 4; ++mt) {
              const int f = 4 * u + gA;
              put2(pos.first, pos.second + 2 * mt, lane, u, (u < 4 && f < q) ? W1[(size_t)f * 64 + 16 * mt + i] : 0.0f);
            }
            for (int b = 0; b < 2; ++b) {
              const int f = 4 * (i & 3) + (i >> 2);
              put2(pos.first, pos.second + 8 + 2 * b, lane, u, f < q ? W1[(size_t)f * 64 + 16 * (2 * b + (u >> 2)) + ku] : 0.0f);
            }
          }
          for (int l = 1; l < NH; ++l) {
            const float *W = Wh + (size_t)(l - 1) * (4096 + 64);
            const auto pf = fpos(l), pb = bpos(l);
            for (int t = 0; t < 4; ++t)
              for (int b = 0; b < 2; ++b) {
                const int k = 16 * (2 * b + (u >> 2)) + ku;
                put2(pf.first, pf.second + 2 * (2 * t + b), lane, u, W[(size_t)k * 64 + 16 * t + i]);              // forward: out tile t
                put2(pb.first, pb.second + 2 * (2 * t + b), lane, u, W[(size_t)(16 * t + i) * 64 + k]);            // backward: in tile t
              }
          }
        }
      }
      for (int tx = 0; tx < NTX; ++tx)      // the head blocks as packed above
        std::memcpy(&sx[(size_t)(FT + tx / BGM_X3_STEP) * step_h + (size_t)(tx % BGM_X3_STEP) * (BGM_X3_BLOCK_BYTES / 2)],
                    &hx3[(size_t)tx * (BGM_X3_BLOCK_BYTES / 2)], BGM_X3_BLOCK_BYTES);
      if (s->sx3_cap < sx.size() * 2) {
        if (s->sx3_dev) BGM_HIP_CHECK(hipFree(s->sx3_dev));
        BGM_HIP_CHECK(hipMalloc((void **)&s->sx3_dev, sx.size() * 2));
        s->sx3_cap = sx.size() * 2;
      }
      BGM_HIP_CHECK(hipMemcpy(s->sx3_dev, sx.data(), sx.size() * 2, hipMemcpyHostToDevice));
      // biases: [b1' (64) | hidden (NH - 1) x 64 | heads 2 x 16 NTX] in the posterior blob's own order, the stage behind them
      BgmMeta &xm = s->sx3_meta;
      xm = m;
      xm.b1 = 0; xm.bh = 64; xm.bhd = 64 + 64 * m.n_hh;
      const int nb = xm.bhd + 2 * 16 * NTX;
      xm.w1 = xm.wh = xm.whd = 0;
      xm.lds_resident = (nb + 3) / 4 * 4; xm.stage = xm.lds_resident; xm.total = xm.lds_resident;
      std::vector<float> xb(xm.lds_resident, 0.0f);
      for (int k = 0; k < 64; ++k) xb[xm.b1 + k] = blob[m.b1 + k];
      for (int k = 0; k < 64 * m.n_hh; ++k) xb[xm.bh + k] = blob[m.bh + k];
      for (int k = 0; k < 2 * 16 * NTX; ++k) xb[xm.bhd + k] = blob[m.bhd + k];
      if (s->sx3_bias_cap < xb.size()) {
        if (s->sx3_bias_dev) BGM_HIP_CHECK(hipFree(s->sx3_bias_dev));
        BGM_HIP_CHECK(hipMalloc((void **)&s->sx3_bias_dev, xb.size() * sizeof(float)));
        s->sx3_bias_cap = xb.size();
      }
      BGM_HIP_CHECK(hipMemcpy(s->sx3_bias_dev, xb.data(), xb.size() * sizeof(float), hipMemcpyHostToDevice));
      s->lds_bytes_sx3 = (xm.stage + 2 * BGM_X3_STEP * (BGM_X3_BLOCK_BYTES / 4)) * 4;
    }
  }
  BGM_HIP_CHECK(hipStreamSynchronize(stream));
  s->blob_valid = true;
  return BGM_OK;
}

// Arithmetic of the posterior kernels' head products (bgm_kernels.h "Split-precision heads"): 0 fp32 (default), 2 f16x3.  Opt-in
// (models: params['hmc_precision'] = 'f16x3'); serves the dual-access-blob shapes (trunk [64] x {3, 5}, z_dim <= 16) at any x_dim
// through the streamed-head kernels; log posterior / gradient / HMC.  Predictive draws and the minibatch steps stay fp32.
extern "C" int bgm_bgm_set_precision(bgm_handle *h, int32_t mode) {
  if (!h || !h->bgm_state || !bst(h)->configured) { bgm_set_error("bgm_bgm_set_precision: not configured"); return BGM_E_STATE; }
  if (mode != 0 && mode != 2) { bgm_set_error("bgm_bgm_set_precision: mode must be 0 (fp32) or 2 (f16x3)"); return BGM_E_INVALID; }
  BgmState *s = bst(h);
  if (mode != 0 && gxb_wanted(s)) { bgm_set_error("bgm_bgm_set_precision: split precision exists for the default trunk shapes ([64] x 3 / [64] x 5, z_dim <= 16)"); return BGM_E_UNSUPPORTED; }
  if (mode != 0 && s->fit_active) { bgm_set_error("bgm_bgm_set_precision: not inside a fit session"); return BGM_E_STATE; }
  if (s->precision != mode) { s->precision = mode; s->blob_valid = false; }
  return BGM_OK;
}

// (KTQ, NTX, NH) variants: z_dim <= 16; x_dim in (16,32] / (96,112] LDS-resident, NTX = 0 = wide (any x_dim,
// head weights streamed through an LDS stage); 5 hidden layers (configs/*.yaml) or 3
#define BGM_BGM_VARIANTS(X) X(1, 2, 5) X(1, 7, 5) X(1, 0, 5) X(1, 2, 3) X(1, 7, 3) X(1, 0, 3)

static int bgm_grid(const bgm_handle *h, long long tiles) {
  return (int)std::max<long long>(1, std::min<long long>((tiles + BGM_WAVES - 1) / BGM_WAVES, h->n_cus));
}
#define BGM_NO_VARIANT(s)                                                                               \
  bgm_set_error("no compiled BGM kernel variant for (KTQ,NTX,NH)=(" + std::to_string((s)->KTQ) + "," +  \
                std::to_string((s)->NTX) + "," + std::to_string((s)->NH) + ")");                        \
  return BGM_E_UNSUPPORTED;

extern "C" int bgm_bgm_logpost(bgm_handle *h, const float *z, const float *x, int64_t n, float *out, float *grad,
                               void *stream_) {
  if (!h || !h->bgm_state || !bst(h)->configured) { bgm_set_error("bgm_bgm_logpost: not configured"); return BGM_E_STATE; }
  if (n <= 0) return BGM_OK;
  if (!z || !x || !out) { bgm_set_error("bgm_bgm_logpost: NULL pointer"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (gxb_wanted(bst(h))) return gxb_logpost(h, bst(h), z, x, n, out, grad, stream);
  int rc = bgm_build_blob(h, stream);
  if (rc) return rc;
  BgmState *s = bst(h);
  const int grid = bgm_grid(h, (n + 15) / 16), lds = s->lds_bytes;
  if (s->precision != 0) {      // split precision: the whole generator as one fp16 fragment stream (bgm_kernels.h, PREC 2)
    const int ldx = s->lds_bytes_sx3;
#define X(KTQ_, NH_)                                                                                                \
    if (s->KTQ == KTQ_ && s->NH == NH_) {                                                                           \
      auto k = (s->sx3_meta.p & 3) == 0 ? bgm_logpost_kernel<KTQ_, 0, NH_, BGM_WAVES, 2, true> : bgm_logpost_kernel<KTQ_, 0, NH_, BGM_WAVES, 2, false>; \
      BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, ldx)); \
      hipLaunchKernelGGL(k, dim3(grid), dim3(64 * BGM_WAVES), ldx, stream, s->sx3_bias_dev, s->sx3_meta, z, x, (long long)n, out, grad, \
                         (const unsigned char *)s->sx3_dev);                                                        \
      BGM_HIP_CHECK(hipGetLastError());                                                                             \
      return BGM_OK;                                                                                                \
    }
    X(1, 5) X(1, 3)
#undef X
    BGM_NO_VARIANT(s)
  }
#define X(KTQ_, NTX_, NH_)                                                                                          \
  if (s->KTQ == KTQ_ && s->NTX == NTX_ && s->NH == NH_) {                                                           \
    auto k = bgm_logpost_kernel<KTQ_, NTX_, NH_, BGM_WAVES>;                                                        \
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * BGM_WAVES), lds, stream, s->blob_dev, s->meta, z, x, (long long)n, out, grad, \
                       (const unsigned char *)nullptr);                                                             \
    BGM_HIP_CHECK(hipGetLastError());                                                                               \
    return BGM_OK;                                                                                                  \
  }
  BGM_BGM_VARIANTS(X)
#undef X
  BGM_NO_VARIANT(s)
}

extern "C" int bgm_bgm_hmc_run(bgm_handle *h, const bgm_hmc_args *a, void *stream_) {
  if (!h || !h->bgm_state || !bst(h)->configured) { bgm_set_error("bgm_bgm_hmc_run: not configured"); return BGM_E_STATE; }
  if (!a) { bgm_set_error("bgm_bgm_hmc_run: NULL args"); return BGM_E_INVALID; }
  if (a->n <= 0 || a->n_iters <= 0) return BGM_OK;
  if (!a->x_dev || !a->state_dev || !a->logp_dev || !a->grad_dev || !a->step_dev) { bgm_set_error("bgm_bgm_hmc_run: NULL pointer"); return BGM_E_INVALID; }
  if (a->n_leapfrog < 1) { bgm_set_error("bgm_bgm_hmc_run: num_leapfrog_steps must be >= 1"); return BGM_E_INVALID; }
  if (a->row_base + a->n > 0xFFFFFFFFll) { bgm_set_error("bgm_bgm_hmc_run: row index exceeds the 32-bit RNG counter"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (gxb_wanted(bst(h))) return gxb_hmc_run(h, bst(h), a, stream);
  int rc = bgm_build_blob(h, stream);
  if (rc) return rc;
  BgmState *s = bst(h);
  BgmHmcKArgs ka{};
  ka.blob = s->blob_dev; ka.x = a->x_dev; ka.n = a->n; ka.row_base = a->row_base;
  ka.state = a->state_dev; ka.logp = a->logp_dev; ka.grad = a->grad_dev; ka.init = a->init;
  ka.it_begin = a->it_begin; ka.n_iters = a->n_iters; ka.burn_in = a->burn_in; ka.n_leapfrog = a->n_leapfrog;
  ka.step = a->step_dev; ka.k0 = (unsigned)(a->seed & 0xFFFFFFFFull); ka.k1 = (unsigned)(a->seed >> 32);
  ka.acc_prob_sum = a->acc_prob_sum_dev; ka.acc_count = a->acc_count_dev; ka.draws = a->draws_dev; ka.m = s->meta;
  const int lds = s->lds_bytes;
  const long long tiles = (a->n + 15) / 16;
  // The wide variant's unit of work is a block pass (W row tiles x all iterations of the launch) and every block makes the same
  // number of passes.  Tiles are dealt wave-major and tile-less waves skip the matrix work (bgm_hmc_kernel), so a partly filled
  // last round costs its ceil(active waves / 4) waves per SIMD: 12 waves (3 per SIMD, the fastest per tile by ~4 %) always.
  auto wide_waves = [&](void) { return std::getenv("BGM_WIDE_W8") ? 8 : BGM_WAVES_WIDE_HMC; };
  const int ww = wide_waves();
#define LAUNCH_HMC(KTQ_, NTX_, NH_, W)                                                                              \
  {                                                                                                                 \
    const int grid = (int)std::max<long long>(1, std::min<long long>((tiles + W - 1) / W, h->n_cus));               \
    auto k = bgm_hmc_kernel<KTQ_, NTX_, NH_, W>;                                                                    \
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * W), lds, stream, ka);                                               \
    BGM_HIP_CHECK(hipGetLastError());                                                                               \
    return BGM_OK;                                                                                                  \
  }
  if (s->precision != 0) {      // split precision (bgm_kernels.h, PREC 2): the biases are the only LDS-resident data, everything else streams
    ka.blob = s->sx3_bias_dev; ka.m = s->sx3_meta; ka.hx3 = s->sx3_dev;
    const int ldx = s->lds_bytes_sx3;
    constexpr int W = BGM_SX3_WAVES_DEFAULT;      // (12 waves: three per SIMD inside the 168-register line; 8 measured slower, round 6)
#define LAUNCH_SX3(KTQ_, NH_, X4_)                                                                                  \
    {                                                                                                               \
      const int grid = (int)std::max<long long>(1, std::min<long long>((tiles + W - 1) / W, h->n_cus));             \
      auto k = bgm_hmc_kernel<KTQ_, 0, NH_, W, 2, X4_>;                                                             \
      BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, ldx)); \
      hipLaunchKernelGGL(k, dim3(grid), dim3(64 * W), ldx, stream, ka);                                             \
      BGM_HIP_CHECK(hipGetLastError());                                                                             \
      return BGM_OK;                                                                                                \
    }
#define X(KTQ_, NH_)                                                                                                \
    if (s->KTQ == KTQ_ && s->NH == NH_) {                                                                           \
      if ((s->sx3_meta.p & 3) == 0) LAUNCH_SX3(KTQ_, NH_, true)                                                              \
      LAUNCH_SX3(KTQ_, NH_, false)                                                                                  \
    }
    X(1, 5) X(1, 3)
#undef X
#undef LAUNCH_SX3
    BGM_NO_VARIANT(s)
  }
#define X(KTQ_, NTX_, NH_)                                                                                          \
  if (s->KTQ == KTQ_ && s->NTX == NTX_ && s->NH == NH_) {                                                           \
    if constexpr (NTX_ == 0) {                                                                                      \
      if (ww == 8) LAUNCH_HMC(KTQ_, NTX_, NH_, 8)                                                                   \
      LAUNCH_HMC(KTQ_, NTX_, NH_, BGM_WAVES_WIDE_HMC)                                                               \
    } else LAUNCH_HMC(KTQ_, NTX_, NH_, BGM_WAVES)                                                                   \
  }
  BGM_BGM_VARIANTS(X)
#undef X
#undef LAUNCH_HMC
  BGM_NO_VARIANT(s)
}

extern "C" int bgm_bgm_hmc_adapt(bgm_handle *h, float *step, const double *acc_prob_sum, int32_t it, double n_chains,
                                 float target, float rate, void *stream_) {
  if (!h || !step || !acc_prob_sum || !(n_chains > 0)) { bgm_set_error("bgm_bgm_hmc_adapt: bad argument"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  hipLaunchKernelGGL(bgm_hmc_adapt_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, step, acc_prob_sum, it, n_chains, target, rate);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bgm_predict_draws(bgm_handle *h, const float *draws, int64_t n, int64_t row_base, int32_t n_draws,
                                     int32_t burn_in, uint64_t seed, const int32_t *slot, int32_t k_slots, float *cells,
                                     float *full, float *var_full, int32_t add_noise, void *stream_) {
  if (!h || !h->bgm_state || !bst(h)->configured) { bgm_set_error("bgm_bgm_predict_draws: not configured"); return BGM_E_STATE; }
  if (n <= 0 || n_draws <= 0) return BGM_OK;
  if (!draws || (!cells && !full && !var_full) || (cells && (!slot || k_slots <= 0))) { bgm_set_error("bgm_bgm_predict_draws: bad pointers"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (gxb_wanted(bst(h))) return gxb_predict_draws(h, bst(h), draws, n, row_base, n_draws, burn_in, seed, slot, k_slots, cells, full, var_full, add_noise, stream);
  int rc = bgm_build_blob(h, stream);
  if (rc) return rc;
  BgmState *s = bst(h);
  BgmPredKArgs ka{};
  ka.blob = s->blob_dev; ka.draws = draws; ka.n = n; ka.row_base = row_base; ka.n_draws = n_draws; ka.burn_in = burn_in;
  ka.k_slots = k_slots; ka.slot = slot; ka.cells = cells; ka.full = full; ka.var_full = var_full; ka.add_noise = add_noise;
  ka.k0 = (unsigned)(seed & 0xFFFFFFFFull); ka.k1 = (unsigned)(seed >> 32); ka.m = s->meta;
  const long long work = ((n + 15) / 16) * (long long)n_draws;
  const int grid = (int)std::max<long long>(1, std::min<long long>((work + BGM_WAVES - 1) / BGM_WAVES, (long long)h->n_cus));
  const int lds = s->lds_bytes;
#define X(KTQ_, NTX_, NH_)                                                                                          \
  if (s->KTQ == KTQ_ && s->NTX == NTX_ && s->NH == NH_) {                                                           \
    auto k = bgm_predict_kernel<KTQ_, NTX_, NH_, BGM_WAVES>;                                                        \
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * BGM_WAVES), lds, stream, ka);                                       \
    BGM_HIP_CHECK(hipGetLastError());                                                                               \
    return BGM_OK;                                                                                                  \
  }
  BGM_BGM_VARIANTS(X)
#undef X
  BGM_NO_VARIANT(s)
}
