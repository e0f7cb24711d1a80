// bnf_api.hip -- host side of the fixed-normalisation Bayesian-network sampling path (bnf_kernels.h): shape check, fragment
// index tables, buffers, launch sequences.  Entered from bgm_bnn_logpost / bgm_bnn_mh_run / bgm_bnn_effects (bnn_sample_api.hip)
// when the session's nets have the default shapes and params['bnn_norm'] = "fixed"; everything else keeps the batch-statistics
// kernels of bnn_sample_kernels.h.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bgm_host.h"
#include "bnf_host.h"
#include "bnf_kernels.h"
#include "bnn_state.h"

namespace {

struct BnfState {
  BnfPlan P{};
  int KSc = 0;                          // compiled k-step count (>= P.KS)
  int KSFc = 0;                         // compiled k-step count of the effects kernel (>= P.KSF)
  BnfLayerDesc lay[14], lay_e[4];       // Flipout kernels of g | h | f (sampler sets) and of f in the effects layout
  int n_w = 0, n_b = 0, n_n = 0, n_we = 0, n_be = 0, n_ne = 0;
  BnfWElem *w_dev = nullptr, *we_dev = nullptr;
  int *npos_dev = nullptr, *npos_e_dev = nullptr;      // noise kernel's position tables
  int n_calls = 0, n_calls_e = 0;
  BnfBElem *b_dev = nullptr, *be_dev = nullptr;
  BnfNElem *n_dev = nullptr, *ne_dev = nullptr;
  float *blob_dev = nullptr, *eblob_dev = nullptr, *sf_dev = nullptr, *esf_dev = nullptr;
  // per-run buffers, grown on demand
  float *dw_dev = nullptr; size_t dw_cap = 0;          // perturbation sets
  void *sg_dev = nullptr; size_t sg_cap = 0;           // sign groups
  float *pair_dev = nullptr;                           // (1, 0): the two treatments of a binary model
  unsigned *queue_dev = nullptr;                       // [16] item counters: 0..7 sampler, 8..15 effects
  int lds_mh = 0, lds_eff = 0;
};

const int kKS[] = {3, 4, 5, 6, 8};
const int kKSF[] = {1, 2, 3, 4, 8};

bool default_head(const BnnNet &n, int in) {
  return n.bn_fixed == 1 && !n.heads && !n.mv && n.n_layers == 4 && n.dims[0] == in && n.dims[1] == 64 && n.dims[2] == 32 && n.dims[3] == 8 &&
         n.dims[4] == 2;
}

// plan + element tables; false when the session is outside this path
struct BnfTabs { std::vector<BnfWElem> W, WE; std::vector<BnfBElem> B, BE; std::vector<BnfNElem> N, NE; };
bool bnf_build(const BnnState *s, BnfState &st, BnfTabs &tb) {
  std::vector<BnfWElem> &W = tb.W; std::vector<BnfBElem> &B = tb.B; std::vector<BnfNElem> &N = tb.N;
  const bgm_bnn_config &c = s->cfg;
  const int q = s->q, p = s->p, z0 = c.z_dims[0], z1 = c.z_dims[1], z2 = c.z_dims[2];
  const BnnNet &G = s->net[BNN_G], &H = s->net[BNN_H], &F = s->net[BNN_F];
  if (G.bn_fixed != 1 || G.heads || G.mv || G.n_layers != 6 || G.dims[0] != q || G.dims[6] != p + 1) return false;
  for (int l = 1; l <= 5; ++l) if (G.dims[l] != 64) return false;
  if (z0 + z2 < 1 || !default_head(H, z0 + z2) || !default_head(F, z0 + z1 + 1)) return false;
  if (q < 1 || q > 31 || p + 1 > 208 || p < 4) return false;
  const int need = (q + 1 + 3) / 4;
  int KS = 0;
  for (int k : kKS) if (k >= need) { KS = k; break; }
  if (!KS) return false;
  const int T0 = (KS + 3) / 4, NTL = (p + 1 + 15) / 16, head = 4 * T0 + 11;
  BnfPlan &P = st.P;
  P = BnfPlan{};
  P.q = q; P.p = p; P.z0 = z0; P.z1 = z1; P.z2 = z2; P.binary = c.binary_treatment;
  P.KS = KS; P.NTL = NTL;
  P.fg0 = 0; P.fgh = 4 * T0; P.fgl = P.fgh + 64; P.fh = P.fgl + 4 * NTL; P.ff = P.fh + head; P.n_frags = P.ff + head;
  P.bg0 = 0; P.bgh = 4; P.bgl = 20; P.bh = 20 + NTL; P.bf = P.bh + 8;
  P.bias_off = P.n_frags * 256;
  P.norm_off = P.bias_off + 16 * (P.bf + 8);
  P.shift_off = P.norm_off + 3 * T0 * 32;
  P.blob_floats = P.shift_off + 3 * T0 * 16;
  P.set_floats = P.n_frags * 256;
  const int f_in = z0 + z1 + 1, need_f = (f_in + 3) / 4;
  int KSF = 0;
  for (int k : kKSF) if (k >= need_f) { KSF = k; break; }
  if (!KSF) return false;
  const int T0F = (KSF + 3) / 4;
  P.KSF = KSF;
  P.e_frags = 4 * T0F + 11;
  P.e_bias_off = P.e_frags * 256;
  P.e_norm_off = P.e_bias_off + 16 * 8;
  P.e_shift_off = P.e_norm_off + T0F * 32;
  P.e_blob_floats = P.e_shift_off + T0F * 16;
  st.KSFc = KSF;
  if ((size_t)P.blob_floats * 4 > 160 * 1024) return false;
  st.KSc = KS;
  st.lds_mh = P.blob_floats * 4;
  st.lds_eff = (((P.e_blob_floats + 3) & ~3) + 16 * BNF_MAX_DOSES) * 4;      // blob + per-wave dose accumulators (<= 16 waves)

  auto ext = [&](int net, int k) { return net == 0 ? k : net == 1 ? (k < z0 ? k : k + z1) : (k < z0 + z1 ? k : q); };
  W.clear(); B.clear(); N.clear();
  st.n_calls = 0; st.n_calls_e = 0;
  const BnnNet *nets[3] = {&G, &H, &F};
  const int fbase[3] = {P.fg0, P.fh, P.ff}, bbase[3] = {P.bg0, P.bh, P.bf};
  int nl = 0;
  for (int ni = 0; ni < 3; ++ni) {
    const BnnNet &n = *nets[ni];
    int fb = fbase[ni], bt = bbase[ni];
    for (int l = 0; l < n.n_layers; ++l) {
      const int in = n.dims[l], out = n.dims[l + 1], T = l == 0 ? T0 : (in + 15) / 16, MT = (out + 15) / 16;
      st.lay[nl] = BnfLayerDesc{(int)W.size(), in * out, l, n.net_id, st.n_calls};
      st.n_calls += (in * out + 3) / 4; ++nl;
      const bool head3 = ni > 0 && l == 2, head4 = ni > 0 && l == 3;
      for (int k = 0; k < in; ++k)
        for (int o = 0; o < out; ++o) {
          BnfWElem e{};
          e.loc = n.woff[l] + k * out + o; e.rho = e.loc + in * out; e.rep = 1; e.scale = l > 0 ? BGM_LRS_W : 1.0f;
          int t, gg, r, mt = o >> 4, j = o & 15;
          if (l == 0) { const int x = ext(ni, k); t = x >> 4; r = (x & 15) >> 2; gg = x & 3; }
          else if (head4) { t = 0; gg = k >> 1; r = k & 1; e.rep = 4; }
          else { t = k >> 4; gg = (k & 15) >> 2; r = k & 3; }
          if (head3) j = 4 * (o >> 1) + (o & 1);
          e.pos = (((fb + mt * T + t) * 64) + gg * 16 + j) * 4 + r;
          W.push_back(e);
        }
      const int boff = n.woff[l] + 2 * in * out;
      for (int o = 0; o < out; ++o) {
        BnfBElem e{};
        e.src = boff + o; e.rep = head4 ? 4 : 1;
        const int j = head3 ? 4 * (o >> 1) + (o & 1) : (o & 15);
        e.pos = 16 * (bt + (o >> 4)) + j;
        B.push_back(e);
      }
      fb += T * MT; bt += MT;
    }
    for (int x = 0; x < 16 * T0; ++x) {
      int k = -1;
      for (int kk = 0; kk < n.dims[0]; ++kk) if (ext(ni, kk) == x) k = kk;
      const int sb = x >> 4, r = (x & 15) >> 2, gg = x & 3;
      BnfNElem e{};
      e.gamma = k >= 0 ? n.off + k : -1; e.beta = k >= 0 ? n.off + n.dims[0] + k : -1;
      e.pos_sc = P.norm_off + ((ni * T0 + sb) * 2) * 16 + gg * 4 + r;
      e.pos_sh = e.pos_sc + 16;
      e.pos_shift = P.shift_off + (ni * T0 + sb) * 16 + gg * 4 + r;
      e.shift = k >= 0 ? 31 - k : 0;
      N.push_back(e);
    }
  }
  // effects layout: the outcome net alone, first layer over its own input (z0, z1, x)
  tb.WE.clear(); tb.BE.clear(); tb.NE.clear();
  {
    const BnnNet &n = F;
    int fb = 0, bt = 0;
    for (int l = 0; l < n.n_layers; ++l) {
      const int in = n.dims[l], out = n.dims[l + 1], T = l == 0 ? T0F : (in + 15) / 16, MT = (out + 15) / 16;
      st.lay_e[l] = BnfLayerDesc{(int)tb.WE.size(), in * out, l, n.net_id, st.n_calls_e};
      st.n_calls_e += (in * out + 3) / 4;
      const bool head3 = l == 2, head4 = l == 3;
      for (int k = 0; k < in; ++k)
        for (int o = 0; o < out; ++o) {
          BnfWElem e{};
          e.loc = n.woff[l] + k * out + o; e.rho = e.loc + in * out; e.rep = 1; e.scale = l > 0 ? BGM_LRS_W : 1.0f;
          int t_, gg, r, mt = o >> 4, j = o & 15;
          if (l == 0) { t_ = k >> 4; r = (k & 15) >> 2; gg = k & 3; }
          else if (head4) { t_ = 0; gg = k >> 1; r = k & 1; e.rep = 4; }
          else { t_ = k >> 4; gg = (k & 15) >> 2; r = k & 3; }
          if (head3) j = 4 * (o >> 1) + (o & 1);
          e.pos = (((fb + mt * T + t_) * 64) + gg * 16 + j) * 4 + r;
          tb.WE.push_back(e);
        }
      const int boff = n.woff[l] + 2 * in * out;
      for (int o = 0; o < out; ++o) {
        BnfBElem e{};
        e.src = boff + o; e.rep = head4 ? 4 : 1;
        const int j = head3 ? 4 * (o >> 1) + (o & 1) : (o & 15);
        e.pos = 16 * (bt + (o >> 4)) + j;
        tb.BE.push_back(e);
      }
      fb += T * MT; bt += MT;
    }
    for (int x = 0; x < 16 * T0F; ++x) {
      const int k = x < n.dims[0] ? x : -1;
      const int sb = x >> 4, r = (x & 15) >> 2, gg = x & 3;
      BnfNElem e{};
      e.gamma = k >= 0 ? n.off + k : -1; e.beta = k >= 0 ? n.off + n.dims[0] + k : -1;
      e.pos_sc = P.e_norm_off + (sb * 2) * 16 + gg * 4 + r;
      e.pos_sh = e.pos_sc + 16;
      e.pos_shift = P.e_shift_off + sb * 16 + gg * 4 + r;
      e.shift = k >= 0 ? 31 - k : 0;
      tb.NE.push_back(e);
    }
  }
  return nl == 14;
}

void bnf_release(BnfState *st) {
  if (!st) return;
  for (void *p : {(void *)st->w_dev, (void *)st->b_dev, (void *)st->n_dev, (void *)st->we_dev, (void *)st->be_dev, (void *)st->ne_dev,
                  (void *)st->esf_dev, (void *)st->npos_dev, (void *)st->npos_e_dev, (void *)st->blob_dev, (void *)st->eblob_dev, (void *)st->sf_dev,
                  (void *)st->dw_dev, st->sg_dev, (void *)st->pair_dev, (void *)st->queue_dev})
    if (p) hipFree(p);
  delete st;
}

// ---- kernel variants: (R row tiles per wave, WAVES per workgroup).  The shipped one is (BNF_R, BNF_W); a development build
// (-D BNF_ALL_CFGS) also carries the others for the bench shape (KS = 3) and picks by the environment variable BGM_BNF_CFG=r<R>w<W>.
#ifndef BNF_R
#define BNF_R 2
#define BNF_W 8
#endif
#ifndef BNF_ER
#define BNF_ER 1
#define BNF_EW 12
#endif
struct BnfCfg { int R, W; };
BnfCfg env_cfg(const char *name, BnfCfg d) {
#ifdef BNF_ALL_CFGS
  if (const char *e = std::getenv(name)) {
    int r = 0, w = 0;
    if (std::sscanf(e, "r%dw%d", &r, &w) == 2) d = BnfCfg{r, w};
  }
#endif
  (void)name;
  return d;
}
using MhFn = void (*)(BnfMhArgs);
using EffFn = void (*)(BnfEffArgs);
template <int KS, int MODE>
MhFn mh_fn_ks(BnfCfg &c) {
#ifdef BNF_ALL_CFGS
  if constexpr (KS == 3) {
    if (c.R == 1 && c.W == 8) return bnf_mh_kernel<KS, 1, 8, MODE>;
    if (c.R == 2 && c.W == 8) return bnf_mh_kernel<KS, 2, 8, MODE>;
    if (c.R == 2 && c.W == 4) return bnf_mh_kernel<KS, 2, 4, MODE>;
    if (c.R == 3 && c.W == 4) return bnf_mh_kernel<KS, 3, 4, MODE>;
    if (c.R == 1 && c.W == 12) return bnf_mh_kernel<KS, 1, 12, MODE>;
  }
#endif
  c = BnfCfg{BNF_R, BNF_W};
  return bnf_mh_kernel<KS, BNF_R, BNF_W, MODE>;
}
template <int KS>
EffFn eff_fn_ks(BnfCfg &c) {
#ifdef BNF_ALL_CFGS
  if constexpr (KS == 1) {
    if (c.R == 1 && c.W == 8) return bnf_effects_kernel<KS, 1, 8>;
    if (c.R == 2 && c.W == 8) return bnf_effects_kernel<KS, 2, 8>;
    if (c.R == 2 && c.W == 4) return bnf_effects_kernel<KS, 2, 4>;
    if (c.R == 4 && c.W == 4) return bnf_effects_kernel<KS, 4, 4>;
    if (c.R == 1 && c.W == 12) return bnf_effects_kernel<KS, 1, 12>;
    if (c.R == 1 && c.W == 16) return bnf_effects_kernel<KS, 1, 16>;
    if (c.R == 2 && c.W == 12) return bnf_effects_kernel<KS, 2, 12>;
  }
#endif
  c = BnfCfg{BNF_ER, BNF_EW};
  return bnf_effects_kernel<KS, BNF_ER, BNF_EW>;
}
// c: in = the requested variant, out = the one returned
template <int MODE>
MhFn mh_fn(int KS, BnfCfg &c) {
  c = env_cfg("BGM_BNF_CFG", BnfCfg{BNF_R, BNF_W});
  switch (KS) {
    case 3: return mh_fn_ks<3, MODE>(c);
    case 4: return mh_fn_ks<4, MODE>(c);
    case 5: return mh_fn_ks<5, MODE>(c);
    case 6: return mh_fn_ks<6, MODE>(c);
    default: return mh_fn_ks<8, MODE>(c);
  }
}
EffFn eff_fn(int KSF, BnfCfg &c) {
  c = env_cfg("BGM_BNF_ECFG", BnfCfg{BNF_ER, BNF_EW});
  switch (KSF) {
    case 1: return eff_fn_ks<1>(c);
    case 2: return eff_fn_ks<2>(c);
    case 3: return eff_fn_ks<3>(c);
    case 4: return eff_fn_ks<4>(c);
    default: return eff_fn_ks<8>(c);
  }
}

int bnf_session(bgm_handle *h, BnnState *s, BnfState *&st, hipStream_t stream) {
  st = static_cast<BnfState *>(s->bnf);
  if (!st) {
    if (std::getenv("BGM_BNF_OFF")) return 1;
    BnfState *n = new BnfState();
    BnfTabs tb;
    if (!bnf_build(s, *n, tb)) { delete n; s->bnf_unsupported = true; return 1; }
    n->n_w = (int)tb.W.size(); n->n_b = (int)tb.B.size(); n->n_n = (int)tb.N.size();
    n->n_we = (int)tb.WE.size(); n->n_be = (int)tb.BE.size(); n->n_ne = (int)tb.NE.size();
    auto up = [&](void **dst, const void *src, size_t bytes) {
      if (hipMalloc(dst, bytes) != hipSuccess) return false;
      return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    static const float pair_host[2] = {1.0f, 0.0f};
    std::vector<int> np(tb.W.size()), npe(tb.WE.size());
    for (size_t i = 0; i < tb.W.size(); ++i) np[i] = tb.W[i].pos | ((tb.W[i].rep - 1) << 28);
    for (size_t i = 0; i < tb.WE.size(); ++i) npe[i] = tb.WE[i].pos | ((tb.WE[i].rep - 1) << 28);
    bool ok = up((void **)&n->npos_dev, np.data(), np.size() * sizeof(int)) && up((void **)&n->npos_e_dev, npe.data(), npe.size() * sizeof(int)) && up((void **)&n->w_dev, tb.W.data(), tb.W.size() * sizeof(BnfWElem)) && up((void **)&n->b_dev, tb.B.data(), tb.B.size() * sizeof(BnfBElem)) &&
              up((void **)&n->n_dev, tb.N.data(), tb.N.size() * sizeof(BnfNElem)) && up((void **)&n->we_dev, tb.WE.data(), tb.WE.size() * sizeof(BnfWElem)) &&
              up((void **)&n->be_dev, tb.BE.data(), tb.BE.size() * sizeof(BnfBElem)) && up((void **)&n->ne_dev, tb.NE.data(), tb.NE.size() * sizeof(BnfNElem)) &&
              up((void **)&n->pair_dev, pair_host, sizeof(pair_host));
    ok = ok && hipMalloc((void **)&n->blob_dev, sizeof(float) * n->P.blob_floats) == hipSuccess &&
         hipMalloc((void **)&n->eblob_dev, sizeof(float) * n->P.e_blob_floats) == hipSuccess &&
         hipMalloc((void **)&n->sf_dev, sizeof(float) * n->P.set_floats) == hipSuccess &&
         hipMalloc((void **)&n->esf_dev, sizeof(float) * n->P.e_frags * 256) == hipSuccess &&
         hipMalloc((void **)&n->queue_dev, 16 * sizeof(unsigned)) == hipSuccess;
    if (!ok) { bnf_release(n); bgm_set_error("bnf: device allocation failed"); return BGM_E_HIP; }
    s->bnf = n; s->bnf_valid = false;
    st = n;
  }
  if (!s->bnf_valid) {
    const BnfPlan &P = st->P;
    BGM_HIP_CHECK(hipMemsetAsync(st->blob_dev, 0, sizeof(float) * P.blob_floats, stream));
    BGM_HIP_CHECK(hipMemsetAsync(st->eblob_dev, 0, sizeof(float) * P.e_blob_floats, stream));
    BGM_HIP_CHECK(hipMemsetAsync(st->sf_dev, 0, sizeof(float) * P.set_floats, stream));
    BGM_HIP_CHECK(hipMemsetAsync(st->esf_dev, 0, sizeof(float) * P.e_frags * 256, stream));
    BnfPackArgs pa{};
    pa.theta = s->theta_dev; pa.w = st->w_dev; pa.n_w = st->n_w; pa.b = st->b_dev; pa.n_b = st->n_b; pa.ne = st->n_dev; pa.n_n = st->n_n;
    pa.blob = st->blob_dev; pa.sf = st->sf_dev; pa.bias_off = P.bias_off;
    hipLaunchKernelGGL(bnf_pack_kernel, dim3(64), dim3(256), 0, stream, pa);
    pa.w = st->we_dev; pa.n_w = st->n_we; pa.b = st->be_dev; pa.n_b = st->n_be; pa.ne = st->ne_dev; pa.n_n = st->n_ne;
    pa.blob = st->eblob_dev; pa.sf = st->esf_dev; pa.bias_off = P.e_bias_off;
    hipLaunchKernelGGL(bnf_pack_kernel, dim3(16), dim3(256), 0, stream, pa);
    BGM_HIP_CHECK(hipGetLastError());
    s->bnf_valid = true;
  }
  (void)h;
  return BGM_OK;
}

// zero: the padding positions of the perturbation sets must be zero and the noise kernel writes real elements only; the set
// boundaries move with (n_blocks, n_doses), so the used part is cleared at every call (tens of MB, once per call)
int grow(void **p, size_t *cap, size_t bytes, hipStream_t stream, bool zero) {
  if (bytes > *cap) {
    BGM_HIP_CHECK(hipStreamSynchronize(stream));
    if (*p) BGM_HIP_CHECK(hipFree(*p));
    *p = nullptr; *cap = 0;
    BGM_HIP_CHECK(hipMalloc(p, bytes));
    *cap = bytes;
  }
  if (zero && bytes) BGM_HIP_CHECK(hipMemsetAsync(*p, 0, bytes, stream));
  return BGM_OK;
}

struct SgLayout { uint4 *g; uint32_t *gout; uint4 *h, *f; size_t bytes; };
SgLayout sg_layout(void *base, long long n, int states_ghf, int states_f_only) {
  SgLayout L{};
  char *p = (char *)base;
  const size_t n16 = (size_t)n * 16;
  L.g = (uint4 *)p; p += (size_t)states_ghf * BNF_NG_G * n16;
  L.gout = (uint32_t *)p; p += (size_t)states_ghf * n * BNF_GOUT * 4;
  L.h = (uint4 *)p; p += (size_t)states_ghf * BNF_NG_H * n16;
  L.f = (uint4 *)p; p += (size_t)std::max(states_ghf, states_f_only) * BNF_NG_H * n16;
  L.bytes = (size_t)(p - (char *)base);
  return L;
}

template <class K>
int set_lds(K kernel, int bytes) {
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return BGM_OK;
}

// effects = true: sets of the outcome net in the effects layout
void launch_noise(const BnfState *st, bool effects, float *dw, long long set_floats, int n_blocks, int n_states, uint64_t seed, uint32_t stream0,
                  int block0, hipStream_t stream) {
  BnfNoiseArgs na{};
  const int n_lay = effects ? 4 : 14;
  for (int i = 0; i < n_lay; ++i) na.lay[i] = effects ? st->lay_e[i] : st->lay[i];
  na.n_lay = n_lay; na.n_calls = effects ? st->n_calls_e : st->n_calls; na.npos = effects ? st->npos_e_dev : st->npos_dev;
  na.sf = effects ? st->esf_dev : st->sf_dev; na.dw = dw; na.set_floats = set_floats;
  na.n_states = n_states; na.k0 = (uint32_t)seed; na.k1 = (uint32_t)(seed >> 32); na.stream0 = stream0; na.block0 = block0;
  hipLaunchKernelGGL(bnf_noise_kernel, dim3((na.n_calls + 1023) / 1024, n_blocks * n_states), dim3(256), 0, stream, na);
}

void launch_signs(const SgLayout &L, long long n, int bs, int block0, int n_states, int nets, uint64_t seed, uint32_t stream0, unsigned *queue,
                  hipStream_t stream) {
  BnfSignArgs sa{};
  sa.g = L.g; sa.gout = L.gout; sa.h = L.h; sa.f = L.f; sa.n = n; sa.bs = bs; sa.block0 = block0; sa.n_states = n_states; sa.nets = nets;
  sa.k0 = (uint32_t)seed; sa.k1 = (uint32_t)(seed >> 32); sa.stream0 = stream0; sa.queue = queue;
  hipLaunchKernelGGL(bnf_signs_kernel, dim3((unsigned)((n + 255) / 256), n_states), dim3(256), 0, stream, sa);
}

int grid_for(const bgm_handle *h) { return std::max(8, (h->n_cus / 8) * 8); }

}  // namespace

void bnf_free(void *state) { bnf_release(static_cast<BnfState *>(state)); }

// returns 1 when the session is outside this path (the caller continues with the batch-statistics kernels), 0 when handled,
// a negative BGM_E_* code on failure
int bnf_logpost(bgm_handle *h, BnnState *s, const float *x, const float *y, const float *v, const float *z, int64_t n, int32_t block_rows,
                int32_t block0, uint64_t seed, uint32_t stream_id, float *out, hipStream_t stream) {
  if (s->bnf_unsupported) return 1;
  BnfState *st;
  int rc = bnf_session(h, s, st, stream);
  if (rc) return rc;
  const BnfPlan &P = st->P;
  BnfCfg c;
  MhFn fn = mh_fn<0>(st->KSc, c);
  const int n_blocks = (int)((n + block_rows - 1) / block_rows);
  rc = grow((void **)&st->dw_dev, &st->dw_cap, sizeof(float) * (size_t)n_blocks * P.set_floats, stream, true);
  if (rc) return rc;
  SgLayout L = sg_layout(nullptr, n, 1, 0);
  rc = grow(&st->sg_dev, &st->sg_cap, L.bytes, stream, false);
  if (rc) return rc;
  L = sg_layout(st->sg_dev, n, 1, 0);
  launch_noise(st, false, st->dw_dev, P.set_floats, n_blocks, 1, seed, stream_id, block0, stream);
  launch_signs(L, n, block_rows, block0, 1, 7, seed, stream_id, st->queue_dev, stream);
  BnfMhArgs a{};
  a.pl = P; a.blob = st->blob_dev; a.dw = st->dw_dev;
  a.sg.g = L.g; a.sg.gout = L.gout; a.sg.h = L.h; a.sg.f = L.f;
  a.x = x; a.y = y; a.v = v; a.z = const_cast<float *>(z); a.n = n; a.row_base = 0;
  a.bs = block_rows; a.n_blocks = n_blocks; a.block0 = block0;
  a.groups_per_block = ((block_rows + 15) / 16 + c.R - 1) / c.R; a.n_items = n_blocks * a.groups_per_block; a.n_states = 1;
  a.mode = 0; a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.out = out; a.queue = st->queue_dev;
  rc = set_lds(fn, st->lds_mh);
  if (rc) return rc;
  hipLaunchKernelGGL(fn, dim3(grid_for(h)), dim3(64 * c.W), st->lds_mh, stream, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

namespace {
// effects of ONE draw (state z) -> adrf column / ite column d
int effects_of(bgm_handle *h, BnfState *st, const float *z, long long n, int bs, int n_blocks, int block0, long long row_base, int n_doses,
               const float *xvals, uint64_t seed, uint32_t stream0, uint32_t it_noise, int sample_y, double *sum_out, long long sum_stride,
               float *ite_out, long long ite_stride, float *dw_eff, const SgLayout &L, hipStream_t stream) {
  const BnfPlan &P = st->P;
  BnfCfg c;
  EffFn fn = eff_fn(st->KSFc, c);
  const long long eset = (long long)P.e_frags * 256;
  launch_noise(st, true, dw_eff, eset, n_blocks, n_doses, seed, stream0, block0, stream);
  launch_signs(L, n, bs, block0, n_doses, 4, seed, stream0, st->queue_dev + 8, stream);
  BnfEffArgs ea{};
  ea.pl = P; ea.eblob = st->eblob_dev; ea.dw = dw_eff; ea.sgf = L.f; ea.z = z; ea.n = n; ea.row_base = row_base;
  ea.bs = bs; ea.n_blocks = n_blocks; ea.block0 = block0;
  ea.groups_per_block = ((bs + 15) / 16 + c.R - 1) / c.R; ea.n_items = n_blocks * ea.groups_per_block; ea.n_doses = n_doses;
  ea.xvals = xvals; ea.k0 = (uint32_t)seed; ea.k1 = (uint32_t)(seed >> 32); ea.sample_y = sample_y; ea.it_noise = it_noise;
  ea.sum_out = sum_out; ea.sum_stride = sum_stride; ea.ite_out = ite_out; ea.ite_stride = ite_stride; ea.queue = st->queue_dev + 8;
  int rc = set_lds(fn, st->lds_eff);
  if (rc) return rc;
  hipLaunchKernelGGL(fn, dim3(grid_for(h)), dim3(64 * c.W), st->lds_eff, stream, ea);
  return BGM_OK;
}
}  // namespace

int bnf_mh_run(bgm_handle *h, BnnState *s, const bgm_bnn_mh_args *g, hipStream_t stream) {
  if (s->bnf_unsupported) return 1;
  const int n_doses = g->effect == 1 ? g->n_doses : (g->effect == 2 ? 2 : 0);
  if (n_doses > BNF_MAX_DOSES) return 1;
  BnfState *st;
  int rc = bnf_session(h, s, st, stream);
  if (rc) return rc;
  const BnfPlan &P = st->P;
  BnfCfg c;
  MhFn fn = mh_fn<1>(st->KSc, c);
  const long long n = g->n;
  const int bs = g->block_rows, n_blocks = (int)((n + bs - 1) / bs), q = s->q;
  const long long eset = (long long)P.e_frags * 256;
  const size_t dw_mh = (size_t)n_blocks * 2 * P.set_floats, dw_eff = (size_t)n_blocks * n_doses * eset;
  rc = grow((void **)&st->dw_dev, &st->dw_cap, sizeof(float) * (dw_mh + dw_eff), stream, true);
  if (rc) return rc;
  SgLayout L = sg_layout(nullptr, n, 2, n_doses);
  rc = grow(&st->sg_dev, &st->sg_cap, L.bytes, stream, false);
  if (rc) return rc;
  L = sg_layout(st->sg_dev, n, 2, n_doses);
  float *dw_eff_dev = st->dw_dev + dw_mh;
  BnfMhArgs a{};
  a.pl = P; a.blob = st->blob_dev; a.dw = st->dw_dev;
  a.sg.g = L.g; a.sg.gout = L.gout; a.sg.h = L.h; a.sg.f = L.f;
  a.x = g->x_dev; a.y = g->y_dev; a.v = g->v_dev; a.z = g->state_dev; a.n = n; a.row_base = g->row_base;
  a.bs = bs; a.n_blocks = n_blocks; a.block0 = g->block0;
  a.groups_per_block = ((bs + 15) / 16 + c.R - 1) / c.R; a.n_items = n_blocks * a.groups_per_block; a.n_states = 2;
  a.mode = 1; a.q_sd = g->q_sd; a.q_sd_blocks = g->q_sd_blocks_dev;
  a.k0 = (uint32_t)g->seed; a.k1 = (uint32_t)(g->seed >> 32); a.acc_count = g->acc_count_dev; a.queue = st->queue_dev;
  rc = set_lds(fn, st->lds_mh);
  if (rc) return rc;
  const int grid = grid_for(h);
#ifdef BNF_PROF
  static unsigned long long *prof_dev = nullptr;
  if (!prof_dev) hipMalloc((void **)&prof_dev, 256);
  hipMemsetAsync(prof_dev, 0, 256, stream);
  a.prof = prof_dev;
#endif
  for (int i = 0; i < g->n_iters; ++i) {
    const int it = g->it_begin + i;
    launch_noise(st, false, st->dw_dev, P.set_floats, n_blocks, 2, g->seed, 2u * (uint32_t)it, g->block0, stream);
    launch_signs(L, n, bs, g->block0, 2, 7, g->seed, 2u * (uint32_t)it, st->queue_dev, stream);
    a.it = it; a.init = (i == 0 && g->init) ? 1 : 0;
    a.acc_blocks = g->acc_blocks_dev ? g->acc_blocks_dev + (long long)i * n_blocks : nullptr;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * c.W), st->lds_mh, stream, a);
    const int d = it - g->burn_in;
    if (d >= 0 && d < g->n_keep) {
      if (g->draws_dev)
        BGM_HIP_CHECK(hipMemcpyAsync(g->draws_dev + (long long)d * n * q, g->state_dev, sizeof(float) * n * q, hipMemcpyDeviceToDevice, stream));
      if (g->effect) {
        rc = effects_of(h, st, g->state_dev, n, bs, n_blocks, g->block0, g->row_base, n_doses, g->effect == 1 ? g->x_values_dev : st->pair_dev,
                        g->seed, 0x40000000u + (uint32_t)d * (uint32_t)n_doses, (uint32_t)it, g->sample_y,
                        g->effect == 1 ? g->adrf_sum_dev + d : nullptr, g->n_keep, g->effect == 2 ? g->ite_dev + d : nullptr, g->n_keep,
                        dw_eff_dev, L, stream);
        if (rc) return rc;
      }
    }
  }
#ifdef BNF_PROF
  {
    unsigned long long hp[32];
    hipStreamSynchronize(stream);
    hipMemcpy(hp, prof_dev, sizeof(hp), hipMemcpyDeviceToHost);
    const double d = (double)grid * std::max(1, g->n_iters);
    fprintf(stderr, "[BNF_PROF] cycles per workgroup-launch (wave 0, R=%d W=%d): prologue %.0f g-first %.0f g-hidden %.0f g-last %.0f h %.0f f %.0f rest %.0f\n",
            c.R, c.W, hp[0] / d, hp[1] / d, hp[2] / d, hp[3] / d, hp[4] / d, hp[5] / d, hp[6] / d);
    const double dw = d * c.W;
    fprintf(stderr, "[BNF_PROF] wave busy cycles per launch: mean %.0f, max (over all launches) %llu; by wave index:", hp[7] / dw, hp[8]);
    for (int w = 0; w < c.W; ++w) fprintf(stderr, " %.0f", hp[9 + w] / d);
    fprintf(stderr, "\n");
  }
#endif
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

int bnf_effects(bgm_handle *h, BnnState *s, const float *draws, int64_t n, int32_t block_rows, int32_t block0, int64_t row_base, int32_t n_keep,
                int32_t it0, uint64_t seed, int32_t effect, int32_t sample_y, const float *x_values, int32_t n_doses, double *adrf_sum,
                float *ite, hipStream_t stream) {
  if (s->bnf_unsupported) return 1;
  const int nd = effect == 1 ? n_doses : 2;
  if (nd > BNF_MAX_DOSES) return 1;
  BnfState *st;
  int rc = bnf_session(h, s, st, stream);
  if (rc) return rc;
  const BnfPlan &P = st->P;
  const int bs = block_rows, n_blocks = (int)((n + bs - 1) / bs), q = s->q;
  const long long eset = (long long)P.e_frags * 256;
  rc = grow((void **)&st->dw_dev, &st->dw_cap, sizeof(float) * (size_t)n_blocks * nd * eset, stream, true);
  if (rc) return rc;
  SgLayout L = sg_layout(nullptr, n, 0, nd);
  rc = grow(&st->sg_dev, &st->sg_cap, L.bytes, stream, false);
  if (rc) return rc;
  L = sg_layout(st->sg_dev, n, 0, nd);
  for (int d = 0; d < n_keep; ++d) {
    rc = effects_of(h, st, draws + (long long)d * n * q, n, bs, n_blocks, block0, row_base, nd, effect == 1 ? x_values : st->pair_dev, seed,
                    0x40000000u + (uint32_t)d * (uint32_t)nd, (uint32_t)(it0 + d), sample_y, effect == 1 ? adrf_sum + d : nullptr, n_keep,
                    effect == 2 ? ite + d : nullptr, n_keep, st->dw_dev, L, stream);
    if (rc) return rc;
  }
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}
