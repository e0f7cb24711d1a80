// bnf_build.h -- host side shared by the two users of the bnf kernels (bnf_api.hip: Bayesian nets with inference-mode normalisation;
// bnf_det_api.hip: deterministic nets of any data width): plan, fragment index tables, device tables, packing.
#pragma once
#include <algorithm>
#include <vector>

#include "bgm_host.h"
#include "bnf_kernels.h"

// One network as the builder sees it: widths and where its parameters sit in the session's flat device array `theta`.
struct BnfNetSrc {
  int n_layers = 0, net_id = 0;
  int dims[9] = {0};
  int woff[8] = {0};      // kernel [in x out] of layer l (loc for a Flipout layer)
  int roff[8] = {0};      // rho of layer l, or -1 (deterministic layer)
  int boff[8] = {0};      // bias
  int gamma = -2, beta = -2;   // input normalisation gamma / beta [in] (inference mode: x * gamma / sqrt(1 + eps) + beta); -2: none (identity)
};
struct BnfSrc { BnfNetSrc g, h, f; int q = 0, p = 0, z0 = 0, z1 = 0, z2 = 0, binary = 0; bool det = false; };

struct BnfState {
  BnfPlan P{};
  bool wide = false;                    // g's last layer streamed from the blob (does not fit the LDS next to the rest)
  int KSc = 0;                          // compiled k-step count (>= P.KS)
  int KSFc = 0;                         // compiled k-step count of the effects kernel (>= P.KSF)
  BnfLayerDesc lay[14], lay_e[4];       // Flipout kernels of g | h | f (sampler sets) and of f in the effects layout
  int n_w = 0, n_b = 0, n_n = 0, n_we = 0, n_be = 0, n_ne = 0;
  BnfWElem *w_dev = nullptr, *we_dev = nullptr;
  int *npos_dev = nullptr, *npos_e_dev = nullptr;      // noise kernel's position tables
  // split precision (bnx_kernels.h, bnx_api.hip): second position tables, re-encoded blobs (built at the first call in that mode)
  int *posx_dev = nullptr, *posx_e_dev = nullptr;
  float *blobx_dev = nullptr, *eblobx_dev = nullptr;
  bool x3_valid = false;
  int n_calls = 0, n_calls_e = 0;
  BnfBElem *b_dev = nullptr, *be_dev = nullptr;
  BnfNElem *n_dev = nullptr, *ne_dev = nullptr;
  float *blob_dev = nullptr, *eblob_dev = nullptr, *sf_dev = nullptr, *esf_dev = nullptr;
  // per-run buffers, grown on demand
  float *dw_dev = nullptr; size_t dw_cap = 0;          // perturbation sets
  float *mh_dev = nullptr; size_t mh_cap = 0;          // split precision: proposals [n x q] | log posteriors [2][n] of an iteration
  void *sg_dev = nullptr; size_t sg_cap = 0;           // sign groups
  float *pair_dev = nullptr;                           // (1, 0): the two treatments of a binary model
  unsigned *queue_dev = nullptr;                       // [16] item counters: 0..7 sampler, 8..15 effects
  float *theta_dev = nullptr;                          // deterministic sessions: their own copy of the parameters
  int lds_mh = 0, lds_eff = 0;
};
struct BnfTabs { std::vector<BnfWElem> W, WE; std::vector<BnfBElem> B, BE; std::vector<BnfNElem> N, NE; };

static const int kBnfKS[] = {3, 4, 5, 6, 8};
static const int kBnfKSF[] = {1, 2, 3, 4, 8};

static inline bool bnf_default_head(const BnfNetSrc &n, int in) {
  return n.n_layers == 4 && n.dims[0] == in && n.dims[1] == 64 && n.dims[2] == 32 && n.dims[3] == 8 && n.dims[4] == 2;
}

// plan + element tables; false when the model is outside this path (default widths, q <= 31, p >= 4; p + 1 <= 208 for Bayesian nets)
static inline bool bnf_build(const BnfSrc &S, BnfState &st, BnfTabs &tb) {
  std::vector<BnfWElem> &W = tb.W; std::vector<BnfBElem> &B = tb.B; std::vector<BnfNElem> &N = tb.N;
  const int q = S.q, p = S.p, z0 = S.z0, z1 = S.z1, z2 = S.z2;
  const BnfNetSrc &G = S.g, &H = S.h, &F = S.f;
  if (G.n_layers != 6 || G.dims[0] != q || G.dims[6] != p + 1) return false;
  for (int l = 1; l <= 5; ++l) if (G.dims[l] != 64) return false;
  if (z0 + z2 < 1 || !bnf_default_head(H, z0 + z2) || !bnf_default_head(F, z0 + z1 + 1)) return false;
  if (q < 1 || q > 31 || p < 4) return false;
  if (!S.det && p + 1 > 208) return false;          // sign words of g's outputs: BNF_GOUT * 32 columns
  const int need = (q + 1 + 3) / 4;
  int KS = 0;
  for (int k : kBnfKS) if (k >= need) { KS = k; break; }
  if (!KS) return false;
  const int T0 = (KS + 3) / 4, NTL = (p + 1 + 15) / 16, head = 4 * T0 + 11;
  BnfPlan &P = st.P;
  P = BnfPlan{};
  P.q = q; P.p = p; P.z0 = z0; P.z1 = z1; P.z2 = z2; P.binary = S.binary;
  P.KS = KS; P.NTL = NTL;
  // fragments [g first | g hidden | g last | h | f]; a wide plan moves g's last layer behind f and keeps it out of the LDS
  const int n_frags = 4 * T0 + 64 + 4 * NTL + 2 * head;
  const int tail = 16 * (20 + NTL + 16) + 3 * T0 * 32 + 3 * T0 * 16;          // bias tiles + norm + shift, floats
  st.wide = (size_t)(n_frags * 256 + tail) * 4 > 160 * 1024;
  if (st.wide && !S.det) return false;
  P.fg0 = 0; P.fgh = 4 * T0;
  if (!st.wide) { P.fgl = P.fgh + 64; P.fh = P.fgl + 4 * NTL; P.ff = P.fh + head; }
  else { P.fh = P.fgh + 64; P.ff = P.fh + head; P.fgl = P.ff + head; }
  P.n_frags = n_frags;
  P.lds_skip = st.wide ? 4 * NTL * 256 : 0;
  P.bg0 = 0; P.bgh = 4; P.bgl = 20; P.bh = 20 + NTL; P.bf = P.bh + 8;
  P.bias_off = P.n_frags * 256;
  P.norm_off = P.bias_off + 16 * (P.bf + 8);
  P.shift_off = P.norm_off + 3 * T0 * 32;
  P.blob_floats = P.shift_off + 3 * T0 * 16;
  P.set_floats = P.n_frags * 256;
  const int f_in = z0 + z1 + 1, need_f = (f_in + 3) / 4;
  int KSF = 0;
  for (int k : kBnfKSF) if (k >= need_f) { KSF = k; break; }
  if (!KSF) return false;
  const int T0F = (KSF + 3) / 4;
  P.KSF = KSF;
  P.e_frags = 4 * T0F + 11;
  P.e_bias_off = P.e_frags * 256;
  P.e_norm_off = P.e_bias_off + 16 * 8;
  P.e_shift_off = P.e_norm_off + T0F * 32;
  P.e_blob_floats = P.e_shift_off + T0F * 16;
  st.KSFc = KSF;
  if ((size_t)(P.blob_floats - P.lds_skip) * 4 > 160 * 1024) return false;
  st.KSc = KS;
  st.lds_mh = (P.blob_floats - P.lds_skip) * 4;
  st.lds_eff = (((P.e_blob_floats + 3) & ~3) + 16 * BNF_MAX_DOSES) * 4;      // blob + per-wave dose accumulators (<= 16 waves)

  auto ext = [&](int net, int k) { return net == 0 ? k : net == 1 ? (k < z0 ? k : k + z1) : (k < z0 + z1 ? k : q); };
  W.clear(); B.clear(); N.clear();
  st.n_calls = 0; st.n_calls_e = 0;
  const BnfNetSrc *nets[3] = {&G, &H, &F};
  const int fbase[3] = {P.fg0, P.fh, P.ff}, bbase[3] = {P.bg0, P.bh, P.bf};
  int nl = 0;
  auto elems = [&](const BnfNetSrc &n, int l, int fb, int T, bool first_ext, int ni, std::vector<BnfWElem> &Wt) {
    const int in = n.dims[l], out = n.dims[l + 1];
    const bool head3 = in == 32 && out == 8 && n.n_layers == 4 && l == 2, head4 = n.n_layers == 4 && l == 3;
    for (int k = 0; k < in; ++k)
      for (int o = 0; o < out; ++o) {
        BnfWElem e{};
        e.loc = n.woff[l] + k * out + o; e.rho = n.roff[l] >= 0 ? n.roff[l] + k * out + o : -1; e.rep = 1; e.scale = l > 0 ? BGM_LRS_W : 1.0f;
        int t_, gg, r, mt = o >> 4, j = o & 15;
        if (l == 0) { const int x = first_ext ? ext(ni, k) : k; t_ = x >> 4; r = (x & 15) >> 2; gg = x & 3; }
        else if (head4) { t_ = 0; gg = k >> 1; r = k & 1; e.rep = 4; }
        else { t_ = k >> 4; gg = (k & 15) >> 2; r = k & 3; }
        if (head3) j = 4 * (o >> 1) + (o & 1);
        e.pos = (((fb + mt * T + t_) * 64) + gg * 16 + j) * 4 + r;
        // split layout (bnx_kernels.h), fp16 units: a K = 32 block is the fragment pair (t_ & ~1, + 1), slot 2 r + (t_ & 1) of lane gg * 16 + j;
        // a layer of one k-tile is a K = 16 block (bit 27), slot r
        e.posx = T == 1 ? (((fb + mt) * 512 + (gg * 16 + j) * 4 + r) | 0x08000000)
                        : ((fb + mt * T + (t_ & ~1)) * 512 + (gg * 16 + j) * 8 + 2 * r + (t_ & 1));
        e.posx |= (e.rep - 1) << 28;
        Wt.push_back(e);
      }
  };
  auto biases = [&](const BnfNetSrc &n, int l, int bt, std::vector<BnfBElem> &Bt) {
    const int out = n.dims[l + 1];
    const bool head3 = n.n_layers == 4 && l == 2, head4 = n.n_layers == 4 && l == 3;
    for (int o = 0; o < out; ++o) {
      BnfBElem e{};
      e.src = n.boff[l] + o; e.rep = head4 ? 4 : 1;
      const int j = head3 ? 4 * (o >> 1) + (o & 1) : (o & 15);
      e.pos = 16 * (bt + (o >> 4)) + j;
      Bt.push_back(e);
    }
  };
  for (int ni = 0; ni < 3; ++ni) {
    const BnfNetSrc &n = *nets[ni];
    int fb = fbase[ni], bt = bbase[ni];
    for (int l = 0; l < n.n_layers; ++l) {
      const int in = n.dims[l], out = n.dims[l + 1], T = l == 0 ? T0 : (in + 15) / 16, MT = (out + 15) / 16;
      if (ni == 0 && l == 5) fb = P.fgl;              // g's last layer: after the hidden layers, or behind f (wide)
      st.lay[nl] = BnfLayerDesc{(int)W.size(), in * out, l, n.net_id, st.n_calls};
      st.n_calls += (in * out + 3) / 4; ++nl;
      elems(n, l, fb, T, true, ni, W);
      biases(n, l, bt, B);
      fb += T * MT; bt += MT;
    }
    for (int x = 0; x < 16 * T0; ++x) {
      int k = -1;
      for (int kk = 0; kk < n.dims[0]; ++kk) if (ext(ni, kk) == x) k = kk;
      const int sb = x >> 4, r = (x & 15) >> 2, gg = x & 3;
      BnfNElem e{};
      e.gamma = k >= 0 ? (n.gamma >= 0 ? n.gamma + k : -2) : -1; e.beta = k >= 0 && n.beta >= 0 ? n.beta + k : -1;
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
    const BnfNetSrc &n = F;
    int fb = 0, bt = 0;
    for (int l = 0; l < n.n_layers; ++l) {
      const int in = n.dims[l], out = n.dims[l + 1], T = l == 0 ? T0F : (in + 15) / 16, MT = (out + 15) / 16;
      st.lay_e[l] = BnfLayerDesc{(int)tb.WE.size(), in * out, l, n.net_id, st.n_calls_e};
      st.n_calls_e += (in * out + 3) / 4;
      elems(n, l, fb, T, false, 2, tb.WE);
      biases(n, l, bt, tb.BE);
      fb += T * MT; bt += MT;
    }
    for (int x = 0; x < 16 * T0F; ++x) {
      const int k = x < n.dims[0] ? x : -1;
      const int sb = x >> 4, r = (x & 15) >> 2, gg = x & 3;
      BnfNElem e{};
      e.gamma = k >= 0 ? (n.gamma >= 0 ? n.gamma + k : -2) : -1; e.beta = k >= 0 && n.beta >= 0 ? n.beta + k : -1;
      e.pos_sc = P.e_norm_off + (sb * 2) * 16 + gg * 4 + r;
      e.pos_sh = e.pos_sc + 16;
      e.pos_shift = P.e_shift_off + sb * 16 + gg * 4 + r;
      e.shift = k >= 0 ? 31 - k : 0;
      tb.NE.push_back(e);
    }
  }
  return nl == 14;
}

static inline void bnf_release(BnfState *st) {
  if (!st) return;
  for (void *p : {(void *)st->w_dev, (void *)st->b_dev, (void *)st->n_dev, (void *)st->we_dev, (void *)st->be_dev, (void *)st->ne_dev,
                  (void *)st->esf_dev, (void *)st->npos_dev, (void *)st->npos_e_dev, (void *)st->blob_dev, (void *)st->eblob_dev, (void *)st->sf_dev,
                  (void *)st->dw_dev, st->sg_dev, (void *)st->pair_dev, (void *)st->queue_dev, (void *)st->theta_dev, (void *)st->posx_dev,
                  (void *)st->posx_e_dev, (void *)st->blobx_dev, (void *)st->eblobx_dev, (void *)st->mh_dev})
    if (p) hipFree(p);
  delete st;
}

// device copies of the tables + the packed arrays (empty); false on an allocation failure
static inline bool bnf_upload(BnfState *n, const BnfTabs &tb) {
  n->n_w = (int)tb.W.size(); n->n_b = (int)tb.B.size(); n->n_n = (int)tb.N.size();
  n->n_we = (int)tb.WE.size(); n->n_be = (int)tb.BE.size(); n->n_ne = (int)tb.NE.size();
  auto up = [&](void **dst, const void *src, size_t bytes) {
    if (hipMalloc(dst, bytes) != hipSuccess) return false;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  static const float pair_host[2] = {1.0f, 0.0f};
  std::vector<int> np(tb.W.size()), npe(tb.WE.size()), px(tb.W.size()), pxe(tb.WE.size());
  for (size_t i = 0; i < tb.W.size(); ++i) { np[i] = tb.W[i].pos | ((tb.W[i].rep - 1) << 28); px[i] = tb.W[i].posx; }
  for (size_t i = 0; i < tb.WE.size(); ++i) { npe[i] = tb.WE[i].pos | ((tb.WE[i].rep - 1) << 28); pxe[i] = tb.WE[i].posx; }
  bool ok = up((void **)&n->npos_dev, np.data(), np.size() * sizeof(int)) && up((void **)&n->npos_e_dev, npe.data(), npe.size() * sizeof(int)) &&
            up((void **)&n->posx_dev, px.data(), px.size() * sizeof(int)) && up((void **)&n->posx_e_dev, pxe.data(), pxe.size() * sizeof(int)) &&
            up((void **)&n->w_dev, tb.W.data(), tb.W.size() * sizeof(BnfWElem)) && up((void **)&n->b_dev, tb.B.data(), tb.B.size() * sizeof(BnfBElem)) &&
            up((void **)&n->n_dev, tb.N.data(), tb.N.size() * sizeof(BnfNElem)) && up((void **)&n->we_dev, tb.WE.data(), tb.WE.size() * sizeof(BnfWElem)) &&
            up((void **)&n->be_dev, tb.BE.data(), tb.BE.size() * sizeof(BnfBElem)) && up((void **)&n->ne_dev, tb.NE.data(), tb.NE.size() * sizeof(BnfNElem)) &&
            up((void **)&n->pair_dev, pair_host, sizeof(pair_host));
  ok = ok && hipMalloc((void **)&n->blob_dev, sizeof(float) * n->P.blob_floats) == hipSuccess &&
       hipMalloc((void **)&n->eblob_dev, sizeof(float) * n->P.e_blob_floats) == hipSuccess &&
       hipMalloc((void **)&n->sf_dev, sizeof(float) * n->P.set_floats) == hipSuccess &&
       hipMalloc((void **)&n->esf_dev, sizeof(float) * n->P.e_frags * 256) == hipSuccess &&
       hipMalloc((void **)&n->queue_dev, 16 * sizeof(unsigned)) == hipSuccess;
  return ok;
}

// (re)build the two blobs and the sigma fragments from the parameters
static inline int bnf_pack(BnfState *st, const float *theta_dev, hipStream_t stream) {
  const BnfPlan &P = st->P;
  BGM_HIP_CHECK(hipMemsetAsync(st->blob_dev, 0, sizeof(float) * P.blob_floats, stream));
  BGM_HIP_CHECK(hipMemsetAsync(st->eblob_dev, 0, sizeof(float) * P.e_blob_floats, stream));
  BGM_HIP_CHECK(hipMemsetAsync(st->sf_dev, 0, sizeof(float) * P.set_floats, stream));
  BGM_HIP_CHECK(hipMemsetAsync(st->esf_dev, 0, sizeof(float) * P.e_frags * 256, stream));
  BnfPackArgs pa{};
  pa.theta = theta_dev; pa.w = st->w_dev; pa.n_w = st->n_w; pa.b = st->b_dev; pa.n_b = st->n_b; pa.ne = st->n_dev; pa.n_n = st->n_n;
  pa.blob = st->blob_dev; pa.sf = st->sf_dev; pa.bias_off = P.bias_off;
  hipLaunchKernelGGL(bnf_pack_kernel, dim3(64), dim3(256), 0, stream, pa);
  pa.w = st->we_dev; pa.n_w = st->n_we; pa.b = st->be_dev; pa.n_b = st->n_be; pa.ne = st->ne_dev; pa.n_n = st->n_ne;
  pa.blob = st->eblob_dev; pa.sf = st->esf_dev; pa.bias_off = P.e_bias_off;
  hipLaunchKernelGGL(bnf_pack_kernel, dim3(16), dim3(256), 0, stream, pa);
  BGM_HIP_CHECK(hipGetLastError());
  st->x3_valid = false;
  return BGM_OK;
}
