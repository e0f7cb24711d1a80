// bnn_egm_kernels.h -- EGM warm start of CausalBGM with Bayesian networks (use_bnn=True) on gfx950.
//
// replaces (src/bayesgm/models/causalbgm/base.py, nets = BayesianFullyConnectedNet of networks/bnn.py:4-38):
//   train_disc_step :305-330  -> bnn_egm_disc_step_kernel  (the encoder call z_ = e_net(v) is a noisy Flipout call)
//   train_gen_step  :332-377  -> bnn_egm_gen_step_kernel   (nine Flipout calls: g x 3, e x 2, f x 2, h x 2)
// The latent discriminator dz_net stays the deterministic Discriminator of networks/base.py:338-385; its forward,
// backward and gradient penalty are the routines of egm_kernels.h.  Gradients follow oracle/bnn.py
// (egm_disc_step_grads, egm_gen_step_grads).  One launch of one 512-thread workgroup per step.
#pragma once
#include "bnn_kernels.h"
#include "egm_kernels.h"
#include "egm_chain.h"
#include "egm_chain_bnn.h"

static_assert(BNN_THREADS == EGM_THREADS, "the Flipout and discriminator routines share one workgroup");

struct BnnEgmArgs {
  BnnNet net[4];                         // g, e, f, h
  EgmDisc dz;
  float *theta, *m, *v, *grad;           // Bayesian nets [g | e | f | h], Adam slots of g_pre_optimizer, gradient
  float *theta_d, *m_d, *v_d, *grad_d;   // discriminator, Adam slots of d_pre_optimizer
  int n_gen, B, q, p, wmax, dmax, z0, z1, z2, binary, use_z_rec;
  const float *z;                        // [B x q] prior sample of the step
  const int *idx;                        // [B] panel rows
  const float *v_, *x_, *y_;             // panel
  float eps;                             // gradient-penalty interpolation coefficient
  uint32_t k0, k1, stream;               // noise key; call c of the step uses stream + c (oracle/bnn.py EGM_CALLS)
  uint32_t row0;                         // sign words of minibatch row b are keyed by row0 + b (a rank's share of a global minibatch; 0 otherwise)
  EgmAdam adam;
  float *ws, *out;
  int apply, disc_lds;
  int wide;                              // general generator step: eps / dW of the nine calls by bnn_egm_gen_noise_wide_kernel, Adam by its own launch
};

// eps and dW = sigma * eps of the discriminator step's one noisy encoder call, over the chip (general encoder widths): the call cache
// sits behind the gathered covariates (and, in bnn_egm_disc_step_kernel's workspace, behind zhat); same draws as bnn_noise
static __global__ __launch_bounds__(EGM_THREADS) void bnn_egm_disc_noise_wide_kernel(BnnEgmArgs a, int with_zhat) {
  const BnnNet &n = a.net[BNN_E];
  float *wp = a.ws;
  auto take = [&](int cnt) { float *r = wp; wp += (cnt + 3) & ~3; return r; };
  float *vb = take(a.B * a.p);
  if (with_zhat) take(a.B * a.q);
  BnnCache k;
  bnn_cache(n, a.B, wp, k, vb);
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.lin[l] * n.lout[l];
    const float *rho = a.theta + n.woff[l] + cnt;
    float *e = k.eps + n.eoff[l], *d = k.dW + n.eoff[l];
    for (int i = blockIdx.x * EGM_THREADS + threadIdx.x; i < (cnt + 3) >> 2; i += gridDim.x * EGM_THREADS) {
      const f32x4 z = box_muller4(philox4x32_10((uint32_t)i, (uint32_t)l | ((uint32_t)n.net_id << 16), a.stream, BNN_TAG_EPS, a.k0, a.k1));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = 4 * i + u;
        if (idx < cnt) { e[idx] = z[u]; d[idx] = (BNN_SCALE_EPS + softplus_acc(rho[idx])) * z[u]; }
      }
    }
  }
}

static __global__ __launch_bounds__(EGM_THREADS) void bnn_egm_disc_step_kernel(BnnEgmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  EgmCtx c{(int)threadIdx.x, egm_lds};
  BnnCtx cb{(int)threadIdx.x, egm_lds};
  const int B = a.B, q = a.q, p = a.p;
  float *wp = a.ws;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  float *vb = take(B * p), *zhat = take(B * q);
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const int b = k / p; vb[k] = a.v_[(long long)a.idx[b] * p + (k - b * p)]; }
  __syncthreads();
  BnnCache ke;
  bnn_cache(a.net[BNN_E], B, wp, ke, vb);
  float *z_ = bnn_fwd(cb, a.theta, a.net[BNN_E], ke, B, a.k0, a.k1, a.stream, a.row0, a.wide != 0);     // noisy encoder call (fixed in this step)
  for (int k = c.tid; k < B * q; k += EGM_THREADS) zhat[k] = a.z[k] * a.eps + z_[k] * (1.0f - a.eps);
  __syncthreads();
  float *arena = a.disc_lds ? (egm_lds + 64) : wp;
  float *ap_ = arena;
  EgmDiscCache kf, kr, kh;
  egm_disc_cache(a.dz, B, ap_, kf, z_);
  float *overlay = ap_;
  egm_disc_cache(a.dz, B, ap_, kr, const_cast<float *>(a.z));
  float *da = ap_; ap_ += B * a.dmax;
  float *du = ap_; ap_ += B * a.dmax;
  egm_disc_fwd(c, a.theta_d, a.dz, kf, B);
  egm_disc_fwd(c, a.theta_d, a.dz, kr, B);
  float sf = 0.0f, sr = 0.0f;
  {
    const float *of_ = egm_dk_out(a.dz, kf, B), *or_ = egm_dk_out(a.dz, kr, B);
    for (int b = c.tid; b < B; b += EGM_THREADS) { sf += of_[b]; sr += or_[b]; }
  }
  sf = egm_block_sum(c, sf); sr = egm_block_sum(c, sr);
  const float dz_loss = (-sr + sf) / (float)B;
  egm_disc_bwd(c, a.theta_d, a.grad_d, a.dz, kf, true, 1.0f / (float)B, nullptr, da, du, nullptr, B, false, 1.0f);
  egm_disc_bwd(c, a.theta_d, a.grad_d, a.dz, kr, true, -1.0f / (float)B, nullptr, da, du, nullptr, B, true, 1.0f);
  kh.in = zhat; kh.base = kf.base;
  egm_disc_fwd(c, a.theta_d, a.dz, kh, B);
  const float gp = egm_disc_gp(c, a.theta_d, a.grad_d, a.dz, kh, overlay, B, 10.0f, a.dmax);
  __syncthreads();
  if (a.apply) egm_adam(c, a.theta_d, a.m_d, a.v_d, a.grad_d, a.dz.n_params, a.adam);
  if (c.tid == 0 && a.out) { a.out[0] = dz_loss; a.out[1] = dz_loss + 10.0f * gp; }
}

// train_disc_step with the discriminator passes as register-chained row tiles (egm_chain.h; inference-mode normalisation, the
// default layer shapes, B = 16 / 32): the noisy encoder call stays on the Flipout routines, its output z_ is laid out as row
// tiles in LDS, and the three discriminator passes, the gradient GEMMs and Adam are ech_disc_tail.
// NTL > 0: the Flipout encoder runs as a row-tile chain too (egm_chain_bnn.h: inference-mode input normalisation, 64-wide hidden
// layers, ceil(p / 16) = NTL); NTL = 0: it stays on the phase-machine routines.
static __global__ __launch_bounds__(EGM_THREADS) void bnn_egm_disc_noise_kernel(BnnEgmArgs a, EcbCall C) {     // grid: ECB_NOISE_PARTS
  ecb_noise(a.theta, a.net[BNN_E], C, a.ws, a.B, a.k0, a.k1, a.stream, threadIdx.x, blockIdx.x, ECB_NOISE_PARTS, a.row0);
}
// T0: latent input tiles (q <= 16 T0); with two, the fourth pass's stash of the tail lives in global scratch (workspace offset C.xh)
template <int NTL, int T1, int T2, int T3, int NB, int T0 = 1>
static __global__ __launch_bounds__(EGM_THREADS) void bnn_egm_disc_chain_kernel(BnnEgmArgs a, EcbCall C) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  BnnCtx cb{(int)threadIdx.x, egm_lds};
  const int tid = threadIdx.x, B = a.B, q = a.q, p = a.p;
  float *wp = a.ws;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  float *vb = take(B * p);
  constexpr int ZW = 16 * T0;
  const EchP P = ech_layout<T1, T2, T3, T0>(a.dz);
  const EchLds<T1, T2, T3, T0> M(egm_lds, P, B);
  if constexpr (NTL > 0) {
    // the call's perturbations and sign words were drawn by the launch before (bnn_egm_disc_noise_kernel)
    ech_fill_params<T1, T2, T3, T0>(M.par, P, a.theta_d, a.dz, tid, EGM_THREADS);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
    if (wave < NB) {
      const int row = 16 * wave + j;
      f32x4 zf[T0];
      ecb_encoder<4, NTL, T0>(a.theta, a.net[BNN_E], a.ws + C.dW, reinterpret_cast<const uint32_t *>(a.ws + C.sg) + (long long)row * a.net[BNN_E].swords,
                              a.v_ + (long long)a.idx[row] * p, zf, j, g);
#pragma unroll
      for (int t = 0; t < T0; ++t) *reinterpret_cast<f32x4 *>(M.zt + row * ZW + 16 * t + 4 * g) = zf[t];
    }
    __syncthreads();
  } else {
    for (int k = tid; k < B * p; k += EGM_THREADS) { const int b = k / p; vb[k] = a.v_[(long long)a.idx[b] * p + (k - b * p)]; }
    __syncthreads();
    BnnCache ke;
    bnn_cache(a.net[BNN_E], B, wp, ke, vb);
    const float *z_ = bnn_fwd(cb, a.theta, a.net[BNN_E], ke, B, a.k0, a.k1, a.stream, a.row0, a.wide != 0);     // noisy encoder call (fixed in this step)
    for (int k = tid; k < ZW * B; k += EGM_THREADS) { const int b = k / ZW, i = k - b * ZW; const float t = z_[b * q + min(i, q - 1)]; M.zt[k] = i < q ? t : 0.0f; }
    ech_fill_params<T1, T2, T3, T0>(M.par, P, a.theta_d, a.dz, tid, EGM_THREADS);
    __syncthreads();
  }
  const EchDiscIo io{a.theta_d, a.m_d, a.v_d, a.grad_d, a.adam, a.apply, q, a.out, a.z, a.eps, a.ws + C.xh};
  ech_disc_tail<T1, T2, T3, NB, T0>(io, a.dz, P, M, tid);
}

// The workspace of the general generator step: gathered batch, the call caches of the nine calls (stream + 0 .. 8: g(z), g(z) again, e(v),
// e(v_), g(z_), f, f again, h, h again), the discriminator cache; `rest` = what follows (the backward's scratch).  One function for the
// step kernel and for the launch that draws the calls' perturbations ahead of it.
struct BnnEgmGenLayout { float *vb, *xb, *yb, *v_, *fin, *hin, *z_, *rest; BnnCache k[9]; EgmDiscCache kd; float *G[9], *GS[9]; };
// net of call c (oracle/bnn.py EGM_CALLS)
__device__ __forceinline__ int bnn_egm_call_net(int call) { return (call == 2 || call == 3) ? BNN_E : (call == 5 || call == 6) ? BNN_F : (call >= 7) ? BNN_H : BNN_G; }
__device__ __forceinline__ void bnn_egm_gen_layout(const BnnEgmArgs &a, BnnEgmGenLayout &L) {
  const int B = a.B, p = a.p;
  const BnnNet &G = a.net[BNN_G], &E = a.net[BNN_E], &F = a.net[BNN_F], &H = a.net[BNN_H];
  float *wp = a.ws;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  L.vb = take(B * p); L.xb = take(B); L.yb = take(B);
  bnn_cache(G, B, wp, L.k[0], a.z);
  bnn_cache(G, B, wp, L.k[1], a.z);
  bnn_cache(E, B, wp, L.k[2], L.vb);
  L.z_ = L.k[2].H + (long long)B * E.hoff[E.n_layers];      // the output of e(v) (bnn_fwd)
  L.v_ = take(B * p);
  bnn_cache(E, B, wp, L.k[3], L.v_);
  bnn_cache(G, B, wp, L.k[4], L.z_);
  egm_disc_cache(a.dz, B, wp, L.kd, L.z_);
  L.fin = take(B * F.dims[0]); L.hin = take(B * H.dims[0]);
  bnn_cache(F, B, wp, L.k[5], L.fin);
  bnn_cache(F, B, wp, L.k[6], L.fin);
  bnn_cache(H, B, wp, L.k[7], L.hin);
  bnn_cache(H, B, wp, L.k[8], L.hin);
  // the upstream gradients of every layer of every call (bnn_bwd's G / GS), kept for the gradient-tile launch (wide steps)
  for (int cidx = 0; cidx < 9; ++cidx) {
    const BnnNet &n = a.net[bnn_egm_call_net(cidx)];
    L.G[cidx] = take(B * n.hoff[n.n_layers + 1]); L.GS[cidx] = take(B * n.hoff[n.n_layers + 1]);
  }
  L.rest = wp;
}
// eps and dW = sigma * eps of the nine calls, over the chip: grid (parts, 9); same draws as bnn_noise
static __global__ __launch_bounds__(EGM_THREADS) void bnn_egm_gen_noise_wide_kernel(BnnEgmArgs a) {
  BnnEgmGenLayout L;
  bnn_egm_gen_layout(a, L);
  const int call = blockIdx.y;
  const BnnNet &n = a.net[bnn_egm_call_net(call)];
  const BnnCache &k = L.k[call];
  const uint32_t stream = a.stream + (uint32_t)call;
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.lin[l] * n.lout[l];
    const float *rho = a.theta + n.woff[l] + cnt;
    float *e = k.eps + n.eoff[l], *d = k.dW + n.eoff[l];
    for (int i = blockIdx.x * EGM_THREADS + threadIdx.x; i < (cnt + 3) >> 2; i += gridDim.x * EGM_THREADS) {
      const f32x4 z = box_muller4(philox4x32_10((uint32_t)i, (uint32_t)l | ((uint32_t)n.net_id << 16), stream, BNN_TAG_EPS, a.k0, a.k1));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = 4 * i + u;
        if (idx < cnt) { e[idx] = z[u]; d[idx] = (BNN_SCALE_EPS + softplus_acc(rho[idx])) * z[u]; }
      }
    }
  }
}

static __global__ __launch_bounds__(EGM_THREADS) void bnn_egm_gen_step_kernel(BnnEgmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  EgmCtx c{(int)threadIdx.x, egm_lds};
  BnnCtx cb{(int)threadIdx.x, egm_lds};
  const int B = a.B, q = a.q, p = a.p, z0 = a.z0, z1 = a.z1, z2 = a.z2;
  const float invB = 1.0f / (float)B;
  BnnEgmGenLayout L;
  bnn_egm_gen_layout(a, L);
  float *wp = L.rest;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  float *vb = L.vb, *xb = L.xb, *yb = L.yb;
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const int b = k / p; vb[k] = a.v_[(long long)a.idx[b] * p + (k - b * p)]; }
  for (int b = c.tid; b < B; b += EGM_THREADS) { xb[b] = a.x_[a.idx[b]]; yb[b] = a.y_[a.idx[b]]; }
  __syncthreads();
  const BnnNet &G = a.net[BNN_G], &E = a.net[BNN_E], &F = a.net[BNN_F], &H = a.net[BNN_H];
  const int wg = p + 1, nf = F.dims[0], nh = H.dims[0], of = F.dims[F.n_layers], oh = H.dims[H.n_layers];
  const bool so = a.wide != 0;      // the calls' eps / dW were drawn by bnn_egm_gen_noise_wide_kernel: sign words only
  // ---- forward: nine calls, noise streams stream + 0..8 in the order of oracle/bnn.py EGM_CALLS
  BnnCache &g1 = L.k[0], &g1s = L.k[1], &e1 = L.k[2], &e2 = L.k[3], &g2 = L.k[4], &cf = L.k[5], &cfs = L.k[6], &ch = L.k[7], &chs = L.k[8];
  const float *gz = bnn_fwd(cb, a.theta, G, g1, B, a.k0, a.k1, a.stream + 0u, a.row0, so);      // g(z): v_ = [:, :p]
  const float *gzs = bnn_fwd(cb, a.theta, G, g1s, B, a.k0, a.k1, a.stream + 1u, a.row0, so);    // g(z) again: variance head penalty
  float *z_ = bnn_fwd(cb, a.theta, E, e1, B, a.k0, a.k1, a.stream + 2u, a.row0, so);            // z_ = e(v)
  float *v_ = L.v_;
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const int b = k / p; v_[k] = gz[b * wg + (k - b * p)]; }
  __syncthreads();
  const float *z__ = bnn_fwd(cb, a.theta, E, e2, B, a.k0, a.k1, a.stream + 3u, a.row0, so);     // z__ = e(v_)
  const float *gv = bnn_fwd(cb, a.theta, G, g2, B, a.k0, a.k1, a.stream + 4u, a.row0, so);      // g(z_): v__ = [:, :p]
  EgmDiscCache &kd = L.kd;
  egm_disc_fwd(c, a.theta_d, a.dz, kd, B);
  float *fin = L.fin, *hin = L.hin;
  for (int k = c.tid; k < B * nf; k += EGM_THREADS) {
    const int b = k / nf, i = k - b * nf;
    fin[k] = (i < z0 + z1) ? z_[b * q + i] : xb[b];
  }
  for (int k = c.tid; k < B * nh; k += EGM_THREADS) {
    const int b = k / nh, i = k - b * nh;
    hin[k] = (i < z0) ? z_[b * q + i] : z_[b * q + z1 + i];
  }
  __syncthreads();
  const float *fo = bnn_fwd(cb, a.theta, F, cf, B, a.k0, a.k1, a.stream + 5u, a.row0, so);
  const float *fs = bnn_fwd(cb, a.theta, F, cfs, B, a.k0, a.k1, a.stream + 6u, a.row0, so);
  const float *ho = bnn_fwd(cb, a.theta, H, ch, B, a.k0, a.k1, a.stream + 7u, a.row0, so);
  const float *hs = bnn_fwd(cb, a.theta, H, chs, B, a.k0, a.k1, a.stream + 8u, a.row0, so);
  // ---- losses
  float l_v = 0.0f, l_z = 0.0f, l_x = 0.0f, l_y = 0.0f, s_g = 0.0f, s_f = 0.0f, s_h = 0.0f, adv = 0.0f;
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const int b = k / p; const float t = vb[k] - gv[b * wg + (k - b * p)]; l_v = fmaf(t, t, l_v); }
  for (int k = c.tid; k < B * q; k += EGM_THREADS) { const float t = a.z[k] - z__[k]; l_z = fmaf(t, t, l_z); }
  const float *dout_ = egm_dk_out(a.dz, kd, B);
  for (int b = c.tid; b < B; b += EGM_THREADS) {
    const float xl = ho[b * oh], yl = fo[b * of];
    if (a.binary) l_x += fmaxf(xl, 0.0f) - xl * xb[b] + log1pf(expf(-fabsf(xl)));
    else l_x += (xl - xb[b]) * (xl - xb[b]);
    l_y += (yl - yb[b]) * (yl - yb[b]);
    s_g += gzs[b * wg + p] * gzs[b * wg + p];
    s_f += fs[b * of + of - 1] * fs[b * of + of - 1];
    s_h += hs[b * oh + oh - 1] * hs[b * oh + oh - 1];
    adv -= dout_[b];
  }
  l_v = egm_block_sum(c, l_v) / (float)(B * p);
  l_z = egm_block_sum(c, l_z) / (float)(B * q);
  l_x = egm_block_sum(c, l_x) * invB;
  l_y = egm_block_sum(c, l_y) * invB;
  const float sig = (egm_block_sum(c, s_g) + egm_block_sum(c, s_f) + egm_block_sum(c, s_h)) * invB;
  adv = egm_block_sum(c, adv) * invB;
  const float zrec = a.use_z_rec ? 1.0f : 0.0f;
  // ---- backward
  const int wmax = a.wmax;
  float *d = take(B * wmax), *ds = take(B * wmax), *t0 = take(B * wmax), *t1 = take(B * wmax);
  float *dv_ = take(B * wmax), *dzsum = take(B * wmax), *dtmp = take(B * wmax), *da = take(B * wmax), *du = take(B * wmax);
  float *dfin = take(B * wmax), *dfin2 = take(B * wmax), *dhin = take(B * wmax), *dhin2 = take(B * wmax);
  // (wide steps: the upstream gradient of a call is written where bnn_bwd keeps the gradients of all its layers, L.G / L.GS)
  float *dd, *dds;
  auto up = [&](int ci, const BnnNet &N, float *dflt) { return a.wide ? L.G[ci] + (long long)B * N.hoff[N.n_layers] : dflt; };
  auto up2 = [&](int ci, const BnnNet &N, float *dflt) { return a.wide ? L.GS[ci] + (long long)B * N.hoff[N.n_layers] : dflt; };
  // z__ branch: e (call 2, input v_) -> g (call 1)
  dd = up(3, E, d); dds = up2(3, E, ds);
  for (int k = c.tid; k < B * q; k += EGM_THREADS) dd[k] = zrec * (-2.0f / (float)(B * q)) * (a.z[k] - z__[k]);
  __syncthreads();
  bnn_bwd(cb, a.theta, a.grad, E, e2, dd, dds, t0, t1, dv_, B, true, false, a.wide ? L.G[3] : nullptr, a.wide ? L.GS[3] : nullptr);
  dd = up(0, G, d); dds = up2(0, G, ds);
  for (int k = c.tid; k < B * wg; k += EGM_THREADS) { const int b = k / wg, i = k - b * wg; dd[k] = (i < p) ? dv_[b * p + i] : 0.0f; }
  __syncthreads();
  bnn_bwd(cb, a.theta, a.grad, G, g1, dd, dds, t0, t1, nullptr, B, true, false, a.wide ? L.G[0] : nullptr, a.wide ? L.GS[0] : nullptr);
  // variance-head penalty of the second g(z) call
  dd = up(1, G, d); dds = up2(1, G, ds);
  for (int k = c.tid; k < B * wg; k += EGM_THREADS) { const int i = k % wg; dd[k] = (i == p) ? 0.001f * 2.0f * gzs[k] * invB : 0.0f; }
  __syncthreads();
  bnn_bwd(cb, a.theta, a.grad, G, g1s, dd, dds, t0, t1, nullptr, B, true, true, a.wide ? L.G[1] : nullptr, a.wide ? L.GS[1] : nullptr);
  // v__ branch: g (call 3, input z_)
  dd = up(4, G, d); dds = up2(4, G, ds);
  for (int k = c.tid; k < B * wg; k += EGM_THREADS) {
    const int b = k / wg, i = k - b * wg;
    dd[k] = (i < p) ? (-2.0f / (float)(B * p)) * (vb[b * p + i] - gv[k]) : 0.0f;
  }
  __syncthreads();
  bnn_bwd(cb, a.theta, a.grad, G, g2, dd, dds, t0, t1, dzsum, B, true, true, a.wide ? L.G[4] : nullptr, a.wide ? L.GS[4] : nullptr);
  // adversarial branch through the fixed discriminator
  float *gd_scratch = take(a.dz.n_params);
  egm_disc_bwd(c, a.theta_d, gd_scratch, a.dz, kd, true, -invB, nullptr, da, du, dtmp, B, false, 1.0f);
  for (int k = c.tid; k < B * q; k += EGM_THREADS) dzsum[k] += dtmp[k];
  __syncthreads();
  // f: mean call, variance call
  dd = up(5, F, d); dds = up2(5, F, ds);
  for (int k = c.tid; k < B * of; k += EGM_THREADS) { const int b = k / of, i = k - b * of; dd[k] = (i == 0) ? 2.0f * (fo[k] - yb[b]) * invB : 0.0f; }
  __syncthreads();
  bnn_bwd(cb, a.theta, a.grad, F, cf, dd, dds, t0, t1, dfin, B, true, false, a.wide ? L.G[5] : nullptr, a.wide ? L.GS[5] : nullptr);
  dd = up(6, F, d); dds = up2(6, F, ds);
  for (int k = c.tid; k < B * of; k += EGM_THREADS) { const int i = k % of; dd[k] = (i == of - 1) ? 0.001f * 2.0f * fs[k] * invB : 0.0f; }
  __syncthreads();
  bnn_bwd(cb, a.theta, a.grad, F, cfs, dd, dds, t0, t1, dfin2, B, true, true, a.wide ? L.G[6] : nullptr, a.wide ? L.GS[6] : nullptr);
  // h: mean call, variance call
  dd = up(7, H, d); dds = up2(7, H, ds);
  for (int k = c.tid; k < B * oh; k += EGM_THREADS) {
    const int b = k / oh, i = k - b * oh;
    dd[k] = (i == 0) ? (a.binary ? (1.0f / (1.0f + expf(-ho[k])) - xb[b]) * invB : 2.0f * (ho[k] - xb[b]) * invB) : 0.0f;
  }
  __syncthreads();
  bnn_bwd(cb, a.theta, a.grad, H, ch, dd, dds, t0, t1, dhin, B, true, false, a.wide ? L.G[7] : nullptr, a.wide ? L.GS[7] : nullptr);
  dd = up(8, H, d); dds = up2(8, H, ds);
  for (int k = c.tid; k < B * oh; k += EGM_THREADS) { const int i = k % oh; dd[k] = (i == oh - 1) ? 0.001f * 2.0f * hs[k] * invB : 0.0f; }
  __syncthreads();
  bnn_bwd(cb, a.theta, a.grad, H, chs, dd, dds, t0, t1, dhin2, B, true, true, a.wide ? L.G[8] : nullptr, a.wide ? L.GS[8] : nullptr);
  dd = up(2, E, d); dds = up2(2, E, ds);
  for (int k = c.tid; k < B * q; k += EGM_THREADS) {
    const int b = k / q, i = k - b * q;
    float t = dzsum[k];
    if (i < z0) t += dfin[b * nf + i] + dfin2[b * nf + i] + dhin[b * nh + i] + dhin2[b * nh + i];
    else if (i < z0 + z1) t += dfin[b * nf + i] + dfin2[b * nf + i];
    else if (i < z0 + z1 + z2) t += dhin[b * nh + (i - z1)] + dhin2[b * nh + (i - z1)];
    dd[k] = t;
  }
  __syncthreads();
  bnn_bwd(cb, a.theta, a.grad, E, e1, dd, dds, t0, t1, nullptr, B, true, true, a.wide ? L.G[2] : nullptr, a.wide ? L.GS[2] : nullptr);
  if (a.apply && !a.wide) egm_adam(c, a.theta, a.m, a.v, a.grad, a.n_gen, a.adam);      // (wide: egm_dp_adam_kernel behind this launch)
  if (c.tid == 0 && a.out) {
    a.out[0] = adv; a.out[1] = l_v; a.out[2] = l_z; a.out[3] = l_x; a.out[4] = l_y;
    a.out[5] = adv + (l_v + zrec * l_z) + (l_x + l_y) + 0.001f * sig;
  }
}

// The parameter-gradient tiles of a wide generator step: every layer of g, e, f, h, the calls of a net in the order the step kernel
// accumulated them (first call assigns, the others add: a tile is always the same wave's, so the sums keep their order).  grid (parts, 4 nets)
static __global__ __launch_bounds__(EGM_THREADS) void bnn_egm_gen_dw_wide_kernel(BnnEgmArgs a) {
  __shared__ float red[32];
  BnnCtx c{(int)threadIdx.x, red};
  BnnEgmGenLayout L;
  bnn_egm_gen_layout(a, L);
  const int id = blockIdx.y;      // BNN_G, BNN_E, BNN_F, BNN_H
  const BnnNet &n = a.net[id];
  const int order[4][3] = {{0, 1, 4}, {3, 2, -1}, {5, 6, -1}, {7, 8, -1}};      // [net id][position]: g1, g1s, g2 | e2, e1 | cf, cfs | ch, chs
  for (int l = 0; l < n.n_layers; ++l)
    for (int j = 0; j < 3; ++j) {
      const int ci = order[id][j];
      if (ci < 0) continue;
      bnn_bwd_params(c, a.theta, a.grad, n, L.k[ci], l, L.G[ci] + (long long)a.B * n.hoff[l + 1], L.GS[ci] + (long long)a.B * n.hoff[l + 1], a.B, j > 0,
                     (int)blockIdx.x, (int)gridDim.x);
    }
}

// train_gen_step as row-tile chains (egm_chain_bnn.h) + its gradient / Adam launch
template <int NTL, int NB, bool PAD = false, int T0 = 1>
static __global__ __launch_bounds__(EGM_THREADS) void bnn_egm_gen_chain_kernel(BnnEgmArgs a, const EcbTab *tab, float *thetaT) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  ecb_gen_chain<BnnEgmArgs, 4, NTL, 4, 2, 1, NB, PAD, T0>(a, *tab, thetaT, egm_lds);
}
static __global__ __launch_bounds__(EGM_THREADS) void bnn_egm_gen_noise_kernel(BnnEgmArgs a, const EcbTab *tab) { ecb_gen_noise<BnnEgmArgs>(a, *tab, a.ws, a.row0); }
template <int NB>
static __global__ __launch_bounds__(EGM_THREADS) void bnn_egm_gen_dw_kernel(BnnEgmArgs a, const EcbTab *tab, const int *tiles, float *thetaT) {
  ecb_gen_dw<BnnEgmArgs, NB>(a, *tab, tiles, thetaT);
}
