// bgm_egm_kernels.h -- EGM warm start of BGM on gfx950.
//
// replaces (src/bayesgm/models/bgm/base.py):
//   train_disc_step :190-244  -> bgm_egm_disc_step_kernel  (LSGAN targets 0.9 / 0.1 on dz_net and dx_net, optional
//                                                           gradient penalties weighted by `gamma`)
//   train_gen_step  :246-289  -> bgm_egm_gen_step_kernel   (generator g_net + encoder e_net)
// The generator is BaseVariationalNet called with its default training=True (networks/base.py:98-111): the input
// BatchNormalization uses the statistics of the batch it is called on and every call moves the moving averages.
// Same execution scheme as egm_kernels.h (one launch of one workgroup per step; dense ops on fp32-MFMA tiles);
// gradient formulas are those of oracle/egm.py (bgm_disc_step_grads / bgm_gen_step_grads, checked against autograd).
#pragma once
#include "egm_kernels.h"

#define BGM_EGM_BN_MOMENTUM 0.99f

// BaseVariationalNet: theta = [gamma | beta | moving mean | moving var] (q each), then the trunk Dense layers and the
// mean head as one MLP (W0,b0,...,W_mean,b_mean -- LeakyReLU after every trunk layer, linear head), then the variance
// head W_var [h x p], b_var [p].
struct EgmVarNet {
  int q, p, hlast;
  EgmMlp mlp;        // dims [q, trunk..., p], off = 4 q
  int wvar, bvar;    // offsets of the variance head
};

struct BgmEgmArgs {
  EgmVarNet g;
  EgmMlp e;
  EgmDisc dz, dx;          // offsets relative to theta_d (dx follows dz)
  float *theta_g, *m_g, *v_g, *grad_g;   // [g | e]
  float *theta_d, *m_d, *v_d, *grad_d;   // [dz | dx]
  int n_gen, n_disc, B, wmax;
  const float *z, *x;      // [B x q] prior sample, [B x p] data rows of this step
  const float *n1, *n2;    // [B x p] standard-normal noise of the reparameterisations (generator calls 1 and 2)
  float eps_z, eps_x, gamma, alpha;
  EgmAdam adam;
  float *ws;
  float *out;              // disc: [dz_loss, dx_loss, d_loss]   gen: [g_adv, e_adv, l2_z, l2_x, reg, total]
  int apply;
};

struct EgmVarCache {
  float *zhat, *inv;       // [B x q], [q]
  EgmMlpCache mlp;         // in = normalised input zn [B x q]
  float *sraw;             // [B x p]
};

__device__ __forceinline__ void bgm_var_cache(const EgmVarNet &g, int B, float *&p, EgmVarCache &k) {
  auto take = [&](int n) { float *r = p; p += (n + 3) & ~3; return r; };
  k.zhat = take(B * g.q);
  k.inv = take(g.q);
  float *zn = take(B * g.q);
  egm_mlp_cache(g.mlp, B, p, k.mlp, zn);
  k.sraw = take(B * g.p);
}

// g_net(z, training=True): batch statistics (biased variance, eps 1e-3), moving averages updated (momentum 0.99).
// Afterwards mean = egm_act(g.mlp, k.mlp, L, B), s_raw = k.sraw.
__device__ __forceinline__ void bgm_var_fwd(const EgmCtx &c, float *theta, const EgmVarNet &g, const EgmVarCache &k, const float *z,
                                            int B) {
  const int q = g.q;
  float *zn = k.mlp.in;
  for (int f = c.tid; f < q; f += EGM_THREADS) {
    float mu = 0.0f;
    for (int b = 0; b < B; ++b) mu += z[b * q + f];
    mu /= (float)B;
    float var = 0.0f;
    for (int b = 0; b < B; ++b) { const float t = z[b * q + f] - mu; var = fmaf(t, t, var); }
    var /= (float)B;
    const float inv = 1.0f / sqrtf(var + EGM_BN_EPS);
    k.inv[f] = inv;
    const float ga = theta[f], be = theta[q + f];
    for (int b = 0; b < B; ++b) {
      const float zh = (z[b * q + f] - mu) * inv;
      k.zhat[b * q + f] = zh;
      zn[b * q + f] = fmaf(zh, ga, be);
    }
    theta[2 * q + f] = theta[2 * q + f] * BGM_EGM_BN_MOMENTUM + mu * (1.0f - BGM_EGM_BN_MOMENTUM);
    theta[3 * q + f] = theta[3 * q + f] * BGM_EGM_BN_MOMENTUM + var * (1.0f - BGM_EGM_BN_MOMENTUM);
  }
  __syncthreads();
  egm_mlp_fwd(c, theta, g.mlp, k.mlp, B);
  const int L = g.mlp.n_layers;
  egm_fwd(c, egm_act(g.mlp, k.mlp, L - 1, B), g.hlast, theta + g.wvar, theta + g.bvar, k.sraw, g.p, B, g.hlast, g.p, false);
}

// Backward given dLoss/dmean and dLoss/d(s_raw) ([B x p] each, destroyed).  Gradients land at grad + the parameter's
// own offset (moving statistics get none).  dz (may be NULL) receives dLoss/dinput.  dmean, ds, t0, t1 are all
// [B x wmax] scratch rows (the first two hold the [B x p] upstream gradients on entry).
__device__ __forceinline__ void bgm_var_bwd(const EgmCtx &c, const float *theta, float *grad, const EgmVarNet &g, const EgmVarCache &k,
                                            float *dmean, float *ds, float *t0, float *t1, float *dz, int B, bool accumulate) {
  const int L = g.mlp.n_layers, q = g.q;
  const float *hl = egm_act(g.mlp, k.mlp, L - 1, B);
  // variance head: its input gradient joins the mean head's before the LeakyReLU derivative of the last trunk layer
  egm_bwd_w(c, hl, g.hlast, ds, g.p, grad + g.wvar, grad + g.bvar, B, g.hlast, g.p, accumulate);
  egm_bwd_in(c, ds, g.p, theta + g.wvar, t1, g.hlast, B, g.hlast, g.p, false);
  float *cur = dmean, *nxt = t0;
  for (int l = L - 1; l >= 0; --l) {
    const int in = g.mlp.dims[l], out = g.mlp.dims[l + 1];
    const float *W = egm_W(theta, g.mlp, l);
    float *gW = grad + (W - theta);
    egm_bwd_layer(c, egm_act(g.mlp, k.mlp, l, B), cur, W, gW, gW + in * out, nxt, B, in, out, accumulate, l > 0,
                  l == L - 1 ? t1 : nullptr);
    float *t = cur; cur = nxt; nxt = t;
    if (l == L - 1) nxt = ds;    // ds is free now; keeps dmean / t0 / ds rotating without touching t1 again
  }
  // cur = dLoss/d zn.  BatchNorm: d gamma, d beta, and (optionally) the input gradient with its cross-sample terms.
  for (int f = c.tid; f < q; f += EGM_THREADS) {
    float gg = 0.0f, gb = 0.0f;
    for (int b = 0; b < B; ++b) { gg = fmaf(cur[b * q + f], k.zhat[b * q + f], gg); gb += cur[b * q + f]; }
    grad[f] = accumulate ? grad[f] + gg : gg;
    grad[q + f] = accumulate ? grad[q + f] + gb : gb;
    if (!accumulate) { grad[2 * q + f] = 0.0f; grad[3 * q + f] = 0.0f; }
    if (dz) {
      const float ga = theta[f], inv = k.inv[f];
      const float m1 = ga * gb / (float)B, m2 = ga * gg / (float)B;
      for (int b = 0; b < B; ++b) dz[b * q + f] = inv * (ga * cur[b * q + f] - m1 - k.zhat[b * q + f] * m2);
    }
  }
  __syncthreads();
}

// x = noise * sqrt(softplus(s_raw) + 1e-6) + mean   (reparameterize, networks/base.py:113-117); also returns s2
__device__ __forceinline__ void bgm_reparam(const EgmCtx &c, const float *mean, const float *sraw, const float *noise, float *xout,
                                            float *s2out, int n) {
  for (int k = c.tid; k < n; k += EGM_THREADS) {
    const float sr = sraw[k];
    const float s2 = fmaxf(sr, 0.0f) + log1pf(expf(-fabsf(sr))) + 1e-6f;
    s2out[k] = s2;
    xout[k] = fmaf(noise[k], sqrtf(s2), mean[k]);
  }
  __syncthreads();
}

// LSGAN forward + backward of one discriminator on one batch: loss term mean((target - D)^2) * w; accumulates
// dLoss/dtheta into gr and (optionally) returns dLoss/dinput.  Returns the loss term.
__device__ __forceinline__ float bgm_lsgan(const EgmCtx &c, const float *th, float *gr, const EgmDisc &d, const EgmDiscCache &k,
                                           float target, float w, float *dvec, float *da, float *du, float *dx, int B,
                                           bool accumulate, bool want_param_grads) {
  egm_disc_fwd(c, th, d, k, B);
  const float *o = egm_dk_out(d, k, B);
  float part = 0.0f;
  for (int b = c.tid; b < B; b += EGM_THREADS) {
    const float r = target - o[b];
    part = fmaf(r, r, part);
    dvec[b] = -2.0f * w * r / (float)B;
  }
  const float loss = egm_block_sum(c, part) * w / (float)B;
  (void)want_param_grads;
  egm_disc_bwd(c, th, gr, d, k, true, 0.0f, nullptr, da, du, dx, B, accumulate, 1.0f, dvec);
  return loss;
}

// ---------------------------------------------------------------------------------------------
// train_disc_step
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(EGM_THREADS) void bgm_egm_disc_step_kernel(BgmEgmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  EgmCtx c{(int)threadIdx.x, egm_lds};
  const int B = a.B, q = a.g.q, p = a.g.p;
  float *wp = a.ws;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  // fakes: z_ = e(x), x_ = reparam(g(z))   (generator and encoder are fixed in this step; g still moves its BN averages)
  EgmMlpCache ce;
  egm_mlp_cache(a.e, B, wp, ce, const_cast<float *>(a.x));
  egm_mlp_fwd(c, a.theta_g, a.e, ce, B);
  float *z_ = egm_act(a.e, ce, a.e.n_layers, B);
  EgmVarCache kg;
  bgm_var_cache(a.g, B, wp, kg);
  bgm_var_fwd(c, a.theta_g, a.g, kg, a.z, B);
  float *x_ = take(B * p), *s2 = take(B * p);
  bgm_reparam(c, egm_act(a.g.mlp, kg.mlp, a.g.mlp.n_layers, B), kg.sraw, a.n1, x_, s2, B * p);
  float *dvec = take(B), *da = take(B * a.wmax), *du = take(B * a.wmax);
  float *gz = a.grad_d, *gx = a.grad_d + a.dz.n_params;
  const float *tz = a.theta_d, *tx = a.theta_d + a.dz.n_params;
  EgmDiscCache k1;
  float *mark = wp;
  // latent discriminator: real z (target 0.9), fake z_ (target 0.1); each term weighted 1/2
  egm_disc_cache(a.dz, B, wp, k1, const_cast<float *>(a.z));
  float dz_loss = bgm_lsgan(c, tz, gz, a.dz, k1, 0.9f, 0.5f, dvec, da, du, nullptr, B, false, true);
  wp = mark; egm_disc_cache(a.dz, B, wp, k1, z_);
  dz_loss += bgm_lsgan(c, tz, gz, a.dz, k1, 0.1f, 0.5f, dvec, da, du, nullptr, B, true, true);
  // data discriminator
  wp = mark; egm_disc_cache(a.dx, B, wp, k1, const_cast<float *>(a.x));
  float dx_loss = bgm_lsgan(c, tx, gx, a.dx, k1, 0.9f, 0.5f, dvec, da, du, nullptr, B, false, true);
  wp = mark; egm_disc_cache(a.dx, B, wp, k1, x_);
  dx_loss += bgm_lsgan(c, tx, gx, a.dx, k1, 0.1f, 0.5f, dvec, da, du, nullptr, B, true, true);
  float d_loss = dz_loss + dx_loss;
  if (a.gamma != 0.0f) {   // gradient penalties on interpolates (bgm/base.py:203-233)
    float *zh = take(B * q), *xh = take(B * p);
    for (int k = c.tid; k < B * q; k += EGM_THREADS) zh[k] = a.z[k] * a.eps_z + z_[k] * (1.0f - a.eps_z);
    for (int k = c.tid; k < B * p; k += EGM_THREADS) xh[k] = a.x[k] * a.eps_x + x_[k] * (1.0f - a.eps_x);
    __syncthreads();
    float *mark2 = wp;
    egm_disc_cache(a.dz, B, wp, k1, zh);
    egm_disc_fwd(c, tz, a.dz, k1, B);
    const float gpz = egm_disc_gp(c, tz, gz, a.dz, k1, wp, B, a.gamma, a.wmax);
    wp = mark2; egm_disc_cache(a.dx, B, wp, k1, xh);
    egm_disc_fwd(c, tx, a.dx, k1, B);
    const float gpx = egm_disc_gp(c, tx, gx, a.dx, k1, wp, B, a.gamma, a.wmax);
    d_loss += a.gamma * (gpz + gpx);
  }
  __syncthreads();
  if (a.apply) egm_adam(c, a.theta_d, a.m_d, a.v_d, a.grad_d, a.n_disc, a.adam);
  if (c.tid == 0 && a.out) { a.out[0] = dz_loss; a.out[1] = dx_loss; a.out[2] = d_loss; }
}

// ---------------------------------------------------------------------------------------------
// train_gen_step
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(EGM_THREADS) void bgm_egm_gen_step_kernel(BgmEgmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  EgmCtx c{(int)threadIdx.x, egm_lds};
  const int B = a.B, q = a.g.q, p = a.g.p, Lg = a.g.mlp.n_layers, Le = a.e.n_layers;
  float *wp = a.ws;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  // ---- forward
  EgmVarCache g1, g2;
  bgm_var_cache(a.g, B, wp, g1);
  bgm_var_fwd(c, a.theta_g, a.g, g1, a.z, B);                                    // g(z)
  float *x_ = take(B * p), *s21 = take(B * p);
  bgm_reparam(c, egm_act(a.g.mlp, g1.mlp, Lg, B), g1.sraw, a.n1, x_, s21, B * p);
  EgmMlpCache e1, e2;
  egm_mlp_cache(a.e, B, wp, e1, const_cast<float *>(a.x));
  egm_mlp_fwd(c, a.theta_g, a.e, e1, B);                                         // z_ = e(x)
  float *z_ = egm_act(a.e, e1, Le, B);
  egm_mlp_cache(a.e, B, wp, e2, x_);
  egm_mlp_fwd(c, a.theta_g, a.e, e2, B);                                         // z__ = e(x_)
  float *z__ = egm_act(a.e, e2, Le, B);
  bgm_var_cache(a.g, B, wp, g2);
  bgm_var_fwd(c, a.theta_g, a.g, g2, z_, B);                                     // g(z_)
  float *x__ = take(B * p), *s22 = take(B * p);
  bgm_reparam(c, egm_act(a.g.mlp, g2.mlp, Lg, B), g2.sraw, a.n2, x__, s22, B * p);
  // ---- losses that need no discriminator
  float l_x = 0.0f, l_z = 0.0f, reg = 0.0f;
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const float t = a.x[k] - x__[k]; l_x = fmaf(t, t, l_x); reg = fmaf(s21[k], s21[k], reg); }
  for (int k = c.tid; k < B * q; k += EGM_THREADS) { const float t = a.z[k] - z__[k]; l_z = fmaf(t, t, l_z); }
  l_x = egm_block_sum(c, l_x) / (float)(B * p);
  l_z = egm_block_sum(c, l_z) / (float)(B * q);
  reg = egm_block_sum(c, reg) / (float)(B * p);
  // ---- backward
  // dmean / ds also serve as ping-pong buffers of the trunk backward: full scratch rows
  float *dmean = take(B * a.wmax), *ds = take(B * a.wmax), *t0 = take(B * a.wmax), *t1 = take(B * a.wmax);
  float *dzsum = take(B * q), *dx_ = take(B * p), *dvec = take(B), *da = take(B * a.wmax), *du = take(B * a.wmax);
  float *dtmp = take(B * a.wmax), *gscr = take(a.n_disc);
  float *grad_e = a.grad_g;   // the encoder's gradients live at its own offsets inside grad_g
  // x__ branch -> g (call 2, input z_): dLoss/dx__ = 10 * (-2 / (B p)) (x - x__); reparam: dmu = dx__, ds2 = dx__ n2 / (2 sqrt s2)
  for (int k = c.tid; k < B * p; k += EGM_THREADS) {
    const float dxk = 10.0f * (-2.0f / (float)(B * p)) * (a.x[k] - x__[k]);
    dmean[k] = dxk;
    const float sr = g2.sraw[k];
    ds[k] = dxk * a.n2[k] * 0.5f / sqrtf(s22[k]) / (1.0f + expf(-sr));
  }
  __syncthreads();
  bgm_var_bwd(c, a.theta_g, a.grad_g, a.g, g2, dmean, ds, t0, t1, dzsum, B, false);   // dzsum = dLoss/dz_ (so far)
  // z__ branch -> e (call 2, input x_)
  for (int k = c.tid; k < B * q; k += EGM_THREADS) t0[k] = 10.0f * (-2.0f / (float)(B * q)) * (a.z[k] - z__[k]);
  __syncthreads();
  egm_mlp_bwd(c, a.theta_g, grad_e, a.e, e2, t0, t1, dx_, B, false);                   // dx_ = dLoss/dx_ (so far)
  // adversarial terms through the fixed discriminators (their parameter gradients go to scratch)
  const float *tz = a.theta_d, *tx = a.theta_d + a.dz.n_params;
  EgmDiscCache kd;
  float *mark = wp;
  egm_disc_cache(a.dx, B, wp, kd, x_);
  const float g_adv = bgm_lsgan(c, tx, gscr, a.dx, kd, 0.9f, 1.0f, dvec, da, du, dtmp, B, false, false);
  for (int k = c.tid; k < B * p; k += EGM_THREADS) dx_[k] += dtmp[k];
  __syncthreads();
  wp = mark; egm_disc_cache(a.dz, B, wp, kd, z_);
  const float e_adv = bgm_lsgan(c, tz, gscr, a.dz, kd, 0.9f, 1.0f, dvec, da, du, dtmp, B, false, false);
  for (int k = c.tid; k < B * q; k += EGM_THREADS) dzsum[k] += dtmp[k];
  __syncthreads();
  // x_ -> g (call 1, input z): dmu = dx_, ds2 = dx_ n1 / (2 sqrt s2) + alpha 2 s2 / (B p)
  for (int k = c.tid; k < B * p; k += EGM_THREADS) {
    const float d = dx_[k];
    dmean[k] = d;
    const float sr = g1.sraw[k];
    ds[k] = (d * a.n1[k] * 0.5f / sqrtf(s21[k]) + a.alpha * 2.0f * s21[k] / (float)(B * p)) / (1.0f + expf(-sr));
  }
  __syncthreads();
  bgm_var_bwd(c, a.theta_g, a.grad_g, a.g, g1, dmean, ds, t0, t1, nullptr, B, true);
  // z_ -> e (call 1, input x)
  egm_mlp_bwd(c, a.theta_g, grad_e, a.e, e1, dzsum, t1, nullptr, B, true);
  if (a.apply) egm_adam(c, a.theta_g, a.m_g, a.v_g, a.grad_g, a.n_gen, a.adam);
  if (c.tid == 0 && a.out) {
    a.out[0] = g_adv; a.out[1] = e_adv; a.out[2] = l_z; a.out[3] = l_x; a.out[4] = reg;
    a.out[5] = g_adv + e_adv + 10.0f * (l_x + l_z) + a.alpha * reg;
  }
}

// e(x) for n rows (Z initialisation, evaluate(data_z=None)): each workgroup takes blocks of `B` rows through the same
// MLP routine with its own workspace slice.
static __global__ __launch_bounds__(EGM_THREADS) void bgm_egm_encode_kernel(EgmMlp e, const float *theta, const float *x, long long n, float *zout,
                                                                    float *ws, long long ws_per_block, int B) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  EgmCtx c{(int)threadIdx.x, egm_lds};
  const int p = e.dims[0], q = e.dims[e.n_layers];
  for (long long r0 = (long long)blockIdx.x * B; r0 < n; r0 += (long long)gridDim.x * B) {
    const int rows = (int)((n - r0) < B ? (n - r0) : B);
    float *wp = ws + blockIdx.x * ws_per_block;
    EgmMlpCache ce;
    egm_mlp_cache(e, rows, wp, ce, const_cast<float *>(x + r0 * p));
    egm_mlp_fwd(c, theta, e, ce, rows);
    const float *z = egm_act(e, ce, e.n_layers, rows);
    for (int k = c.tid; k < rows * q; k += EGM_THREADS) zout[r0 * q + k] = z[k];
    __syncthreads();
  }
}
