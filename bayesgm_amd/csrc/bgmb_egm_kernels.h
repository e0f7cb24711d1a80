// bgmb_egm_kernels.h -- EGM warm start of BGM with the Bayesian generator (use_bnn=True) on gfx950.
//
// replaces (src/bayesgm/models/bgm/base.py with g_net = BayesianVariationalNet, networks/bnn.py:40-99):
//   train_disc_step :190-244  -> bgmb_egm_disc_step_kernel
//   train_gen_step  :246-289  -> bgmb_egm_gen_step_kernel   (two Flipout generator calls; the KL term is commented out
//                                                            in the reference, :280-283, and is not added here)
// e_net, dz_net, dx_net stay deterministic (bgm/base.py:74-79): their routines are those of egm_kernels.h /
// bgm_egm_kernels.h; the generator calls are the Flipout routines of bnn_kernels.h (training-mode BatchNorm, moving
// averages updated by every call).  Gradients follow oracle/bgm_bnn.py (egm_disc_step_grads, egm_gen_step_grads).
#pragma once
#include "bnn_kernels.h"
#include "bgm_egm_kernels.h"

static_assert(BNN_THREADS == EGM_THREADS, "the Flipout and discriminator routines share one workgroup");

struct BgmbEgmArgs {
  BnnNet g;
  EgmMlp e;                // off = g.n_params
  EgmDisc dz, dx;          // offsets relative to theta_d (dx follows dz)
  float *theta_g, *m_g, *v_g, *grad_g;   // [g | e]
  float *theta_d, *m_d, *v_d, *grad_d;   // [dz | dx]
  int n_gen, n_disc, B, q, p, wmax;
  const float *z, *x;      // [B x q] prior sample, [B x p] data rows of this step
  const float *n1, *n2;    // [B x p] standard normals of the two reparameterisations
  float eps_z, eps_x, gamma, alpha;
  uint32_t k0, k1, stream; // Flipout noise: generator call c of the step uses stream + c
  EgmAdam adam;
  float *ws;
  float *out;
  int apply;
};

static __global__ __launch_bounds__(EGM_THREADS) void bgmb_egm_disc_step_kernel(BgmbEgmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  EgmCtx c{(int)threadIdx.x, egm_lds};
  BnnCtx cb{(int)threadIdx.x, egm_lds};
  const int B = a.B, q = a.q, p = a.p;
  float *wp = a.ws;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  EgmMlpCache ce;
  egm_mlp_cache(a.e, B, wp, ce, const_cast<float *>(a.x));
  egm_mlp_fwd(c, a.theta_g, a.e, ce, B);
  float *z_ = egm_act(a.e, ce, a.e.n_layers, B);
  BnnCache kg;
  bnn_cache(a.g, B, wp, kg, a.z);
  const float *o = bnn_fwd(cb, a.theta_g, a.g, kg, B, a.k0, a.k1, a.stream);
  bnn_bn_move(cb, a.theta_g, a.g, kg);
  float *x_ = take(B * p), *s2 = take(B * p);
  bgm_reparam(c, o, o + B * p, a.n1, x_, s2, B * p);
  float *dvec = take(B), *da = take(B * a.wmax), *du = take(B * a.wmax);
  float *gz = a.grad_d, *gx = a.grad_d + a.dz.n_params;
  const float *tz = a.theta_d, *tx = a.theta_d + a.dz.n_params;
  EgmDiscCache k1;
  float *mark = wp;
  egm_disc_cache(a.dz, B, wp, k1, const_cast<float *>(a.z));
  float dz_loss = bgm_lsgan(c, tz, gz, a.dz, k1, 0.9f, 0.5f, dvec, da, du, nullptr, B, false, true);
  wp = mark; egm_disc_cache(a.dz, B, wp, k1, z_);
  dz_loss += bgm_lsgan(c, tz, gz, a.dz, k1, 0.1f, 0.5f, dvec, da, du, nullptr, B, true, true);
  wp = mark; egm_disc_cache(a.dx, B, wp, k1, const_cast<float *>(a.x));
  float dx_loss = bgm_lsgan(c, tx, gx, a.dx, k1, 0.9f, 0.5f, dvec, da, du, nullptr, B, false, true);
  wp = mark; egm_disc_cache(a.dx, B, wp, k1, x_);
  dx_loss += bgm_lsgan(c, tx, gx, a.dx, k1, 0.1f, 0.5f, dvec, da, du, nullptr, B, true, true);
  float d_loss = dz_loss + dx_loss;
  if (a.gamma != 0.0f) {
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

static __global__ __launch_bounds__(EGM_THREADS) void bgmb_egm_gen_step_kernel(BgmbEgmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  EgmCtx c{(int)threadIdx.x, egm_lds};
  BnnCtx cb{(int)threadIdx.x, egm_lds};
  const int B = a.B, q = a.q, p = a.p, Le = a.e.n_layers;
  float *wp = a.ws;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  // ---- forward
  BnnCache g1, g2;
  bnn_cache(a.g, B, wp, g1, a.z);
  const float *o1 = bnn_fwd(cb, a.theta_g, a.g, g1, B, a.k0, a.k1, a.stream);          // g(z)
  bnn_bn_move(cb, a.theta_g, a.g, g1);
  float *x_ = take(B * p), *s21 = take(B * p);
  bgm_reparam(c, o1, o1 + B * p, a.n1, x_, s21, B * p);
  EgmMlpCache e1, e2;
  egm_mlp_cache(a.e, B, wp, e1, const_cast<float *>(a.x));
  egm_mlp_fwd(c, a.theta_g, a.e, e1, B);                                               // z_ = e(x)
  float *z_ = egm_act(a.e, e1, Le, B);
  egm_mlp_cache(a.e, B, wp, e2, x_);
  egm_mlp_fwd(c, a.theta_g, a.e, e2, B);                                               // z__ = e(x_)
  float *z__ = egm_act(a.e, e2, Le, B);
  bnn_cache(a.g, B, wp, g2, z_);
  const float *o2 = bnn_fwd(cb, a.theta_g, a.g, g2, B, a.k0, a.k1, a.stream + 1u);     // g(z_)
  bnn_bn_move(cb, a.theta_g, a.g, g2);
  float *x__ = take(B * p), *s22 = take(B * p);
  bgm_reparam(c, o2, o2 + B * p, a.n2, x__, s22, B * p);
  float l_x = 0.0f, l_z = 0.0f, reg = 0.0f;
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const float t = a.x[k] - x__[k]; l_x = fmaf(t, t, l_x); reg = fmaf(s21[k], s21[k], reg); }
  for (int k = c.tid; k < B * q; k += EGM_THREADS) { const float t = a.z[k] - z__[k]; l_z = fmaf(t, t, l_z); }
  l_x = egm_block_sum(c, l_x) / (float)(B * p);
  l_z = egm_block_sum(c, l_z) / (float)(B * q);
  reg = egm_block_sum(c, reg) / (float)(B * p);
  // ---- backward (d holds dLoss/dmean [B x p] followed by dLoss/d raw variance [B x p])
  float *d = take(B * a.wmax), *ds = take(B * a.wmax), *t0 = take(B * a.wmax), *t1 = take(B * a.wmax);
  float *dzsum = take(B * q), *dx_ = take(B * p), *dvec = take(B), *da = take(B * a.wmax), *du = take(B * a.wmax);
  float *dtmp = take(B * a.wmax), *gscr = take(a.n_disc);
  float *grad_e = a.grad_g;
  const float *sr2 = o2 + B * p, *sr1 = o1 + B * p;
  for (int k = c.tid; k < B * p; k += EGM_THREADS) {
    const float dxk = 10.0f * (-2.0f / (float)(B * p)) * (a.x[k] - x__[k]);
    d[k] = dxk;
    d[B * p + k] = dxk * a.n2[k] * 0.5f / sqrtf(s22[k]) / (1.0f + expf(-sr2[k]));
  }
  __syncthreads();
  bnn_bwd(cb, a.theta_g, a.grad_g, a.g, g2, d, ds, t0, t1, dzsum, B, true, false);     // dzsum = dLoss/dz_ (so far)
  for (int k = c.tid; k < B * q; k += EGM_THREADS) t0[k] = 10.0f * (-2.0f / (float)(B * q)) * (a.z[k] - z__[k]);
  __syncthreads();
  egm_mlp_bwd(c, a.theta_g, grad_e, a.e, e2, t0, t1, dx_, B, false);                   // dx_ = dLoss/dx_ (so far)
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
  for (int k = c.tid; k < B * p; k += EGM_THREADS) {
    const float dd = dx_[k];
    d[k] = dd;
    d[B * p + k] = (dd * a.n1[k] * 0.5f / sqrtf(s21[k]) + a.alpha * 2.0f * s21[k] / (float)(B * p)) / (1.0f + expf(-sr1[k]));
  }
  __syncthreads();
  bnn_bwd(cb, a.theta_g, a.grad_g, a.g, g1, d, ds, t0, t1, nullptr, B, true, true);
  egm_mlp_bwd(c, a.theta_g, grad_e, a.e, e1, dzsum, t1, nullptr, B, true);
  if (a.apply) egm_adam(c, a.theta_g, a.m_g, a.v_g, a.grad_g, a.n_gen, a.adam);
  if (c.tid == 0 && a.out) {
    a.out[0] = g_adv; a.out[1] = e_adv; a.out[2] = l_z; a.out[3] = l_x; a.out[4] = reg;
    a.out[5] = g_adv + e_adv + 10.0f * (l_x + l_z) + a.alpha * reg;
  }
}
