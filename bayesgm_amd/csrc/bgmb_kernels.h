// bgmb_kernels.h -- BGM with the Bayesian generator (use_bnn=True) on gfx950 (BayesianVariationalNet).
//
// replaces (src/bayesgm/models/networks/bnn.py:40-99 BayesianVariationalNet = BatchNormalization that honours `training` +
// DenseFlipout trunk + two sibling DenseFlipout heads, kernel AND bias prior N(0, 0.1^2)) inside
//   BGM.update_g_net                 bgm/base.py:145-164 (KL term :155-157)  -> bgmb_theta_step_kernel
//   BGM.update_latent_variable_sgd   bgm/base.py:167-187                     -> bgmb_z_step_kernel
//   BGM.get_log_posterior            bgm/base.py:665-705 (training=False)    -> bgmb_logpost_kernel
//   BGM.tfp_mcmc_sampler             bgm/base.py:709-830                     -> bgmb_hmc_kernel
//   predict_on_posteriors / generate / evaluate  :511-525, :478-509, :444-476 -> bgmb_decode_kernel
// Arithmetic and noise layout: oracle/bgm_bnn.py.  DenseFlipout perturbs the kernel in every call, also at inference, so
// the HMC target is stochastic: each gradient evaluation is one generator call (one perturbation shared by all rows, signs
// per row keyed by the global row).
//
// Everything here is built from the Flipout building blocks of bnn_kernels.h (two-accumulator fp32-MFMA GEMMs over
// workspace in HBM / L2).  The minibatch steps are one workgroup; the large-batch kernels give each workgroup a tile of
// BGMB_RT rows which it carries through whole HMC transitions (chains are independent; the per-call perturbation is a pure
// function of (seed, stream), so every workgroup regenerates the same one).  This is the first correct version of this
// path, not a tuned one (DESIGN_HISTORY.md section 7).
#pragma once
#include <hip/hip_runtime.h>

#include "bnn_kernels.h"

#define BGMB_RT 64
#define BGMB_MAX_BATCH 4096        // rows of a minibatch step (bgmb_theta_step_kernel, bgmb_z_step_kernel)
#define BGMB_STREAM_PREDICT 0x40000000u
#define BGMB_STREAM_DECODE 0x50000000u

// feature f of oracle/rng.py normals(rows, it, n_feat, tag)
__device__ __forceinline__ float bgmb_normal(uint32_t row, uint32_t it, int f, uint32_t tag, uint32_t k0, uint32_t k1) {
  const int g = f & 3, s = f >> 2;
  const f32x4 e = box_muller4(philox4x32_10(row, it, (uint32_t)(g + 4 * (s >> 2)), tag, k0, k1));
  const int u = s & 3;
  return u == 0 ? e[0] : (u == 1 ? e[1] : (u == 2 ? e[2] : e[3]));
}

// Per-cell Gaussian terms of rows [0, B): o = mean [B x p] | raw variance [B x p]; x [B x p] with NaN = missing.
//   ll[i]      = (x - mean)^2 / (2 s2) + log(s2) / 2           (0 for a missing cell)
//   d[i]       = w * d ll / d mean,  d[B p + i] = w * d ll / d raw
// Returns the thread's partial sum of squared residuals.
__device__ __forceinline__ float bgmb_cells(const BnnCtx &c, const float *o, const float *x, int B, int p, float w, float *d, float *ll) {
  float sq = 0.0f;
  const int Bp = B * p;
  for (int i = c.tid; i < Bp; i += BNN_THREADS) {
    const float xv = x[i], raw = o[Bp + i];
    const bool ok = xv == xv;
    const float r = ok ? xv - o[i] : 0.0f;
    const float s2 = softplus_acc(raw) + BGM_EPS;
    ll[i] = ok ? r * r / (2.0f * s2) + 0.5f * logf(s2) : 0.0f;
    d[i] = -r / s2 * w;
    d[Bp + i] = ok ? (-r * r / (2.0f * s2 * s2) + 0.5f / s2) * sigmoid_f(raw) * w : 0.0f;
    sq = fmaf(r, r, sq);
  }
  return sq;
}

// rowv[b] = sign * (sum_j ll[b p + j] (+ 0.5 |z_b|^2 when z != NULL)): one wave per row, lanes stride over the columns (coalesced), fixed
// reduction order.  No barrier at the end.
__device__ __forceinline__ void bgmb_row_sums(const BnnCtx &c, const float *ll, int B, int p, const float *z, int q, float *rowv,
                                              float sign = 1.0f) {
  const int lane = c.tid & 63, wave = c.tid >> 6;
  for (int b = wave; b < B; b += BNN_THREADS / 64) {
    float s = 0.0f;
    for (int j = lane; j < p; j += 64) s += ll[b * p + j];
    if (z) for (int j = lane; j < q; j += 64) s = fmaf(0.5f * z[b * q + j], z[b * q + j], s);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) rowv[b] = sign * s;
  }
}

struct BgmbArgs {
  BnnNet net;                            // generator: heads = 1, mv = 1, bn_fixed = 0
  float *theta, *m, *v, *grad;
  int B, q, p, wmax;
  float kl_weight;
  float *data_z;                         // [N x q]
  const int *idx;                        // [B]
  const float *x_;                       // panel [N x p]
  uint32_t k0, k1, stream;
  BnnAdam adam;
  int apply;                             // theta step: 1 = Adam inside the kernel, 0 = gradient stays in grad
  float inv_B;                           // 1 / global batch
  float z_lr_t, z_b1, z_b2, z_eps;       // fresh-slot Adam of the batch latents (bgm/base.py:402)
  float *ws;
  float *out;                            // theta: [loss_x + kl_weight KL, loss_mse];  z: [loss_postrior_z]
  // steps spread over the chip (bgmb_api.hip): the call's eps / dW by bgmb_noise_kernel, and for the theta step the parameter-gradient
  // tiles (bgmb_dw_kernel), the KL terms (bgmb_kl_kernel / bgmb_kl_finish_kernel) and Adam (bnn_adam_kernel) behind the step kernel
  int wide;
  float *kl_part;                        // [BNN_KL_PARTS]
};

struct BgmbWs { float *zb, *xb, *d, *ds, *t0, *t1, *ll, *dx; };
__device__ __forceinline__ void bgmb_take(float *&wp, BgmbWs &w, int B, int q, int p, int wmax) {
  auto take = [&](long long n) { float *r = wp; wp += (n + 3) & ~3LL; return r; };
  w.zb = take((long long)B * q); w.xb = take((long long)B * p);
  w.d = take((long long)B * wmax); w.ds = take((long long)B * wmax); w.t0 = take((long long)B * wmax); w.t1 = take((long long)B * wmax);
  w.ll = take((long long)B * p); w.dx = take((long long)B * q);
}
inline size_t bgmb_ws_floats(int B, int q, int p, int wmax) {
  return (size_t)B * (2 * (size_t)q + 2 * (size_t)p + 4 * (size_t)wmax) + 64;
}

// update_g_net (bgm/base.py:145-164)
static __global__ __launch_bounds__(BNN_THREADS) void bgmb_theta_step_kernel(BgmbArgs a) {
  __shared__ float red[32];
  __shared__ float rowv[BGMB_MAX_BATCH];
  BnnCtx c{(int)threadIdx.x, red};
  const BnnNet &n = a.net;
  const int B = a.B, p = a.p, q = a.q;
  float *wp = a.ws;
  BgmbWs w;
  bgmb_take(wp, w, B, q, p, a.wmax);
  for (int i = c.tid; i < B * q; i += BNN_THREADS) w.zb[i] = a.data_z[(long long)a.idx[i / q] * q + i % q];
  for (int i = c.tid; i < B * p; i += BNN_THREADS) w.xb[i] = a.x_[(long long)a.idx[i / p] * p + i % p];
  __syncthreads();
  BnnCache k;
  bnn_cache(n, B, wp, k, w.zb);
  float *G = nullptr, *GS = nullptr, *d = w.d, *ds = w.ds;
  if (a.wide) {      // every layer's upstream gradient stays (in a second call cache) for bgmb_dw_kernel
    BnnCache k2;
    bnn_cache(n, B, wp, k2, nullptr);
    G = k2.H; GS = k2.HS;
    d = G + (long long)B * n.hoff[n.n_layers - 1]; ds = GS + (long long)B * n.hoff[n.n_layers - 1];
  }
  const float *o = bnn_fwd(c, a.theta, n, k, B, a.k0, a.k1, a.stream, 0u, a.wide != 0);
  float sq = bgmb_cells(c, o, w.xb, B, p, a.inv_B, d, w.ll);
  sq = bnn_block_sum(c, sq);
  bgmb_row_sums(c, w.ll, B, p, nullptr, 0, rowv);
  __syncthreads();
  float loss = 0.0f;
  for (int b = 0; b < B; ++b) loss += rowv[b];
  bnn_bwd(c, a.theta, a.grad, n, k, d, ds, w.t0, w.t1, nullptr, B, true, false, G, GS);
  const float klv = a.wide ? 0.0f : bnn_kl(c, a.theta, a.grad, n, a.kl_weight);
  __syncthreads();
  bnn_bn_move(c, a.theta, n, k);
  __syncthreads();
  if (a.apply && !a.wide) bnn_adam(c, a.theta + n.off, a.m + n.off, a.v + n.off, a.grad + n.off, n.n_params, a.adam);
  if (c.tid == 0 && a.out) { a.out[0] = loss * a.inv_B + a.kl_weight * klv; a.out[1] = sq / (float)(B * p); }
}

// ---- the elementwise parts and the gradient tiles of a step, over the chip (the step kernels are one workgroup) ----------------------
// eps and dW = sigma * eps of the step's call, written into the call cache of the step kernel (same pointer arithmetic); grid (parts)
static __global__ __launch_bounds__(BNN_THREADS) void bgmb_noise_kernel(BgmbArgs a) {
  const BnnNet &n = a.net;
  float *wp = a.ws;
  BgmbWs w;
  bgmb_take(wp, w, a.B, a.q, a.p, a.wmax);
  BnnCache k;
  bnn_cache(n, a.B, wp, k, w.zb);
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.lin[l] * n.lout[l];
    const float *rho = a.theta + n.woff[l] + cnt;
    float *e = k.eps + n.eoff[l], *d = k.dW + n.eoff[l];
    for (int i = blockIdx.x * BNN_THREADS + threadIdx.x; i < (cnt + 3) >> 2; i += gridDim.x * BNN_THREADS) {
      const f32x4 z = box_muller4(philox4x32_10((uint32_t)i, (uint32_t)l | ((uint32_t)n.net_id << 16), a.stream, BNN_TAG_EPS, a.k0, a.k1));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = 4 * i + u;
        if (idx < cnt) { e[idx] = z[u]; d[idx] = (BNN_SCALE_EPS + softplus_acc(rho[idx])) * z[u]; }
      }
    }
  }
}
// the parameter-gradient tiles of all layers (trunk and both heads) behind bgmb_theta_step_kernel; grid (parts)
static __global__ __launch_bounds__(BNN_THREADS) void bgmb_dw_kernel(BgmbArgs a) {
  __shared__ float red[32];
  BnnCtx c{(int)threadIdx.x, red};
  const BnnNet &n = a.net;
  float *wp = a.ws;
  BgmbWs w;
  bgmb_take(wp, w, a.B, a.q, a.p, a.wmax);
  BnnCache k, k2;
  bnn_cache(n, a.B, wp, k, w.zb);
  bnn_cache(n, a.B, wp, k2, nullptr);
  for (int l = 0; l < n.n_layers; ++l)
    bnn_bwd_params(c, a.theta, a.grad, n, k, l, k2.H + (long long)a.B * n.hoff[l + 1], k2.HS + (long long)a.B * n.hoff[l + 1], a.B, false,
                   (int)blockIdx.x, (int)gridDim.x);
}
// grad += w * dKL/dtheta as bnn_kl; the workgroup's share of sum(net.losses) -> kl_part.  grid (BNN_KL_PARTS)
static __global__ __launch_bounds__(256) void bgmb_kl_kernel(BgmbArgs a) {
  __shared__ float red[4];
  const BnnNet &n = a.net;
  const float iv = n.prior_iv, ls = n.prior_logs, w = a.kl_weight;
  float acc = 0.0f;
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.lin[l] * n.lout[l];
    const float *loc = a.theta + n.woff[l], *rho = loc + cnt;
    float *gloc = a.grad + n.woff[l], *grho = gloc + cnt;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256) {
      const float sg = BNN_SCALE_EPS + softplus_acc(rho[i]), mu = loc[i];
      acc += -logf(sg) + 0.5f * (sg * sg + mu * mu) * iv - 0.5f + ls;
      gloc[i] += w * mu * iv;
      grho[i] += w * (-1.0f / sg + sg * iv) * sigmoid_f(rho[i]);
    }
    if (n.bias_prior) {
      const float *b = rho + cnt;
      float *gb = grho + cnt;
      for (int i = blockIdx.x * 256 + threadIdx.x; i < n.lout[l]; i += gridDim.x * 256) {
        acc += 0.5f * b[i] * b[i] * iv + ls + 0.9189385332046727f;
        gb[i] += w * b[i] * iv;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.kl_part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
static __global__ void bgmb_kl_finish_kernel(BgmbArgs a) {
  if (threadIdx.x != 0 || !a.out) return;
  float t = 0.0f;
  for (int i = 0; i < BNN_KL_PARTS; ++i) t += a.kl_part[i];
  a.out[0] += a.kl_weight * t;
}

// update_latent_variable_sgd (bgm/base.py:167-187) + the fresh-slot Adam step on the batch rows (:402)
static __global__ __launch_bounds__(BNN_THREADS) void bgmb_z_step_kernel(BgmbArgs a) {
  __shared__ float red[32];
  __shared__ float rowv[BGMB_MAX_BATCH];
  BnnCtx c{(int)threadIdx.x, red};
  const BnnNet &n = a.net;
  const int B = a.B, p = a.p, q = a.q;
  float *wp = a.ws;
  BgmbWs w;
  bgmb_take(wp, w, B, q, p, a.wmax);
  for (int i = c.tid; i < B * q; i += BNN_THREADS) w.zb[i] = a.data_z[(long long)a.idx[i / q] * q + i % q];
  for (int i = c.tid; i < B * p; i += BNN_THREADS) w.xb[i] = a.x_[(long long)a.idx[i / p] * p + i % p];
  __syncthreads();
  BnnCache k;
  bnn_cache(n, B, wp, k, w.zb);
  const float *o = bnn_fwd(c, a.theta, n, k, B, a.k0, a.k1, a.stream, 0u, a.wide != 0);
  bgmb_cells(c, o, w.xb, B, p, a.inv_B, w.d, w.ll);
  __syncthreads();
  bgmb_row_sums(c, w.ll, B, p, w.zb, q, rowv);
  __syncthreads();
  float loss = 0.0f;
  for (int b = 0; b < B; ++b) loss += rowv[b];
  bnn_bwd(c, a.theta, a.grad, n, k, w.d, w.ds, w.t0, w.t1, w.dx, B, false, false);
  for (int i = c.tid; i < B * q; i += BNN_THREADS) {
    const float z = w.zb[i], g = w.dx[i] + z * a.inv_B;
    const float m = (1.0f - a.z_b1) * g, v = (1.0f - a.z_b2) * g * g;
    a.data_z[(long long)a.idx[i / q] * q + i % q] = z - a.z_lr_t * m / (sqrtf(v) + a.z_eps);
  }
  bnn_bn_move(c, a.theta, n, k);
  if (c.tid == 0 && a.out) a.out[0] = loss * a.inv_B;
}

// ---------------------------------------------------------------------------------------------
// large batches, training = False: row tiles of BGMB_RT rows per workgroup
// ---------------------------------------------------------------------------------------------
struct BgmbBigArgs {
  BnnNet net;                  // generator with bn_fixed = 2 (moving statistics)
  const float *theta;
  int q, p, wmax;
  const float *x;              // [n x p], NaN = missing
  long long n, row_base;
  float *state, *logp, *grad;  // HMC: [n x q], [n], [n x q] in/out;  logpost: z in (state), outputs logp / grad (grad may be NULL)
  int rt;                      // rows per workgroup tile (<= BGMB_RT; smaller when few rows would leave CUs idle)
  long long n_tiles;           // ceil(n / rt); a workgroup walks tiles blockIdx.x, + gridDim.x, ... with ONE workspace slice
  int init, it_begin, n_iters, burn_in, n_leapfrog;
  int frozen;                  // 1: every gradient evaluation of the HMC run reuses generator call 0 (deterministic target)
  const float *step;
  uint32_t k0, k1, stream;
  double *acc_prob_sum;
  uint32_t *acc_count;
  float *draws;                // [n_keep x n x q]
  const float *dw;             // perturbations of the launch's generator calls (bgmb_noise_kernel), slot-major
  long long dw_stride;
  float *ws;
  long long ws_stride;
};

// sigma * eps of whole generator calls, produced ONCE per call for all workgroups (slot s of the launch -> stream[s]):
// dw[s * stride + eoff[l] + idx] as in a call cache.  grid = (element blocks, slots).
struct BgmbNoiseArgs {
  BnnNet net;
  const float *theta;
  float *dw;
  long long stride;
  uint32_t k0, k1;
  uint32_t stream0, stream_step;   // slot s < n_seq: stream0 + s * stream_step
  int n_seq;
  uint32_t stream_extra;           // slot n_seq (if launched): this stream (the bootstrap evaluation)
};
static __global__ __launch_bounds__(256) void bgmb_noise_kernel(BgmbNoiseArgs a) {
  const int slot = blockIdx.y;
  const uint32_t stream = slot < a.n_seq ? a.stream0 + (uint32_t)slot * a.stream_step : a.stream_extra;
  float *dw = a.dw + (long long)slot * a.stride;
  const BnnNet &n = a.net;
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.lin[l] * n.lout[l];
    const float *rho = a.theta + n.woff[l] + cnt;
    float *d = dw + n.eoff[l];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (cnt + 3) >> 2; i += gridDim.x * blockDim.x) {
      const f32x4 z = box_muller4(philox4x32_10((uint32_t)i, (uint32_t)l | ((uint32_t)n.net_id << 16), stream, BNN_TAG_EPS, a.k0, a.k1));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = 4 * i + u;
        if (idx < cnt) d[idx] = (BNN_SCALE_EPS + softplus_acc(rho[idx])) * z[u];
      }
    }
  }
}

struct BgmbTile { float *z, *gr, *zc, *pc, *grc, *mom, *xb, *d, *ds, *t0, *t1, *ll; float *cache; };
__device__ __forceinline__ void bgmb_tile_take(float *wp, BgmbTile &t, int q, int p, int wmax) {
  auto take = [&](long long n) { float *r = wp; wp += (n + 3) & ~3LL; return r; };
  const long long B = BGMB_RT;
  t.z = take(B * q); t.gr = take(B * q); t.zc = take(B * q); t.pc = take(B * q); t.grc = take(B * q); t.mom = take(B * q);
  t.xb = take(B * p);
  t.d = take(B * wmax); t.ds = take(B * wmax); t.t0 = take(B * wmax); t.t1 = take(B * wmax);
  t.ll = take(B * p);
  t.cache = wp;
}
inline size_t bgmb_tile_floats(const BnnNet &n, int q, int p, int wmax) {
  return (size_t)BGMB_RT * (6 * (size_t)q + 2 * (size_t)p + 4 * (size_t)wmax) + 96 + bnn_cache_floats(n, BGMB_RT);
}

// log p(z | x_obs) + const and its gradient for the B rows of a tile, ONE generator call (`stream`); lp -> lpv [B] (LDS),
// gradient -> gr [B x q].  Ends with a barrier.
__device__ __forceinline__ void bgmb_lpg(const BnnCtx &c, const BgmbBigArgs &a, const BgmbTile &t, const float *zin, int B,
                                         uint32_t stream, int slot, uint32_t row0, float *lpv, float *gr) {
  const BnnNet &n = a.net;
  float *wp = t.cache;
  BnnCache k;
  bnn_cache(n, B, wp, k, zin);
  k.dW = const_cast<float *>(a.dw + (long long)slot * a.dw_stride);      // shared by all workgroups; only the signs are per row
  const float *o = bnn_fwd(c, a.theta, n, k, B, a.k0, a.k1, stream, row0, true);
  bgmb_cells(c, o, t.xb, B, a.p, -1.0f, t.d, t.ll);
  __syncthreads();
  bgmb_row_sums(c, t.ll, B, a.p, zin, a.q, lpv, -1.0f);
  if (gr) {
    bnn_bwd(c, a.theta, nullptr, n, k, t.d, t.ds, t.t0, t.t1, gr, B, false, false);
    for (int i = c.tid; i < B * a.q; i += BNN_THREADS) gr[i] -= zin[i];
  }
  __syncthreads();
}

// get_log_posterior for given z (state): logp [n], grad [n x q] (optional)
static __global__ __launch_bounds__(BNN_THREADS) void bgmb_logpost_kernel(BgmbBigArgs a) {
  __shared__ float red[32];
  __shared__ float lpv[BGMB_RT];
  BnnCtx c{(int)threadIdx.x, red};
  const int q = a.q, p = a.p;
  BgmbTile t;
  bgmb_tile_take(a.ws + (long long)blockIdx.x * a.ws_stride, t, q, p, a.wmax);
  for (long long tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const long long r0 = tile * a.rt;
    const int B = (int)min((long long)a.rt, a.n - r0);
    for (int i = c.tid; i < B * q; i += BNN_THREADS) t.z[i] = a.state[r0 * q + i];
    for (int i = c.tid; i < B * p; i += BNN_THREADS) t.xb[i] = a.x[r0 * p + i];
    __syncthreads();
    bgmb_lpg(c, a, t, t.z, B, a.stream, 0, (uint32_t)(a.row_base + r0), lpv, a.grad ? t.gr : nullptr);
    for (int b = c.tid; b < B; b += BNN_THREADS) a.logp[r0 + b] = lpv[b];
    if (a.grad) for (int i = c.tid; i < B * q; i += BNN_THREADS) a.grad[r0 * q + i] = t.gr[i];
    __syncthreads();
  }
}

// HMC transitions [it_begin, it_begin + n_iters) of the rows of a tile (oracle/bgm_bnn.py hmc_sampler)
static __global__ __launch_bounds__(BNN_THREADS) void bgmb_hmc_kernel(BgmbBigArgs a) {
  __shared__ float red[32];
  __shared__ float lpv[BGMB_RT], lpc[BGMB_RT], ke0[BGMB_RT];
  __shared__ int accv[BGMB_RT];
  BnnCtx c{(int)threadIdx.x, red};
  const int q = a.q, p = a.p, L = a.n_leapfrog;
  BgmbTile t;
  bgmb_tile_take(a.ws + (long long)blockIdx.x * a.ws_stride, t, q, p, a.wmax);
  for (long long tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
  const long long r0 = tile * a.rt;
  const int B = (int)min((long long)a.rt, a.n - r0);
  const uint32_t row0 = (uint32_t)(a.row_base + r0);
  for (int i = c.tid; i < B * p; i += BNN_THREADS) t.xb[i] = a.x[r0 * p + i];
  if (a.init) {
    for (int i = c.tid; i < B * q; i += BNN_THREADS) t.z[i] = bgmb_normal(row0 + (uint32_t)(i / q), 0u, i % q, TAG_INIT, a.k0, a.k1);
    __syncthreads();
    bgmb_lpg(c, a, t, t.z, B, 0u, a.frozen ? 0 : a.n_iters * L, row0, lpv, t.gr);
  } else {
    for (int i = c.tid; i < B * q; i += BNN_THREADS) { t.z[i] = a.state[r0 * q + i]; t.gr[i] = a.grad[r0 * q + i]; }
    for (int b = c.tid; b < B; b += BNN_THREADS) lpv[b] = a.logp[r0 + b];
    __syncthreads();
  }
  for (int it = a.it_begin; it < a.it_begin + a.n_iters; ++it) {
    const float e = *a.step;
    for (int i = c.tid; i < B * q; i += BNN_THREADS) {
      const float mo = bgmb_normal(row0 + (uint32_t)(i / q), (uint32_t)it, i % q, TAG_MOM, a.k0, a.k1);
      t.mom[i] = mo;
      t.zc[i] = t.z[i];
      t.pc[i] = mo + 0.5f * e * t.gr[i];
    }
    __syncthreads();
    for (int b = c.tid; b < B; b += BNN_THREADS) {
      float s = 0.0f;
      for (int j = 0; j < q; ++j) s = fmaf(t.mom[b * q + j], t.mom[b * q + j], s);
      ke0[b] = s;
    }
    for (int l = 0; l < L; ++l) {
      for (int i = c.tid; i < B * q; i += BNN_THREADS) t.zc[i] = fmaf(e, t.pc[i], t.zc[i]);
      __syncthreads();
      bgmb_lpg(c, a, t, t.zc, B, a.frozen ? 0u : 1u + (uint32_t)it * (uint32_t)L + (uint32_t)l, a.frozen ? 0 : (it - a.it_begin) * L + l, row0,
               lpc, t.grc);
      const float h = (l < L - 1) ? e : 0.5f * e;
      for (int i = c.tid; i < B * q; i += BNN_THREADS) t.pc[i] = fmaf(h, t.grc[i], t.pc[i]);
      __syncthreads();
    }
    float pa = 0.0f;
    int na = 0;
    for (int b = c.tid; b < B; b += BNN_THREADS) {
      float s = 0.0f;
      for (int j = 0; j < q; ++j) s = fmaf(t.pc[b * q + j], t.pc[b * q + j], s);
      float lr = -((-lpc[b] + 0.5f * s) - (-lpv[b] + 0.5f * ke0[b]));
      lr = (lr == lr && fabsf(lr) != INFINITY) ? lr : -INFINITY;
      const uint4 w4 = philox4x32_10(row0 + (uint32_t)b, (uint32_t)it >> 2, 0u, TAG_HACC, a.k0, a.k1);
      const uint32_t w_ = (it & 2) ? ((it & 1) ? w4.w : w4.z) : ((it & 1) ? w4.y : w4.x);
      const bool acc = logf(u01_open(w_)) < lr;
      accv[b] = acc ? 1 : 0;
      if (acc) lpv[b] = lpc[b];
      pa += expf(fminf(lr, 0.0f));
      na += acc ? 1 : 0;
    }
    __syncthreads();
    for (int i = c.tid; i < B * q; i += BNN_THREADS)
      if (accv[i / q]) { t.z[i] = t.zc[i]; t.gr[i] = t.grc[i]; }
    pa = bnn_block_sum(c, pa);
    const float nf = bnn_block_sum(c, (float)na);
    if (c.tid == 0) {
      if (a.acc_prob_sum) atomicAdd(a.acc_prob_sum + it, (double)pa);
      if (a.acc_count) atomicAdd(a.acc_count + it, (uint32_t)(nf + 0.5f));
    }
    __syncthreads();
    if (a.draws && it >= a.burn_in)
      for (int i = c.tid; i < B * q; i += BNN_THREADS) a.draws[((long long)(it - a.burn_in) * a.n + r0) * q + i] = t.z[i];
  }
  __syncthreads();
  for (int i = c.tid; i < B * q; i += BNN_THREADS) { a.state[r0 * q + i] = t.z[i]; a.grad[r0 * q + i] = t.gr[i]; }
  for (int b = c.tid; b < B; b += BNN_THREADS) a.logp[r0 + b] = lpv[b];
  __syncthreads();
  }   // tiles of this workgroup
}

// g_net(z, training=False) over the flattened rows [n_draws x n] (ONE call: one perturbation; the signs of draw d, row r are
// keyed by d * sign_stride + sign_off + r), then x = mean + sqrt(var) * noise (tag 6 keyed by the GLOBAL row and
// burn_in + draw, as the deterministic path).  A workgroup tile never spans two draws.
struct BgmbDecodeArgs {
  BnnNet net;
  const float *theta;
  int q, p;
  const float *draws;          // [n_draws x n x q]
  long long n, row_base;
  int n_draws, burn_in;
  uint32_t k0, k1, stream;     // Flipout noise key / stream of the call
  uint32_t sign_stride, sign_off;
  uint32_t x0, x1;             // key of the x-noise
  const int *slot; int k_slots;
  float *cells, *full, *var_full;
  int add_noise;
  const float *dw;             // the call's perturbation (bgmb_noise_kernel, one slot)
  float *ws;
  long long ws_stride;
};
static __global__ __launch_bounds__(BNN_THREADS) void bgmb_decode_kernel(BgmbDecodeArgs a) {
  __shared__ float red[32];
  BnnCtx c{(int)threadIdx.x, red};
  const int q = a.q, p = a.p;
  const long long tiles_per_draw = (a.n + BGMB_RT - 1) / BGMB_RT, tiles = tiles_per_draw * a.n_draws;
  float *base = a.ws + (long long)blockIdx.x * a.ws_stride;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long d = t / tiles_per_draw, r0 = (t - d * tiles_per_draw) * BGMB_RT;
    const int B = (int)min((long long)BGMB_RT, a.n - r0);
    const long long f0 = d * a.n + r0;
    float *wp = base;
    BnnCache k;
    bnn_cache(a.net, B, wp, k, a.draws + f0 * q);
    k.dW = const_cast<float *>(a.dw);
    const float *o = bnn_fwd(c, a.theta, a.net, k, B, a.k0, a.k1, a.stream, (uint32_t)d * a.sign_stride + a.sign_off + (uint32_t)r0, true);
    for (int i = c.tid; i < B * p; i += BNN_THREADS) {
      const long long row = r0 + i / p, f = d * a.n + row;
      const int col = i % p;
      const float s2 = softplus_acc(o[B * p + i]) + BGM_EPS;
      float xp = o[i];
      if (a.add_noise) {
        const f32x4 e = box_muller4(philox4x32_10((uint32_t)(a.row_base + row), (uint32_t)(a.burn_in + d), (uint32_t)(col >> 2), TAG_XNOISE, a.x0, a.x1));
        const int u = col & 3;
        xp = fmaf(sqrtf(s2), u == 0 ? e[0] : (u == 1 ? e[1] : (u == 2 ? e[2] : e[3])), xp);
      }
      if (a.full) a.full[f * p + col] = xp;
      if (a.var_full) a.var_full[f * p + col] = s2;
      if (a.cells) {
        const int sl = a.slot[row * (long long)p + col];
        if (sl >= 0) a.cells[(row * (long long)a.k_slots + sl) * a.n_draws + d] = xp;
      }
    }
    __syncthreads();
  }
}
