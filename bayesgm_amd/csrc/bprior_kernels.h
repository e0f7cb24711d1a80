// bprior_kernels.h -- the conditional-prior network of IdentifiableCausalBGM with use_bnn=True: a BayesianFullyConnectedNet
// (n_segments -> prior_units -> q + 1) on the one-hot segment of every row.
//
// replaces (src/bayesgm/models/causalbgm/identifiable.py, networks/bnn.py:4-38):
//   prior_net = BayesianFullyConnectedNet(...)                                                               :66-67
//   update_latent_variable_sgd: conditional-prior term + kl_weight * sum(prior_net.losses), joint step on the batch
//   latents (fresh Adam slots) and on the prior net (prior_optimizer)                                            :195-226
//   prior_net(data_u) inside get_log_posterior: one noisy call per log-posterior evaluation                       :541-551
// Semantics and noise streams as the other Bayesian nets (oracle/bnn.py): input BatchNormalization (statistics of the batch at hand, or
// the build's fixed mean 0 / variance 1), DenseFlipout layers  y = x loc + ((x s_in)(sigma eps)) s_out + b,  sigma = eps32 + softplus(rho),
// ONE eps per call and layer (Philox TAG_EPS, net id 4), per-row sign words (TAG_SIGN), KL(N(loc, sigma^2) || N(0, 1)) per kernel.
// Parameter layout: gamma [k], beta [k], then per layer loc [in x out], rho [in x out], bias [out].
// A minibatch is 32 rows and the net ~1.5 k parameters: one workgroup, everything in LDS, every parameter's gradient formed by the thread
// that owns it (fixed summation order) and consumed by its Adam update on the spot.  The sampling-side kernel evaluates the net for
// the rows of a block in chunks of 64 rows per pass.
#pragma once
#include <hip/hip_runtime.h>
#include "bgm_device.h"

#include "bprior_types.h"

__device__ __forceinline__ float bprior_softplus(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float bprior_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float bprior_sign(const unsigned *w, int c) { return ((w[c >> 5] >> (c & 31)) & 1u) ? -1.0f : 1.0f; }

// sigma * eps (dw) and eps (ep, optional) of every layer of one call -> LDS, layer l at offset sum of the earlier in x out
__device__ __forceinline__ void bprior_noise(const BPriorNet &n, const float *theta, float *dw, float *ep, unsigned stream, unsigned k0, unsigned k1) {
  int off = 0;
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.dims[l] * n.dims[l + 1];
    const float *rho = theta + n.rho_off[l];
    for (int i = threadIdx.x; i < (cnt + 3) / 4; i += blockDim.x) {
      const f32x4 z = box_muller4(philox4x32_10((unsigned)i, (unsigned)l | (BPRIOR_NET_ID << 16), stream, BPRIOR_TAG_EPS, k0, k1));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = 4 * i + u;
        if (idx < cnt) {
          dw[off + idx] = (BPRIOR_SCALE_EPS + bprior_softplus(rho[idx])) * z[u];
          if (ep) ep[off + idx] = z[u];
        }
      }
    }
    off += cnt;
  }
}

// sign words of `rows` rows (row r of the call is keyed by row0 + r) -> LDS [rows][words]
__device__ __forceinline__ void bprior_signs(const BPriorNet &n, unsigned *sg, int rows, unsigned row0, unsigned stream, unsigned k0, unsigned k1) {
  const int calls = n.words / 4;
  for (int e = threadIdx.x; e < rows * calls; e += blockDim.x) {
    const int r = e / calls, c = e - r * calls;
    const uint4 w = philox4x32_10(row0 + (unsigned)r, (unsigned)c | (BPRIOR_NET_ID << 16), stream, BPRIOR_TAG_SIGN, k0, k1);
    unsigned *o = sg + r * n.words + 4 * c;
    o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = w.w;
  }
}

// Flipout layers on `rows` rows whose normalised input sits in act[0 .. rows x k); layer l's output goes to act + a_off[l + 1]
// (row-major [rows x dims[l + 1]]; activated except the last).  pre (optional): the pre-activations' signs are recoverable from the
// outputs (LeakyReLU keeps the sign), so nothing else is stored.
__device__ __forceinline__ void bprior_layers(const BPriorNet &n, const float *theta, const float *dw, const unsigned *sg, float *act,
                                              const int *a_off, int rows) {
  int woff = 0;
  for (int l = 0; l < n.n_layers; ++l) {
    const int din = n.dims[l], dout = n.dims[l + 1];
    const float *loc = theta + n.loc_off[l], *bias = theta + n.bias_off[l], *d = dw + woff;
    const float *in = act + a_off[l];
    float *o = act + a_off[l + 1];
    for (int e = threadIdx.x; e < rows * dout; e += blockDim.x) {
      const int b = e / dout, c = e - b * dout;
      const unsigned *w = sg + b * n.words;
      float s0 = bias[c], s1 = 0.0f;
      for (int i = 0; i < din; ++i) {
        const float x = in[b * din + i];
        s0 = fmaf(x, loc[i * dout + c], s0);
        s1 = fmaf(x * bprior_sign(w + n.sin_w[l], i), d[i * dout + c], s1);
      }
      const float pre = fmaf(s1, bprior_sign(w + n.sout_w[l], c), s0);
      o[e] = (l + 1 < n.n_layers) ? (pre > 0.0f ? pre : BPRIOR_LEAK * pre) : pre;
    }
    woff += din * dout;
    __syncthreads();
  }
}

struct BPriorStepArgs {
  BPriorNet net;
  float *theta, *m, *v;                // prior parameters and their Adam slots
  const int *seg;                      // [n_rows]
  float *data_z;                       // [n_rows x q]
  const int *idx;                      // [B]
  const float *dz;                     // [B x q]: gradient of the batch-mean negative log joint with the STANDARD-normal prior
  int B, q;
  float lr_t_z, lr_t_p, b1, b2, eps, kl_weight;
  unsigned k0, k1, stream, row0;       // noise key and call id; row0: position of this rank's first row in the global minibatch
  float *out;                          // [3]: batch-mean conditional-prior term, batch-mean |z|^2 / 2, sum of the KL terms
  float inv_B;                         // 1 / (global) minibatch
  float *grad;                         // data-parallel: the net's gradient (without the KL part, which every rank adds in bprior_adam) or NULL
  int apply;                           // 1: Adam in place
};

// LDS (floats): act [B x sum(dims)] | xhat [B x k] | dw [n_kernel] | ep [n_kernel] | zb [B x q] | dlt [B x wmax] | dprev [B x wmax] |
//               red [3 B] | stat [2 k] | signs [B x words] (uint) | seg [B] (int)
static __global__ __launch_bounds__(BPRIOR_THREADS) void bprior_step_kernel(BPriorStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BPriorNet &n = a.net;
  const int B = a.B, q = a.q, L = n.n_layers, k = n.dims[0], tid = threadIdx.x;
  int a_off[BPRIOR_MAX_LAYERS + 1];
  int tot = 0;
  for (int l = 0; l <= L; ++l) { a_off[l] = tot; tot += B * n.dims[l]; }
  float *act = lds, *xhat = act + tot, *dw = xhat + B * k, *ep = dw + n.n_kernel, *zb = ep + n.n_kernel;
  float *dlt = zb + B * q, *dprev = dlt + B * n.wmax, *red = dprev + B * n.wmax, *stat = red + 3 * B;
  unsigned *sgw = reinterpret_cast<unsigned *>(stat + 2 * k);
  int *sg = reinterpret_cast<int *>(sgw + B * n.words);
  for (int e = tid; e < B * q; e += blockDim.x) zb[e] = a.data_z[(long long)a.idx[e / q] * q + (e % q)];
  for (int b = tid; b < B; b += blockDim.x) sg[b] = a.seg[a.idx[b]];
  bprior_noise(n, a.theta, dw, ep, a.stream, a.k0, a.k1);
  bprior_signs(n, sgw, B, a.row0, a.stream, a.k0, a.k1);
  __syncthreads();
  // input BatchNormalization of the one-hot rows (bnn.py:26)
  for (int i = tid; i < k; i += blockDim.x) {
    float mean = 0.0f, var = 1.0f;
    if (n.norm_mode == 0) {
      int cnt = 0;
      for (int b = 0; b < B; ++b) cnt += (sg[b] == i);
      mean = (float)cnt / (float)B;
      var = mean * (1.0f - mean);              // biased variance of a 0 / 1 column
    }
    stat[i] = mean; stat[k + i] = 1.0f / sqrtf(var + BPRIOR_BN_EPS);
  }
  __syncthreads();
  for (int e = tid; e < B * k; e += blockDim.x) {
    const int b = e / k, i = e - b * k;
    const float xh = ((sg[b] == i ? 1.0f : 0.0f) - stat[i]) * stat[k + i];
    xhat[e] = xh;
    act[e] = fmaf(xh, a.theta[n.gamma_off + i], a.theta[n.beta_off + i]);
  }
  __syncthreads();
  bprior_layers(n, a.theta, dw, sgw, act, a_off, B);
  const float *out = act + a_off[L];
  const float invB = a.inv_B;
  // loss and d loss / d out (:203-211)
  for (int b = tid; b < B; b += blockDim.x) {
    const float s2 = bprior_softplus(out[b * (q + 1) + q]) + BPRIOR_EPS;
    float ssq = 0.0f, zsq = 0.0f;
    for (int j = 0; j < q; ++j) {
      const float d = zb[b * q + j] - out[b * (q + 1) + j];
      ssq = fmaf(d, d, ssq);
      zsq = fmaf(zb[b * q + j], zb[b * q + j], zsq);
      dlt[b * n.wmax + j] = -d / s2 * invB;
    }
    dlt[b * n.wmax + q] = (-ssq / (2.0f * s2 * s2) + (float)q / (2.0f * s2)) * invB * bprior_sigmoid(out[b * (q + 1) + q]);
    red[b] = ssq / (2.0f * s2) + 0.5f * (float)q * logf(s2);
    red[B + b] = 0.5f * zsq;
  }
  __syncthreads();
  // latent step with fresh Adam slots (:216-217 on the Variable created at :304): g = dz(standard prior) - z / B + d / (s2 B)
  for (int e = tid; e < B * q; e += blockDim.x) {
    const int b = e / q, j = e - b * q;
    const float g = a.dz[e] - zb[e] * invB - dlt[b * n.wmax + j];
    const float m_ = (1.0f - a.b1) * g, v_ = (1.0f - a.b2) * g * g;
    a.data_z[(long long)a.idx[b] * q + j] = zb[e] - a.lr_t_z * m_ / (sqrtf(v_) + a.eps);
  }
  auto adam = [&](int p, float g, float g_kl) {
    if (a.grad) a.grad[p] = g;                   // (data term only: the KL part is the same on every rank and added after the all-reduce)
    if (!a.apply) return;
    g += g_kl;
    const float m_ = a.b1 * a.m[p] + (1.0f - a.b1) * g, v_ = a.b2 * a.v[p] + (1.0f - a.b2) * g * g;
    a.m[p] = m_; a.v[p] = v_;
    a.theta[p] -= a.lr_t_p * m_ / (sqrtf(v_) + a.eps);
  };
  // backward through the Flipout layers (oracle/bnn.py backward), Adam on every parameter by its owner thread (:220-222)
  float kl_part = 0.0f;
  int woff = n.n_kernel;
  for (int l = L - 1; l >= 0; --l) {
    const int din = n.dims[l], dout = n.dims[l + 1];
    woff -= din * dout;
    const float *in = act + a_off[l];
    const float *loc = a.theta + n.loc_off[l], *rho = a.theta + n.rho_off[l];
    // delta of this layer's input (through the OLD parameters), before the activation mask of the layer below
    for (int e = tid; e < B * din; e += blockDim.x) {
      const int b = e / din, i = e - b * din;
      const unsigned *w = sgw + b * n.words;
      float s0 = 0.0f, s1 = 0.0f;
      for (int c = 0; c < dout; ++c) {
        const float d = dlt[b * n.wmax + c];
        s0 = fmaf(d, loc[i * dout + c], s0);
        s1 = fmaf(d * bprior_sign(w + n.sout_w[l], c), dw[woff + i * dout + c], s1);
      }
      float s = fmaf(s1, bprior_sign(w + n.sin_w[l], i), s0);
      if (l > 0) s *= (in[e] > 0.0f ? 1.0f : BPRIOR_LEAK);
      dprev[b * n.wmax + i] = s;
    }
    __syncthreads();
    for (int e = tid; e < din * dout + dout; e += blockDim.x) {
      if (e < din * dout) {
        const int i = e / dout, c = e - i * dout;
        float gl = 0.0f, gd = 0.0f;
        for (int b = 0; b < B; ++b) {
          const unsigned *w = sgw + b * n.words;
          const float x = in[b * din + i], d = dlt[b * n.wmax + c];
          gl = fmaf(x, d, gl);
          gd = fmaf(x * bprior_sign(w + n.sin_w[l], i), d * bprior_sign(w + n.sout_w[l], c), gd);
        }
        const float r = rho[e], sgm = bprior_sigmoid(r), sc = BPRIOR_SCALE_EPS + bprior_softplus(r), mu = loc[e];
        kl_part += -logf(sc) + 0.5f * (sc * sc + mu * mu) - 0.5f;
        adam(n.loc_off[l] + e, gl, a.kl_weight * mu);
        adam(n.rho_off[l] + e, gd * ep[woff + e] * sgm, a.kl_weight * (sc - 1.0f / sc) * sgm);
      } else {
        const int c = e - din * dout;
        float g = 0.0f;
        for (int b = 0; b < B; ++b) g += dlt[b * n.wmax + c];
        adam(n.bias_off[l] + c, g, 0.0f);
      }
    }
    __syncthreads();
    float *t = dlt; dlt = dprev; dprev = t;
  }
  // gamma, beta of the input normalisation (the input is data: nothing flows further)
  for (int i = tid; i < k; i += blockDim.x) {
    float gg = 0.0f, gb = 0.0f;
    for (int b = 0; b < B; ++b) { gg = fmaf(dlt[b * n.wmax + i], xhat[b * k + i], gg); gb += dlt[b * n.wmax + i]; }
    adam(n.gamma_off + i, gg, 0.0f);
    adam(n.beta_off + i, gb, 0.0f);
  }
  // outputs: batch means and the KL sum (block reduction in a fixed order)
  __shared__ float klred[BPRIOR_THREADS];
  klred[tid] = kl_part;
  __syncthreads();
  if (tid == 0 && a.out) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    for (int b = 0; b < B; ++b) { s0 += red[b]; s1 += red[B + b]; }
    for (int t = 0; t < BPRIOR_THREADS; ++t) s2 += klred[t];
    a.out[0] = s0 * invB; a.out[1] = s1 * invB; a.out[2] = s2;
  }
}

// Adam step from a gradient buffer (data-parallel fit, after the all-reduce of the data term) + the KL part every rank adds itself
static __global__ void bprior_adam_kernel(BPriorNet n, float *theta, float *m, float *v, const float *grad, float lr_t, float b1, float b2, float eps,
                                          float kl_weight) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n.n_params) return;
  float g = grad[p];
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.dims[l] * n.dims[l + 1];
    if (p >= n.loc_off[l] && p < n.loc_off[l] + cnt) g += kl_weight * theta[p];
    if (p >= n.rho_off[l] && p < n.rho_off[l] + cnt) {
      const float r = theta[p], sc = BPRIOR_SCALE_EPS + bprior_softplus(r);
      g += kl_weight * (sc - 1.0f / sc) * bprior_sigmoid(r);
    }
  }
  const float m_ = b1 * m[p] + (1.0f - b1) * g, v_ = b2 * v[p] + (1.0f - b2) * g * g;
  m[p] = m_; v[p] = v_;
  theta[p] -= lr_t * m_ / (sqrtf(v_) + eps);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Sampling side: one noisy call of the prior net per (block of rows, log-posterior evaluation).  rows_out [n_states][n][q + 2] =
// mu [q], 1 / sigma^2, (q / 2) log sigma^2 of every row -- what the Metropolis-Hastings kernel reads (bnf_kernels.h).
// grid = (parts per block x n_blocks, n_states); a workgroup regenerates the call's sigma * eps (1.3 k values) and walks its share of
// the block's rows 64 at a time.  Block b's noise key is seed + (block0 + b) << 32, state s's call id stream0 + s; row r of the block
// draws its sign words as row r (the other Bayesian nets' convention).  Fixed normalisation only (the bnf sampling family's mode).
// ---------------------------------------------------------------------------------------------------------------------------
#define BPRIOR_ROWS_CHUNK 64
struct BPriorRowsArgs {
  BPriorNet net;
  const float *theta;
  const int *seg;                      // [n] segments of the rows of this call
  long long n;
  int bs, n_blocks, block0, parts, q;
  unsigned k0, k1, stream0;
  float *rows_out;                     // [n_states][n][q + 2]
  unsigned rib0;                       // position of the call's first row inside its block (0 unless a block's rows are split over ranks)
  const int *hist;                     // norm_mode 0: rows per (block, segment) [n_blocks][k] (bprior_hist_kernel)
};

// Batch statistics of the one-hot input (norm_mode 0, the reference as written: bnn.py:26 normalises with the statistics of the rows of
// the call = of the block): the mean of column i is the share of the block's rows in segment i.  One workgroup per block.
static __global__ __launch_bounds__(256) void bprior_hist_kernel(const int *seg, long long n, int bs, int k, int *hist) {
  __shared__ int cnt[64];
  const int blk = blockIdx.x;
  for (int i = threadIdx.x; i < k; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  const long long lo = (long long)blk * bs, hi = min(n, lo + bs);
  for (long long r = lo + threadIdx.x; r < hi; r += blockDim.x) {
    const int sg = seg[r];
    if (sg >= 0 && sg < k) atomicAdd(&cnt[sg], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < k; i += blockDim.x) hist[(long long)blk * k + i] = cnt[i];
}

static __global__ __launch_bounds__(BPRIOR_THREADS) void bprior_rows_kernel(BPriorRowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BPriorNet &n = a.net;
  const int L = n.n_layers, k = n.dims[0], q = a.q, tid = threadIdx.x, R = BPRIOR_ROWS_CHUNK;
  const int blk = blockIdx.x / a.parts, part = blockIdx.x - blk * a.parts, st = blockIdx.y;
  int a_off[BPRIOR_MAX_LAYERS + 1];
  int tot = 0;
  for (int l = 0; l <= L; ++l) { a_off[l] = tot; tot += R * n.dims[l]; }
  float *act = lds, *dw = act + tot;
  unsigned *sgw = reinterpret_cast<unsigned *>(dw + n.n_kernel);
  int *sg = reinterpret_cast<int *>(sgw + R * n.words);
  const unsigned k0 = a.k0, k1 = a.k1 + (unsigned)(a.block0 + blk), stream = a.stream0 + (unsigned)st;
  bprior_noise(n, a.theta, dw, nullptr, stream, k0, k1);
  const long long blk_lo = (long long)blk * a.bs;
  const int blk_n = (int)min((long long)a.bs, a.n - blk_lo);
  const float inv = 1.0f / sqrtf(1.0f + BPRIOR_BN_EPS);
  float *dst = a.rows_out + (long long)st * a.n * (q + 2);
  for (int r0 = part * R; r0 < blk_n; r0 += a.parts * R) {
    const int rows = min(R, blk_n - r0);
    __syncthreads();
    for (int b = tid; b < rows; b += blockDim.x) sg[b] = a.seg[blk_lo + r0 + b];
    bprior_signs(n, sgw, rows, a.rib0 + (unsigned)r0, stream, k0, k1);
    __syncthreads();
    for (int e = tid; e < rows * k; e += blockDim.x) {
      const int b = e / k, i = e - b * k;
      float xh = (sg[b] == i ? 1.0f : 0.0f) * inv;
      if (n.norm_mode == 0) {          // batch statistics of the block (bprior_step_kernel's arithmetic)
        const float mean = (float)a.hist[(long long)blk * k + i] / (float)blk_n;
        xh = ((sg[b] == i ? 1.0f : 0.0f) - mean) * (1.0f / sqrtf(mean * (1.0f - mean) + BPRIOR_BN_EPS));
      }
      act[e] = fmaf(xh, a.theta[n.gamma_off + i], a.theta[n.beta_off + i]);
    }
    __syncthreads();
    bprior_layers(n, a.theta, dw, sgw, act, a_off, rows);
    const float *o = act + a_off[L];
    for (int e = tid; e < rows * (q + 2); e += blockDim.x) {
      const int b = e / (q + 2), c = e - b * (q + 2);
      const float s2 = bprior_softplus(o[b * (q + 1) + q]) + BPRIOR_EPS;
      dst[(blk_lo + r0 + b) * (q + 2) + c] = c < q ? o[b * (q + 1) + c] : (c == q ? 1.0f / s2 : 0.5f * (float)q * logf(s2));
    }
  }
}
