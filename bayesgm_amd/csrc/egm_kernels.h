// egm_kernels.h -- EGM warm start of CausalBGM on gfx950.
//
// replaces (src/bayesgm/models/causalbgm/base.py):
//   train_disc_step :305-330  -> egm_disc_step_kernel   (WGAN-GP on the latent discriminator dz_net)
//   train_gen_step  :332-377  -> egm_gen_step_kernel    (g, e, f, h against the fixed discriminator)
// and the Discriminator of models/networks/base.py:338-385 (Dense -> BatchNorm(batch statistics) -> tanh).
//
// These are B = 32 minibatch steps: ~10 MFLOP each, 180 000 of them per default fit.  The work is
// latency, not throughput: one step = ONE launch of ONE 1024-thread workgroup that walks the whole
// forward / backward / (double-backward) / Adam sequence with workgroup barriers between the tiny
// dense ops; every weight matrix is staged through LDS once per use (odd row stride, so the forward
// W[i][o] and the transposed W^T accesses are both conflict-free), activations live in an L2-resident
// workspace.  No MFMA: at B = 32 the matrix pipe would idle on the barriers anyway.
// All gradient formulas are the hand-derived ones of oracle/egm.py (checked there against autograd);
// the gradient penalty needs reverse mode through the backward pass of a batch-normalised network.
#pragma once
#include <hip/hip_runtime.h>

#define EGM_THREADS 1024
#define EGM_MAX_LAYERS 8
#define EGM_LEAK 0.2f
#define EGM_BN_EPS 1e-3f

struct EgmMlp {            // BaseFullyConnectedNet: parameters at theta + off as W0,b0,W1,b1,... (Keras order)
  int n_layers;            // Dense layers (hidden + output)
  int dims[EGM_MAX_LAYERS + 1];
  int off;
};
struct EgmDisc {           // Discriminator: hidden layers with BatchNorm + tanh, linear scalar output
  int n_hidden;
  int dims[EGM_MAX_LAYERS + 1];          // in, h1..hL, 1
  int w[EGM_MAX_LAYERS], b[EGM_MAX_LAYERS], gamma[EGM_MAX_LAYERS], beta[EGM_MAX_LAYERS];   // offsets into theta_d
  int n_params;
};
struct EgmAdam { float lr_t, b1, b2, eps; };

struct EgmArgs {
  EgmMlp g, e, f, h;
  EgmDisc dz;
  float *theta_g, *m_g, *v_g, *grad_g;   // generator-side parameters [g|e|f|h], Adam slots, gradient scratch
  float *theta_d, *m_d, *v_d, *grad_d;   // discriminator parameters
  int n_gen, B, q, p;
  int wmax;                              // widest layer of any network (scratch row width)
  int z0, z1, z2;                        // z_dims[0..2]
  int binary, use_z_rec;
  const float *z;                        // [B x q] prior sample of this step
  const int *idx;                        // [B] rows of the panel
  const float *v, *x, *y;                // panel [N x p], [N], [N]
  float eps;                             // interpolation coefficient of the gradient penalty
  EgmAdam adam;
  float *ws;                             // workspace
  float *out;                            // disc: [dz_loss, d_loss]   gen: [e_adv, l2_v, l2_z, l2_x, l2_y, total]
  int apply;                             // 1: Adam step; 0: leave the gradients in grad_* (parity tests)
};

struct EgmCtx {
  int tid;
  float *lds;      // weight stage
  float *red;      // [32] reduction scratch (LDS)
};

__device__ __forceinline__ int egm_ldw(int out) { return out | 1; }

// stage W [in x out] (row-major, Keras) into LDS with an odd row stride
__device__ __forceinline__ void egm_stage(const EgmCtx &c, const float *W, int in, int out) {
  const int ld = egm_ldw(out);
  for (int k = c.tid; k < in * out; k += EGM_THREADS) {
    const int i = k / out, o = k - i * out;
    c.lds[i * ld + o] = W[k];
  }
  __syncthreads();
}

// Y = X W + bias  (optionally LeakyReLU);  W staged
__device__ __forceinline__ void egm_fwd(const EgmCtx &c, const float *X, int ldx, const float *W, const float *bias, float *Y,
                                        int ldy, int B, int in, int out, bool act) {
  egm_stage(c, W, in, out);
  const int ld = egm_ldw(out);
  for (int k = c.tid; k < B * out; k += EGM_THREADS) {
    const int b = k / out, o = k - b * out;
    float acc = bias ? bias[o] : 0.0f;
    const float *xr = X + (long long)b * ldx;
    for (int i = 0; i < in; ++i) acc = fmaf(xr[i], c.lds[i * ld + o], acc);
    if (act) acc = fmaxf(acc, EGM_LEAK * acc);
    Y[(long long)b * ldy + o] = acc;
  }
  __syncthreads();
}
// dX (+)= dY W^T
__device__ __forceinline__ void egm_bwd_in(const EgmCtx &c, const float *dY, int ldy, const float *W, float *dX, int ldx, int B,
                                           int in, int out, bool accumulate) {
  egm_stage(c, W, in, out);
  const int ld = egm_ldw(out);
  for (int k = c.tid; k < B * in; k += EGM_THREADS) {
    const int b = k / in, i = k - b * in;
    float acc = 0.0f;
    const float *dr = dY + (long long)b * ldy;
    for (int o = 0; o < out; ++o) acc = fmaf(dr[o], c.lds[i * ld + o], acc);
    float *dst = dX + (long long)b * ldx + i;
    *dst = accumulate ? *dst + acc : acc;
  }
  __syncthreads();
}
// gW (+)= s * X^T dY ;  gb (+)= s * column sums of dY
__device__ __forceinline__ void egm_bwd_w(const EgmCtx &c, const float *X, int ldx, const float *dY, int ldy, float *gW, float *gb,
                                          int B, int in, int out, bool accumulate, float s = 1.0f) {
  for (int k = c.tid; k < in * out + out; k += EGM_THREADS) {
    float acc = 0.0f;
    if (k < in * out) {
      const int i = k / out, o = k - i * out;
      for (int b = 0; b < B; ++b) acc = fmaf(X[(long long)b * ldx + i], dY[(long long)b * ldy + o], acc);
      gW[k] = accumulate ? gW[k] + s * acc : s * acc;
    } else if (gb) {
      const int o = k - in * out;
      for (int b = 0; b < B; ++b) acc += dY[(long long)b * ldy + o];
      gb[o] = accumulate ? gb[o] + s * acc : s * acc;
    }
  }
  __syncthreads();
}
// workgroup sum (every thread gets the result)
__device__ __forceinline__ float egm_block_sum(const EgmCtx &c, float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((c.tid & 63) == 0) c.red[c.tid >> 6] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < EGM_THREADS / 64; ++w) t += c.red[w];
  return t;
}

// ---------------------------------------------------------------------------------------------
// MLP forward (activations kept) / backward
// ---------------------------------------------------------------------------------------------
struct EgmMlpCache { float *act[EGM_MAX_LAYERS + 1]; };   // act[0] = input, act[L] = output; act[l] is [B x dims[l]]

__device__ __forceinline__ const float *egm_W(const float *theta, const EgmMlp &n, int l) {
  int o = n.off;
  for (int k = 0; k < l; ++k) o += n.dims[k] * n.dims[k + 1] + n.dims[k + 1];
  return theta + o;
}
__device__ __forceinline__ void egm_mlp_fwd(const EgmCtx &c, const float *theta, const EgmMlp &n, const EgmMlpCache &a, int B) {
  for (int l = 0; l < n.n_layers; ++l) {
    const float *W = egm_W(theta, n, l);
    egm_fwd(c, a.act[l], n.dims[l], W, W + n.dims[l] * n.dims[l + 1], a.act[l + 1], n.dims[l + 1], B, n.dims[l], n.dims[l + 1],
            l < n.n_layers - 1);
  }
}
// d: [B x dims[L]] upstream gradient (destroyed); tmp: scratch [B x max width]; grads at grad + n.off (same layout as theta)
// dx (may be NULL): [B x dims[0]] receives dLoss/dinput
__device__ __forceinline__ void egm_mlp_bwd(const EgmCtx &c, const float *theta, float *grad, const EgmMlp &n, const EgmMlpCache &a,
                                            float *d, float *tmp, float *dx, int B, bool accumulate) {
  float *cur = d, *nxt = tmp;
  for (int l = n.n_layers - 1; l >= 0; --l) {
    const int in = n.dims[l], out = n.dims[l + 1];
    if (l < n.n_layers - 1) {   // through LeakyReLU: the sign of the activation is the sign of the pre-activation
      for (int k = c.tid; k < B * out; k += EGM_THREADS) cur[k] *= (a.act[l + 1][k] > 0.0f) ? 1.0f : EGM_LEAK;
      __syncthreads();
    }
    const float *W = egm_W(theta, n, l);
    float *gW = grad + (W - theta);
    egm_bwd_w(c, a.act[l], in, cur, out, gW, gW + in * out, B, in, out, accumulate);
    if (l > 0) {
      egm_bwd_in(c, cur, out, W, nxt, in, B, in, out, false);
      float *t = cur; cur = nxt; nxt = t;
    } else if (dx) {
      egm_bwd_in(c, cur, out, W, dx, in, B, in, out, false);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Discriminator: forward with batch statistics, backward, gradient penalty (double backward)
// ---------------------------------------------------------------------------------------------
struct EgmDiscCache {
  float *a[EGM_MAX_LAYERS + 1];     // a[0] = input [B x dims[0]], a[l] = tanh output of hidden layer l
  float *uhat[EGM_MAX_LAYERS];      // [B x dims[l+1]]
  float *sigma[EGM_MAX_LAYERS];     // [dims[l+1]]
  float *out;                       // [B]
};

__device__ __forceinline__ void egm_disc_fwd(const EgmCtx &c, const float *th, const EgmDisc &d, const EgmDiscCache &k, int B) {
  const int L = d.n_hidden;
  for (int l = 0; l < L; ++l) {
    const int in = d.dims[l], out = d.dims[l + 1];
    egm_fwd(c, k.a[l], in, th + d.w[l], th + d.b[l], k.uhat[l], out, B, in, out, false);   // u (normalised in place below)
    for (int o = c.tid; o < out; o += EGM_THREADS) {
      float mu = 0.0f;
      for (int b = 0; b < B; ++b) mu += k.uhat[l][b * out + o];
      mu /= (float)B;
      float var = 0.0f;
      for (int b = 0; b < B; ++b) { const float t = k.uhat[l][b * out + o] - mu; var = fmaf(t, t, var); }
      var /= (float)B;
      const float sg = sqrtf(var + EGM_BN_EPS);
      k.sigma[l][o] = sg;
      const float ga = th[d.gamma[l] + o], be = th[d.beta[l] + o];
      for (int b = 0; b < B; ++b) {
        const float uh = (k.uhat[l][b * out + o] - mu) / sg;
        k.uhat[l][b * out + o] = uh;
        k.a[l + 1][b * out + o] = tanhf(fmaf(uh, ga, be));
      }
    }
    __syncthreads();
  }
  egm_fwd(c, k.a[L], d.dims[L], th + d.w[L], th + d.b[L], k.out, 1, B, d.dims[L], 1, false);
}

// x - mean_b x - uhat * mean_b(x uhat), per feature column o (one thread per column)
__device__ __forceinline__ void egm_bn_proj_col(float *x, const float *uhat, int B, int out, int o, float scale) {
  float m1 = 0.0f, m2 = 0.0f;
  for (int b = 0; b < B; ++b) { m1 += x[b * out + o]; m2 = fmaf(x[b * out + o], uhat[b * out + o], m2); }
  m1 /= (float)B; m2 /= (float)B;
  for (int b = 0; b < B; ++b) x[b * out + o] = (x[b * out + o] - m1 - uhat[b * out + o] * m2) * scale;
}

struct EgmDiscAdj {   // extra adjoints on forward nodes (gradient-penalty reverse pass); NULL pointers = none
  float *a_bar[EGM_MAX_LAYERS], *uhat_bar[EGM_MAX_LAYERS], *sigma_bar[EGM_MAX_LAYERS];
};

// Ordinary backward.  dout_val: dLoss/dout of every row (has_dout), grads accumulate (scaled by s) when `accumulate`.
// da / du: scratch [B x max width].  dx (may be NULL) receives dLoss/dinput [B x dims[0]].
__device__ __forceinline__ void egm_disc_bwd(const EgmCtx &c, const float *th, float *gr, const EgmDisc &d, const EgmDiscCache &k,
                                             bool has_dout, float dout_val, const EgmDiscAdj *adj, float *da, float *du, float *dx,
                                             int B, bool accumulate, float s) {
  const int L = d.n_hidden, nL = d.dims[L];
  if (has_dout) {
    for (int i = c.tid; i < nL + 1; i += EGM_THREADS) {
      float acc = 0.0f;
      if (i < nL) { for (int b = 0; b < B; ++b) acc += k.a[L][b * nL + i]; acc *= dout_val; }
      else acc = dout_val * (float)B;
      float *dst = gr + (i < nL ? d.w[L] + i : d.b[L]);
      *dst = accumulate ? *dst + s * acc : s * acc;
    }
    for (int t = c.tid; t < B * nL; t += EGM_THREADS) da[t] = dout_val * th[d.w[L] + (t % nL)];
  } else {
    for (int t = c.tid; t < B * nL; t += EGM_THREADS) da[t] = 0.0f;
    if (!accumulate)
      for (int i = c.tid; i < nL + 1; i += EGM_THREADS) gr[i < nL ? d.w[L] + i : d.b[L]] = 0.0f;
  }
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    const int in = d.dims[l], out = d.dims[l + 1];
    // dy = (da + a_bar) (1 - a^2);  dgamma, dbeta;  duhat = dy gamma + uhat_bar;  du = proj(duhat)/sigma + sigma_bar uhat / B
    for (int o = c.tid; o < out; o += EGM_THREADS) {
      float gg = 0.0f, gb = 0.0f;
      const float ga = th[d.gamma[l] + o];
      for (int b = 0; b < B; ++b) {
        const int t = b * out + o;
        float dav = da[t];
        if (adj && adj->a_bar[l]) dav += adj->a_bar[l][t];
        const float av = k.a[l + 1][t];
        const float dy = dav * (1.0f - av * av);
        gg = fmaf(dy, k.uhat[l][t], gg);
        gb += dy;
        float dh = dy * ga;
        if (adj && adj->uhat_bar[l]) dh += adj->uhat_bar[l][t];
        du[t] = dh;
      }
      float *pg = gr + d.gamma[l] + o, *pb = gr + d.beta[l] + o;
      *pg = accumulate ? *pg + s * gg : s * gg;
      *pb = accumulate ? *pb + s * gb : s * gb;
      egm_bn_proj_col(du, k.uhat[l], B, out, o, 1.0f / k.sigma[l][o]);
      if (adj && adj->sigma_bar[l]) {
        const float sb = adj->sigma_bar[l][o] / (float)B;
        for (int b = 0; b < B; ++b) du[b * out + o] = fmaf(sb, k.uhat[l][b * out + o], du[b * out + o]);
      }
    }
    __syncthreads();
    egm_bwd_w(c, k.a[l], in, du, out, gr + d.w[l], gr + d.b[l], B, in, out, accumulate, s);
    if (l > 0) egm_bwd_in(c, du, out, th + d.w[l], da, in, B, in, out, false);
    else if (dx) egm_bwd_in(c, du, out, th + d.w[l], dx, in, B, in, out, false);
  }
}

// Gradient penalty GP = mean_b (||g_b|| - 1)^2 on the batch whose forward cache is k; accumulates s * dGP/dtheta
// into gr (which must already hold valid values) and returns GP.  Scratch: per hidden layer dy, dhat, du, da
// ([B x width] each) inside `scr`, plus the adjoint buffers.
__device__ __forceinline__ float egm_disc_gp(const EgmCtx &c, const float *th, float *gr, const EgmDisc &d, const EgmDiscCache &k,
                                             float *scr, int B, float s, int wmax) {
  const int L = d.n_hidden;
  // ---- adjoint network: g = d(sum_b out_b)/d input
  float *da_[EGM_MAX_LAYERS + 1], *dy_[EGM_MAX_LAYERS], *dhat_[EGM_MAX_LAYERS], *du_[EGM_MAX_LAYERS];
  EgmDiscAdj adj;
  int off = 0;
  auto take = [&](int n) { float *p_ = scr + off; off += (n + 3) & ~3; return p_; };
  for (int l = 0; l <= L; ++l) da_[l] = take(B * d.dims[l]);
  for (int l = 0; l < L; ++l) {
    const int w = B * d.dims[l + 1];
    dy_[l] = take(w); dhat_[l] = take(w); du_[l] = take(w);
    adj.a_bar[l] = take(w); adj.uhat_bar[l] = take(w); adj.sigma_bar[l] = take(d.dims[l + 1]);
  }
  float *bar_a = take(B * wmax), *bar_b = take(B * wmax), *tmp = take(B * wmax);
  const int nL = d.dims[L];
  for (int t = c.tid; t < B * nL; t += EGM_THREADS) da_[L][t] = th[d.w[L] + (t % nL)];
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    const int in = d.dims[l], out = d.dims[l + 1];
    for (int o = c.tid; o < out; o += EGM_THREADS) {
      const float ga = th[d.gamma[l] + o];
      for (int b = 0; b < B; ++b) {
        const int t = b * out + o;
        const float av = k.a[l + 1][t];
        const float dy = da_[l + 1][t] * (1.0f - av * av);
        dy_[l][t] = dy;
        dhat_[l][t] = dy * ga;
        du_[l][t] = dy * ga;
      }
      egm_bn_proj_col(du_[l], k.uhat[l], B, out, o, 1.0f / k.sigma[l][o]);
    }
    __syncthreads();
    egm_bwd_in(c, du_[l], out, th + d.w[l], da_[l], in, B, in, out, false);
  }
  // ---- penalty and its adjoint on g = da_[0]
  const int q = d.dims[0];
  float part = 0.0f;
  for (int b = c.tid; b < B; b += EGM_THREADS) {
    float n2 = 0.0f;
    for (int i = 0; i < q; ++i) n2 = fmaf(da_[0][b * q + i], da_[0][b * q + i], n2);
    const float nrm = sqrtf(n2);
    part += (nrm - 1.0f) * (nrm - 1.0f);
    const float coef = 2.0f * (nrm - 1.0f) / nrm / (float)B;
    for (int i = 0; i < q; ++i) bar_a[b * q + i] = coef * da_[0][b * q + i];
  }
  const float gp = egm_block_sum(c, part) / (float)B;
  // ---- reverse through the adjoint network, bottom (l = 0) to top
  float *da_bar = bar_a, *du_bar = bar_b;
  for (int l = 0; l < L; ++l) {
    const int in = d.dims[l], out = d.dims[l + 1];
    // da_{l-1} = du W^T :  W_bar += da_bar^T du ;  du_bar = da_bar W
    egm_bwd_w(c, da_bar, in, du_[l], out, gr + d.w[l], nullptr, B, in, out, true, s);
    egm_fwd(c, da_bar, in, th + d.w[l], nullptr, du_bar, out, B, in, out, false);
    // du = (dhat - m1 - uhat m2) / sigma
    for (int o = c.tid; o < out; o += EGM_THREADS) {
      const float sg = k.sigma[l][o], ga = th[d.gamma[l] + o];
      float sb = 0.0f, m2 = 0.0f, tu = 0.0f;
      for (int b = 0; b < B; ++b) {
        const int t = b * out + o;
        sb = fmaf(du_bar[t], du_[l][t], sb);
        m2 = fmaf(dhat_[l][t], k.uhat[l][t], m2);
        tu = fmaf(du_bar[t] / sg, k.uhat[l][t], tu);
      }
      adj.sigma_bar[l][o] = -sb / sg;
      m2 /= (float)B; tu /= (float)B;
      for (int b = 0; b < B; ++b) {
        const int t = b * out + o;
        const float tt = du_bar[t] / sg;
        adj.uhat_bar[l][t] = -(tt * m2 + dhat_[l][t] * tu);
        tmp[t] = tt;
      }
      egm_bn_proj_col(tmp, k.uhat[l], B, out, o, 1.0f);          // dhat_bar
      float gg = 0.0f;
      for (int b = 0; b < B; ++b) {
        const int t = b * out + o;
        gg = fmaf(tmp[t], dy_[l][t], gg);
        const float dyb = tmp[t] * ga;
        const float av = k.a[l + 1][t];
        adj.a_bar[l][t] = dyb * da_[l + 1][t] * (-2.0f * av);
        tmp[t] = dyb * (1.0f - av * av);                          // da_bar of the layer above
      }
      gr[d.gamma[l] + o] += s * gg;
    }
    __syncthreads();
    // tmp now holds da_bar for layer l+1 : move it into the ping-pong buffer
    for (int t = c.tid; t < B * out; t += EGM_THREADS) da_bar[t] = tmp[t];
    __syncthreads();
  }
  for (int i = c.tid; i < nL; i += EGM_THREADS) {
    float acc = 0.0f;
    for (int b = 0; b < B; ++b) acc += da_bar[b * nL + i];
    gr[d.w[L] + i] += s * acc;
  }
  __syncthreads();
  // ---- ... and on through the forward pass (scratch: bar_b, tmp)
  egm_disc_bwd(c, th, gr, d, k, false, 0.0f, &adj, bar_b, tmp, nullptr, B, true, s);
  return gp;
}

__device__ __forceinline__ void egm_adam(const EgmCtx &c, float *theta, float *m, float *v, const float *g, int n, const EgmAdam &a) {
  for (int i = c.tid; i < n; i += EGM_THREADS) {
    const float gi = g[i];
    const float mi = a.b1 * m[i] + (1.0f - a.b1) * gi;
    const float vi = a.b2 * v[i] + (1.0f - a.b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    theta[i] -= a.lr_t * mi / (sqrtf(vi) + a.eps);
  }
}

// carve a discriminator cache out of the workspace
__device__ __forceinline__ void egm_disc_cache(const EgmDisc &d, int B, float *&p, EgmDiscCache &k, float *input) {
  auto take = [&](int n) { float *r = p; p += (n + 3) & ~3; return r; };
  k.a[0] = input;
  for (int l = 0; l < d.n_hidden; ++l) {
    k.a[l + 1] = take(B * d.dims[l + 1]);
    k.uhat[l] = take(B * d.dims[l + 1]);
    k.sigma[l] = take(d.dims[l + 1]);
  }
  k.out = take(B);
}
__device__ __forceinline__ void egm_mlp_cache(const EgmMlp &n, int B, float *&p, EgmMlpCache &a, float *input) {
  auto take = [&](int k) { float *r = p; p += (k + 3) & ~3; return r; };
  a.act[0] = input;
  for (int l = 0; l < n.n_layers; ++l) a.act[l + 1] = take(B * n.dims[l + 1]);
}

// ---------------------------------------------------------------------------------------------
// train_disc_step
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EGM_THREADS) void egm_disc_step_kernel(EgmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  EgmCtx c{(int)threadIdx.x, egm_lds + 32, egm_lds};
  const int B = a.B, q = a.q, p = a.p;
  float *wp = a.ws;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  float *vb = take(B * p), *zhat = take(B * q);
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const int b = k / p; vb[k] = a.v[(long long)a.idx[b] * p + (k - b * p)]; }
  __syncthreads();
  EgmMlpCache ce;
  egm_mlp_cache(a.e, B, wp, ce, vb);
  egm_mlp_fwd(c, a.theta_g, a.e, ce, B);                      // z_ = e(v)   (encoder fixed in this step)
  float *z_ = ce.act[a.e.n_layers];
  for (int k = c.tid; k < B * q; k += EGM_THREADS) zhat[k] = a.z[k] * a.eps + z_[k] * (1.0f - a.eps);
  __syncthreads();
  EgmDiscCache kf, kr, kh;
  egm_disc_cache(a.dz, B, wp, kf, z_);
  egm_disc_cache(a.dz, B, wp, kr, const_cast<float *>(a.z));
  egm_disc_cache(a.dz, B, wp, kh, zhat);
  float *da = take(B * a.wmax), *du = take(B * a.wmax);
  egm_disc_fwd(c, a.theta_d, a.dz, kf, B);
  egm_disc_fwd(c, a.theta_d, a.dz, kr, B);
  egm_disc_fwd(c, a.theta_d, a.dz, kh, B);
  float sf = 0.0f, sr = 0.0f;
  for (int b = c.tid; b < B; b += EGM_THREADS) { sf += kf.out[b]; sr += kr.out[b]; }
  sf = egm_block_sum(c, sf); sr = egm_block_sum(c, sr);
  const float dz_loss = (-sr + sf) / (float)B;
  egm_disc_bwd(c, a.theta_d, a.grad_d, a.dz, kf, true, 1.0f / (float)B, nullptr, da, du, nullptr, B, false, 1.0f);
  egm_disc_bwd(c, a.theta_d, a.grad_d, a.dz, kr, true, -1.0f / (float)B, nullptr, da, du, nullptr, B, true, 1.0f);
  const float gp = egm_disc_gp(c, a.theta_d, a.grad_d, a.dz, kh, wp, B, 10.0f, a.wmax);
  __syncthreads();
  if (a.apply) egm_adam(c, a.theta_d, a.m_d, a.v_d, a.grad_d, a.dz.n_params, a.adam);
  if (c.tid == 0 && a.out) { a.out[0] = dz_loss; a.out[1] = dz_loss + 10.0f * gp; }
}

// ---------------------------------------------------------------------------------------------
// train_gen_step
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EGM_THREADS) void egm_gen_step_kernel(EgmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  EgmCtx c{(int)threadIdx.x, egm_lds + 32, egm_lds};
  const int B = a.B, q = a.q, p = a.p, z0 = a.z0, z1 = a.z1, z2 = a.z2;
  const float invB = 1.0f / (float)B;
  float *wp = a.ws;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  float *vb = take(B * p), *xb = take(B), *yb = take(B);
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const int b = k / p; vb[k] = a.v[(long long)a.idx[b] * p + (k - b * p)]; }
  for (int b = c.tid; b < B; b += EGM_THREADS) { xb[b] = a.x[a.idx[b]]; yb[b] = a.y[a.idx[b]]; }
  __syncthreads();
  const int Lg = a.g.n_layers, Le = a.e.n_layers, Lf = a.f.n_layers, Lh = a.h.n_layers;
  const int wg = p + 1, nf = a.f.dims[0], nh = a.h.dims[0], of = a.f.dims[Lf], oh = a.h.dims[Lh];
  // ---- forward
  EgmMlpCache g1, e1, e2, g2, cf, ch;
  egm_mlp_cache(a.g, B, wp, g1, const_cast<float *>(a.z));
  egm_mlp_fwd(c, a.theta_g, a.g, g1, B);                       // g(z): v_ = [:, :p], sigma head [:, p]
  float *gz = g1.act[Lg];
  egm_mlp_cache(a.e, B, wp, e1, vb);
  egm_mlp_fwd(c, a.theta_g, a.e, e1, B);                       // z_ = e(v)
  float *z_ = e1.act[Le];
  float *v_ = take(B * p);                                     // contiguous copy of g(z)[:, :p]
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const int b = k / p; v_[k] = gz[b * wg + (k - b * p)]; }
  __syncthreads();
  egm_mlp_cache(a.e, B, wp, e2, v_);
  egm_mlp_fwd(c, a.theta_g, a.e, e2, B);                       // z__ = e(v_)
  float *z__ = e2.act[Le];
  egm_mlp_cache(a.g, B, wp, g2, z_);
  egm_mlp_fwd(c, a.theta_g, a.g, g2, B);                       // g(z_): v__ = [:, :p]
  float *gv = g2.act[Lg];
  EgmDiscCache kd;
  egm_disc_cache(a.dz, B, wp, kd, z_);
  egm_disc_fwd(c, a.theta_d, a.dz, kd, B);
  float *fin = take(B * nf), *hin = take(B * nh);
  for (int k = c.tid; k < B * nf; k += EGM_THREADS) {
    const int b = k / nf, i = k - b * nf;
    fin[k] = (i < z0 + z1) ? z_[b * q + i] : xb[b];
  }
  for (int k = c.tid; k < B * nh; k += EGM_THREADS) {
    const int b = k / nh, i = k - b * nh;
    hin[k] = (i < z0) ? z_[b * q + i] : z_[b * q + z1 + i];     // z2 block starts at z0 + z1
  }
  __syncthreads();
  egm_mlp_cache(a.f, B, wp, cf, fin);
  egm_mlp_fwd(c, a.theta_g, a.f, cf, B);
  egm_mlp_cache(a.h, B, wp, ch, hin);
  egm_mlp_fwd(c, a.theta_g, a.h, ch, B);
  float *fo = cf.act[Lf], *ho = ch.act[Lh];
  // ---- losses
  float l_v = 0.0f, l_z = 0.0f, l_x = 0.0f, l_y = 0.0f, s_g = 0.0f, s_f = 0.0f, s_h = 0.0f, adv = 0.0f;
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const int b = k / p; const float t = vb[k] - gv[b * wg + (k - b * p)]; l_v = fmaf(t, t, l_v); }
  for (int k = c.tid; k < B * q; k += EGM_THREADS) { const float t = a.z[k] - z__[k]; l_z = fmaf(t, t, l_z); }
  for (int b = c.tid; b < B; b += EGM_THREADS) {
    const float xl = ho[b * oh], yl = fo[b * of];
    if (a.binary) l_x += fmaxf(xl, 0.0f) - xl * xb[b] + log1pf(expf(-fabsf(xl)));
    else l_x += (xl - xb[b]) * (xl - xb[b]);
    l_y += (yl - yb[b]) * (yl - yb[b]);
    s_g += gz[b * wg + p] * gz[b * wg + p];
    s_f += fo[b * of + of - 1] * fo[b * of + of - 1];
    s_h += ho[b * oh + oh - 1] * ho[b * oh + oh - 1];
    adv -= kd.out[b];
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
  float *d0 = take(B * wmax), *d1 = take(B * wmax), *dzsum = take(B * q), *dtmp = take(B * wmax);
  float *da = take(B * wmax), *du = take(B * wmax), *dfin = take(B * nf), *dhin = take(B * nh);
  // z__ branch: e (call 2, input v_) -> g (call 1)
  for (int k = c.tid; k < B * q; k += EGM_THREADS) d0[k] = zrec * (-2.0f / (float)(B * q)) * (a.z[k] - z__[k]);
  __syncthreads();
  egm_mlp_bwd(c, a.theta_g, a.grad_g, a.e, e2, d0, d1, dtmp, B, false);                // dtmp = dLoss/dv_  [B x p]
  for (int k = c.tid; k < B * wg; k += EGM_THREADS) {
    const int b = k / wg, i = k - b * wg;
    d0[k] = (i < p) ? dtmp[b * p + i] : 0.001f * 2.0f * gz[k] * invB;
  }
  __syncthreads();
  egm_mlp_bwd(c, a.theta_g, a.grad_g, a.g, g1, d0, d1, nullptr, B, false);
  // v__ branch: g (call 2, input z_)
  for (int k = c.tid; k < B * wg; k += EGM_THREADS) {
    const int b = k / wg, i = k - b * wg;
    d0[k] = (i < p) ? (-2.0f / (float)(B * p)) * (vb[b * p + i] - gv[k]) : 0.0f;
  }
  __syncthreads();
  egm_mlp_bwd(c, a.theta_g, a.grad_g, a.g, g2, d0, d1, dzsum, B, true);                // dzsum = dLoss/dz_ (so far)
  // adversarial branch through the fixed discriminator (its gradients go to a scratch area past the workspace use)
  float *gd_scratch = take(a.dz.n_params);
  egm_disc_bwd(c, a.theta_d, gd_scratch, a.dz, kd, true, -invB, nullptr, da, du, dtmp, B, false, 1.0f);
  for (int k = c.tid; k < B * q; k += EGM_THREADS) dzsum[k] += dtmp[k];
  __syncthreads();
  // f, h branches
  for (int k = c.tid; k < B * of; k += EGM_THREADS) {
    const int b = k / of, i = k - b * of;
    float t = 0.0f;
    if (i == 0) t += 2.0f * (fo[k] - yb[b]) * invB;
    if (i == of - 1) t += 0.001f * 2.0f * fo[k] * invB;
    d0[k] = t;
  }
  __syncthreads();
  egm_mlp_bwd(c, a.theta_g, a.grad_g, a.f, cf, d0, d1, dfin, B, false);
  for (int k = c.tid; k < B * oh; k += EGM_THREADS) {
    const int b = k / oh, i = k - b * oh;
    float t = 0.0f;
    if (i == 0) t += a.binary ? (1.0f / (1.0f + expf(-ho[k])) - xb[b]) * invB : 2.0f * (ho[k] - xb[b]) * invB;
    if (i == oh - 1) t += 0.001f * 2.0f * ho[k] * invB;
    d0[k] = t;
  }
  __syncthreads();
  egm_mlp_bwd(c, a.theta_g, a.grad_g, a.h, ch, d0, d1, dhin, B, false);
  for (int k = c.tid; k < B * q; k += EGM_THREADS) {
    const int b = k / q, i = k - b * q;
    float t = dzsum[k];
    if (i < z0) t += dfin[b * nf + i] + dhin[b * nh + i];
    else if (i < z0 + z1) t += dfin[b * nf + i];
    else if (i < z0 + z1 + z2) t += dhin[b * nh + (i - z1)];
    d0[k] = t;
  }
  __syncthreads();
  egm_mlp_bwd(c, a.theta_g, a.grad_g, a.e, e1, d0, d1, nullptr, B, true);
  if (a.apply) egm_adam(c, a.theta_g, a.m_g, a.v_g, a.grad_g, a.n_gen, a.adam);
  if (c.tid == 0 && a.out) {
    a.out[0] = adv; a.out[1] = l_v; a.out[2] = l_z; a.out[3] = l_x; a.out[4] = l_y;
    a.out[5] = adv + (l_v + zrec * l_z) + (l_x + l_y) + 0.001f * sig;
  }
}
