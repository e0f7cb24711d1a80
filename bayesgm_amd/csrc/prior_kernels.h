// prior_kernels.h -- the conditional-prior network of IdentifiableCausalBGM on the device.
//
// replaces (src/bayesgm/models/causalbgm/identifiable.py):
//   prior_net = BaseFullyConnectedNet(n_segments -> prior_units -> q + 1)                              :76-78
//   update_latent_variable_sgd: prior term of the loss, joint step on the batch latents and the prior net  :195-226
//   the per-segment (mu, sigma^2) the sampler's log-posterior uses                                     :541-551
// A minibatch is 32 rows and the network a few thousand parameters: one workgroup, activations in LDS, every parameter's
// gradient formed in a register by the thread that owns the parameter (fixed summation order: deterministic) and consumed by
// its Adam update on the spot.  No MFMA: 32 x 64 x 11 MACs are not worth a tile.
#pragma once
#include <hip/hip_runtime.h>

#define PRIOR_MAX_LAYERS 4
#define PRIOR_THREADS 256
#define PRIOR_LEAK 0.2f
#define PRIOR_EPS 1e-6f

struct PriorNet {
  int n_layers;                        // dense layers
  int dims[PRIOR_MAX_LAYERS + 1];      // n_segments, hidden..., q + 1
  int w_off[PRIOR_MAX_LAYERS], b_off[PRIOR_MAX_LAYERS];   // Keras order: W_l [in x out] row-major, then b_l
  int a_off[PRIOR_MAX_LAYERS + 1];     // LDS offsets of the layer outputs [B x dims[l + 1]] (a_off[l] = output of layer l)
  int n_params, wmax;
};

struct PriorStepArgs {
  PriorNet net;
  float *theta, *m, *v;                // prior parameters and their Adam slots
  const int *seg;                      // [n_rows]
  float *data_z;                       // [n_rows x q]
  const int *idx;                      // [B]
  const float *dz;                     // [B x q]: gradient of the batch-mean negative log joint with the STANDARD-normal prior
  int B, q;
  float lr_t_z, lr_t_p, b1, b2, eps;
  float *out;                          // [2]: batch-mean conditional-prior term, batch-mean |z|^2 / 2
  float inv_B;                         // 1 / (global) minibatch: data-parallel steps pass the rows of all ranks (the local sums then add up)
  float *grad;                         // data-parallel: the prior net's gradient goes here [n_params] and the Adam step is left to
  int apply;                           //   prior_adam_kernel after the caller's all-reduce (apply = 0); apply = 1: Adam in place
};

__device__ __forceinline__ float prior_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// forward of rows whose first-layer input is the one-hot of seg_of(b): layer 0 is a row lookup
template <class SegOf>
__device__ __forceinline__ void prior_forward(const PriorNet &n, const float *theta, float *lds, int rows, SegOf seg_of) {
  for (int l = 0; l < n.n_layers; ++l) {
    const int din = n.dims[l], dout = n.dims[l + 1];
    const float *W = theta + n.w_off[l], *bias = theta + n.b_off[l];
    float *o = lds + n.a_off[l];
    const float *in = l ? lds + n.a_off[l - 1] : nullptr;
    for (int e = threadIdx.x; e < rows * dout; e += blockDim.x) {
      const int b = e / dout, c = e - b * dout;
      float s = bias[c];
      if (l == 0) s += W[seg_of(b) * dout + c];
      else
        for (int i = 0; i < din; ++i) s = fmaf(in[b * din + i], W[i * dout + c], s);
      o[e] = (l + 1 < n.n_layers) ? (s > 0.0f ? s : PRIOR_LEAK * s) : s;
    }
    __syncthreads();
  }
}

// table [n_segments x (q + 2)] = mu, 1 / sigma^2, (q / 2) log sigma^2  (what bgm_causal_set_prior takes)
static __global__ __launch_bounds__(PRIOR_THREADS) void prior_table_kernel(PriorNet n, const float *theta, float *table, int q) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int k = n.dims[0];
  prior_forward(n, theta, lds, k, [](int b) { return b; });
  const float *o = lds + n.a_off[n.n_layers - 1];
  for (int e = threadIdx.x; e < k * (q + 2); e += blockDim.x) {
    const int s = e / (q + 2), c = e - s * (q + 2);
    const float s2 = prior_softplus(o[s * (q + 1) + q]) + PRIOR_EPS;
    table[e] = c < q ? o[s * (q + 1) + c] : (c == q ? 1.0f / s2 : 0.5f * (float)q * logf(s2));
  }
}

static __global__ __launch_bounds__(PRIOR_THREADS) void prior_step_kernel(PriorStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const PriorNet &n = a.net;
  const int B = a.B, q = a.q, L = n.n_layers, tid = threadIdx.x;
  float *zb = lds + n.a_off[L];                 // [B x q] batch latents
  float *dlt = zb + B * q;                      // [B x wmax] delta of the layer being processed
  float *dprev = dlt + B * n.wmax;              // [B x wmax] delta handed to the layer below
  float *red = dprev + B * n.wmax;              // [2 x B] per-row loss terms
  int *sg = reinterpret_cast<int *>(red + 2 * B);   // [B] segments of the batch rows
  for (int e = tid; e < B * q; e += blockDim.x) zb[e] = a.data_z[(long long)a.idx[e / q] * q + (e % q)];
  for (int b = tid; b < B; b += blockDim.x) sg[b] = a.seg[a.idx[b]];
  __syncthreads();
  prior_forward(n, a.theta, lds, B, [sg](int b) { return sg[b]; });
  const float *out = lds + n.a_off[L - 1];
  const float invB = a.inv_B;
  // loss and d loss / d out (:203-211); one thread per row for the row sums
  for (int b = tid; b < B; b += blockDim.x) {
    const float s2 = prior_softplus(out[b * (q + 1) + q]) + PRIOR_EPS;
    float ssq = 0.0f, zsq = 0.0f;
    for (int j = 0; j < q; ++j) {
      const float d = zb[b * q + j] - out[b * (q + 1) + j];
      ssq = fmaf(d, d, ssq);
      zsq = fmaf(zb[b * q + j], zb[b * q + j], zsq);
      dlt[b * n.wmax + j] = -d / s2 * invB;
    }
    const float sgm = 1.0f / (1.0f + expf(-out[b * (q + 1) + q]));
    dlt[b * n.wmax + q] = (-ssq / (2.0f * s2 * s2) + (float)q / (2.0f * s2)) * invB * sgm;
    red[b] = ssq / (2.0f * s2) + 0.5f * (float)q * logf(s2);
    red[B + b] = 0.5f * zsq;
  }
  __syncthreads();
  if (tid == 0 && a.out) {
    float s0 = 0.0f, s1 = 0.0f;
    for (int b = 0; b < B; ++b) { s0 += red[b]; s1 += red[B + b]; }
    a.out[0] = s0 * invB; a.out[1] = s1 * invB;
  }
  // latent step with fresh Adam slots (:216-217 on the Variable created at :304): g = dz(standard prior) - z / B + d / (s2 B)
  for (int e = tid; e < B * q; e += blockDim.x) {
    const int b = e / q, j = e - b * q;
    const float g = a.dz[e] - zb[e] * invB - dlt[b * n.wmax + j];
    const float m_ = (1.0f - a.b1) * g, v_ = (1.0f - a.b2) * g * g;
    a.data_z[(long long)a.idx[b] * q + j] = zb[e] - a.lr_t_z * m_ / (sqrtf(v_) + a.eps);
  }
  // backward through the prior net, Adam on every parameter by its owner thread (:220-222)
  auto adam = [&](int p, float g) {
    if (a.grad) a.grad[p] = g;
    if (!a.apply) return;
    const float m_ = a.b1 * a.m[p] + (1.0f - a.b1) * g, v_ = a.b2 * a.v[p] + (1.0f - a.b2) * g * g;
    a.m[p] = m_; a.v[p] = v_;
    a.theta[p] -= a.lr_t_p * m_ / (sqrtf(v_) + a.eps);
  };
  for (int l = L - 1; l >= 0; --l) {
    const int din = n.dims[l], dout = n.dims[l + 1];
    const float *in = l ? lds + n.a_off[l - 1] : nullptr;
    const float *W = a.theta + n.w_off[l];
    if (l > 0) {          // delta of the layer below, with the OLD weights of this layer
      for (int e = tid; e < B * din; e += blockDim.x) {
        const int b = e / din, i = e - b * din;
        float s = 0.0f;
        for (int c = 0; c < dout; ++c) s = fmaf(W[i * dout + c], dlt[b * n.wmax + c], s);
        dprev[b * n.wmax + i] = s * (in[e] > 0.0f ? 1.0f : PRIOR_LEAK);
      }
    }
    __syncthreads();
    for (int e = tid; e < din * dout + dout; e += blockDim.x) {
      float g = 0.0f;
      if (e < din * dout) {
        const int i = e / dout, c = e - i * dout;
        if (l == 0) { for (int b = 0; b < B; ++b) g += (sg[b] == i) ? dlt[b * n.wmax + c] : 0.0f; }
        else for (int b = 0; b < B; ++b) g = fmaf(in[b * din + i], dlt[b * n.wmax + c], g);
        adam(n.w_off[l] + e, g);
      } else {
        const int c = e - din * dout;
        for (int b = 0; b < B; ++b) g += dlt[b * n.wmax + c];
        adam(n.b_off[l] + c, g);
      }
    }
    __syncthreads();
    float *t = dlt; dlt = dprev; dprev = t;
  }
}


// the Adam step of prior_step_kernel from a gradient buffer (data-parallel fit: after the all-reduce)
static __global__ void prior_adam_kernel(float *theta, float *m, float *v, const float *grad, int n, float lr_t, float b1, float b2, float eps) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float g = grad[p];
  const float m_ = b1 * m[p] + (1.0f - b1) * g, v_ = b2 * v[p] + (1.0f - b2) * g * g;
  m[p] = m_; v[p] = v_;
  theta[p] -= lr_t * m_ / (sqrtf(v_) + eps);
}
