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
// dense ops.  Each dense op is a handful of 16x16 fp32-MFMA tiles dealt to the 16 waves, operands read
// straight from the L1/L2-resident parameter arrays and activation workspace (see egm_gemm_tiles).
// All gradient formulas are the hand-derived ones of oracle/egm.py (checked there against autograd);
// the gradient penalty needs reverse mode through the backward pass of a batch-normalised network.
#pragma once
#include <hip/hip_runtime.h>

#include "bgm_device.h"

#define EGM_THREADS 512
#define EGM_MAX_LAYERS 8
#define EGM_LEAK 0.2f
#define EGM_BN_EPS 1e-3f

// Prefix tables (woff, aoff, csum, dsum) are filled on the host (egm_finish_mlp / egm_finish_disc): computing
// them in the kernel meant a loop of dependent scalar loads from the kernel-argument segment at the head of every
// phase (~2-4k cycles of a ~15k-cycle phase).
struct EgmMlp {            // BaseFullyConnectedNet: parameters at theta + off as W0,b0,W1,b1,... (Keras order)
  int n_layers;            // Dense layers (hidden + output)
  int dims[EGM_MAX_LAYERS + 1];
  int off;
  int woff[EGM_MAX_LAYERS];       // offset of W_l in theta (b_l follows it)
  int aoff[EGM_MAX_LAYERS + 2];   // activation of layer l (l >= 1) at base + B * aoff[l]; aoff[L+1] = total width
};
struct EgmDisc {           // Discriminator: hidden layers with BatchNorm + tanh, linear scalar output
  int n_hidden;
  int dims[EGM_MAX_LAYERS + 1];          // in, h1..hL, 1
  int w[EGM_MAX_LAYERS], b[EGM_MAX_LAYERS], gamma[EGM_MAX_LAYERS], beta[EGM_MAX_LAYERS];   // offsets into theta_d
  int n_params;
  int fixed_norm;                  // 0: BatchNormalization on batch statistics; 1: inference mode on the initial moving averages (mean 0, variance 1)
  int csum[EGM_MAX_LAYERS + 1];   // dims[1] + ... + dims[l]             (cache block of layer l at (2B+1) * csum[l])
  int dsum[EGM_MAX_LAYERS + 2];   // dims[0] + ... + dims[l-1]           (gradient-penalty da_l at B * dsum[l])
};
inline void egm_finish_mlp(EgmMlp &m) {
  int o = m.off, a = 0;
  m.aoff[0] = 0; m.aoff[1] = 0;
  for (int l = 0; l < m.n_layers; ++l) {
    m.woff[l] = o;
    o += m.dims[l] * m.dims[l + 1] + m.dims[l + 1];
    a += m.dims[l + 1];
    m.aoff[l + 2] = a;
  }
}
inline void egm_finish_disc(EgmDisc &d) {
  int c = 0, s_ = 0;
  d.csum[0] = 0; d.dsum[0] = 0;
  for (int l = 0; l <= d.n_hidden; ++l) { s_ += d.dims[l]; d.dsum[l + 1] = s_; }
  for (int l = 0; l < d.n_hidden; ++l) { c += d.dims[l + 1]; d.csum[l + 1] = c; }
}
struct EgmAdam { float lr_t, b1, b2, eps; };

struct EgmArgs {
  EgmMlp g, e, f, h;
  EgmDisc dz;
  float *theta_g, *m_g, *v_g, *grad_g;   // generator-side parameters [g|e|f|h], Adam slots, gradient scratch
  float *theta_d, *m_d, *v_d, *grad_d;   // discriminator parameters
  int n_gen, B, q, p;
  int wmax;                              // widest layer of any network (scratch row width)
  int dmax;                              // widest discriminator layer incl. its input
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
  int disc_lds;                          // 1: the discriminator working set of the step lives in LDS (it fits)
#ifdef EGM_PHASE_CLOCK
  unsigned long long *stamps;
#endif
};

struct EgmCtx {
  int tid;
  float *red;      // [32] reduction scratch (LDS)
#ifdef EGM_PHASE_CLOCK
  unsigned long long *stamps;
#endif
};
// Development aid (-DEGM_PHASE_CLOCK, never in the shipped library): thread 0 records (source line, shader clock) after every
// workgroup barrier of a step; egm_api.hip prints the per-phase cycle counts of one step.
#ifdef EGM_PHASE_CLOCK
#define EGM_STAMP(c) do { if ((c).tid == 0) { int *n_ = reinterpret_cast<int *>((c).red + 48); (c).stamps[2 * *n_] = __LINE__; (c).stamps[2 * *n_ + 1] = clock64(); ++*n_; } } while (0)
#else
#define EGM_STAMP(c) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------
// Tiny GEMMs on the matrix pipe.  C[M x N] = A[M x K] B[K x N] with arbitrary element strides, one
// 16x16 output tile per wave and round, K in steps of 4 on v_mfma_f32_16x16x4_f32 (exact fp32 fma
// chain).  Lane (j = lane & 15, g = lane >> 4) feeds A(m0 + j, 4s + g) and B(4s + g, n0 + j) and
// receives C(m0 + 4g + r, n0 + j) in accumulator register r.  Operands come straight from the
// L1/L2-resident workspace and parameter arrays: two loads per 1024 multiply-adds.  (The first,
// scalar version of these kernels issued two loads per multiply-add and was bound by the vector
// memory pipe of its single CU: 1.2 ms per generator step.)
// ---------------------------------------------------------------------------------------------
struct EgmMat { const float *p; int s0, s1; };   // element (i, j) at p[i * s0 + j * s1]

template <class Epi>
__device__ __forceinline__ void egm_gemm_tiles(const EgmCtx &c, EgmMat A, EgmMat Bm, int M, int N, int K, int tile_begin,
                                               int tile_stride, Epi epi) {
  const int lane = c.tid & 63, j = lane & 15, g = lane >> 4;
  const int tn_count = (N + 15) >> 4, tiles = ((M + 15) >> 4) * tn_count;
  const bool vec = A.s1 == 1 && (A.s0 & 3) == 0 && ((unsigned long long)A.p & 15ull) == 0;
  for (int t = tile_begin; t < tiles; t += tile_stride) {
    const int m0 = (t / tn_count) << 4, n0 = (t % tn_count) << 4;
    // Loads are unconditional (indices clamped into the matrix, contributions masked to zero).
    const float am = (m0 + j < M) ? 1.0f : 0.0f, bn = (n0 + j < N) ? 1.0f : 0.0f;
    const float *ap = A.p + (long long)min(m0 + j, M - 1) * A.s0, *bp = Bm.p + (long long)min(n0 + j, N - 1) * Bm.s1;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    // K in chunks of 16 steps: all 32 operand loads of a chunk are in flight before its first MFMA.
    for (int k0 = 0; k0 < K; k0 += 64) {
      float av[16], bv[16];
      if (vec && k0 + 64 <= K) {
        // row-major A (activation rows): a lane's four K values of a group of four steps are adjacent -> one 16-byte load per
        // group instead of four scalar loads that each touch 16 cache lines; step u then uses
        // k = k0 + 16 (u / 4) + 4 g + u % 4 on BOTH operands (same set of k, another summation order).  See DESIGN_HISTORY.md 7b.
        const f32x4 *q = reinterpret_cast<const f32x4 *>(ap + k0 + 4 * g);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const f32x4 v = q[4 * c4];
#pragma unroll
          for (int r = 0; r < 4; ++r) av[4 * c4 + r] = v[r] * am;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) bv[u] = bp[(long long)(k0 + 16 * (u >> 2) + 4 * g + (u & 3)) * Bm.s0];
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int k = k0 + 4 * u + g;
          const int kc = min(k, K - 1);
          av[u] = ap[kc * A.s1] * ((k < K) ? am : 0.0f);
          bv[u] = bp[kc * Bm.s0];
        }
      }
      BGM_NO_HOIST();
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = BGM_MFMA(av[u], bv[u] * bn, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * g + r, n = n0 + j;
      if (m < M && n < N) epi(m, n, acc[r]);
    }
  }
}

// Y = X W + bias  (optionally LeakyReLU)
__device__ __forceinline__ void egm_fwd(const EgmCtx &c, const float *X, int ldx, const float *W, const float *bias, float *Y,
                                        int ldy, int B, int in, int out, bool act) {
  egm_gemm_tiles(c, EgmMat{X, ldx, 1}, EgmMat{W, out, 1}, B, out, in, c.tid >> 6, EGM_THREADS / 64,
                 [&](int m, int n, float v) {
                   v += bias ? bias[n] : 0.0f;
                   if (act) v = fmaxf(v, EGM_LEAK * v);
                   Y[(long long)m * ldy + n] = v;
                 });
  __syncthreads();
  EGM_STAMP(c);
}
// dX (+)= dY W^T
__device__ __forceinline__ void egm_bwd_in(const EgmCtx &c, const float *dY, int ldy, const float *W, float *dX, int ldx, int B,
                                           int in, int out, bool accumulate) {
  egm_gemm_tiles(c, EgmMat{dY, ldy, 1}, EgmMat{W, 1, out}, B, in, out, c.tid >> 6, EGM_THREADS / 64,
                 [&](int m, int n, float v) {
                   float *dst = dX + (long long)m * ldx + n;
                   *dst = accumulate ? *dst + v : v;
                 });
  __syncthreads();
  EGM_STAMP(c);
}
__device__ __forceinline__ void egm_colsum(const EgmCtx &c, const float *dY, int ldy, float *gb, int B, int out, bool accumulate,
                                           float s) {
  for (int o = c.tid; o < out; o += EGM_THREADS) {
    float acc = 0.0f;
    for (int b = 0; b < B; ++b) acc += dY[(long long)b * ldy + o];
    gb[o] = accumulate ? gb[o] + s * acc : s * acc;
  }
}
// gW (+)= s * X^T dY ;  gb (+)= s * column sums of dY
__device__ __forceinline__ void egm_bwd_w(const EgmCtx &c, const float *X, int ldx, const float *dY, int ldy, float *gW, float *gb,
                                          int B, int in, int out, bool accumulate, float s = 1.0f) {
  egm_gemm_tiles(c, EgmMat{X, 1, ldx}, EgmMat{dY, ldy, 1}, in, out, B, c.tid >> 6, EGM_THREADS / 64,
                 [&](int m, int n, float v) {
                   float *dst = gW + (long long)m * out + n;
                   *dst = accumulate ? *dst + s * v : s * v;
                 });
  if (gb) egm_colsum(c, dY, ldy, gb, B, out, accumulate, s);
  __syncthreads();
  EGM_STAMP(c);
}
// One backward phase of a Dense layer: gW, gb (+)= X^T dY, sums of dY;  dX = (dY W^T) * lrelu'(X) when `mask`
// (X is the LeakyReLU output of the layer below, so dX is that layer's PRE-activation gradient), all behind one
// barrier: the weight-gradient tiles and the input-gradient tiles are dealt to the waves together.  dX may be NULL.
__device__ __forceinline__ void egm_bwd_layer(const EgmCtx &c, const float *X, const float *dY, const float *W, float *gW, float *gb,
                                              float *dX, int B, int in, int out, bool accumulate, bool mask,
                                              const float *add = nullptr) {
  const int nw = EGM_THREADS / 64, wave = c.tid >> 6;
  const int t_w = ((in + 15) >> 4) * ((out + 15) >> 4);
  egm_gemm_tiles(c, EgmMat{X, 1, in}, EgmMat{dY, out, 1}, in, out, B, wave, nw,
                 [&](int m, int n, float v) {
                   float *dst = gW + (long long)m * out + n;
                   *dst = accumulate ? *dst + v : v;
                 });
  if (dX) {
    const int first = (wave - t_w % nw + nw) % nw;      // continue the round-robin deal after the last dW tile
    egm_gemm_tiles(c, EgmMat{dY, out, 1}, EgmMat{W, 1, out}, B, in, out, first, nw,
                   [&](int m, int n, float v) {
                     const long long t = (long long)m * in + n;
                     if (add) v += add[t];     // a second consumer of X (the variance head of a variational net)
                     if (mask) v *= (X[t] > 0.0f) ? 1.0f : EGM_LEAK;
                     dX[t] = v;
                   });
  }
  egm_colsum(c, dY, out, gb, B, out, accumulate, 1.0f);
  __syncthreads();
  EGM_STAMP(c);
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
// Activation cache of one MLP call: layer l's output [B x dims[l]] (l >= 1) lives at base + B * (dims[1] + ... +
// dims[l-1]); l = 0 is the caller's input.  Pointers are COMPUTED (scalar ALU on kernel-argument dims): arrays of
// pointers indexed by a runtime layer number live in scratch memory and put a private-memory round trip at the
// head of every phase.
struct EgmMlpCache { float *in, *base; };
__device__ __forceinline__ float *egm_act(const EgmMlp &n, const EgmMlpCache &a, int l, int B) {
  return l == 0 ? a.in : a.base + (long long)B * n.aoff[l];
}

__device__ __forceinline__ const float *egm_W(const float *theta, const EgmMlp &n, int l) { return theta + n.woff[l]; }
__device__ __forceinline__ void egm_mlp_fwd(const EgmCtx &c, const float *theta, const EgmMlp &n, const EgmMlpCache &a, int B) {
  for (int l = 0; l < n.n_layers; ++l) {
    const float *W = egm_W(theta, n, l);
    egm_fwd(c, egm_act(n, a, l, B), n.dims[l], W, W + n.dims[l] * n.dims[l + 1], egm_act(n, a, l + 1, B), n.dims[l + 1], B, n.dims[l], n.dims[l + 1],
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
    const float *W = egm_W(theta, n, l);
    float *gW = grad + (W - theta);
    // act[l] (l > 0) is a LeakyReLU output: its sign is the sign of the pre-activation
    egm_bwd_layer(c, egm_act(n, a, l, B), cur, W, gW, gW + in * out, l > 0 ? nxt : dx, B, in, out, accumulate, l > 0);
    float *t = cur; cur = nxt; nxt = t;
  }
}

// ---------------------------------------------------------------------------------------------
// Discriminator: forward with batch statistics, backward, gradient penalty (double backward)
// ---------------------------------------------------------------------------------------------
// Forward cache of one discriminator call; computed pointers as above.  Per hidden layer l the block
// [a_{l+1} (B x w) | uhat_l (B x w) | sigma_l (w)], w = dims[l+1]; the scalar outputs [B] follow the last block.
struct EgmDiscCache { float *in, *base; };
__device__ __forceinline__ int egm_dk_off(const EgmDisc &d, int l, int B) { return (2 * B + 1) * d.csum[l]; }
__device__ __forceinline__ float *egm_dk_a(const EgmDisc &d, const EgmDiscCache &k, int l, int B) {   // a_l, l = 0..L
  return l == 0 ? k.in : k.base + egm_dk_off(d, l - 1, B);
}
__device__ __forceinline__ float *egm_dk_uhat(const EgmDisc &d, const EgmDiscCache &k, int l, int B) {
  return k.base + egm_dk_off(d, l, B) + B * d.dims[l + 1];
}
__device__ __forceinline__ float *egm_dk_sigma(const EgmDisc &d, const EgmDiscCache &k, int l, int B) {
  return k.base + egm_dk_off(d, l, B) + 2 * B * d.dims[l + 1];
}
__device__ __forceinline__ float *egm_dk_out(const EgmDisc &d, const EgmDiscCache &k, int B) {
  return k.base + egm_dk_off(d, d.n_hidden, B);
}

// Per-column passes of the discriminator (BatchNormalization statistics, their backward projections, gamma / beta gradients)
// are reductions over the B rows of a column.  A wave handles two columns at a time, its 32-lane halves the rows of one column
// each (rows beyond 32 in further rounds), and a column sum is five cross-lane adds -- instead of one thread walking the B rows
// of its column in a chain of dependent LDS accesses (the first version: ~9 k cycles per pass, ~90 passes per discriminator step).
// The strided row accesses of a column are bank-conflicted (stride = layer width), which costs tens of cycles, not thousands.
__device__ __forceinline__ float egm_sum32(float v) {          // sum over the 32 lanes of this half-wave, result in all of them
  v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
  return v;
}
// f(o, valid column, row lane bl in [0, 32)) for every column o of a layer of width `out`; all 64 lanes of a wave call f together
template <class F>
__device__ __forceinline__ void egm_for_cols(const EgmCtx &c, int out, F f) {
  const int lane = c.tid & 63, w = c.tid >> 6;
  for (int o0 = 2 * w; o0 < out; o0 += 2 * (EGM_THREADS / 64)) {
    const int o = o0 + (lane >> 5);
    f(o < out ? o : out - 1, o < out, lane & 31);
  }
}

__device__ __forceinline__ void egm_disc_fwd(const EgmCtx &c, const float *th, const EgmDisc &d, const EgmDiscCache &k, int B) {
  const int L = d.n_hidden;
  for (int l = 0; l < L; ++l) {
    const int in = d.dims[l], out = d.dims[l + 1];
    float *uh_ = egm_dk_uhat(d, k, l, B), *sg_ = egm_dk_sigma(d, k, l, B), *ao_ = egm_dk_a(d, k, l + 1, B);
    egm_fwd(c, egm_dk_a(d, k, l, B), in, th + d.w[l], th + d.b[l], uh_, out, B, in, out, false);   // u (normalised in place below)
    egm_for_cols(c, out, [&](int o, bool ok, int bl) {
      float mu = 0.0f, var = 1.0f;
      if (!d.fixed_norm) {
        float s1 = 0.0f;
        for (int b = bl; b < B; b += 32) s1 += uh_[b * out + o];
        mu = egm_sum32(s1) / (float)B;
        float s2 = 0.0f;
        for (int b = bl; b < B; b += 32) { const float t = uh_[b * out + o] - mu; s2 = fmaf(t, t, s2); }
        var = egm_sum32(s2) / (float)B;
      }
      const float sg = sqrtf(var + EGM_BN_EPS);
      const float ga = th[d.gamma[l] + o], be = th[d.beta[l] + o];
      if (ok) {
        if (bl == 0) sg_[o] = sg;
        for (int b = bl; b < B; b += 32) {
          const float uh = (uh_[b * out + o] - mu) / sg;
          uh_[b * out + o] = uh;
          ao_[b * out + o] = tanhf(fmaf(uh, ga, be));
        }
      }
    });
    __syncthreads();
    EGM_STAMP(c);
  }
  egm_fwd(c, egm_dk_a(d, k, L, B), d.dims[L], th + d.w[L], th + d.b[L], egm_dk_out(d, k, B), 1, B, d.dims[L], 1, false);
}

// x - mean_b x - uhat * mean_b(x uhat), per feature column o, by the 32 row lanes of a half-wave (see egm_for_cols)
// (fixed statistics: the normalisation is a constant per-column scale, nothing flows through mean / variance)
__device__ __forceinline__ void egm_bn_proj_col(float *x, const float *uhat, int B, int out, int o, bool ok, int bl, float scale,
                                                int fixed_norm) {
  float m1 = 0.0f, m2 = 0.0f;
  if (!fixed_norm) {
    for (int b = bl; b < B; b += 32) { m1 += x[b * out + o]; m2 = fmaf(x[b * out + o], uhat[b * out + o], m2); }
    m1 = egm_sum32(m1) / (float)B; m2 = egm_sum32(m2) / (float)B;
  }
  if (ok)
    for (int b = bl; b < B; b += 32) x[b * out + o] = (x[b * out + o] - m1 - uhat[b * out + o] * m2) * scale;
}

// Scratch of the gradient-penalty pass (computed pointers).  First the adjoint-network activations da_l [B x dims[l]],
// l = 0..L; then per hidden layer l the block [dy | dhat | du | a_bar | uhat_bar (B x w each) | sigma_bar (w)].
struct EgmGpScr { float *base; };
__device__ __forceinline__ float *egm_gp_da(const EgmDisc &d, const EgmGpScr &g, int l, int B) { return g.base + B * d.dsum[l]; }
__device__ __forceinline__ float *egm_gp_blk(const EgmDisc &d, const EgmGpScr &g, int l, int B, int which) {
  return g.base + B * d.dsum[d.n_hidden + 1] + (5 * B + 1) * d.csum[l] + which * B * d.dims[l + 1];
}
__device__ __forceinline__ int egm_gp_floats(const EgmDisc &d, int B) {
  return (B * d.dsum[d.n_hidden + 1] + (5 * B + 1) * d.csum[d.n_hidden] + 3) & ~3;
}
enum { GP_DY = 0, GP_DHAT = 1, GP_DU = 2, GP_ABAR = 3, GP_UBAR = 4, GP_SBAR = 5 };

// Ordinary backward.  dout_val: dLoss/dout of every row (has_dout), grads accumulate (scaled by s) when `accumulate`.
// adj (may be NULL): extra adjoints a_bar / uhat_bar / sigma_bar on the forward nodes (gradient-penalty reverse pass).
// da / du: scratch [B x max width].  dx (may be NULL) receives dLoss/dinput [B x dims[0]].
__device__ __forceinline__ void egm_disc_bwd(const EgmCtx &c, const float *th, float *gr, const EgmDisc &d, const EgmDiscCache &k,
                                             bool has_dout, float dout_val, const EgmGpScr *adj, float *da, float *du, float *dx,
                                             int B, bool accumulate, float s, const float *dout_vec = nullptr) {
  const int L = d.n_hidden, nL = d.dims[L];
  const float *aL = egm_dk_a(d, k, L, B);
  if (has_dout) {   // dLoss/dout_b = dout_vec[b] (LSGAN) or the same dout_val for every row (Wasserstein)
    for (int i = c.tid; i < nL + 1; i += EGM_THREADS) {
      float acc = 0.0f;
      if (i < nL) { for (int b = 0; b < B; ++b) acc = fmaf(dout_vec ? dout_vec[b] : dout_val, aL[b * nL + i], acc); }
      else { for (int b = 0; b < B; ++b) acc += dout_vec ? dout_vec[b] : dout_val; }
      float *dst = gr + (i < nL ? d.w[L] + i : d.b[L]);
      *dst = accumulate ? *dst + s * acc : s * acc;
    }
    for (int t = c.tid; t < B * nL; t += EGM_THREADS) da[t] = (dout_vec ? dout_vec[t / nL] : dout_val) * th[d.w[L] + (t % nL)];
  } else {
    for (int t = c.tid; t < B * nL; t += EGM_THREADS) da[t] = 0.0f;
    if (!accumulate)
      for (int i = c.tid; i < nL + 1; i += EGM_THREADS) gr[i < nL ? d.w[L] + i : d.b[L]] = 0.0f;
  }
  __syncthreads();
  EGM_STAMP(c);
  for (int l = L - 1; l >= 0; --l) {
    const int in = d.dims[l], out = d.dims[l + 1];
    const float *uh_ = egm_dk_uhat(d, k, l, B), *sg_ = egm_dk_sigma(d, k, l, B), *ao_ = egm_dk_a(d, k, l + 1, B);
    const float *abar = adj ? egm_gp_blk(d, *adj, l, B, GP_ABAR) : nullptr;
    const float *ubar = adj ? egm_gp_blk(d, *adj, l, B, GP_UBAR) : nullptr;
    const float *sbar = adj ? egm_gp_blk(d, *adj, l, B, GP_SBAR) : nullptr;
    // dy = (da + a_bar) (1 - a^2);  dgamma, dbeta;  duhat = dy gamma + uhat_bar;  du = proj(duhat)/sigma + sigma_bar uhat / B
    egm_for_cols(c, out, [&](int o, bool ok, int bl) {
      float gg = 0.0f, gb = 0.0f;
      const float ga = th[d.gamma[l] + o];
      for (int b = bl; b < B; b += 32) {
        const int t = b * out + o;
        float dav = da[t];
        if (abar) dav += abar[t];
        const float av = ao_[t];
        const float dy = dav * (1.0f - av * av);
        gg = fmaf(dy, uh_[t], gg);
        gb += dy;
        float dh = dy * ga;
        if (ubar) dh += ubar[t];
        if (ok) du[t] = dh;
      }
      gg = egm_sum32(gg); gb = egm_sum32(gb);
      if (ok && bl == 0) {
        float *pg = gr + d.gamma[l] + o, *pb = gr + d.beta[l] + o;
        *pg = accumulate ? *pg + s * gg : s * gg;
        *pb = accumulate ? *pb + s * gb : s * gb;
      }
      egm_bn_proj_col(du, uh_, B, out, o, ok, bl, 1.0f / sg_[o], d.fixed_norm);
      if (sbar && !d.fixed_norm && ok) {
        const float sb = sbar[o] / (float)B;
        for (int b = bl; b < B; b += 32) du[b * out + o] = fmaf(sb, uh_[b * out + o], du[b * out + o]);
      }
    });
    __syncthreads();
    EGM_STAMP(c);
    const float *ai_ = egm_dk_a(d, k, l, B);
    egm_bwd_w(c, ai_, in, du, out, gr + d.w[l], gr + d.b[l], B, in, out, accumulate, s);
    if (l > 0) egm_bwd_in(c, du, out, th + d.w[l], da, in, B, in, out, false);
    else if (dx) egm_bwd_in(c, du, out, th + d.w[l], dx, in, B, in, out, false);
  }
}

// Gradient penalty GP = mean_b (||g_b|| - 1)^2 on the batch whose forward cache is k; accumulates s * dGP/dtheta
// into gr (which must already hold valid values) and returns GP.
__device__ __forceinline__ float egm_disc_gp(const EgmCtx &c, const float *th, float *gr, const EgmDisc &d, const EgmDiscCache &k,
                                             float *scr, int B, float s, int wmax) {
  const int L = d.n_hidden;
  EgmGpScr G{scr};
  float *rest = scr + egm_gp_floats(d, B);
  float *bar_a = rest, *bar_b = rest + B * wmax, *tmp = rest + 2 * B * wmax;
  // ---- adjoint network: g = d(sum_b out_b)/d input
  const int nL = d.dims[L];
  {
    float *daL = egm_gp_da(d, G, L, B);
    for (int t = c.tid; t < B * nL; t += EGM_THREADS) daL[t] = th[d.w[L] + (t % nL)];
  }
  __syncthreads();
  EGM_STAMP(c);
  for (int l = L - 1; l >= 0; --l) {
    const int in = d.dims[l], out = d.dims[l + 1];
    const float *uh_ = egm_dk_uhat(d, k, l, B), *sg_ = egm_dk_sigma(d, k, l, B), *ao_ = egm_dk_a(d, k, l + 1, B);
    const float *da_up = egm_gp_da(d, G, l + 1, B);
    float *dy_ = egm_gp_blk(d, G, l, B, GP_DY), *dhat_ = egm_gp_blk(d, G, l, B, GP_DHAT), *du_ = egm_gp_blk(d, G, l, B, GP_DU);
    egm_for_cols(c, out, [&](int o, bool ok, int bl) {
      const float ga = th[d.gamma[l] + o];
      if (ok)
        for (int b = bl; b < B; b += 32) {
          const int t = b * out + o;
          const float av = ao_[t];
          const float dy = da_up[t] * (1.0f - av * av);
          dy_[t] = dy;
          dhat_[t] = dy * ga;
          du_[t] = dy * ga;
        }
      egm_bn_proj_col(du_, uh_, B, out, o, ok, bl, 1.0f / sg_[o], d.fixed_norm);
    });
    __syncthreads();
    EGM_STAMP(c);
    egm_bwd_in(c, du_, out, th + d.w[l], egm_gp_da(d, G, l, B), in, B, in, out, false);
  }
  // ---- penalty and its adjoint on g = da_0
  const int q = d.dims[0];
  const float *g0 = egm_gp_da(d, G, 0, B);
  float part = 0.0f;
  for (int b = c.tid; b < B; b += EGM_THREADS) {
    float n2 = 0.0f;
    for (int i = 0; i < q; ++i) n2 = fmaf(g0[b * q + i], g0[b * q + i], n2);
    const float nrm = sqrtf(n2);
    part += (nrm - 1.0f) * (nrm - 1.0f);
    const float coef = 2.0f * (nrm - 1.0f) / nrm / (float)B;
    for (int i = 0; i < q; ++i) bar_a[b * q + i] = coef * g0[b * q + i];
  }
  const float gp = egm_block_sum(c, part) / (float)B;
  // ---- reverse through the adjoint network, bottom (l = 0) to top
  float *da_bar = bar_a, *du_bar = bar_b;
  for (int l = 0; l < L; ++l) {
    const int in = d.dims[l], out = d.dims[l + 1];
    const float *uh_ = egm_dk_uhat(d, k, l, B), *sg_ = egm_dk_sigma(d, k, l, B), *ao_ = egm_dk_a(d, k, l + 1, B);
    const float *da_up = egm_gp_da(d, G, l + 1, B);
    const float *dy_ = egm_gp_blk(d, G, l, B, GP_DY), *dhat_ = egm_gp_blk(d, G, l, B, GP_DHAT), *du_ = egm_gp_blk(d, G, l, B, GP_DU);
    float *abar = egm_gp_blk(d, G, l, B, GP_ABAR), *ubar = egm_gp_blk(d, G, l, B, GP_UBAR), *sbar = egm_gp_blk(d, G, l, B, GP_SBAR);
    // da_{l-1} = du W^T :  W_bar += da_bar^T du ;  du_bar = da_bar W
    egm_bwd_w(c, da_bar, in, du_, out, gr + d.w[l], nullptr, B, in, out, true, s);
    egm_fwd(c, da_bar, in, th + d.w[l], nullptr, du_bar, out, B, in, out, false);
    // du = (dhat - m1 - uhat m2) / sigma
    egm_for_cols(c, out, [&](int o, bool ok, int bl) {
      const float sg = sg_[o], ga = th[d.gamma[l] + o];
      float sb = 0.0f, m2 = 0.0f, tu = 0.0f;
      for (int b = bl; b < B; b += 32) {
        const int t = b * out + o;
        sb = fmaf(du_bar[t], du_[t], sb);
        m2 = fmaf(dhat_[t], uh_[t], m2);
        tu = fmaf(du_bar[t] / sg, uh_[t], tu);
      }
      sb = egm_sum32(sb); m2 = egm_sum32(m2) / (float)B; tu = egm_sum32(tu) / (float)B;
      if (ok) {
        if (bl == 0) sbar[o] = d.fixed_norm ? 0.0f : -sb / sg;
        for (int b = bl; b < B; b += 32) {
          const int t = b * out + o;
          const float tt = du_bar[t] / sg;
          ubar[t] = d.fixed_norm ? 0.0f : -(tt * m2 + dhat_[t] * tu);
          tmp[t] = tt;
        }
      }
      egm_bn_proj_col(tmp, uh_, B, out, o, ok, bl, 1.0f, d.fixed_norm);          // dhat_bar
      float gg = 0.0f;
      for (int b = bl; b < B; b += 32) {
        const int t = b * out + o;
        gg = fmaf(tmp[t], dy_[t], gg);
        const float dyb = tmp[t] * ga;
        const float av = ao_[t];
        if (ok) {
          abar[t] = dyb * da_up[t] * (-2.0f * av);
          tmp[t] = dyb * (1.0f - av * av);                    // da_bar of the layer above
        }
      }
      gg = egm_sum32(gg);
      if (ok && bl == 0) gr[d.gamma[l] + o] += s * gg;
    });
    __syncthreads();
    EGM_STAMP(c);
    // tmp now holds da_bar for layer l+1 : move it into the ping-pong buffer
    for (int t = c.tid; t < B * out; t += EGM_THREADS) da_bar[t] = tmp[t];
    __syncthreads();
    EGM_STAMP(c);
  }
  for (int i = c.tid; i < nL; i += EGM_THREADS) {
    float acc = 0.0f;
    for (int b = 0; b < B; ++b) acc += da_bar[b * nL + i];
    gr[d.w[L] + i] += s * acc;
  }
  __syncthreads();
  EGM_STAMP(c);
  // ---- ... and on through the forward pass (scratch: bar_b, tmp)
  egm_disc_bwd(c, th, gr, d, k, false, 0.0f, &G, bar_b, tmp, nullptr, B, true, s);
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

// ---------------------------------------------------------------------------------------------
// data-parallel warm start: a step run with apply = 0 leaves its gradient in the session; the caller all-reduces it across ranks
// and the Adam step follows from the reduced buffer (bgm_causal_egm_grad / _apply, bgm_bnn_egm_grad / _apply)
// ---------------------------------------------------------------------------------------------
static __global__ void egm_dp_scale_kernel(const float *src, float *dst, int n, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] * scale;
}
// thetaT / mirror: the transposed copy the generator chains read (NULL when the session keeps none)
static __global__ void egm_dp_adam_kernel(float *theta, float *m, float *v, const float *g, int n, EgmAdam a, float *thetaT, const int *mirror) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = a.b1 * m[i] + (1.0f - a.b1) * gi;
  const float vi = a.b2 * v[i] + (1.0f - a.b2) * gi * gi;
  const float tn = theta[i] - a.lr_t * mi / (sqrtf(vi) + a.eps);
  m[i] = mi; v[i] = vi; theta[i] = tn;
  if (thetaT) { const int e = mirror[i]; if (e >= 0) thetaT[e] = tn; }
}

// carve a discriminator cache out of the workspace
__device__ __forceinline__ void egm_disc_cache(const EgmDisc &d, int B, float *&p, EgmDiscCache &k, float *input) {
  k.in = input;
  k.base = p;
  p += (egm_dk_off(d, d.n_hidden, B) + B + 3) & ~3;
}
__device__ __forceinline__ void egm_mlp_cache(const EgmMlp &n, int B, float *&p, EgmMlpCache &a, float *input) {
  auto take = [&](int k) { float *r = p; p += (k + 3) & ~3; return r; };
  a.in = input;
  a.base = take(B * n.aoff[n.n_layers + 1]);
}

// ---------------------------------------------------------------------------------------------
// train_disc_step
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(EGM_THREADS) void egm_disc_step_kernel(EgmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  EgmCtx c{(int)threadIdx.x, egm_lds};
#ifdef EGM_PHASE_CLOCK
  c.stamps = a.stamps;
  if (c.tid == 0) *reinterpret_cast<int *>(c.red + 48) = 0;
  EGM_STAMP(c);
#endif
  const int B = a.B, q = a.q, p = a.p;
  float *wp = a.ws;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  float *vb = take(B * p), *zhat = take(B * q);
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const int b = k / p; vb[k] = a.v[(long long)a.idx[b] * p + (k - b * p)]; }
  __syncthreads();
  EGM_STAMP(c);
  EgmMlpCache ce;
  egm_mlp_cache(a.e, B, wp, ce, vb);
  egm_mlp_fwd(c, a.theta_g, a.e, ce, B);                      // z_ = e(v)   (encoder fixed in this step)
  float *z_ = egm_act(a.e, ce, a.e.n_layers, B);
  for (int k = c.tid; k < B * q; k += EGM_THREADS) zhat[k] = a.z[k] * a.eps + z_[k] * (1.0f - a.eps);
  __syncthreads();
  EGM_STAMP(c);
  // Discriminator working set: [cache A | cache B, da, du  (later overlaid by the gradient-penalty scratch)].
  // In LDS when it fits (B = 32, dz_units [64, 32, 8]: 147 KB): its many small per-column passes are chains of
  // dependent accesses and an LDS access costs ~1/15 of an L2 round trip.  Cache A holds D(z_) first and
  // D(zhat) afterwards.
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
  EGM_STAMP(c);
  if (a.apply) egm_adam(c, a.theta_d, a.m_d, a.v_d, a.grad_d, a.dz.n_params, a.adam);
  if (c.tid == 0 && a.out) { a.out[0] = dz_loss; a.out[1] = dz_loss + 10.0f * gp; }
  EGM_STAMP(c);
}

// ---------------------------------------------------------------------------------------------
// train_gen_step
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(EGM_THREADS) void egm_gen_step_kernel(EgmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float egm_lds[];
  EgmCtx c{(int)threadIdx.x, egm_lds};
#ifdef EGM_PHASE_CLOCK
  c.stamps = a.stamps;
  if (c.tid == 0) *reinterpret_cast<int *>(c.red + 48) = 0;
  EGM_STAMP(c);
#endif
  const int B = a.B, q = a.q, p = a.p, z0 = a.z0, z1 = a.z1, z2 = a.z2;
  const float invB = 1.0f / (float)B;
  float *wp = a.ws;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  float *vb = take(B * p), *xb = take(B), *yb = take(B);
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const int b = k / p; vb[k] = a.v[(long long)a.idx[b] * p + (k - b * p)]; }
  for (int b = c.tid; b < B; b += EGM_THREADS) { xb[b] = a.x[a.idx[b]]; yb[b] = a.y[a.idx[b]]; }
  __syncthreads();
  EGM_STAMP(c);
  const int Lg = a.g.n_layers, Le = a.e.n_layers, Lf = a.f.n_layers, Lh = a.h.n_layers;
  const int wg = p + 1, nf = a.f.dims[0], nh = a.h.dims[0], of = a.f.dims[Lf], oh = a.h.dims[Lh];
  // ---- forward
  EgmMlpCache g1, e1, e2, g2, cf, ch;
  egm_mlp_cache(a.g, B, wp, g1, const_cast<float *>(a.z));
  egm_mlp_fwd(c, a.theta_g, a.g, g1, B);                       // g(z): v_ = [:, :p], sigma head [:, p]
  float *gz = egm_act(a.g, g1, Lg, B);
  egm_mlp_cache(a.e, B, wp, e1, vb);
  egm_mlp_fwd(c, a.theta_g, a.e, e1, B);                       // z_ = e(v)
  float *z_ = egm_act(a.e, e1, Le, B);
  float *v_ = take(B * p);                                     // contiguous copy of g(z)[:, :p]
  for (int k = c.tid; k < B * p; k += EGM_THREADS) { const int b = k / p; v_[k] = gz[b * wg + (k - b * p)]; }
  __syncthreads();
  EGM_STAMP(c);
  egm_mlp_cache(a.e, B, wp, e2, v_);
  egm_mlp_fwd(c, a.theta_g, a.e, e2, B);                       // z__ = e(v_)
  float *z__ = egm_act(a.e, e2, Le, B);
  egm_mlp_cache(a.g, B, wp, g2, z_);
  egm_mlp_fwd(c, a.theta_g, a.g, g2, B);                       // g(z_): v__ = [:, :p]
  float *gv = egm_act(a.g, g2, Lg, B);
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
  EGM_STAMP(c);
  egm_mlp_cache(a.f, B, wp, cf, fin);
  egm_mlp_fwd(c, a.theta_g, a.f, cf, B);
  egm_mlp_cache(a.h, B, wp, ch, hin);
  egm_mlp_fwd(c, a.theta_g, a.h, ch, B);
  float *fo = egm_act(a.f, cf, Lf, B), *ho = egm_act(a.h, ch, Lh, B);
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
    s_g += gz[b * wg + p] * gz[b * wg + p];
    s_f += fo[b * of + of - 1] * fo[b * of + of - 1];
    s_h += ho[b * oh + oh - 1] * ho[b * oh + oh - 1];
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
  float *d0 = take(B * wmax), *d1 = take(B * wmax), *dzsum = take(B * q), *dtmp = take(B * wmax);
  float *da = take(B * wmax), *du = take(B * wmax), *dfin = take(B * nf), *dhin = take(B * nh);
  // z__ branch: e (call 2, input v_) -> g (call 1)
  for (int k = c.tid; k < B * q; k += EGM_THREADS) d0[k] = zrec * (-2.0f / (float)(B * q)) * (a.z[k] - z__[k]);
  __syncthreads();
  EGM_STAMP(c);
  egm_mlp_bwd(c, a.theta_g, a.grad_g, a.e, e2, d0, d1, dtmp, B, false);                // dtmp = dLoss/dv_  [B x p]
  for (int k = c.tid; k < B * wg; k += EGM_THREADS) {
    const int b = k / wg, i = k - b * wg;
    d0[k] = (i < p) ? dtmp[b * p + i] : 0.001f * 2.0f * gz[k] * invB;
  }
  __syncthreads();
  EGM_STAMP(c);
  egm_mlp_bwd(c, a.theta_g, a.grad_g, a.g, g1, d0, d1, nullptr, B, false);
  // v__ branch: g (call 2, input z_)
  for (int k = c.tid; k < B * wg; k += EGM_THREADS) {
    const int b = k / wg, i = k - b * wg;
    d0[k] = (i < p) ? (-2.0f / (float)(B * p)) * (vb[b * p + i] - gv[k]) : 0.0f;
  }
  __syncthreads();
  EGM_STAMP(c);
  egm_mlp_bwd(c, a.theta_g, a.grad_g, a.g, g2, d0, d1, dzsum, B, true);                // dzsum = dLoss/dz_ (so far)
  // adversarial branch through the fixed discriminator (its gradients go to a scratch area past the workspace use)
  float *gd_scratch = take(a.dz.n_params);
  egm_disc_bwd(c, a.theta_d, gd_scratch, a.dz, kd, true, -invB, nullptr, da, du, dtmp, B, false, 1.0f);
  for (int k = c.tid; k < B * q; k += EGM_THREADS) dzsum[k] += dtmp[k];
  __syncthreads();
  EGM_STAMP(c);
  // f, h branches
  for (int k = c.tid; k < B * of; k += EGM_THREADS) {
    const int b = k / of, i = k - b * of;
    float t = 0.0f;
    if (i == 0) t += 2.0f * (fo[k] - yb[b]) * invB;
    if (i == of - 1) t += 0.001f * 2.0f * fo[k] * invB;
    d0[k] = t;
  }
  __syncthreads();
  EGM_STAMP(c);
  egm_mlp_bwd(c, a.theta_g, a.grad_g, a.f, cf, d0, d1, dfin, B, false);
  for (int k = c.tid; k < B * oh; k += EGM_THREADS) {
    const int b = k / oh, i = k - b * oh;
    float t = 0.0f;
    if (i == 0) t += a.binary ? (1.0f / (1.0f + expf(-ho[k])) - xb[b]) * invB : 2.0f * (ho[k] - xb[b]) * invB;
    if (i == oh - 1) t += 0.001f * 2.0f * ho[k] * invB;
    d0[k] = t;
  }
  __syncthreads();
  EGM_STAMP(c);
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
  EGM_STAMP(c);
  egm_mlp_bwd(c, a.theta_g, a.grad_g, a.e, e1, d0, d1, nullptr, B, true);
  if (a.apply) egm_adam(c, a.theta_g, a.m_g, a.v_g, a.grad_g, a.n_gen, a.adam);
  if (c.tid == 0 && a.out) {
    a.out[0] = adv; a.out[1] = l_v; a.out[2] = l_z; a.out[3] = l_x; a.out[4] = l_y;
    a.out[5] = adv + (l_v + zrec * l_z) + (l_x + l_y) + 0.001f * sig;
  }
  EGM_STAMP(c);
}
