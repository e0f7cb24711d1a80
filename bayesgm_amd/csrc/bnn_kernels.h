// bnn_kernels.h -- CausalBGM with Bayesian networks (use_bnn=True) on gfx950: minibatch steps.
//
// replaces (src/bayesgm/models/networks/bnn.py:4-38 BayesianFullyConnectedNet, i.e. input BatchNormalization on BATCH
// statistics + tfp.layers.DenseFlipout stack) inside
//   update_g_net / update_h_net / update_f_net   causalbgm/base.py:156-243   -> bnn_theta_step_kernel
//   update_latent_variable_sgd                   causalbgm/base.py:246-302   -> bnn_z_grad_kernel
// Semantics (TFP 0.18 DenseFlipout, Keras BatchNormalization) and the counter-based noise layout are restated in
// oracle/bnn.py, whose hand-derived gradients these kernels follow.
//
// Like the EGM steps these are B = 32 minibatch steps where latency, not throughput, is the cost: ONE launch of ONE
// 512-thread workgroup walks noise generation, forward, backward, KL and Adam with workgroup barriers between the
// phases.  A Flipout layer is two GEMMs over the same index space, y = h loc + ((h * s_in) dW) * s_out + b: both run
// in one pass of 16x16 fp32-MFMA tiles with two accumulators (bnn_gemm2), and so do the two halves of each
// backward product.  eps and dW = sigma * eps of a call are materialised once (they are needed again in backward).
#pragma once
#include "z_replay.h"
#include "fit_sync.h"
#include <hip/hip_runtime.h>

#include "bgm_device.h"

#define BNN_THREADS 512
#define BNN_MAX_LAYERS 8
#define BNN_LEAK 0.2f
#define BNN_BN_EPS 1e-3f
#define BNN_SCALE_EPS 1.1920928955078125e-07f
#define BNN_TAG_EPS 8u
#define BNN_TAG_SIGN 9u
#define BNN_MAX_BATCH 4096         // rows of a minibatch step (one workgroup per net walks them: a correctness path beyond a few hundred rows; the row-tile chains serve 16 / 32)

// One BayesianFullyConnectedNet.  Parameters (flat, at theta + off): gamma[in], beta[in], then per layer
// loc[in x out], rho[in x out], bias[out].  All prefix tables are filled on the host (bnn_finish_net).
struct BnnNet {
  int n_layers;                       // DenseFlipout layers (hidden + output; with `heads` the last two are the sibling heads)
  int dims[BNN_MAX_LAYERS + 1];
  int off, n_params, net_id;
  int bn_fixed;                       // 0: batch statistics; 1: mean 0 / variance 1; 2: the moving statistics (needs mv)
  int heads;                          // 1: BayesianVariationalNet (networks/bnn.py:40-99): layers L-2 (mean) and L-1 (var) both read
                                      //    the output of layer L-3; dims = [in, trunk..., p, p]
  int mv;                             // 1: moving mean[in], moving variance[in] stored after beta (they get a zero gradient)
  float prior_iv, prior_logs;         // kernel prior N(0, s^2): 1 / s^2 and log s
  int bias_prior;                     // 1: the bias has the same prior; KL(point mass || prior) = -log prior(bias)
  int lin[BNN_MAX_LAYERS], lout[BNN_MAX_LAYERS];   // fan-in / fan-out of layer l
  int hin[BNN_MAX_LAYERS];            // input of layer l at B * hin[l] (H) ...
  int hsin[BNN_MAX_LAYERS];           // ... and its sign-flipped copy at B * hsin[l] (HS)
  int hs_total;                       // HS floats per row
  int woff[BNN_MAX_LAYERS];           // loc of layer l (rho at + in*out, bias at + 2*in*out)
  int hoff[BNN_MAX_LAYERS + 2];       // output of layer l at B * hoff[l + 1]; hoff[L+1] = total width of H
  int eoff[BNN_MAX_LAYERS + 1];       // eps / dW of layer l inside one call's noise block; eoff[L] = kernel elements
  int sin_w[BNN_MAX_LAYERS], sout_w[BNN_MAX_LAYERS], swords;   // sign words per row (oracle/bnn.py sign_layout)
};
inline void bnn_finish_net(BnnNet &n) {
  const int L = n.n_layers;
  int o = n.off + (n.mv ? 4 : 2) * n.dims[0], h = 0, e = 0, w = 0;
  if (n.prior_iv == 0.0f) { n.prior_iv = 1.0f; n.prior_logs = 0.0f; }
  for (int l = 0; l < L; ++l) {
    const bool var_head = n.heads && l == L - 1;
    const int in = var_head ? n.dims[L - 2] : n.dims[l], out = n.dims[l + 1];
    n.lin[l] = in; n.lout[l] = out;
    n.woff[l] = o; o += 2 * in * out + out;
    n.hoff[l] = h; h += n.dims[l];
    n.hin[l] = var_head ? n.hoff[L - 2] : n.hoff[l];
    n.hsin[l] = n.hin[l];
    n.eoff[l] = e; e += in * out;
    n.sin_w[l] = w; w += (in + 31) / 32;
    n.sout_w[l] = w; w += (out + 31) / 32;
  }
  n.hoff[L] = h; n.hoff[L + 1] = h + n.dims[L];
  n.hs_total = n.hoff[L + 1];
  if (n.heads) { n.hsin[L - 1] = n.hs_total; n.hs_total += n.dims[L - 2]; }
  n.eoff[L] = e;
  n.swords = (w + 3) / 4 * 4;
  n.n_params = o - n.off;
}
// floats of one call cache (bnn_cache) for a batch of B rows
inline size_t bnn_cache_floats(const BnnNet &n, int B) {
  return (size_t)B * n.dims[0] + 3 * (size_t)n.dims[0] + (size_t)B * (n.hoff[n.n_layers + 1] + n.hs_total) + 2 * (size_t)n.eoff[n.n_layers] +
         (size_t)B * n.swords + 64;
}

struct BnnAdam { float lr_t, b1, b2, eps; };

struct BnnCtx { int tid; float *red; };

__device__ __forceinline__ float bnn_block_sum(const BnnCtx &c, float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((c.tid & 63) == 0) c.red[c.tid >> 6] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < BNN_THREADS / 64; ++w) t += c.red[w];
  return t;
}

// sign of (row, column) in the bit string that starts at word w0 of the row: +1 / -1
__device__ __forceinline__ float bnn_sign(const uint32_t *sg, int swords, int row, int w0, int col) {
  const uint32_t w = sg[(long long)row * swords + w0 + (col >> 5)];
  return ((w >> (col & 31)) & 1u) ? -1.0f : 1.0f;
}

// ---------------------------------------------------------------------------------------------
// Two GEMMs over one index space: C1 = A1 B1, C2 = A2 B2, [M x K] x [K x N], arbitrary element strides.  One 16x16
// output tile per wave and round, K in steps of 4 (v_mfma_f32_16x16x4_f32): lane (j = lane & 15, g = lane >> 4)
// feeds A(m0 + j, 4s + g), B(4s + g, n0 + j) and receives C(m0 + 4g + r, n0 + j) in register r.  K in chunks of 8
// steps with all 32 operand loads of a chunk in flight before its first MFMA.
// ---------------------------------------------------------------------------------------------
struct BnnMat { const float *p; int s0, s1; };   // element (i, j) at p[i * s0 + j * s1]

template <class Epi>
__device__ __forceinline__ void bnn_gemm2(int tid, BnnMat A1, BnnMat A2, BnnMat B1, BnnMat B2, int M, int N, int K,
                                          int tile_begin, int tile_stride, Epi epi) {
  const int lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int tn_count = (N + 15) >> 4, tiles = ((M + 15) >> 4) * tn_count;
  const bool vec = A1.s1 == 1 && (A1.s0 & 3) == 0 && (((unsigned long long)A1.p | (unsigned long long)A2.p) & 15ull) == 0;
  for (int t = tile_begin; t < tiles; t += tile_stride) {
    const int m0 = (t / tn_count) << 4, n0 = (t % tn_count) << 4;
    const float am = (m0 + j < M) ? 1.0f : 0.0f, bn = (n0 + j < N) ? 1.0f : 0.0f;
    const long long ao = (long long)min(m0 + j, M - 1) * A1.s0, bo = (long long)min(n0 + j, N - 1) * B1.s1;
    f32x4 c1 = {0.0f, 0.0f, 0.0f, 0.0f}, c2 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int k0 = 0; k0 < K; k0 += 32) {
      float a1[8], a2[8], b1[8], b2[8];
      if (vec && k0 + 32 <= K) {
        // row-major A: a lane's four K values of a group of four steps are adjacent -> one 16-byte load per operand and
        // group instead of four scalar loads that each touch 16 cache lines (the activation rows).  Step u then uses
        // k = k0 + 16 (u / 4) + 4 g + u % 4 on BOTH operands (the sum over k is over the same set, in another order).
        // (The same trick on the transposed weights of the input-gradient product, B with unit K stride, measured no gain:
        // those lines are shared by all workgroups and stay in L2.)
        const f32x4 *q1 = reinterpret_cast<const f32x4 *>(A1.p + ao + k0 + 4 * g), *q2 = reinterpret_cast<const f32x4 *>(A2.p + ao + k0 + 4 * g);
        const f32x4 v10 = q1[0], v11 = q1[4], v20 = q2[0], v21 = q2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a1[u] = v10[u] * am; a1[4 + u] = v11[u] * am; a2[u] = v20[u] * am; a2[4 + u] = v21[u] * am; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const long long kk = (long long)(k0 + 16 * (u >> 2) + 4 * g + (u & 3)) * B1.s0;
          b1[u] = B1.p[bo + kk];
          b2[u] = B2.p[bo + kk];
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 4 * u + g;
          const int kc = min(k, K - 1);
          const float mk = (k < K) ? am : 0.0f;
          a1[u] = A1.p[ao + (long long)kc * A1.s1] * mk;
          a2[u] = A2.p[ao + (long long)kc * A1.s1] * mk;
          b1[u] = B1.p[bo + (long long)kc * B1.s0];
          b2[u] = B2.p[bo + (long long)kc * B1.s0];
        }
      }
      BGM_NO_HOIST();
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c1 = BGM_MFMA(a1[u], b1[u] * bn, c1);
        c2 = BGM_MFMA(a2[u], b2[u] * bn, c2);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * g + r, n = n0 + j;
      if (m < M && n < N) epi(m, n, c1[r], c2[r]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same pair of products for minibatches of up to 32 rows when BOTH operands run along K in memory -- A [M x K] row-major and the
// weights as Wt [N x K] row-major (the input-gradient products of the backward pass: A = the upstream gradient, Wt = loc / dW as stored,
// [in][out]) -- with all eight waves on ONE tile row: the workgroup walks K in chunks of 16 through a double-buffered LDS stage
// (A1 | A2: 32 rows, W1 | W2: 256 columns of a pass, 36 KiB per half; every element fetched once per workgroup and pass), a wave owns a
// 32 x 32 block of the pass = 2 x 2 tiles for each product, fragments as ds_read_b128, the schedule of a chunk as in bnw_kernels.h
// (stage writes, fetches and fragment reads between the K steps, odd waves shifted).  16 chunks of a 256 x 256 layer instead of 4 tiles x 8
// chunks of a lone wave each.  rowinfo(m, n0) / eleminfo(m, n): what the epilogue loads, requested before the chunk loop;
// epi(m, n, c1, c2, row info, element info, bit = n - n0).
// ---------------------------------------------------------------------------------------------
#define BNN_R32_KC 16
#define BNN_R32_NP 256
#define BNN_R32_ROWS (2 * 32 + 2 * BNN_R32_NP)
#define BNN_R32_STAGE_FLOATS (BNN_R32_ROWS * BNN_R32_KC)      // one half of the stage
#define BNN_R32_LDS_BYTES (2 * BNN_R32_STAGE_FLOATS * sizeof(float))
struct BnnR32Frag { f32x4 a1[2], a2[2], b1[2], b2[2]; };
template <class RowInfo, class ElemInfo, class Epi>
__device__ __forceinline__ void bnn_gemm2_rows32(int tid, float *stage, const float *A1, const float *A2, int lda, const float *W1, const float *W2,
                                                 int ldw, int M, int N, int K, RowInfo rowinfo, ElemInfo eleminfo, Epi epi) {
  constexpr int KC = BNN_R32_KC, SLOTS = (BNN_R32_ROWS * 4 + BNN_THREADS - 1) / BNN_THREADS;
  const int lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6, nw = wave << 5;
  const bool late = (wave & 1) != 0;
  const bool vec = (lda & 3) == 0 && (ldw & 3) == 0 &&
                   ((((unsigned long long)A1 | (unsigned long long)A2 | (unsigned long long)W1 | (unsigned long long)W2) & 15ull) == 0);
  const int nc = (K + KC - 1) / KC;
  auto fetch = [&](const float *p, int k) -> f32x4 {
    if (vec && k + 3 < K) return *reinterpret_cast<const f32x4 *>(p + k);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k + e < K) ? p[min(k + e, K - 1)] : 0.0f;
    return v;
  };
  int ar[2], br[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { ar[i] = (16 * i + j) * KC + 4 * g; br[i] = (64 + nw + 16 * i + j) * KC + 4 * g; }
  auto frags = [&](BnnR32Frag &f, const float *h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f.a1[i] = *reinterpret_cast<const f32x4 *>(h + ar[i]); f.a2[i] = *reinterpret_cast<const f32x4 *>(h + 32 * KC + ar[i]);
      f.b1[i] = *reinterpret_cast<const f32x4 *>(h + br[i]); f.b2[i] = *reinterpret_cast<const f32x4 *>(h + BNN_R32_NP * KC + br[i]);
    }
  };
  const int fq = 4 * (tid & 3);
  for (int np = 0; np < N; np += BNN_R32_NP) {
    const float *fp[SLOTS];
    int so[SLOTS];      // stage offset of the slot (-1: none)
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      const int row = (tid + BNN_THREADS * sl) >> 2;           // stage row: A1 [0, 32) | A2 [32, 64) | W1 [64, 320) | W2 [320, 576)
      so[sl] = row < BNN_R32_ROWS ? row * KC + fq : -1;
      if (row < 32) fp[sl] = A1 + (long long)min(row, M - 1) * lda;
      else if (row < 64) fp[sl] = A2 + (long long)min(row - 32, M - 1) * lda;
      else if (row < 64 + BNN_R32_NP) fp[sl] = W1 + (long long)min(np + row - 64, N - 1) * ldw;
      else fp[sl] = W2 + (long long)min(np + min(row, BNN_R32_ROWS - 1) - 64 - BNN_R32_NP, N - 1) * ldw;
    }
    f32x4 c1[2][2], c2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) { c1[i][i2] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; c2[i][i2] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    const bool live = np + nw < N;
    f32x4 rg[SLOTS];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) if (so[sl] >= 0) rg[sl] = fetch(fp[sl], fq);
    decltype(rowinfo(0, 0)) rinfo[2][4];
    decltype(eleminfo(0, 0)) einfo[2][2][4];
    if (live) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = min(16 * i + 4 * g + r, M - 1);
          rinfo[i][r] = rowinfo(m, np + nw);
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) einfo[i][i2][r] = eleminfo(m, min(np + nw + 16 * i2 + j, N - 1));
        }
    }
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) if (so[sl] >= 0) *reinterpret_cast<f32x4 *>(stage + so[sl]) = rg[sl];
    if (nc > 1) {
#pragma unroll
      for (int sl = 0; sl < SLOTS; ++sl) if (so[sl] >= 0) rg[sl] = fetch(fp[sl], KC + fq);
    }
    __syncthreads();
    BnnR32Frag f0, f1;
    if (live) frags(f0, stage);
    auto step = [&](const BnnR32Frag &f, int u) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          c1[i][i2] = BGM_MFMA(f.a1[i][u], f.b1[i2][u], c1[i][i2]);
          c2[i][i2] = BGM_MFMA(f.a2[i][u], f.b2[i2][u], c2[i][i2]);
        }
    };
    auto chunk = [&](int c, const BnnR32Frag &cur, BnnR32Frag &nxt) {
      float *other = stage + ((c + 1) & 1) * BNN_R32_STAGE_FLOATS;
      auto put = [&]() {
        if (c + 1 < nc) {
#pragma unroll
          for (int sl = 0; sl < SLOTS; ++sl) if (so[sl] >= 0) *reinterpret_cast<f32x4 *>(other + so[sl]) = rg[sl];
          if (c + 2 < nc) {
            const int k = (c + 2) * KC + fq;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) if (so[sl] >= 0) rg[sl] = fetch(fp[sl], k);
          }
        }
      };
      if (late) put();
      BGM_NO_HOIST();
      if (live) step(cur, 0);
      BGM_NO_HOIST();
      if (!late) put();
      BGM_NO_HOIST();
      if (live) step(cur, 1);
      __syncthreads();
      if (!late && live && c + 1 < nc) frags(nxt, other);
      BGM_NO_HOIST();
      if (live) step(cur, 2);
      BGM_NO_HOIST();
      if (late && live && c + 1 < nc) frags(nxt, other);
      BGM_NO_HOIST();
      if (live) step(cur, 3);
    };
    int c = 0;
    for (; c + 2 <= nc; c += 2) { chunk(c, f0, f1); chunk(c + 1, f1, f0); }
    if (c < nc) chunk(c, f0, f1);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): no load is outstanding (keeps per-element waits out of the epilogue)
    if (live) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 16 * i + 4 * g + r, n = np + nw + 16 * i2 + j;
            if (m < M && n < N) epi(m, n, c1[i][i2][r], c2[i][i2][r], rinfo[i][r], einfo[i][i2][r], 16 * i2 + j);
          }
    }
    __syncthreads();      // (the stage is reused by the next pass / the next product)
  }
}

// ---------------------------------------------------------------------------------------------
// One call of a net on a batch of B rows: cache, noise, forward, backward
// ---------------------------------------------------------------------------------------------
struct BnnCache {
  const float *x;     // [B x in] raw input of the call
  float *xhat, *inv;  // normalised input [B x in], 1 / sqrt(var + eps) [in]
  float *mu;          // nets with moving statistics: batch mean [in] | batch variance [in]
  float *H, *HS;      // h_l and h_l * s_in(l) at B * hin[l] / B * hsin[l]
  float *eps, *dW;    // [kernel elements]
  uint32_t *sg;       // [B x swords]
  const float *ext = nullptr;   // given statistics of the input columns, mean [in] | variance [in] (bnw_kernels.h: the statistics of a BLOCK of rows the call's tile belongs to)
};
__device__ __forceinline__ void bnn_cache(const BnnNet &n, int B, float *&p, BnnCache &k, const float *input) {
  auto take = [&](long long cnt) { float *r = p; p += (cnt + 3) & ~3LL; return r; };
  k.x = input;
  k.xhat = take((long long)B * n.dims[0]);
  k.inv = take(n.dims[0]);
  k.mu = n.mv ? take(2 * n.dims[0]) : nullptr;     // batch mean | batch variance
  k.H = take((long long)B * n.hoff[n.n_layers + 1]);
  k.HS = take((long long)B * n.hs_total);
  k.eps = take(n.eoff[n.n_layers]);
  k.dW = take(n.eoff[n.n_layers]);
  k.sg = (uint32_t *)take((long long)B * n.swords);
}

// eps, dW = sigma * eps and the sign words of call `stream` (oracle/bnn.py draw_noise).  No barrier at the end.
// signs_only: the perturbation dW of the call was produced elsewhere (k.dW points at it); only the per-row sign words are drawn.
__device__ __forceinline__ void bnn_noise(const BnnCtx &c, const float *theta, const BnnNet &n, const BnnCache &k, int B,
                                          uint32_t k0, uint32_t k1, uint32_t stream, uint32_t row0 = 0u, bool signs_only = false) {
  if (!signs_only)
    for (int l = 0; l < n.n_layers; ++l) {
      const int cnt = n.lin[l] * n.lout[l];
      const float *rho = theta + n.woff[l] + cnt;
      float *e = k.eps + n.eoff[l], *d = k.dW + n.eoff[l];
      for (int i = c.tid; i < (cnt + 3) >> 2; i += BNN_THREADS) {
        const f32x4 z = box_muller4(philox4x32_10((uint32_t)i, (uint32_t)l | ((uint32_t)n.net_id << 16), stream, BNN_TAG_EPS, k0, k1));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = 4 * i + u;
          if (idx < cnt) { e[idx] = z[u]; d[idx] = (BNN_SCALE_EPS + softplus_acc(rho[idx])) * z[u]; }
        }
      }
    }
  const int calls = n.swords >> 2;
  for (int i = c.tid; i < B * calls; i += BNN_THREADS) {
    const int r = i / calls, cc = i - r * calls;
    const uint4 w = philox4x32_10(row0 + (uint32_t)r, (uint32_t)cc | ((uint32_t)n.net_id << 16), stream, BNN_TAG_SIGN, k0, k1);
    uint32_t *dst = k.sg + (long long)r * n.swords + 4 * cc;
    dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w;
  }
}

// BatchNormalization, h_0 = gamma * xhat + beta, and hs_0.  Needs the sign words (barrier before).
__device__ __forceinline__ void bnn_bn_fwd(const BnnCtx &c, const float *theta, const BnnNet &n, const BnnCache &k, int B) {
  const int in = n.dims[0];
  const float *gamma = theta + n.off, *beta = gamma + in, *mvs = beta + in;
  if (k.ext || n.bn_fixed) {      // statistics known before the call: one thread per element (rows x columns) instead of one per column
    for (int idx = c.tid; idx < B * in; idx += BNN_THREADS) {
      const int b = idx / in, i = idx - b * in;
      float mu, var;
      if (k.ext) { mu = k.ext[i]; var = k.ext[in + i]; }
      else if (n.bn_fixed == 1) { mu = 0.0f; var = 1.0f; }
      else { mu = mvs[i]; var = mvs[in + i]; }
      const float inv = 1.0f / sqrtf(var + BNN_BN_EPS);
      if (b == 0) {
        k.inv[i] = inv;
        if (k.mu) { k.mu[i] = mu; k.mu[in + i] = var; }
      }
      const float xh = (k.x[idx] - mu) * inv;
      const float h = xh * gamma[i] + beta[i];
      k.xhat[idx] = xh;
      k.H[idx] = h;
      k.HS[idx] = h * bnn_sign(k.sg, n.swords, b, n.sin_w[0], i);
    }
    return;
  }
  for (int i = c.tid; i < in; i += BNN_THREADS) {
    float mu, var;
    if (k.ext) { mu = k.ext[i]; var = k.ext[in + i]; }
    else if (n.bn_fixed == 1) { mu = 0.0f; var = 1.0f; }
    else if (n.bn_fixed == 2) { mu = mvs[i]; var = mvs[in + i]; }
    else {
      float s = 0.0f;
      for (int b = 0; b < B; ++b) s += k.x[(long long)b * in + i];
      mu = s / (float)B;
      float v = 0.0f;
      for (int b = 0; b < B; ++b) { const float d = k.x[(long long)b * in + i] - mu; v = fmaf(d, d, v); }
      var = v / (float)B;
    }
    const float inv = 1.0f / sqrtf(var + BNN_BN_EPS);
    k.inv[i] = inv;
    if (k.mu) { k.mu[i] = mu; k.mu[in + i] = var; }
    for (int b = 0; b < B; ++b) {
      const float xh = (k.x[(long long)b * in + i] - mu) * inv;
      const float h = xh * gamma[i] + beta[i];
      k.xhat[(long long)b * in + i] = xh;
      k.H[(long long)b * in + i] = h;
      k.HS[(long long)b * in + i] = h * bnn_sign(k.sg, n.swords, b, n.sin_w[0], i);
    }
  }
}

// moving averages of a training-mode call (momentum 0.99, biased batch variance; Keras BatchNormalization)
__device__ __forceinline__ void bnn_bn_move(const BnnCtx &c, float *theta, const BnnNet &n, const BnnCache &k) {
  const int in = n.dims[0];
  float *mvs = theta + n.off + 2 * in;
  for (int i = c.tid; i < in; i += BNN_THREADS) {
    mvs[i] = mvs[i] * 0.99f + k.mu[i] * (1.0f - 0.99f);
    mvs[in + i] = mvs[in + i] * 0.99f + k.mu[in + i] * (1.0f - 0.99f);
  }
}

// forward of the Flipout stack (after bnn_noise + barrier + bnn_bn_fwd + barrier)
// stage, locT, dWT (nets without heads, up to 32 rows): the LDS stage of bnn_gemm2_rows32 and the call's weights TRANSPOSED -- posterior means
// and perturbation, layer l at eoff[l] as [out][in] (bnn_step_noise_kernel writes both) -- so that both operands of the forward products run along K
__device__ __forceinline__ void bnn_layers_fwd(const BnnCtx &c, const float *theta, const BnnNet &n, const BnnCache &k, int B, float *stage = nullptr,
                                               const float *locT = nullptr, const float *dWT = nullptr) {
  const int L = n.n_layers;
  for (int l = 0; l < L; ++l) {
    const int in = n.lin[l], out = n.lout[l];
    const float *loc = theta + n.woff[l], *bias = loc + 2 * in * out;
    const float *h = k.H + (long long)B * n.hin[l], *hs = k.HS + (long long)B * n.hsin[l];
    float *y = k.H + (long long)B * n.hoff[l + 1], *ys = k.HS + (long long)B * n.hoff[l + 1];
    const bool last = (l == L - 1) || (n.heads && l == L - 2);
    const bool feeds_heads = n.heads && l == L - 3;
    float *ys2 = k.HS + (long long)B * n.hsin[L - 1];
    const int so = n.sout_w[l], si = last ? 0 : n.sin_w[l + 1], si2 = n.sin_w[L - 1];
    if (stage && locT && !n.heads && B <= 32) {
      struct Rw { uint32_t so, si; };
      bnn_gemm2_rows32(c.tid, stage, h, hs, in, locT + n.eoff[l], dWT + n.eoff[l], in, B, out, in,
                       [&](int m, int n0) {
                         const uint32_t *sg = k.sg + (long long)m * n.swords + (n0 >> 5);
                         Rw w; w.so = sg[so]; w.si = last ? 0u : sg[si];
                         return w;
                       },
                       [&](int, int o) { return bias[o]; },
                       [&](int m, int o, float c1, float c2, const Rw &w, float bo, int bit) {
                         float v = c1 + bo + (((w.so >> bit) & 1u) ? -c2 : c2);
                         if (!last) v = fmaxf(v, BNN_LEAK * v);
                         y[(long long)m * out + o] = v;
                         if (!last) ys[(long long)m * out + o] = ((w.si >> bit) & 1u) ? -v : v;
                       });
      continue;      // (bnn_gemm2_rows32 ends with a barrier)
    }
    bnn_gemm2(c.tid, BnnMat{h, in, 1}, BnnMat{hs, in, 1}, BnnMat{loc, out, 1}, BnnMat{k.dW + n.eoff[l], out, 1}, B, out, in,
              c.tid >> 6, BNN_THREADS / 64, [&](int m, int o, float c1, float c2) {
                float v = c1 + bias[o] + bnn_sign(k.sg, n.swords, m, so, o) * c2;
                if (!last) v = fmaxf(v, BNN_LEAK * v);
                y[(long long)m * out + o] = v;
                if (!last) ys[(long long)m * out + o] = v * bnn_sign(k.sg, n.swords, m, si, o);
                if (feeds_heads) ys2[(long long)m * out + o] = v * bnn_sign(k.sg, n.swords, m, si2, o);
              });
    if (!(n.heads && l == L - 2)) __syncthreads();     // the two heads are independent
  }
}

__device__ __forceinline__ float *bnn_fwd(const BnnCtx &c, const float *theta, const BnnNet &n, const BnnCache &k, int B,
                                          uint32_t k0, uint32_t k1, uint32_t stream, uint32_t row0 = 0u, bool signs_only = false, float *stage = nullptr,
                                          const float *locT = nullptr, const float *dWT = nullptr) {
  bnn_noise(c, theta, n, k, B, k0, k1, stream, row0, signs_only);
  __syncthreads();
  bnn_bn_fwd(c, theta, n, k, B);
  __syncthreads();
  bnn_layers_fwd(c, theta, n, k, B, stage, locT, dWT);
  return k.H + (long long)B * n.hoff[n.heads ? n.n_layers - 1 : n.n_layers];     // heads: mean [B x p], then var-raw [B x p]
}

// Parameter gradients of layer l from its input (h, hs) and the upstream gradient (cur, curs = cur * s_out).
// (wg, n_wg): this workgroup's share when the tiles of a layer are spread over n_wg workgroups (bnn_dw_kernel)
__device__ __forceinline__ void bnn_bwd_params(const BnnCtx &c, const float *theta, float *grad, const BnnNet &n, const BnnCache &k,
                                               int l, const float *cur, const float *curs, int B, bool accumulate, int wg = 0, int n_wg = 1) {
  const int in = n.lin[l], out = n.lout[l], nw = n_wg * (BNN_THREADS / 64), wave = wg * (BNN_THREADS / 64) + (c.tid >> 6);
  const float *rho = theta + n.woff[l] + in * out;
  const float *h = k.H + (long long)B * n.hin[l], *hs = k.HS + (long long)B * n.hsin[l];
  float *gloc = grad + n.woff[l], *grho = gloc + in * out, *gb = grho + in * out;
  const float *eps = k.eps + n.eoff[l];
  bnn_gemm2(c.tid, BnnMat{h, 1, in}, BnnMat{hs, 1, in}, BnnMat{cur, out, 1}, BnnMat{curs, out, 1}, in, out, B, wave, nw,
            [&](int i, int o, float c1, float c2) {
              const int t = i * out + o;
              const float r = c2 * eps[t] * sigmoid_f(rho[t]);
              gloc[t] = accumulate ? gloc[t] + c1 : c1;
              grho[t] = accumulate ? grho[t] + r : r;
            });
  for (int o = wg * BNN_THREADS + c.tid; o < out; o += n_wg * BNN_THREADS) {
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s += cur[(long long)b * out + o];
    gb[o] = accumulate ? gb[o] + s : s;
  }
}

// Backward of one call.  d: upstream gradient (destroyed) -- [B x out_L], or with heads [B x p] for the mean followed by
// [B x p] for the raw variance; ds, t0, t1: scratch of the same size (at least B x the widest layer);
// grad (layout of theta) receives / accumulates the parameter gradients when `want_params`; dx [B x in] (may be NULL)
// receives the gradient w.r.t. the raw input (through the batch statistics).
__device__ __forceinline__ void bnn_bwd(const BnnCtx &c, const float *theta, float *grad, const BnnNet &n, const BnnCache &k,
                                        float *d, float *ds, float *t0, float *t1, float *dx, int B, bool want_params,
                                        bool accumulate, float *G = nullptr, float *GS = nullptr, float *stage = nullptr) {
  // stage: LDS of BNN_R32_LDS_BYTES -- minibatches of up to 32 rows run the trunk's input-gradient products on bnn_gemm2_rows32
  // G, GS (want_params): the upstream gradients of EVERY layer are kept -- layer l's at B * hoff[l + 1] of G / GS, where
  // d / ds must point for the last layer -- and the parameter-gradient tiles are left to bnn_dw_kernel (all layers at once, over the chip)
  const bool defer = G && want_params;      // (with heads: d / ds at B * hoff[L - 1] of G / GS -- mean head's rows, then the variance head's)
  const int L = n.n_layers, nw = BNN_THREADS / 64, wave = c.tid >> 6;
  float *cur = d, *curs = ds, *nxt = t0, *nxts = t1;
  int l_top = L - 1;
  if (n.heads) {
    const int p = n.lout[L - 1], in = n.lin[L - 1];
    float *dv = d + (long long)B * p, *dvs = ds + (long long)B * p;
    for (int i = c.tid; i < B * p; i += BNN_THREADS) {
      ds[i] = d[i] * bnn_sign(k.sg, n.swords, i / p, n.sout_w[L - 2], i % p);
      dvs[i] = dv[i] * bnn_sign(k.sg, n.swords, i / p, n.sout_w[L - 1], i % p);
    }
    __syncthreads();
    if (want_params && !defer) {
      bnn_bwd_params(c, theta, grad, n, k, L - 2, d, ds, B, accumulate);
      bnn_bwd_params(c, theta, grad, n, k, L - 1, dv, dvs, B, accumulate);
    }
    if (defer) { t0 = G + (long long)B * n.hoff[L - 2]; t1 = GS + (long long)B * n.hoff[L - 2]; }      // the trunk output's gradient stays as well
    const float *h = k.H + (long long)B * n.hin[L - 1];
    {   // variance head -> t1 (raw), then mean head adds its share, applies the activation mask of the trunk output
      const float *loc = theta + n.woff[L - 1];
      const int si = n.sin_w[L - 1];
      bnn_gemm2(c.tid, BnnMat{dv, p, 1}, BnnMat{dvs, p, 1}, BnnMat{loc, 1, p}, BnnMat{k.dW + n.eoff[L - 1], 1, p}, B, in, p, wave, nw,
                [&](int m, int i, float c1, float c2) { t1[(long long)m * in + i] = c1 + bnn_sign(k.sg, n.swords, m, si, i) * c2; });
    }
    __syncthreads();
    {
      const float *loc = theta + n.woff[L - 2];
      const int si = n.sin_w[L - 2], sop = L >= 3 ? n.sout_w[L - 3] : 0;
      bnn_gemm2(c.tid, BnnMat{d, p, 1}, BnnMat{ds, p, 1}, BnnMat{loc, 1, p}, BnnMat{k.dW + n.eoff[L - 2], 1, p}, B, in, p, wave, nw,
                [&](int m, int i, float c1, float c2) {
                  const long long t = (long long)m * in + i;
                  float v = c1 + bnn_sign(k.sg, n.swords, m, si, i) * c2 + t1[t];
                  if (L >= 3) {
                    v *= (h[t] > 0.0f) ? 1.0f : BNN_LEAK;
                    t1[t] = v * bnn_sign(k.sg, n.swords, m, sop, i);
                  }
                  t0[t] = v;
                });
    }
    __syncthreads();
    cur = t0; curs = t1; nxt = d; nxts = ds;
    l_top = L - 3;
  } else {
    const int out = n.lout[L - 1];
    for (int i = c.tid; i < B * out; i += BNN_THREADS) ds[i] = d[i] * bnn_sign(k.sg, n.swords, i / out, n.sout_w[L - 1], i % out);
    __syncthreads();
  }
  for (int l = l_top; l >= 0; --l) {
    const int in = n.lin[l], out = n.lout[l];
    const float *loc = theta + n.woff[l];
    const float *h = k.H + (long long)B * n.hin[l];
    int first = wave;
    if (defer) { nxt = G + (long long)B * n.hoff[l]; nxts = GS + (long long)B * n.hoff[l]; }
    if (want_params && !defer) {
      bnn_bwd_params(c, theta, grad, n, k, l, cur, curs, B, accumulate);
      const int t_w = ((in + 15) >> 4) * ((out + 15) >> 4);
      first = (wave - t_w % nw + nw) % nw;
    }
    if (stage && B <= 32 && (l > 0 || dx || want_params)) {
      const int si = n.sin_w[l], sop = l > 0 ? n.sout_w[l - 1] : 0;
      struct Rw { uint32_t si, sop; };
      bnn_gemm2_rows32(c.tid, stage, cur, curs, out, loc, k.dW + n.eoff[l], out, B, in, out,
                       [&](int m, int n0) {
                         const uint32_t *sg = k.sg + (long long)m * n.swords + (n0 >> 5);
                         Rw w; w.si = sg[si]; w.sop = l > 0 ? sg[sop] : 0u;
                         return w;
                       },
                       [&](int m, int i) { return l > 0 ? h[(long long)m * in + i] : 1.0f; },
                       [&](int m, int i, float c1, float c2, const Rw &w, float hv, int bit) {
                         const long long t = (long long)m * in + i;
                         float v = c1 + (((w.si >> bit) & 1u) ? -c2 : c2);
                         if (l > 0) {
                           v *= (hv > 0.0f) ? 1.0f : BNN_LEAK;
                           nxts[t] = ((w.sop >> bit) & 1u) ? -v : v;
                         }
                         nxt[t] = v;
                       });
    } else if (l > 0 || dx || want_params) {
      const int si = n.sin_w[l], sop = l > 0 ? n.sout_w[l - 1] : 0;
      bnn_gemm2(c.tid, BnnMat{cur, out, 1}, BnnMat{curs, out, 1}, BnnMat{loc, 1, out}, BnnMat{k.dW + n.eoff[l], 1, out}, B, in, out,
                first, nw, [&](int m, int i, float c1, float c2) {
                  const long long t = (long long)m * in + i;
                  float v = c1 + bnn_sign(k.sg, n.swords, m, si, i) * c2;
                  if (l > 0) {
                    v *= (h[t] > 0.0f) ? 1.0f : BNN_LEAK;
                    nxts[t] = v * bnn_sign(k.sg, n.swords, m, sop, i);
                  }
                  nxt[t] = v;
                });
    }
    __syncthreads();
    float *t = cur; cur = nxt; nxt = t;
    t = curs; curs = nxts; nxts = t;
  }
  // cur = dLoss/dh_0 [B x in]: gamma, beta and the input gradient (through the batch statistics unless they are constants)
  const int in = n.dims[0];
  const float *gamma = theta + n.off;
  const bool fixed = n.bn_fixed != 0;
  for (int i = c.tid; i < in; i += BNN_THREADS) {
    float sg_ = 0.0f, sb = 0.0f;
    for (int b = 0; b < B; ++b) { const float v = cur[(long long)b * in + i]; sg_ = fmaf(v, k.xhat[(long long)b * in + i], sg_); sb += v; }
    if (want_params) {
      float *gg = grad + n.off, *gbt = gg + in;
      gg[i] = accumulate ? gg[i] + sg_ : sg_;
      gbt[i] = accumulate ? gbt[i] + sb : sb;
    }
    if (dx) {
      const float m1 = fixed ? 0.0f : sb * gamma[i] / (float)B, m2 = fixed ? 0.0f : sg_ * gamma[i] / (float)B, inv = k.inv[i];
      for (int b = 0; b < B; ++b) {
        const long long t = (long long)b * in + i;
        dx[t] = inv * (cur[t] * gamma[i] - m1 - k.xhat[t] * m2);
      }
    }
  }
  __syncthreads();
}

// grad += w * dKL/dtheta for every kernel (and, with bias_prior, every bias) of the net; returns sum(net.losses) (every thread).
__device__ __forceinline__ float bnn_kl(const BnnCtx &c, const float *theta, float *grad, const BnnNet &n, float w) {
  float acc = 0.0f;
  const float iv = n.prior_iv, ls = n.prior_logs;
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.lin[l] * n.lout[l];
    const float *loc = theta + n.woff[l], *rho = loc + cnt;
    float *gloc = grad + n.woff[l], *grho = gloc + cnt;
    for (int i = c.tid; i < cnt; i += BNN_THREADS) {
      const float sg = BNN_SCALE_EPS + softplus_acc(rho[i]), mu = loc[i];
      acc += -logf(sg) + 0.5f * (sg * sg + mu * mu) * iv - 0.5f + ls;
      gloc[i] += w * mu * iv;
      grho[i] += w * (-1.0f / sg + sg * iv) * sigmoid_f(rho[i]);
    }
    if (n.bias_prior) {
      const float *b = rho + cnt;
      float *gb = grho + cnt;
      for (int i = c.tid; i < n.lout[l]; i += BNN_THREADS) {
        acc += 0.5f * b[i] * b[i] * iv + ls + 0.9189385332046727f;
        gb[i] += w * b[i] * iv;
      }
    }
  }
  return bnn_block_sum(c, acc);
}

// one Adam update, roundings spelled out (every site that steps a parameter of the Bayesian nets goes through it: the results of the
// fused and the split forms of a step must agree to the bit whatever the surrounding code lets the compiler contract)
struct BnnAdamOut { float m, v, th; };
__device__ __forceinline__ BnnAdamOut bnn_adam_one(float th, float m, float v, float g, const BnnAdam &a) {
  BnnAdamOut o;
  o.m = fmaf(a.b1, m, __fmul_rn(1.0f - a.b1, g));
  o.v = fmaf(a.b2, v, __fmul_rn(__fmul_rn(1.0f - a.b2, g), g));
  o.th = __fsub_rn(th, __fdiv_rn(__fmul_rn(a.lr_t, o.m), __fadd_rn(sqrtf(o.v), a.eps)));
  return o;
}
__device__ __forceinline__ void bnn_adam(const BnnCtx &c, float *theta, float *m, float *v, const float *g, int n, const BnnAdam &a) {
  for (int i = c.tid; i < n; i += BNN_THREADS) {
    const BnnAdamOut o = bnn_adam_one(theta[i], m[i], v[i], g[i], a);
    m[i] = o.m; v[i] = o.v; theta[i] = o.th;
  }
}

// ---------------------------------------------------------------------------------------------
// step kernels
// ---------------------------------------------------------------------------------------------
enum { BNN_G = 0, BNN_E = 1, BNN_F = 2, BNN_H = 3 };

struct BnnArgs {
  BnnNet net[4];                         // g, e, f, h
  float *theta, *m, *v, *grad;           // flat [g | e | f | h] parameters, Adam slots, gradient
  int B, q, p, z0, z1, z2, binary, wmax;
  float kl_weight;
  const float *data_z;                   // [N x q] latent table (rows gathered through idx)
  const int *idx;                        // [B] rows of the panel
  const float *v_, *x_, *y_;             // panel [N x p], [N], [N]
  uint32_t k0, k1, stream;               // noise key and call id
  BnnAdam adam;
  int apply;                             // 1: Adam on g, h, f inside the kernel; 0: gradients stay in grad
  float inv_B;                           // 1 / global batch (data-parallel steps divide by the global batch)
  float *ws;                             // workspace (one slice of ws_stride floats per workgroup)
  long long ws_stride;
  float *dz_part, *loss_part;            // z step: per-net partials [3][B x q], [3] (summed by bnn_z_combine_kernel)
  float *out;                            // theta: [loss_v, mse_v, loss_x, aux_x, loss_y, mse_y];  z: [loss_posterior]
  float *dz;                             // z step: [B x q] gradient w.r.t. the batch rows of data_z
  float sig2[3];                         // fixed sigma_v^2, sigma_x^2, sigma_y^2 (params['sigma_*']); <= 0: the net's variance head
  // general (any-width) steps spread over the chip (bnn_api.hip): 1 = eps / dW of the calls were written by bnn_step_noise_kernel (the
  // step kernels draw the sign words only), the KL terms and the Adam step follow in their own launches (bnn_kl_adam_kernel)
  int wide;
  float *kl_part;                        // [3][BNN_KL_PARTS] partial sums of the nets' KL terms
};
#define BNN_KL_PARTS 32
#define BNN_NOISE_PARTS 48

// gather the minibatch: zb [B x q], vb [B x p], xb, yb [B], f input [B x nf], h input [B x nh]
struct BnnBatch { float *zb, *vb, *xb, *yb, *fin, *hin; };
__device__ __forceinline__ void bnn_gather_ptrs(const BnnArgs &a, float *&wp, BnnBatch &bt) {
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  const int B = a.B, q = a.q, p = a.p, nf = a.net[BNN_F].dims[0], nh = a.net[BNN_H].dims[0];
  bt.zb = take(B * q); bt.vb = take(B * p); bt.xb = take(B); bt.yb = take(B); bt.fin = take(B * nf); bt.hin = take(B * nh);
}
__device__ __forceinline__ void bnn_gather(const BnnCtx &c, const BnnArgs &a, float *&wp, BnnBatch &bt) {
  const int B = a.B, q = a.q, p = a.p, nf = a.net[BNN_F].dims[0], nh = a.net[BNN_H].dims[0];
  bnn_gather_ptrs(a, wp, bt);
  for (int k = c.tid; k < B * p; k += BNN_THREADS) { const int b = k / p; bt.vb[k] = a.v_[(long long)a.idx[b] * p + (k - b * p)]; }
  for (int k = c.tid; k < B * q; k += BNN_THREADS) { const int b = k / q; bt.zb[k] = a.data_z[(long long)a.idx[b] * q + (k - b * q)]; }
  for (int b = c.tid; b < B; b += BNN_THREADS) { bt.xb[b] = a.x_[a.idx[b]]; bt.yb[b] = a.y_[a.idx[b]]; }
  __syncthreads();
  for (int k = c.tid; k < B * nf; k += BNN_THREADS) {
    const int b = k / nf, i = k - b * nf;
    bt.fin[k] = (i < a.z0 + a.z1) ? bt.zb[b * q + i] : bt.xb[b];
  }
  for (int k = c.tid; k < B * nh; k += BNN_THREADS) {
    const int b = k / nh, i = k - b * nh;
    bt.hin[k] = (i < a.z0) ? bt.zb[b * q + i] : bt.zb[b * q + a.z1 + i];
  }
  __syncthreads();
}

// ssq[b] = sum_j (v[b p + j] - o[b wo + j])^2: one wave per row, lanes stride over the columns (coalesced), fixed order.
__device__ __forceinline__ void bnn_row_ssq(const BnnCtx &c, const float *v, const float *o, int B, int p, int wo, float *ssq) {
  const int lane = c.tid & 63, wave = c.tid >> 6;
  for (int b = wave; b < B; b += BNN_THREADS / 64) {
    float s = 0.0f;
    for (int j = lane; j < p; j += 64) { const float t = v[b * p + j] - o[b * wo + j]; s = fmaf(t, t, s); }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) ssq[b] = s;
  }
}

// Gaussian head: loss_b = ssq / (2 s2) + dim * log(s2) / 2 with s2 = softplus(raw) + 1e-6; returns d loss_b / d raw.
// fix2 > 0: params['sigma_v' | 'sigma_x' | 'sigma_y'] given (causalbgm/base.py:161,195,224,257,268,283): s2 = sigma^2, the variance head
// is not read and receives no gradient.
__device__ __forceinline__ float bnn_gauss(float ssq, float raw, float dim, float &loss_b, float &s2, float fix2 = 0.0f) {
  if (fix2 > 0.0f) {
    s2 = fix2;
    loss_b = ssq / (2.0f * s2) + dim * logf(s2) * 0.5f;
    return 0.0f;
  }
  s2 = softplus_acc(raw) + BGM_EPS;
  loss_b = ssq / (2.0f * s2) + dim * logf(s2) * 0.5f;
  return (-ssq / (2.0f * s2 * s2) + dim / (2.0f * s2)) * sigmoid_f(raw);
}

// update_g_net, update_h_net, update_f_net (causalbgm/base.py:156-243) with use_bnn: the three updates are independent
// given the batch (each reads the latents of BEFORE the step), so one launch does all three, one workgroup per net.
#ifdef BNN_PROF      // development: shader-clock cycles of wave 0 of one net's workgroup (BNN_PROF_WG: 0 g, 1 h, 2 f) per phase of the general theta step (bnn_api.hip prints them)
#ifndef BNN_PROF_WG
#define BNN_PROF_WG 0
#endif
__device__ unsigned long long bnn_prof_acc[8];
__device__ unsigned long long bnn_prof_span[8];
#define BNN_T(i) do { const unsigned long long t2_ = __builtin_amdgcn_s_memtime(); if (c.tid == 0 && blockIdx.x == BNN_PROF_WG) atomicAdd(&bnn_prof_acc[i], t2_ - bnn_t_); bnn_t_ = t2_; } while (0)
#else
#define BNN_T(i) do {} while (0)
#endif
static __global__ __launch_bounds__(BNN_THREADS) void bnn_theta_step_kernel(BnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float bnn_r32_stage[];      // BNN_R32_LDS_BYTES when a.wide (bnn_api.hip), else none
  __shared__ float red[32];
  __shared__ float ssq_row[BNN_MAX_BATCH];
  BnnCtx c{(int)threadIdx.x, red};
  const int B = a.B, p = a.p;
  float *wp = a.ws + (long long)blockIdx.x * a.ws_stride;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  BnnBatch bt;
#ifdef BNN_PROF
  unsigned long long bnn_t_ = __builtin_amdgcn_s_memtime();
  const unsigned long long bnn_rt0_ = wall_clock64();      // (the constant 100 MHz counter: what the shader clock was during the kernel)
#endif
  bnn_gather(c, a, wp, bt);
  BNN_T(0);
  float *d = take(B * a.wmax), *ds = take(B * a.wmax), *t0 = take(B * a.wmax), *t1 = take(B * a.wmax);
  float *cache_base = wp;
  {   // grid = 3 workgroups: g, h, f are independent given the batch
    const int which = blockIdx.x;
    const int id = which == 0 ? BNN_G : (which == 1 ? BNN_H : BNN_F);
    const BnnNet &n = a.net[id];
    wp = cache_base;
    BnnCache k;
    bnn_cache(n, B, wp, k, id == BNN_G ? bt.zb : (id == BNN_H ? bt.hin : bt.fin));
    float *G = nullptr, *GS = nullptr;
    const float *locT = nullptr, *dWT = nullptr;
    if (a.wide && !n.heads) {      // the upstream gradients of all layers stay (in the second call cache of the slice) for bnn_dw_kernel
      BnnCache k2;
      bnn_cache(n, B, wp, k2, nullptr);
      G = k2.H; GS = k2.HS;
      locT = k2.eps; dWT = k2.dW;      // (and its noise arrays hold the transposed weights of the call: bnn_step_noise_kernel)
      d = G + (long long)B * n.hoff[n.n_layers]; ds = GS + (long long)B * n.hoff[n.n_layers];
    }
    bnn_noise(c, a.theta, n, k, B, a.k0, a.k1, a.stream, 0u, a.wide != 0);      // (the steps of bnn_fwd)
    __syncthreads();
    BNN_T(1);
    bnn_bn_fwd(c, a.theta, n, k, B);
    __syncthreads();
    bnn_layers_fwd(c, a.theta, n, k, B, a.wide ? bnn_r32_stage : nullptr, locT, dWT);
    const float *o = k.H + (long long)B * n.hoff[n.heads ? n.n_layers - 1 : n.n_layers];
    BNN_T(2);
    const int wo = n.dims[n.n_layers];
    float loss = 0.0f, aux = 0.0f;
    if (id == BNN_G) {
      // per-row sum of squares, then the head
      bnn_row_ssq(c, bt.vb, o, B, p, wo, ssq_row);
      __syncthreads();
      for (int i = c.tid; i < B * wo; i += BNN_THREADS) {
        const int b = i / wo, j = i - b * wo;
        float lb, s2;
        const float dr = bnn_gauss(ssq_row[b], o[b * wo + wo - 1], (float)p, lb, s2, a.sig2[0]);
        if (j < p) d[i] = -(bt.vb[b * p + j] - o[i]) / s2 * a.inv_B;
        else if (j == wo - 1) { d[i] = dr * a.inv_B; loss += lb; aux += ssq_row[b]; }
        else d[i] = 0.0f;
      }
      loss = bnn_block_sum(c, loss) * a.inv_B;
      aux = bnn_block_sum(c, aux) * a.inv_B / (float)p;
    } else {
      const float *tgt = id == BNN_H ? bt.xb : bt.yb;
      for (int i = c.tid; i < B * wo; i += BNN_THREADS) d[i] = 0.0f;
      __syncthreads();
      for (int b = c.tid; b < B; b += BNN_THREADS) {
        const float l = o[b * wo];
        if (id == BNN_H && a.binary) {
          const float e = fmaxf(l, 0.0f) - l * tgt[b] + log1pf(expf(-fabsf(l)));
          loss += e; aux += e;
          d[b * wo] = (sigmoid_f(l) - tgt[b]) * a.inv_B;
        } else {
          const float r = tgt[b] - l;
          float lb, s2;
          const float dr = bnn_gauss(r * r, o[b * wo + wo - 1], 1.0f, lb, s2, a.sig2[id == BNN_H ? 1 : 2]);
          loss += lb; aux += r * r;
          d[b * wo] = -r / s2 * a.inv_B;
          d[b * wo + wo - 1] += dr * a.inv_B;
        }
      }
      loss = bnn_block_sum(c, loss) * a.inv_B;
      aux = bnn_block_sum(c, aux) * a.inv_B;
    }
    __syncthreads();
    BNN_T(3);
    bnn_bwd(c, a.theta, a.grad, n, k, d, ds, t0, t1, nullptr, B, true, false, G, GS, a.wide ? bnn_r32_stage : nullptr);
    BNN_T(4);
    const float klv = a.wide ? 0.0f : bnn_kl(c, a.theta, a.grad, n, a.kl_weight);      // (wide: bnn_kl_adam_kernel / bnn_kl_finish_kernel)
    __syncthreads();
    BNN_T(5);
    if (a.apply && !a.wide) bnn_adam(c, a.theta + n.off, a.m + n.off, a.v + n.off, a.grad + n.off, n.n_params, a.adam);
    BNN_T(6);
#ifdef BNN_PROF
    if (c.tid == 0 && blockIdx.x == BNN_PROF_WG) atomicAdd(&bnn_prof_acc[7], wall_clock64() - bnn_rt0_);
    if (c.tid == 0) { bnn_prof_span[2 * blockIdx.x] = bnn_rt0_; bnn_prof_span[2 * blockIdx.x + 1] = wall_clock64(); }      // (last call's start / end of each workgroup)
#endif
    if (c.tid == 0 && a.out) { a.out[2 * which] = loss + a.kl_weight * klv; a.out[2 * which + 1] = aux; }
    __syncthreads();
  }
}

// update_latent_variable_sgd (causalbgm/base.py:246-302) with use_bnn: every net is called twice with independent
// noise (mean from the first call, variance head from the second); dz [B x q] = d loss / d (batch rows of data_z).
static __global__ __launch_bounds__(BNN_THREADS) void bnn_z_grad_kernel(BnnArgs a) {
  __shared__ float red[32];
  __shared__ float ssq_row[BNN_MAX_BATCH];
  BnnCtx c{(int)threadIdx.x, red};
  const int B = a.B, p = a.p, q = a.q;
  float *wp = a.ws + (long long)blockIdx.x * a.ws_stride;
  auto take = [&](int n) { float *r = wp; wp += (n + 3) & ~3; return r; };
  BnnBatch bt;
  bnn_gather(c, a, wp, bt);
  float *d = take(B * a.wmax), *ds = take(B * a.wmax), *t0 = take(B * a.wmax), *t1 = take(B * a.wmax);
  float *dx1 = take(B * a.wmax), *dx2 = take(B * a.wmax);
  float *cache_base = wp;
  // grid = 3 workgroups (g, h, f); each writes its share of the gradient and of the loss, workgroup 0 also the prior term
  const int which = blockIdx.x;
  float *dzp = a.dz_part + (long long)which * B * q;
  float total = 0.0f;
  for (int i = c.tid; i < B * q; i += BNN_THREADS) {
    const float z = bt.zb[i];
    dzp[i] = which == 0 ? z * a.inv_B : 0.0f;
    if (which == 0) total += 0.5f * z * z;
  }
  total = bnn_block_sum(c, total) * a.inv_B;
  {
    const int id = which == 0 ? BNN_G : (which == 1 ? BNN_H : BNN_F);
    const BnnNet &n = a.net[id];
    wp = cache_base;
    const float *input = id == BNN_G ? bt.zb : (id == BNN_H ? bt.hin : bt.fin);
    const int wo = n.dims[n.n_layers], in = n.dims[0];
    const bool two = !(id == BNN_H && a.binary);
    BnnCache k1, k2;
    bnn_cache(n, B, wp, k1, input);
    const float *o1 = bnn_fwd(c, a.theta, n, k1, B, a.k0, a.k1, a.stream, 0u, a.wide != 0);
    const float *o2 = o1;
    if (two) { bnn_cache(n, B, wp, k2, input); o2 = bnn_fwd(c, a.theta, n, k2, B, a.k0, a.k1, a.stream + 1u, 0u, a.wide != 0); }
    float loss = 0.0f;
    // upstream gradients: d for call 1 (mean), t-buffers reused for call 2 (variance head) after the first backward
    if (id == BNN_G) {
      bnn_row_ssq(c, bt.vb, o1, B, p, wo, ssq_row);
      __syncthreads();
      for (int i = c.tid; i < B * wo; i += BNN_THREADS) {
        const int b = i / wo, j = i - b * wo;
        float lb, s2;
        bnn_gauss(ssq_row[b], o2[b * wo + wo - 1], (float)p, lb, s2, a.sig2[0]);
        d[i] = (j < p) ? -(bt.vb[b * p + j] - o1[i]) / s2 * a.inv_B : 0.0f;
        if (j == wo - 1) loss += lb;
      }
    } else {
      const float *tgt = id == BNN_H ? bt.xb : bt.yb;
      for (int i = c.tid; i < B * wo; i += BNN_THREADS) d[i] = 0.0f;
      __syncthreads();
      for (int b = c.tid; b < B; b += BNN_THREADS) {
        const float l = o1[b * wo];
        if (!two) {
          loss += fmaxf(l, 0.0f) - l * tgt[b] + log1pf(expf(-fabsf(l)));
          d[b * wo] = (sigmoid_f(l) - tgt[b]) * a.inv_B;
        } else {
          const float r = tgt[b] - l;
          float lb, s2;
          bnn_gauss(r * r, o2[b * wo + wo - 1], 1.0f, lb, s2, a.sig2[id == BNN_H ? 1 : 2]);
          loss += lb;
          d[b * wo] = -r / s2 * a.inv_B;
          ssq_row[b] = r * r;
        }
      }
    }
    total += bnn_block_sum(c, loss) * a.inv_B;
    __syncthreads();
    bnn_bwd(c, a.theta, a.grad, n, k1, d, ds, t0, t1, dx1, B, false, false);
    if (two) {
      const float dim = id == BNN_G ? (float)p : 1.0f;
      for (int i = c.tid; i < B * wo; i += BNN_THREADS) {
        const int b = i / wo, j = i - b * wo;
        float lb, s2;
        d[i] = (j == wo - 1) ? bnn_gauss(ssq_row[b], o2[b * wo + wo - 1], dim, lb, s2, a.sig2[id == BNN_G ? 0 : (id == BNN_H ? 1 : 2)]) * a.inv_B : 0.0f;
      }
      __syncthreads();
      bnn_bwd(c, a.theta, a.grad, n, k2, d, ds, t0, t1, dx2, B, false, false);
    }
    // scatter the input gradient into dz
    for (int i = c.tid; i < B * in; i += BNN_THREADS) {
      const int b = i / in, j = i - b * in;
      const float v = dx1[i] + (two ? dx2[i] : 0.0f);
      int col = -1;
      if (id == BNN_G) col = j;
      else if (id == BNN_F) col = (j < a.z0 + a.z1) ? j : -1;
      else col = (j < a.z0) ? j : j + a.z1;
      if (col >= 0) dzp[b * q + col] += v;      // one thread per (b, col)
    }
    __syncthreads();
  }
  if (c.tid == 0) a.loss_part[which] = total;
}

// The latent step's gradient with the two noise calls of a net on workgroups of their own (general widths, bgm_bnn_z_step): grid = 6
// workgroups (net = blockIdx.x >> 1, call = blockIdx.x & 1), each on its own workspace slice.  bnn_z_fwd_kernel: gather + forward of the
// call (eps / dW from bnn_step_noise_kernel); bnn_z_bwd_kernel: the call's upstream gradient (call 0: the mean, with the other call's
// variance output; call 1: the variance head, with call 0's residuals recomputed) and its backward; the input gradients land in
// dz_part [6][B x q] and bnn_z_combine6_kernel adds them up as bnn_z_grad_kernel does: the same sums in the same order.
struct BnnZSlice { BnnBatch bt; float *d, *ds, *t0, *t1, *dx; BnnCache k; const float *o; const float *locT, *dWT; };
__device__ __forceinline__ void bnn_z_slice(const BnnArgs &a, const BnnNet &n, int id, int slice, BnnZSlice &z) {
  float *wp = a.ws + (long long)slice * a.ws_stride;
  auto take = [&](int cnt) { float *r = wp; wp += (cnt + 3) & ~3; return r; };
  bnn_gather_ptrs(a, wp, z.bt);
  z.d = take(a.B * a.wmax); z.ds = take(a.B * a.wmax); z.t0 = take(a.B * a.wmax); z.t1 = take(a.B * a.wmax);
  z.dx = take(a.B * a.wmax); take(a.B * a.wmax);
  bnn_cache(n, a.B, wp, z.k, id == BNN_G ? z.bt.zb : (id == BNN_H ? z.bt.hin : z.bt.fin));
  z.o = z.k.H + (long long)a.B * n.hoff[n.heads ? n.n_layers - 1 : n.n_layers];
  BnnCache k2;      // the slice's second cache: its noise arrays hold the call's transposed weights (bnn_step_noise_kernel)
  bnn_cache(n, a.B, wp, k2, nullptr);
  z.locT = k2.eps; z.dWT = k2.dW;
}
static __global__ __launch_bounds__(BNN_THREADS) void bnn_z_fwd_kernel(BnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float bnn_r32_stage[];      // BNN_R32_LDS_BYTES
  __shared__ float red[32];
  BnnCtx c{(int)threadIdx.x, red};
  const int which = blockIdx.x >> 1, call = blockIdx.x & 1;
  const int id = which == 0 ? BNN_G : (which == 1 ? BNN_H : BNN_F);
  if (call == 1 && id == BNN_H && a.binary) return;
  const BnnNet &n = a.net[id];
  float *wp = a.ws + (long long)blockIdx.x * a.ws_stride;
  BnnBatch bt;
  bnn_gather(c, a, wp, bt);
  BnnZSlice z;
  bnn_z_slice(a, n, id, blockIdx.x, z);
  bnn_fwd(c, a.theta, n, z.k, a.B, a.k0, a.k1, a.stream + (uint32_t)call, 0u, true, bnn_r32_stage, n.heads ? nullptr : z.locT, z.dWT);
}
static __global__ __launch_bounds__(BNN_THREADS) void bnn_z_bwd_kernel(BnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float bnn_r32_stage[];      // BNN_R32_LDS_BYTES
  __shared__ float red[32];
  __shared__ float ssq_row[BNN_MAX_BATCH];
  BnnCtx c{(int)threadIdx.x, red};
  const int B = a.B, p = a.p, q = a.q;
  const int which = blockIdx.x >> 1, call = blockIdx.x & 1;
  const int id = which == 0 ? BNN_G : (which == 1 ? BNN_H : BNN_F);
  const bool two = !(id == BNN_H && a.binary);
  if (call == 1 && !two) return;
  const BnnNet &n = a.net[id];
  BnnZSlice z, zo;
  bnn_z_slice(a, n, id, blockIdx.x, z);
  bnn_z_slice(a, n, id, blockIdx.x ^ 1, zo);
  const BnnBatch &bt = z.bt;
  const float *o1 = call ? zo.o : z.o, *o2 = two ? (call ? z.o : zo.o) : o1;
  const int wo = n.dims[n.n_layers], in = n.dims[0];
  float *d = z.d;
  float *dzp = a.dz_part + (long long)blockIdx.x * B * q;
  for (int i = c.tid; i < B * q; i += BNN_THREADS) {
    dzp[i] = 0.0f;
    if (!call && !two) dzp[(long long)B * q + i] = 0.0f;      // (the call that does not exist)
  }
  float total = 0.0f;
  if (!call) {
    if (which == 0) for (int i = c.tid; i < B * q; i += BNN_THREADS) { const float zz = bt.zb[i]; total += 0.5f * zz * zz; }
    total = bnn_block_sum(c, total) * a.inv_B;
    float loss = 0.0f;
    if (id == BNN_G) {
      bnn_row_ssq(c, bt.vb, o1, B, p, wo, ssq_row);
      __syncthreads();
      for (int i = c.tid; i < B * wo; i += BNN_THREADS) {
        const int b = i / wo, j = i - b * wo;
        float lb, s2;
        bnn_gauss(ssq_row[b], o2[b * wo + wo - 1], (float)p, lb, s2, a.sig2[0]);
        d[i] = (j < p) ? -(bt.vb[b * p + j] - o1[i]) / s2 * a.inv_B : 0.0f;
        if (j == wo - 1) loss += lb;
      }
    } else {
      const float *tgt = id == BNN_H ? bt.xb : bt.yb;
      for (int i = c.tid; i < B * wo; i += BNN_THREADS) d[i] = 0.0f;
      __syncthreads();
      for (int b = c.tid; b < B; b += BNN_THREADS) {
        const float l = o1[b * wo];
        if (!two) {
          loss += fmaxf(l, 0.0f) - l * tgt[b] + log1pf(expf(-fabsf(l)));
          d[b * wo] = (sigmoid_f(l) - tgt[b]) * a.inv_B;
        } else {
          const float r = tgt[b] - l;
          float lb, s2;
          bnn_gauss(r * r, o2[b * wo + wo - 1], 1.0f, lb, s2, a.sig2[id == BNN_H ? 1 : 2]);
          loss += lb;
          d[b * wo] = -r / s2 * a.inv_B;
        }
      }
    }
    total += bnn_block_sum(c, loss) * a.inv_B;
  } else {
    if (id == BNN_G) bnn_row_ssq(c, bt.vb, o1, B, p, wo, ssq_row);
    else {
      const float *tgt = id == BNN_H ? bt.xb : bt.yb;
      for (int b = c.tid; b < B; b += BNN_THREADS) { const float r = tgt[b] - o1[b * wo]; ssq_row[b] = r * r; }
    }
    __syncthreads();
    const float dim = id == BNN_G ? (float)p : 1.0f;
    for (int i = c.tid; i < B * wo; i += BNN_THREADS) {
      const int b = i / wo, j = i - b * wo;
      float lb, s2;
      d[i] = (j == wo - 1) ? bnn_gauss(ssq_row[b], o2[b * wo + wo - 1], dim, lb, s2, a.sig2[id == BNN_G ? 0 : (id == BNN_H ? 1 : 2)]) * a.inv_B : 0.0f;
    }
  }
  __syncthreads();
  bnn_bwd(c, a.theta, a.grad, n, z.k, d, z.ds, z.t0, z.t1, z.dx, B, false, false, nullptr, nullptr, bnn_r32_stage);
  for (int i = c.tid; i < B * in; i += BNN_THREADS) {
    const int b = i / in, j = i - b * in;
    int col = -1;
    if (id == BNN_G) col = j;
    else if (id == BNN_F) col = (j < a.z0 + a.z1) ? j : -1;
    else col = (j < a.z0) ? j : j + a.z1;
    if (col >= 0) dzp[b * q + col] = z.dx[i];      // one thread per (b, col)
  }
  if (!call && c.tid == 0) a.loss_part[which] = total;
}
// dz = (z / B + (g call 0 + g call 1)) + (h 0 + h 1) + (f 0 + f 1): the sums of bnn_z_grad_kernel + bnn_z_combine_kernel, term by term
static __global__ void bnn_z_combine6_kernel(const float *dz_part, const float *loss_part, float *dz, float *out, int n, const float *data_z,
                                             const int *idx, int q, float inv_B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int b = i / q, col = i - b * q;
    const float p0 = __fadd_rn(__fmul_rn(data_z[(long long)idx[b] * q + col], inv_B), __fadd_rn(dz_part[i], dz_part[n + i]));
    const float p1 = __fadd_rn(0.0f, __fadd_rn(dz_part[2 * n + i], dz_part[3 * n + i]));
    const float p2 = __fadd_rn(0.0f, __fadd_rn(dz_part[4 * n + i], dz_part[5 * n + i]));
    dz[i] = __fadd_rn(__fadd_rn(p0, p1), p2);
  }
  if (i == 0 && out) out[0] = loss_part[0] + loss_part[1] + loss_part[2];
}

// dz = sum of the three per-net partials; loss_postrior_z likewise
// ---- the elementwise parts of the general steps, spread over the chip (one workgroup per net walks everything else) ----------------
// eps and dW = sigma * eps of the calls of a step, written where the step kernel's call caches will look for them: the same pointer
// arithmetic as bnn_theta_step_kernel (scratch = 4 row buffers, one call per net) / bnn_z_grad_kernel (6, two calls unless the
// treatment is binary).  grid (BNN_NOISE_PARTS, 3 nets, calls); same draws as bnn_noise.
static __global__ __launch_bounds__(BNN_THREADS) void bnn_step_noise_kernel(BnnArgs a, int n_scratch, int split) {
  // split: every (net, call) has a workspace slice of its own (bnn_z_fwd_kernel / bnn_z_bwd_kernel), its cache is the slice's first
  const int which = blockIdx.y, call = blockIdx.z;
  const int id = which == 0 ? BNN_G : (which == 1 ? BNN_H : BNN_F);
  if (call == 1 && id == BNN_H && a.binary) return;
  const BnnNet &n = a.net[id];
  float *wp = a.ws + (long long)(split ? 2 * which + call : which) * a.ws_stride;
  BnnBatch bt;
  bnn_gather_ptrs(a, wp, bt);
  wp += (long long)n_scratch * ((a.B * a.wmax + 3) & ~3);
  BnnCache k, k2;
  bnn_cache(n, a.B, wp, k, nullptr);
  if (call == 1 && !split) bnn_cache(n, a.B, wp, k, nullptr);
  bnn_cache(n, a.B, wp, k2, nullptr);      // the cache behind the call's: its noise arrays take the call's weights transposed ([out][in] per layer:
  const bool tr = !n.heads;                //   posterior means -> k2.eps, dW -> k2.dW) for the forward products of bnn_gemm2_rows32
  const uint32_t stream = a.stream + (uint32_t)call;
  for (int l = 0; l < n.n_layers; ++l) {
    const int in = n.lin[l], out = n.lout[l], cnt = in * out;
    const float *loc = a.theta + n.woff[l], *rho = loc + cnt;
    float *e = k.eps + n.eoff[l], *d = k.dW + n.eoff[l], *lt = k2.eps + n.eoff[l], *dt = k2.dW + n.eoff[l];
    for (int i = blockIdx.x * BNN_THREADS + threadIdx.x; i < (cnt + 3) >> 2; i += gridDim.x * BNN_THREADS) {
      const f32x4 z = box_muller4(philox4x32_10((uint32_t)i, (uint32_t)l | ((uint32_t)n.net_id << 16), stream, BNN_TAG_EPS, a.k0, a.k1));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = 4 * i + u;
        if (idx < cnt) {
          const float dw = (BNN_SCALE_EPS + softplus_acc(rho[idx])) * z[u];
          e[idx] = z[u]; d[idx] = dw;
          if (tr) { const int ki = idx / out, no = idx - ki * out; lt[no * in + ki] = loc[idx]; dt[no * in + ki] = dw; }
        }
      }
    }
  }
}
// the parameter-gradient tiles of a general theta step (d loc, d rho, d bias of every Flipout layer of g, h, f), left by the step kernel:
// the layers' inputs are in the call cache, their upstream gradients in G / GS (bnn_bwd).  grid (BNN_DW_PARTS, 3 nets): the tiles of a
// layer over all waves of the net's workgroups, the same arithmetic per tile as bnn_bwd_params inside the step kernel.
#define BNN_DW_PARTS 48
static __global__ __launch_bounds__(BNN_THREADS) void bnn_dw_kernel(BnnArgs a) {
  __shared__ float red[32];
  BnnCtx c{(int)threadIdx.x, red};
  const int which = blockIdx.y, id = which == 0 ? BNN_G : (which == 1 ? BNN_H : BNN_F);
  const BnnNet &n = a.net[id];
  if (n.heads) return;
  float *wp = a.ws + (long long)which * a.ws_stride;
  BnnBatch bt;
  bnn_gather_ptrs(a, wp, bt);
  wp += 4LL * ((a.B * a.wmax + 3) & ~3);
  BnnCache k, k2;
  bnn_cache(n, a.B, wp, k, nullptr);
  bnn_cache(n, a.B, wp, k2, nullptr);
  for (int l = 0; l < n.n_layers; ++l)
    bnn_bwd_params(c, a.theta, a.grad, n, k, l, k2.H + (long long)a.B * n.hoff[l + 1], k2.HS + (long long)a.B * n.hoff[l + 1], a.B, false,
                   (int)blockIdx.x, (int)gridDim.x);
}
// grad += w * dKL/dtheta as bnn_kl (a slice of every layer per workgroup, the slice's share of sum(net.losses) -> kl_part) and, with
// a.apply, the Adam step of every parameter of g, h, f in the same launch: an element's KL gradient and its
// Adam step are the same thread's (no ordering between workgroups needed); the parameters the KL terms do not touch (gamma, beta, the
// moving statistics, the biases without a prior) take their step in a second loop.  grid (BNN_KL_PARTS, 3 nets)
static __global__ __launch_bounds__(256) void bnn_kl_adam_kernel(BnnArgs a) {
  __shared__ float red[4];
  const int which = blockIdx.y, id = which == 0 ? BNN_G : (which == 1 ? BNN_H : BNN_F);
  const BnnNet &n = a.net[id];
  const float iv = n.prior_iv, ls = n.prior_logs, w = a.kl_weight;
  const int gt = blockIdx.x * 256 + threadIdx.x, gs = gridDim.x * 256;
  auto adam = [&](int j) {      // parameter j of the flat vector
    const BnnAdamOut o = bnn_adam_one(a.theta[j], a.m[j], a.v[j], a.grad[j], a.adam);
    a.m[j] = o.m; a.v[j] = o.v; a.theta[j] = o.th;
  };
  float acc = 0.0f;
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.lin[l] * n.lout[l], o_loc = n.woff[l], o_rho = o_loc + cnt, o_b = o_rho + cnt;
    for (int i = gt; i < cnt; i += gs) {
      const float rho = a.theta[o_rho + i], mu = a.theta[o_loc + i];
      const float sg = BNN_SCALE_EPS + softplus_acc(rho);
      acc += -logf(sg) + 0.5f * (sg * sg + mu * mu) * iv - 0.5f + ls;
      a.grad[o_loc + i] += w * mu * iv;
      a.grad[o_rho + i] += w * (-1.0f / sg + sg * iv) * sigmoid_f(rho);
      if (a.apply) { adam(o_loc + i); adam(o_rho + i); }
    }
    for (int i = gt; i < n.lout[l]; i += gs) {
      if (n.bias_prior) {
        const float b = a.theta[o_b + i];
        acc += 0.5f * b * b * iv + ls + 0.9189385332046727f;
        a.grad[o_b + i] += w * b * iv;
      }
      if (a.apply) adam(o_b + i);
    }
  }
  if (a.apply) for (int i = gt; i < (n.mv ? 4 : 2) * n.dims[0]; i += gs) adam(n.off + i);      // gamma, beta (, moving mean, moving variance)
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) a.kl_part[which * BNN_KL_PARTS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// out[2 net] += kl_weight * KL(net): the partial sums in a fixed order
static __global__ void bnn_kl_finish_kernel(BnnArgs a) {
  const int which = threadIdx.x;
  if (which >= 3 || !a.out) return;
  float t = 0.0f;
  for (int i = 0; i < BNN_KL_PARTS; ++i) t += a.kl_part[which * BNN_KL_PARTS + i];
  a.out[2 * which] += a.kl_weight * t;
}

static __global__ void bnn_z_combine_kernel(const float *dz_part, const float *loss_part, float *dz, float *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dz[i] = dz_part[i] + dz_part[n + i] + dz_part[2 * n + i];
  if (i == 0 && out) out[0] = loss_part[0] + loss_part[1] + loss_part[2];
}

// Adam on the [N x q] latent table, Keras `_resource_apply_sparse` semantics (oracle/fit.py adam_rows):
// dense-decay = decay every row's slots, add the batch rows' gradient, update EVERY row; lazy = batch rows only.
static __global__ void bnn_z_decay_kernel(float *zm, float *zv, long long n, float b1, float b2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { zm[i] *= b1; zv[i] *= b2; }
}
// one element of a batch row (lazy / replay forms), roundings spelled out: the latent chain kernel applies the same step in its epilogue
// (bgm_bnn_fit_epoch, EcbZRows) and the two must agree to the bit
__device__ __forceinline__ void bnn_z_row_adam(float *data_z, float *zm, float *zv, long long t, float g, float lr_t, float b1, float b2, float eps) {
  const float m = fmaf(b1, zm[t], __fmul_rn(1.0f - b1, g)), v = fmaf(b2, zv[t], __fmul_rn(__fmul_rn(1.0f - b2, g), g));
  zm[t] = m; zv[t] = v;
  data_z[t] = __fsub_rn(data_z[t], __fdiv_rn(__fmul_rn(lr_t, m), __fadd_rn(sqrtf(v), eps)));
}
static __global__ void bnn_z_rows_kernel(float *data_z, float *zm, float *zv, const float *dz, const int *idx, int B, int q,
                                         float lr_t, float b1, float b2, float eps, int lazy, int *t_last = nullptr, int t_now = 0,
                                         FitSync sy = FitSync{}) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * q) {
    const int b = i / q, j = i - b * q;
    const long long t = (long long)idx[b] * q + j;
    const float g = dz[i];
    if (lazy) {
      bnn_z_row_adam(data_z, zm, zv, t, g, lr_t, b1, b2, eps);
      if (t_last && j == 0) t_last[idx[b]] = t_now;      // replay mode (z_replay.h): the row is current to this step
    } else {
      zm[t] += (1.0f - b1) * g;
      zv[t] += (1.0f - b2) * g * g;
    }
  }
  fit_sync_done(sy);      // bgm_bnn_fit_epoch: the latent phase of this minibatch is complete
}
static __global__ void bnn_z_apply_kernel(float *data_z, const float *zm, const float *zv, long long n, float lr_t, float eps) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) data_z[i] -= lr_t * zm[i] / (sqrtf(zv[i]) + eps);
}
static __global__ void bnn_adam_kernel(float *theta, float *m, float *v, const float *g, int n, BnnAdam a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const BnnAdamOut o = bnn_adam_one(theta[i], m[i], v[i], g[i], a);
    m[i] = o.m; v[i] = o.v; theta[i] = o.th;
  }
}
