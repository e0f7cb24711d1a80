// gx_bgm_kernels.h -- BGM on the general-width engine (gx_device.h): masked log-posterior and its latent gradient, persistent HMC,
// posterior-predictive draws and the minibatch fit steps for a generator g_net = BaseVariationalNet of ANY trunk widths / depth and any
// latent width (params['g_units'], ['z_dim'] of bgm/base.py:59-80; networks/base.py:53-117).
//
// replaces (src/bayesgm/models/bgm/base.py):
//   get_log_posterior :665-705 (+ its gradient, the target of tfp.mcmc.HamiltonianMonteCarlo :798-821) -> gx_bgm_logp_grad (device),
//                                                                                   gx_bgm_logpost_kernel, gx_bgm_hmc_kernel
//   predict_on_posteriors :511-525, g_net(z, training=False) :468,503                -> gx_bgm_predict_kernel
//   update_g_net :145-164, update_latent_variable_sgd :167-187 (training-mode net)   -> gx_bgm_fit_kernel (+ the shared bn / dW kernels)
// RNG streams, counters and output layouts are those of bgm_kernels.h (oracle/rng.py).
//
// The two p-wide heads are ONE padded layer [H][2 Pp] (mean columns, then variance columns at Pp = pad32(p)), processed in column
// chunks that fit an LDS buffer: chunk forward -> masked Gaussian terms and their derivatives in place -> chunk backward
// accumulated into d logp / d h.  LeakyReLU derivatives of the trunk are kept as bit pairs (one byte per two features).
#pragma once
#include "gx_device.h"

struct GxBgmModel {
  GxNet g;                 // trunk layers 0 .. L-2, heads layer L-1 (width 2 Pp)
  const float *pack, *packT;
  const float *bnp;        // gamma | beta | moving mean | moving variance, [4 q] (canonical order, device)
  int q, p, Pp, ld, ch;    // ch: head columns per chunk (multiple of 32, <= ld - 8)
  int moff[GX_MAXL];       // byte offset of the mask of trunk layer l
  int mask_bytes;
};

struct GxBgmLds {
  float *B0, *B1, *B2, *B3, *zs, *zc, *mom, *gr, *gc, *lp, *lpc, *ll, *sc, *sh, *red;
  unsigned char *mask;
};
__host__ __device__ inline int gx_bgm_lds_bytes(int ld, int q, int mask_bytes) {
  return 4 * (4 * GX_ROWS * ld + 5 * GX_ROWS * q + 3 * GX_ROWS + 2 * q + 128) + ((mask_bytes + 15) & ~15);
}
__device__ __forceinline__ GxBgmLds gx_bgm_carve(float *lds, int ld, int q) {
  GxBgmLds L;
  L.B0 = lds; L.B1 = L.B0 + GX_ROWS * ld; L.B2 = L.B1 + GX_ROWS * ld; L.B3 = L.B2 + GX_ROWS * ld;
  L.zs = L.B3 + GX_ROWS * ld; L.zc = L.zs + GX_ROWS * q; L.mom = L.zc + GX_ROWS * q; L.gr = L.mom + GX_ROWS * q; L.gc = L.gr + GX_ROWS * q;
  L.lp = L.gc + GX_ROWS * q; L.lpc = L.lp + GX_ROWS; L.ll = L.lpc + GX_ROWS; L.sc = L.ll + GX_ROWS; L.sh = L.sc + q; L.red = L.sh + q;
  L.mask = reinterpret_cast<unsigned char *>(L.red + 128);
  return L;
}

// y = lrelu(acc + b) -> Y, and the sign bits of the pair -> mask[row * (N / 2) + (n0 / 2 + j)]
struct GxStoreMask {
  float *Y; int ldy; const float *bias; unsigned char *mask; int mrow;
  __device__ __forceinline__ void operator()(int rt, int n0, const f32x4 &a0, const f32x4 &a1) const {
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const f32x2 bb = *reinterpret_cast<const f32x2 *>(bias + n0 + 2 * j);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p0 = a0[r] + bb[0], p1 = a1[r] + bb[1];
      const int row = 16 * rt + 4 * g + r;
      f32x2 o = {lrelu(p0), lrelu(p1)};
      *reinterpret_cast<f32x2 *>(Y + (size_t)row * ldy + n0 + 2 * j) = o;
      mask[row * mrow + (n0 >> 1) + j] = (unsigned char)((p0 > 0.0f ? 1 : 0) | (p1 > 0.0f ? 2 : 0));
    }
  }
};
// dX * LeakyReLU'(pre-activation of the layer below) -> Y
struct GxMaskBack {
  float *Y; int ldy; const unsigned char *mask; int mrow;
  __device__ __forceinline__ void operator()(int rt, int n0, const f32x4 &a0, const f32x4 &a1) const {
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rt + 4 * g + r;
      const unsigned mk = mask[row * mrow + (n0 >> 1) + j];
      f32x2 o = {a0[r] * ((mk & 1u) ? 1.0f : BGM_LEAK), a1[r] * ((mk & 2u) ? 1.0f : BGM_LEAK)};
      *reinterpret_cast<f32x2 *>(Y + (size_t)row * ldy + n0 + 2 * j) = o;
    }
  }
};
// Y (+)= acc
struct GxAccum {
  float *Y; int ldy; bool first;
  __device__ __forceinline__ void operator()(int rt, int n0, const f32x4 &a0, const f32x4 &a1) const {
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x2 *p = reinterpret_cast<f32x2 *>(Y + (size_t)(16 * rt + 4 * g + r) * ldy + n0 + 2 * j);
      f32x2 o = {a0[r], a1[r]};
      if (!first) { const f32x2 old = *p; o[0] += old[0]; o[1] += old[1]; }
      *p = o;
    }
  }
};

__device__ __forceinline__ void gx_bgm_affine(const GxBgmModel &m, const GxBgmLds &L) {     // inference-mode BatchNorm as z * scale + shift
  for (int c = threadIdx.x; c < m.q; c += GX_THREADS) {
    const float scale = m.bnp[c] / sqrtf(m.bnp[3 * m.q + c] + 1e-3f);
    L.sc[c] = scale; L.sh[c] = m.bnp[m.q + c] - m.bnp[2 * m.q + c] * scale;
  }
  __syncthreads();
}

// trunk forward of the rows z (LDS [32][q]); returns the buffer holding the last hidden activation (B0 or B1)
template <bool MASKS>
__device__ __forceinline__ float *gx_bgm_trunk(const GxBgmModel &m, const GxBgmLds &L, const float *z) {
  const int q = m.q, ld = m.ld, T = m.g.L - 1;
  for (int i = threadIdx.x; i < GX_ROWS * m.g.pad[0]; i += GX_THREADS) {
    const int r = i / m.g.pad[0], c = i - r * m.g.pad[0];
    L.B0[r * ld + c] = c < q ? fmaf(z[r * q + c], L.sc[c], L.sh[c]) : 0.0f;
  }
  __syncthreads();
  float *cur = L.B0, *oth = L.B1;
  for (int l = 0; l < T; ++l) {
    if (MASKS) gx_dense(m.pack + m.g.w[l], m.g.pad[l], m.g.pad[l + 1], cur, ld, GxStoreMask{oth, ld, m.pack + m.g.b[l], L.mask + m.moff[l], m.g.pad[l + 1] >> 1});
    else gx_dense(m.pack + m.g.w[l], m.g.pad[l], m.g.pad[l + 1], cur, ld, GxStore<true>{oth, ld, m.pack + m.g.b[l]});
    __syncthreads();
    float *t = cur; cur = oth; oth = t;
  }
  return cur;
}

// log p(z | x_obs) + const (-> lp[32]) and, GRAD, d logp / d z (-> grad [32][q]) for the tile's rows; x rows with NaN = missing.
template <bool GRAD>
__device__ __forceinline__ void gx_bgm_logp_grad(const GxBgmModel &m, const GxBgmLds &L, const float *z, const float *x, long long row0,
                                                 long long n, float *lp, float *grad) {
  const int q = m.q, ld = m.ld, T = m.g.L - 1, Hp = m.g.pad[T], P2 = m.g.pad[T + 1], Pp = m.Pp, p = m.p;
  float *cur = gx_bgm_trunk<GRAD>(m, L, z);
  float *Mb = (cur == L.B0) ? L.B1 : L.B0, *Sb = L.B3, *dH = L.B2;
  const float *Wh = m.pack + m.g.w[T], *bh = m.pack + m.g.b[T], *WTh = m.packT + m.g.wt[T];
  const int row = threadIdx.x >> 3, sub = threadIdx.x & 7;
  long long gr_ = row0 + row; gr_ = gr_ < n ? gr_ : n - 1;
  const float *xr = x + gr_ * (long long)p;
  float ll = 0.0f;
  for (int c0 = 0; c0 < Pp; c0 += m.ch) {
    const int ch = min(m.ch, Pp - c0);
    gx_dense_ld(Wh + c0, P2, Hp, ch, cur, ld, GxStore<false>{Mb, ld, bh + c0});
    gx_dense_ld(Wh + Pp + c0, P2, Hp, ch, cur, ld, GxStore<false>{Sb, ld, bh + Pp + c0});
    __syncthreads();
    for (int c = sub; c < ch; c += 8) {
      const float xv = (c0 + c < p) ? xr[c0 + c] : __builtin_nanf("");
      float dm = 0.0f, ds = 0.0f;
      if (xv == xv) {          // observed cell (bgm/base.py:689-700: missing cells carry a zero mask)
        const float mu = Mb[row * ld + c], sraw = Sb[row * ld + c];
        const float s2 = softplus_f(sraw) + BGM_EPS, is2 = fast_rcp(s2), d = xv - mu;
        ll += 0.5f * (d * d * is2 + fast_log(s2));
        if (GRAD) { dm = d * is2; ds = (0.5f * d * d * is2 * is2 - 0.5f * is2) * fast_rcp(1.0f + fast_exp(-sraw)); }
      }
      if (GRAD) { Mb[row * ld + c] = dm; Sb[row * ld + c] = ds; }
    }
    __syncthreads();
    if (GRAD) {
      gx_dense_ld(WTh + (size_t)c0 * Hp, Hp, ch, Hp, Mb, ld, GxAccum{dH, ld, c0 == 0});
      gx_dense_ld(WTh + (size_t)(Pp + c0) * Hp, Hp, ch, Hp, Sb, ld, GxAccum{dH, ld, false});
      __syncthreads();
    }
  }
  ll += __shfl_xor(ll, 1); ll += __shfl_xor(ll, 2); ll += __shfl_xor(ll, 4);
  if (sub == 0) {
    float zz = 0.0f;
    for (int c = 0; c < q; ++c) zz = fmaf(z[row * q + c], z[row * q + c], zz);
    lp[row] = -(ll + 0.5f * zz);
  }
  if (GRAD) {
    // d logp / d h_T -> pre-activation of the last trunk layer, then down the trunk
    const unsigned char *mk = L.mask + m.moff[T - 1];
    for (int i = threadIdx.x; i < GX_ROWS * Hp; i += GX_THREADS) {
      const int r = i / Hp, c = i - r * Hp;
      if (!((mk[r * (Hp >> 1) + (c >> 1)] >> (c & 1)) & 1u)) dH[r * ld + c] *= BGM_LEAK;
    }
    __syncthreads();
    float *A = dH, *o1 = Mb;         // outputs alternate between the two chunk buffers; dH is only ever read
    for (int l = T - 1; l >= 1; --l) {
      gx_dense(m.packT + m.g.wt[l], m.g.pad[l + 1], m.g.pad[l], A, ld, GxMaskBack{o1, ld, L.mask + m.moff[l - 1], m.g.pad[l] >> 1});
      __syncthreads();
      A = o1; o1 = (A == Mb) ? Sb : Mb;
    }
    gx_dense(m.packT + m.g.wt[0], m.g.pad[1], m.g.pad[0], A, ld, GxRawStore{o1, ld});
    __syncthreads();
    for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
      const int r = i / q, c = i - r * q;
      grad[i] = fmaf(o1[r * ld + c], L.sc[c], -z[i]);
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(GX_THREADS) void gx_bgm_logpost_kernel(GxBgmModel m, const float *z, const float *x, long long n, float *out,
                                                                    float *grad) {
  extern __shared__ float lds[];
  const GxBgmLds L = gx_bgm_carve(lds, m.ld, m.q);
  const int q = m.q;
  gx_bgm_affine(m, L);
  const long long tiles = (n + GX_ROWS - 1) / GX_ROWS;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long row0 = t * GX_ROWS;
    for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
      long long gr = row0 + i / q; gr = gr < n ? gr : n - 1;
      L.zs[i] = z[gr * q + i % q];
    }
    __syncthreads();
    if (grad) gx_bgm_logp_grad<true>(m, L, L.zs, x, row0, n, L.lp, L.gr);
    else gx_bgm_logp_grad<false>(m, L, L.zs, x, row0, n, L.lp, nullptr);
    if (threadIdx.x < GX_ROWS && row0 + threadIdx.x < n) out[row0 + threadIdx.x] = L.lp[threadIdx.x];
    if (grad)
      for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
        const long long gr = row0 + i / q;
        if (gr < n) grad[gr * q + i % q] = L.gr[i];
      }
    __syncthreads();
  }
}

struct GxHmcArgs {
  GxBgmModel m;
  const float *x;
  long long n, row_base;
  float *state, *logp, *grad;
  int init, it_begin, n_iters, burn_in, n_leapfrog;
  const float *step;
  unsigned k0, k1;
  double *acc_prob_sum;
  unsigned *acc_count;
  float *draws;
};

// One HamiltonianMonteCarlo.one_step per iteration for the tile's 32 chains (bgm/base.py:798-821; oracle/bgm.py hmc_transition).
static __global__ __launch_bounds__(GX_THREADS) void gx_bgm_hmc_kernel(GxHmcArgs a) {
  extern __shared__ float lds[];
  const GxBgmModel &m = a.m;
  const GxBgmLds L = gx_bgm_carve(lds, m.ld, m.q);
  const int q = m.q, ncall = (q + 15) >> 4;
  gx_bgm_affine(m, L);
  const long long n = a.n, tiles = (n + GX_ROWS - 1) / GX_ROWS;
  float *ke0 = L.red, *ke1 = L.red + 32, *flag = L.red + 64;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long row0 = t * GX_ROWS;
    if (a.init) {      // initial_state ~ N(0, 1)  (bgm/base.py:778), RNG tag 0
      for (int i = threadIdx.x; i < GX_ROWS * 4 * ncall; i += GX_THREADS) {
        const int r = i / (4 * ncall), c = i - r * 4 * ncall, g = c & 3, tt = c >> 2;
        const f32x4 nz = box_muller4(philox4x32_10((unsigned)(a.row_base + row0 + r), 0u, (unsigned)(g + 4 * tt), TAG_INIT, a.k0, a.k1));
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int f = 16 * tt + 4 * w + g; if (f < q) L.zs[r * q + f] = nz[w]; }
      }
      __syncthreads();
      gx_bgm_logp_grad<true>(m, L, L.zs, a.x, row0, n, L.lp, L.gr);
    } else {
      for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
        long long gr = row0 + i / q; gr = gr < n ? gr : n - 1;
        L.zs[i] = a.state[gr * q + i % q]; L.gr[i] = a.grad[gr * q + i % q];
      }
      if (threadIdx.x < GX_ROWS) { long long gr = row0 + threadIdx.x; gr = gr < n ? gr : n - 1; L.lp[threadIdx.x] = a.logp[gr]; }
      __syncthreads();
    }
    for (int it = a.it_begin; it < a.it_begin + a.n_iters; ++it) {
      const float eps = *a.step;
      for (int i = threadIdx.x; i < GX_ROWS * 4 * ncall; i += GX_THREADS) {       // momentum ~ N(0, I), RNG tag 4
        const int r = i / (4 * ncall), c = i - r * 4 * ncall, g = c & 3, tt = c >> 2;
        const f32x4 nz = box_muller4(philox4x32_10((unsigned)(a.row_base + row0 + r), (unsigned)it, (unsigned)(g + 4 * tt), TAG_MOM, a.k0, a.k1));
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int f = 16 * tt + 4 * w + g; if (f < q) L.mom[r * q + f] = nz[w]; }
      }
      __syncthreads();
      if (threadIdx.x < GX_ROWS) {
        float s = 0.0f;
        for (int c = 0; c < q; ++c) s = fmaf(L.mom[threadIdx.x * q + c], L.mom[threadIdx.x * q + c], s);
        ke0[threadIdx.x] = s;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) { L.mom[i] = fmaf(0.5f * eps, L.gr[i], L.mom[i]); L.zc[i] = L.zs[i]; }   // first half kick
      __syncthreads();
      for (int l = 0; l < a.n_leapfrog; ++l) {
        for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) L.zc[i] = fmaf(eps, L.mom[i], L.zc[i]);
        __syncthreads();
        gx_bgm_logp_grad<true>(m, L, L.zc, a.x, row0, n, L.lpc, L.gc);
        const float kick = (l < a.n_leapfrog - 1) ? eps : 0.5f * eps;
        for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) L.mom[i] = fmaf(kick, L.gc[i], L.mom[i]);
        __syncthreads();
      }
      if (threadIdx.x < 64) {
        const int r = threadIdx.x & 31;
        const bool me = threadIdx.x < GX_ROWS, ok = me && (row0 + r < n);
        bool acc = false; float pa = 0.0f;
        if (me) {
          float s = 0.0f;
          for (int c = 0; c < q; ++c) s = fmaf(L.mom[r * q + c], L.mom[r * q + c], s);
          float log_ratio = -((-L.lpc[r] + 0.5f * s) - (-L.lp[r] + 0.5f * ke0[r]));
          log_ratio = (log_ratio == log_ratio && fabsf(log_ratio) != INFINITY) ? log_ratio : -INFINITY;
          const uint4 w4 = philox4x32_10((unsigned)(a.row_base + row0 + r), (unsigned)it >> 2, 0u, TAG_HACC, a.k0, a.k1);
          const unsigned w_ = (it & 2) ? ((it & 1) ? w4.w : w4.z) : ((it & 1) ? w4.y : w4.x);
          acc = logf(u01_open(w_)) < log_ratio;
          flag[r] = acc ? 1.0f : 0.0f;
          if (acc) L.lp[r] = L.lpc[r];
          pa = ok ? expf(fminf(log_ratio, 0.0f)) : 0.0f;
        }
        for (int off = 16; off > 0; off >>= 1) pa += __shfl_xor(pa, off);
        const unsigned cnt = (unsigned)__popcll(__ballot(acc && ok));
        if (threadIdx.x == 0) {
          if (a.acc_prob_sum) atomicAdd(a.acc_prob_sum + it, (double)pa);
          if (a.acc_count) atomicAdd(a.acc_count + it, cnt);
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS)
        if (flag[i / q] != 0.0f) { L.zs[i] = L.zc[i]; L.gr[i] = L.gc[i]; }
      __syncthreads();
      if (a.draws && it >= a.burn_in)
        for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
          const long long gr = row0 + i / q;
          if (gr < n) a.draws[((long long)(it - a.burn_in) * n + gr) * q + i % q] = L.zs[i];
        }
    }
    for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
      const long long gr = row0 + i / q;
      if (gr < n) { a.state[gr * q + i % q] = L.zs[i]; a.grad[gr * q + i % q] = L.gr[i]; }
    }
    if (threadIdx.x < GX_ROWS && row0 + threadIdx.x < n) a.logp[row0 + threadIdx.x] = L.lp[threadIdx.x];
    __syncthreads();
  }
}

// predict_on_posteriors (bgm/base.py:511-525): x ~ N(mu(z_d), sigma^2(z_d)) for every retained draw d; outputs as bgm_predict_kernel
struct GxPredArgs {
  GxBgmModel m;
  const float *draws;
  long long n, row_base;
  int n_draws, burn_in, k_slots;
  const int *slot;
  float *cells, *full, *var_full;
  int add_noise;
  unsigned k0, k1;
};
static __global__ __launch_bounds__(GX_THREADS) void gx_bgm_predict_kernel(GxPredArgs a) {
  extern __shared__ float lds[];
  const GxBgmModel &m = a.m;
  const GxBgmLds L = gx_bgm_carve(lds, m.ld, m.q);
  const int q = m.q, ld = m.ld, T = m.g.L - 1, Hp = m.g.pad[T], P2 = m.g.pad[T + 1], Pp = m.Pp, p = m.p;
  gx_bgm_affine(m, L);
  const long long n = a.n, tiles = (n + GX_ROWS - 1) / GX_ROWS, work = tiles * a.n_draws;
  const float *Wh = m.pack + m.g.w[T], *bh = m.pack + m.g.b[T];
  for (long long wk = blockIdx.x; wk < work; wk += gridDim.x) {
    const long long t = wk / a.n_draws; const int d = (int)(wk - t * a.n_draws);
    const long long row0 = t * GX_ROWS;
    for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
      long long gr = row0 + i / q; gr = gr < n ? gr : n - 1;
      L.zs[i] = a.draws[((long long)d * n + gr) * q + i % q];
    }
    __syncthreads();
    float *cur = gx_bgm_trunk<false>(m, L, L.zs);
    float *Mb = (cur == L.B0) ? L.B1 : L.B0, *Sb = L.B3;
    for (int c0 = 0; c0 < Pp; c0 += m.ch) {
      const int ch = min(m.ch, Pp - c0);
      gx_dense_ld(Wh + c0, P2, Hp, ch, cur, ld, GxStore<false>{Mb, ld, bh + c0});
      gx_dense_ld(Wh + Pp + c0, P2, Hp, ch, cur, ld, GxStore<false>{Sb, ld, bh + Pp + c0});
      __syncthreads();
      for (int i = threadIdx.x; i < GX_ROWS * (ch >> 2); i += GX_THREADS) {      // four columns = one Philox call (sequential layout, tag 6)
        const int r = i / (ch >> 2), c4 = (i - r * (ch >> 2)) << 2;
        const long long gr = row0 + r;
        if (gr >= n || c0 + c4 >= p) continue;
        f32x4 nz = {0.0f, 0.0f, 0.0f, 0.0f};
        if (a.add_noise) nz = box_muller4(philox4x32_10((unsigned)(a.row_base + gr), (unsigned)(a.burn_in + d), (unsigned)((c0 + c4) >> 2), TAG_XNOISE, a.k0, a.k1));
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int c = c0 + c4 + w;
          if (c >= p) break;
          const float mu = Mb[r * ld + c4 + w], s2 = softplus_f(Sb[r * ld + c4 + w]) + BGM_EPS;
          const float xv = a.add_noise ? fmaf(__builtin_sqrtf(s2), nz[w], mu) : mu;
          const long long e = ((long long)d * n + gr) * p + c;
          if (a.full) a.full[e] = xv;
          if (a.var_full) a.var_full[e] = s2;
          if (a.cells) {
            const int sl = a.slot[gr * p + c];
            if (sl >= 0) a.cells[(gr * a.k_slots + sl) * (long long)a.n_draws + d] = xv;
          }
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// fit: forward / backward of one minibatch with the training-mode input BatchNorm (statistics from bgm_bn_stats_kernel)
// ---------------------------------------------------------------------------------------------------------------------------
struct GxBgmFitArgs {
  GxBgmModel m;
  long long act[GX_MAXL], dy[GX_MAXL];     // workspace offsets: layer inputs / pre-activation gradients (dy[L-1]: the heads' [B][2 Pp])
  long long zhat, dzn;                     // [B][KQ]
  float *ws;
  const float *x, *data_z;
  const int *idx;
  int B; float inv_B;
  const float *bn;                         // [4 KQ]: mu_B | inv_std | gamma | beta (bgm_bn_stats_kernel)
  double *loss;                            // [2] += sum loss_x, sum |x - mu|^2
};
static __global__ __launch_bounds__(GX_THREADS) void gx_bgm_fit_kernel(GxBgmFitArgs a) {
  extern __shared__ float lds[];
  const GxBgmModel &m = a.m;
  const int q = m.q, ld = m.ld, Ln = m.g.L, T = Ln - 1, KQ = m.g.pad[0], P2 = m.g.pad[Ln], Pp = m.Pp, p = m.p;
  float *bufA = lds, *bufB = bufA + GX_ROWS * ld;
  long long *rowg = reinterpret_cast<long long *>(bufB + GX_ROWS * ld);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles = (a.B + GX_ROWS - 1) / GX_ROWS;
  double accl = 0.0, accs = 0.0;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long b0 = (long long)t * GX_ROWS;
    __syncthreads();
    if (threadIdx.x < GX_ROWS) { long long b = b0 + threadIdx.x; b = b < a.B ? b : a.B - 1; rowg[threadIdx.x] = a.idx[b]; }
    __syncthreads();
    {
      float *A0 = a.ws + a.act[0] + b0 * KQ, *ZH = a.ws + a.zhat + b0 * KQ;
      for (int i = threadIdx.x; i < GX_ROWS * KQ; i += GX_THREADS) {
        const int r = i / KQ, c = i - r * KQ;
        float zh = 0.0f, zn = 0.0f;
        if (c < q) { zh = (a.data_z[rowg[r] * q + c] - a.bn[c]) * a.bn[KQ + c]; zn = fmaf(zh, a.bn[2 * KQ + c], a.bn[3 * KQ + c]); }
        bufA[r * ld + c] = zn; A0[(long long)r * KQ + c] = zn; ZH[(long long)r * KQ + c] = zh;
      }
    }
    __syncthreads();
    float *cur = bufA, *oth = bufB;
    for (int l = 0; l < T; ++l) {
      gx_dense(m.pack + m.g.w[l], m.g.pad[l], m.g.pad[l + 1], cur, ld,
               GxStoreWs<true>{oth, ld, m.pack + m.g.b[l], a.ws + a.act[l + 1] + b0 * m.g.pad[l + 1], m.g.pad[l + 1]});
      __syncthreads();
      float *tt = cur; cur = oth; oth = tt;
    }
    float *out = a.ws + a.dy[T] + b0 * P2;
    gx_dense(m.pack + m.g.w[T], m.g.pad[T], P2, cur, ld, GxStoreWs<false>{nullptr, 0, m.pack + m.g.b[T], out, P2});
    __syncthreads();
    // losses (bgm/base.py:148-153) and their derivatives over the raw head outputs, in place
    for (int r = wave; r < GX_ROWS; r += GX_WAVES) {
      const bool ok = b0 + r < a.B;
      float *o = out + (long long)r * P2;
      const float *xr = a.x + rowg[r] * (long long)p;
      float lsum = 0.0f, ssum = 0.0f;
      for (int c = lane; c < p; c += 64) {
        const float mu = o[c], sraw = o[Pp + c];
        const float s2 = softplus_acc(sraw) + BGM_EPS, d = xr[c] - mu;
        lsum += d * d / (2.0f * s2) + 0.5f * logf(s2); ssum = fmaf(d, d, ssum);
        o[c] = ok ? -d / s2 * a.inv_B : 0.0f;
        o[Pp + c] = ok ? (-d * d / (2.0f * s2 * s2) + 0.5f / s2) * sigmoid_f(sraw) * a.inv_B : 0.0f;
      }
      for (int off = 32; off; off >>= 1) { lsum += __shfl_xor(lsum, off); ssum += __shfl_xor(ssum, off); }
      if (lane == 0 && ok) { accl += (double)lsum; accs += (double)ssum; }
    }
    __syncthreads();
    for (int l = T; l >= 1; --l) {
      const GxBackStore epi{oth, ld, a.ws + a.act[l] + b0 * m.g.pad[l], m.g.pad[l], a.ws + a.dy[l - 1] + b0 * m.g.pad[l], m.g.pad[l]};
      if (l == T) gx_dense<true>(m.packT + m.g.wt[l], P2, m.g.pad[l], out, P2, epi);
      else gx_dense(m.packT + m.g.wt[l], m.g.pad[l + 1], m.g.pad[l], cur, ld, epi);
      __syncthreads();
      float *tt = cur; cur = oth; oth = tt;
    }
    gx_dense(m.packT + m.g.wt[0], m.g.pad[1], KQ, cur, ld, GxRawStore{a.ws + a.dzn + b0 * KQ, KQ});
    __syncthreads();
  }
  if (a.loss) {
    // lane 0 of every wave holds that wave's rows
    if (lane == 0) { atomicAdd(a.loss, accl); atomicAdd(a.loss + 1, accs); }
  }
}
