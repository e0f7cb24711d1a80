// gx_flipout.h -- DenseFlipout layers on the general-width engine (gx_device.h), and on them the HMC sampler of BGM with the
// Bayesian generator in the shipped noise mode (params['bnn_mcmc_noise'] = 'frozen': ONE weight perturbation and one sign string per
// row for a whole run, DESIGN_HISTORY.md section 7b).
//
// replaces: BGM.tfp_mcmc_sampler bgm/base.py:709-830 on the target bgm/base.py:665-705 with g_net = BayesianVariationalNet
// (networks/bnn.py:40-99; tfp.layers.DenseFlipout restated in oracle/bnn.py, oracle/bgm_bnn.py) -- the job of bgmb_hmc_kernel
// (bgmb_kernels.h), which carries every activation through an HBM workspace (0.13 - 0.18 of the fp32-MFMA peak).  Here a tile of 32
// chains keeps its activations in LDS, the posterior means `loc` and the perturbation dW = sigma * eps are two padded packs in L2
// read by ONE pass of the matrix pipe per layer:
//     y = h loc + ((h * s_in) dW) * s_out + b        lane: acc_loc += a * w_loc,  acc_dw += (a ^ sign bits) * w_dw
// (the sign flip of the A fragment is an XOR with four bits of the row's sign word, the output signs are applied to acc_dw in the
// epilogue), and the backward products d loc^T + ((d * s_out) dW^T) * s_in run through the same routine on the transposed packs.
// The sign words of the tile's rows (Philox, oracle/bnn.py `draw_noise`) are drawn once per tile: frozen noise.
#pragma once
#include "bnn_kernels.h"
#include "gx_bgm_kernels.h"

// Y = A Wl + ((A * s_k) Wd) * s_n for the workgroup's 32 rows: the two-pack form of gx_dense_ld.  sg: the rows' sign words in LDS
// [32][sw]; K-side signs start at word kbase (bit k of the layer side = feature k), N-side signs at word nbase.
template <class Epi>
__device__ __forceinline__ void gx_dense2_ld(const float *__restrict__ Wl, const float *__restrict__ Wd, int ldw, int K, int N, const float *A,
                                             int lda, const unsigned *sg, int sw, int kbase, int nbase, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int units = 2 * (N >> 5);
  for (int u = wave; u < units; u += GX_WAVES) {
    const int rt = u & 1, n0 = (u >> 1) << 5;
    f32x4 l0 = {0.0f, 0.0f, 0.0f, 0.0f}, l1 = l0, d0 = l0, d1 = l0;
    const float *ap = A + (size_t)(16 * rt + j) * lda + 4 * g;
    const unsigned *sk = sg + (16 * rt + j) * sw + kbase;
    const size_t wo = (size_t)(4 * g) * ldw + n0 + 2 * j;
    const float *wl = Wl + wo, *wd = Wd + wo;
    for (int k0 = 0; k0 < K; k0 += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4 *>(ap + k0);
      const unsigned bits = sk[(k0 + 4 * g) >> 5] >> ((k0 + 4 * g) & 31);
      f32x2 bl[4], bd[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bl[s] = *reinterpret_cast<const f32x2 *>(wl + (size_t)s * ldw);
        bd[s] = *reinterpret_cast<const f32x2 *>(wd + (size_t)s * ldw);
      }
      wl += (size_t)16 * ldw; wd += (size_t)16 * ldw;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float av = a[s];      // (a scalar copy: hipcc 7.2 reads element 0 for __builtin_bit_cast of a vector-element lvalue)
        const float as = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, av) ^ (((bits >> s) & 1u) << 31));
        l0 = BGM_MFMA(av, bl[s][0], l0); l1 = BGM_MFMA(av, bl[s][1], l1);
        d0 = BGM_MFMA(as, bd[s][0], d0); d1 = BGM_MFMA(as, bd[s][1], d1);
      }
    }
    // output signs of columns n0 + 2j, n0 + 2j + 1 for the lane's four rows
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned w = sg[(16 * rt + 4 * g + r) * sw + nbase + ((n0 + 2 * j) >> 5)] >> ((n0 + 2 * j) & 31);
      l0[r] += (w & 1u) ? -d0[r] : d0[r];
      l1[r] += (w & 2u) ? -d1[r] : d1[r];
    }
    epi(rt, n0, l0, l1);
  }
}

struct GxfModel {
  GxBgmModel m;                 // pack / packT = the posterior means loc (and the biases); bnp = gamma | beta | moving mean | moving variance
  const float *dw, *dwT;        // the perturbation packs, same offsets as pack / packT
  int sin_w[GX_MAXL + 1], sout_w[GX_MAXL + 1];    // sign-word offsets of the trunk layers 0 .. T-1, the mean head (T) and the variance head (T + 1)
  int swords;                   // sign words per row (multiple of 4)
};

// log p(z | x_obs) + const and d logp / d z for the tile's rows under the frozen perturbation; sg: the rows' sign words
__device__ __forceinline__ void gxf_logp_grad(const GxfModel &f, const GxBgmLds &L, const unsigned *sg, const float *z, const float *x,
                                              long long row0, long long n, float *lp, float *grad) {
  const GxBgmModel &m = f.m;
  const int q = m.q, ld = m.ld, T = m.g.L - 1, Hp = m.g.pad[T], P2 = m.g.pad[T + 1], Pp = m.Pp, p = m.p, sw = f.swords;
  for (int i = threadIdx.x; i < GX_ROWS * m.g.pad[0]; i += GX_THREADS) {
    const int r = i / m.g.pad[0], c = i - r * m.g.pad[0];
    L.B0[r * ld + c] = c < q ? fmaf(z[r * q + c], L.sc[c], L.sh[c]) : 0.0f;
  }
  __syncthreads();
  float *cur = L.B0, *oth = L.B1;
  for (int l = 0; l < T; ++l) {
    gx_dense2_ld(m.pack + m.g.w[l], f.dw + m.g.w[l], m.g.pad[l + 1], m.g.pad[l], m.g.pad[l + 1], cur, ld, sg, sw, f.sin_w[l], f.sout_w[l],
                 GxStoreMask{oth, ld, m.pack + m.g.b[l], L.mask + m.moff[l], m.g.pad[l + 1] >> 1});
    __syncthreads();
    float *t = cur; cur = oth; oth = t;
  }
  float *Mb = (cur == L.B0) ? L.B1 : L.B0, *Sb = L.B3, *dH = L.B2;
  const int wh = m.g.w[T], wth = m.g.wt[T];
  const float *bh = m.pack + m.g.b[T];
  const int row = threadIdx.x >> 3, sub = threadIdx.x & 7;
  long long gr_ = row0 + row; gr_ = gr_ < n ? gr_ : n - 1;
  const float *xr = x + gr_ * (long long)p;
  float ll = 0.0f;
  for (int c0 = 0; c0 < Pp; c0 += m.ch) {
    const int ch = min(m.ch, Pp - c0);
    gx_dense2_ld(m.pack + wh + c0, f.dw + wh + c0, P2, Hp, ch, cur, ld, sg, sw, f.sin_w[T], f.sout_w[T] + (c0 >> 5), GxStore<false>{Mb, ld, bh + c0});
    gx_dense2_ld(m.pack + wh + Pp + c0, f.dw + wh + Pp + c0, P2, Hp, ch, cur, ld, sg, sw, f.sin_w[T + 1], f.sout_w[T + 1] + (c0 >> 5),
                 GxStore<false>{Sb, ld, bh + Pp + c0});
    __syncthreads();
    for (int c = sub; c < ch; c += 8) {
      const float xv = (c0 + c < p) ? xr[c0 + c] : __builtin_nanf("");
      float dm = 0.0f, ds = 0.0f;
      if (xv == xv) {
        const float mu = Mb[row * ld + c], sraw = Sb[row * ld + c];
        const float s2 = softplus_f(sraw) + BGM_EPS, is2 = fast_rcp(s2), d = xv - mu;
        ll += 0.5f * (d * d * is2 + fast_log(s2));
        dm = d * is2; ds = (0.5f * d * d * is2 * is2 - 0.5f * is2) * fast_rcp(1.0f + fast_exp(-sraw));
      }
      Mb[row * ld + c] = dm; Sb[row * ld + c] = ds;
    }
    __syncthreads();
    gx_dense2_ld(m.packT + wth + (size_t)c0 * Hp, f.dwT + wth + (size_t)c0 * Hp, Hp, ch, Hp, Mb, ld, sg, sw, f.sout_w[T] + (c0 >> 5), f.sin_w[T],
                 GxAccum{dH, ld, c0 == 0});
    gx_dense2_ld(m.packT + wth + (size_t)(Pp + c0) * Hp, f.dwT + wth + (size_t)(Pp + c0) * Hp, Hp, ch, Hp, Sb, ld, sg, sw, f.sout_w[T + 1] + (c0 >> 5),
                 f.sin_w[T + 1], GxAccum{dH, ld, false});
    __syncthreads();
  }
  ll += __shfl_xor(ll, 1); ll += __shfl_xor(ll, 2); ll += __shfl_xor(ll, 4);
  if (sub == 0) {
    float zz = 0.0f;
    for (int c = 0; c < q; ++c) zz = fmaf(z[row * q + c], z[row * q + c], zz);
    lp[row] = -(ll + 0.5f * zz);
  }
  const unsigned char *mk = L.mask + m.moff[T - 1];
  for (int i = threadIdx.x; i < GX_ROWS * Hp; i += GX_THREADS) {
    const int r = i / Hp, c = i - r * Hp;
    if (!((mk[r * (Hp >> 1) + (c >> 1)] >> (c & 1)) & 1u)) dH[r * ld + c] *= BGM_LEAK;
  }
  __syncthreads();
  float *A = dH, *o1 = Mb;
  for (int l = T - 1; l >= 1; --l) {
    gx_dense2_ld(m.packT + m.g.wt[l], f.dwT + m.g.wt[l], m.g.pad[l], m.g.pad[l + 1], m.g.pad[l], A, ld, sg, sw, f.sout_w[l], f.sin_w[l],
                 GxMaskBack{o1, ld, L.mask + m.moff[l - 1], m.g.pad[l] >> 1});
    __syncthreads();
    A = o1; o1 = (A == Mb) ? Sb : Mb;
  }
  gx_dense2_ld(m.packT + m.g.wt[0], f.dwT + m.g.wt[0], m.g.pad[0], m.g.pad[1], m.g.pad[0], A, ld, sg, sw, f.sout_w[0], f.sin_w[0], GxRawStore{o1, ld});
  __syncthreads();
  for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
    const int r = i / q, c = i - r * q;
    grad[i] = fmaf(o1[r * ld + c], L.sc[c], -z[i]);
  }
  __syncthreads();
}

struct GxfHmcArgs {
  GxfModel f;
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

// HMC transitions of 32-chain tiles on the frozen-noise target (oracle/bgm_bnn.py hmc_sampler(frozen=True)): the transition logic of
// gx_bgm_hmc_kernel, the target evaluated by gxf_logp_grad.  Dynamic LDS: gx_bgm_lds_bytes(...) + 32 * swords * 4 bytes of sign words.
static __global__ __launch_bounds__(GX_THREADS) void gxf_bgm_hmc_kernel(GxfHmcArgs a) {
  extern __shared__ float lds[];
  const GxfModel &f = a.f;
  const GxBgmModel &m = f.m;
  const GxBgmLds L = gx_bgm_carve(lds, m.ld, m.q);
  unsigned *sg = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(lds) + gx_bgm_lds_bytes(m.ld, m.q, m.mask_bytes));
  const int q = m.q, ncall = (q + 15) >> 4, sw = f.swords;
  gx_bgm_affine(m, L);
  const long long n = a.n, tiles = (n + GX_ROWS - 1) / GX_ROWS;
  float *ke0 = L.red, *flag = L.red + 64;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long row0 = t * GX_ROWS;
    // sign words of the tile's rows: word w of row r = Philox(ctr = (r, w >> 2 | net 0 << 16, stream 0, TAG_SIGN))[w & 3], r = GLOBAL row
    for (int i = threadIdx.x; i < GX_ROWS * (sw >> 2); i += GX_THREADS) {
      const int r = i / (sw >> 2), c = i - r * (sw >> 2);
      const uint4 w4 = philox4x32_10((unsigned)(a.row_base + row0 + r), (unsigned)c, 0u, BNN_TAG_SIGN, a.k0, a.k1);
      unsigned *d = sg + r * sw + 4 * c;
      d[0] = w4.x; d[1] = w4.y; d[2] = w4.z; d[3] = w4.w;
    }
    if (a.init) {
      for (int i = threadIdx.x; i < GX_ROWS * 4 * ncall; i += GX_THREADS) {
        const int r = i / (4 * ncall), c = i - r * 4 * ncall, g = c & 3, tt = c >> 2;
        const f32x4 nz = box_muller4(philox4x32_10((unsigned)(a.row_base + row0 + r), 0u, (unsigned)(g + 4 * tt), TAG_INIT, a.k0, a.k1));
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int ff = 16 * tt + 4 * w + g; if (ff < q) L.zs[r * q + ff] = nz[w]; }
      }
      __syncthreads();
      gxf_logp_grad(f, L, sg, L.zs, a.x, row0, n, L.lp, L.gr);
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
      for (int i = threadIdx.x; i < GX_ROWS * 4 * ncall; i += GX_THREADS) {
        const int r = i / (4 * ncall), c = i - r * 4 * ncall, g = c & 3, tt = c >> 2;
        const f32x4 nz = box_muller4(philox4x32_10((unsigned)(a.row_base + row0 + r), (unsigned)it, (unsigned)(g + 4 * tt), TAG_MOM, a.k0, a.k1));
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int ff = 16 * tt + 4 * w + g; if (ff < q) L.mom[r * q + ff] = nz[w]; }
      }
      __syncthreads();
      if (threadIdx.x < GX_ROWS) {
        float s = 0.0f;
        for (int c = 0; c < q; ++c) s = fmaf(L.mom[threadIdx.x * q + c], L.mom[threadIdx.x * q + c], s);
        ke0[threadIdx.x] = s;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) { L.mom[i] = L.mom[i] + 0.5f * eps * L.gr[i]; L.zc[i] = L.zs[i]; }
      __syncthreads();
      for (int l = 0; l < a.n_leapfrog; ++l) {
        for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) L.zc[i] = fmaf(eps, L.mom[i], L.zc[i]);
        __syncthreads();
        gxf_logp_grad(f, L, sg, L.zc, a.x, row0, n, L.lpc, L.gc);
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
          float lr = -((-L.lpc[r] + 0.5f * s) - (-L.lp[r] + 0.5f * ke0[r]));
          lr = (lr == lr && fabsf(lr) != INFINITY) ? lr : -INFINITY;
          const uint4 w4 = philox4x32_10((unsigned)(a.row_base + row0 + r), (unsigned)it >> 2, 0u, TAG_HACC, a.k0, a.k1);
          const unsigned w_ = (it & 2) ? ((it & 1) ? w4.w : w4.z) : ((it & 1) ? w4.y : w4.x);
          acc = logf(u01_open(w_)) < lr;
          flag[r] = acc ? 1.0f : 0.0f;
          if (acc) L.lp[r] = L.lpc[r];
          pa = ok ? expf(fminf(lr, 0.0f)) : 0.0f;
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

// Packs of one BayesianVariationalNet from the session's parameter vector (gamma | beta | moving mean | moving variance, then per
// Flipout layer loc [in x out], rho [in x out], bias [out]; layers = trunk ..., mean head, variance head): posterior means and
// biases -> pack / packT; the perturbation dW = (eps32 + softplus(rho)) * N(0, 1) of the run's generator call, drawn by
// bgmb_noise_kernel in the parameter vector's order (the same buffer the workspace kernel reads) -> dw / dwT.
struct GxfPackArgs {
  GxNet g;                     // trunk + concatenated heads layer (width 2 Pp)
  int n_flip;                  // Flipout layers of the parameter vector (= g.L + 1)
  int lin[GX_MAXL + 1], lout[GX_MAXL + 1], woff[GX_MAXL + 1], eoff[GX_MAXL + 1];
  int Pp;
  const float *theta;
  const float *dwc;            // the perturbation of the generator call in the parameter vector's own order (bgmb_noise_kernel, slot 0)
  float *pack, *packT, *dw, *dwT;
};
static __global__ __launch_bounds__(256) void gxf_pack_kernel(GxfPackArgs a) {
  const int l = blockIdx.y;                              // Flipout layer of the parameter vector
  const int T = a.g.L - 1;
  const int gl = l < T ? l : T, head = l < T ? 0 : (l - T) * a.Pp;      // layer of the packs, column offset of the head
  const int in = a.lin[l], out = a.lout[l], cnt = in * out;
  const int Kp = a.g.pad[gl], Np = a.g.pad[gl + 1];
  const float *loc = a.theta + a.woff[l], *bias = loc + 2 * cnt, *dwc = a.dwc + a.eoff[l];
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < cnt; idx += gridDim.x * blockDim.x) {
    const int i = idx / out, o = idx - i * out;
    const float w = loc[idx], d = dwc[idx];
    const size_t pf = (size_t)a.g.w[gl] + (size_t)i * Np + head + o, pt = (size_t)a.g.wt[gl] + (size_t)(head + o) * Kp + i;
    a.pack[pf] = w; a.packT[pt] = w; a.dw[pf] = d; a.dwT[pt] = d;
  }
  if (blockIdx.x == 0)
    for (int o = threadIdx.x; o < out; o += blockDim.x) a.pack[a.g.b[gl] + head + o] = bias[o];
}
