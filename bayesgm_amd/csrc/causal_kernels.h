// causal_kernels.h -- CausalBGM log-posterior / Metropolis-Hastings kernels (gfx950).
//
// replaces (reference file:line, all in src/bayesgm/models/causalbgm/base.py):
//   get_log_posterior            :765-817   -> causal_logp<> (device), causal_logpost_kernel
//   metropolis_hastings_sampler  :820-904   -> causal_mh_kernel (persistent over iterations)
//   infer_from_latent_posterior  :671-763   -> fused into causal_mh_kernel (EFFECT != 0)
//
// One wave owns R groups of 16 chains for the whole segment of iterations: its V
// rows (R x NTL x 4 VGPRs), x, y, the chain state and the cached log-posterior
// stay in registers; the g/f/h weights (about 150 KB fp32 at p = 200) stay in
// LDS, shared by the block's waves.  Per transition only the Philox counters
// change; HBM is touched again only to emit retained draws / effects.
#pragma once
#include "bgm_device.h"

struct CausalMeta {
  int q, p;
  int sig_pc, sig_slot; // g's variance column (feature p) sits at element sig_pc of the LAST output tile = padded column sig_slot
  int binary;
  float sig2_v, sig2_x, sig2_y;  // fixed variances (sigma^2) if > 0
  int n_gh;             // number of hidden->hidden 64x64 layers of g
  // LDS blob offsets (floats)
  int w1g, w1f, w1h, b1g, b1f, b1h;
  int wg, bg;           // n_gh consecutive [64][64] / [64]
  int wgl, bgl;         // last g layer [64][16*NTL] / [16*NTL]
  int wf2, bf2, wf3, bf3, wf4, bf4;
  int wh2, bh2, wh3, bh3, wh4, bh4;
  int wxf;              // f L1 x-row, accumulator layout [64]
  int total;            // blob floats
};

struct CausalMhKArgs {
  const float *blob;    // packed weights (global)
  const float *x, *y, *v;
  long long n, row_base;
  float *state, *logp;
  int init, it_begin, n_iters, burn_in;
  float q_sd;
  unsigned k0, k1;
  unsigned *acc_count;
  float *draws;
  int n_keep, sample_y, n_doses;
  const float *x_values;
  float *adrf_partial;
  float *ite;
  unsigned long long *clk;
  const int *seg;           // conditional prior (IdentifiableCausalBGM): segment of every local row, or NULL
  const float *prior_tab;   // [n_segments][q + 2]: mu(u) [q], 1 / sigma^2(u), (q / 2) log sigma^2(u)
  float *eff_cache;         // ADRF kernels with one row tile per wave: [n_slots][ceil(n_doses / 4)][64][2] (mean, sd) of the lane's dose
  int eff_skip;             // 1: a retained iteration in which no chain of the wave moved reuses the cached (mean, sd)
  unsigned long long *eff_stats;   // [0] += retained tile-iterations served from the cache (the total is known on the host), or NULL
  // EFFECT == 3 (causal_event_kernels.h): the retained iterations only append EVENTS -- (chain, first retained draw, state) of every
  // accepted move -- to the wave slot's region; the outcome net runs on dense event tiles afterwards
  float *ev_z;              // [n_slots * ev_cap][q]
  unsigned *ev_meta;        // [n_slots * ev_cap]: (iteration - it_begin) << 4 | chain of the tile
  int *tile_ev;             // [n_tiles][2]: first event of the tile within its slot's region, number of events
  int *slot_cnt;            // [n_slots]: events of the slot
  long long ev_cap;         // events a slot's region holds (a multiple of 16)
  int ev_first;             // 1: every chain emits its state at the launch's first iteration (nothing is carried from an earlier launch)
  CausalMeta m;
};

// ---------------------------------------------------------------------------
// log p(z | x, y, v) for R x 16 chains held by one wave.
//   zin  : L1 input tiles, feature 16 t + 4 r + g  (z features, then x at index q, then 0)
//   vreg : (bias of g's last layer) - (V row), feature 16 t + 4 g + r (load_v_rows); the variance slot holds the bias alone
// Returns logp[rr] replicated over the four lane groups.
// ---------------------------------------------------------------------------
template <int T0, int KT, int NTL, int R>
__device__ __forceinline__ void g_last_groups(const float *wl, const float *bl, int lane_off, int g,
                                              int sig_r, const f32x4 (&in)[R][KT],
                                              const f32x4 (&vreg)[R][NTL], float (&ssq)[R],
                                              float (&sraw)[R]) {
  if constexpr (T0 < NTL) {
    constexpr int GS = group_size(NTL - T0);
    f32x4 acc[R][NTL];  // only [T0, T0+GS) is touched; the rest is dead and never allocated
    constexpr int K_ROWS = 16 * KT;
    constexpr int NKS = 4 * KT;
    const float *base = wl + K_ROWS * 16 * T0 + lane_off * GS;
    bool done = false;
#ifndef BGM_NO_ASM_DENSE
    if constexpr (R == 1 && KT == 4 && GS == 4) {
      dense_group4_k64_asm_c(lds_byte_addr(base), in[0], vreg[0][T0], vreg[0][T0 + 1], vreg[0][T0 + 2], vreg[0][T0 + 3], acc[0][T0],
                             acc[0][T0 + 1], acc[0][T0 + 2], acc[0][T0 + 3]);
      done = true;
    }
#endif
    if (!done) {       // accumulators start from bias - v
#pragma unroll
      for (int u = 0; u < GS; ++u)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) acc[rr][T0 + u] = vreg[rr][T0 + u];
    }
    if constexpr (R == 1 && NKS * GS <= BGM_PREFETCH_ALL_MAX) {   // the single trailing tile: see dense_groups
      AFrag<GS> af[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) af[ks].load(base + (16 * (ks >> 2) + (ks & 3)) * 16 * GS);
      BGM_NO_HOIST();
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int u = 0; u < GS; ++u) acc[0][T0 + u] = BGM_MFMA(af[ks].get(u), in[0][ks >> 2][ks & 3], acc[0][T0 + u]);
      done = true;
    }
    if (!done) {
      AFrag<GS> a_cur, a_nxt;
      a_cur.load(base);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int t = ks >> 2, r = ks & 3;
        if (ks + 1 < NKS) a_nxt.load(base + (16 * ((ks + 1) >> 2) + ((ks + 1) & 3)) * 16 * GS);
#pragma unroll
        for (int u = 0; u < GS; ++u)
#pragma unroll
          for (int rr = 0; rr < R; ++rr)
            acc[rr][T0 + u] = BGM_MFMA(a_cur.get(u), in[rr][t][r], acc[rr][T0 + u]);
        a_cur = a_nxt;
      }
    }
#pragma unroll
    for (int u = 0; u < GS; ++u) {
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float d = acc[rr][T0 + u][r];            // mu - v
          if (T0 + u == NTL - 1) {  // tile holding the variance column (feature p)
            const bool is_sig = (r == sig_r);
            sraw[rr] = is_sig ? acc[rr][T0 + u][r] : sraw[rr];
            d = is_sig ? 0.0f : d;
          }
          ssq[rr] = fmaf(d, d, ssq[rr]);
        }
      }
    }
    g_last_groups<T0 + GS, KT, NTL, R>(wl, bl, lane_off, g, sig_r, in, vreg, ssq, sraw);
  }
}

// f / h tail:  64 -> 32 -> 8 -> 2   (f_units = h_units = [64, 32, 8]); a1 is the
// activated first hidden layer.  Returns the raw outputs (mu, s) of every row group, valid in every lane
// (the sampling blob replicates the two output columns for all lane groups, see above causal_effects).
template <int R>
__device__ __forceinline__ void fh_tail(const float *lds, int w2, int b2, int w3, int b3, int w4,
                                        int b4, int lane_off, int g,
                                        const f32x4 (&a1)[R][4], float (&mu)[R], float (&sr)[R]) {
  f32x4 a2[R][2];
  dense<4, 4, 2, R>(lds + w2, lds + b2, lane_off, g, a1, a2);
  lrelu_s_inplace<2, R>(a2);
  f32x4 a3[R][1];
  dense<2, 4, 1, R>(lds + w3, lds + b3, lane_off, g, a2, a3);
  lrelu_s_inplace<1, R>(a3);
  f32x4 a4[R][1];
  dense<1, 2, 1, R>(lds + w4, lds + b4, lane_off, g, a3, a4);   // two K-steps: see the sampling-blob layout above causal_effects
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    mu[rr] = a4[rr][0][0];
    sr[rr] = a4[rr][0][1];
  }
}

#ifdef BGM_PROF
#define PMARK(i) do { const unsigned long long _t = __builtin_readcyclecounter(); tsec[i] += _t - tlast; tlast = _t; } while (0)
#else
#define PMARK(i)
#endif
template <int KT1, int KSL1, int NTL, int R>
__device__ __forceinline__ void causal_logp(const float *lds, const CausalMeta &m, int lane_off, int g,
                                            int j, const f32x4 (&zin)[R][KT1],
                                            const f32x4 (&vreg)[R][NTL], const float (&xr)[R],
                                            const float (&yr)[R], float (&logp)[R]
#ifdef BGM_PROF
                                            , unsigned long long (&tsec)[8], unsigned long long &tlast
#endif
                                            ) {
  // ---- g : z -> (mu_v, s_v), Gaussian NLL over p covariates (base.py:779-801)
  float ssq[R], sraw_v[R];
  {
    f32x4 h[R][4];
    dense<KT1, KSL1, 4, R>(lds + m.w1g, lds + m.b1g, lane_off, g, zin, h);
    lrelu_s_inplace<4, R>(h);
    PMARK(1);
    bool hidden_done = false;
#ifndef BGM_NO_ASM_HIDDEN4
    if constexpr (R == 1) {
      if (m.n_gh == 4) {   // g_units = [64]*5 (every shipped config): all four hidden layers as one scheduled block
        f32x4 q[4];
        dense_hidden4_asm(lds_byte_addr(lds + m.wg + lane_off * 4), lds_byte_addr(lds + m.bg + 4 * g), h[0], q);
        lrelu_s_inplace<4, 1>(h);
        hidden_done = true;
      }
    }
#endif
    if (!hidden_done) {
      for (int l = 0; l < m.n_gh; ++l) {
        BGM_NO_HOIST();
        f32x4 h2[R][4];
        dense<4, 4, 4, R>(lds + m.wg + l * 4096, lds + m.bg + l * 64, lane_off, g, h, h2);
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[rr][t][r] = lrelu_s(h2[rr][t][r]);
      }
    }
    PMARK(2);
#pragma unroll
    for (int rr = 0; rr < R; ++rr) { ssq[rr] = 0.0f; sraw_v[rr] = 0.0f; }
    const int pc = m.sig_pc;  // position of the variance column inside the last tile (see bgm_g_last_padded)
    g_last_groups<0, 4, NTL, R>(lds + m.wgl, lds + m.bgl, lane_off, g, pc - 4 * g, h, vreg, ssq, sraw_v);
#pragma unroll
    for (int rr = 0; rr < R; ++rr) sraw_v[rr] = __shfl(sraw_v[rr], j + 16 * (pc >> 2));
    PMARK(3);
  }
  // ---- f : (z0, z1, x) -> (mu_y, s_y)   (base.py:793-798)  and  h : (z0, z2) -> (mu_x | logit, s_x)   (:786-791)
  float mu_y[R], sr_y[R], mu_x[R], sr_x[R];
  if constexpr (R == 1) {   // the two small nets in lock step (see dense_pair)
    f32x4 f1[1][4], h1[1][4];
    dense_pair<KT1, KSL1, 4>(lds + m.w1f, lds + m.b1f, lds + m.w1h, lds + m.b1h, lane_off, g, zin, zin, f1, h1);
    lrelu_s_inplace<4, 1>(f1); lrelu_s_inplace<4, 1>(h1);
    f32x4 f2[1][2], h2[1][2];
    dense_pair<4, 4, 2>(lds + m.wf2, lds + m.bf2, lds + m.wh2, lds + m.bh2, lane_off, g, f1, h1, f2, h2);
    lrelu_s_inplace<2, 1>(f2); lrelu_s_inplace<2, 1>(h2);
    f32x4 f3[1][1], h3[1][1];
    dense_pair<2, 4, 1>(lds + m.wf3, lds + m.bf3, lds + m.wh3, lds + m.bh3, lane_off, g, f2, h2, f3, h3);
    lrelu_s_inplace<1, 1>(f3); lrelu_s_inplace<1, 1>(h3);
    f32x4 f4[1][1], h4[1][1];
    dense_pair<1, 2, 1>(lds + m.wf4, lds + m.bf4, lds + m.wh4, lds + m.bh4, lane_off, g, f3, h3, f4, h4);   // two K-steps (permuted layer-3 outputs)
    mu_y[0] = f4[0][0][0]; sr_y[0] = f4[0][0][1];
    mu_x[0] = h4[0][0][0]; sr_x[0] = h4[0][0][1];
  } else {
    {
      f32x4 a1[R][4];
      dense<KT1, KSL1, 4, R>(lds + m.w1f, lds + m.b1f, lane_off, g, zin, a1);
      lrelu_s_inplace<4, R>(a1);
      fh_tail<R>(lds, m.wf2, m.bf2, m.wf3, m.bf3, m.wf4, m.bf4, lane_off, g, a1, mu_y, sr_y);
    }
    {
      f32x4 a1[R][4];
      dense<KT1, KSL1, 4, R>(lds + m.w1h, lds + m.b1h, lane_off, g, zin, a1);
      lrelu_s_inplace<4, R>(a1);
      fh_tail<R>(lds, m.wh2, m.bh2, m.wh3, m.bh3, m.wh4, m.bh4, lane_off, g, a1, mu_x, sr_x);
    }
  }
  PMARK(4);
  // ---- assemble -(loss_v + loss_x + loss_y + |z|^2/2)   (base.py:800-816)
  // the x / y losses are identical in the four lane groups of a row: they are counted once (lane group 0) and
  // folded into the per-lane partial that is summed over g, so one reduction serves all terms.
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    float zsq = 0.0f;
#pragma unroll
    for (int t = 0; t < KT1; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float zz = zin[rr][t][r];
        zsq = (16 * t + 4 * r + g < m.q) ? fmaf(zz, zz, zsq) : zsq;
      }
    const float s2v = (m.sig2_v > 0.0f) ? m.sig2_v : softplus_f(sraw_v[rr]) + BGM_EPS;
    float loss_x;
    if (m.binary) {
      const float l = mu_x[rr];
      const float e = fast_exp(-fabsf(l));
      loss_x = vmax(l, 0.0f) - l * xr[rr] + ((e < 2.44140625e-4f) ? e * (1.0f - 0.5f * e) : fast_log(1.0f + e));
    } else {
      const float s2x = (m.sig2_x > 0.0f) ? m.sig2_x : softplus_f(sr_x[rr]) + BGM_EPS;
      const float dx = xr[rr] - mu_x[rr];
      loss_x = 0.5f * (dx * dx * fast_rcp(s2x) + fast_log(s2x));
    }
    const float s2y = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(sr_y[rr]) + BGM_EPS;
    const float dy = yr[rr] - mu_y[rr];
    const float loss_y = 0.5f * (dy * dy * fast_rcp(s2y) + fast_log(s2y));
    float part = 0.5f * (ssq[rr] * fast_rcp(s2v) + zsq);
    part += (g == 0) ? (loss_x + loss_y) : 0.0f;
    logp[rr] = -(sum_over_g(part) + 0.5f * (float)m.p * fast_log(s2v));
  }
}

// ---- helpers to move rows between HBM and the register layouts ---------------
// raw V rows in the same layout (the encoder's input)
template <int NTL, int R>
__device__ __forceinline__ void load_v_rows(const float *v, long long n, int p, long long row0, int j,
                                            int g, f32x4 (&vreg)[R][NTL]) {
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    long long row = row0 + 16 * rr + j;
    row = row < n ? row : n - 1;
    const float *vr = v + row * (long long)p;
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * t + 4 * g + r;
        vreg[rr][t][r] = (c < p) ? vr[c] : 0.0f;
      }
  }
}

// vreg = (bias of g's last layer, from LDS) - (V row): the last layer accumulates on top of it (g_last_groups)
template <int NTL, int R>
__device__ __forceinline__ void load_v_rows(const float *v, const float *bl, long long n, int p, long long row0, int j,
                                            int g, f32x4 (&vreg)[R][NTL]) {
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    long long row = row0 + 16 * rr + j;
    row = row < n ? row : n - 1;
    const float *vr = v + row * (long long)p;
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * t + 4 * g + r;
        vreg[rr][t][r] = bl[c] - ((c < p) ? vr[c] : 0.0f);
      }
  }
}

// z row (q floats) + x -> L1 input tiles (feature 16 t + 4 r + g)
template <int KT1, int R>
__device__ __forceinline__ void load_z_rows(const float *z, long long n, int q, long long row0, int j,
                                            int g, const float (&xr)[R], f32x4 (&zin)[R][KT1]) {
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    long long row = row0 + 16 * rr + j;
    row = row < n ? row : n - 1;
    const float *zr = z + row * (long long)q;
#pragma unroll
    for (int t = 0; t < KT1; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * t + 4 * r + g;
        zin[rr][t][r] = (f < q) ? zr[f] : (f == q ? xr[rr] : 0.0f);
      }
  }
}

template <int KT1, int R>
__device__ __forceinline__ void store_z_rows(float *z, long long n, int q, long long row0, int j, int g,
                                             const f32x4 (&zin)[R][KT1]) {
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    const long long row = row0 + 16 * rr + j;
    if (row < n) {
      float *zr = z + row * (long long)q;
#pragma unroll
      for (int t = 0; t < KT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * r + g;
          if (f < q) zr[f] = zin[rr][t][r];
        }
    }
  }
}


// ---------------------------------------------------------------------------
// Conditional latent prior of IdentifiableCausalBGM (models/causalbgm/identifiable.py:195-211, 541-551):
// Z | U ~ N(mu(U), sigma^2(U) I) with U the one-hot segment of the row, so -log p(z | u) = |z - mu|^2 / (2 sigma^2) + (q/2) log sigma^2
// replaces |z|^2 / 2.  causal_logp keeps the standard-normal term; kernels instantiated with PRIOR = 1 hold the row's
// (mu, 1/sigma^2, (q/2) log sigma^2) in registers and add the difference to every log-posterior they evaluate.
// ---------------------------------------------------------------------------
template <int KT1>
struct PriorRow {
  f32x4 mu[KT1];      // feature 16 t + 4 r + g, as the L1 input tiles
  float is2, lc;
  __device__ __forceinline__ void load(const int *seg, const float *tab, long long row, int q, int g) {
    const float *t = tab + (long long)seg[row] * (q + 2);
#pragma unroll
    for (int tt = 0; tt < KT1; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * tt + 4 * r + g;
        const float val = t[f < q ? f : q + 1];       // unconditional in-bounds load (the row has q + 2 entries), then mask: a load under
                                                      // the select makes hipcc 7.2 fail with its "$src_shared_base" illegal-instruction error
        mu[tt][r] = (f < q) ? val : 0.0f;
      }
    is2 = t[q];
    lc = t[q + 1];
  }
  // log p_cond(z) - log p_std(z) = |z|^2 / 2 - is2 |z - mu|^2 / 2 - lc      (same value in the four lane groups)
  __device__ __forceinline__ float correction(const f32x4 (&z)[KT1], int q, int g) const {
    float part = 0.0f;
#pragma unroll
    for (int tt = 0; tt < KT1; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float zz = z[tt][r], d = zz - mu[tt][r];
        const float term = 0.5f * (zz * zz - is2 * d * d);
        part += (16 * tt + 4 * r + g < q) ? term : 0.0f;
      }
    return sum_over_g(part) - lc;
  }
};

// ---------------------------------------------------------------------------
// get_log_posterior for n rows (one evaluation)
// ---------------------------------------------------------------------------
template <int KT1, int KSL1, int NTL, int R, int WAVES, int PRIOR = 0>
__global__ __launch_bounds__(64 * WAVES) void causal_logpost_kernel(const float *blob, CausalMeta m,
                                                                    const float *x, const float *y,
                                                                    const float *v, const float *z,
                                                                    long long n, float *out, const int *seg = nullptr,
                                                                    const float *prior_tab = nullptr) {
  static_assert(PRIOR == 0 || R == 1, "conditional prior: one row tile per wave");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lds_fill(lds, blob, m.total);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4, lane_off = 64 * g + j;
  const long long n_tiles = (n + 16 * R - 1) / (16 * R);
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles;
       tile += (long long)gridDim.x * WAVES) {
    BGM_NO_HOIST();
    const long long row0 = tile * 16 * R;
    float xr[R], yr[R], lp[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      long long row = row0 + 16 * rr + j;
      row = row < n ? row : n - 1;
      xr[rr] = x[row];
      yr[rr] = y[row];
    }
    f32x4 vreg[R][NTL];
    load_v_rows<NTL, R>(v, lds + m.bgl, n, m.p, row0, j, g, vreg);
    f32x4 zin[R][KT1];
    load_z_rows<KT1, R>(z, n, m.q, row0, j, g, xr, zin);
#ifdef BGM_PROF
    unsigned long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    causal_logp<KT1, KSL1, NTL, R>(lds, m, lane_off, g, j, zin, vreg, xr, yr, lp, tsec, tlast);
#else
    causal_logp<KT1, KSL1, NTL, R>(lds, m, lane_off, g, j, zin, vreg, xr, yr, lp);
#endif
    if constexpr (PRIOR) {
      PriorRow<KT1> pr;
      const long long rowc = (row0 + j < n) ? row0 + j : n - 1;
      pr.load(seg, prior_tab, rowc, m.q, g);
      lp[0] += pr.correction(zin[0], m.q, g);
    }
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const long long row = row0 + 16 * rr + j;
      if (g == 0 && row < n) out[row] = lp[rr];
    }
  }
}

// ---------------------------------------------------------------------------
// infer_from_latent_posterior (base.py:671-763) for the R x 16 states held by one wave:
// f first layer once at x = 0, doses as rank-1 updates, DB doses per pass.
//   EFFECT 1: adrf_slot[d * n_doses + k] += sum over the wave's valid rows of y_k   (draw-major: the doses of a retained draw share one
//             or two cache lines, so the 4-lane atomics of its passes are one L2 line instead of one line per dose)
//   EFFECT 2: ite[row * n_keep + d] = y(x=1) - y(x=0)
// Sampling-blob layout of the two small nets' tails (causal_scale_blob_kernel): the 8 outputs of layer 3 sit at tile positions
// 4 (f >> 1) + (f & 1) -- accumulator registers r = 0, 1 of every lane group -- so layer 4 contracts over two K-steps instead of
// four, and the two output columns of layer 4 (mu, s) are replicated at positions 4 g' + {0, 1}: every lane holds (mu, s) of its
// row in registers 0 / 1 of the output tile.
// Dose-response sums (EFFECT 1, one row tile per wave): a pass evaluates four doses at once (independent MFMA chains sharing the
// A fragments); afterwards lane group g owns dose e = g of the pass: ONE softplus / sqrt / noise FMA, ONE DPP row reduction and
// ONE 4-lane atomic per pass instead of four of each.  Outcome noise of dose k = word (k & 3) of Philox(row, it, k >> 2,
// TAG_YNOISE) (oracle/causal.py).  Sixteen doses at a time, lane group g evaluates Philox call 4c + g ONCE and pass p uses its
// word p for dose 16c + 4g + p, so the noise is consumed by the lane group that produced it (2 instead of 5 Philox evaluations
// per kept draw at 20 doses); a remainder of fewer than four calls is evaluated by all lane groups (call t, dose 4t + g = word g).
// ---------------------------------------------------------------------------
// value of the lane's own group: v[g]
__device__ __forceinline__ float pick_by_group(int g, float v0, float v1, float v2, float v3) {
  return g == 0 ? v0 : (g == 1 ? v1 : (g == 2 ? v2 : v3));
}

template <int KT1, int KSL1, int R, int EFFECT, bool GROUPING = true, bool CACHE = false, bool EVAL_ONLY = false>
__device__ __forceinline__ void causal_effects(const float *lds, const CausalMeta &m, int lane_off, int g, int j,
                                               int lane, const f32x4 (&zs)[R][KT1], const unsigned (&rowid)[R],
                                               const bool (&valid)[R], long long row0, long long n, unsigned it,
                                               long long d, int n_keep, int sample_y, int n_doses,
                                               const float *x_values, float *adrf_slot, float *ite, unsigned k0,
                                               unsigned k1, float2 *cache = nullptr, float *ite_c = nullptr) {
  // ite_c (EFFECT == 2, one row tile per wave, CACHE): [4] registers of the caller receiving (mean, sd) of the two arms
  BGM_NO_HOIST();
  f32x4 z0in[R][KT1];
#pragma unroll
  for (int rr = 0; rr < R; ++rr)
#pragma unroll
    for (int t = 0; t < KT1; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) z0in[rr][t][r] = (16 * t + 4 * r + g == m.q) ? 0.0f : zs[rr][t][r];
  f32x4 base[R][4];
  dense<KT1, KSL1, 4, R>(lds + m.w1f, lds + m.b1f, lane_off, g, z0in, base);
  f32x4 wx[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wx[t] = *reinterpret_cast<const f32x4 *>(lds + m.wxf + 16 * t + 4 * g);
  const int nd = (EFFECT == 2) ? (CACHE && n_doses == 0 ? 0 : 2) : n_doses;     // (CACHE: n_doses = 0 switches the evaluation off)
  constexpr int DB = (EFFECT == 2) ? 2 : 4;  // doses evaluated per pass (independent MFMA chains)
  constexpr bool GROUPED = (EFFECT == 1 && R == 1 && GROUPING);   // lane group g finishes dose e = g of a pass (see above)
  const int n_calls = (nd + 3) >> 2, n_own = GROUPED ? (n_calls & ~3) : 0;   // Philox calls [0, n_own) in groups of four
  f32x4 nz[R];
#pragma unroll
  for (int rr = 0; rr < R; ++rr) nz[rr] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int kb = 0; kb < n_calls; ++kb) {       // one pass of DB doses per Philox call
    BGM_NO_HOIST();
    const bool own = kb < n_own;
    const int c4 = kb & ~3, p4 = kb & 3;
    if (!EVAL_ONLY && sample_y && (!own || p4 == 0)) {
#pragma unroll
      for (int rr = 0; rr < R; ++rr) nz[rr] = box_muller4(philox4x32_10(rowid[rr], it, (unsigned)(own ? c4 + g : kb), TAG_YNOISE, k0, k1));
    }
    float xk[DB];
#pragma unroll
    for (int e = 0; e < DB; ++e) {
      const int k = own ? 4 * (c4 + e) + p4 : 4 * kb + e;
      xk[e] = (EFFECT == 2) ? (e == 0 ? 1.0f : 0.0f) : x_values[k < nd ? k : nd - 1];
    }
    float mu[DB * R], sr[DB * R];
#ifndef BGM_NO_JIT_EFFECTS
    if constexpr (R == 1) {
      // The dose-specific first-layer activations a1[e] = lrelu_s(base + wx * x_e) are never materialised: element
      // (t, r) is the B operand of K-step 4t + r of the second layer, so it is computed right there, between the
      // MFMAs of the same wave (the two waves of a SIMD share its VALU issue slots as well as its matrix pipe: a
      // separate 768-instruction VALU block per 4 doses cost its full issue time, VALU placed inside the MFMA stream
      // of the same wave is covered by it).  Same for the LeakyReLU feeding layers 3 and 4.
      f32x4 a2[DB][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(lds + m.bf2 + 16 * u + 4 * g);
#pragma unroll
        for (int e = 0; e < DB; ++e) a2[e][u] = b;
      }
      const float *w2 = lds + m.wf2 + lane_off * 2;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int t = s >> 2, r = s & 3;
        AFrag<2> af;
        af.load(w2 + (16 * t + r) * 16 * 2);
        float bop[DB];
#pragma unroll
        for (int e = 0; e < DB; ++e) bop[e] = lrelu_s_pinned(fmaf(wx[t][r], xk[e], base[0][t][r]));
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < DB; ++e) a2[e][u] = BGM_MFMA(af.get(u), bop[e], a2[e][u]);
        // (Forcing MFMA, VALU, VALU, MFMA ... with sched_group_barrier -- with or without software-pipelining the
        // next step's operands and fragment -- produced the intended instruction stream and was 2-3 % SLOWER than
        // leaving the placement to the compiler: measured 5.30-5.36 s vs 5.18 s for the keep phase.)
      }
      f32x4 a3[DB];
      {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(lds + m.bf3 + 4 * g);
#pragma unroll
        for (int e = 0; e < DB; ++e) a3[e] = b;
      }
      const float *w3 = lds + m.wf3 + lane_off;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int t = s >> 2, r = s & 3;
        const float af = w3[(16 * t + r) * 16];
#pragma unroll
        for (int e = 0; e < DB; ++e) a3[e] = BGM_MFMA(af, lrelu_s(a2[e][t][r]), a3[e]);
      }
      f32x4 a4[DB];
      {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(lds + m.bf4 + 4 * g);
#pragma unroll
        for (int e = 0; e < DB; ++e) a4[e] = b;
      }
      const float *w4 = lds + m.wf4 + lane_off;
#pragma unroll
      for (int s = 0; s < 2; ++s) {        // the 8 inputs occupy K-steps 0 and 1 (permuted layer-3 outputs)
        const float af = w4[s * 16];
#pragma unroll
        for (int e = 0; e < DB; ++e) a4[e] = BGM_MFMA(af, lrelu_s(a3[e][s]), a4[e]);
      }
#pragma unroll
      for (int e = 0; e < DB; ++e) { mu[e] = a4[e][0]; sr[e] = a4[e][1]; }
    } else
#endif
    {
      f32x4 a1[DB * R][4];
#pragma unroll
      for (int e = 0; e < DB; ++e)
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) a1[e * R + rr][t][r] = lrelu_s(fmaf(wx[t][r], xk[e], base[rr][t][r]));
      fh_tail<DB * R>(lds, m.wf2, m.bf2, m.wf3, m.bf3, m.wf4, m.bf4, lane_off, g, a1, mu, sr);
    }
    if constexpr (GROUPED) {
      const int k = own ? 4 * (c4 + g) + p4 : 4 * kb + g;          // this lane group's dose of the pass ...
      const float mu_m = pick_by_group(g, mu[0], mu[1], mu[2], mu[3]), sr_m = pick_by_group(g, sr[0], sr[1], sr[2], sr[3]);
      const float s2 = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(sr_m) + BGM_EPS;
      const float sd_m = __builtin_sqrtf(s2);
      if constexpr (CACHE) cache[kb * 64 + lane] = make_float2(mu_m, sd_m);        // for causal_effects_cached below
      if constexpr (EVAL_ONLY) continue;                                             // (causal_event_f_kernel: the pairs are all that is wanted)
      // ... and its noise word: word p4 of its own call (rotated into word 0 pass by pass), or word g of the shared call
      const float noise = own ? nz[0][0] : pick_by_group(g, nz[0][0], nz[0][1], nz[0][2], nz[0][3]);
      if (own) nz[0] = f32x4{nz[0][1], nz[0][2], nz[0][3], nz[0][0]};
      float y = sample_y ? fmaf(sd_m, noise, mu_m) : mu_m;
      y = (valid[0] && k < nd) ? y : 0.0f;
      const float tot = sum_over_j_to_lane15(y);
      if (j == 15 && k < nd) unsafeAtomicAdd(adrf_slot + (long long)d * nd + k, tot);
      continue;
    }
    float yk[DB][R];
#pragma unroll
    for (int e = 0; e < DB; ++e)
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const float s2 = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(sr[e * R + rr]) + BGM_EPS;
        const float sd = __builtin_sqrtf(s2);
        if constexpr (CACHE && EFFECT == 2 && R == 1) { ite_c[2 * e] = mu[e * R + rr]; ite_c[2 * e + 1] = sd; }
        yk[e][rr] = sample_y ? fmaf(sd, nz[rr][e], mu[e * R + rr]) : mu[e * R + rr];
      }
    if constexpr (EVAL_ONLY) continue;             // (causal_event_f_ite_kernel: the two arms' (mean, sd) in ite_c are all that is wanted)
    if constexpr (EFFECT == 1) {
#pragma unroll
      for (int e = 0; e < DB; ++e) {
        float tot = 0.0f;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) tot += valid[rr] ? yk[e][rr] : 0.0f;
        tot = sum_over_j_to_lane15(tot);  // values are valid in lane group 0 -> lane 15
        const int k = 4 * kb + e;
        if (lane == 15 && k < nd) unsafeAtomicAdd(adrf_slot + (long long)d * nd + k, tot);
      }
    } else {
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const long long row = row0 + 16 * rr + j;
        if (g == 0 && row < n) ite[row * (long long)n_keep + d] = yk[0][rr] - yk[1][rr];
      }
    }
  }
}

// The ADRF contribution of a retained iteration in which none of the wave's 16 chains moved.  The outcome net is a deterministic
// function of the chain state, so (mean, sd) of every dose are those of the previous retained iteration -- causal_effects left them in
// `cache` ([pass][lane]: the dose this lane group finishes in that pass) -- and only the outcome noise is new: same Philox calls, same
// fma, same reduction order as the GROUPED path of causal_effects, i.e. bit-identical sums without the 20 dose evaluations.  (With the
// reference's q_sd = 1 a trained model accepts 2-10 % of its proposals: 27-80 % of the tiles see no move in an iteration.)
__device__ __forceinline__ void causal_effects_cached(int g, int j, int lane, unsigned rowid, bool valid, unsigned it, long long d,
                                                      int sample_y, int n_doses, float *adrf_slot, unsigned k0, unsigned k1,
                                                      const float2 *cache) {
  const int nd = n_doses;
  const int n_calls = (nd + 3) >> 2, n_own = n_calls & ~3;
  f32x4 nz = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float2 c = cache[lane];
  for (int kb = 0; kb < n_calls; ++kb) {
    const float2 cn = cache[(kb + 1 < n_calls ? kb + 1 : kb) * 64 + lane];     // next pass's pair under this pass's Philox call
    const bool own = kb < n_own;
    const int c4 = kb & ~3, p4 = kb & 3;
    if (sample_y && (!own || p4 == 0)) nz = box_muller4(philox4x32_10(rowid, it, (unsigned)(own ? c4 + g : kb), TAG_YNOISE, k0, k1));
    const int k = own ? 4 * (c4 + g) + p4 : 4 * kb + g;
    const float noise = own ? nz[0] : pick_by_group(g, nz[0], nz[1], nz[2], nz[3]);
    if (own) nz = f32x4{nz[1], nz[2], nz[3], nz[0]};
    float y = sample_y ? fmaf(c.y, noise, c.x) : c.x;
    y = (valid && k < nd) ? y : 0.0f;
    const float tot = sum_over_j_to_lane15(y);
    if (j == 15 && k < nd) unsafeAtomicAdd(adrf_slot + (long long)d * nd + k, tot);
    c = cn;
  }
}

// The same for the individual treatment effects of a binary treatment (EFFECT == 2): y(1) - y(0) of a retained iteration in which no
// chain of the wave moved, from the (mean, sd) of the two arms kept in registers; the ungrouped noise plan of causal_effects (one
// Philox call, word e for arm e).
__device__ __forceinline__ void causal_ite_cached(int g, int j, unsigned rowid, long long row, long long n, unsigned it, long long d,
                                                  int n_keep, int sample_y, float *ite, unsigned k0, unsigned k1, const float (&c)[4]) {
  f32x4 nz = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  if (sample_y) nz = box_muller4(philox4x32_10(rowid, it, 0u, TAG_YNOISE, k0, k1));
  const float y1 = sample_y ? fmaf(c[1], nz[0], c[0]) : c[0];
  const float y0 = sample_y ? fmaf(c[3], nz[1], c[2]) : c[2];
  if (g == 0 && row < n) ite[row * (long long)n_keep + d] = y1 - y0;
}

// Event form of the retained phase (causal_event_kernels.h): the chains of a wave whose flag `ev` is set append (chain, iteration,
// state) to the wave slot's region, in lane order; ev_cnt is the slot's running event count (wave-uniform).
template <int KT1>
__device__ __forceinline__ void causal_event_append(const CausalMhKArgs &a, int q, long long slot, int &ev_cnt, bool ev, int g, int j,
                                                    const f32x4 (&z)[KT1], int it) {
  const unsigned m16 = (unsigned)(__ballot(ev && g == 0) & 0xFFFFull);
  if (m16 == 0u) return;
  if (ev) {
    const long long e = slot * a.ev_cap + ev_cnt + __popc(m16 & ((1u << j) - 1u));
    float *ez = a.ev_z + e * (long long)q;
#pragma unroll
    for (int t = 0; t < KT1; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * t + 4 * r + g;
        if (f < q) ez[f] = z[t][r];
      }
    if (g == 0) a.ev_meta[e] = ((unsigned)(it - a.it_begin) << 4) | (unsigned)j;
  }
  ev_cnt += __popc(m16);
}

// ---------------------------------------------------------------------------
// Persistent random-walk Metropolis-Hastings over a segment of iterations.
// ---------------------------------------------------------------------------
template <int KT1, int KSL1, int NTL, int R, int WAVES, int EFFECT, int PRIOR = 0>
__global__ __launch_bounds__(64 * WAVES) void causal_mh_kernel(CausalMhKArgs a) {
  static_assert(PRIOR == 0 || R == 1, "conditional prior: one row tile per wave");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const CausalMeta &m = a.m;
  lds_fill(lds, a.blob, m.total);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4, lane_off = 64 * g + j;
  const long long n = a.n;
  const long long n_tiles = (n + 16 * R - 1) / (16 * R);
  const long long slot = (long long)blockIdx.x * WAVES + wave;
  const long long n_slots = (long long)gridDim.x * WAVES;
  const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#ifdef BGM_PROF
  unsigned long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // [WAVES] progress counters after the blob.  An LDS-typed pointer: as a generic one its volatile accesses stay FLAT, and the
  // "{offset, src_shared_base} != null" test that comes with them is what hipcc 7.2 intermittently fails to encode ("$src_shared_base"
  // illegal instruction) when unrelated code in the kernel changes
  volatile __attribute__((address_space(3))) int *prog =
      (volatile __attribute__((address_space(3))) int *)((__attribute__((address_space(3))) float *)lds + m.total);
  if (lane == 0) prog[wave_u] = 0;
  int tiles_done = 0;
  [[maybe_unused]] int ev_cnt = 0;          // EFFECT == 3: events this wave slot has appended

  for (long long tile = slot; tile < n_tiles; tile += n_slots) {
    const long long row0 = tile * 16 * R;
    float xr[R], yr[R], lp[R];
    unsigned rowid[R];
    bool valid[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      long long row = row0 + 16 * rr + j;
      valid[rr] = row < n;
      row = row < n ? row : n - 1;
      xr[rr] = a.x[row];
      yr[rr] = a.y[row];
      rowid[rr] = (unsigned)(a.row_base + row);
    }
    f32x4 vreg[R][NTL];
    load_v_rows<NTL, R>(a.v, lds + m.bgl, n, m.p, row0, j, g, vreg);
    PriorRow<KT1> pr;
    if constexpr (PRIOR) pr.load(a.seg, a.prior_tab, (row0 + j < n) ? row0 + j : n - 1, m.q, g);
    f32x4 zs[R][KT1];
    if (a.init) {
      // current_state ~ N(0,1)  (base.py:842), RNG spec tag 0
#pragma unroll
      for (int rr = 0; rr < R; ++rr)
#pragma unroll
        for (int t = 0; t < KT1; ++t) {
          const f32x4 e = box_muller4(philox4x32_10(rowid[rr], 0u, (unsigned)(g + 4 * t), TAG_INIT, a.k0, a.k1));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = 16 * t + 4 * r + g;
            zs[rr][t][r] = (f < m.q) ? e[r] : (f == m.q ? xr[rr] : 0.0f);
          }
        }
#ifdef BGM_PROF
      { unsigned long long tsec0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast0 = 0;
        causal_logp<KT1, KSL1, NTL, R>(lds, m, lane_off, g, j, zs, vreg, xr, yr, lp, tsec0, tlast0); }
#else
      causal_logp<KT1, KSL1, NTL, R>(lds, m, lane_off, g, j, zs, vreg, xr, yr, lp);
#endif
      if constexpr (PRIOR) lp[0] += pr.correction(zs[0], m.q, g);
    } else {
      load_z_rows<KT1, R>(a.state, n, m.q, row0, j, g, xr, zs);
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        long long row = row0 + 16 * rr + j;
        row = row < n ? row : n - 1;
        lp[rr] = a.logp[row];
      }
    }

    uint4 uacc[R];
    [[maybe_unused]] const int ev_tile0 = ev_cnt;
    bool eff_cached = false;      // the slot's cache holds the outcome-net values of the tile's current states
    unsigned n_eff_skipped = 0u;
    float ite_c[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // EFFECT == 2: (mean, sd) of the two arms at the tile's current states
#ifdef BGM_PROF
    tlast = __builtin_readcyclecounter();
#endif
    for (int it = a.it_begin; it < a.it_begin + a.n_iters; ++it) {
      BGM_NO_HOIST();
      PMARK(7);
      // The two waves that share a SIMD (w, w+4) are arbitrated oldest-first: left alone the older one
      // runs ahead, finishes ~30 % early and leaves the younger wave by itself on the SIMD at low MFMA
      // utilisation (measured: waves 0-3 done at 45 ms, waves 4-7 at 63 ms).  Each wave publishes its
      // progress in LDS and raises its priority only while it is behind its partner.
      // The two waves that share a SIMD (w, w+4) are arbitrated oldest-first: left alone the older one runs
      // ahead, finishes ~30 % early and leaves the younger wave by itself on the SIMD (measured: waves 0-3 done
      // at 45 ms, waves 4-7 at 63 ms).  Each wave publishes its progress in LDS and raises its priority only
      // while it is behind its partner.
      if constexpr (WAVES == 8) {
        const int mine = tiles_done * a.n_iters + (it - a.it_begin);
        if (lane == 0) prog[wave_u] = mine;
        const int other = __builtin_amdgcn_readfirstlane(prog[wave_u ^ 4]);
        if (mine < other) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      }
      // ---- proposal  z' = z + q_sd * eps   (base.py:862)
      f32x4 zp[R][KT1];
#pragma unroll
      for (int rr = 0; rr < R; ++rr)
#pragma unroll
        for (int t = 0; t < KT1; ++t) {
          const f32x4 e = box_muller4(philox4x32_10(rowid[rr], (unsigned)it, (unsigned)(g + 4 * t), TAG_PROP, a.k0, a.k1));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int f = 16 * t + 4 * r + g;
            zp[rr][t][r] = (f < m.q) ? fmaf(a.q_sd, e[r], zs[rr][t][r]) : zs[rr][t][r];
          }
        }
      float lpp[R];
      PMARK(0);
#ifdef BGM_PROF
      causal_logp<KT1, KSL1, NTL, R>(lds, m, lane_off, g, j, zp, vreg, xr, yr, lpp, tsec, tlast);
#else
      causal_logp<KT1, KSL1, NTL, R>(lds, m, lane_off, g, j, zp, vreg, xr, yr, lpp);
#endif
      if constexpr (PRIOR) lpp[0] += pr.correction(zp[0], m.q, g);
      PMARK(5);
      // ---- accept / reject   (base.py:868-871).  u(it) = word (it & 3) of Philox(row, it >> 2, 0, TAG_ACC)
      if ((it & 3) == 0 || it == a.it_begin) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) uacc[rr] = philox4x32_10(rowid[rr], (unsigned)it >> 2, 0u, TAG_ACC, a.k0, a.k1);
      }
      unsigned long long accmask = 0ull;
      [[maybe_unused]] bool moved[R];      // (the same in the four lanes of a chain: lp / lpp are replicated over the lane groups)
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const unsigned w = (it & 2) ? ((it & 1) ? uacc[rr].w : uacc[rr].z) : ((it & 1) ? uacc[rr].y : uacc[rr].x);
        const float u = u01_open(w);
        const float ratio = fast_exp(fminf(lpp[rr] - lp[rr], 0.0f));
        const bool acc = u < ratio;
        moved[rr] = acc;
#pragma unroll
        for (int t = 0; t < KT1; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) zs[rr][t][r] = acc ? zp[rr][t][r] : zs[rr][t][r];
        lp[rr] = acc ? lpp[rr] : lp[rr];
        accmask += __popcll(__ballot(acc && valid[rr] && g == 0));
      }
      // per-(wave slot, iteration) counter, slot-private: a shared per-iteration word would take ~1e8
      // atomics/s on one address and saturate it (measured: -30 % kernel throughput)
      // (fire-and-forget atomic: the read-modify-write form made the wave wait for the load of its own counter every iteration)
      if (a.acc_count != nullptr && lane == 0) atomicAdd(&a.acc_count[slot * (long long)a.n_iters + (it - a.it_begin)], (unsigned)accmask);

      PMARK(6);
      if (it >= a.burn_in) {
        const long long d = it - a.burn_in;
        if (a.draws != nullptr) {  // samples.append(current_state.copy())  (base.py:896)
#pragma unroll
          for (int rr = 0; rr < R; ++rr) {
            const long long row = row0 + 16 * rr + j;
            if (row < n) {
              float *dr = a.draws + (d * n + row) * (long long)m.q;
#pragma unroll
              for (int t = 0; t < KT1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int f = 16 * t + 4 * r + g;
                  if (f < m.q) dr[f] = zs[rr][t][r];
                }
            }
          }
        }
        if constexpr (EFFECT == 3) {
          // event mode: a chain whose proposal was accepted (or every chain at the launch's first iteration, ev_first) appends
          // (chain, iteration, state) to the slot's region, in time order; nothing else happens at a retained iteration
          static_assert(R == 1, "event mode: one row tile per wave");
          causal_event_append<KT1>(a, m.q, slot, ev_cnt, valid[0] && (moved[0] || (a.ev_first && it == a.it_begin)), g, j, zs[0], it);
        } else if constexpr (EFFECT == 1 && R == 1) {
          float2 *cache = reinterpret_cast<float2 *>(a.eff_cache) + slot * (long long)((a.n_doses + 3) >> 2) * 64;
          float *adrf_slot = a.adrf_partial + slot * (long long)a.n_doses * a.n_keep;
          const bool skip = a.eff_skip && eff_cached && accmask == 0ull;                // wave-uniform: nobody moved
          // (the evaluation is switched off through its trip count: its first layer -- 12 MFMAs -- still runs)
          causal_effects<KT1, KSL1, R, EFFECT, true, true>(lds, m, lane_off, g, j, lane, zs, rowid, valid, row0, n, (unsigned)it, d, a.n_keep,
                                                            a.sample_y, skip ? 0 : a.n_doses, a.x_values, adrf_slot, a.ite, a.k0, a.k1, cache);
          if (skip) causal_effects_cached(g, j, lane, rowid[0], valid[0], (unsigned)it, d, a.sample_y, a.n_doses, adrf_slot, a.k0, a.k1, cache);
          eff_cached = true;
          n_eff_skipped += skip ? 1u : 0u;
        } else if constexpr (EFFECT == 2 && R == 1) {
          const bool skip = a.eff_skip && eff_cached && accmask == 0ull;                // wave-uniform: nobody moved
          causal_effects<KT1, KSL1, R, EFFECT, true, true>(lds, m, lane_off, g, j, lane, zs, rowid, valid, row0, n, (unsigned)it, d, a.n_keep,
                                                            a.sample_y, skip ? 0 : 2, a.x_values, a.adrf_partial, a.ite, a.k0, a.k1, nullptr, ite_c);
          if (skip) causal_ite_cached(g, j, rowid[0], row0 + j, n, (unsigned)it, d, a.n_keep, a.sample_y, a.ite, a.k0, a.k1, ite_c);
          eff_cached = true;
          n_eff_skipped += skip ? 1u : 0u;
        } else if constexpr (EFFECT != 0) {
          causal_effects<KT1, KSL1, R, EFFECT>(lds, m, lane_off, g, j, lane, zs, rowid, valid, row0, n, (unsigned)it, d,
                                                a.n_keep, a.sample_y, a.n_doses, a.x_values,
                                                a.adrf_partial + slot * (long long)((EFFECT == 2) ? 2 : a.n_doses) * a.n_keep,
                                                a.ite, a.k0, a.k1);
        }
      }
    }
    if constexpr (EFFECT != 0 && R == 1) {
      if (a.eff_stats != nullptr && lane == 0 && n_eff_skipped != 0u) atomicAdd(&a.eff_stats[0], (unsigned long long)n_eff_skipped);
    }
    if constexpr (EFFECT == 3) {
      if (lane == 0) { a.tile_ev[2 * tile] = ev_tile0; a.tile_ev[2 * tile + 1] = ev_cnt - ev_tile0; }
    }
    // ---- write the chain state back
    store_z_rows<KT1, R>(a.state, n, m.q, row0, j, g, zs);
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const long long row = row0 + 16 * rr + j;
      if (g == 0 && row < n) a.logp[row] = lp[rr];
    }
    ++tiles_done;
  }
  if (lane == 0) prog[wave_u] = 0x7fffffff;   // finished: the partner never needs to catch up
  if constexpr (EFFECT == 3) {
    if (lane == 0) a.slot_cnt[slot] = ev_cnt;
  }
#ifdef BGM_PROF
  if (a.clk != nullptr && lane == 0 && slot == 0) for (int i = 0; i < 8; ++i) a.clk[4ll * gridDim.x * WAVES + i] = tsec[i];
#endif
  if (a.clk != nullptr && lane == 0) {   // [n_slots][4]: cycles, 100 MHz ticks, start tick, XCC id
    a.clk[4 * slot + 0] = __builtin_readcyclecounter() - clk_c0;
    a.clk[4 * slot + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    a.clk[4 * slot + 2] = clk_r0;
    a.clk[4 * slot + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
  }
}

// ---------------------------------------------------------------------------
// evaluate (base.py:534-570): reconstruction errors of g, h, f at the given latent matrix and the
// plug-in causal estimate (ITE per row, or the dose-response curve summed over rows).
// sums[0..2] += sum |v - mu_v|^2, sum (x - x_pred)^2, sum (y - mu_y)^2   (x_pred = sigmoid(logit) if binary)
// ---------------------------------------------------------------------------
struct CausalEvalKArgs {
  const float *blob, *x, *y, *v, *z;
  long long n;
  double *sums;
  const float *x_values;
  int n_doses;
  float *adrf_partial;   // [n_slots][n_doses]
  float *ite;            // [n]
  CausalMeta m;
};

template <int KT1, int KSL1, int NTL, int WAVES, int EFFECT>
__global__ __launch_bounds__(64 * WAVES) void causal_eval_kernel(CausalEvalKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const CausalMeta &m = a.m;
  lds_fill(lds, a.blob, m.total);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4, lane_off = 64 * g + j;
  const long long n = a.n, n_tiles = (n + 15) / 16;
  const long long slot = (long long)blockIdx.x * WAVES + wave;
  double sv = 0.0, sx = 0.0, sy = 0.0;
  for (long long tile = slot; tile < n_tiles; tile += (long long)gridDim.x * WAVES) {
    BGM_NO_HOIST();
    const long long row0 = tile * 16;
    long long row = row0 + j;
    bool valid[1] = {row < n};
    row = row < n ? row : n - 1;
    float xr[1] = {a.x[row]}, yr[1] = {a.y[row]};
    unsigned rowid[1] = {0u};
    f32x4 vreg[1][NTL];
    load_v_rows<NTL, 1>(a.v, lds + m.bgl, n, m.p, row0, j, g, vreg);
    f32x4 zin[1][KT1];
    load_z_rows<KT1, 1>(a.z, n, m.q, row0, j, g, xr, zin);
    float ssq[1] = {0.0f}, sraw[1] = {0.0f};
    {
      f32x4 h[1][4];
      dense<KT1, KSL1, 4, 1>(lds + m.w1g, lds + m.b1g, lane_off, g, zin, h);
      lrelu_s_inplace<4, 1>(h);
      for (int l = 0; l < m.n_gh; ++l) {
        BGM_NO_HOIST();
        f32x4 h2[1][4];
        dense<4, 4, 4, 1>(lds + m.wg + l * 4096, lds + m.bg + l * 64, lane_off, g, h, h2);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) h[0][t][r] = lrelu_s(h2[0][t][r]);
      }
      const int pc = m.sig_pc;
      g_last_groups<0, 4, NTL, 1>(lds + m.wgl, lds + m.bgl, lane_off, g, pc - 4 * g, h, vreg, ssq, sraw);
    }
    float mu_y[1], sr_y[1], mu_x[1], sr_x[1];
    {
      f32x4 a1[1][4];
      dense<KT1, KSL1, 4, 1>(lds + m.w1f, lds + m.b1f, lane_off, g, zin, a1);
      lrelu_s_inplace<4, 1>(a1);
      fh_tail<1>(lds, m.wf2, m.bf2, m.wf3, m.bf3, m.wf4, m.bf4, lane_off, g, a1, mu_y, sr_y);
    }
    {
      f32x4 a1[1][4];
      dense<KT1, KSL1, 4, 1>(lds + m.w1h, lds + m.b1h, lane_off, g, zin, a1);
      lrelu_s_inplace<4, 1>(a1);
      fh_tail<1>(lds, m.wh2, m.bh2, m.wh3, m.bh3, m.wh4, m.bh4, lane_off, g, a1, mu_x, sr_x);
    }
    if (valid[0]) {
      sv += (double)ssq[0];   // every lane group holds its own partial of |v - mu|^2
      if (g == 0) {
        const float xp = m.binary ? 1.0f / (1.0f + expf(-mu_x[0])) : mu_x[0];
        sx += (double)((xr[0] - xp) * (xr[0] - xp));
        sy += (double)((yr[0] - mu_y[0]) * (yr[0] - mu_y[0]));
      }
    }
    if constexpr (EFFECT != 0)
      causal_effects<KT1, KSL1, 1, EFFECT>(lds, m, lane_off, g, j, lane, zin, rowid, valid, row0, n, 0u, 0, 1, 0,
                                           a.n_doses, a.x_values, a.adrf_partial + slot * (long long)a.n_doses,
                                           a.ite, 0u, 0u);
  }
  for (int off = 32; off > 0; off >>= 1) {
    sv += __shfl_xor(sv, off);
    sx += __shfl_xor(sx, off);
    sy += __shfl_xor(sy, off);
  }
  if (lane == 0) {
    atomicAdd(a.sums + 0, sv);
    atomicAdd(a.sums + 1, sx);
    atomicAdd(a.sums + 2, sy);
  }
}

// ---------------------------------------------------------------------------
// infer_from_latent_posterior (base.py:671-763) on a GIVEN tensor of posterior draws [n_keep x n x q]: the same
// effect routine and the same noise counters (iteration = burn_in + d) as the pass fused into the MH kernel, so
// both routes give the same numbers for the same draws.
// ---------------------------------------------------------------------------
struct CausalEffKArgs {
  const float *blob, *x, *draws;
  long long n, row_base;
  int n_keep, burn_in, sample_y, n_doses;
  const float *x_values;
  float *adrf_partial;   // [n_slots][n_keep][n_doses]
  float *ite;            // [n][n_keep]
  unsigned k0, k1;
  CausalMeta m;
};

template <int KT1, int KSL1, int WAVES, int EFFECT>
__global__ __launch_bounds__(64 * WAVES) void causal_effects_kernel(CausalEffKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const CausalMeta &m = a.m;
  lds_fill(lds, a.blob, m.total);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4, lane_off = 64 * g + j;
  const long long n = a.n, n_tiles = (n + 15) / 16;
  const long long slot = (long long)blockIdx.x * WAVES + wave;
  for (long long tile = slot; tile < n_tiles; tile += (long long)gridDim.x * WAVES) {
    const long long row0 = tile * 16;
    long long row = row0 + j;
    bool valid[1] = {row < n};
    row = row < n ? row : n - 1;
    float xr[1] = {a.x[row]};
    unsigned rowid[1] = {(unsigned)(a.row_base + row)};
    for (int d = 0; d < a.n_keep; ++d) {
      BGM_NO_HOIST();
      f32x4 zin[1][KT1];
      load_z_rows<KT1, 1>(a.draws + (long long)d * n * m.q, n, m.q, row0, j, g, xr, zin);
      causal_effects<KT1, KSL1, 1, EFFECT>(lds, m, lane_off, g, j, lane, zin, rowid, valid, row0, n, (unsigned)(a.burn_in + d), d,
                                           a.n_keep, a.sample_y, a.n_doses, a.x_values,
                                           a.adrf_partial + slot * (long long)a.n_doses * a.n_keep /* unused when EFFECT == 2 */, a.ite,
                                           a.k0, a.k1);
    }
  }
}

// acc[it_begin + i] += sum over wave slots of scratch[slot][i].  A block of 1024 threads takes 64 consecutive iterations: thread
// (i & 63, slot group) adds every 16th slot (rows of 256 contiguous bytes), the 16 groups are combined in LDS (integer sums: any order).
static __global__ __launch_bounds__(1024) void acc_reduce_kernel(const unsigned *scratch, int n_slots, int n_iters, unsigned *acc) {
  __shared__ unsigned part[16][64];
  const int ti = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + ti;
  unsigned s = 0;
  if (i < n_iters)
    for (int k = sg; k < n_slots; k += 16) s += scratch[(long long)k * n_iters + i];
  part[sg][ti] = s;
  __syncthreads();
  if (sg == 0 && i < n_iters) {
#pragma unroll
    for (int u = 1; u < 16; ++u) s += part[u][ti];
    acc[i] += s;
  }
}
