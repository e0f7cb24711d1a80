// egm_chain.h -- train_disc_step of the EGM warm start as register-chained row tiles (CausalBGM base.py:305-330,
// Discriminator of models/networks/base.py:338-385 in inference-mode normalisation, the shipped default).
//
// The phase machine of egm_kernels.h spends a workgroup barrier, an operand round trip and a few hundred address
// instructions on every one of the ~80 tiny dense ops of a step (measured: 6 k cycles for a single 64x64 layer on 32 rows,
// 650 k cycles per step).  With fixed normalisation statistics every row of the minibatch is independent until the
// parameter gradients are summed, so here ONE WAVE OWNS A 16-ROW TILE and walks a whole network with the activations in
// registers: the MFMA accumulator of a layer (lane (j, g), tile t, register r = feature 16 t + 4 g + r of row j) is
// directly the B operand of the next layer's K-step (t, r); the A operand of that step is W[16 t + 4 g + r][16 u + i],
// read in place from the canonical Keras array (64 contiguous bytes per lane group, no packing pass).
//
//   waves 0,1  encoder forward on the two row tiles (weights streamed from L2) -> z_ ; then D(z_): forward + backward
//   waves 2,3  D(z) forward + backward, concurrently with the encoder
//   waves 4,5  (after z_ is known) D(zhat): forward, adjoint network, gradient-penalty reverse pass, backward
//   all waves  (after one more barrier) parameter gradients as X^T D GEMMs over the rows stashed in LDS, Adam.
//
// The discriminator's parameters (3 k floats) and their transposes sit in LDS with padded leading dimensions.
// The formulas are those of egm_disc_fwd / egm_disc_bwd / egm_disc_gp in egm_kernels.h with fixed_norm = 1
// (oracle/egm.py); the batch-statistics mode couples the rows and stays on the phase machine.
#pragma once
#include "egm_kernels.h"

#define ECH_WAVES 8
#define ECH_THREADS (64 * ECH_WAVES)
#define ECH_ROLE_WAVES 6          // waves that own a discriminator pass (and a slot of per-wave partial sums)
#ifdef EGM_PHASE_CLOCK
#define ECH_STAMP(k) do { if ((threadIdx.x & 63) == 0) a.stamps[4096 + (threadIdx.x >> 6) * 16 + (k)] = clock64(); } while (0)
#else
#define ECH_STAMP(k) do {} while (0)
#endif

__device__ __forceinline__ float ech_ld(const float *p, int i, int n) {   // p[i] for i < n, else 0; the load itself is unconditional
  const float v = p[i < n ? i : n - 1];
  return i < n ? v : 0.0f;
}

// out[16 u + 4 g + r] (+ bias) = sum_f in[f] W[f * ld + o]: one row tile through one Dense layer.
template <int KT, int NT, bool EXACT>
__device__ __forceinline__ void ech_dense(const float *W, int ld, const float *bias, int n_in, int n_out, const f32x4 (&in)[KT],
                                          f32x4 (&out)[NT], int j, int g) {
  f32x4 bv[NT];
  if (bias) {      // requested first, added last: the MFMAs do not wait for it
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * u + 4 * g + r;
        bv[u][r] = EXACT ? bias[o] : ech_ld(bias, o, n_out);
      }
  }
#pragma unroll
  for (int u = 0; u < NT; ++u) out[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * t + 4 * g + r;
      const bool fok = EXACT || f < n_in;
      const float *wr = W + (EXACT ? f : min(f, n_in - 1)) * ld;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int o = 16 * u + j;
        float a;
        if (EXACT) a = wr[o];
        else {
          const float w = wr[min(o, n_out - 1)];
          a = (fok && o < n_out) ? w : 0.0f;
        }
        out[u] = BGM_MFMA(a, in[t][r], out[u]);
      }
    }
  if (bias) {
#pragma unroll
    for (int u = 0; u < NT; ++u) out[u] += bv[u];
  }
}

// ---------------------------------------------------------------------------------------------
// LDS parameter block of a three-hidden-layer discriminator (offsets in floats from its base)
// ---------------------------------------------------------------------------------------------
struct EchP {
  int d0, d1, d2, d3;
  int ld0, ld1, ld2, lt0, lt1, lt2;     // leading dimensions of W_l (16 NT + 4) and W_l^T (16 KT + 4): lane groups land on distinct banks
  int W0, W1, W2, T0, T1, T2;           // W_l [16 KT x ld_l], W_l^T [16 NT x lt_l], zero-padded to whole tiles (no masks in the passes)
  int b0, b1, b2, ga0, ga1, ga2, be0, be1, be2, wo, bo;
  int total;
};
template <int T1, int T2, int T3, int T0 = 1>
__host__ __device__ inline EchP ech_layout(const EgmDisc &d) {
  EchP P;
  P.d0 = d.dims[0]; P.d1 = d.dims[1]; P.d2 = d.dims[2]; P.d3 = d.dims[3];
  P.ld0 = 16 * T1 + 4; P.ld1 = 16 * T2 + 4; P.ld2 = 16 * T3 + 4;
  P.lt0 = 16 * T0 + 4; P.lt1 = 16 * T1 + 4; P.lt2 = 16 * T2 + 4;
  int o = 0;
  auto take = [&](int n) { const int r = o; o += (n + 3) & ~3; return r; };
  P.W0 = take(16 * T0 * P.ld0); P.W1 = take(16 * T1 * P.ld1); P.W2 = take(16 * T2 * P.ld2);
  P.T0 = take(16 * T1 * P.lt0); P.T1 = take(16 * T2 * P.lt1); P.T2 = take(16 * T3 * P.lt2);
  P.b0 = take(16 * T1); P.b1 = take(16 * T2); P.b2 = take(16 * T3);
  P.ga0 = take(16 * T1); P.ga1 = take(16 * T2); P.ga2 = take(16 * T3);
  P.be0 = take(16 * T1); P.be1 = take(16 * T2); P.be2 = take(16 * T3);
  P.wo = take(16 * T3); P.bo = take(1);
  P.total = o;
  return P;
}
// stash of one pass (floats per row): inputs X_l of the three hidden layers, then their (scaled) pre-activation gradients D_l
template <int T1, int T2, int T3, int T0 = 1> struct EchDims {
  static constexpr int XW = 16 * (T0 + T1 + T2), DW = 16 * (T1 + T2 + T3), SW = XW + DW;
  static constexpr int SL = 16 * (T1 + T2 + T3);            // per-wave partial sums: gamma [SL] | beta [SL] | w_out [16 T3] | scalars [16]
  static constexpr int SLOT = 2 * SL + 16 * T3 + 16;
  static constexpr int TILES = T0 * T1 + T1 * T2 + T2 * T3;      // weight-gradient tiles
};
template <int T1, int T2, int T3, int T0 = 1>
__host__ __device__ inline int ech_disc_lds_floats(const EgmDisc &d, int B) {
  using D = EchDims<T1, T2, T3, T0>;
  return 64 + ech_layout<T1, T2, T3, T0>(d).total + ECH_ROLE_WAVES * D::SLOT + 16 * T0 * B + (T0 > 1 ? 3 : 4) * B * D::SW;
}

template <int T1, int T2, int T3, int T0 = 1>
__device__ __forceinline__ void ech_fill_params(float *par, const EchP &P, const float *th, const EgmDisc &d, int tid, int nthr) {
  const int dims_in[3] = {P.d0, P.d1, P.d2}, dims_out[3] = {P.d1, P.d2, P.d3};
  const int kp[3] = {16 * T0, 16 * T1, 16 * T2}, np_[3] = {16 * T1, 16 * T2, 16 * T3};       // padded extents
  const int ld[3] = {P.ld0, P.ld1, P.ld2}, lt[3] = {P.lt0, P.lt1, P.lt2}, ow[3] = {P.W0, P.W1, P.W2}, ot[3] = {P.T0, P.T1, P.T2};
  const int ob[3] = {P.b0, P.b1, P.b2}, og[3] = {P.ga0, P.ga1, P.ga2}, oe[3] = {P.be0, P.be1, P.be2};
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const int n_in = dims_in[l], n_out = dims_out[l];
    for (int e = tid; e < kp[l] * np_[l]; e += nthr) {
      const int f = e / np_[l], o = e - f * np_[l];
      const float w = th[d.w[l] + min(f, n_in - 1) * n_out + min(o, n_out - 1)];
      const float wz = (f < n_in && o < n_out) ? w : 0.0f;
      par[ow[l] + f * ld[l] + o] = wz;
      par[ot[l] + o * lt[l] + f] = wz;
    }
    for (int o = tid; o < np_[l]; o += nthr) {
      const int oc = min(o, n_out - 1);
      const float bv = th[d.b[l] + oc], gv = th[d.gamma[l] + oc], ev = th[d.beta[l] + oc];
      par[ob[l] + o] = o < n_out ? bv : 0.0f;
      par[og[l] + o] = o < n_out ? gv : 0.0f;
      par[oe[l] + o] = o < n_out ? ev : 0.0f;
    }
  }
  for (int f = tid; f < 16 * T3; f += nthr) { const float w = th[d.w[3] + min(f, P.d3 - 1)]; par[P.wo + f] = f < P.d3 ? w : 0.0f; }
  if (tid == 0) par[P.bo] = th[d.b[3]];
}

// ---------------------------------------------------------------------------------------------
// one discriminator evaluation on a row tile
// ---------------------------------------------------------------------------------------------
template <int T1, int T2, int T3, int T0 = 1>
struct EchFwd {
  f32x4 a0[T0], a1[T1], a2[T2], a3[T3];    // layer inputs / tanh outputs (T0 tiles of input: q <= 16 T0)
  f32x4 u1[T1], u2[T2], u3[T3];            // normalised pre-activations (uhat)
  float out;
};
template <int T1, int T2, int T3, int T0 = 1>
struct EchAcc {                            // per-row contributions to the vector-parameter gradients (summed over rows at the end)
  f32x4 gam1[T1], gam2[T2], gam3[T3], bet1[T1], bet2[T2], bet3[T3], wo[T3];
};
template <int NT>
__device__ __forceinline__ void ech_zero(f32x4 (&x)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) x[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}
template <int T1, int T2, int T3, int T0 = 1>
__device__ __forceinline__ void ech_zero_acc(EchAcc<T1, T2, T3, T0> &acc) {
  ech_zero<T1>(acc.gam1); ech_zero<T2>(acc.gam2); ech_zero<T3>(acc.gam3);
  ech_zero<T1>(acc.bet1); ech_zero<T2>(acc.bet2); ech_zero<T3>(acc.bet3); ech_zero<T3>(acc.wo);
}
__device__ __forceinline__ float ech_c() { return 1.0f / sqrtf(1.0f + EGM_BN_EPS); }   // 1 / sqrt(moving variance 1 + epsilon)

// tanh(x) = 1 - 2 / (exp(2 x) + 1) on the transcendental unit (absolute error ~2e-7; libm's tanhf is ~60 instructions per value)
__device__ __forceinline__ float ech_tanh(float x) { return 1.0f - 2.0f * fast_rcp(fast_exp(2.0f * x) + 1.0f); }

template <int NT>
__device__ __forceinline__ void ech_act(const float *ga, const float *be, int n, int g, const f32x4 (&u)[NT], f32x4 (&uh)[NT],
                                        f32x4 (&a)[NT]) {
  const float c = ech_c();
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * t + 4 * g + r;
      uh[t][r] = u[t][r] * c;
      a[t][r] = ech_tanh(fmaf(uh[t][r], ga[o], be[o]));
    }
}
template <int NT>
__device__ __forceinline__ void ech_stash(float *base, int row, int g, const f32x4 (&x)[NT], float s) {
#pragma unroll
  for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4 *>(base + row * (16 * NT) + 16 * t + 4 * g) = x[t] * s;
}

template <int T1, int T2, int T3, int T0 = 1>
__device__ __forceinline__ void ech_disc_fwd(const float *par, const EchP &P, EchFwd<T1, T2, T3, T0> &F, int j, int g) {
  f32x4 u1[T1], u2[T2], u3[T3];
  ech_dense<T0, T1, true>(par + P.W0, P.ld0, par + P.b0, P.d0, P.d1, F.a0, u1, j, g);
  ech_act<T1>(par + P.ga0, par + P.be0, P.d1, g, u1, F.u1, F.a1);
  ech_dense<T1, T2, true>(par + P.W1, P.ld1, par + P.b1, P.d1, P.d2, F.a1, u2, j, g);
  ech_act<T2>(par + P.ga1, par + P.be1, P.d2, g, u2, F.u2, F.a2);
  ech_dense<T2, T3, true>(par + P.W2, P.ld2, par + P.b2, P.d2, P.d3, F.a2, u3, j, g);
  ech_act<T3>(par + P.ga2, par + P.be2, P.d3, g, u3, F.u3, F.a3);
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < T3; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) s = fmaf(F.a3[t][r], par[P.wo + 16 * t + 4 * g + r], s);
  F.out = sum_over_g(s) + par[P.bo];
}

// dy = da (1 - a^2);  gamma / beta contributions;  du = dy gamma c
template <int NT>
__device__ __forceinline__ void ech_bwd_act(const float *ga, int n, int g, const f32x4 (&da)[NT], const f32x4 (&a)[NT], const f32x4 (&uh)[NT],
                                            float s, f32x4 (&gam)[NT], f32x4 (&bet)[NT], f32x4 (&du)[NT]) {
  const float c = ech_c();
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float av = a[t][r];
      const float dy = da[t][r] * (1.0f - av * av);
      gam[t][r] = fmaf(s * dy, uh[t][r], gam[t][r]);
      bet[t][r] = fmaf(s, dy, bet[t][r]);
      du[t][r] = dy * ga[16 * t + 4 * g + r] * c;
    }
}

// Ordinary backward of one evaluation: dLoss/dout = dout on every row (ADJ = false), or no output gradient but node adjoints
// ab* on the tanh outputs (ADJ = true: the second half of the gradient penalty).  Stashes (a_{l-1}, s du_l) of the three
// hidden layers for the weight-gradient GEMMs of the last phase; `st` = this pass's stash block.
template <int T1, int T2, int T3, bool ADJ, int T0 = 1>
__device__ __forceinline__ void ech_disc_bwd(const float *par, const EchP &P, const EchFwd<T1, T2, T3, T0> &F, float dout, const f32x4 (&ab1)[T1],
                                             const f32x4 (&ab2)[T2], const f32x4 (&ab3)[T3], float s, float *st, int B, int row,
                                             EchAcc<T1, T2, T3, T0> &acc, int j, int g) {
  using D = EchDims<T1, T2, T3, T0>;
  float *X0 = st, *X1 = st + B * 16 * T0, *X2 = st + B * (16 * T0 + 16 * T1);
  float *D0 = st + B * D::XW, *D1 = D0 + B * 16 * T1, *D2 = D1 + B * 16 * T2;
  f32x4 da3[T3], du3[T3];
#pragma unroll
  for (int t = 0; t < T3; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      da3[t][r] = ADJ ? ab3[t][r] : dout * par[P.wo + 16 * t + 4 * g + r];
      if (!ADJ) acc.wo[t][r] = fmaf(dout, F.a3[t][r], acc.wo[t][r]);
    }
  ech_bwd_act<T3>(par + P.ga2, P.d3, g, da3, F.a3, F.u3, s, acc.gam3, acc.bet3, du3);
  ech_stash<T2>(X2, row, g, F.a2, 1.0f);
  ech_stash<T3>(D2, row, g, du3, s);
  f32x4 da2[T2], du2[T2];
  ech_dense<T3, T2, true>(par + P.T2, P.lt2, nullptr, P.d3, P.d2, du3, da2, j, g);
  if (ADJ) {
#pragma unroll
    for (int t = 0; t < T2; ++t) da2[t] += ab2[t];
  }
  ech_bwd_act<T2>(par + P.ga1, P.d2, g, da2, F.a2, F.u2, s, acc.gam2, acc.bet2, du2);
  ech_stash<T1>(X1, row, g, F.a1, 1.0f);
  ech_stash<T2>(D1, row, g, du2, s);
  f32x4 da1[T1], du1[T1];
  ech_dense<T2, T1, true>(par + P.T1, P.lt1, nullptr, P.d2, P.d1, du2, da1, j, g);
  if (ADJ) {
#pragma unroll
    for (int t = 0; t < T1; ++t) da1[t] += ab1[t];
  }
  ech_bwd_act<T1>(par + P.ga0, P.d1, g, da1, F.a1, F.u1, s, acc.gam1, acc.bet1, du1);
  ech_stash<T0>(X0, row, g, F.a0, 1.0f);
  ech_stash<T1>(D0, row, g, du1, s);
}

// adjoint-network layer: du = dA (1 - a^2) gamma c
template <int NT>
__device__ __forceinline__ void ech_adj_act(const float *ga, int g, const f32x4 (&dA)[NT], const f32x4 (&a)[NT], f32x4 (&du)[NT]) {
  const float c = ech_c();
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float av = a[t][r];
      du[t][r] = dA[t][r] * (1.0f - av * av) * ga[16 * t + 4 * g + r] * c;
    }
}
// reverse of an adjoint-network layer: dub = abar_{l-1} W_l;  gamma contribution;  abar on the forward node;  abar_l for the layer above
template <int NT>
__device__ __forceinline__ void ech_rev_act(const float *ga, int g, const f32x4 (&dub)[NT], const f32x4 (&dA)[NT], const f32x4 (&a)[NT], float s,
                                            f32x4 (&gam)[NT], f32x4 (&ab)[NT], f32x4 (&an)[NT]) {
  const float c = ech_c();
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float av = a[t][r], om = 1.0f - av * av;
      const float tt = dub[t][r] * c;
      gam[t][r] = fmaf(s * tt, dA[t][r] * om, gam[t][r]);     // dy of the adjoint network = dA (1 - a^2)
      const float dyb = tt * ga[16 * t + 4 * g + r];
      ab[t][r] = dyb * dA[t][r] * (-2.0f * av);
      an[t][r] = dyb * om;
    }
}

// Gradient penalty on the evaluation F (of zhat): s * d/dtheta mean_b (||d out_b / d input_b|| - 1)^2.  st_rev / st_bwd: stash blocks of
// the reverse pass through the adjoint network and of the backward pass through the forward network.  Returns (||g|| - 1)^2 of the row.
template <int T1, int T2, int T3, int T0 = 1>
__device__ __forceinline__ float ech_disc_gp(const float *par, const EchP &P, const EchFwd<T1, T2, T3, T0> &F, float s, float *st_rev,
                                             float *st_bwd, int B, int row, EchAcc<T1, T2, T3, T0> &acc, int j, int g) {
  using D = EchDims<T1, T2, T3, T0>;
  float *X0 = st_rev, *X1 = st_rev + B * 16 * T0, *X2 = st_rev + B * (16 * T0 + 16 * T1);
  float *D0 = st_rev + B * D::XW, *D1 = D0 + B * 16 * T1, *D2 = D1 + B * 16 * T2;
  // ---- adjoint network: g = d out / d input  (its du_l go straight to the stash: W_l receives abar_{l-1}^T du_l below)
  f32x4 dA3[T3], dA2[T2], dA1[T1], g0[T0];
  {
    f32x4 du3[T3];
#pragma unroll
    for (int t = 0; t < T3; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dA3[t][r] = par[P.wo + 16 * t + 4 * g + r];
    ech_adj_act<T3>(par + P.ga2, g, dA3, F.a3, du3);
    ech_stash<T3>(D2, row, g, du3, s);
    ech_dense<T3, T2, true>(par + P.T2, P.lt2, nullptr, P.d3, P.d2, du3, dA2, j, g);
    f32x4 du2[T2];
    ech_adj_act<T2>(par + P.ga1, g, dA2, F.a2, du2);
    ech_stash<T2>(D1, row, g, du2, s);
    ech_dense<T2, T1, true>(par + P.T1, P.lt1, nullptr, P.d2, P.d1, du2, dA1, j, g);
    f32x4 du1[T1];
    ech_adj_act<T1>(par + P.ga0, g, dA1, F.a1, du1);
    ech_stash<T1>(D0, row, g, du1, s);
    ech_dense<T1, T0, true>(par + P.T0, P.lt0, nullptr, P.d1, P.d0, du1, g0, j, g);
  }
  float n2 = 0.0f;
#pragma unroll
  for (int t = 0; t < T0; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) n2 = fmaf(g0[t][r], g0[t][r], n2);
  n2 = sum_over_g(n2);
  const float nrm = sqrtf(n2);
  const float part = (nrm - 1.0f) * (nrm - 1.0f);
  const float coef = 2.0f * (nrm - 1.0f) / nrm / (float)B;
  // ---- reverse through the adjoint network, bottom to top
  f32x4 ab1[T1], ab2[T2], ab3[T3];
  {
    f32x4 an0[T0];
#pragma unroll
    for (int t = 0; t < T0; ++t) an0[t] = g0[t] * coef;
    ech_stash<T0>(X0, row, g, an0, 1.0f);
    f32x4 dub1[T1], an1[T1];
    ech_dense<T0, T1, true>(par + P.W0, P.ld0, nullptr, P.d0, P.d1, an0, dub1, j, g);
    ech_rev_act<T1>(par + P.ga0, g, dub1, dA1, F.a1, s, acc.gam1, ab1, an1);
    ech_stash<T1>(X1, row, g, an1, 1.0f);
    f32x4 dub2[T2], an2[T2];
    ech_dense<T1, T2, true>(par + P.W1, P.ld1, nullptr, P.d1, P.d2, an1, dub2, j, g);
    ech_rev_act<T2>(par + P.ga1, g, dub2, dA2, F.a2, s, acc.gam2, ab2, an2);
    ech_stash<T2>(X2, row, g, an2, 1.0f);
    f32x4 dub3[T3], an3[T3];
    ech_dense<T2, T3, true>(par + P.W2, P.ld2, nullptr, P.d2, P.d3, an2, dub3, j, g);
    ech_rev_act<T3>(par + P.ga2, g, dub3, dA3, F.a3, s, acc.gam3, ab3, an3);
#pragma unroll
    for (int t = 0; t < T3; ++t) acc.wo[t] += an3[t] * s;      // d out / d a_L = w_out: its adjoint is abar_L summed over rows
  }
  // ---- ... and on through the forward pass
  ech_disc_bwd<T1, T2, T3, true, T0>(par, P, F, 0.0f, ab1, ab2, ab3, s, st_bwd, B, row, acc, j, g);
  return part;
}

// sums over the 16 rows of the tile -> this wave's slot (lane j = 15 of each group holds the totals)
template <int T1, int T2, int T3, int T0 = 1>
__device__ __forceinline__ void ech_write_slot(float *slot, const EchAcc<T1, T2, T3, T0> &acc, float out_signed, float gp_part, int j, int g) {
  using D = EchDims<T1, T2, T3, T0>;
  auto put = [&](float *dst, const f32x4 &v) {
    f32x4 s;
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = sum_over_j_to_lane15(v[r]);
    if (j == 15) *reinterpret_cast<f32x4 *>(dst + 4 * g) = s;
  };
#pragma unroll
  for (int t = 0; t < T1; ++t) { put(slot + 16 * t, acc.gam1[t]); put(slot + D::SL + 16 * t, acc.bet1[t]); }
#pragma unroll
  for (int t = 0; t < T2; ++t) { put(slot + 16 * (T1 + t), acc.gam2[t]); put(slot + D::SL + 16 * (T1 + t), acc.bet2[t]); }
#pragma unroll
  for (int t = 0; t < T3; ++t) {
    put(slot + 16 * (T1 + T2 + t), acc.gam3[t]); put(slot + D::SL + 16 * (T1 + T2 + t), acc.bet3[t]);
    put(slot + 2 * D::SL + 16 * t, acc.wo[t]);
  }
  const float so = sum_over_j_to_lane15(out_signed), sg = sum_over_j_to_lane15(gp_part);
  if (j == 15 && g == 0) { slot[2 * D::SL + 16 * T3] = so; slot[2 * D::SL + 16 * T3 + 1] = sg; }
}

// ---------------------------------------------------------------------------------------------
// encoder forward on one row tile (weights streamed from the canonical array in L2; hidden width 16 HT)
// ---------------------------------------------------------------------------------------------
// The weights come from L2 at best (every launch starts with a cold L1, and the generator step that last wrote them ran on
// another XCD more often than not): every layer's A fragments are requested one layer (first layer: two K tiles) ahead of the
// MFMAs that consume them.
template <int HT>
__device__ __forceinline__ void ech_load_a(const float *W, int ld, float (&av)[4 * HT][HT], int j, int g) {   // A fragments of a [16 HT x 16 HT] layer
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < HT; ++u) av[4 * t + r][u] = W[(16 * t + 4 * g + r) * ld + 16 * u + j];
}
// first layer, K extent known at compile time: the row's inputs are requested up front (one HBM round trip for the whole row
// instead of one per K tile), the weights of K tile t + 2 under the MFMAs of tile t
template <int HT, int KT0>
__device__ __forceinline__ void ech_encoder_l1_k(const float *W, int p, const float *vrow, f32x4 (&h)[HT], int j, int g) {
  constexpr int H = 16 * HT;
  f32x4 xin[KT0];
  const bool vec = (p & 3) == 0 && (reinterpret_cast<unsigned long long>(vrow) & 15ull) == 0;
  if (vec) {
#pragma unroll
    for (int t = 0; t < KT0; ++t) {
      const int f = 16 * t + 4 * g;
      const f32x4 x = *reinterpret_cast<const f32x4 *>(vrow + min(f, p - 4));
      xin[t] = f < p ? x : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  } else {
#pragma unroll
    for (int t = 0; t < KT0; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) xin[t][r] = ech_ld(vrow, 16 * t + 4 * g + r, p);
  }
  float a[3][4][HT];
  auto load = [&](int t, float (&av)[4][HT]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float *wr = W + min(16 * t + 4 * g + r, p - 1) * H + j;
#pragma unroll
      for (int u = 0; u < HT; ++u) av[r][u] = wr[16 * u];
    }
  };
  load(0, a[0]);
  if (KT0 > 1) load(1, a[1]);
#pragma unroll
  for (int t = 0; t < KT0; ++t) {
    if (t + 2 < KT0) load(t + 2, a[(t + 2) % 3]);
    BGM_NO_HOIST();          // pins the issue order: without it the scheduler sinks each load to just above its MFMA
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < HT; ++u) h[u] = BGM_MFMA(a[t % 3][r][u], xin[t][r], h[u]);
    BGM_NO_HOIST();
  }
}
// first layer, any K extent: three K tiles in three fixed register sets (rotating them with moves would make every move wait for
// the load it copies); a tile beyond the input contributes zeros
template <int HT>
__device__ __forceinline__ void ech_encoder_l1(const float *W, int p, const float *vrow, f32x4 (&h)[HT], int j, int g) {
  constexpr int H = 16 * HT;
  const int KT = (p + 15) >> 4;
  float a0[4][HT], x0[4], a1[4][HT], x1[4], a2[4][HT], x2[4];
  auto load = [&](int t, float (&av)[4][HT], float (&xv)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * t + 4 * g + r;
      const int fc = min(f, p - 1);
      const float x = vrow[fc];
      xv[r] = f < p ? x : 0.0f;
#pragma unroll
      for (int u = 0; u < HT; ++u) av[r][u] = W[fc * H + 16 * u + j];
    }
  };
  auto mm = [&](const float (&av)[4][HT], const float (&xv)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < HT; ++u) h[u] = BGM_MFMA(av[r][u], xv[r], h[u]);
  };
  load(0, a0, x0);
  load(1, a1, x1);
  for (int t = 0; t < KT; t += 3) {
    BGM_NO_HOIST();
    load(t + 2, a2, x2);
    BGM_NO_HOIST();
    mm(a0, x0);
    BGM_NO_HOIST();
    load(t + 3, a0, x0);
    BGM_NO_HOIST();
    mm(a1, x1);
    BGM_NO_HOIST();
    load(t + 4, a1, x1);
    BGM_NO_HOIST();
    mm(a2, x2);
  }
}

// KT0 = ceil(p / 16) when that shape is compiled, 0 = any p
template <int HT, int KT0, int T0 = 1>
__device__ __forceinline__ void ech_encoder(const float *theta, const EgmMlp &n, const float *vrow, f32x4 (&z)[T0], int j, int g) {
  const int p = n.dims[0], L = n.n_layers, q = n.dims[L];
  constexpr int H = 16 * HT;
  f32x4 h[HT];
  float wa[4 * HT][HT];                       // A fragments of the next hidden layer
  {
    const float *W = theta + n.woff[0], *bias = W + p * H;
    f32x4 bv[HT];
#pragma unroll
    for (int u = 0; u < HT; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) { bv[u][r] = bias[16 * u + 4 * g + r]; h[u][r] = 0.0f; }
    if constexpr (KT0 > 0) ech_encoder_l1_k<HT, KT0>(W, p, vrow, h, j, g);
    else ech_encoder_l1<HT>(W, p, vrow, h, j, g);
    if (L > 2) ech_load_a<HT>(theta + n.woff[1], H, wa, j, g);
    BGM_NO_HOIST();
#pragma unroll
    for (int u = 0; u < HT; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float t = h[u][r] + bv[u][r]; h[u][r] = fmaxf(t, EGM_LEAK * t); }
  }
  const float *Wl = theta + n.woff[L - 1];
  float wz[4 * HT];                            // A fragments of the output layer [16 HT x q]
  for (int l = 1; l < L - 1; ++l) {
    BGM_NO_HOIST();
    const float *W = theta + n.woff[l];
    f32x4 h2[HT], bv[HT];
#pragma unroll
    for (int u = 0; u < HT; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) { bv[u][r] = W[H * H + 16 * u + 4 * g + r]; h2[u][r] = 0.0f; }
    float wn[4 * HT][HT];
    ech_load_a<HT>(theta + n.woff[min(l + 1, L - 2)], H, wn, j, g);     // next hidden layer (the last one re-reads itself: no branch)
    BGM_NO_HOIST();
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < HT; ++u) h2[u] = BGM_MFMA(wa[4 * t + r][u], h[t][r], h2[u]);
#pragma unroll
    for (int u = 0; u < HT; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float t = h2[u][r] + bv[u][r]; h[u][r] = fmaxf(t, EGM_LEAK * t); }
#pragma unroll
    for (int k = 0; k < 4 * HT; ++k)
#pragma unroll
      for (int u = 0; u < HT; ++u) wa[k][u] = wn[k][u];
  }
#pragma unroll
  for (int tz = 0; tz < T0; ++tz) {            // output tiles (q <= 16 T0)
    const int o = 16 * tz + j;
#pragma unroll
    for (int k = 0; k < 4 * HT; ++k) wz[k] = Wl[(16 * (k >> 2) + 4 * g + (k & 3)) * q + min(o, q - 1)];
    f32x4 bz;
#pragma unroll
    for (int r = 0; r < 4; ++r) { bz[r] = ech_ld(Wl + H * q, 16 * tz + 4 * g + r, q); z[tz][r] = 0.0f; }
    const float jm = o < q ? 1.0f : 0.0f;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[tz] = BGM_MFMA(wz[4 * t + r] * jm, h[t][r], z[tz]);
    z[tz] += bz;
  }
}

// what the discriminator passes need from a step's arguments (CausalBGM and its Bayesian-network variant share them)
struct EchDiscIo {
  float *theta_d, *m_d, *v_d, *grad_d;
  EgmAdam adam;
  int apply, q;
  float *out;
  const float *z;      // [B x q] prior sample
  float eps;           // gradient-penalty interpolation coefficient
  float *gstash;       // T0 > 1 only: [B x SW] floats of global scratch for the fourth pass's stash (LDS holds three passes then)
};
// Adam on one parameter with its state already loaded
__device__ __forceinline__ void ech_adam(const EchDiscIo &a, int e, float gi, float th, float m0, float v0) {
  a.grad_d[e] = gi;
  if (a.apply) {
    const float mi = a.adam.b1 * m0 + (1.0f - a.adam.b1) * gi;
    const float vi = a.adam.b2 * v0 + (1.0f - a.adam.b2) * gi * gi;
    a.m_d[e] = mi; a.v_d[e] = vi;
    a.theta_d[e] = th - a.adam.lr_t * mi / (sqrtf(vi) + a.adam.eps);
  }
}

// LDS map behind the 64 reduction words: parameter block | per-wave partial sums | z_ tiles [B x 16] | stash of the four passes
template <int T1, int T2, int T3, int T0 = 1>
struct EchLds {
  float *par, *slots, *zt, *stash;
  __device__ __forceinline__ EchLds(float *lds, const EchP &P, int B) {
    using D = EchDims<T1, T2, T3, T0>;
    par = lds + 64;
    slots = par + P.total;
    zt = slots + ECH_ROLE_WAVES * D::SLOT;
    stash = zt + 16 * T0 * B;
  }
};

// The three discriminator passes, the gradient GEMMs and Adam.  On entry (after a workgroup barrier): `par` filled, z_ = e(v) of
// the minibatch in `zt` ([B x 16], zero beyond q).  Waves 0,1: D(z_); 2,3: D(z); 4,5: D(zhat) with the gradient penalty.
template <int T1, int T2, int T3, int NB, int T0 = 1>
__device__ __forceinline__ void ech_disc_tail(const EchDiscIo &a, const EgmDisc &dz, const EchP &P, const EchLds<T1, T2, T3, T0> &M, int tid) {
  using D = EchDims<T1, T2, T3, T0>;
  constexpr int B = 16 * NB;
  const int lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int q = a.q;
  float *par = M.par, *slots = M.slots, *zt = M.zt, *stash = M.stash;
  const int role = wave >> 1, tile = wave & 1;
  const bool active = tile < NB;
  const int row = 16 * tile + j;                      // row of the minibatch owned by this lane
  const float invB = 1.0f / (float)B;
  f32x4 nul1[T1], nul2[T2], nul3[T3];
  ech_zero<T1>(nul1); ech_zero<T2>(nul2); ech_zero<T3>(nul3);
  if (role == 0 && active) {
    EchAcc<T1, T2, T3, T0> acc;
    ech_zero_acc(acc);
    EchFwd<T1, T2, T3, T0> F;
#pragma unroll
    for (int t = 0; t < T0; ++t) F.a0[t] = *reinterpret_cast<const f32x4 *>(zt + row * (16 * T0) + 16 * t + 4 * g);
    ech_disc_fwd<T1, T2, T3, T0>(par, P, F, j, g);
    ech_disc_bwd<T1, T2, T3, false, T0>(par, P, F, invB, nul1, nul2, nul3, 1.0f, stash, B, row, acc, j, g);
    ech_write_slot<T1, T2, T3, T0>(slots + wave * D::SLOT, acc, F.out, 0.0f, j, g);
  } else if (role == 1 && active) {
    EchAcc<T1, T2, T3, T0> acc;
    ech_zero_acc(acc);
    EchFwd<T1, T2, T3, T0> F;
#pragma unroll
    for (int t = 0; t < T0; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) F.a0[t][r] = ech_ld(a.z + (long long)row * q, 16 * t + 4 * g + r, q);
    ech_disc_fwd<T1, T2, T3, T0>(par, P, F, j, g);
    ech_disc_bwd<T1, T2, T3, false, T0>(par, P, F, -invB, nul1, nul2, nul3, 1.0f, stash + 1 * B * D::SW, B, row, acc, j, g);
    ech_write_slot<T1, T2, T3, T0>(slots + wave * D::SLOT, acc, -F.out, 0.0f, j, g);
  } else if (role == 2 && active) {
    EchAcc<T1, T2, T3, T0> acc;
    ech_zero_acc(acc);
    EchFwd<T1, T2, T3, T0> F;
#pragma unroll
    for (int t = 0; t < T0; ++t) {
      const f32x4 ze = *reinterpret_cast<const f32x4 *>(zt + row * (16 * T0) + 16 * t + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) F.a0[t][r] = ech_ld(a.z + (long long)row * q, 16 * t + 4 * g + r, q) * a.eps + ze[r] * (1.0f - a.eps);
    }
    ech_disc_fwd<T1, T2, T3, T0>(par, P, F, j, g);
    const float part = ech_disc_gp<T1, T2, T3, T0>(par, P, F, 10.0f, stash + 2 * B * D::SW, T0 > 1 ? a.gstash : stash + 3 * B * D::SW, B, row, acc, j, g);
    ech_write_slot<T1, T2, T3, T0>(slots + wave * D::SLOT, acc, 0.0f, part, j, g);
  }
  if (T0 > 1) __threadfence();
  __syncthreads();
  // ---- parameter gradients: W_l += sum over the four passes of X^T D (rows = K), one 16x16 tile per wave and round
  const float c = ech_c();
  for (int tau = wave; tau < D::TILES; tau += ECH_WAVES) {
    int l, u, v, xw, dw, xo, dofs, n_in, n_out;
    if (tau < T0 * T1) { l = 0; u = tau / T1; v = tau - u * T1; xw = 16 * T0; dw = 16 * T1; xo = 0; dofs = 0; n_in = P.d0; n_out = P.d1; }
    else if (tau < T0 * T1 + T1 * T2) { const int k = tau - T0 * T1; l = 1; u = k / T2; v = k - u * T2; xw = 16 * T1; dw = 16 * T2; xo = 16 * T0; dofs = 16 * T1; n_in = P.d1; n_out = P.d2; }
    else { const int k = tau - T0 * T1 - T1 * T2; l = 2; u = k / T3; v = k - u * T3; xw = 16 * T2; dw = 16 * T3; xo = 16 * T0 + 16 * T1; dofs = 16 * (T1 + T2); n_in = P.d2; n_out = P.d3; }
    // Adam state of this lane's four elements: requested before the GEMM, consumed after it
    const int o = 16 * v + j;
    int e[4]; float th[4], m0[4], v0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * u + 4 * g + r;
      e[r] = dz.w[l] + min(f, n_in - 1) * n_out + min(o, n_out - 1);
      th[r] = a.theta_d[e[r]]; m0[r] = a.m_d[e[r]]; v0[r] = a.v_d[e[r]];
    }
    BGM_NO_HOIST();
    f32x4 w = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const float *pb = (T0 > 1 && pass == 3) ? a.gstash : stash + pass * B * D::SW;      // (unrolled: the address space is static)
      const float *xb = pb + B * xo + 16 * u + j;
      const float *db = pb + B * (D::XW + dofs) + 16 * v + j;
#pragma unroll
      for (int s4 = 0; s4 < NB * 4; ++s4) {
        const int rr = 4 * s4 + g;
        w = BGM_MFMA(xb[rr * xw], db[rr * dw], w);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * u + 4 * g + r;
      if (f < n_in && o < n_out) ech_adam(a, e[r], w[r], th[r], m0[r], v0[r]);
    }
  }
  // ---- vector parameters: fixed-order sums of the per-wave partials
  auto slot_sum = [&](int k) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < ECH_ROLE_WAVES; ++w)
      if ((w & 1) < NB) s += slots[w * D::SLOT + k];
    return s;
  };
  auto adam1 = [&](int e, float gi) { ech_adam(a, e, gi, a.theta_d[e], a.m_d[e], a.v_d[e]); };
  for (int k = tid; k < D::SL; k += ECH_THREADS) {
    int l, o;
    if (k < 16 * T1) { l = 0; o = k; } else if (k < 16 * (T1 + T2)) { l = 1; o = k - 16 * T1; } else { l = 2; o = k - 16 * (T1 + T2); }
    const int n_out = l == 0 ? P.d1 : (l == 1 ? P.d2 : P.d3);
    if (o < n_out) {
      const float gg = slot_sum(k), gb = slot_sum(D::SL + k);
      const float ga = par[(l == 0 ? P.ga0 : (l == 1 ? P.ga1 : P.ga2)) + o];
      adam1(dz.gamma[l] + o, gg);
      adam1(dz.beta[l] + o, gb);
      adam1(dz.b[l] + o, ga * c * gb);          // du = dy gamma c on every pass: the bias gradient is the beta gradient scaled
    }
  }
  // the output layer's vector and bias on another wave than the one that starts the loop above
  if (tid >= 256 && tid < 256 + P.d3) adam1(dz.w[3] + tid - 256, slot_sum(2 * D::SL + tid - 256));
  if (tid == 320) {
    adam1(dz.b[3], 0.0f);                        // sum_b dLoss/dout_b = B / B - B / B
    const float dz_loss = slot_sum(2 * D::SL + 16 * T3) * invB, gp = slot_sum(2 * D::SL + 16 * T3 + 1) * invB;
    if (a.out) { a.out[0] = dz_loss; a.out[1] = dz_loss + 10.0f * gp; }
  }
}

template <int HT, int KT0, int T1, int T2, int T3, int NB, int T0 = 1>
static __global__ __launch_bounds__(ECH_THREADS) void egm_disc_chain_kernel(EgmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ech_lds[];
  constexpr int B = 16 * NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const EchP P = ech_layout<T1, T2, T3, T0>(a.dz);
  const EchLds<T1, T2, T3, T0> M(ech_lds, P, B);
  ECH_STAMP(0);
  const int role = wave >> 1, tile = wave & 1;
  const int row = 16 * tile + j;
  if (role == 0) {
    if (tile < NB) {
      f32x4 zf[T0];
      ech_encoder<HT, KT0, T0>(a.theta_g, a.e, a.v + (long long)a.idx[row] * a.p, zf, j, g);
#pragma unroll
      for (int t = 0; t < T0; ++t) *reinterpret_cast<f32x4 *>(M.zt + row * (16 * T0) + 16 * t + 4 * g) = zf[t];
    }
  } else {
    // the other six waves: the discriminator's parameter block, then pull the encoder's weights into this XCD's L2 ahead of
    // the two waves that stream them
    const float *w = a.theta_g + a.e.off;
    const int n = a.e.woff[a.e.n_layers - 1] - a.e.off;
    float sink = 0.0f;
    for (int i = tid - 128; i < n; i += (ECH_THREADS - 128) * 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) sink += w[min(i + (ECH_THREADS - 128) * k, n - 1)];
    }
    asm volatile("" ::"v"(sink));
    ech_fill_params<T1, T2, T3, T0>(M.par, P, a.theta_d, a.dz, tid - 128, ECH_THREADS - 128);
  }
  ECH_STAMP(1);
  __syncthreads();
  ECH_STAMP(2);
  const EchDiscIo io{a.theta_d, a.m_d, a.v_d, a.grad_d, a.adam, a.apply, a.q, a.out, a.z, a.eps, a.ws};
  ech_disc_tail<T1, T2, T3, NB, T0>(io, a.dz, P, M, tid);
  ECH_STAMP(6);
}
