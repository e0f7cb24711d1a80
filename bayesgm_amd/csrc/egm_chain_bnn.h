// egm_chain_bnn.h -- Flipout networks (BayesianFullyConnectedNet, networks/bnn.py:4-38; input BatchNormalization in inference mode,
// the shipped default) on the register-chained row tiles of egm_chain.h / egm_chain_gen.h.
//
// A DenseFlipout layer is two products over one index space, y = h loc + ((h * s_in) dW) * s_out + b (bnn_kernels.h): on a row
// tile that is two sweeps of the pipelined sub-layer primitive (ecg_sub) over the same columns -- A fragments from `loc`, then from
// the call's perturbation dW = sigma * eps (materialised once per call by bnn_noise, as the phase-machine kernels do) -- and a sign
// flip per (row, feature) taken from the call's sign words.
#pragma once
#include "bnn_kernels.h"
#include "egm_chain_gen.h"

// x with the sign of every (row, feature) flipped where the bit string that starts at word w0 of the row's sign words has a 1
// (bnn_sign); feature 16 t + 4 g + r of the tile layout = bit (16 (t & 1) + 4 g + r) of word w0 + (t >> 1)
template <int NT>
__device__ __forceinline__ void ecb_flip(const uint32_t *roww, int w0, int g, const f32x4 (&x)[NT], f32x4 (&xs)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const uint32_t w = roww[w0 + (t >> 1)] >> (16 * (t & 1) + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) xs[t][r] = __uint_as_float(__float_as_uint(x[t][r]) ^ (((w >> r) & 1u) << 31));
  }
}

// One Flipout layer on a row tile whose first two K tiles of `loc` are in A; leaves the first two K tiles of `wnext` in An.
// v = h loc + bias + s_out * (hs dW)   (no activation here)
template <int KT, int NT, int NTN, bool CX, bool CXN>
__device__ __forceinline__ void ecb_layer(const float *loc, const float *dW, const float *bias, int ld, int n_in, int n_out, const uint32_t *roww,
                                          int sout_w, const f32x4 (&h)[KT], const f32x4 (&hs)[KT], f32x4 (&v)[NT], EcgA<NT> &A,
                                          const EcgW &wnext, EcgA<NTN> &An, int j, int g) {
  const EcgW wl{loc, ld, n_in, n_out, 0}, wd{dW, ld, n_in, n_out, 0};
  f32x4 c1[NT], c2[NT], c2s[NT];
  ech_zero<NT>(c1);
  ech_zero<NT>(c2);
  EcgA<NT> Ad;
  ecg_sub<KT, NT, NT, CX, CX>(wl, h, c1, A, wd, Ad, j, g);
  ecg_sub<KT, NT, NTN, CX, CXN>(wd, hs, c2, Ad, wnext, An, j, g);
  ecb_flip<NT>(roww, sout_w, g, c2, c2s);
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[u][r] = c1[u][r] + ech_ld(bias, 16 * u + 4 * g + r, n_out) + c2s[u][r];
}

// Noisy encoder call z_ = e(v) on one row tile (no gradient: train_disc_step keeps the encoder fixed).  dW: the call's
// perturbations (BnnCache::dW), roww: this row's sign words.  Hidden width 16 HT, NTL input tiles, q <= 16 outputs.
template <int HT, int NTL>
__device__ __forceinline__ void ecb_encoder(const float *theta, const BnnNet &n, const float *dW, const uint32_t *roww, const float *vrow,
                                            f32x4 (&z)[1], int j, int g) {
  constexpr int H = 16 * HT;
  const int L = n.n_layers, p = n.dims[0], q = n.dims[L];
  const float *gamma = theta + n.off, *beta = gamma + p;
  const float inv = 1.0f / sqrtf(1.0f + BNN_BN_EPS);
  f32x4 x0[NTL], xs0[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * t + 4 * g + r;
      const float xv = ech_ld(vrow, f, p);
      x0[t][r] = f < p ? fmaf(xv * inv, gamma[min(f, p - 1)], beta[min(f, p - 1)]) : 0.0f;
    }
  ecb_flip<NTL>(roww, n.sin_w[0], g, x0, xs0);
  f32x4 h[HT], hs[HT];
  EcgA<HT> A, An;
  {
    const float *loc = theta + n.woff[0];
    const EcgW w0{loc, H, p, H, 0};
    ecg_prime<HT, true>(w0, A, j, g);
    const EcgW wn{theta + n.woff[min(1, L - 1)], H, H, H, 0};
    if (L > 2) ecb_layer<NTL, HT, HT, true, true>(loc, dW + n.eoff[0], loc + 2 * p * H, H, p, H, roww, n.sout_w[0], x0, xs0, h, A, wn, An, j, g);
    else ecb_layer<NTL, HT, HT, true, true>(loc, dW + n.eoff[0], loc + 2 * p * H, H, p, H, roww, n.sout_w[0], x0, xs0, h, A, w0, An, j, g);
    ecg_lrelu<HT>(h);
    ecb_flip<HT>(roww, n.sin_w[1], g, h, hs);
    ecg_copy<HT>(A, An);
  }
  for (int l = 1; l < L - 1; ++l) {
    BGM_NO_HOIST();
    const float *loc = theta + n.woff[l];
    const EcgW wn{theta + n.woff[min(l + 1, L - 2)], H, H, H, 0};
    f32x4 v[HT];
    ecb_layer<HT, HT, HT, true, true>(loc, dW + n.eoff[l], loc + 2 * H * H, H, H, H, roww, n.sout_w[l], h, hs, v, A, wn, An, j, g);
    ecg_lrelu<HT>(v);
#pragma unroll
    for (int u = 0; u < HT; ++u) h[u] = v[u];
    ecb_flip<HT>(roww, n.sin_w[l + 1], g, h, hs);
    ecg_copy<HT>(A, An);
  }
  {
    const float *loc = theta + n.woff[L - 1];
    const EcgW wl{loc, q, H, q, 0};
    EcgA<1> A1, Ad;
    ecg_prime<1, false>(wl, A1, j, g);
    ecb_layer<HT, 1, 1, false, false>(loc, dW + n.eoff[L - 1], loc + 2 * H * q, q, H, q, roww, n.sout_w[L - 1], h, hs, z, A1, wl, Ad, j, g);
  }
}
