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
// perturbations (BnnCache::dW), roww: this row's sign words.  Hidden width 16 HT, NTL input tiles, q <= 16 T0 outputs.
template <int HT, int NTL, int T0 = 1>
__device__ __forceinline__ void ecb_encoder(const float *theta, const BnnNet &n, const float *dW, const uint32_t *roww, const float *vrow,
                                            f32x4 (&z)[T0], int j, int g) {
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
    EcgA<T0> A1, Ad;
    ecg_prime<T0, false>(wl, A1, j, g);
    ecb_layer<HT, T0, T0, false, false>(loc, dW + n.eoff[L - 1], loc + 2 * H * q, q, H, q, roww, n.sout_w[L - 1], h, hs, z, A1, wl, Ad, j, g);
  }
}

// =============================================================================================
// train_gen_step with Bayesian nets (nine Flipout calls: g x 3, e x 2, f x 2, h x 2; bnn_egm_gen_step_kernel) as row-tile chains
// =============================================================================================
#define ECB_CALLS 9
#define ECB_TILE_INTS 24
// call indices = noise streams of the step (oracle/bnn.py EGM_CALLS)
enum { ECB_G1 = 0, ECB_G1S = 1, ECB_E1 = 2, ECB_E2 = 3, ECB_G2 = 4, ECB_F = 5, ECB_FS = 6, ECB_H = 7, ECB_HS = 8 };

struct EcbCall {
  int net;                               // BNN_G / BNN_E / BNN_F / BNN_H
  int soff;                              // noise stream of the call = the step's stream id + soff
  int eps, dW, dWT;                      // workspace offset (floats) of the call's perturbations dW = sigma * eps (eps, dWT: unused, 0)
  int sg;                                // sign words [B x swords]
  int xh;                                // normalised input xhat [B x 16 KT0]
  int bnp;                               // per-row-tile sums for the input normalisation's gamma / beta: [NB][2][16 KT0]
  int x[BNN_MAX_LAYERS], xs[BNN_MAX_LAYERS], d[BNN_MAX_LAYERS], ds[BNN_MAX_LAYERS];    // layer inputs h, h * s_in; gradients cur, cur * s_out
};
struct EcbTab {
  EcbCall c[ECB_CALLS];
  int n_tiles;                           // weight-gradient tiles (tiles[] = [n_tiles][ECB_TILE_INTS])
  int kt0[4];                            // input tiles of the four nets
  int net_ncalls[4], net_calls[4][3];
  int n_warm;
  int klp;                               // theta step: per-workgroup KL partial sums of the noise launch, [n_calls][ECB_NOISE_PARTS]
};

// dW = sigma * eps and the sign words of one call (bnn_noise; eps itself is not stored: the gradient kernel recovers it as dW / sigma).
// `part` of `parts` workgroup-sized slices: the noise of the nine calls is 40 k Philox blocks -- 110 us on the chain kernel's one CU
// (integer multiplies at quarter rate, two waves per SIMD), a few microseconds as its own launch over 144 workgroups.
// RAW: the standard normals themselves (bgm_bnn_fit_epoch prepares a step's noise while the parameters it will be scaled with are still
// being updated: the gradient-tile kernel multiplies them by sigma of the NEW rho, ecb_theta_dw -- the same single product).
template <bool RAW = false>
__device__ __forceinline__ void ecb_noise(const float *theta, const BnnNet &n, const EcbCall &C, float *ws, int B, uint32_t k0, uint32_t k1,
                                          uint32_t stream, int tid, int part, int parts, uint32_t row0 = 0u) {
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.lin[l] * n.lout[l];
    const float *rho = theta + n.woff[l] + cnt;
    float *d = ws + C.dW + n.eoff[l];
    for (int i = part * BNN_THREADS + tid; i < (cnt + 3) >> 2; i += parts * BNN_THREADS) {
      const f32x4 z = box_muller4(philox4x32_10((uint32_t)i, (uint32_t)l | ((uint32_t)n.net_id << 16), stream, BNN_TAG_EPS, k0, k1));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = 4 * i + u;
        if (idx < cnt) d[idx] = RAW ? z[u] : (BNN_SCALE_EPS + softplus_f(rho[idx])) * z[u];      // transcendental-unit softplus: ~2 ulp on the scale of a random perturbation
      }
    }
  }
  const int calls = n.swords >> 2;
  uint32_t *sg = reinterpret_cast<uint32_t *>(ws + C.sg);
  for (int i = part * BNN_THREADS + tid; i < B * calls; i += parts * BNN_THREADS) {
    const int r = i / calls, cc = i - r * calls;
    const uint4 w = philox4x32_10(row0 + (uint32_t)r, (uint32_t)cc | ((uint32_t)n.net_id << 16), stream, BNN_TAG_SIGN, k0, k1);   // row0: first row of this rank's share (data-parallel warm start)
    uint32_t *dst = sg + (long long)r * n.swords + 4 * cc;
    dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w;
  }
}
#define ECB_NOISE_PARTS 16
// sum over this net's kernels (and biases) of KL(q || prior) (bnn_kl, value only), by `nthr` threads
__device__ __forceinline__ float ecb_kl_value(const float *theta, const BnnNet &n, int t, int nthr) {
  float acc = 0.0f;
  const float iv = n.prior_iv, ls = n.prior_logs;
  for (int l = 0; l < n.n_layers; ++l) {
    const int cnt = n.lin[l] * n.lout[l];
    const float *loc = theta + n.woff[l], *rho = loc + cnt;
    for (int i = t; i < cnt; i += nthr) {
      const float sg = BNN_SCALE_EPS + softplus_acc(rho[i]), mu = loc[i];
      acc += -logf(sg) + 0.5f * (sg * sg + mu * mu) * iv - 0.5f + ls;
    }
    if (n.bias_prior) {
      const float *b = rho + cnt;
      for (int i = t; i < n.lout[l]; i += nthr) acc += 0.5f * b[i] * b[i] * iv + ls + 0.9189385332046727f;
    }
  }
  return acc;
}

// KL(q || prior) of the slice of the call's net this workgroup visits (bnn_kl, value only) -> ws[klp + blockIdx.x]; fixed order.
__device__ __forceinline__ void ecb_kl_partial(const float *theta, const BnnNet &n, float *dst, int part, int parts, float *red) {
  const int tid = threadIdx.x;
  float acc = ecb_kl_value(theta, n, part * BNN_THREADS + tid, parts * BNN_THREADS);
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) { float t = 0.0f; for (int w = 0; w < BNN_THREADS / 64; ++w) t += red[w]; *dst = t; }
}
template <class Args>
__device__ __forceinline__ void ecb_gen_noise(const Args &a, const EcbTab &tab, float *ws_, uint32_t row0 = 0u) {      // grid: n_calls * ECB_NOISE_PARTS workgroups
  const int c = blockIdx.x / ECB_NOISE_PARTS, part = blockIdx.x % ECB_NOISE_PARTS;
  ecb_noise(a.theta, a.net[tab.c[c].net], tab.c[c], ws_, a.B, a.k0, a.k1, a.stream + (uint32_t)tab.c[c].soff, threadIdx.x, part, ECB_NOISE_PARTS, row0);
}

// bgm_bnn_fit_epoch: the noise of a step off the critical path.  A step's perturbations are sigma(rho) * eps with eps independent of the
// parameters: RIDER workgroups of the theta-chain launch of minibatch k (the chains occupy six workgroups, the chip is idle beside them)
// write eps and the sign words of (i) the theta phase of minibatch k + 1 and (ii) the latent phase of minibatch k into the workspaces
// those phases will read, and the KL partial sums of THIS step (the chain workgroups wait for them on a counter before their last
// addition); the gradient-tile kernel that follows, in the thread that has just written a weight's new rho, scales the weight's eps in
// both workspaces.  The noise launches (15-19 us each, in front of the chains of both streams) disappear; values are the same bits.
struct EcbRider {
  float *ws_next; uint32_t stream_next;                       // theta phase of the next minibatch (NULL: not prepared)
  const EcbTab *tab_z; float *ws_z; uint32_t stream_z;        // latent phase of this minibatch (NULL: not prepared)
  unsigned *kl_cnt;                                           // rider workgroups that have written their KL partial sum
};
struct EcbAhead {
  float *ws_next; const EcbTab *tab_z; float *ws_z;           // workspaces holding raw normals to be scaled (NULL: none)
  int dw_t[4], dw_z[4][2];                                    // per net: the perturbation offsets (EcbCall::dW) of its theta call / its two latent calls
  float *zero_t, *zero_z;                                     // loss words the next theta / this latent phase accumulates into
  unsigned *kl_cnt;                                           // reset for the workspace's next use
};
// the latent chain kernel applies the Adam step of its own 16 batch rows in its epilogue (the thread that has assembled an element of
// dz owns that element of the latent table): no row-update launch behind it.  zm == NULL: dz only.
struct EcbZRows { float *data_z, *zm, *zv; float lr_t, b1, b2, eps; int *t_last; int t_now; };
#define ECB_RIDERS (9 * ECB_NOISE_PARTS)
template <class Args>
__device__ __forceinline__ void ecb_rider(const Args &a, const EcbTab &tab, float *ws, const EcbRider &rd, int r, float *red) {
  const int c = r / ECB_NOISE_PARTS, part = r % ECB_NOISE_PARTS;
  if (r < 3 * ECB_NOISE_PARTS) {
    ecb_kl_partial(a.theta, a.net[tab.c[c].net], ws + tab.klp + r, part, ECB_NOISE_PARTS, red);
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(rd.kl_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (rd.ws_next)
      ecb_noise<true>(a.theta, a.net[tab.c[c].net], tab.c[c], rd.ws_next, a.B, a.k0, a.k1, rd.stream_next + (uint32_t)tab.c[c].soff, threadIdx.x, part,
                      ECB_NOISE_PARTS);
  } else if (r < ECB_RIDERS && rd.ws_z) {
    const EcbCall &C = rd.tab_z->c[c - 3];
    ecb_noise<true>(a.theta, a.net[C.net], C, rd.ws_z, a.B, a.k0, a.k1, rd.stream_z + (uint32_t)C.soff, threadIdx.x, part, ECB_NOISE_PARTS);
  }
}

// v = a W1 + flip(as W2): the Flipout product pair in either direction.  KC: W1, W2 are given K-contiguously (ecg_load_tile): the
// backward products read the canonical `loc` and the canonical perturbation that way, so neither needs a transposed copy.
template <int KT, int NT, bool KC, bool PAD>
__device__ __forceinline__ void ecb_pair_wide(const float *W1, const float *W2, int ld, int n_in, int n_out, const uint32_t *roww, int flipw,
                                              const f32x4 (&a)[KT], const f32x4 (&as)[KT], f32x4 (&v)[NT], int j, int g) {
  f32x4 c2[NT], c2s[NT];
  ecg_wide<KT, NT, KC, PAD>(W1, ld, n_in, n_out, a, v, j, g);
  ecg_wide<KT, NT, KC, PAD>(W2, ld, n_in, n_out, as, c2, j, g);
  ecb_flip<NT>(roww, flipw, g, c2, c2s);
#pragma unroll
  for (int u = 0; u < NT; ++u) v[u] += c2s[u];
}
template <int KT, int NT, bool CX, bool KC = false, bool PAD = false>
__device__ __forceinline__ void ecb_pair(const float *W1, const float *W2, int ld, int n_in, int n_out, const uint32_t *roww, int flipw,
                                         const f32x4 (&a)[KT], const f32x4 (&as)[KT], f32x4 (&v)[NT], int j, int g) {
  if constexpr (NT > 4) ecb_pair_wide<KT, NT, KC, PAD>(W1, W2, ld, n_in, n_out, roww, flipw, a, as, v, j, g);
  else {
    const EcgW w1{W1, ld, n_in, n_out, 0}, w2{W2, ld, n_in, n_out, 0};
    EcgA<NT> A, Ad, Ad2;
    ecg_prime<NT, CX, KC>(w1, A, j, g);
    f32x4 c2[NT], c2s[NT];
    ech_zero<NT>(v);
    ech_zero<NT>(c2);
    ecg_sub<KT, NT, NT, CX, CX, KC, KC>(w1, a, v, A, w2, Ad, j, g);
    ecg_sub<KT, NT, NT, CX, CX, KC, KC>(w2, as, c2, Ad, w2, Ad2, j, g);
    ecb_flip<NT>(roww, flipw, g, c2, c2s);
#pragma unroll
    for (int u = 0; u < NT; ++u) v[u] += c2s[u];
  }
}
template <int NT>
__device__ __forceinline__ void ecb_add_bias(const float *bias, int n, int g, f32x4 (&v)[NT]) { ecg_bias<NT>(bias, n, 0, g, v); }

// Flipout MLP with hidden width 16 HT (g: KT0 = 1, NTO = output tiles; e: KT0 = input tiles, NTO = 1): forward with stash.
// xraw: the raw input tiles (zero beyond dims[0]).
// ecb_mlp_fwd_hidden: everything up to the input (h, flipped copy hs) of the last layer, stashed; ecb_mlp_fwd adds the last layer.
template <int KT0, int HT, bool PAD = false>
__device__ __forceinline__ void ecb_mlp_fwd_hidden(const float *theta, const BnnNet &n, const EcbCall &C, float *ws, int row, const f32x4 (&xraw)[KT0],
                                                   f32x4 (&h)[HT], f32x4 (&hs)[HT], int j, int g) {
  constexpr int H = 16 * HT;
  const int L = n.n_layers, d0 = n.dims[0];
  const uint32_t *roww = reinterpret_cast<const uint32_t *>(ws + C.sg) + (long long)row * n.swords;
  const float *dW = ws + C.dW;
  const float *gamma = theta + n.off, *beta = gamma + d0;
  const float inv = 1.0f / sqrtf(1.0f + BNN_BN_EPS);
  f32x4 xh[KT0], h0[KT0], hs0[KT0];
#pragma unroll
  for (int t = 0; t < KT0; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * t + 4 * g + r, fc = min(f, d0 - 1);
      xh[t][r] = xraw[t][r] * inv;
      h0[t][r] = f < d0 ? fmaf(xh[t][r], gamma[fc], beta[fc]) : 0.0f;
    }
  ecb_flip<KT0>(roww, n.sin_w[0], g, h0, hs0);
  ecg_put<KT0>(ws + C.xh, row, g, xh);
  ecg_put<KT0>(ws + C.x[0], row, g, h0);
  ecg_put<KT0>(ws + C.xs[0], row, g, hs0);
  {
    const float *loc = theta + n.woff[0];
    ecb_pair<KT0, HT, true>(loc, dW + n.eoff[0], H, d0, H, roww, n.sout_w[0], h0, hs0, h, j, g);
    ecb_add_bias<HT>(loc + 2 * d0 * H, H, g, h);
    ecg_lrelu<HT>(h);
    ecb_flip<HT>(roww, n.sin_w[1], g, h, hs);
  }
  if (L > 2) {
    EcgA<HT> A, An;
    ecg_prime<HT, true>(EcgW{theta + n.woff[1], H, H, H, 0}, A, j, g);
    for (int l = 1; l < L - 1; ++l) {
      BGM_NO_HOIST();
      ecg_put<HT>(ws + C.x[l], row, g, h);
      ecg_put<HT>(ws + C.xs[l], row, g, hs);
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
  }
  ecg_put<HT>(ws + C.x[L - 1], row, g, h);
  ecg_put<HT>(ws + C.xs[L - 1], row, g, hs);
}
template <int KT0, int HT, int NTO, bool PAD = false>
__device__ __forceinline__ void ecb_mlp_fwd(const float *theta, const BnnNet &n, const EcbCall &C, float *ws, int row, const f32x4 (&xraw)[KT0],
                                            f32x4 (&out)[NTO], int j, int g) {
  constexpr int H = 16 * HT;
  const int L = n.n_layers, no = n.dims[L];
  const uint32_t *roww = reinterpret_cast<const uint32_t *>(ws + C.sg) + (long long)row * n.swords;
  const float *dW = ws + C.dW;
  f32x4 h[HT], hs[HT];
  ecb_mlp_fwd_hidden<KT0, HT, PAD>(theta, n, C, ws, row, xraw, h, hs, j, g);
  const float *loc = theta + n.woff[L - 1];
  ecb_pair<HT, NTO, false, false, PAD>(loc, dW + n.eoff[L - 1], no, H, no, roww, n.sout_w[L - 1], h, hs, out, j, g);
  ecb_add_bias<NTO>(loc + 2 * H * no, no, g, out);
}

// backward with stash.  dout: dLoss/d output (zero beyond n_out).  dxraw (WANT_DX): dLoss/d raw input.  The per-row products for the
// input normalisation's gamma / beta are summed over the tile's rows into bnp [2][16 KT0] (this tile's slot).
// ecb_mlp_bwd_hidden: from dh = dLoss / d(pre-activation of layer L - 2) (already masked) down to the input; ecb_mlp_bwd runs the last
// layer's backward product first.
template <int KT0, int HT, bool WANT_DX, bool PAD = false>
__device__ __forceinline__ void ecb_mlp_bwd_hidden(const float *theta, const BnnNet &n, const EcbCall &C, float *ws, int row, int tile, f32x4 (&dh)[HT],
                                                   f32x4 (&dxraw)[KT0], int j, int g);
template <int KT0, int HT, int NTO, bool WANT_DX, bool PAD = false>
__device__ __forceinline__ void ecb_mlp_bwd(const float *theta, const float *thetaT, const BnnNet &n, const EcbCall &C, float *ws, int row, int tile,
                                            const f32x4 (&dout)[NTO], f32x4 (&dxraw)[KT0], int j, int g) {
  constexpr int H = 16 * HT;
  const int L = n.n_layers, no = n.dims[L];
  const uint32_t *roww = reinterpret_cast<const uint32_t *>(ws + C.sg) + (long long)row * n.swords;
  const float *dW = ws + C.dW;
  f32x4 dh[HT], xl[HT];
  {
    f32x4 douts[NTO];
    ecb_flip<NTO>(roww, n.sout_w[L - 1], g, dout, douts);
    ecg_put<NTO>(ws + C.d[L - 1], row, g, dout);
    ecg_put<NTO>(ws + C.ds[L - 1], row, g, douts);
    ecb_pair<NTO, HT, true, true>(theta + n.woff[L - 1], dW + n.eoff[L - 1], no, no, H, roww, n.sin_w[L - 1], dout, douts, dh, j, g);
    ecg_get<HT>(ws + C.x[L - 1], row, g, xl);
    ecg_mask<HT>(dh, xl);
  }
  ecb_mlp_bwd_hidden<KT0, HT, WANT_DX, PAD>(theta, n, C, ws, row, tile, dh, dxraw, j, g);
  (void)thetaT;
}
template <int KT0, int HT, bool WANT_DX, bool PAD>
__device__ __forceinline__ void ecb_mlp_bwd_hidden(const float *theta, const BnnNet &n, const EcbCall &C, float *ws, int row, int tile, f32x4 (&dh)[HT],
                                                   f32x4 (&dxraw)[KT0], int j, int g) {
  constexpr int H = 16 * HT;
  const int L = n.n_layers, d0 = n.dims[0];
  const uint32_t *roww = reinterpret_cast<const uint32_t *>(ws + C.sg) + (long long)row * n.swords;
  const float *dW = ws + C.dW;
  f32x4 dhs[HT], xl[HT];
  for (int l = L - 2; l >= 1; --l) {
    BGM_NO_HOIST();
    ecb_flip<HT>(roww, n.sout_w[l], g, dh, dhs);
    ecg_put<HT>(ws + C.d[l], row, g, dh);
    ecg_put<HT>(ws + C.ds[l], row, g, dhs);
    f32x4 d2[HT];
    ecb_pair<HT, HT, true, true>(theta + n.woff[l], dW + n.eoff[l], H, H, H, roww, n.sin_w[l], dh, dhs, d2, j, g);
    ecg_get<HT>(ws + C.x[l], row, g, xl);
    ecg_mask<HT>(d2, xl);
#pragma unroll
    for (int u = 0; u < HT; ++u) dh[u] = d2[u];
  }
  ecb_flip<HT>(roww, n.sout_w[0], g, dh, dhs);
  ecg_put<HT>(ws + C.d[0], row, g, dh);
  ecg_put<HT>(ws + C.ds[0], row, g, dhs);
  f32x4 dh0[KT0], xh[KT0];
  ecb_pair<HT, KT0, false, true, PAD>(theta + n.woff[0], dW + n.eoff[0], H, H, d0, roww, n.sin_w[0], dh, dhs, dh0, j, g);
  ecg_get<KT0>(ws + C.xh, row, g, xh);
  float *bnp = ws + C.bnp + tile * (2 * 16 * KT0);
  const float *gamma = theta + n.off;
  const float inv = 1.0f / sqrtf(1.0f + BNN_BN_EPS);
#pragma unroll
  for (int t = 0; t < KT0; ++t) {
    f32x4 sg_, sb;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sg_[r] = sum_over_j_to_lane15(dh0[t][r] * xh[t][r]);
      sb[r] = sum_over_j_to_lane15(dh0[t][r]);
      if (WANT_DX) dxraw[t][r] = inv * dh0[t][r] * ech_ld(gamma, 16 * t + 4 * g + r, d0);
    }
    if (j == 15) {
      *reinterpret_cast<f32x4 *>(bnp + 16 * t + 4 * g) = sg_;
      *reinterpret_cast<f32x4 *>(bnp + 16 * KT0 + 16 * t + 4 * g) = sb;
    }
  }
}

// head networks f, h: [in <= 16] -> 16 T1 -> 16 T2 -> 16 T3 -> [out <= 16]
template <int T1, int T2, int T3>
__device__ __forceinline__ void ecb_head_fwd(const float *theta, const BnnNet &n, const EcbCall &C, float *ws, int row, const f32x4 (&xraw)[1],
                                             f32x4 (&out)[1], int j, int g) {
  const int d0 = n.dims[0], d1 = n.dims[1], d2 = n.dims[2], d3 = n.dims[3], d4 = n.dims[4];
  const uint32_t *roww = reinterpret_cast<const uint32_t *>(ws + C.sg) + (long long)row * n.swords;
  const float *dW = ws + C.dW;
  const float *gamma = theta + n.off, *beta = gamma + d0;
  const float inv = 1.0f / sqrtf(1.0f + BNN_BN_EPS);
  f32x4 xh[1], h0[1], hs0[1];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 4 * g + r, fc = min(f, d0 - 1);
    xh[0][r] = xraw[0][r] * inv;
    h0[0][r] = f < d0 ? fmaf(xh[0][r], gamma[fc], beta[fc]) : 0.0f;
  }
  ecb_flip<1>(roww, n.sin_w[0], g, h0, hs0);
  ecg_put<1>(ws + C.xh, row, g, xh);
  ecg_put<1>(ws + C.x[0], row, g, h0);
  ecg_put<1>(ws + C.xs[0], row, g, hs0);
  f32x4 a1[T1], a1s[T1], a2[T2], a2s[T2], a3[T3], a3s[T3];
  const float *l0 = theta + n.woff[0], *l1 = theta + n.woff[1], *l2 = theta + n.woff[2], *l3 = theta + n.woff[3];
  ecb_pair<1, T1, true>(l0, dW + n.eoff[0], d1, d0, d1, roww, n.sout_w[0], h0, hs0, a1, j, g);
  ecb_add_bias<T1>(l0 + 2 * d0 * d1, d1, g, a1);
  ecg_lrelu<T1>(a1);
  ecb_flip<T1>(roww, n.sin_w[1], g, a1, a1s);
  ecg_put<T1>(ws + C.x[1], row, g, a1);
  ecg_put<T1>(ws + C.xs[1], row, g, a1s);
  ecb_pair<T1, T2, true>(l1, dW + n.eoff[1], d2, d1, d2, roww, n.sout_w[1], a1, a1s, a2, j, g);
  ecb_add_bias<T2>(l1 + 2 * d1 * d2, d2, g, a2);
  ecg_lrelu<T2>(a2);
  ecb_flip<T2>(roww, n.sin_w[2], g, a2, a2s);
  ecg_put<T2>(ws + C.x[2], row, g, a2);
  ecg_put<T2>(ws + C.xs[2], row, g, a2s);
  ecb_pair<T2, T3, false>(l2, dW + n.eoff[2], d3, d2, d3, roww, n.sout_w[2], a2, a2s, a3, j, g);
  ecb_add_bias<T3>(l2 + 2 * d2 * d3, d3, g, a3);
  ecg_lrelu<T3>(a3);
  ecb_flip<T3>(roww, n.sin_w[3], g, a3, a3s);
  ecg_put<T3>(ws + C.x[3], row, g, a3);
  ecg_put<T3>(ws + C.xs[3], row, g, a3s);
  ecb_pair<T3, 1, false>(l3, dW + n.eoff[3], d4, d3, d4, roww, n.sout_w[3], a3, a3s, out, j, g);
  ecb_add_bias<1>(l3 + 2 * d3 * d4, d4, g, out);
}
template <int T1, int T2, int T3>
__device__ __forceinline__ void ecb_head_bwd(const float *theta, const float *thetaT, const BnnNet &n, const EcbCall &C, float *ws, int row, int tile,
                                             const f32x4 (&dout)[1], f32x4 (&dxraw)[1], int j, int g) {
  const int d0 = n.dims[0], d1 = n.dims[1], d2 = n.dims[2], d3 = n.dims[3], d4 = n.dims[4];
  const uint32_t *roww = reinterpret_cast<const uint32_t *>(ws + C.sg) + (long long)row * n.swords;
  const float *dW = ws + C.dW;
  f32x4 douts[1], e3[T3], e3s[T3], e2[T2], e2s[T2], e1[T1], e1s[T1], x3[T3], x2[T2], x1[T1], dh0[1], xh[1];
  ecb_flip<1>(roww, n.sout_w[3], g, dout, douts);
  ecg_put<1>(ws + C.d[3], row, g, dout);
  ecg_put<1>(ws + C.ds[3], row, g, douts);
  ecb_pair<1, T3, false, true>(theta + n.woff[3], dW + n.eoff[3], d4, d4, d3, roww, n.sin_w[3], dout, douts, e3, j, g);
  ecg_get<T3>(ws + C.x[3], row, g, x3);
  ecg_mask<T3>(e3, x3);
  ecb_flip<T3>(roww, n.sout_w[2], g, e3, e3s);
  ecg_put<T3>(ws + C.d[2], row, g, e3);
  ecg_put<T3>(ws + C.ds[2], row, g, e3s);
  ecb_pair<T3, T2, true, true>(theta + n.woff[2], dW + n.eoff[2], d3, d3, d2, roww, n.sin_w[2], e3, e3s, e2, j, g);
  ecg_get<T2>(ws + C.x[2], row, g, x2);
  ecg_mask<T2>(e2, x2);
  ecb_flip<T2>(roww, n.sout_w[1], g, e2, e2s);
  ecg_put<T2>(ws + C.d[1], row, g, e2);
  ecg_put<T2>(ws + C.ds[1], row, g, e2s);
  ecb_pair<T2, T1, true, true>(theta + n.woff[1], dW + n.eoff[1], d2, d2, d1, roww, n.sin_w[1], e2, e2s, e1, j, g);
  ecg_get<T1>(ws + C.x[1], row, g, x1);
  ecg_mask<T1>(e1, x1);
  ecb_flip<T1>(roww, n.sout_w[0], g, e1, e1s);
  ecg_put<T1>(ws + C.d[0], row, g, e1);
  ecg_put<T1>(ws + C.ds[0], row, g, e1s);
  ecb_pair<T1, 1, false, true>(theta + n.woff[0], dW + n.eoff[0], d1, d1, d0, roww, n.sin_w[0], e1, e1s, dh0, j, g);
  ecg_get<1>(ws + C.xh, row, g, xh);
  float *bnp = ws + C.bnp + tile * (2 * 16);
  const float *gamma = theta + n.off;
  const float inv = 1.0f / sqrtf(1.0f + BNN_BN_EPS);
  f32x4 sg_, sb;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sg_[r] = sum_over_j_to_lane15(dh0[0][r] * xh[0][r]);
    sb[r] = sum_over_j_to_lane15(dh0[0][r]);
    dxraw[0][r] = inv * dh0[0][r] * ech_ld(gamma, 4 * g + r, d0);
  }
  if (j == 15) {
    *reinterpret_cast<f32x4 *>(bnp + 4 * g) = sg_;
    *reinterpret_cast<f32x4 *>(bnp + 16 + 4 * g) = sb;
  }
}

// LDS behind the discriminator's parameter block: loss partials [8 x 16] | flags [16] | z_ | head-input gradient scratch | head
// contribution to dLoss/dz_  ([B x 16 T0], [B x 16], [B x 16 T0])
template <int T1, int T2, int T3, int T0 = 1>
__host__ __device__ inline int ecb_lds_floats(const EgmDisc &d, int B) {
  return 64 + ech_layout<T1, T2, T3, T0>(d).total + 8 * 16 + 16 + (2 * 16 * T0 + 16) * B;
}

struct BnnEgmArgs;   // bnn_egm_kernels.h

// Chains (waves):  0,1  g(z) [G1] -> v_ -> e(v_) [E2] -> backward of both          (latent reconstruction)
//                  2,3  e(v) [E1] -> z_ -> g(z_) [G2] -> backward, then e backward with the summed dLoss/dz_
//                  4,5  g(z) [G1S] (variance-head penalty) forward + backward; then, from z_: D, f [F, FS], h [H, HS]
//                  6,7  discriminator parameter block, L2 warm-up
// The noise of the nine calls has been drawn by the launch before (ecb_gen_noise).
template <class Args, int HT, int NTL, int T1, int T2, int T3, int NB, bool PAD = false, int T0 = 1>
__device__ __forceinline__ void ecb_gen_chain(const Args &a, const EcbTab &tab, float *thetaT_, float *ech_lds) {
  constexpr int B = 16 * NB, ZW = 16 * T0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int q = a.q, p = a.p, z0 = a.z0, z1 = a.z1, z2 = a.z2;
  const EchP P = ech_layout<T1, T2, T3, T0>(a.dz);
  float *par = ech_lds + 64;
  float *part = par + P.total;
  volatile int *flag = reinterpret_cast<volatile int *>(part + 8 * 16);
  float *zt = part + 8 * 16 + 16, *dzt = zt + ZW * B, *dzh = dzt + 16 * B;
  float *ws = a.ws;
  const float *th = a.theta, *tT = thetaT_;
  if (tid < 16) flag[tid] = 0;
  __syncthreads();
  const int role = wave >> 1, tile = wave & 1;
  const bool active = tile < NB;
  const int row = 16 * tile + j;
  const float invB = 1.0f / (float)B;
  const float zrec = a.use_z_rec ? 1.0f : 0.0f;
  const BnnNet &G = a.net[BNN_G], &E = a.net[BNN_E], &F = a.net[BNN_F], &Hn = a.net[BNN_H];
  float ls[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};     // l_v, l_z, l_x, l_y, s_g, s_f, s_h, adv
  if (role == 0 && active) {
    f32x4 zin[T0];
#pragma unroll
    for (int t = 0; t < T0; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) zin[t][r] = ech_ld(a.z + (long long)row * q, 16 * t + 4 * g + r, q);
    f32x4 gz[NTL];
    ecb_mlp_fwd<T0, HT, NTL, PAD>(th, G, tab.c[ECB_G1], ws, row, zin, gz, j, g);
    f32x4 vin[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) vin[t][r] = (16 * t + 4 * g + r < p) ? gz[t][r] : 0.0f;
    f32x4 zz[T0];
    ecb_mlp_fwd<NTL, HT, T0, PAD>(th, E, tab.c[ECB_E2], ws, row, vin, zz, j, g);
    f32x4 dzz[T0];
    float lz = 0.0f;
#pragma unroll
    for (int tz = 0; tz < T0; ++tz)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = (16 * tz + 4 * g + r < q) ? zin[tz][r] - zz[tz][r] : 0.0f;
        lz = fmaf(t, t, lz);
        dzz[tz][r] = zrec * (-2.0f / (float)(B * q)) * t;
      }
    ls[1] = sum_over_g(lz);
    f32x4 dv[NTL];
    ecb_mlp_bwd<NTL, HT, T0, true, PAD>(th, tT, E, tab.c[ECB_E2], ws, row, tile, dzz, dv, j, g);
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dv[t][r] = (16 * t + 4 * g + r < p) ? dv[t][r] : 0.0f;
    f32x4 dnone[T0];
    ecb_mlp_bwd<T0, HT, NTL, false, PAD>(th, tT, G, tab.c[ECB_G1], ws, row, tile, dv, dnone, j, g);
  } else if (role == 1 && active) {
    const float *vrow = a.v_ + (long long)a.idx[row] * p;
    f32x4 vin[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) vin[t][r] = ech_ld(vrow, 16 * t + 4 * g + r, p);
    f32x4 ze[T0];
    ecb_mlp_fwd<NTL, HT, T0, PAD>(th, E, tab.c[ECB_E1], ws, row, vin, ze, j, g);
#pragma unroll
    for (int t = 0; t < T0; ++t) *reinterpret_cast<f32x4 *>(zt + row * ZW + 16 * t + 4 * g) = ze[t];
    __threadfence_block();
    if (lane == 0) flag[2 + tile] = 1;
    f32x4 dz[T0];
    {
      f32x4 gv[NTL];
      ecb_mlp_fwd<T0, HT, NTL, PAD>(th, G, tab.c[ECB_G2], ws, row, ze, gv, j, g);
      float lv = 0.0f;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * g + r;
          const float d = f < p ? ech_ld(vrow, f, p) - gv[t][r] : 0.0f;
          lv = fmaf(d, d, lv);
          gv[t][r] = (-2.0f / (float)(B * p)) * d;
        }
      ls[0] = sum_over_g(lv);
      ecb_mlp_bwd<T0, HT, NTL, true, PAD>(th, tT, G, tab.c[ECB_G2], ws, row, tile, gv, dz, j, g);
    }
    while (flag[4 + tile] == 0) __builtin_amdgcn_s_sleep(2);
    __threadfence_block();
#pragma unroll
    for (int t = 0; t < T0; ++t) dz[t] += *reinterpret_cast<const f32x4 *>(dzh + row * ZW + 16 * t + 4 * g);
    f32x4 dnone[NTL];
    ecb_mlp_bwd<NTL, HT, T0, false, PAD>(th, tT, E, tab.c[ECB_E1], ws, row, tile, dz, dnone, j, g);
  } else if (role == 2 && active) {
    {   // second g(z) call: the variance-head penalty
      f32x4 zin[T0];
#pragma unroll
      for (int t = 0; t < T0; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) zin[t][r] = ech_ld(a.z + (long long)row * q, 16 * t + 4 * g + r, q);
      f32x4 gzs[NTL];
      ecb_mlp_fwd<T0, HT, NTL, PAD>(th, G, tab.c[ECB_G1S], ws, row, zin, gzs, j, g);
      float sgv = 0.0f;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) sgv += (16 * t + 4 * g + r == p) ? gzs[t][r] : 0.0f;
      sgv = sum_over_g(sgv);
      ls[4] = sgv * sgv;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) gzs[t][r] = (16 * t + 4 * g + r == p) ? 0.001f * 2.0f * sgv * invB : 0.0f;
      f32x4 dnone[T0];
      ecb_mlp_bwd<T0, HT, NTL, false, PAD>(th, tT, G, tab.c[ECB_G1S], ws, row, tile, gzs, dnone, j, g);
    }
    const long long prow = a.idx[row];
    const float xv = a.x_[prow], yv = a.y_[prow];
    while (flag[0] < 2) __builtin_amdgcn_s_sleep(2);
    while (flag[2 + tile] == 0) __builtin_amdgcn_s_sleep(2);
    __threadfence_block();
    f32x4 ze[T0], dz[T0];
#pragma unroll
    for (int t = 0; t < T0; ++t) ze[t] = *reinterpret_cast<const f32x4 *>(zt + row * ZW + 16 * t + 4 * g);
    {
      EchFwd<T1, T2, T3, T0> Fw;
#pragma unroll
      for (int t = 0; t < T0; ++t) Fw.a0[t] = ze[t];
      ech_disc_fwd<T1, T2, T3, T0>(par, P, Fw, j, g);
      ls[7] = -Fw.out;
      ecg_disc_dx<T1, T2, T3, T0>(par, P, Fw, -invB, dz, j, g);
    }
    {   // f: mean call, variance-penalty call
      f32x4 fin[1], fo[1], dfo[1], dfin[1];
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int f = 4 * g + r; fin[0][r] = f < z0 + z1 ? ze[0][r] : (f == z0 + z1 ? xv : 0.0f); }
      const int of = F.dims[F.n_layers];
      ecb_head_fwd<T1, T2, T3>(th, F, tab.c[ECB_F], ws, row, fin, fo, j, g);
      const float mu = __shfl(fo[0][0], j);
      ls[3] = (mu - yv) * (mu - yv);
#pragma unroll
      for (int r = 0; r < 4; ++r) dfo[0][r] = (4 * g + r == 0) ? 2.0f * (mu - yv) * invB : 0.0f;
      ecb_head_bwd<T1, T2, T3>(th, tT, F, tab.c[ECB_F], ws, row, tile, dfo, dfin, j, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) dz[0][r] += (4 * g + r < z0 + z1) ? dfin[0][r] : 0.0f;
      ecb_head_fwd<T1, T2, T3>(th, F, tab.c[ECB_FS], ws, row, fin, fo, j, g);
      float sg = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) sg += (4 * g + r == of - 1) ? fo[0][r] : 0.0f;
      sg = sum_over_g(sg);
      ls[5] = sg * sg;
#pragma unroll
      for (int r = 0; r < 4; ++r) dfo[0][r] = (4 * g + r == of - 1) ? 0.001f * 2.0f * sg * invB : 0.0f;
      ecb_head_bwd<T1, T2, T3>(th, tT, F, tab.c[ECB_FS], ws, row, tile, dfo, dfin, j, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) dz[0][r] += (4 * g + r < z0 + z1) ? dfin[0][r] : 0.0f;
    }
    {   // h: mean call, variance-penalty call (input: z_[0:z0] and z_[z0+z1 : z0+z1+z2], a lane permutation through LDS)
      f32x4 hin[1], ho[1], dho[1], dhin[1], dhin2[1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 4 * g + r;
        const float t = zt[row * ZW + min(f < z0 ? f : f + z1, ZW - 1)];
        hin[0][r] = f < z0 + z2 ? t : 0.0f;
      }
      const int oh = Hn.dims[Hn.n_layers];
      ecb_head_fwd<T1, T2, T3>(th, Hn, tab.c[ECB_H], ws, row, hin, ho, j, g);
      const float mu = __shfl(ho[0][0], j);
      float dmu;
      if (a.binary) {
        ls[2] = fmaxf(mu, 0.0f) - mu * xv + log1pf(expf(-fabsf(mu)));
        dmu = (1.0f / (1.0f + expf(-mu)) - xv) * invB;
      } else {
        ls[2] = (mu - xv) * (mu - xv);
        dmu = 2.0f * (mu - xv) * invB;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dho[0][r] = (4 * g + r == 0) ? dmu : 0.0f;
      ecb_head_bwd<T1, T2, T3>(th, tT, Hn, tab.c[ECB_H], ws, row, tile, dho, dhin, j, g);
      ecb_head_fwd<T1, T2, T3>(th, Hn, tab.c[ECB_HS], ws, row, hin, ho, j, g);
      float sg = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) sg += (4 * g + r == oh - 1) ? ho[0][r] : 0.0f;
      sg = sum_over_g(sg);
      ls[6] = sg * sg;
#pragma unroll
      for (int r = 0; r < 4; ++r) dho[0][r] = (4 * g + r == oh - 1) ? 0.001f * 2.0f * sg * invB : 0.0f;
      ecb_head_bwd<T1, T2, T3>(th, tT, Hn, tab.c[ECB_HS], ws, row, tile, dho, dhin2, j, g);
      dhin[0] += dhin2[0];
      *reinterpret_cast<f32x4 *>(dzt + row * 16 + 4 * g) = dhin[0];
      __threadfence_block();
#pragma unroll
      for (int tz = 0; tz < T0; ++tz)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * tz + 4 * g + r;
          const int src = f < z0 ? f : f - z1;
          const float t = dzt[row * 16 + min(max(src, 0), 15)];
          dz[tz][r] += (f < z0 || (f >= z0 + z1 && f < z0 + z1 + z2)) ? t : 0.0f;
        }
    }
#pragma unroll
    for (int t = 0; t < T0; ++t) *reinterpret_cast<f32x4 *>(dzh + row * ZW + 16 * t + 4 * g) = dz[t];
    __threadfence_block();
    if (lane == 0) flag[4 + tile] = 1;
  } else if (role == 3) {
    ech_fill_params<T1, T2, T3, T0>(par, P, a.theta_d, a.dz, tid - 384, 128);
    __threadfence_block();
    if (lane == 0) atomicAdd(const_cast<int *>(flag), 1);
    float sink = 0.0f;
    const int n4 = tab.n_warm >> 2;
    for (int arr = 0; arr < (tT ? 2 : 1); ++arr) {
      const f32x4 *src = reinterpret_cast<const f32x4 *>(arr == 0 ? th : tT);
      for (int i = tid - 384; i < n4; i += 128 * 16) {
        f32x4 acc4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 16; ++k) acc4 += src[min(i + 128 * k, n4 - 1)];
        sink += acc4[0] + acc4[1] + acc4[2] + acc4[3];
      }
    }
    asm volatile("" ::"v"(sink));
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float s = sum_over_j_to_lane15(ls[k]);
    if (j == 15 && g == 0) part[wave * 16 + k] = s;
  }
  __syncthreads();
  if (tid == 0 && a.out) {
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.0f; for (int w = 0; w < 6; ++w) if ((w & 1) < NB) s[k] += part[w * 16 + k]; }
    const float l_v = s[0] / (float)(B * p), l_z = s[1] / (float)(B * q), l_x = s[2] * invB, l_y = s[3] * invB;
    const float sig = (s[4] + s[5] + s[6]) * invB, adv = s[7] * invB;
    a.out[0] = adv; a.out[1] = l_v; a.out[2] = l_z; a.out[3] = l_x; a.out[4] = l_y;
    a.out[5] = adv + (l_v + zrec * l_z) + (l_x + l_y) + 0.001f * sig;
  }
}

// Parameter gradients + Adam of the Flipout nets: per 16x16 tile of a layer, loc gets sum_calls h^T cur, rho gets
// sum_calls (hs^T curs) * eps_call * sigmoid(rho) (bnn_bwd_params); bias from the column sums of cur.  The last workgroup
// reduces the per-tile sums of the input normalisations' gamma / beta.  Tile entry (ints): woff, n_in, n_out, u, v, xw, dw, ncalls,
// then per call: x, xs, d, ds, eps (workspace offsets of that layer).
template <class Args, int NB>
__device__ __forceinline__ void ecb_gen_dw(const Args &a, const EcbTab &tab, const int *tiles, float *thetaT) {
  constexpr int B = 16 * NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const float *ws = a.ws;
  auto adam = [&](int ei, float gi, int et) {
    a.grad[ei] = gi;
    if (a.apply) {
      const float mi = a.adam.b1 * a.m[ei] + (1.0f - a.adam.b1) * gi;
      const float vi = a.adam.b2 * a.v[ei] + (1.0f - a.adam.b2) * gi * gi;
      const float tn = a.theta[ei] - a.adam.lr_t * mi / (sqrtf(vi) + a.adam.eps);
      a.m[ei] = mi; a.v[ei] = vi; a.theta[ei] = tn;
      if (thetaT && et >= 0) thetaT[et] = tn;      // (no mirror in the Bayesian chains: both directions read the canonical array)
    }
  };
  const int n_tile_blocks = (tab.n_tiles + ECH_WAVES - 1) / ECH_WAVES;
  if ((int)blockIdx.x == n_tile_blocks) {     // gamma / beta of the four input normalisations
    for (int k = 0; k < 4; ++k) {
      const BnnNet &n = a.net[k];
      const int in = n.dims[0], w16 = 16 * tab.kt0[k];
      for (int f = tid; f < in; f += ECH_THREADS) {
        float sg = 0.0f, sb = 0.0f;
        for (int c = 0; c < tab.net_ncalls[k]; ++c) {
          const float *bp = ws + tab.c[tab.net_calls[k][c]].bnp;
          for (int t = 0; t < NB; ++t) { sg += bp[t * 2 * w16 + f]; sb += bp[t * 2 * w16 + w16 + f]; }
        }
        adam(n.off + f, sg, -1);
        adam(n.off + in + f, sb, -1);
      }
    }
    return;
  }
  const int tau = blockIdx.x * ECH_WAVES + wave;
  if (tau >= tab.n_tiles) return;
  const int *td = tiles + tau * ECB_TILE_INTS;
  const int woff = td[0], n_in = td[1], n_out = td[2], u = td[3], v = td[4], xw = td[5], dw = td[6], ncalls = td[7];
  const int o = 16 * v + j;
  f32x4 c1 = {0.0f, 0.0f, 0.0f, 0.0f}, rr = {0.0f, 0.0f, 0.0f, 0.0f};
  float bs = 0.0f;
  for (int c = 0; c < ncalls; ++c) {
    const int xo = td[8 + 5 * c], xso = td[9 + 5 * c], dofs = td[10 + 5 * c], dso = td[11 + 5 * c], eo = td[12 + 5 * c];
    f32x4 c2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s4 = 0; s4 < NB * 4; ++s4) {
      const int r_ = 4 * s4 + g;
      const float dv = ws[dofs + r_ * dw + o];
      c1 = BGM_MFMA(ws[xo + r_ * xw + 16 * u + j], dv, c1);
      c2 = BGM_MFMA(ws[xso + r_ * xw + 16 * u + j], ws[dso + r_ * dw + o], c2);
      bs += dv;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = min(16 * u + 4 * g + r, n_in - 1);
      rr[r] = fmaf(c2[r], ws[eo + f * n_out + min(o, n_out - 1)], rr[r]);      // eo: the call's dW of this layer; divided by sigma below
    }
  }
  bs = sum_over_g(bs);
  const int cnt = n_in * n_out;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 16 * u + 4 * g + r;
    if (f < n_in && o < n_out) {
      const int t = f * n_out + o;
      adam(woff + t, c1[r], woff + o * n_in + f);
      const float rho_ = a.theta[woff + cnt + t];
      adam(woff + cnt + t, rr[r] / (BNN_SCALE_EPS + softplus_acc(rho_)) * sigmoid_f(rho_), -1);      // sum_calls c2 eps, eps = dW / sigma
    }
  }
  if (u == 0 && g == 0 && o < n_out) adam(woff + 2 * cnt + o, bs, -1);
}


// =============================================================================================
// Iterative-update steps of CausalBGM with Bayesian nets as row-tile chains (bnn_theta_step_kernel / bnn_z_grad_kernel):
// update_g_net / update_h_net / update_f_net (causalbgm/base.py:156-243) and update_latent_variable_sgd (:246-302).
// Calls of the theta step: [0] g, [1] h, [2] f (one noise stream); of the latent step: [0,1] g, [2,3] h, [4,5] f (mean call on the
// step's stream, variance-head call on stream + 1; a binary treatment head has no variance call).
// =============================================================================================
__device__ __forceinline__ float ecb_gauss(float ssq, float raw, float dim, float &loss_b, float &s2, float fix2) {   // bnn_gauss
  if (fix2 > 0.0f) {          // fixed params['sigma_*']: the variance head is not read and gets no gradient
    s2 = fix2;
    loss_b = ssq / (2.0f * s2) + dim * logf(s2) * 0.5f;
    return 0.0f;
  }
  s2 = softplus_acc(raw) + BGM_EPS;
  loss_b = ssq / (2.0f * s2) + dim * logf(s2) * 0.5f;
  return (-ssq / (2.0f * s2 * s2) + dim / (2.0f * s2)) * sigmoid_f(raw);
}
// value of feature `f` of a one-row tile set (held by one lane group), in every lane of the row
template <int NT>
__device__ __forceinline__ float ecb_pick(const f32x4 (&x)[NT], int f, int g) {
  float v = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) v += (16 * t + 4 * g + r == f) ? x[t][r] : 0.0f;
  return sum_over_g(v);
}
// the data row of a minibatch row as NTL register tiles (zero beyond p), requested up front: a random row of the panel is a trip to HBM,
// and loaded where the likelihood consumes it that latency sits on the chain's critical path
template <int NTL>
__device__ __forceinline__ void ecb_load_row(const float *vrow, int p, int g, f32x4 (&vv)[NTL]) {
  if ((p & 3) == 0 && (reinterpret_cast<unsigned long long>(vrow) & 15ull) == 0) {      // one 16-byte request per tile and lane
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int f = 16 * t + 4 * g;
      const f32x4 x_ = *reinterpret_cast<const f32x4 *>(vrow + min(f, p - 4));
      vv[t] = f < p ? x_ : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  } else {
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) vv[t][r] = ech_ld(vrow, 16 * t + 4 * g + r, p);
  }
}
// inputs of the three nets from a row of the latent table
template <int T0>
__device__ __forceinline__ void ecb_inputs(const float *zrow, float xv, int q, int z0, int z1, int z2, int g, f32x4 (&zin)[T0], f32x4 (&fin)[1],
                                           f32x4 (&hin)[1]) {
#pragma unroll
  for (int t = 1; t < T0; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) zin[t][r] = ech_ld(zrow, 16 * t + 4 * g + r, q);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 4 * g + r;
    zin[0][r] = ech_ld(zrow, f, q);
    fin[0][r] = f < z0 + z1 ? ech_ld(zrow, f, q) : (f == z0 + z1 ? xv : 0.0f);
    const float t = ech_ld(zrow, f < z0 ? f : f + z1, q);
    hin[0][r] = f < z0 + z2 ? t : 0.0f;
  }
}

// WS (NB = 1, HT = 4, T0 = 1): g's last layer on the workgroup's idle waves, as fit_chain_kernel<WS> does for the deterministic nets --
// here every column group is TWO products (loc and the call's perturbation) and two sign flips.  LDS behind the 64 loss words:
// h | hs [16 x 64 each] | residual partials [2][4][16] | backward partials [4][16 x 64] | three counters.
#define ECB_WS_LDS_FLOATS (64 + 2 * 1024 + 128 + 4 * 1024 + 8)
template <class Args, int HT, int NTL, int T1, int T2, int T3, int NB, bool PAD = false, int T0 = 1, bool WS = false>
__device__ __forceinline__ void ecb_theta_chain(const Args &a, const EcbTab &tab, float *ws, float *lds, unsigned *kl_cnt = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int q = a.q, p = a.p;
  // a minibatch may be spread over workgroups: blockIdx.x owns NB row tiles, blockIdx.y (gridDim.y = 3) one network -- on one CU the
  // six chain waves share one address unit for their weight streams; a.out is then accumulated (zeroed by the noise launch)
  const int role = wave >> 1, ltile = wave & 1, tile = NB * (int)blockIdx.x + ltile;
  const bool active = ltile < NB && (gridDim.y == 1 || role == (int)blockIdx.y);
  const int row = 16 * tile + j;
  const float *th = a.theta;
  float *part = lds;                 // [8 waves][4]: loss, aux
  float ls0 = 0.0f, ls1 = 0.0f;
  if constexpr (WS) {
    static_assert(NB == 1 && HT == 4 && T0 == 1 && !PAD, "worker split: one plain row tile per workgroup");
    constexpr int NW = (NTL + 3) / 4;
    float *hb = lds + 64, *hsb = hb + 1024, *pp = hsb + 1024, *dhp = pp + 128;
    int *fl = reinterpret_cast<int *>(dhp + 4096);
    if (tid < 8) fl[tid] = 0;
    __syncthreads();
#define ECB_WAIT(f, need) { while (__hip_atomic_load(fl + (f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (need)) __builtin_amdgcn_s_sleep(1); __threadfence_block(); }
#define ECB_POST(f) { __threadfence_block(); if (lane == 0) __hip_atomic_fetch_add(fl + (f), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    const int wk = wave == 0 ? 0 : ((wave & 1) && wave < 6 ? 1 + (wave >> 1) : -1);       // waves 0 | 1, 3, 5
    if ((gridDim.y == 1 || blockIdx.y == 0) && wk >= 0 && wk < NW) {
      const BnnNet &G = a.net[BNN_G];
      const EcbCall &C = tab.c[0];
      const int L = G.n_layers, no = G.dims[L];
      const int wrow = 16 * (int)blockIdx.x + j;
      const long long prow = a.idx[wrow];
      const uint32_t *roww = reinterpret_cast<const uint32_t *>(ws + C.sg) + (long long)wrow * G.swords;
      const float *loc = th + G.woff[L - 1], *dWl = ws + C.dW + G.eoff[L - 1];
      f32x4 vv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[t][r] = ech_ld(a.v_ + prow * p, 64 * wk + 16 * t + 4 * g + r, p);
      f32x4 h[4], hs[4];
      if (wk == 0) {
        f32x4 zin[1], fin[1], hin[1];
        ecb_inputs<1>(a.data_z + prow * q, a.x_[prow], q, a.z0, a.z1, a.z2, g, zin, fin, hin);
        ecb_mlp_fwd_hidden<1, 4, false>(th, G, C, ws, wrow, zin, h, hs, j, g);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          *reinterpret_cast<f32x4 *>(hb + j * 64 + 16 * t + 4 * g) = h[t];
          *reinterpret_cast<f32x4 *>(hsb + j * 64 + 16 * t + 4 * g) = hs[t];
        }
        ECB_POST(0);
      } else {
        ECB_WAIT(0, 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          h[t] = *reinterpret_cast<const f32x4 *>(hb + j * 64 + 16 * t + 4 * g);
          hs[t] = *reinterpret_cast<const f32x4 *>(hsb + j * 64 + 16 * t + 4 * g);
        }
      }
      f32x4 o4[4], c2[4], c2s[4];
      ech_zero<4>(o4);
      ech_zero<4>(c2);
      {
        const EcgW w1{loc, no, 64, no, 64 * wk}, w2{dWl, no, 64, no, 64 * wk};
        EcgA<4> A, Ad, Ad2;
        ecg_prime<4, false>(w1, A, j, g);
        ecg_sub<4, 4, 4, false, false>(w1, h, o4, A, w2, Ad, j, g);
        ecg_sub<4, 4, 4, false, false>(w2, hs, c2, Ad, w2, Ad2, j, g);
      }
      ecb_flip<4>(roww, G.sout_w[L - 1] + 2 * wk, g, c2, c2s);
#pragma unroll
      for (int u = 0; u < 4; ++u) o4[u] += c2s[u];
      ecg_bias<4>(loc + 2 * 64 * no, no, 64 * wk, g, o4);
      float ssq = 0.0f, raw = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 64 * wk + 16 * t + 4 * g + r;
          const float d = f < p ? vv[t][r] - o4[t][r] : 0.0f;
          ssq = fmaf(d, d, ssq);
          raw += f == p ? o4[t][r] : 0.0f;
          o4[t][r] = d;
        }
      ssq = sum_over_g(ssq);
      raw = sum_over_g(raw);
      if (g == 0) { pp[wk * 16 + j] = ssq; pp[64 + wk * 16 + j] = raw; }
      ECB_POST(1);
      ECB_WAIT(1, NW);
      ssq = 0.0f; raw = 0.0f;
#pragma unroll
      for (int k = 0; k < NW; ++k) { ssq += pp[k * 16 + j]; raw += pp[64 + k * 16 + j]; }
      float lb, s2;
      const float dr = ecb_gauss(ssq, raw, (float)p, lb, s2, a.sig2[0]);
      if (wk == 0) { ls0 = lb; ls1 = ssq; }
      f32x4 douts[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 64 * wk + 16 * t + 4 * g + r;
          o4[t][r] = f < p ? -o4[t][r] / s2 * a.inv_B : (f == p ? dr * a.inv_B : 0.0f);
        }
      ecb_flip<4>(roww, G.sout_w[L - 1] + 2 * wk, g, o4, douts);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (4 * wk + t < NTL) {
          *reinterpret_cast<f32x4 *>(ws + C.d[L - 1] + (long long)wrow * (16 * NTL) + 16 * (4 * wk + t) + 4 * g) = o4[t];
          *reinterpret_cast<f32x4 *>(ws + C.ds[L - 1] + (long long)wrow * (16 * NTL) + 16 * (4 * wk + t) + 4 * g) = douts[t];
        }
      // backward partial over this worker's 64 output features (K-contiguous reads of the canonical arrays, offset by 64 wk)
      f32x4 dh[4];
      {
        const EcgW w1{loc + 64 * wk, no, no, 64, 0}, w2{dWl + 64 * wk, no, no, 64, 0};
        EcgA<4> A, Ad, Ad2;
        ecg_prime<4, true, true>(w1, A, j, g);
        ech_zero<4>(dh);
        ech_zero<4>(c2);
        ecg_sub<4, 4, 4, true, true, true, true>(w1, o4, dh, A, w2, Ad, j, g);
        ecg_sub<4, 4, 4, true, true, true, true>(w2, douts, c2, Ad, w2, Ad2, j, g);
        ecb_flip<4>(roww, G.sin_w[L - 1], g, c2, c2s);
#pragma unroll
        for (int u = 0; u < 4; ++u) dh[u] += c2s[u];
      }
      if (wk != 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4 *>(dhp + wk * 1024 + j * 64 + 16 * u + 4 * g) = dh[u];
        ECB_POST(2);
      } else {
        ECB_WAIT(2, NW - 1);
#pragma unroll
        for (int k = 1; k < NW; ++k)
#pragma unroll
          for (int u = 0; u < 4; ++u) dh[u] += *reinterpret_cast<const f32x4 *>(dhp + k * 1024 + j * 64 + 16 * u + 4 * g);
        ecg_mask<4>(dh, h);
        f32x4 dnone[1];
        ecb_mlp_bwd_hidden<1, 4, false, false>(th, G, C, ws, wrow, (int)blockIdx.x, dh, dnone, j, g);
      }
    }
  }
  if (role < 3 && active && !(WS && role == 0)) {
    const long long prow = a.idx[row];
    const float xv = a.x_[prow], yv = a.y_[prow];
    f32x4 zin[T0], fin[1], hin[1];
    ecb_inputs<T0>(a.data_z + prow * q, xv, q, a.z0, a.z1, a.z2, g, zin, fin, hin);
    if (role == 0) {
      const BnnNet &G = a.net[BNN_G];
      f32x4 o[NTL], vv[NTL];
      ecb_load_row<NTL>(a.v_ + prow * p, p, g, vv);
      ecb_mlp_fwd<T0, HT, NTL, PAD>(th, G, tab.c[0], ws, row, zin, o, j, g);
      float ssq = 0.0f;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * g + r;
          const float d = f < p ? vv[t][r] - o[t][r] : 0.0f;
          ssq = fmaf(d, d, ssq);
        }
      ssq = sum_over_g(ssq);
      const float raw = ecb_pick<NTL>(o, p, g);
      float lb, s2;
      const float dr = ecb_gauss(ssq, raw, (float)p, lb, s2, a.sig2[0]);
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * g + r;
          o[t][r] = f < p ? -(vv[t][r] - o[t][r]) / s2 * a.inv_B : (f == p ? dr * a.inv_B : 0.0f);
        }
      ls0 = lb; ls1 = ssq;
      f32x4 dnone[T0];
      ecb_mlp_bwd<T0, HT, NTL, false, PAD>(th, th, G, tab.c[0], ws, row, tile, o, dnone, j, g);
    } else {
      const bool is_h = role == 1;
      const BnnNet &N = a.net[is_h ? BNN_H : BNN_F];
      const EcbCall &C = tab.c[is_h ? 1 : 2];
      f32x4 o[1], d[1], dnone[1];
      ecb_head_fwd<T1, T2, T3>(th, N, C, ws, row, is_h ? hin : fin, o, j, g);
      const int wo = N.dims[N.n_layers];
      const float l = __shfl(o[0][0], j), raw = ecb_pick<1>(o, wo - 1, g), tgt = is_h ? xv : yv;
      float d0, dl = 0.0f;
      if (is_h && a.binary) {
        const float e = fmaxf(l, 0.0f) - l * tgt + log1pf(expf(-fabsf(l)));
        ls0 = e; ls1 = e;
        d0 = (sigmoid_f(l) - tgt) * a.inv_B;
      } else {
        const float r_ = tgt - l;
        float lb, s2;
        const float dr = ecb_gauss(r_ * r_, raw, 1.0f, lb, s2, a.sig2[is_h ? 1 : 2]);
        ls0 = lb; ls1 = r_ * r_;
        d0 = -r_ / s2 * a.inv_B;
        dl = dr * a.inv_B;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int f = 4 * g + r; d[0][r] = (f == 0 ? d0 : 0.0f) + (f == wo - 1 ? dl : 0.0f); }
      ecb_head_bwd<T1, T2, T3>(th, th, N, C, ws, row, tile, d, dnone, j, g);
    }
  }
  {
    const float s0 = sum_over_j_to_lane15(ls0), s1 = sum_over_j_to_lane15(ls1);
    if (j == 15 && g == 0) { part[wave * 4] = s0; part[wave * 4 + 1] = s1; }
  }
  __syncthreads();
  if (tid < 3 && a.out) {            // which = 0 g, 1 h, 2 f
    float kl = 0.0f;             // call tid of the theta step is net which = tid; its KL partials come from the noise launch
    if (kl_cnt) {                // the partials come from rider workgroups of this very launch (ecb_rider)
      while (__hip_atomic_load(kl_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 3u * ECB_NOISE_PARTS) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int t = 0; t < ECB_NOISE_PARTS; ++t) kl += __hip_atomic_load(ws + tab.klp + tid * ECB_NOISE_PARTS + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else
    for (int t = 0; t < ECB_NOISE_PARTS; ++t) kl += ws[tab.klp + tid * ECB_NOISE_PARTS + t];
    float l0 = 0.0f, l1 = 0.0f;
    for (int w = 2 * tid; w < 2 * tid + NB; ++w) { l0 += part[w * 4]; l1 += part[w * 4 + 1]; }
    const bool kl_here = blockIdx.x == 0 && (gridDim.y == 1 || (int)blockIdx.y == tid);
    if (gridDim.x * gridDim.y == 1) {
      a.out[2 * tid] = l0 * a.inv_B + a.kl_weight * kl;
      a.out[2 * tid + 1] = tid == 0 ? l1 * a.inv_B / (float)p : l1 * a.inv_B;
    } else {
      atomicAdd(a.out + 2 * tid, l0 * a.inv_B + (kl_here ? a.kl_weight * kl : 0.0f));
      atomicAdd(a.out + 2 * tid + 1, tid == 0 ? l1 * a.inv_B / (float)p : l1 * a.inv_B);
    }
  }
}

// gradient tiles of the theta step: ecb_gen_dw with one call per net, plus the KL terms of bnn_kl; Adam when a.apply.
// Two phases: everything that only READS (products, the old parameters and Adam slots, the new values) first; then `late` -- the wait
// for the previous latent phase, which reads the parameters this kernel is about to overwrite (bgm_bnn_fit_epoch; NULL: none) --
// and behind it the stores alone: the products and the loads run beside the latent chains instead of behind them.
template <class Args, int NB>
__device__ __forceinline__ void ecb_theta_dw(const Args &a, const EcbTab &tab, const int *tiles, const float *ws, const EcbAhead &ah = EcbAhead{},
                                             const FitSync *late = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  auto calc = [&](int ei, float gi) { return a.apply ? bnn_adam_one(a.theta[ei], a.m[ei], a.v[ei], gi, a.adam) : BnnAdamOut{0.0f, 0.0f, 0.0f}; };
  auto put = [&](int ei, float gi, const BnnAdamOut &o) {
    a.grad[ei] = gi;
    if (a.apply) { a.m[ei] = o.m; a.v[ei] = o.v; a.theta[ei] = o.th; }
  };
  const int n_tile_blocks = (tab.n_tiles + ECH_WAVES - 1) / ECH_WAVES;
  if ((int)blockIdx.x == n_tile_blocks) {              // input-normalisation parameters (a few hundred), the loss words, the KL counter
    if (late) fit_sync_wait(*late);
    if (tid == 0 && ah.kl_cnt) *ah.kl_cnt = 0u;
    if (tid < 6 && ah.zero_t) ah.zero_t[tid] = 0.0f;
    if (tid == 0 && ah.zero_z) ah.zero_z[0] = 0.0f;
    for (int k = 0; k < 4; ++k) {
      if (tab.net_ncalls[k] == 0) continue;
      const BnnNet &n = a.net[k];
      const int in = n.dims[0], w16 = 16 * tab.kt0[k];
      for (int f = tid; f < in; f += ECH_THREADS) {
        float sg = 0.0f, sb = 0.0f;
        for (int c = 0; c < tab.net_ncalls[k]; ++c) {
          const float *bp = ws + tab.c[tab.net_calls[k][c]].bnp;
          for (int t = 0; t < NB; ++t) { sg += bp[t * 2 * w16 + f]; sb += bp[t * 2 * w16 + w16 + f]; }
        }
        put(n.off + f, sg, calc(n.off + f, sg));
        put(n.off + in + f, sb, calc(n.off + in + f, sb));
      }
    }
    return;
  }
  const int tau_ = blockIdx.x * ECH_WAVES + wave;
  const bool live = tau_ < tab.n_tiles;               // (every wave reaches the wait below)
  const int tau = live ? tau_ : tab.n_tiles - 1;
  const int *td = tiles + tau * ECB_TILE_INTS;
  const int woff = td[0], n_in = td[1], n_out = td[2], u = td[3], v = td[4], xw = td[5], dw = td[6], ncalls = td[7], net = td[23];
  const float iv = a.net[net].prior_iv, klw = a.kl_weight;
  const int bias_prior = a.net[net].bias_prior;
  const int o = 16 * v + j;
  const int cnt = n_in * n_out;
  // this lane's four weights: old parameters, Adam slots and (prepared steps, EcbAhead) standard normals, requested with the activations
  const bool sc_t = ah.ws_next != nullptr, sc_z = ah.ws_z != nullptr;
  const int el = td[12] - ah.dw_t[net];                // the layer's offset inside a call's perturbations
  float mu[4], rho_[4], ml[4], vl[4], mr[4], vr[4], e_t[4], e_z0[4], e_z1[4];
  bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 16 * u + 4 * g + r;
    ok[r] = live && f < n_in && o < n_out;
    const int t = min(f, n_in - 1) * n_out + min(o, n_out - 1);
    mu[r] = a.theta[woff + t]; rho_[r] = a.theta[woff + cnt + t];
    ml[r] = a.m[woff + t]; vl[r] = a.v[woff + t]; mr[r] = a.m[woff + cnt + t]; vr[r] = a.v[woff + cnt + t];
    e_t[r] = sc_t ? ah.ws_next[td[12] + t] : 0.0f;
    e_z0[r] = sc_z ? ah.ws_z[ah.dw_z[net][0] + el + t] : 0.0f;
    e_z1[r] = sc_z ? ah.ws_z[ah.dw_z[net][1] + el + t] : 0.0f;
  }
  const int eb = woff + 2 * cnt + min(o, n_out - 1);
  const float b_old = a.theta[eb], mb = a.m[eb], vb = a.v[eb];
  f32x4 c1 = {0.0f, 0.0f, 0.0f, 0.0f}, rr = {0.0f, 0.0f, 0.0f, 0.0f};
  float bs = 0.0f;
  for (int c = 0; c < ncalls; ++c) {
    const int xo = td[8 + 5 * c], xso = td[9 + 5 * c], dofs = td[10 + 5 * c], dso = td[11 + 5 * c], eo = td[12 + 5 * c];
    f32x4 c2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s4 = 0; s4 < NB * 4; ++s4) {
      const int r_ = 4 * s4 + g;
      const float dv = ws[dofs + r_ * dw + o];
      c1 = BGM_MFMA(ws[xo + r_ * xw + 16 * u + j], dv, c1);
      c2 = BGM_MFMA(ws[xso + r_ * xw + 16 * u + j], ws[dso + r_ * dw + o], c2);
      bs += dv;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = min(16 * u + 4 * g + r, n_in - 1);
      rr[r] = fmaf(c2[r], ws[eo + f * n_out + min(o, n_out - 1)], rr[r]);
    }
  }
  bs = sum_over_g(bs);
  float gl[4], gr[4], s_t[4], s_z0[4], s_z1[4];
  BnnAdamOut ol[4], orh[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float sg = BNN_SCALE_EPS + softplus_acc(rho_[r]), sgm = sigmoid_f(rho_[r]);
    gl[r] = c1[r] + klw * mu[r] * iv;
    gr[r] = rr[r] / sg * sgm + klw * (-1.0f / sg + sg * iv) * sgm;
    ol[r] = a.apply ? bnn_adam_one(mu[r], ml[r], vl[r], gl[r], a.adam) : BnnAdamOut{0.0f, 0.0f, 0.0f};
    orh[r] = a.apply ? bnn_adam_one(rho_[r], mr[r], vr[r], gr[r], a.adam) : BnnAdamOut{0.0f, 0.0f, 0.0f};
    const float sn = BNN_SCALE_EPS + softplus_f(a.apply ? orh[r].th : rho_[r]);      // sigma of the NEW rho (ecb_noise's product)
    s_t[r] = sn * e_t[r]; s_z0[r] = sn * e_z0[r]; s_z1[r] = sn * e_z1[r];
  }
  const bool bias_here = live && u == 0 && g == 0 && o < n_out;
  const float gb = bs + (bias_prior ? klw * b_old * iv : 0.0f);
  const BnnAdamOut ob = a.apply ? bnn_adam_one(b_old, mb, vb, gb, a.adam) : BnnAdamOut{0.0f, 0.0f, 0.0f};
  if (late) fit_sync_wait(*late);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (ok[r]) {
      const int t = (16 * u + 4 * g + r) * n_out + o;
      put(woff + t, gl[r], ol[r]);
      put(woff + cnt + t, gr[r], orh[r]);
      if (sc_t) ah.ws_next[td[12] + t] = s_t[r];
      if (sc_z) { ah.ws_z[ah.dw_z[net][0] + el + t] = s_z0[r]; ah.ws_z[ah.dw_z[net][1] + el + t] = s_z1[r]; }
    }
  }
  if (bias_here) put(eb, gb, ob);
}

// latent step: dz [B x q] = d loss / d (batch rows of data_z), out[0] = loss_postrior_z.  Waves 0,1: g mean call; 2,3: g variance-head
// call (they exchange the row's sum of squares and raw variance through LDS); 4,5: h (both calls); 6,7: f (both calls).
template <class Args, int HT, int NTL, int T1, int T2, int T3, int NB, bool PAD = false, int T0 = 1, bool WS = false>
__device__ __forceinline__ void ecb_z_chain(const Args &a, const EcbTab &tab, float *ws, float *lds, const EcbZRows &zr = EcbZRows{}) {
  constexpr int B = 16 * NB, ZW = 16 * T0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int q = a.q, p = a.p, z0 = a.z0, z1 = a.z1, z2 = a.z2;
  // blockIdx.x owns NB row tiles of the minibatch (two workgroups of one tile each instead of one of two: the chain waves of a CU share
  // its address unit): `lt` / `lrow` index this workgroup's LDS, `tile` / `row` the minibatch
  const int role = wave >> 1, lt = wave & 1, tile = NB * (int)blockIdx.x + lt;
  const bool active = lt < NB;
  const int lrow = 16 * lt + j, row = 16 * tile + j;
  const float *th = a.theta;
  float *part = lds;                                   // [8 waves] loss partials
  volatile int *flag = reinterpret_cast<volatile int *>(lds + 16);    // [0 + tile]: ssq written; [2 + tile]: raw written
  float *xch = lds + 32;                               // [B] ssq | [B] raw of the g calls
  float *dzc = lds + 32 + 2 * B;                       // [4 roles][B x 16 T0] input gradients (the head nets use the first 16 columns)
  if (tid < 16) flag[tid] = 0;
  __syncthreads();
  float lsum = 0.0f;
  // WS (NB = 1): the mean call's last layer over waves 0 | 1, 3, 5 (as ecb_theta_chain<WS>); the variance-head call (wave 2) needs ONE
  // output column of its last layer -- it multiplies that column's tile alone, forward and backward.  LDS behind the latent-gradient
  // tiles: h | hs [16 x 64] | residual partials [4][16] | backward partials [4][16 x 64] | three counters.
  if constexpr (WS) {
    static_assert(NB == 1 && HT == 4 && T0 == 1 && !PAD, "worker split: one plain row tile per workgroup");
    constexpr int NW = (NTL + 3) / 4;
    float *hb = dzc + 4 * B * ZW, *hsb = hb + 1024, *pp = hsb + 1024, *dhp = pp + 64;
    int *fl = reinterpret_cast<int *>(dhp + 4096);
    if (tid < 8) fl[tid] = 0;
    __syncthreads();
    const int wk = wave == 0 ? 0 : ((wave & 1) && wave < 6 ? 1 + (wave >> 1) : -1);       // waves 0 | 1, 3, 5
    if (wk >= 0 && wk < NW) {
      const BnnNet &G = a.net[BNN_G];
      const EcbCall &C = tab.c[0];
      const int L = G.n_layers, no = G.dims[L];
      const int wrow = 16 * (int)blockIdx.x + j;
      const long long prow = a.idx[wrow];
      const uint32_t *roww = reinterpret_cast<const uint32_t *>(ws + C.sg) + (long long)wrow * G.swords;
      const float *loc = th + G.woff[L - 1], *dWl = ws + C.dW + G.eoff[L - 1];
      f32x4 vv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[t][r] = ech_ld(a.v_ + prow * p, 64 * wk + 16 * t + 4 * g + r, p);
      f32x4 h[4], hs[4], zin[1], fin[1], hin[1];
      if (wk == 0) {
        ecb_inputs<1>(a.data_z + prow * q, a.x_[prow], q, z0, z1, z2, g, zin, fin, hin);
        ecb_mlp_fwd_hidden<1, 4, false>(th, G, C, ws, wrow, zin, h, hs, j, g);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          *reinterpret_cast<f32x4 *>(hb + j * 64 + 16 * t + 4 * g) = h[t];
          *reinterpret_cast<f32x4 *>(hsb + j * 64 + 16 * t + 4 * g) = hs[t];
        }
        ECB_POST(0);
      } else {
        ECB_WAIT(0, 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          h[t] = *reinterpret_cast<const f32x4 *>(hb + j * 64 + 16 * t + 4 * g);
          hs[t] = *reinterpret_cast<const f32x4 *>(hsb + j * 64 + 16 * t + 4 * g);
        }
      }
      f32x4 o4[4], c2[4], c2s[4];
      ech_zero<4>(o4);
      ech_zero<4>(c2);
      {
        const EcgW w1{loc, no, 64, no, 64 * wk}, w2{dWl, no, 64, no, 64 * wk};
        EcgA<4> A, Ad, Ad2;
        ecg_prime<4, false>(w1, A, j, g);
        ecg_sub<4, 4, 4, false, false>(w1, h, o4, A, w2, Ad, j, g);
        ecg_sub<4, 4, 4, false, false>(w2, hs, c2, Ad, w2, Ad2, j, g);
      }
      ecb_flip<4>(roww, G.sout_w[L - 1] + 2 * wk, g, c2, c2s);
#pragma unroll
      for (int u = 0; u < 4; ++u) o4[u] += c2s[u];
      ecg_bias<4>(loc + 2 * 64 * no, no, 64 * wk, g, o4);
      float ssq = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 64 * wk + 16 * t + 4 * g + r;
          const float d = f < p ? vv[t][r] - o4[t][r] : 0.0f;
          ssq = fmaf(d, d, ssq);
          o4[t][r] = d;
        }
      ssq = sum_over_g(ssq);
      if (g == 0) pp[wk * 16 + j] = ssq;
      ECB_POST(1);
      ECB_WAIT(1, NW);
      ssq = 0.0f;
#pragma unroll
      for (int k = 0; k < NW; ++k) ssq += pp[k * 16 + j];
      if (wk == 0) {                                  // hand the row's sum of squares to the variance-head call
        if (g == 0) xch[j] = ssq;
        __threadfence_block();
        if (lane == 0) flag[0] = 1;
      }
      while (flag[2] == 0) __builtin_amdgcn_s_sleep(2);
      __threadfence_block();
      const float raw = xch[B + j];
      float lb, s2;
      ecb_gauss(ssq, raw, (float)p, lb, s2, a.sig2[0]);
      f32x4 douts[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[t][r] = (64 * wk + 16 * t + 4 * g + r < p) ? -o4[t][r] / s2 * a.inv_B : 0.0f;
      ecb_flip<4>(roww, G.sout_w[L - 1] + 2 * wk, g, o4, douts);
      f32x4 dh[4];
      {
        const EcgW w1{loc + 64 * wk, no, no, 64, 0}, w2{dWl + 64 * wk, no, no, 64, 0};
        EcgA<4> A, Ad, Ad2;
        ecg_prime<4, true, true>(w1, A, j, g);
        ech_zero<4>(dh);
        ech_zero<4>(c2);
        ecg_sub<4, 4, 4, true, true, true, true>(w1, o4, dh, A, w2, Ad, j, g);
        ecg_sub<4, 4, 4, true, true, true, true>(w2, douts, c2, Ad, w2, Ad2, j, g);
        ecb_flip<4>(roww, G.sin_w[L - 1], g, c2, c2s);
#pragma unroll
        for (int u = 0; u < 4; ++u) dh[u] += c2s[u];
      }
      if (wk != 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4 *>(dhp + wk * 1024 + j * 64 + 16 * u + 4 * g) = dh[u];
        ECB_POST(2);
      } else {
        ECB_WAIT(2, NW - 1);
#pragma unroll
        for (int k = 1; k < NW; ++k)
#pragma unroll
          for (int u = 0; u < 4; ++u) dh[u] += *reinterpret_cast<const f32x4 *>(dhp + k * 1024 + j * 64 + 16 * u + 4 * g);
        ecg_mask<4>(dh, h);
        f32x4 dx[1];
        ecb_mlp_bwd_hidden<1, 4, true, false>(th, G, C, ws, wrow, (int)blockIdx.x, dh, dx, j, g);
        *reinterpret_cast<f32x4 *>(dzc + (0 * B + j) * ZW + 4 * g) = dx[0];
        float zz = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) zz = fmaf(zin[0][r], zin[0][r], zz);
        lsum = lb + 0.5f * sum_over_g(zz);             // the row's likelihood term of v and its prior term
      }
    }
  }
  if (active && !(WS && role == 0)) {
    const long long prow = a.idx[row];
    const float xv = a.x_[prow], yv = a.y_[prow];
    const float *zrow = a.data_z + prow * q;
    f32x4 zin[T0], fin[1], hin[1], dx[T0];
    ecb_inputs<T0>(zrow, xv, q, z0, z1, z2, g, zin, fin, hin);
    ech_zero<T0>(dx);
    if (role == 0) {
      const BnnNet &G = a.net[BNN_G];
      f32x4 o[NTL], vv[NTL];
      ecb_load_row<NTL>(a.v_ + prow * p, p, g, vv);
      ecb_mlp_fwd<T0, HT, NTL, PAD>(th, G, tab.c[0], ws, row, zin, o, j, g);
      float ssq = 0.0f;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * g + r;
          const float d = f < p ? vv[t][r] - o[t][r] : 0.0f;
          ssq = fmaf(d, d, ssq);
          o[t][r] = d;
        }
      ssq = sum_over_g(ssq);
      if (g == 0) xch[lrow] = ssq;
      __threadfence_block();
      if (lane == 0) flag[lt] = 1;
      while (flag[2 + lt] == 0) __builtin_amdgcn_s_sleep(2);
      __threadfence_block();
      const float raw = xch[B + lrow];
      float lb, s2;
      ecb_gauss(ssq, raw, (float)p, lb, s2, a.sig2[0]);
      lsum = lb;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t][r] = (16 * t + 4 * g + r < p) ? -o[t][r] / s2 * a.inv_B : 0.0f;
      ecb_mlp_bwd<T0, HT, NTL, true, PAD>(th, th, G, tab.c[0], ws, row, tile, o, dx, j, g);
    } else if (role == 1) {
     if constexpr (WS) {
      // the variance-head call: only output column p of the last layer is used -- its tile alone, forward and backward
      const BnnNet &G = a.net[BNN_G];
      const EcbCall &C = tab.c[1];
      const int L = G.n_layers, no = G.dims[L], tp = p >> 4;
      const uint32_t *roww = reinterpret_cast<const uint32_t *>(ws + C.sg) + (long long)row * G.swords;
      const float *loc = th + G.woff[L - 1], *dWl = ws + C.dW + G.eoff[L - 1];
      f32x4 h[4], hs[4];
      ecb_mlp_fwd_hidden<1, 4, false>(th, G, C, ws, row, zin, h, hs, j, g);
      f32x4 o1[1], c2[1];
      ech_zero<1>(o1);
      ech_zero<1>(c2);
      {
        const EcgW w1{loc, no, 64, no, 16 * tp}, w2{dWl, no, 64, no, 16 * tp};
        EcgA<1> A, Ad, Ad2;
        ecg_prime<1, false>(w1, A, j, g);
        ecg_sub<4, 1, 1, false, false>(w1, h, o1, A, w2, Ad, j, g);
        ecg_sub<4, 1, 1, false, false>(w2, hs, c2, Ad, w2, Ad2, j, g);
      }
      const uint32_t so = roww[G.sout_w[L - 1] + (tp >> 1)] >> (16 * (tp & 1) + 4 * g);       // sign bits of tile tp (ecb_flip's layout)
      float raw = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = o1[0][r] + __uint_as_float(__float_as_uint(c2[0][r]) ^ (((so >> r) & 1u) << 31)) + ech_ld(loc + 2 * 64 * no, 16 * tp + 4 * g + r, no);
        raw += (16 * tp + 4 * g + r == p) ? v : 0.0f;
      }
      raw = sum_over_g(raw);
      if (g == 0) xch[B + lrow] = raw;
      __threadfence_block();
      if (lane == 0) flag[2 + lt] = 1;
      while (flag[lt] == 0) __builtin_amdgcn_s_sleep(2);
      __threadfence_block();
      const float ssq = xch[lrow];
      float lb, s2;
      const float dr = ecb_gauss(ssq, raw, (float)p, lb, s2, a.sig2[0]);
      f32x4 d1[1], d1s[1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d1[0][r] = (16 * tp + 4 * g + r == p) ? dr * a.inv_B : 0.0f;
        d1s[0][r] = __uint_as_float(__float_as_uint(d1[0][r]) ^ (((so >> r) & 1u) << 31));
      }
      f32x4 dh[4], c4[4], c4s[4];
      {
        const EcgW w1{loc + 16 * tp, no, no, 64, 0}, w2{dWl + 16 * tp, no, no, 64, 0};
        EcgA<4> A, Ad, Ad2;
        ecg_prime<4, true, true>(w1, A, j, g);
        ech_zero<4>(dh);
        ech_zero<4>(c4);
        ecg_sub<1, 4, 4, true, true, true, true>(w1, d1, dh, A, w2, Ad, j, g);
        ecg_sub<1, 4, 4, true, true, true, true>(w2, d1s, c4, Ad, w2, Ad2, j, g);
        ecb_flip<4>(roww, G.sin_w[L - 1], g, c4, c4s);
#pragma unroll
        for (int u = 0; u < 4; ++u) dh[u] += c4s[u];
      }
      ecg_mask<4>(dh, h);
      ecb_mlp_bwd_hidden<1, 4, true, false>(th, G, C, ws, row, tile, dh, dx, j, g);
     } else {
      const BnnNet &G = a.net[BNN_G];
      f32x4 o[NTL];
      ecb_mlp_fwd<T0, HT, NTL, PAD>(th, G, tab.c[1], ws, row, zin, o, j, g);
      const float raw = ecb_pick<NTL>(o, p, g);
      if (g == 0) xch[B + lrow] = raw;
      __threadfence_block();
      if (lane == 0) flag[2 + lt] = 1;
      while (flag[lt] == 0) __builtin_amdgcn_s_sleep(2);
      __threadfence_block();
      const float ssq = xch[lrow];
      float lb, s2;
      const float dr = ecb_gauss(ssq, raw, (float)p, lb, s2, a.sig2[0]);
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t][r] = (16 * t + 4 * g + r == p) ? dr * a.inv_B : 0.0f;
      ecb_mlp_bwd<T0, HT, NTL, true, PAD>(th, th, G, tab.c[1], ws, row, tile, o, dx, j, g);
     }
    } else {
      const bool is_h = role == 2;
      const BnnNet &N = a.net[is_h ? BNN_H : BNN_F];
      const int c0 = is_h ? 2 : 4;
      const bool two = !(is_h && a.binary);
      f32x4 o1[1], o2[1], d[1], dx1[1], dx2[1];
      ecb_head_fwd<T1, T2, T3>(th, N, tab.c[c0], ws, row, is_h ? hin : fin, o1, j, g);
      if (two) ecb_head_fwd<T1, T2, T3>(th, N, tab.c[c0 + 1], ws, row, is_h ? hin : fin, o2, j, g);
      const int wo = N.dims[N.n_layers];
      const float l = __shfl(o1[0][0], j), tgt = is_h ? xv : yv;
      float d0, dl = 0.0f;
      if (!two) {
        lsum = fmaxf(l, 0.0f) - l * tgt + log1pf(expf(-fabsf(l)));
        d0 = (sigmoid_f(l) - tgt) * a.inv_B;
      } else {
        const float raw = ecb_pick<1>(o2, wo - 1, g), r_ = tgt - l;
        float lb, s2;
        const float dr = ecb_gauss(r_ * r_, raw, 1.0f, lb, s2, a.sig2[is_h ? 1 : 2]);
        lsum = lb;
        d0 = -r_ / s2 * a.inv_B;
        dl = dr * a.inv_B;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) d[0][r] = (4 * g + r == 0) ? d0 : 0.0f;
      ecb_head_bwd<T1, T2, T3>(th, th, N, tab.c[c0], ws, row, tile, d, dx1, j, g);
      dx[0] = dx1[0];
      if (two) {
#pragma unroll
        for (int r = 0; r < 4; ++r) d[0][r] = (4 * g + r == wo - 1) ? dl : 0.0f;
        ecb_head_bwd<T1, T2, T3>(th, th, N, tab.c[c0 + 1], ws, row, tile, d, dx2, j, g);
        dx[0] += dx2[0];
      }
    }
#pragma unroll
    for (int t = 0; t < T0; ++t) *reinterpret_cast<f32x4 *>(dzc + (role * B + lrow) * ZW + 16 * t + 4 * g) = dx[t];
    if (role == 0) {            // the prior term of the row
      float zz = 0.0f;
#pragma unroll
      for (int t = 0; t < T0; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) zz = fmaf(zin[t][r], zin[t][r], zz);
      lsum += 0.5f * sum_over_g(zz);
    }
  }
  {
    const float s0 = sum_over_j_to_lane15(lsum);
    if (j == 15 && g == 0) part[wave] = s0;
  }
  __syncthreads();
  const int rb = B * (int)blockIdx.x;
  for (int i0 = tid; i0 < B * q; i0 += ECH_THREADS) {
    const int b = i0 / q, col = i0 - b * q, i = (rb + b) * q + col;
    float v = a.data_z[(long long)a.idx[rb + b] * q + col] * a.inv_B + dzc[(0 * B + b) * ZW + col] + dzc[(1 * B + b) * ZW + col];
    if (col < z0 + z1) v += dzc[(3 * B + b) * ZW + col];                                  // f: (z0, z1, x)
    if (col < z0) v += dzc[(2 * B + b) * ZW + col];                                       // h: (z0, z2)
    else if (col >= z0 + z1 && col < z0 + z1 + z2) v += dzc[(2 * B + b) * ZW + col - z1];
    a.dz[i] = v;
    if (zr.zm) {
      const long long t = (long long)a.idx[rb + b] * q + col;
      bnn_z_row_adam(zr.data_z, zr.zm, zr.zv, t, v, zr.lr_t, zr.b1, zr.b2, zr.eps);
      if (zr.t_last && col == 0) zr.t_last[a.idx[rb + b]] = zr.t_now;
    }
  }
  if (tid == 0 && a.out) {
    float t = 0.0f;
    for (int w = 0; w < 8; ++w) if ((w & 1) < NB) t += part[w];
    if (gridDim.x == 1) a.out[0] = t * a.inv_B;
    else atomicAdd(a.out, t * a.inv_B);             // (zeroed by the noise launch)
  }
}


// ---------------------------------------------------------------------------------------------
// host: workspace layout of a list of calls + the gradient-tile table (shared by the EGM and the iterative-update steps)
// ---------------------------------------------------------------------------------------------
#include <vector>
// Returns the workspace floats used.  tile_nets: the nets whose parameters the step trains (tiles are emitted for them).
inline size_t ecb_build_tab(const BnnNet *nets, const int *call_net, const int *call_soff, int n_calls, int B, int ntl, EcbTab &tab,
                            std::vector<int> &tiles, const int *tile_nets, int n_tile_nets) {
  auto tl = [](int n) { return (n + 15) / 16; };
  size_t off = 0;
  auto take = [&](size_t n) { const size_t r = off; off += (n + 3) / 4 * 4; return (int)r; };
  const int NBt = (B + 15) / 16;
  for (int k = 0; k < 4; ++k) { tab.net_ncalls[k] = 0; tab.kt0[k] = k == BNN_E ? ntl : (k == BNN_G ? tl(nets[BNN_G].dims[0]) : 1); }
  int xw[ECB_CALLS][BNN_MAX_LAYERS], dw[ECB_CALLS][BNN_MAX_LAYERS];
  for (int c = 0; c < n_calls; ++c) {
    const BnnNet &m = nets[call_net[c]];
    EcbCall &C = tab.c[c];
    C.net = call_net[c];
    C.soff = call_soff[c];
    tab.net_calls[C.net][tab.net_ncalls[C.net]++] = c;
    const size_t E_ = (size_t)m.eoff[m.n_layers];
    C.eps = 0; C.dWT = 0;
    C.dW = take(E_ + 256);          // K-contiguous tile loads of padded shapes read up to ~200 floats past the last layer: zeros, never written
    C.sg = take((size_t)B * m.swords);
    const int kt0 = tab.kt0[C.net];
    C.xh = take((size_t)B * 16 * kt0);
    C.bnp = take((size_t)NBt * 2 * 16 * kt0);
    for (int l = 0; l < m.n_layers; ++l) {
      xw[c][l] = 16 * tl(m.dims[l]); dw[c][l] = 16 * tl(m.dims[l + 1]);
      if (C.net == BNN_E && l == 0) xw[c][l] = 16 * ntl;
      if (C.net == BNN_G && l == m.n_layers - 1) dw[c][l] = 16 * ntl;
      C.x[l] = take((size_t)B * xw[c][l]); C.xs[l] = take((size_t)B * xw[c][l]);
      C.d[l] = take((size_t)B * dw[c][l]); C.ds[l] = take((size_t)B * dw[c][l]);
    }
  }
  tiles.clear();
  for (int kk = 0; kk < n_tile_nets; ++kk) {
    const int k = tile_nets[kk];
    const BnnNet &m = nets[k];
    if (tab.net_ncalls[k] == 0) continue;
    for (int l = 0; l < m.n_layers; ++l) {
      const int ni = m.dims[l], no = m.dims[l + 1], c0 = tab.net_calls[k][0];
      for (int u = 0; u < tl(ni); ++u)
        for (int v = 0; v < tl(no); ++v) {
          int en[ECB_TILE_INTS] = {m.woff[l], ni, no, u, v, xw[c0][l], dw[c0][l], tab.net_ncalls[k]};
          for (int c = 0; c < tab.net_ncalls[k]; ++c) {
            const EcbCall &C = tab.c[tab.net_calls[k][c]];
            en[8 + 5 * c] = C.x[l]; en[9 + 5 * c] = C.xs[l]; en[10 + 5 * c] = C.d[l]; en[11 + 5 * c] = C.ds[l]; en[12 + 5 * c] = C.dW + m.eoff[l];
          }
          en[23] = k;
          tiles.insert(tiles.end(), en, en + ECB_TILE_INTS);
        }
    }
  }
  tab.n_tiles = (int)(tiles.size() / ECB_TILE_INTS);
  return off + 64;
}
// shapes the row-tile chains are compiled for
inline bool ecb_shapes_ok(const BnnNet *nets, int q, int p, bool need_e) {
  bool ok = q <= 32;               // one latent input tile, or two (B = 32, 13-tile kernels)
  const int ntl = (p + 1 + 15) / 16;
  ok = ok && ntl <= 13;            // 13 and 7 are compiled exactly; anything narrower runs the 13-tile kernels with masked columns
  for (int k = 0; k < 4; ++k) {
    if (k == BNN_E && !need_e) continue;
    ok = ok && nets[k].bn_fixed == 1 && !nets[k].heads && !nets[k].mv;
  }
  const BnnNet &G = nets[BNN_G], &E = nets[BNN_E];
  ok = ok && G.n_layers >= 3 && G.dims[0] == q && G.dims[G.n_layers] == p + 1;
  for (int l = 1; l < G.n_layers; ++l) ok = ok && G.dims[l] == 64;
  if (need_e) {
    ok = ok && E.n_layers >= 3 && E.dims[E.n_layers] == q && E.dims[0] == p;
    for (int l = 1; l < E.n_layers; ++l) ok = ok && E.dims[l] == 64;
  }
  for (int k : {BNN_F, BNN_H}) {
    const BnnNet &m = nets[k];
    ok = ok && m.n_layers == 4 && m.dims[0] <= 16 && m.dims[1] == 64 && m.dims[2] == 32 && m.dims[3] >= 1 && m.dims[3] <= 16 && m.dims[4] >= 1 &&
         m.dims[4] <= 16;
  }
  return ok;
}
