// egm_chain_gen.h -- train_gen_step of the EGM warm start as register-chained row tiles (CausalBGM base.py:332-377).
//
// Same idea as egm_chain.h: the generator loss splits into two chains that share parameters but no activations,
//   chain A:  z -> g -> v_ -> e -> z__                      (latent reconstruction, the variance-head regulariser of g)
//   chain B:  v -> e -> z_ -> { g -> v__ ,  D ,  f ,  h }   (data reconstruction, adversarial term, outcome / treatment heads)
// and every row of the minibatch is independent until the parameter gradients are summed.  Waves 0,1 walk chain A on the two
// 16-row tiles, waves 2,3 chain B; the activations stay in registers, the weights are read in place from the canonical arrays
// (forward) and from a transposed mirror kept by the Adam step (backward), always two K tiles ahead of the MFMAs that consume
// them.  Layer inputs X_l and pre-activation gradients D_l go to an HBM/L2 workspace; a second launch over ~35 workgroups
// (egm_gen_dw_kernel: one 16x16 tile of one layer per wave, from a host-built table) turns them into parameter gradients
// (X^T D, rows = K) and applies Adam.  (Inside the single chain workgroup the same phase took as long as the chains: 270 tiles x
// ~70 vector-memory instructions through one CU's address unit.)
// Formulas: egm_gen_step_kernel (egm_kernels.h) / oracle/egm.py gen_step_grads; discriminator in inference-mode normalisation.
#pragma once
#include "egm_chain.h"

#define ECG_PASS_G1 0
#define ECG_PASS_E2 1
#define ECG_PASS_E1 2
#define ECG_PASS_G2 3
#define ECG_PASS_F 4
#define ECG_PASS_H 5
#define ECG_TILE_INTS 16

// -D FITC_CLOCK: cycle stamps of the first wave of fit_chain_kernel along the g chain (dev tool: scripts/probe_fit.py prints them)
#ifdef FITC_CLOCK
__device__ unsigned long long g_fitc[64];
#define FITC_T(k) { if (threadIdx.x == 0) g_fitc[k] = clock64(); }
#else
#define FITC_T(k)
#endif

struct EcgTab {
  int x[6][EGM_MAX_LAYERS], d[6][EGM_MAX_LAYERS];   // workspace offsets (floats) of X_l [B x 16 KT_l] and D_l [B x 16 NT_l] per pass
  const int *tiles;                                 // [n_tiles][ECG_TILE_INTS]: xo0, xo1, do0, do1, xw, dw, u, v, woff, n_in, n_out, boff
  int n_tiles;
  float *thetaT;                                    // W_l^T [n_out x n_in] at the offset of W_l
  int n_warm;                                       // floats of theta_g / thetaT pulled into L2 by the idle waves
};

// ---------------------------------------------------------------------------------------------
// pipelined sub-layer: out (+)= in W[:, col0 : col0 + 16 NT], A fragments ECG_DEPTH K tiles ahead
// ---------------------------------------------------------------------------------------------
struct EcgW { const float *W; int ld, n_in, n_out, col0; };
#ifndef ECG_DEPTH
#define ECG_DEPTH 2       // K tiles requested ahead of the one being multiplied (ring of ECG_DEPTH + 1 buffers); 3 measured +2 % on the fit
                          // chains, nothing on the EGM chains, for 32 more registers per fragment set -- the loads are not what the chains wait for
#endif
template <int NT> struct EcgA { float v[ECG_DEPTH + 1][4][NT]; };      // v[0 .. DEPTH-1]: the first K tiles of the sub-layer about to run

// CX: the columns col0 .. col0 + 16 NT all exist (no column clamp / mask: one lane base + immediate offsets);  clamp_rows: the K tile
// may reach beyond n_in (only the last K tile of a layer can).
// KC ("K-contiguous"): the matrix is given in the layout in which the contraction index is the fast one, element (k, col) at
// W[col * ld + k] -- the canonical [in x out] array seen from a BACKWARD product (k = output feature), or a transposed mirror seen
// from a forward one.  A lane's four K values of a tile are then one 16-byte load.  That removes the need for a transposed copy (the
// Bayesian chains read their per-call perturbations this way: nothing to transpose, nothing to scatter), but it is NOT faster per
// se: the address unit pays per cache line touched (16 lines per request here, 4 for the row-contiguous 4-byte form), and the
// deterministic generator step measured 73 -> 88 us with every product switched to it -- so those chains keep the mirror.
// K values beyond n_in read neighbouring finite numbers that meet zero B operands.
template <int NT, bool CX, bool KC = false>
__device__ __forceinline__ void ecg_load_tile(const EcgW &w, int t, float (&av)[4][NT], int j, int g, bool clamp_rows) {
  if constexpr (KC) {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int o = w.col0 + 16 * u + j;
      const int oc = CX ? o : min(o, w.n_out - 1);
      const float *q = w.W + (oc * w.ld + 16 * t + 4 * g);
      float x0 = q[0], x1 = q[1], x2 = q[2], x3 = q[3];      // contiguous: the compiler merges them into one dwordx4 request
      if (!CX && o >= w.n_out) { x0 = 0.0f; x1 = 0.0f; x2 = 0.0f; x3 = 0.0f; }
      av[0][u] = x0; av[1][u] = x1; av[2][u] = x2; av[3][u] = x3;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = 16 * t + 4 * g + r;
      if (clamp_rows) row = min(row, w.n_in - 1);
      const float *wr = w.W + (row * w.ld + w.col0 + j);
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        if (CX) av[r][u] = wr[16 * u];
        else {
          const int o = w.col0 + 16 * u + j;
          const float x = w.W[row * w.ld + min(o, w.n_out - 1)];
          av[r][u] = o < w.n_out ? x : 0.0f;      // padded output columns stay exactly zero (they are the next layer's padded inputs)
        }
      }
    }
  }
}
template <int NT, bool CX, bool KC = false>
__device__ __forceinline__ void ecg_prime(const EcgW &w, EcgA<NT> &A, int j, int g) {
#pragma unroll
  for (int k = 0; k < ECG_DEPTH; ++k) ecg_load_tile<NT, CX, KC>(w, k, A.v[k], j, g, true);       // (a short sub-layer reads clamped duplicates)
}
// Runs the sub-layer whose first two K tiles are in A and leaves the first two K tiles of the next sub-layer `wn` in An.
template <int KT, int NT, int NTN, bool CX, bool CXN, bool KC = false, bool KCN = false>
__device__ __forceinline__ void ecg_sub(const EcgW &w, const f32x4 (&in)[KT], f32x4 (&out)[NT], EcgA<NT> &A, const EcgW &wn, EcgA<NTN> &An,
                                        int j, int g) {
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    if (t + ECG_DEPTH < KT) ecg_load_tile<NT, CX, KC>(w, t + ECG_DEPTH, A.v[(t + ECG_DEPTH) % (ECG_DEPTH + 1)], j, g, KT > 4 || t + ECG_DEPTH == KT - 1);
#pragma unroll
    for (int k = 0; k < ECG_DEPTH; ++k)      // tile k of the next sub-layer takes the request slot this one no longer needs
      if (t == (KT - ECG_DEPTH + k > 0 ? KT - ECG_DEPTH + k : 0)) ecg_load_tile<NTN, CXN, KCN>(wn, k, An.v[k], j, g, true);
    BGM_NO_HOIST();          // pins the issue order (the scheduler would sink every load to just above its MFMA)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < NT; ++u) out[u] = BGM_MFMA(A.v[t % (ECG_DEPTH + 1)][r][u], in[t][r], out[u]);
    BGM_NO_HOIST();
  }
}
template <int NT>
__device__ __forceinline__ void ecg_copy(EcgA<NT> &A, const EcgA<NT> &An) {
#pragma unroll
  for (int b = 0; b < ECG_DEPTH; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < NT; ++u) A.v[b][r][u] = An.v[b][r][u];
}
template <int NT>
__device__ __forceinline__ void ecg_bias(const float *b, int n, int col0, int g, f32x4 (&x)[NT]) {
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) x[u][r] += ech_ld(b, col0 + 16 * u + 4 * g + r, n);
}
template <int NT>
__device__ __forceinline__ void ecg_lrelu(f32x4 (&x)[NT]) {
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) x[u][r] = fmaxf(x[u][r], EGM_LEAK * x[u][r]);
}
template <int NT>
__device__ __forceinline__ void ecg_mask(f32x4 (&d)[NT], const f32x4 (&x)[NT]) {     // d *= lrelu'(pre-activation), from the sign of the output x
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) d[u][r] *= x[u][r] > 0.0f ? 1.0f : EGM_LEAK;
}
template <int NT>
__device__ __forceinline__ void ecg_put(float *base, int row, int g, const f32x4 (&x)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4 *>(base + (long long)row * (16 * NT) + 16 * t + 4 * g) = x[t];
}
template <int NT>
__device__ __forceinline__ void ecg_get(const float *base, int row, int g, f32x4 (&x)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) x[t] = *reinterpret_cast<const f32x4 *>(base + (long long)row * (16 * NT) + 16 * t + 4 * g);
}

// Wide layers (more than four output tiles) run as column groups of four tiles, each a full K sweep over the same input.
// NT = 4 Q + R column tiles, R in {0, 1, 2, 3}.
// PAD: n_out may be anywhere in (0, 16 NT] (a compiled extent used for a narrower layer): every group masks its columns.
template <int KT, int NT, bool KC = false, bool PAD = false>
__device__ __forceinline__ void ecg_wide(const float *W, int ld, int n_in, int n_out, const f32x4 (&in)[KT], f32x4 (&out)[NT], int j, int g) {
  constexpr int Q = NT / 4, R = NT % 4;
  ech_zero<NT>(out);
  EcgW w{W, ld, n_in, n_out, 0};
  if constexpr (Q > 0) {
    // 16 (NT - 1) < n_out <= 16 NT: every group but the very last is complete
    EcgA<4> A, An;
    ecg_prime<4, !PAD && (Q > 1 || R > 0), KC>(w, A, j, g);
#pragma unroll
    for (int c = 0; c < Q; ++c) {
      f32x4 o4[4];
      ech_zero<4>(o4);
      EcgW wn{W, ld, n_in, n_out, 64 * (c + 1 < Q ? c + 1 : c)};
      w.col0 = 64 * c;
      constexpr bool last_full = R > 0 && !PAD;      // the last group of four is complete when a remainder group follows
      if (PAD) ecg_sub<KT, 4, 4, false, false, KC, KC>(w, in, o4, A, wn, An, j, g);
      else if (c + 1 < Q) {
        if (c + 2 < Q || last_full) ecg_sub<KT, 4, 4, true, true, KC, KC>(w, in, o4, A, wn, An, j, g);
        else ecg_sub<KT, 4, 4, true, false, KC, KC>(w, in, o4, A, wn, An, j, g);
      } else {
        if (last_full) ecg_sub<KT, 4, 4, true, true, KC, KC>(w, in, o4, A, wn, An, j, g);
        else ecg_sub<KT, 4, 4, false, false, KC, KC>(w, in, o4, A, wn, An, j, g);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) out[4 * c + u] = o4[u];
      ecg_copy<4>(A, An);
    }
  }
  if constexpr (R > 0) {
    EcgA<R> Ar, Ad;
    EcgW wr{W, ld, n_in, n_out, 64 * Q};
    ecg_prime<R, false, KC>(wr, Ar, j, g);
    f32x4 orr[R];
    ech_zero<R>(orr);
    ecg_sub<KT, R, R, false, false, KC, KC>(wr, in, orr, Ar, wr, Ad, j, g);
#pragma unroll
    for (int u = 0; u < R; ++u) out[4 * Q + u] = orr[u];
  }
}

// ---------------------------------------------------------------------------------------------
// LeakyReLU MLPs with hidden width 16 HT (g, e): forward with stash, backward with stash
// ---------------------------------------------------------------------------------------------
// hidden layers 1 .. L-2 (all [16 HT x 16 HT]); h: output of layer 0 on entry, of layer L-2 on exit
template <int HT>
__device__ __forceinline__ void ecg_hidden_fwd(const float *theta, const EgmMlp &n, const int *xo, float *ws, int row, f32x4 (&h)[HT], int j, int g) {
  constexpr int H = 16 * HT;
  const int L = n.n_layers;
  if (L <= 2) return;
  EcgA<HT> A, An;
  EcgW w{theta + n.woff[1], H, H, H, 0};
  ecg_prime<HT, true>(w, A, j, g);
  for (int l = 1; l < L - 1; ++l) {
    BGM_NO_HOIST();
    ecg_put<HT>(ws + xo[l], row, g, h);
    w.W = theta + n.woff[l];
    EcgW wn{theta + n.woff[min(l + 1, L - 2)], H, H, H, 0};
    f32x4 h2[HT];
    ech_zero<HT>(h2);
    ecg_sub<HT, HT, HT, true, true>(w, h, h2, A, wn, An, j, g);
    FITC_T(30 + 2 * l);
    ecg_bias<HT>(w.W + H * H, H, 0, g, h2);
    ecg_lrelu<HT>(h2);
#pragma unroll
    for (int u = 0; u < HT; ++u) h[u] = h2[u];
    ecg_copy<HT>(A, An);
    FITC_T(31 + 2 * l);
  }
}
// dh: gradient with respect to the OUTPUT of layer L-2 on entry (already masked: D_{L-2}); on exit D_0.  Stashes D_{L-2} .. D_1 (not D_0).
template <int HT>
__device__ __forceinline__ void ecg_hidden_bwd(const float *thetaT, const EgmMlp &n, const int *xo, const int *dofs, float *ws, int row,
                                               f32x4 (&dh)[HT], int j, int g) {
  constexpr int H = 16 * HT;
  const int L = n.n_layers;
  if (L <= 2) return;
  EcgA<HT> A, An;
  EcgW w{thetaT + n.woff[L - 2], H, H, H, 0};
  ecg_prime<HT, true>(w, A, j, g);
  for (int l = L - 2; l >= 1; --l) {
    BGM_NO_HOIST();
    ecg_put<HT>(ws + dofs[l], row, g, dh);
    w.W = thetaT + n.woff[l];
    EcgW wn{thetaT + n.woff[max(l - 1, 1)], H, H, H, 0};
    f32x4 d2[HT], xl[HT];
    ech_zero<HT>(d2);
    ecg_get<HT>(ws + xo[l], row, g, xl);          // x_l = LeakyReLU output of layer l-1
    ecg_sub<HT, HT, HT, true, true>(w, dh, d2, A, wn, An, j, g);
    FITC_T(40 + 2 * l);
    ecg_mask<HT>(d2, xl);
#pragma unroll
    for (int u = 0; u < HT; ++u) dh[u] = d2[u];
    ecg_copy<HT>(A, An);
    FITC_T(41 + 2 * l);
  }
}

// generator g: [q <= 16] -> 16 HT x (L-2) -> [n_out = p + 1], NTL output tiles
template <int HT, int NTL, bool PAD = false, int T0 = 1>
__device__ __forceinline__ void ecg_g_fwd(const float *theta, const EgmMlp &n, const int *xo, float *ws, int row, const f32x4 (&zin)[T0],
                                          f32x4 (&out)[NTL], int j, int g) {
  constexpr int H = 16 * HT;
  const int L = n.n_layers, q = n.dims[0], no = n.dims[L];
  ecg_put<T0>(ws + xo[0], row, g, zin);
  f32x4 h[HT];
  {
    EcgA<HT> A, Ad;
    EcgW w{theta + n.woff[0], H, q, H, 0};
    ecg_prime<HT, true>(w, A, j, g);
    ech_zero<HT>(h);
    ecg_sub<T0, HT, HT, true, true>(w, zin, h, A, w, Ad, j, g);
    ecg_bias<HT>(w.W + q * H, H, 0, g, h);
    ecg_lrelu<HT>(h);
  }
  FITC_T(2);
  ecg_hidden_fwd<HT>(theta, n, xo, ws, row, h, j, g);
  FITC_T(3);
  ecg_put<HT>(ws + xo[L - 1], row, g, h);
  const float *Wl = theta + n.woff[L - 1];
  ecg_wide<HT, NTL, false, PAD>(Wl, no, H, no, h, out, j, g);
  FITC_T(4);
  ecg_bias<NTL>(Wl + H * no, no, 0, g, out);
}
// dout: dLoss/d output (NTL tiles, zero beyond n_out).  dx (WANT_DX): dLoss/d input.
template <int HT, int NTL, bool WANT_DX, int T0 = 1>
__device__ __forceinline__ void ecg_g_bwd(const float *thetaT, const EgmMlp &n, const int *xo, const int *dofs, float *ws, int row,
                                          const f32x4 (&dout)[NTL], f32x4 (&dx)[T0], int j, int g) {
  constexpr int H = 16 * HT;
  const int L = n.n_layers, q = n.dims[0], no = n.dims[L];
  ecg_put<NTL>(ws + dofs[L - 1], row, g, dout);
  f32x4 dh[HT], xl[HT];
  {
    EcgA<HT> A, Ad;
    EcgW w{thetaT + n.woff[L - 1], H, no, H, 0};      // W^T [n_out x H]
    ecg_prime<HT, true>(w, A, j, g);
    ech_zero<HT>(dh);
    ecg_get<HT>(ws + xo[L - 1], row, g, xl);
    ecg_sub<NTL, HT, HT, true, true>(w, dout, dh, A, w, Ad, j, g);
    ecg_mask<HT>(dh, xl);
  }
  FITC_T(7);
  ecg_hidden_bwd<HT>(thetaT, n, xo, dofs, ws, row, dh, j, g);
  FITC_T(8);
  ecg_put<HT>(ws + dofs[0], row, g, dh);
  if (WANT_DX) {
    EcgA<T0> A, Ad;
    EcgW w{thetaT + n.woff[0], q, H, q, 0};           // W^T [H x q]
    ecg_prime<T0, false>(w, A, j, g);
    ech_zero<T0>(dx);
    ecg_sub<HT, T0, T0, false, false>(w, dh, dx, A, w, Ad, j, g);
  }
}

// encoder e: [p] -> 16 HT x (L-2) -> [q <= 16];  xin: the NTL input tiles (zero beyond p), already in the stash as X_0
template <int HT, int NTL, int T0 = 1>
__device__ __forceinline__ void ecg_e_fwd(const float *theta, const EgmMlp &n, const int *xo, float *ws, int row, const f32x4 (&xin)[NTL],
                                          f32x4 (&z)[T0], int j, int g) {
  constexpr int H = 16 * HT;
  const int L = n.n_layers, p = n.dims[0], q = n.dims[L];
  f32x4 h[HT];
  {
    EcgA<HT> A, Ad;
    EcgW w{theta + n.woff[0], H, p, H, 0};
    ecg_prime<HT, true>(w, A, j, g);
    ech_zero<HT>(h);
    ecg_sub<NTL, HT, HT, true, true>(w, xin, h, A, w, Ad, j, g);
    ecg_bias<HT>(w.W + p * H, H, 0, g, h);
    ecg_lrelu<HT>(h);
  }
  ecg_hidden_fwd<HT>(theta, n, xo, ws, row, h, j, g);
  ecg_put<HT>(ws + xo[L - 1], row, g, h);
  {
    EcgA<T0> A, Ad;
    EcgW w{theta + n.woff[L - 1], q, H, q, 0};
    ecg_prime<T0, false>(w, A, j, g);
    ech_zero<T0>(z);
    ecg_sub<HT, T0, T0, false, false>(w, h, z, A, w, Ad, j, g);
    ecg_bias<T0>(w.W + H * q, q, 0, g, z);
  }
}
template <int HT, int NTL, bool WANT_DX, bool PAD = false, int T0 = 1>
__device__ __forceinline__ void ecg_e_bwd(const float *thetaT, const EgmMlp &n, const int *xo, const int *dofs, float *ws, int row,
                                          const f32x4 (&dz)[T0], f32x4 (&dx)[NTL], int j, int g) {
  constexpr int H = 16 * HT;
  const int L = n.n_layers, p = n.dims[0], q = n.dims[L];
  ecg_put<T0>(ws + dofs[L - 1], row, g, dz);
  f32x4 dh[HT], xl[HT];
  {
    EcgA<HT> A, Ad;
    EcgW w{thetaT + n.woff[L - 1], H, q, H, 0};       // W^T [q x H]
    ecg_prime<HT, true>(w, A, j, g);
    ech_zero<HT>(dh);
    ecg_get<HT>(ws + xo[L - 1], row, g, xl);
    ecg_sub<T0, HT, HT, true, true>(w, dz, dh, A, w, Ad, j, g);
    ecg_mask<HT>(dh, xl);
  }
  ecg_hidden_bwd<HT>(thetaT, n, xo, dofs, ws, row, dh, j, g);
  ecg_put<HT>(ws + dofs[0], row, g, dh);
  if (WANT_DX) ecg_wide<HT, NTL, false, PAD>(thetaT + n.woff[0], p, H, p, dh, dx, j, g);      // W^T [H x p]
}

// head networks f, h: [in <= 16] -> 16 T1 -> 16 T2 -> 16 T3 -> [out <= 16], LeakyReLU
template <int T1, int T2, int T3>
__device__ __forceinline__ void ecg_head_fwd(const float *theta, const EgmMlp &n, const int *xo, float *ws, int row, const f32x4 (&in)[1],
                                             f32x4 (&out)[1], int j, int g) {
  const int d0 = n.dims[0], d1 = n.dims[1], d2 = n.dims[2], d3 = n.dims[3], d4 = n.dims[4];
  ecg_put<1>(ws + xo[0], row, g, in);
  f32x4 a1[T1], a2[T2], a3[T3];
  EcgA<T1> A1; EcgA<T2> A2; EcgA<T3> A3; EcgA<1> A4, Ad;
  EcgW w0{theta + n.woff[0], d1, d0, d1, 0}, w1{theta + n.woff[1], d2, d1, d2, 0}, w2{theta + n.woff[2], d3, d2, d3, 0},
       w3{theta + n.woff[3], d4, d3, d4, 0};
  ecg_prime<T1, true>(w0, A1, j, g);
  ech_zero<T1>(a1);
  ecg_sub<1, T1, T2, true, true>(w0, in, a1, A1, w1, A2, j, g);
  ecg_bias<T1>(w0.W + d0 * d1, d1, 0, g, a1);
  ecg_lrelu<T1>(a1);
  ecg_put<T1>(ws + xo[1], row, g, a1);
  ech_zero<T2>(a2);
  ecg_sub<T1, T2, T3, true, false>(w1, a1, a2, A2, w2, A3, j, g);
  ecg_bias<T2>(w1.W + d1 * d2, d2, 0, g, a2);
  ecg_lrelu<T2>(a2);
  ecg_put<T2>(ws + xo[2], row, g, a2);
  ech_zero<T3>(a3);
  ecg_sub<T2, T3, 1, false, false>(w2, a2, a3, A3, w3, A4, j, g);
  ecg_bias<T3>(w2.W + d2 * d3, d3, 0, g, a3);
  ecg_lrelu<T3>(a3);
  ecg_put<T3>(ws + xo[3], row, g, a3);
  ech_zero<1>(out);
  ecg_sub<T3, 1, 1, false, false>(w3, a3, out, A4, w3, Ad, j, g);
  ecg_bias<1>(w3.W + d3 * d4, d4, 0, g, out);
}
template <int T1, int T2, int T3>
__device__ __forceinline__ void ecg_head_bwd(const float *thetaT, const EgmMlp &n, const int *xo, const int *dofs, float *ws, int row,
                                             const f32x4 (&dout)[1], f32x4 (&dx)[1], int j, int g) {
  const int d0 = n.dims[0], d1 = n.dims[1], d2 = n.dims[2], d3 = n.dims[3], d4 = n.dims[4];
  ecg_put<1>(ws + dofs[3], row, g, dout);
  EcgA<T3> A3; EcgA<T2> A2; EcgA<T1> A1; EcgA<1> A0, Ad;
  EcgW w3{thetaT + n.woff[3], d3, d4, d3, 0}, w2{thetaT + n.woff[2], d2, d3, d2, 0}, w1{thetaT + n.woff[1], d1, d2, d1, 0},
       w0{thetaT + n.woff[0], d0, d1, d0, 0};
  f32x4 e3[T3], e2[T2], e1[T1], x3[T3], x2[T2], x1[T1];
  ecg_prime<T3, false>(w3, A3, j, g);
  ecg_get<T3>(ws + xo[3], row, g, x3);
  ech_zero<T3>(e3);
  ecg_sub<1, T3, T2, false, true>(w3, dout, e3, A3, w2, A2, j, g);
  ecg_mask<T3>(e3, x3);
  ecg_put<T3>(ws + dofs[2], row, g, e3);
  ecg_get<T2>(ws + xo[2], row, g, x2);
  ech_zero<T2>(e2);
  ecg_sub<T3, T2, T1, true, true>(w2, e3, e2, A2, w1, A1, j, g);
  ecg_mask<T2>(e2, x2);
  ecg_put<T2>(ws + dofs[1], row, g, e2);
  ecg_get<T1>(ws + xo[1], row, g, x1);
  ech_zero<T1>(e1);
  ecg_sub<T2, T1, 1, true, false>(w1, e2, e1, A1, w0, A0, j, g);
  ecg_mask<T1>(e1, x1);
  ecg_put<T1>(ws + dofs[0], row, g, e1);
  ech_zero<1>(dx);
  ecg_sub<T1, 1, 1, false, false>(w0, e1, dx, A0, w0, Ad, j, g);
}

// dLoss/d input of the (fixed) discriminator for dLoss/d out = dout on every row
template <int T1, int T2, int T3, int T0 = 1>
__device__ __forceinline__ void ecg_disc_dx(const float *par, const EchP &P, const EchFwd<T1, T2, T3, T0> &F, float dout, f32x4 (&dx)[T0], int j, int g) {
  const float c = ech_c();
  f32x4 du3[T3], du2[T2], du1[T1], da2[T2], da1[T1];
#pragma unroll
  for (int t = 0; t < T3; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float av = F.a3[t][r];
      du3[t][r] = dout * par[P.wo + 16 * t + 4 * g + r] * (1.0f - av * av) * par[P.ga2 + 16 * t + 4 * g + r] * c;
    }
  ech_dense<T3, T2, true>(par + P.T2, P.lt2, nullptr, P.d3, P.d2, du3, da2, j, g);
#pragma unroll
  for (int t = 0; t < T2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float av = F.a2[t][r]; du2[t][r] = da2[t][r] * (1.0f - av * av) * par[P.ga1 + 16 * t + 4 * g + r] * c; }
  ech_dense<T2, T1, true>(par + P.T1, P.lt1, nullptr, P.d2, P.d1, du2, da1, j, g);
#pragma unroll
  for (int t = 0; t < T1; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float av = F.a1[t][r]; du1[t][r] = da1[t][r] * (1.0f - av * av) * par[P.ga0 + 16 * t + 4 * g + r] * c; }
  ech_dense<T1, T0, true>(par + P.T0, P.lt0, nullptr, P.d1, P.d0, du1, dx, j, g);
}

// LDS words behind the parameter block: per-wave loss partials [8 x 16] | flags [16] | z_ tiles | head-input gradient tiles | head
// contribution to dLoss/dz_  (each [B x 16])
template <int T1, int T2, int T3, int T0 = 1>
__host__ __device__ inline int ecg_lds_floats(const EgmDisc &d, int B) { return 64 + ech_layout<T1, T2, T3, T0>(d).total + 8 * 16 + 16 + 3 * 16 * T0 * B; }

// ---------------------------------------------------------------------------------------------
// the step
// ---------------------------------------------------------------------------------------------
template <int HT, int NTL, int T1, int T2, int T3, int NB, bool PAD = false, int T0 = 1>
static __global__ __launch_bounds__(ECH_THREADS) void egm_gen_chain_kernel(EgmArgs a, EcgTab tab) {
  extern __shared__ __attribute__((aligned(16))) float ech_lds[];
  constexpr int B = 16 * NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int q = a.q, p = a.p, z0 = a.z0, z1 = a.z1, z2 = a.z2;
  constexpr int ZW = 16 * T0;                 // floats per row of the latent tiles in LDS
  const EchP P = ech_layout<T1, T2, T3, T0>(a.dz);
  float *par = ech_lds + 64;
  float *part = par + P.total;                 // [8 waves][16] loss partial sums
  volatile int *flag = reinterpret_cast<volatile int *>(part + 8 * 16);   // [0]: parameter block filled (counts 4 waves); [2 + tile]: z_ of
                                                                          // the tile written; [4 + tile]: head contribution to dz written
  float *zt = part + 8 * 16 + 16;              // z_ of chain B, [B x 16]
  float *dzt = zt + ZW * B;                    // dLoss/d hin, [B x 16] (lane permutation scratch of the head waves)
  float *dzh = dzt + ZW * B;                   // dLoss/dz_ through D, f and h, [B x ZW]
  if (tid < 16) flag[tid] = 0;
  __syncthreads();
  ECH_STAMP(0);
  const int role = wave >> 1, tile = wave & 1;
  const bool active = tile < NB;
  const int row = 16 * tile + j;
  const float invB = 1.0f / (float)B;
  const float zrec = a.use_z_rec ? 1.0f : 0.0f;
  float *ws = a.ws;
  const float *th = a.theta_g, *tT = tab.thetaT;
  float ls[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};     // l_v, l_z, l_x, l_y, s_g, s_f, s_h, adv (sums over this lane's row)
  if (role == 0 && active) {
    // ================= chain A =================
    f32x4 zin[T0];
#pragma unroll
    for (int t = 0; t < T0; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) zin[t][r] = ech_ld(a.z + (long long)row * q, 16 * t + 4 * g + r, q);
    f32x4 gz[NTL];
    ecg_g_fwd<HT, NTL, PAD, T0>(th, a.g, tab.x[ECG_PASS_G1], ws, row, zin, gz, j, g);
    ECH_STAMP(1);
    float sgv = 0.0f;                          // variance head g(z)[:, p] of this row (held by one lane group)
    f32x4 vin[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * t + 4 * g + r;
        vin[t][r] = f < p ? gz[t][r] : 0.0f;
        sgv += f == p ? gz[t][r] : 0.0f;
      }
    sgv = sum_over_g(sgv);
    ls[4] = sgv * sgv;
    ecg_put<NTL>(ws + tab.x[ECG_PASS_E2][0], row, g, vin);
    f32x4 zz[T0];
    ecg_e_fwd<HT, NTL, T0>(th, a.e, tab.x[ECG_PASS_E2], ws, row, vin, zz, j, g);
    ECH_STAMP(2);
    f32x4 dzz[T0];
    float lz = 0.0f;
#pragma unroll
    for (int tz = 0; tz < T0; ++tz)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = zin[tz][r] - zz[tz][r];    // both are zero beyond q
        lz = fmaf(t, t, lz);
        dzz[tz][r] = zrec * (-2.0f / (float)(B * q)) * t;
      }
    ls[1] = sum_over_g(lz);
    f32x4 dv[NTL];
    ecg_e_bwd<HT, NTL, true, PAD, T0>(tT, a.e, tab.x[ECG_PASS_E2], tab.d[ECG_PASS_E2], ws, row, dzz, dv, j, g);
    ECH_STAMP(3);
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * t + 4 * g + r;
        dv[t][r] = f < p ? dv[t][r] : (f == p ? 0.001f * 2.0f * sgv * invB : 0.0f);
      }
    f32x4 dnone[T0];
    ecg_g_bwd<HT, NTL, false, T0>(tT, a.g, tab.x[ECG_PASS_G1], tab.d[ECG_PASS_G1], ws, row, dv, dnone, j, g);
  } else if (role == 1 && active) {
    // ================= chain B =================
    const long long prow = a.idx[row];
    const float xv = a.x[prow], yv = a.y[prow];
    const float *vrow = a.v + prow * p;
    f32x4 vin[NTL];
    if ((p & 3) == 0 && (reinterpret_cast<unsigned long long>(vrow) & 15ull) == 0) {      // one 16-byte request per tile and lane
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const int f = 16 * t + 4 * g;
        const f32x4 x = *reinterpret_cast<const f32x4 *>(vrow + min(f, p - 4));
        vin[t] = f < p ? x : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    } else {
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) vin[t][r] = ech_ld(vrow, 16 * t + 4 * g + r, p);
    }
    ecg_put<NTL>(ws + tab.x[ECG_PASS_E1][0], row, g, vin);
    f32x4 ze[T0];
    ecg_e_fwd<HT, NTL, T0>(th, a.e, tab.x[ECG_PASS_E1], ws, row, vin, ze, j, g);
    ECH_STAMP(1);
#pragma unroll
    for (int t = 0; t < T0; ++t) *reinterpret_cast<f32x4 *>(zt + row * ZW + 16 * t + 4 * g) = ze[t];
    __threadfence_block();
    if (lane == 0) flag[2 + tile] = 1;         // waves 4,5 take D, f and h from here
    f32x4 dz[T0];                              // dLoss/dz_ accumulates here
    {
      f32x4 gv[NTL];
      ecg_g_fwd<HT, NTL, PAD, T0>(th, a.g, tab.x[ECG_PASS_G2], ws, row, ze, gv, j, g);
      float lv = 0.0f;
      ecg_get<NTL>(ws + tab.x[ECG_PASS_E1][0], row, g, vin);
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * g + r;
          const float d = f < p ? vin[t][r] - gv[t][r] : 0.0f;
          lv = fmaf(d, d, lv);
          gv[t][r] = (-2.0f / (float)(B * p)) * d;
        }
      ls[0] = sum_over_g(lv);
      ecg_g_bwd<HT, NTL, true, T0>(tT, a.g, tab.x[ECG_PASS_G2], tab.d[ECG_PASS_G2], ws, row, gv, dz, j, g);
    }
    ECH_STAMP(2);
    // D, f, h ran on waves 4,5 meanwhile: add their part of dLoss/dz_
    while (flag[4 + tile] == 0) __builtin_amdgcn_s_sleep(2);
    __threadfence_block();
#pragma unroll
    for (int t = 0; t < T0; ++t) dz[t] += *reinterpret_cast<const f32x4 *>(dzh + row * ZW + 16 * t + 4 * g);
    ECH_STAMP(3);
    f32x4 dnone[NTL];
    ecg_e_bwd<HT, NTL, false, false, T0>(tT, a.e, tab.x[ECG_PASS_E1], tab.d[ECG_PASS_E1], ws, row, dz, dnone, j, g);
  } else if (role >= 2) {
    // ================= waves 4..7: the discriminator's parameter block; pull the generator-side weights into this XCD's L2 (waves
    // 4,5 the forward arrays, 6,7 the transposed mirror); then waves 4,5 take D, f and h of chain B as soon as its z_ exists ======
    ech_fill_params<T1, T2, T3, T0>(par, P, a.theta_d, a.dz, tid - 256, 256);
    __threadfence_block();
    if (lane == 0) atomicAdd(const_cast<int *>(flag), 1);
    {
      float sink = 0.0f;
      const int n4 = tab.n_warm >> 2;                         // both arrays start 16-byte aligned
      const f32x4 *src = reinterpret_cast<const f32x4 *>(role == 2 ? th : tT);
      for (int i = (tid & 127); i < n4; i += 128 * 16) {
        f32x4 acc4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 16; ++k) acc4 += src[min(i + 128 * k, n4 - 1)];
        sink += acc4[0] + acc4[1] + acc4[2] + acc4[3];
      }
      asm volatile("" ::"v"(sink));
    }
    if (role == 2 && active) {
      const long long prow = a.idx[row];
      const float xv = a.x[prow], yv = a.y[prow];
      while (flag[0] < 4) __builtin_amdgcn_s_sleep(2);
      while (flag[2 + tile] == 0) __builtin_amdgcn_s_sleep(2);
      __threadfence_block();
      f32x4 ze[T0], dz[T0];
#pragma unroll
      for (int t = 0; t < T0; ++t) ze[t] = *reinterpret_cast<const f32x4 *>(zt + row * ZW + 16 * t + 4 * g);
    // ---- adversarial term through the fixed discriminator
    {
      EchFwd<T1, T2, T3, T0> F;
#pragma unroll
      for (int t = 0; t < T0; ++t) F.a0[t] = ze[t];
      ech_disc_fwd<T1, T2, T3, T0>(par, P, F, j, g);
      ls[7] = -F.out;
      ecg_disc_dx<T1, T2, T3, T0>(par, P, F, -invB, dz, j, g);
    }
    // ---- f(z0, z1, x) -> y
    {
      f32x4 fin[1], fo[1], dfo[1], dfin[1];
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int f = 4 * g + r; fin[0][r] = f < z0 + z1 ? ze[0][r] : (f == z0 + z1 ? xv : 0.0f); }
      ecg_head_fwd<T1, T2, T3>(th, a.f, tab.x[ECG_PASS_F], ws, row, fin, fo, j, g);
      const int of = a.f.dims[a.f.n_layers];
      const float mu = __shfl(fo[0][0], j);                         // feature 0 lives in lane group 0
      float sg = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) sg += (4 * g + r == of - 1) ? fo[0][r] : 0.0f;
      sg = sum_over_g(sg);
      ls[3] = (mu - yv) * (mu - yv);
      ls[5] = sg * sg;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 4 * g + r;
        float t = 0.0f;
        if (f == 0) t += 2.0f * (mu - yv) * invB;
        if (f == of - 1) t += 0.001f * 2.0f * sg * invB;
        dfo[0][r] = t;
      }
      ecg_head_bwd<T1, T2, T3>(tT, a.f, tab.x[ECG_PASS_F], tab.d[ECG_PASS_F], ws, row, dfo, dfin, j, g);
#pragma unroll
      for (int r = 0; r < 4; ++r) dz[0][r] += (4 * g + r < z0 + z1) ? dfin[0][r] : 0.0f;
    }
    // ---- h(z0, z2) -> x   (its input gathers z_[0:z0] and z_[z0+z1 : z0+z1+z2]: a lane permutation through LDS)
    {
      f32x4 hin[1], ho[1], dho[1], dhin[1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 4 * g + r;
        const float t = zt[row * ZW + min(f < z0 ? f : f + z1, ZW - 1)];
        hin[0][r] = f < z0 + z2 ? t : 0.0f;
      }
      ecg_head_fwd<T1, T2, T3>(th, a.h, tab.x[ECG_PASS_H], ws, row, hin, ho, j, g);
      const int oh = a.h.dims[a.h.n_layers];
      const float mu = __shfl(ho[0][0], j);
      float sg = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) sg += (4 * g + r == oh - 1) ? ho[0][r] : 0.0f;
      sg = sum_over_g(sg);
      float dmu;
      if (a.binary) {
        ls[2] = fmaxf(mu, 0.0f) - mu * xv + log1pf(expf(-fabsf(mu)));
        dmu = (1.0f / (1.0f + expf(-mu)) - xv) * invB;
      } else {
        ls[2] = (mu - xv) * (mu - xv);
        dmu = 2.0f * (mu - xv) * invB;
      }
      ls[6] = sg * sg;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 4 * g + r;
        float t = 0.0f;
        if (f == 0) t += dmu;
        if (f == oh - 1) t += 0.001f * 2.0f * sg * invB;
        dho[0][r] = t;
      }
      ecg_head_bwd<T1, T2, T3>(tT, a.h, tab.x[ECG_PASS_H], tab.d[ECG_PASS_H], ws, row, dho, dhin, j, g);
      *reinterpret_cast<f32x4 *>(dzt + row * 16 + 4 * g) = dhin[0];
      __threadfence_block();
#pragma unroll
      for (int tz = 0; tz < T0; ++tz)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * tz + 4 * g + r;
          const int src = f < z0 ? f : f - z1;            // z_[f] fed hin[f] (f < z0) or hin[f - z1] (z0 + z1 <= f < z0 + z1 + z2)
          const float t = dzt[row * 16 + min(max(src, 0), 15)];
          dz[tz][r] += (f < z0 || (f >= z0 + z1 && f < z0 + z1 + z2)) ? t : 0.0f;
        }
    }
#pragma unroll
      for (int t = 0; t < T0; ++t) *reinterpret_cast<f32x4 *>(dzh + row * ZW + 16 * t + 4 * g) = dz[t];
      __threadfence_block();
      if (lane == 0) flag[4 + tile] = 1;
    }
  }
  // per-row loss terms -> per-wave sums (lane j = 15 of group 0 holds the totals)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float s = sum_over_j_to_lane15(ls[k]);
    if (j == 15 && g == 0) part[wave * 16 + k] = s;
  }
  ECH_STAMP(4);
  __syncthreads();
  ECH_STAMP(5);
  ECH_STAMP(6);
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

// ---------------------------------------------------------------------------------------------
// parameter gradients + Adam: one 16x16 tile of one layer per wave
// ---------------------------------------------------------------------------------------------
template <int NB>
static __global__ __launch_bounds__(ECH_THREADS) void egm_gen_dw_kernel(EgmArgs a, EcgTab tab) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int tau = blockIdx.x * ECH_WAVES + wave;
  if (tau >= tab.n_tiles) return;
  const int *td = tab.tiles + tau * ECG_TILE_INTS;
  const int xo0 = td[0], xo1 = td[1], do0 = td[2], do1 = td[3], xw = td[4], dw = td[5], u = td[6], v = td[7], woff = td[8],
            n_in = td[9], n_out = td[10], boff = td[11];
  const float *ws = a.ws;
  const int o = 16 * v + j;
  int e[4]; float t0[4], m0[4], v0[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    e[r] = woff + min(16 * u + 4 * g + r, n_in - 1) * n_out + min(o, n_out - 1);
    t0[r] = a.theta_g[e[r]]; m0[r] = a.m_g[e[r]]; v0[r] = a.v_g[e[r]];
  }
  const int eb = max(boff + min(o, n_out - 1), 0);
  const float tb = a.theta_g[eb], mb = a.m_g[eb], vb = a.v_g[eb];
  float xa[2][NB * 4], da[2][NB * 4];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int xo = pass == 0 ? xo0 : max(xo1, 0), dofs = pass == 0 ? do0 : max(do1, 0);
    const float *xp = ws + xo + g * xw + 16 * u + j, *dp = ws + dofs + g * dw + 16 * v + j;
#pragma unroll
    for (int s4 = 0; s4 < NB * 4; ++s4) { xa[pass][s4] = xp[4 * s4 * xw]; da[pass][s4] = dp[4 * s4 * dw]; }
  }
  f32x4 w = {0.0f, 0.0f, 0.0f, 0.0f};
  float bs = 0.0f;
  const float second = xo1 >= 0 ? 1.0f : 0.0f;
#pragma unroll
  for (int s4 = 0; s4 < NB * 4; ++s4) { w = BGM_MFMA(xa[0][s4], da[0][s4], w); bs += da[0][s4]; }
#pragma unroll
  for (int s4 = 0; s4 < NB * 4; ++s4) { w = BGM_MFMA(xa[1][s4] * second, da[1][s4], w); bs += da[1][s4] * second; }
  bs = sum_over_g(bs);
  auto adam = [&](int ei, float gi, float th0, float m_0, float v_0, int et) {
    a.grad_g[ei] = gi;
    if (a.apply) {
      const float mi = a.adam.b1 * m_0 + (1.0f - a.adam.b1) * gi;
      const float vi = a.adam.b2 * v_0 + (1.0f - a.adam.b2) * gi * gi;
      const float tn = th0 - a.adam.lr_t * mi / (sqrtf(vi) + a.adam.eps);
      a.m_g[ei] = mi; a.v_g[ei] = vi; a.theta_g[ei] = tn;
      if (et >= 0) tab.thetaT[et] = tn;
    }
  };
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 16 * u + 4 * g + r;
    if (f < n_in && o < n_out) adam(e[r], w[r], t0[r], m0[r], v0[r], woff + o * n_in + f);
  }
  if (boff >= 0 && g == 0 && o < n_out) adam(boff + o, bs, tb, mb, vb, -1);
}
