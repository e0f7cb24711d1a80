// fit_chain.h -- CausalBGM iterative-update steps at the reference batch size (B = 16 / 32) as register-chained row tiles.
//
// replaces, for that case, the LDS-blob kernels of fit_kernels.h (update_g_net / update_h_net / update_f_net,
// causalbgm/base.py:156-243, and update_latent_variable_sgd :246-302): at two row tiles the 157 KB blob fill of every launch
// (4 per minibatch) costs more than the arithmetic.  Here the weights are read in place (canonical array forward, transposed
// mirror backward), one wave per 16-row tile and network:
//   theta phase   waves 0,1: g forward + backward;  2,3: f;  4,5: h   -> stash of layer inputs / pre-activation gradients
//                 fit_chain_dw_kernel (34 workgroups): gradient tiles X^T D -> grad[] in canonical parameter order
//                 (then the optional all-reduce and fit_adam_theta_kernel, which also refreshes the blobs and the mirror)
//   latent phase  the same chains with input gradients; dz = sum of the three + z / B -> the Adam-on-Z kernels.
// Loss bookkeeping and gradient formulas are those of fit_fwd_kernel / fit_bwd_kernel.
#pragma once
#include "egm_chain_gen.h"
#include "fit_types.h"
#include "z_replay.h"

struct FitChainArgs {
  EgmMlp g, f, h;                 // offsets into theta (canonical [g | f | h])
  const float *theta, *thetaT;
  float *ws;
  int xo[3][EGM_MAX_LAYERS], dofs[3][EGM_MAX_LAYERS];     // stash offsets per net (g, f, h)
  const int *tiles; int n_tiles;  // ECG_TILE_INTS per tile (single pass)
  float *grad;                    // [n_params]
  const float *x, *y, *v, *data_z;
  const int *idx; long long row_lo;
  int q, p, z0, z1, z2, binary;
  float sig2_v, sig2_x, sig2_y;   // > 0: fixed variances
  float inv_B;
  double *loss;                   // [7] accumulators or NULL
  float *dz;                      // latent phase: [B x q]
  int *pos; long long pos_n; int epoch;      // latent phase, dense Adam on Z: batch position / step stamp of the minibatch rows (fit_set_pos_kernel's job)
  float *zm, *zv; float *z_out; int *t_last; int t_now; float lr_t, b1, b2, eps;   // latent phase, zm != NULL: the Adam step on the
                                  // minibatch rows is applied here (lazy = 1 / 2; t_last != NULL marks the rows current to step t_now)
  int n_warm;                     // floats of theta (= of the transposed mirror) the idle wave pair pulls into this XCD's L2
  int n_valid;                    // rows of the minibatch (<= 16 NB): the tile rows behind them are masked -- they read row n_valid - 1,
                                  // contribute zero loss and zero output gradients, hence nothing to any parameter gradient
  FitSync sy;                     // bgm_causal_fit_epoch: the chain kernel waits at entry (both phases); the latent phase's workgroups and
                                  // the gradient-tile kernel's count themselves done at their end
  int rp_first, rp_n, rp_t_to;    // latent phase, rp_n > 0: workgroups blockIdx.x >= rp_first replay the pending zero-gradient steps of
  const int *rp_idx; const int *rp_tlast; float rp_lr;      // the rows rp_idx[0 .. rp_n) to step rp_t_to (fit_adam_z_replay_kernel's job, riding along)
  FitAdamTheta ad;                // theta phase, ad.on: the gradient-tile kernel applies the Adam step itself (bgm_causal_fit_epoch)
};

__device__ __forceinline__ long long fitc_row(const FitChainArgs &a, int b) {
  b = min(b, a.n_valid - 1);
  return a.idx ? (long long)a.idx[b] : a.row_lo + b;
}

// Z_MODE 0: theta phase (stash, no input gradients); 1: latent phase (input gradients -> dz)
// T0: latent input tiles of g (q <= 16 T0); the head networks' inputs (z0 + z1 + 1, z0 + z2) stay within one tile
// WS (one row tile per workgroup, HT = 4, T0 = 1): g's last layer -- 13 of the chain's ~60 output tiles but 40 % of its time, a
// straight line of 2 x 208 dependent MFMAs -- is spread over the workgroup's idle waves.  The chain wave publishes the last hidden
// activation in LDS; it and the waves 1, 3, 5 each take one column group of four output tiles: forward product, their part of the
// row's residual sum (met in LDS), the output gradients of their columns (stashed for the gradient-tile kernel) and their share
// of the backward product into the hidden layer, summed by the chain wave, which walks on.  Three LDS flags order the phases.
template <int HT, int NTL, int T1, int T2, int T3, int NB, int Z_MODE, bool PAD = false, int T0 = 1, bool WS = false>
static __global__ __launch_bounds__(ECH_THREADS) void fit_chain_kernel(FitChainArgs a) {
  constexpr int ZW = 16 * T0;
  __shared__ float dzc[3 * 32 * ZW];
  __shared__ double lsum[8 * 8];
  constexpr int NW = WS ? (NTL + 3) / 4 : 1;           // column groups of g's last layer = worker waves
  __shared__ float ws_h[WS ? 16 * 64 : 4];
  __shared__ float ws_part[2][WS ? 4 : 1][16];
  __shared__ float ws_dh[WS ? 4 : 1][WS ? 16 * 64 : 4];
  __shared__ int ws_flag[4];
  static_assert(!WS || (NB == 1 && HT == 4 && T0 == 1 && !PAD), "worker split: one plain row tile per workgroup");
  constexpr int B = 16 * NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  if (Z_MODE == 1 && a.rp_n > 0 && (int)blockIdx.x >= a.rp_first) {      // a rider workgroup: rows of a later minibatch, nothing to wait for
    fit_adam_z_replay_one((long long)((int)blockIdx.x - a.rp_first) * ECH_THREADS + tid, tid & 15, a.z_out, a.zm, a.zv, a.rp_tlast, a.q, a.rp_idx,
                          a.rp_n, a.rp_t_to, a.rp_lr, a.b1, a.b2, a.eps);
    fit_sync_done(a.sy);
    return;
  }
  fit_sync_wait(a.sy);
  const int role = wave >> 1, tile = wave & 1;
  const bool active = tile < NB && role < 3 && (gridDim.y == 1 || role == (int)blockIdx.y);     // gridDim.y = 3: one network per workgroup
  const int rb = 16 * NB * (int)blockIdx.x;          // a minibatch may be split over workgroups (blockIdx.x: NB row tiles each): rows rb ..
  const int row = rb + 16 * tile + j;
  const float mk = row < a.n_valid ? 1.0f : 0.0f;       // short minibatches (the last one of an epoch, a rank's share under data parallelism)
  const int q = a.q, p = a.p, z0 = a.z0, z1 = a.z1, z2 = a.z2;
  const float *th = a.theta, *tT = a.thetaT;
  float *ws = a.ws;
  float l0 = 0.0f, l1 = 0.0f, l2 = 0.0f;          // this wave's loss terms per row: (nll, sse) of its net; role 0 also 0.5 |z|^2
  FITC_T(0);
  if constexpr (WS) {
    if (tid < 4) ws_flag[tid] = 0;
    __syncthreads();
    const int wk = wave == 0 ? 0 : ((wave & 1) && wave < 6 ? 1 + (wave >> 1) : -1);       // waves 0 | 1, 3, 5
    if ((gridDim.y == 1 || blockIdx.y == 0) && wk >= 0 && wk < NW) {
#define FITC_WAIT(f, need) { while (__hip_atomic_load(&ws_flag[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (need)) __builtin_amdgcn_s_sleep(1); __threadfence_block(); }
#define FITC_POST(f) { __threadfence_block(); if (lane == 0) __hip_atomic_fetch_add(&ws_flag[f], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      const EgmMlp &G = a.g;
      const int L = G.n_layers, no = G.dims[L];
      const int wrow = rb + j;
      const float wmk = wrow < a.n_valid ? 1.0f : 0.0f;
      const long long prow = fitc_row(a, wrow);
      const float *vrow = a.v + prow * p;
      f32x4 vv[4];                                   // this worker's four tiles of the data row, requested first (a trip to HBM)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[t][r] = ech_ld(vrow, 64 * wk + 16 * t + 4 * g + r, p);
      f32x4 h[4], zin[1];
      if (wk == 0) {                                 // the chain wave: input, first layer, hidden layers (ecg_g_fwd up to the last layer)
        const float *zrow = a.data_z + prow * q;
        float zsq = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { zin[0][r] = ech_ld(zrow, 4 * g + r, q); zsq = fmaf(zin[0][r], zin[0][r], zsq); }
        l2 = wmk * 0.5f * sum_over_g(zsq);
        FITC_T(1);
        ecg_put<1>(ws + a.xo[0][0], wrow, g, zin);
        {
          EcgA<4> A, Ad;
          EcgW w{th + G.woff[0], 64, q, 64, 0};
          ecg_prime<4, true>(w, A, j, g);
          ech_zero<4>(h);
          ecg_sub<1, 4, 4, true, true>(w, zin, h, A, w, Ad, j, g);
          ecg_bias<4>(w.W + q * 64, 64, 0, g, h);
          ecg_lrelu<4>(h);
        }
        ecg_hidden_fwd<4>(th, G, a.xo[0], ws, wrow, h, j, g);
        ecg_put<4>(ws + a.xo[0][L - 1], wrow, g, h);
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4 *>(ws_h + j * 64 + 16 * t + 4 * g) = h[t];
        FITC_POST(0);
      } else {
        FITC_WAIT(0, 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) h[t] = *reinterpret_cast<const f32x4 *>(ws_h + j * 64 + 16 * t + 4 * g);
      }
      // forward of this worker's column group
      const float *Wl = th + G.woff[L - 1];
      f32x4 o4[4];
      ech_zero<4>(o4);
      {
        EcgW w{Wl, no, 64, no, 64 * wk};
        EcgA<4> A, Ad;
        ecg_prime<4, false>(w, A, j, g);
        ecg_sub<4, 4, 4, false, false>(w, h, o4, A, w, Ad, j, g);
      }
      ecg_bias<4>(Wl + 64 * no, no, 64 * wk, g, o4);
      FITC_T(5);
      float ssq = 0.0f, sraw = 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 64 * wk + 16 * t + 4 * g + r;
          const float d = f < p ? vv[t][r] - o4[t][r] : 0.0f;
          ssq = fmaf(d, d, ssq);
          sraw += f == p ? o4[t][r] : 0.0f;
          o4[t][r] = d;
        }
      ssq = sum_over_g(ssq);
      sraw = sum_over_g(sraw);
      if (g == 0) { ws_part[0][wk][j] = ssq; ws_part[1][wk][j] = sraw; }
      FITC_POST(1);
      FITC_WAIT(1, NW);
      ssq = 0.0f; sraw = 0.0f;
#pragma unroll
      for (int k = 0; k < NW; ++k) { ssq += ws_part[0][k][j]; sraw += ws_part[1][k][j]; }
      float s2, dsraw;
      if (a.sig2_v > 0.0f) { s2 = a.sig2_v; dsraw = 0.0f; }
      else {
        s2 = softplus_acc(sraw) + BGM_EPS;
        dsraw = wmk * (-ssq / (2.0f * s2 * s2) + 0.5f * (float)p / s2) * a.inv_B * sigmoid_f(sraw);
      }
      if (wk == 0) {
        l0 = wmk * (ssq / (2.0f * s2) + 0.5f * (float)p * logf(s2));
        l1 = wmk * ssq;
      }
      const float cmu = -wmk * a.inv_B / s2;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 64 * wk + 16 * t + 4 * g + r;
          o4[t][r] = f < p ? cmu * o4[t][r] : (f == p ? dsraw : 0.0f);
        }
        if (4 * wk + t < NTL) *reinterpret_cast<f32x4 *>(ws + a.dofs[0][L - 1] + (long long)wrow * (16 * NTL) + 16 * (4 * wk + t) + 4 * g) = o4[t];
      }
      FITC_T(6);
      // this worker's share of dh = W_last^T dout: the K range is its 64 output columns (rows of the mirror), clamped to the matrix
      {
        const float *WT = tT + G.woff[L - 1];
        f32x4 dh[4];
        ech_zero<4>(dh);
        float av[2][4][4];
        auto ld = [&](int t, float (&x)[4][4]) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float *wr = WT + (long long)min(64 * wk + 16 * t + 4 * g + r, no - 1) * 64 + j;
#pragma unroll
            for (int u = 0; u < 4; ++u) x[r][u] = wr[16 * u];
          }
        };
        ld(0, av[0]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t + 1 < 4) ld(t + 1, av[(t + 1) & 1]);
          BGM_NO_HOIST();
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) dh[u] = BGM_MFMA(av[t & 1][r][u], o4[t][r], dh[u]);
          BGM_NO_HOIST();
        }
        if (wk != 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4 *>(&ws_dh[wk][j * 64 + 16 * u + 4 * g]) = dh[u];
          FITC_POST(2);
        } else {
          FITC_WAIT(2, NW - 1);
#pragma unroll
          for (int k = 1; k < NW; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) dh[u] += *reinterpret_cast<const f32x4 *>(&ws_dh[k][j * 64 + 16 * u + 4 * g]);
          ecg_mask<4>(dh, h);
          FITC_T(7);
          ecg_hidden_bwd<4>(tT, G, a.xo[0], a.dofs[0], ws, wrow, dh, j, g);
          FITC_T(8);
          ecg_put<4>(ws + a.dofs[0][0], wrow, g, dh);
          if (Z_MODE == 1) {
            f32x4 dx[1];
            EcgA<1> A, Ad;
            EcgW w{tT + G.woff[0], q, 64, q, 0};           // W^T [H x q]
            ecg_prime<1, false>(w, A, j, g);
            ech_zero<1>(dx);
            ecg_sub<4, 1, 1, false, false>(w, dh, dx, A, w, Ad, j, g);
            *reinterpret_cast<f32x4 *>(dzc + wrow * ZW + 4 * g) = dx[0];
          }
          FITC_T(9);
        }
      }
    }
  }
  if (active && !(WS && role == 0)) {
    const long long prow = fitc_row(a, row);
    const float xv = a.x[prow], yv = a.y[prow];
    const float *zrow = a.data_z + prow * q;
    if (role == 0) {
      // the data row is requested first (a random row of the panel: a trip to HBM) and consumed by the likelihood after the forward
      // pass -- loaded there, that latency sat on the chain's critical path (a quarter of the kernel)
      const float *vrow = a.v + prow * p;
      f32x4 vv[NTL];
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
      f32x4 zin[T0], dx[T0];
      ech_zero<T0>(dx);
      float zsq = 0.0f;
#pragma unroll
      for (int t = 0; t < T0; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { zin[t][r] = ech_ld(zrow, 16 * t + 4 * g + r, q); zsq = fmaf(zin[t][r], zin[t][r], zsq); }
      l2 = mk * 0.5f * sum_over_g(zsq);
      f32x4 o[NTL];
      FITC_T(1);
      ecg_g_fwd<HT, NTL, PAD, T0>(th, a.g, a.xo[0], ws, row, zin, o, j, g);
      FITC_T(5);
      float ssq = 0.0f, sraw = 0.0f;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * g + r;
          const float d = f < p ? vv[t][r] - o[t][r] : 0.0f;
          ssq = fmaf(d, d, ssq);
          sraw += f == p ? o[t][r] : 0.0f;
          o[t][r] = d;
        }
      ssq = sum_over_g(ssq);
      sraw = sum_over_g(sraw);
      float s2, dsraw;
      if (a.sig2_v > 0.0f) { s2 = a.sig2_v; dsraw = 0.0f; }
      else {
        s2 = softplus_acc(sraw) + BGM_EPS;
        dsraw = mk * (-ssq / (2.0f * s2 * s2) + 0.5f * (float)p / s2) * a.inv_B * sigmoid_f(sraw);
      }
      l0 = mk * (ssq / (2.0f * s2) + 0.5f * (float)p * logf(s2));
      l1 = mk * ssq;
      const float cmu = -mk * a.inv_B / s2;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * g + r;
          o[t][r] = f < p ? cmu * o[t][r] : (f == p ? dsraw : 0.0f);
        }
      FITC_T(6);
      ecg_g_bwd<HT, NTL, Z_MODE == 1, T0>(tT, a.g, a.xo[0], a.dofs[0], ws, row, o, dx, j, g);
      FITC_T(9);
      if (Z_MODE == 1) {
#pragma unroll
        for (int t = 0; t < T0; ++t) *reinterpret_cast<f32x4 *>(dzc + row * ZW + 16 * t + 4 * g) = dx[t];
      }
    } else {
      const bool is_f = role == 1;
      const EgmMlp &N = is_f ? a.f : a.h;
      f32x4 in[1], o[1], d[1], dx[1];
      ech_zero<1>(dx);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 4 * g + r;
        if (is_f) in[0][r] = f < z0 + z1 ? ech_ld(zrow, f, q) : (f == z0 + z1 ? xv : 0.0f);
        else { const float t = ech_ld(zrow, f < z0 ? f : f + z1, q); in[0][r] = f < z0 + z2 ? t : 0.0f; }
      }
      ecg_head_fwd<T1, T2, T3>(th, N, a.xo[role], ws, row, in, o, j, g);
      const float mu = __shfl(o[0][0], j), sr = __shfl(o[0][1], j);       // outputs (mu, s_raw) live in lane group 0
      const float tgt = is_f ? yv : xv, fixed = is_f ? a.sig2_y : a.sig2_x;
      float dmu, dsr;
      if (!is_f && a.binary) {
        l0 = mk * (fmaxf(mu, 0.0f) - mu * tgt + log1pf(expf(-fabsf(mu))));
        l1 = l0;
        dmu = mk * (sigmoid_f(mu) - tgt) * a.inv_B;
        dsr = 0.0f;
      } else {
        const float dd = tgt - mu;
        float s2;
        if (fixed > 0.0f) { s2 = fixed; dsr = 0.0f; }
        else {
          s2 = softplus_acc(sr) + BGM_EPS;
          dsr = mk * (-dd * dd / (2.0f * s2 * s2) + 0.5f / s2) * a.inv_B * sigmoid_f(sr);
        }
        l0 = mk * (dd * dd / (2.0f * s2) + 0.5f * logf(s2));
        l1 = mk * dd * dd;
        dmu = -mk * dd / s2 * a.inv_B;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int f = 4 * g + r; d[0][r] = f == 0 ? dmu : (f == 1 ? dsr : 0.0f); }
      ecg_head_bwd<T1, T2, T3>(tT, N, a.xo[role], a.dofs[role], ws, row, d, dx, j, g);
      if (Z_MODE == 1) *reinterpret_cast<f32x4 *>(dzc + (role * 32 + row) * ZW + 4 * g) = dx[0];
    }
  }
  if (Z_MODE == 1 && a.pos && tid >= 384 && tid - 384 < min(B, a.n_valid - rb)) {      // row -> batch position map of the dense Adam step that follows
    const int b = rb + tid - 384;
    const long long r_ = fitc_row(a, b);
    a.pos[r_] = b; a.pos[a.pos_n + r_] = a.epoch;
  }
  if (role == 3) {
    // waves 6,7: Adam rewrote theta and its mirror from other CUs since the last launch, so the chains' first touch of every weight
    // tile would be a serial trip to memory; stream both arrays through this XCD's L2 while the chains start (wave 6 the forward
    // array, wave 7 the mirror the backward sweeps read)
    const f32x4 *src = reinterpret_cast<const f32x4 *>(tile == 0 ? th : tT);
    const int n4 = a.n_warm >> 2;
    float sink = 0.0f;
    for (int i = lane; i < n4; i += 64 * 16) {
      f32x4 acc4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < 16; ++k) acc4 += src[min(i + 64 * k, n4 - 1)];
      sink += acc4[0] + acc4[1] + acc4[2] + acc4[3];
    }
    asm volatile("" ::"v"(sink));
  }
  {   // per-wave sums in double (the accumulators are doubles)
    const float s0 = sum_over_j_to_lane15(l0), s1 = sum_over_j_to_lane15(l1), s2_ = sum_over_j_to_lane15(l2);
    if (j == 15 && g == 0) { lsum[wave * 8] = s0; lsum[wave * 8 + 1] = s1; lsum[wave * 8 + 2] = s2_; }
  }
  __syncthreads();
  if (Z_MODE == 1) {
    for (int i0 = tid; i0 < min(B, a.n_valid - rb) * q; i0 += ECH_THREADS) {
      const int b = rb + i0 / q, col = i0 % q, i = b * q + col;
      float v = a.data_z[fitc_row(a, b) * q + col] * a.inv_B + dzc[(0 * 32 + b) * ZW + col];
      if (col < z0 + z1) v += dzc[(1 * 32 + b) * ZW + col];                                  // f: (z0, z1, x)
      if (col < z0) v += dzc[(2 * 32 + b) * ZW + col];                                       // h: (z0, z2)
      else if (col >= z0 + z1 && col < z0 + z1 + z2) v += dzc[(2 * 32 + b) * ZW + col - z1];
      a.dz[i] = v;
      if (a.zm) {                                            // fit_adam_z_kernel's batch-rows form, fused
        const long long e = fitc_row(a, b) * q + col;
        const float m_ = a.b1 * a.zm[e] + (1.0f - a.b1) * v, v_ = a.b2 * a.zv[e] + (1.0f - a.b2) * v * v;
        a.zm[e] = m_; a.zv[e] = v_;
        a.z_out[e] -= a.lr_t * m_ / (sqrtf(v_) + a.eps);
        if (a.t_last && col == 0) a.t_last[fitc_row(a, b)] = a.t_now;
      }
    }
  }
  FITC_T(10);
  if (tid == 0 && a.loss) {
    double s[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int rl = 0; rl < 3; ++rl)
      for (int t = 0; t < NB; ++t)
        for (int k = 0; k < 3; ++k) s[rl][k] += lsum[(2 * rl + t) * 8 + k];
    // [loss_v, sse_v, loss_x, sse_x | bce, loss_y, sse_y, total]   (roles: 0 g -> v, 1 f -> y, 2 h -> x)
    atomicAdd(a.loss + 0, s[0][0]); atomicAdd(a.loss + 1, s[0][1]);
    atomicAdd(a.loss + 2, s[2][0]); atomicAdd(a.loss + 3, s[2][1]);
    atomicAdd(a.loss + 4, s[1][0]); atomicAdd(a.loss + 5, s[1][1]);
    atomicAdd(a.loss + 6, s[0][0] + s[2][0] + s[1][0] + s[0][2]);
  }
  if (Z_MODE == 1) fit_sync_done(a.sy);
}

// gradient tiles: grad[W_l] = X_l^T D_l, grad[b_l] = column sums of D_l  (one pass; canonical parameter order)
template <int NB>
static __global__ __launch_bounds__(ECH_THREADS) void fit_chain_dw_kernel(FitChainArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const int tau = blockIdx.x * ECH_WAVES + wave;
  if (tau < a.n_tiles) {
  const int *td = a.tiles + tau * ECG_TILE_INTS;
  const int xo = td[0], dofs = td[2], xw = td[4], dw = td[5], u = td[6], v = td[7], woff = td[8], n_in = td[9], n_out = td[10], boff = td[11];
  const float *ws = a.ws;
  const int o = 16 * v + j;
  f32x4 w = {0.0f, 0.0f, 0.0f, 0.0f};
  float bs = 0.0f;
  const float *xp = ws + xo + g * xw + 16 * u + j, *dp = ws + dofs + g * dw + o;
  // fused Adam: the parameters' state is requested before the stash, so that the two trips to memory overlap
  const bool fused = a.ad.on != 0, own_b = boff >= 0 && g == 0 && o < n_out;
  bool own[4];
  FitAdamPre pre[4], pre_b;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 16 * u + 4 * g + r;
    own[r] = f < n_in && o < n_out;
    if (fused && own[r]) pre[r] = fit_adam_theta_load(woff + f * n_out + o, a.theta, a.ad);
  }
  if (fused && own_b) pre_b = fit_adam_theta_load(boff + o, a.theta, a.ad);
  float xa[NB * 4], da[NB * 4];
#pragma unroll
  for (int s4 = 0; s4 < NB * 4; ++s4) { xa[s4] = xp[4 * s4 * xw]; da[s4] = dp[4 * s4 * dw]; }
#pragma unroll
  for (int s4 = 0; s4 < NB * 4; ++s4) { w = BGM_MFMA(xa[s4], da[s4], w); bs += da[s4]; }
  bs = sum_over_g(bs);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 16 * u + 4 * g + r;
    if (own[r]) {
      if (fused) fit_adam_theta_apply(woff + f * n_out + o, w[r], pre[r], a.ad);      // (every parameter belongs to exactly one tile)
      else a.grad[woff + f * n_out + o] = w[r];
    }
  }
  if (own_b) {
    if (fused) fit_adam_theta_apply(boff + o, bs, pre_b, a.ad);
    else a.grad[boff + o] = bs;
  }
  }
  fit_sync_done(a.sy);
}
