// fit_kernels.h -- CausalBGM iterative-update step functions on gfx950.
//
// replaces (src/bayesgm/models/causalbgm/base.py):
//   update_g_net :156-180, update_h_net :183-214, update_f_net :217-243  (theta phase)
//   update_latent_variable_sgd :246-302                                  (z phase)
//
// One minibatch = {forward, backward, dW GEMM, reduce, [RCCL all-reduce], Adam+repack} for theta,
// then {forward, backward, Adam-on-Z} with the updated networks.  Forward keeps the forward-packed
// weights LDS-resident, backward the transposed-packed weights (both at once exceed 160 KiB); the
// layer activations round-trip through an HBM workspace in row-major [batch][width] form, which is
// also exactly what the weight-gradient GEMM (contraction over rows) wants to read.
#pragma once
#include "z_replay.h"
#include "fit_types.h"




__device__ __forceinline__ long long fit_row(const FitKArgs &a, long long b) {
  return a.idx ? (long long)a.idx[b] : a.row_lo + b;
}

template <int NT>
__device__ __forceinline__ void store_tiles(float *base, int width, long long brow, bool ok, int g,
                                            const f32x4 (&r)[1][NT]) {
  if (ok) {
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4 *>(base + brow * width + 16 * t + 4 * g) = r[0][t];
  }
}
template <int NT>
__device__ __forceinline__ void load_tiles(const float *base, int width, long long brow, int g, f32x4 (&r)[1][NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) r[0][t] = *reinterpret_cast<const f32x4 *>(base + brow * width + 16 * t + 4 * g);
}

// ---------------------------------------------------------------------------
// forward: activations + head outputs + per-row losses
// ---------------------------------------------------------------------------
template <int KT1, int KSL1, int NTL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void fit_fwd_kernel(FitKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const CausalMeta &m = a.m;
  lds_fill_fast(lds, a.blob, m.total);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4, lane_off = 64 * g + j;
  const long long n_tiles = (a.B + 15) / 16;
  float *ws = a.wsp;
  double acc_loss[7] = {0, 0, 0, 0, 0, 0, 0};
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * WAVES) {
    BGM_NO_HOIST();
    long long b = tile * 16 + j;
    const bool ok = b < a.B;
    b = ok ? b : a.B - 1;
    const long long row = fit_row(a, b);
    float xr[1] = {a.x[row]}, yr[1] = {a.y[row]};
    f32x4 zin[1][KT1];
    {
      const float *zr = a.data_z + row * (long long)m.q;
#pragma unroll
      for (int t = 0; t < KT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * r + g;
          const float val = (f < m.q) ? zr[f] : (f == m.q ? xr[0] : 0.0f);
          zin[0][t][r] = val;
          if (ok) ws[a.ws.zin + b * (16 * KT1) + f] = val;
        }
    }
    // ---- g
    float ssq = 0.0f, sraw_v = 0.0f;
    {
      f32x4 h[1][4];
      dense<KT1, KSL1, 4, 1>(lds + m.w1g, lds + m.b1g, lane_off, g, zin, h);
      lrelu_inplace<4, 1>(h);
      store_tiles<4>(ws + a.ws.ag, 64, b, ok, g, h);
      for (int l = 0; l < m.n_gh; ++l) {
        BGM_NO_HOIST();
        f32x4 h2[1][4];
        dense<4, 4, 4, 1>(lds + m.wg + l * 4096, lds + m.bg + l * 64, lane_off, g, h, h2);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) h[0][t][r] = lrelu(h2[0][t][r]);
        store_tiles<4>(ws + a.ws.ag + (long long)(l + 1) * a.ws.B * 64, 64, b, ok, g, h);
      }
      f32x4 o[1][NTL];
      dense<4, 4, NTL, 1>(lds + m.wgl, lds + m.bgl, lane_off, g, h, o);
      store_tiles<NTL>(ws + a.ws.outg, 16 * NTL, b, ok, g, o);
      const float *vr = a.v + row * (long long)m.p;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * t + 4 * g + r;
          if (c < m.p) { const float d = vr[c] - o[0][t][r]; ssq = fmaf(d, d, ssq); }
          if (c == m.sig_slot) sraw_v = o[0][t][r];
        }
      ssq = sum_over_g(ssq);
      sraw_v = sum_over_g(sraw_v);  // only one lane group contributed a non-zero term
    }
    // ---- f
    float mu_y, sr_y;
    {
      f32x4 a1[1][4];
      dense<KT1, KSL1, 4, 1>(lds + m.w1f, lds + m.b1f, lane_off, g, zin, a1);
      lrelu_inplace<4, 1>(a1);
      store_tiles<4>(ws + a.ws.af1, 64, b, ok, g, a1);
      f32x4 a2[1][2];
      dense<4, 4, 2, 1>(lds + m.wf2, lds + m.bf2, lane_off, g, a1, a2);
      lrelu_inplace<2, 1>(a2);
      store_tiles<2>(ws + a.ws.af2, 32, b, ok, g, a2);
      f32x4 a3[1][1];
      dense<2, 4, 1, 1>(lds + m.wf3, lds + m.bf3, lane_off, g, a2, a3);
      lrelu_inplace<1, 1>(a3);
      store_tiles<1>(ws + a.ws.af3, 16, b, ok, g, a3);
      f32x4 a4[1][1];
      dense<1, 4, 1, 1>(lds + m.wf4, lds + m.bf4, lane_off, g, a3, a4);
      store_tiles<1>(ws + a.ws.outf, 16, b, ok, g, a4);
      mu_y = __shfl(a4[0][0][0], j);
      sr_y = __shfl(a4[0][0][1], j);
    }
    // ---- h
    float mu_x, sr_x;
    {
      f32x4 a1[1][4];
      dense<KT1, KSL1, 4, 1>(lds + m.w1h, lds + m.b1h, lane_off, g, zin, a1);
      lrelu_inplace<4, 1>(a1);
      store_tiles<4>(ws + a.ws.ah1, 64, b, ok, g, a1);
      f32x4 a2[1][2];
      dense<4, 4, 2, 1>(lds + m.wh2, lds + m.bh2, lane_off, g, a1, a2);
      lrelu_inplace<2, 1>(a2);
      store_tiles<2>(ws + a.ws.ah2, 32, b, ok, g, a2);
      f32x4 a3[1][1];
      dense<2, 4, 1, 1>(lds + m.wh3, lds + m.bh3, lane_off, g, a2, a3);
      lrelu_inplace<1, 1>(a3);
      store_tiles<1>(ws + a.ws.ah3, 16, b, ok, g, a3);
      f32x4 a4[1][1];
      dense<1, 4, 1, 1>(lds + m.wh4, lds + m.bh4, lane_off, g, a3, a4);
      store_tiles<1>(ws + a.ws.outh, 16, b, ok, g, a4);
      mu_x = __shfl(a4[0][0][0], j);
      sr_x = __shfl(a4[0][0][1], j);
    }
    // ---- per-row losses (accurate log/log1p: these are reported values)  base.py:164-169,191-203,229-232
    if (ok && g == 0) {
      float zsq = 0.0f;
      const float *zr = a.data_z + row * (long long)m.q;
      for (int f = 0; f < m.q; ++f) zsq = fmaf(zr[f], zr[f], zsq);
      const float s2v = (m.sig2_v > 0.0f) ? m.sig2_v : (fmaxf(sraw_v, 0.0f) + log1pf(expf(-fabsf(sraw_v)))) + BGM_EPS;
      const float lv = ssq / (2.0f * s2v) + 0.5f * (float)m.p * logf(s2v);
      float lx, ex;
      if (m.binary) {
        lx = fmaxf(mu_x, 0.0f) - mu_x * xr[0] + log1pf(expf(-fabsf(mu_x)));
        ex = lx;
      } else {
        const float s2x = (m.sig2_x > 0.0f) ? m.sig2_x : (fmaxf(sr_x, 0.0f) + log1pf(expf(-fabsf(sr_x)))) + BGM_EPS;
        const float dx = xr[0] - mu_x;
        lx = dx * dx / (2.0f * s2x) + 0.5f * logf(s2x);
        ex = dx * dx;
      }
      const float s2y = (m.sig2_y > 0.0f) ? m.sig2_y : (fmaxf(sr_y, 0.0f) + log1pf(expf(-fabsf(sr_y)))) + BGM_EPS;
      const float dy = yr[0] - mu_y;
      const float ly = dy * dy / (2.0f * s2y) + 0.5f * logf(s2y);
      acc_loss[0] += lv; acc_loss[1] += ssq; acc_loss[2] += lx; acc_loss[3] += ex;
      acc_loss[4] += ly; acc_loss[5] += dy * dy; acc_loss[6] += lv + lx + ly + 0.5f * zsq;
    }
  }
  if (a.loss != nullptr) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      double s = acc_loss[k];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (lane == 0 && s != 0.0) atomicAdd(a.loss + k, s);
    }
  }
}

// ---------------------------------------------------------------------------
// backward: head gradients -> dpre of every layer (theta phase) / dz (z phase)
// ---------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void mul_lrelu_grad(f32x4 (&d)[1][NT], const f32x4 (&act)[1][NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) d[0][t][r] *= (act[0][t][r] > 0.0f) ? 1.0f : BGM_LEAK;
}
template <int NT>
__device__ __forceinline__ void zero_tiles(f32x4 (&d)[1][NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) d[0][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}
// transposed dense (no bias): out = W . in
template <int KT, int NT>
__device__ __forceinline__ void dense_t(const float *wl, int lane_off, const f32x4 (&in)[1][KT], f32x4 (&out)[1][NT]) {
  zero_tiles<NT>(out);
  dense_groups<0, KT, 4, NT, 1>(wl, lane_off, in, out);
}


template <int KT1, int KSL1, int NTL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void fit_bwd_kernel(FitKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const CausalMeta &m = a.m;
  const FitMeta &bm = a.bm;
  lds_fill_fast(lds, a.blob, bm.total);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4, lane_off = 64 * g + j;
  const long long n_tiles = (a.B + 15) / 16;
  float *ws = a.wsp;
  const bool theta = a.z_mode == 0;
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * WAVES) {
    BGM_NO_HOIST();
    long long b = tile * 16 + j;
    const bool ok = b < a.B;
    b = ok ? b : a.B - 1;
    const long long row = fit_row(a, b);
    const float xr = a.x[row], yr = a.y[row];
    f32x4 dzin[1][KT1];
    zero_tiles<KT1>(dzin);
    // ================= g =================
    {
      f32x4 o[1][NTL];
      load_tiles<NTL>(ws + a.ws.outg, 16 * NTL, b, g, o);
      const float *vr = a.v + row * (long long)m.p;
      float ssq = 0.0f, sraw = 0.0f;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * t + 4 * g + r;
          float d = 0.0f;
          if (c < m.p) { d = vr[c] - o[0][t][r]; ssq = fmaf(d, d, ssq); }
          if (c == m.sig_slot) sraw = o[0][t][r];
          o[0][t][r] = d;  // residual v - mu
        }
      ssq = sum_over_g(ssq);
      sraw = sum_over_g(sraw);
      float s2v, dsraw;
      if (m.sig2_v > 0.0f) { s2v = m.sig2_v; dsraw = 0.0f; }
      else {
        s2v = softplus_acc(sraw) + BGM_EPS;
        dsraw = (-ssq / (2.0f * s2v * s2v) + 0.5f * (float)m.p / s2v) * a.inv_B * sigmoid_f(sraw);
      }
      const float cmu = -a.inv_B / s2v;
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * t + 4 * g + r;
          o[0][t][r] = (c < m.p) ? cmu * o[0][t][r] : (c == m.sig_slot ? dsraw : 0.0f);
        }
      if (theta) store_tiles<NTL>(ws + a.ws.dgl, 16 * NTL, b, ok, g, o);
      f32x4 dh[1][4];
      dense_t<NTL, 4>(lds + bm.wgl, lane_off, o, dh);
      for (int l = m.n_gh; l >= 0; --l) {   // hidden activations h_{l+1}, l = n_gh .. 0
        BGM_NO_HOIST();
        f32x4 act[1][4];
        load_tiles<4>(ws + a.ws.ag + (long long)l * a.ws.B * 64, 64, b, g, act);
        mul_lrelu_grad<4>(dh, act);
        if (theta) store_tiles<4>(ws + a.ws.dg + (long long)l * a.ws.B * 64, 64, b, ok, g, dh);
        if (l > 0) {
          f32x4 dn[1][4];
          dense_t<4, 4>(lds + bm.wg + (l - 1) * 4096, lane_off, dh, dn);
#pragma unroll
          for (int t = 0; t < 4; ++t) dh[0][t] = dn[0][t];
        }
      }
      if (!theta) {
        f32x4 dzg[1][KT1];
        dense_t<4, KT1>(lds + bm.w1g, lane_off, dh, dzg);
#pragma unroll
        for (int t = 0; t < KT1; ++t) dzin[0][t] += dzg[0][t];
      }
    }
    // ================= f and h tails =================
#pragma unroll
    for (int net = 0; net < 2; ++net) {
      BGM_NO_HOIST();
      const long long o_out = net == 0 ? a.ws.outf : a.ws.outh;
      const long long o_a1 = net == 0 ? a.ws.af1 : a.ws.ah1, o_a2 = net == 0 ? a.ws.af2 : a.ws.ah2,
                      o_a3 = net == 0 ? a.ws.af3 : a.ws.ah3;
      const long long o_d1 = net == 0 ? a.ws.df1 : a.ws.dh1, o_d2 = net == 0 ? a.ws.df2 : a.ws.dh2,
                      o_d3 = net == 0 ? a.ws.df3 : a.ws.dh3, o_d4 = net == 0 ? a.ws.df4 : a.ws.dh4;
      const int w2 = net == 0 ? bm.wf2 : bm.wh2, w3 = net == 0 ? bm.wf3 : bm.wh3, w4 = net == 0 ? bm.wf4 : bm.wh4,
                w1 = net == 0 ? bm.w1f : bm.w1h;
      f32x4 o[1][1];
      load_tiles<1>(ws + o_out, 16, b, g, o);
      const float mu = o[0][0][0], sr = o[0][0][1];  // valid in lane group 0
      float dmu, dsr;
      if (net == 1 && m.binary) {        // BCE with logits (base.py:191)
        dmu = (sigmoid_f(mu) - xr) * a.inv_B;
        dsr = 0.0f;
      } else {
        const float target = net == 0 ? yr : xr;
        const float fixed = net == 0 ? m.sig2_y : m.sig2_x;
        const float d = target - mu;
        float s2;
        if (fixed > 0.0f) { s2 = fixed; dsr = 0.0f; }
        else {
          s2 = softplus_acc(sr) + BGM_EPS;
          dsr = (-d * d / (2.0f * s2 * s2) + 0.5f / s2) * a.inv_B * sigmoid_f(sr);
        }
        dmu = -d / s2 * a.inv_B;
      }
      f32x4 d4[1][1];
      d4[0][0] = f32x4{(g == 0) ? dmu : 0.0f, (g == 0) ? dsr : 0.0f, 0.0f, 0.0f};
      if (theta) store_tiles<1>(ws + o_d4, 16, b, ok, g, d4);
      f32x4 d3[1][1], act3[1][1];
      dense_t<1, 1>(lds + w4, lane_off, d4, d3);
      load_tiles<1>(ws + o_a3, 16, b, g, act3);
      mul_lrelu_grad<1>(d3, act3);
      if (theta) store_tiles<1>(ws + o_d3, 16, b, ok, g, d3);
      f32x4 d2[1][2], act2[1][2];
      dense_t<1, 2>(lds + w3, lane_off, d3, d2);
      load_tiles<2>(ws + o_a2, 32, b, g, act2);
      mul_lrelu_grad<2>(d2, act2);
      if (theta) store_tiles<2>(ws + o_d2, 32, b, ok, g, d2);
      f32x4 d1[1][4], act1[1][4];
      dense_t<2, 4>(lds + w3 * 0 + w2, lane_off, d2, d1);
      load_tiles<4>(ws + o_a1, 64, b, g, act1);
      mul_lrelu_grad<4>(d1, act1);
      if (theta) store_tiles<4>(ws + o_d1, 64, b, ok, g, d1);
      if (!theta) {
        f32x4 dzn[1][KT1];
        dense_t<4, KT1>(lds + w1, lane_off, d1, dzn);
#pragma unroll
        for (int t = 0; t < KT1; ++t) dzin[0][t] += dzn[0][t];
      }
    }
    if (!theta && ok) {   // dz = dloss/dz + z/B   (prior term, base.py:292-293)
      const float *zr = a.data_z + row * (long long)m.q;
#pragma unroll
      for (int t = 0; t < KT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * r + g;
          if (f < m.q) ws[a.ws.dz + b * m.q + f] = dzin[0][t][r] + zr[f] * a.inv_B;
        }
    }
  }
}

// ---------------------------------------------------------------------------
// weight-gradient GEMM: dW[K x N] = A^T D over the rows of one slice, bias sums.
// blockIdx.y = layer, blockIdx.x = row slice; wave w owns input tiles ti = w, w+4, ...
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void fit_dw_kernel(DwArgs a) {
  const DwLayer L = a.layer[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int KT = L.K / 16, NT = L.N / 16;
  const long long r0 = (long long)blockIdx.x * a.rows_per_slice;
  const long long r1 = min((long long)a.B, r0 + a.rows_per_slice);
  float *out = a.partial + (long long)blockIdx.x * a.partial_stride + L.out_off;
  const float *A = a.ws + L.a_off, *D = a.ws + L.d_off;
  constexpr int NT_MAX = 13;
  // output tiles in register-sized chunks (one chunk up to N = 208), the chunks of a wide layer on gridDim.z workgroups
  for (int to0 = (int)blockIdx.z * NT_MAX; to0 < NT; to0 += (int)gridDim.z * NT_MAX)
  for (int ti = wave; ti < KT; ti += 4) {
    f32x4 acc[NT_MAX];
#pragma unroll
    for (int to = 0; to < NT_MAX; ++to) acc[to] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float bsum[NT_MAX];
#pragma unroll
    for (int to = 0; to < NT_MAX; ++to) bsum[to] = 0.0f;
    constexpr int G = 2;                        // groups of four rows per trip: their loads are in flight together (4: no further gain)
    for (long long r = r0; r < r1; r += 4 * G) {
      float av[G], dv[G][NT_MAX];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const long long rr = r + 4 * u + kk;
        const bool ok = rr < r1;
        av[u] = ok ? A[rr * L.K + 16 * ti + i] : 0.0f;
#pragma unroll
        for (int to = 0; to < NT_MAX; ++to) dv[u][to] = (ok && to0 + to < NT) ? D[rr * L.N + 16 * (to0 + to) + i] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < G; ++u)
#pragma unroll
        for (int to = 0; to < NT_MAX; ++to) {
          if (to0 + to < NT) {
            acc[to] = BGM_MFMA(av[u], dv[u][to], acc[to]);
            bsum[to] += dv[u][to];
          }
        }
    }
    // D tile [in feature 16 ti + 4 kk + r][out feature 16 to + i]
#pragma unroll
    for (int to = 0; to < NT_MAX; ++to) {
      if (to0 + to < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(long long)(16 * ti + 4 * kk + r) * L.N + 16 * (to0 + to) + i] = acc[to][r];
        if (ti == 0) {   // bias gradient = column sums of dpre (wave 0 only: ti == wave for the first pass)
          float s = bsum[to];
          s += __shfl_xor(s, 16);
          s += __shfl_xor(s, 32);
          if (kk == 0) out[(long long)L.K * L.N + 16 * (to0 + to) + i] = s;
        }
      }
    }
  }
}

// gridDim.z of a fit_dw_kernel launch: one workgroup per 13-tile chunk of the widest layer (at most 8)
static inline int fit_dw_chunks(const DwArgs &a) {
  int nt = 1;
  for (int l = 0; l < a.n_layers; ++l) nt = nt > a.layer[l].N / 16 ? nt : a.layer[l].N / 16;
  const int z = (nt + 12) / 13;
  return z < 8 ? z : 8;
}

// grad[c] = sum over slices of partial[slice][src[c]]   (fixed order -> deterministic)
__global__ void fit_grad_reduce_kernel(const float *partial, long long stride, int n_slices, const int *src,
                                       int n_params, float *grad) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_params) return;
  const int s = src[c];
  float acc = 0.0f;
  if (s >= 0)
    for (int k = 0; k < n_slices; ++k) acc += partial[(long long)k * stride + s];
  grad[c] = acc;
}

// Keras Adam (optimizer_v2) on the canonical parameters + scatter into both packed blobs.
__global__ void fit_adam_theta_kernel(float *theta, float *m1, float *m2, const float *grad, int n_params,
                                      float lr_t, float b1, float b2, float eps, float *fwd_blob,
                                      float *bwd_blob, const int *fwd_dst, const int *fwd_dst2, const int *bwd_dst,
                                      float *mirror = nullptr, const int *mirror_dst = nullptr, float *theta_out = nullptr) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_params) return;
  FitAdamTheta ad;
  ad.on = 1; ad.lr_t = lr_t; ad.b1 = b1; ad.b2 = b2; ad.eps = eps;
  ad.m1 = m1; ad.m2 = m2; ad.theta_out = theta_out ? theta_out : theta;      // theta_out: the other of two parameter buffers (bgm_causal_fit_epoch)
  ad.fwd_blob = fwd_blob; ad.bwd_blob = bwd_blob; ad.mirror = mirror;
  ad.fwd_dst = fwd_dst; ad.fwd_dst2 = fwd_dst2; ad.bwd_dst = bwd_dst; ad.mirror_dst = mirror_dst;
  fit_adam_theta_one(c, grad[c], theta, ad);
}

// The same step as its own launch BETWEEN two device-ordered phases (bgm_causal_fit_epoch_dp: the gradient was summed over the ranks
// in front of it): waits at entry for the latent phase that last read the buffer written here, counts the step done at its end.
__global__ void fit_adam_theta_sync_kernel(float *theta, float *m1, float *m2, const float *grad, int n_params,
                                           float lr_t, float b1, float b2, float eps, float *fwd_blob,
                                           float *bwd_blob, const int *fwd_dst, const int *fwd_dst2, const int *bwd_dst,
                                           float *mirror, const int *mirror_dst, float *theta_out, FitSync sy) {
  fit_sync_wait(sy);
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n_params) {
    FitAdamTheta ad;
    ad.on = 1; ad.lr_t = lr_t; ad.b1 = b1; ad.b2 = b2; ad.eps = eps;
    ad.m1 = m1; ad.m2 = m2; ad.theta_out = theta_out ? theta_out : theta;
    ad.fwd_blob = fwd_blob; ad.bwd_blob = bwd_blob; ad.mirror = mirror;
    ad.fwd_dst = fwd_dst; ad.fwd_dst2 = fwd_dst2; ad.bwd_dst = bwd_dst; ad.mirror_dst = mirror_dst;
    fit_adam_theta_one(c, grad[c], theta, ad);
  }
  fit_sync_done(sy);
}

// Adam on the latent matrix.  mode 0 = Keras sparse path (decay + apply on ALL rows, base.py:301),
// mode 1 = lazy (batch rows only).  `pos[row]` = position of the row in the batch of step `epoch` if pos[n + row] == epoch
// (fit_set_pos_kernel stamps the rows of a batch; older stamps are simply stale, nothing has to be cleared).
__global__ void fit_adam_z_kernel(float *z, float *zm, float *zv, const float *dz, const int *pos, long long n,
                                  int q, float lr_t, float b1, float b2, float eps, int lazy, const int *idx, int B, int epoch) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (lazy) {
    if (i >= (long long)B * q) return;
    const long long b = i / q;
    const int f = (int)(i - b * q);
    const long long e = (long long)idx[b] * q + f;
    const float g = dz[i];
    const float m = b1 * zm[e] + (1.0f - b1) * g;
    const float v = b2 * zv[e] + (1.0f - b2) * g * g;
    zm[e] = m; zv[e] = v;
    z[e] -= lr_t * m / (sqrtf(v) + eps);
    if (lazy == 2 && f == 0) const_cast<int *>(pos)[idx[b]] = epoch;      // replay mode: `pos` is t_last, `epoch` this step
    return;
  }
  if (i >= n * q) return;
  const long long row = i / q;
  const int f = (int)(i - row * q);
  const int pb = (pos[n + row] == epoch) ? pos[row] : -1;
  float m = b1 * zm[i], v = b2 * zv[i];
  if (pb >= 0) {
    const float g = dz[(long long)pb * q + f];
    m += (1.0f - b1) * g;
    v += (1.0f - b2) * g * g;
  }
  zm[i] = m; zv[i] = v;
  z[i] -= lr_t * m / (sqrtf(v) + eps);
}

__global__ void fit_set_pos_kernel(int *pos, long long n, const int *idx, int B, int epoch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { pos[idx[b]] = b; pos[n + idx[b]] = epoch; }
}
