// gx_fit_kernels.h -- the minibatch steps of CausalBGM.fit on the general-width engine (gx_device.h), for deterministic networks of
// any hidden widths / depths.
//
// replaces (src/bayesgm/models/causalbgm/base.py):
//   update_g_net :156-180, update_h_net :183-214, update_f_net :217-243  -> gx_causal_fit_kernel (z_mode = 0) + fit_dw_kernel
//   update_latent_variable_sgd :246-302 (gradient half)                   -> gx_causal_fit_kernel (z_mode = 1)
// Losses and their derivatives are the formulas of fit_kernels.h / oracle/fit.py (accurate log / log1p: reported values).
//
// A workgroup owns 32 rows of the minibatch and ONE of the three networks (blockIdx.y: g, f, h run side by side -- a minibatch of 32 rows is a
// single latency chain, and the chain through g alone is 12 of the 28 layer steps of g, f, h in sequence); it walks its net forward with every layer input copied to the
// HBM workspace ([B][padded width] row-major -- exactly what the weight-gradient GEMM fit_dw_kernel contracts over rows), the
// loss derivative written over the last layer's output in place, backward through the transposed pack with the pre-activation
// gradients stored next to the activations.  g's p-wide last layer never enters LDS: its output goes to the workspace and is the
// A operand of the first backward product from there.  z_mode = 1: each net's workgroup leaves its share of d loss / dz in a partial buffer and
// the LAST of a tile's three workgroups to finish (a counter per tile) adds them in the fixed order g + f + h + prior: deterministic sums.
// Every layer's first weight block (and bias pair) is requested before the products of the layer in front of it (gx_prefetch).
#pragma once
#include "gx_causal_kernels.h"

struct GxFitNet {
  long long act[GX_MAXL];   // input of layer l:                      ws + act[l], [B][pad[l]]
  long long dy[GX_MAXL];    // d loss / d pre-activation of layer l:  ws + dy[l],  [B][pad[l+1]]  (the last one also holds the raw output)
};

struct GxFitArgs {
  GxCausalModel m;
  const float *packT;       // transposed pack
  GxFitNet wg, wf, wh;
  float *ws;
  long long dz_off;         // [B][q] latent gradients (z_mode = 1)
  long long dzp_off;        // [3][Bcap][q] the networks' shares of them; Bcap = dzp_rows
  long long cnt_off;        // [tiles] arrival counters of the tiles' three workgroups (zero between launches)
  int dzp_rows;
  const float *x, *y, *v, *data_z;
  const int *idx; long long row_lo;
  int B; float inv_B; int z_mode;
  double *loss;             // [8] (see bgm_causal_fit_theta_grad) or NULL
};

__host__ __device__ inline int gx_fit_lds_floats(int ld, int q) { return 2 * GX_ROWS * ld + GX_ROWS * q + 8 * GX_ROWS + 64; }

// NET: 0 = g, 1 = f, 2 = h
template <int NET>
__device__ __forceinline__ void gx_fit_net(const GxFitArgs &a, const GxNet &net, const GxFitNet &w, float *bufA, float *bufB, float *dzacc,
                                           float *lossr, const long long *rowg, long long b0) {
  const GxCausalModel &m = a.m;
  const int ld = m.ld, q = m.q, Ln = net.L;
  const bool theta = a.z_mode == 0;
  const int zf = m.z0 + m.z1;
  // ---- input rows -> LDS and workspace
  {
    const int wp = net.pad[0];
    float *A0 = a.ws + w.act[0] + b0 * wp;
    for (int i = threadIdx.x; i < GX_ROWS * wp; i += GX_THREADS) {
      const int r = i / wp, c = i - r * wp;
      const float *zr = a.data_z + rowg[r] * q;
      float val;
      if (NET == 0) val = c < q ? zr[c] : 0.0f;
      else if (NET == 1) val = c < zf ? zr[c] : (c == zf ? a.x[rowg[r]] : 0.0f);
      else val = c < m.z0 ? zr[c] : (c < m.z0 + m.z2 ? zr[m.z1 + c] : 0.0f);
      bufA[r * ld + c] = val;
      A0[(long long)r * wp + c] = val;
    }
  }
  GxPre pre = gx_prefetch(m.pack + net.w[0], net.pad[1], net.pad[1], m.pack + net.b[0]);
  __syncthreads();
  // ---- forward
  float *cur = bufA, *oth = bufB;
  for (int l = 0; l < Ln - 1; ++l) {
    const GxPre nx = gx_prefetch(m.pack + net.w[l + 1], net.pad[l + 2], net.pad[l + 2], m.pack + net.b[l + 1]);
    gx_dense(m.pack + net.w[l], net.pad[l], net.pad[l + 1], cur, ld,
             GxStoreWs<true>{oth, ld, nullptr, a.ws + w.act[l + 1] + b0 * net.pad[l + 1], net.pad[l + 1]}, 2, m.pack + net.b[l], &pre);
    __syncthreads();
    pre = nx;
    float *t = cur; cur = oth; oth = t;
  }
  const int lo = Ln - 1, NO = net.pad[Ln];
  float *out = a.ws + w.dy[lo] + b0 * NO;
  gx_dense(m.pack + net.w[lo], net.pad[lo], NO, cur, ld, GxStoreWs<false>{nullptr, 0, nullptr, out, NO}, 2, m.pack + net.b[lo], &pre);
  // the first backward product's first block (transposed pack; no bias)
  GxPre pb;
  pb.valid = 0;
  if (Ln - 1 >= (theta ? 1 : 0)) pb = gx_prefetch(a.packT + net.wt[Ln - 1], net.pad[Ln - 1], net.pad[Ln - 1], nullptr);
  __syncthreads();
  // ---- losses and their derivatives, written over the raw output
  if (NET == 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < GX_ROWS; r += GX_WAVES) {
      const bool ok = b0 + r < a.B;
      float *o = out + (long long)r * NO;
      const float *vr = a.v + rowg[r] * (long long)m.p;
      const float sraw = o[m.p];
      const float s2 = (m.sig2_v > 0.0f) ? m.sig2_v : softplus_acc(sraw) + BGM_EPS;
      const float cmu = ok ? a.inv_B / s2 : 0.0f;
      float ssq = 0.0f;
      for (int c = lane; c < NO; c += 64) {
        float d = 0.0f;
        if (c < m.p) { d = o[c] - vr[c]; ssq = fmaf(d, d, ssq); }
        if (c != m.p) o[c] = cmu * d;
      }
      for (int off = 32; off; off >>= 1) ssq += __shfl_xor(ssq, off);
      if (lane == 0) {
        o[m.p] = (ok && !(m.sig2_v > 0.0f)) ? (-ssq / (2.0f * s2 * s2) + 0.5f * (float)m.p / s2) * a.inv_B * sigmoid_f(sraw) : 0.0f;
        lossr[0 * GX_ROWS + r] = ssq / (2.0f * s2) + 0.5f * (float)m.p * logf(s2);
        lossr[1 * GX_ROWS + r] = ssq;
      }
    }
  } else if (threadIdx.x < GX_ROWS) {
    const int r = threadIdx.x;
    const bool ok = b0 + r < a.B;
    float *o = out + (long long)r * NO;
    const float mu = o[0], sr = o[1];
    float dmu, dsr = 0.0f, lv, ex;
    if (NET == 2 && m.binary) {           // BCE with logits (base.py:191)
      const float xr = a.x[rowg[r]];
      dmu = (sigmoid_f(mu) - xr) * a.inv_B;
      lv = fmaxf(mu, 0.0f) - mu * xr + log1pf(expf(-fabsf(mu)));
      ex = lv;
    } else {
      const float target = NET == 1 ? a.y[rowg[r]] : a.x[rowg[r]];
      const float fixed = NET == 1 ? m.sig2_y : m.sig2_x;
      const float d = target - mu;
      float s2;
      if (fixed > 0.0f) s2 = fixed;
      else { s2 = softplus_acc(sr) + BGM_EPS; dsr = (-d * d / (2.0f * s2 * s2) + 0.5f / s2) * a.inv_B * sigmoid_f(sr); }
      dmu = -d / s2 * a.inv_B;
      lv = d * d / (2.0f * s2) + 0.5f * logf(s2);
      ex = d * d;
    }
    o[0] = ok ? dmu : 0.0f; o[1] = ok ? dsr : 0.0f;
    const int k = NET == 1 ? 4 : 2;
    lossr[k * GX_ROWS + r] = lv; lossr[(k + 1) * GX_ROWS + r] = ex;
  }
  __syncthreads();
  // ---- backward: d pre-activation of layer l - 1 from layer l
  for (int l = Ln - 1; l >= 1; --l) {
    const GxBackStore epi{oth, ld, a.ws + w.act[l] + b0 * net.pad[l], net.pad[l], theta ? a.ws + w.dy[l - 1] + b0 * net.pad[l] : nullptr, net.pad[l]};
    GxPre nx;
    nx.valid = 0;
    if (l - 1 >= (theta ? 1 : 0)) nx = gx_prefetch(a.packT + net.wt[l - 1], net.pad[l - 1], net.pad[l - 1], nullptr);
    if (l == Ln - 1) gx_dense<true>(a.packT + net.wt[l], net.pad[l + 1], net.pad[l], out, NO, epi, 2, nullptr, &pb);
    else gx_dense(a.packT + net.wt[l], net.pad[l + 1], net.pad[l], cur, ld, epi, 2, nullptr, &pb);
    __syncthreads();
    pb = nx;
    float *t = cur; cur = oth; oth = t;
  }
  if (!theta) {       // gradient with respect to the network input -> the latent columns it was gathered from
    if (Ln == 1) gx_dense<true>(a.packT + net.wt[0], net.pad[1], net.pad[0], out, NO, GxRawStore{oth, ld}, 2, nullptr, &pb);
    else gx_dense(a.packT + net.wt[0], net.pad[1], net.pad[0], cur, ld, GxRawStore{oth, ld}, 2, nullptr, &pb);
    __syncthreads();
    const int nin = NET == 0 ? q : (NET == 1 ? zf : m.z0 + m.z2);
    for (int i = threadIdx.x; i < GX_ROWS * nin; i += GX_THREADS) {
      const int r = i / nin, c = i - r * nin;
      const int zc = NET == 2 ? (c < m.z0 ? c : m.z1 + c) : c;
      dzacc[r * q + zc] += oth[r * ld + c];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(GX_THREADS) void gx_causal_fit_kernel(GxFitArgs a) {
  extern __shared__ float lds[];
  __shared__ int last_s;
  const GxCausalModel &m = a.m;
  const int ld = m.ld, q = m.q;
  const int net = blockIdx.y;      // 0 g, 1 f, 2 h: the three networks of a tile run as three workgroups
  float *bufA = lds, *bufB = bufA + GX_ROWS * ld, *dzacc = bufB + GX_ROWS * ld, *lossr = dzacc + GX_ROWS * q;
  long long *rowg = reinterpret_cast<long long *>(lossr + 6 * GX_ROWS);        // 32 x 8 bytes = 64 floats
  const int tiles = (a.B + GX_ROWS - 1) / GX_ROWS;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long b0 = (long long)t * GX_ROWS;
    __syncthreads();
    if (threadIdx.x < GX_ROWS) {
      long long b = b0 + threadIdx.x; b = b < a.B ? b : a.B - 1;
      rowg[threadIdx.x] = a.idx ? (long long)a.idx[b] : a.row_lo + b;
    }
    for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) dzacc[i] = 0.0f;
    for (int i = threadIdx.x; i < 6 * GX_ROWS; i += GX_THREADS) lossr[i] = 0.0f;
    __syncthreads();
    if (net == 0) gx_fit_net<0>(a, m.g, a.wg, bufA, bufB, dzacc, lossr, rowg, b0);
    else if (net == 1) gx_fit_net<1>(a, m.f, a.wf, bufA, bufB, dzacc, lossr, rowg, b0);
    else gx_fit_net<2>(a, m.h, a.wh, bufA, bufB, dzacc, lossr, rowg, b0);
    if (a.z_mode) {      // this net's share of d loss / dz; the tile's last workgroup forms dz = g + f + h shares + z / B (prior term, base.py:292-293)
      float *part = a.ws + a.dzp_off;
      const long long ps = (long long)a.dzp_rows * q;
      for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
        const int r = i / q;
        if (b0 + r < a.B) part[net * ps + b0 * q + i] = dzacc[i];
      }
      __threadfence();
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned *cnt = reinterpret_cast<unsigned *>(a.ws + a.cnt_off) + t;
        const unsigned old = atomicAdd(cnt, 1u);
        last_s = old == 2u;
        if (old == 2u) *cnt = 0u;          // zero again for the next launch
      }
      __syncthreads();
      if (last_s) {
        __threadfence();
        float *dz = a.ws + a.dz_off;
        for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
          const int r = i / q, c = i - r * q;
          if (b0 + r < a.B) {
            const long long o = b0 * q + i;
            const float pg = __builtin_nontemporal_load(part + o), pf = __builtin_nontemporal_load(part + ps + o),
                        ph = __builtin_nontemporal_load(part + 2 * ps + o);
            dz[o] = ((pg + pf) + ph) + a.data_z[rowg[r] * q + c] * a.inv_B;
          }
        }
      }
    }
    if (threadIdx.x < GX_ROWS && b0 + threadIdx.x < a.B) {
      const int r = threadIdx.x;
      const float lv = lossr[r], ssq = lossr[GX_ROWS + r], lx = lossr[2 * GX_ROWS + r], ex = lossr[3 * GX_ROWS + r], ly = lossr[4 * GX_ROWS + r],
                  ey = lossr[5 * GX_ROWS + r];
      float tot = lv + lx + ly;          // (the other nets' entries are zero in this workgroup)
      if (net == 0) {
        float zsq = 0.0f;
        const float *zr = a.data_z + rowg[r] * q;
        for (int c = 0; c < q; ++c) zsq = fmaf(zr[c], zr[c], zsq);
        tot += 0.5f * zsq;
      }
      acc[0] += lv; acc[1] += ssq; acc[2] += lx; acc[3] += ex; acc[4] += ly; acc[5] += ey; acc[6] += tot;
    }
  }
  if (a.loss != nullptr && threadIdx.x < 64) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      double s = threadIdx.x < GX_ROWS ? acc[k] : 0.0;
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      if (threadIdx.x == 0 && s != 0.0) atomicAdd(a.loss + k, s);
    }
  }
}
