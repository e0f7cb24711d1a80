// bgm_fit_kernels.h -- BGM iterative-update step functions on gfx950.
//
// replaces (src/bayesgm/models/bgm/base.py):
//   update_g_net :145-164, update_latent_variable_sgd :167-187 and the loop body :399-413,
//   with g_net = BaseVariationalNet called with training=True (networks/base.py:98-111): the input
//   BatchNormalization uses the minibatch statistics and updates its moving averages.
// One dual-access weight blob (bgm_kernels.h) serves the forward and the backward kernel; layer
// activations / pre-activation gradients round-trip through an HBM workspace read by the shared
// weight-gradient GEMM (fit_dw_kernel).
#pragma once
#include "bgm_kernels.h"
#include "fit_types.h"

struct BgmFitWs {
  int B;
  long long zn;           // [B][16*KTQ]  normalised input (natural feature order)
  long long zhat;         // [B][16*KTQ]  (z - mu_B) * inv_std
  long long act;          // [NH][B][64]
  long long omean, osraw; // [B][16*NTX] each
  long long dact;         // [NH][B][64]  dpre of the trunk layers
  long long dmean, dsraw; // [B][16*NTX]
  long long dzn;          // [B][16*KTQ]
  long long dz;           // [B][q]
  long long total;
};

struct BgmFitKArgs {
  const float *blob;      // training blob (dual-access, first layer NOT folded with BN)
  BgmMeta m;
  BgmFitWs ws;
  float *wsp;
  const float *x;         // [N x p] data
  const float *data_z;    // [N x q]
  const int *idx;         // [B]
  int B;
  float inv_B;
  const float *bn;        // [4*KQ]: mu_B | inv_std | gamma | beta   (KQ = 16*KTQ, zero padded)
  double *loss;           // [4]: sum loss_x, sum |x-mu|^2, -, -
  // minibatches of at most 32 rows (split != 0): the head tiles are dealt over the waves of gridDim.x workgroups (BgmFitSplit below)
  int split;
  float *part;            // [gridDim.x * BGM_FIT_S][32][64]: each wave's share of d loss / d (last hidden activation)
  unsigned *part_ctr;     // workgroups done (the last one adds the shares up and walks the trunk back)
  // split mode: the batch statistics of the input BatchNorm are formed at the head of the forward kernel (every workgroup for itself,
  // workgroup 0 publishes them in bn_w for the backward kernels and moves the running averages) -- bgm_bn_stats_kernel's job
  float *bn_w; const float *bn_theta; float *bn_moving; int bn_update;
};

// Minibatches of <= 32 rows (the reference's batch_size): with one row tile per wave only two of a workgroup's waves had work and walked
// all 2 x ntx head tiles one after the other (x_dim = 500: 120 + 143 us per pass).  Here wave w takes row tile w & 1 and every
// BGM_FIT_S-th head tile pair from s = w >> 1 on (workgroups interleaved the same way); the trunk is evaluated by every wave (cheap),
// stashed once.  Streamed heads (NTX = 0): BGM_FIT_S tile pairs per round through the LDS stage, the next round's on their way in
// registers meanwhile.  Backward: each wave's share of the hidden gradient goes to `part`; the last workgroup to arrive adds the
// shares in index order (deterministic) and continues with the trunk.
#define BGM_FIT_S 4
struct BgmHeadRounds {
  const float *src;      // global: [ntx][BGM_PAIR]
  float *buf;            // LDS:    [BGM_FIT_S][BGM_PAIR]
  int tid, ntx;
  static constexpr int NR = (BGM_FIT_S * (BGM_PAIR / 4) + 511) / 512;      // f32x4 per thread and round (512 threads)
  f32x4 r[NR];
  __device__ __forceinline__ void fetch(int tx0) {
    const int n4 = min(BGM_FIT_S, ntx - tx0) * (BGM_PAIR / 4);
    const f32x4 *s = reinterpret_cast<const f32x4 *>(src + (long long)tx0 * BGM_PAIR);
#pragma unroll
    for (int k = 0; k < NR; ++k)
      if (tid + 512 * k < n4) r[k] = s[tid + 512 * k];
  }
  __device__ __forceinline__ void commit() {      // (every wave is through with the round in the stage)
    __syncthreads();
    f32x4 *d = reinterpret_cast<f32x4 *>(buf);
#pragma unroll
    for (int k = 0; k < NR; ++k)
      if (tid + 512 * k < BGM_FIT_S * (BGM_PAIR / 4)) d[tid + 512 * k] = r[k];
    __syncthreads();
  }
  __device__ __forceinline__ const float *tile(int s) const { return buf + s * BGM_PAIR; }
};

// ---- batch-norm statistics of the batch latents (single block) + moving-average update
// threads 0..255 of a workgroup; `out` = bn (global) or an LDS copy, `pub` != NULL: also published there (+ the running averages)
__device__ __forceinline__ void bgm_bn_stats_body(const float *data_z, const int *idx, int B, int q, int KQ, const float *theta, float *out,
                                                  float *pub, float *moving, int update_moving) {
  // 16 features at a time, 16 lanes per feature (rows lane, lane + 16, ...), double sums met through shuffles: no block barrier
  const int fl = threadIdx.x >> 4, l = threadIdx.x & 15;
  for (int f = fl; f < KQ; f += 16) {
    double a = 0.0, b2 = 0.0;
    if (f < q)
      for (int b = l; b < B; b += 16) {
        const double v = data_z[(long long)idx[b] * q + f];
        a += v; b2 += v * v;
      }
#pragma unroll
    for (int o = 8; o; o >>= 1) { a += __shfl_xor(a, o, 16); b2 += __shfl_xor(b2, o, 16); }
    if (l == 0) {
      float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
      if (f < q) {
        const double mu = a / B, var = fmax(b2 / B - mu * mu, 0.0);   // biased variance, as Keras
        v0 = (float)mu;
        v1 = (float)(1.0 / sqrt(var + 1e-3));
        v2 = theta[f];
        v3 = theta[q + f];
        if (update_moving && moving) {   // moving = moving * 0.99 + batch * 0.01
          moving[f] = moving[f] * 0.99f + (float)mu * 0.01f;
          moving[q + f] = moving[q + f] * 0.99f + (float)var * 0.01f;
        }
      }
      out[f] = v0; out[KQ + f] = v1; out[2 * KQ + f] = v2; out[3 * KQ + f] = v3;
      if (pub) { pub[f] = v0; pub[KQ + f] = v1; pub[2 * KQ + f] = v2; pub[3 * KQ + f] = v3; }
    }
  }
}
static __global__ __launch_bounds__(256) void bgm_bn_stats_kernel(const float *data_z, const int *idx, int B, int q, int KQ,
                                                          const float *theta /* gamma|beta|mmean|mvar */, float *bn,
                                                          float *moving /* mmean|mvar in theta, updated */,
                                                          int update_moving) {
  bgm_bn_stats_body(data_z, idx, B, q, KQ, theta, bn, nullptr, moving, update_moving);
}

template <int NT>
__device__ __forceinline__ void st_tiles(float *base, int width, long long brow, bool ok, int g, const f32x4 (&r)[NT]) {
  if (ok) {
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4 *>(base + brow * width + 16 * t + 4 * g) = r[t];
  }
}
template <int NT>
__device__ __forceinline__ void ld_tiles(const float *base, int width, long long brow, int g, f32x4 (&r)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) r[t] = *reinterpret_cast<const f32x4 *>(base + brow * width + 16 * t + 4 * g);
}

template <int KTQ, int NTX, int NH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void bgm_fit_fwd_kernel(BgmFitKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BgmMeta &m = a.m;
  lds_fill_fast(lds, a.blob, m.lds_resident);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  constexpr int KQ = 16 * KTQ;
  const int XW = 16 * m.ntx;
  float *ws = a.wsp;
  double l0 = 0.0, l1 = 0.0;
  if (a.split) {
    static_assert(WAVES == 8, "split mode: two row tiles x BGM_FIT_S head groups");
    const int rt = wave & 1, sp = wave >> 1, GS = (int)gridDim.x * BGM_FIT_S;
    long long b = 16 * rt + j;
    const bool ok = b < a.B, lead = ok && blockIdx.x == 0 && sp == 0;
    b = b < a.B ? b : a.B - 1;
    const long long row = a.idx[b];
    BgmHeadRounds hr;
    if constexpr (NTX == 0) {
      hr.src = a.blob + m.whd; hr.buf = lds + m.stage; hr.tid = threadIdx.x; hr.ntx = m.ntx;
      hr.fetch((int)blockIdx.x * BGM_FIT_S);
    }
    __shared__ float bnl[4 * KQ];
    if (threadIdx.x < 256)
      bgm_bn_stats_body(a.data_z, a.idx, a.B, m.q, KQ, a.bn_theta, bnl, blockIdx.x == 0 ? a.bn_w : nullptr,
                        blockIdx.x == 0 ? a.bn_moving : nullptr, a.bn_update);
    __syncthreads();
    f32x4 zn[KTQ];
#pragma unroll
    for (int t = 0; t < KTQ; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * t + 4 * r + g;
        float zh = 0.0f, v = 0.0f;
        if (f < m.q) {
          zh = (a.data_z[row * (long long)m.q + f] - bnl[f]) * bnl[KQ + f];
          v = fmaf(zh, bnl[2 * KQ + f], bnl[3 * KQ + f]);
        }
        zn[t][r] = v;
        if (lead) { ws[a.ws.zn + b * KQ + f] = v; ws[a.ws.zhat + b * KQ + f] = zh; }
      }
    f32x4 h[4], h2[4];
    bias17<4>(lds + m.b1, g, h);
    fwd17<KTQ, 4>(lds + m.w1, j, g, zn, h);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[t][r] = lrelu(h[t][r]);
    st_tiles<4>(ws + a.ws.act, 64, b, lead, g, h);
    for (int l = 1; l < NH; ++l) {
      BGM_NO_HOIST();
      bias17<4>(lds + m.bh + (l - 1) * 64, g, h2);
      fwd17<4, 4>(lds + m.wh + (l - 1) * (4 * 64 * 17), j, g, h, h2);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[t][r] = lrelu(h2[t][r]);
      st_tiles<4>(ws + a.ws.act + (long long)l * a.ws.B * 64, 64, b, lead, g, h);
    }
    float nll = 0.0f, sse = 0.0f;
    const float *xr = a.x + row * (long long)m.p;
    if constexpr (NTX == 0) hr.commit();
#pragma unroll 1
    for (int tx0 = (int)blockIdx.x * BGM_FIT_S; tx0 < m.ntx; tx0 += GS) {
      BGM_NO_HOIST();
      const int tx = tx0 + sp;
      if constexpr (NTX == 0) { if (tx0 + GS < m.ntx) hr.fetch(tx0 + GS); }
      if (tx < m.ntx) {
        const float *wl = NTX == 0 ? hr.tile(sp) : lds + m.whd + tx * BGM_PAIR;
        f32x4 ms[2];
        ms[0] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * tx + 4 * g);
        ms[1] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * (m.ntx + tx) + 4 * g);
        heads_fwd17(wl, j, g, h, ms);
        if (ok) {
          *reinterpret_cast<f32x4 *>(ws + a.ws.omean + b * XW + 16 * tx + 4 * g) = ms[0];
          *reinterpret_cast<f32x4 *>(ws + a.ws.osraw + b * XW + 16 * tx + 4 * g) = ms[1];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * tx + 4 * g + r;
          if (c < m.p) {
            const float s2 = softplus_acc(ms[1][r]) + BGM_EPS;
            const float d = xr[c] - ms[0][r];
            nll += d * d / (2.0f * s2) + 0.5f * logf(s2);
            sse += d * d;
          }
        }
      }
      if constexpr (NTX == 0) { if (tx0 + GS < m.ntx) hr.commit(); }
    }
    nll = sum_over_g(nll);
    sse = sum_over_g(sse);
    if (ok && g == 0) { l0 += nll; l1 += sse; }
    if (a.loss != nullptr) {
      for (int off = 32; off > 0; off >>= 1) { l0 += __shfl_xor(l0, off); l1 += __shfl_xor(l1, off); }
      if (lane == 0) { atomicAdd(a.loss + 0, l0); atomicAdd(a.loss + 1, l1); }
    }
    return;
  }
  BgmHeadStream hs;
  if constexpr (NTX == 0) hs.begin(a.blob, m, lds);
  const long long n_tiles = (a.B + 15) / 16, passes = bgm_block_passes(n_tiles, WAVES);
  for (long long ps = 0; ps < passes; ++ps) {
    BGM_NO_HOIST();
    long long tile = (ps * gridDim.x + blockIdx.x) * WAVES + wave;
    const bool tile_ok = tile < n_tiles;
    if (NTX > 0 && !tile_ok) break;
    tile = tile_ok ? tile : n_tiles - 1;
    long long b = tile * 16 + j;
    const bool ok = tile_ok && b < a.B;
    b = b < a.B ? b : a.B - 1;
    const long long row = a.idx[b];
    f32x4 zn[KTQ];
#pragma unroll
    for (int t = 0; t < KTQ; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * t + 4 * r + g;
        float zh = 0.0f, v = 0.0f;
        if (f < m.q) {
          zh = (a.data_z[row * (long long)m.q + f] - a.bn[f]) * a.bn[KQ + f];
          v = fmaf(zh, a.bn[2 * KQ + f], a.bn[3 * KQ + f]);
        }
        zn[t][r] = v;
        if (ok) { ws[a.ws.zn + b * KQ + f] = v; ws[a.ws.zhat + b * KQ + f] = zh; }
      }
    f32x4 h[4], h2[4];
    bias17<4>(lds + m.b1, g, h);
    fwd17<KTQ, 4>(lds + m.w1, j, g, zn, h);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[t][r] = lrelu(h[t][r]);
    st_tiles<4>(ws + a.ws.act, 64, b, ok, g, h);
    for (int l = 1; l < NH; ++l) {
      BGM_NO_HOIST();
      bias17<4>(lds + m.bh + (l - 1) * 64, g, h2);
      fwd17<4, 4>(lds + m.wh + (l - 1) * (4 * 64 * 17), j, g, h, h2);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[t][r] = lrelu(h2[t][r]);
      st_tiles<4>(ws + a.ws.act + (long long)l * a.ws.B * 64, 64, b, ok, g, h);
    }
    float nll = 0.0f, sse = 0.0f;
    const float *xr = a.x + row * (long long)m.p;
    auto head = [&](int tx, const float *wl) {
      f32x4 ms[2];
      ms[0] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * tx + 4 * g);
      ms[1] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * (m.ntx + tx) + 4 * g);
      heads_fwd17(wl, j, g, h, ms);
      if (ok) {
        *reinterpret_cast<f32x4 *>(ws + a.ws.omean + b * XW + 16 * tx + 4 * g) = ms[0];
        *reinterpret_cast<f32x4 *>(ws + a.ws.osraw + b * XW + 16 * tx + 4 * g) = ms[1];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * tx + 4 * g + r;
        if (c < m.p) {
          const float s2 = softplus_acc(ms[1][r]) + BGM_EPS;
          const float d = xr[c] - ms[0][r];
          nll += d * d / (2.0f * s2) + 0.5f * logf(s2);
          sse += d * d;
        }
      }
    };
    if constexpr (NTX > 0) {
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) {
        BGM_NO_HOIST();
        head(tx, lds + m.whd + tx * BGM_PAIR);
      }
    } else {
#pragma unroll 1
      for (int tx = 0; tx < m.ntx; ++tx) {
        BGM_NO_HOIST();
        hs.fetch(tx + 1 < m.ntx ? tx + 1 : 0);
        head(tx, hs.tile());
        hs.commit();
      }
    }
    nll = sum_over_g(nll);
    sse = sum_over_g(sse);
    if (ok && g == 0) { l0 += nll; l1 += sse; }
  }
  if (a.loss != nullptr) {
    for (int off = 32; off > 0; off >>= 1) { l0 += __shfl_xor(l0, off); l1 += __shfl_xor(l1, off); }
    if (lane == 0) { atomicAdd(a.loss + 0, l0); atomicAdd(a.loss + 1, l1); }
  }
}

template <int KTQ, int NTX, int NH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void bgm_fit_bwd_kernel(BgmFitKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BgmMeta &m = a.m;
  lds_fill_fast(lds, a.blob, m.lds_resident);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  constexpr int KQ = 16 * KTQ;
  const int XW = 16 * m.ntx;
  float *ws = a.wsp;
  if (a.split) {
    __shared__ int last_wg;
    const int rt = wave & 1, sp = wave >> 1, GS = (int)gridDim.x * BGM_FIT_S;
    const int slot = 16 * rt + j;
    long long b = slot;
    const bool ok = b < a.B;
    b = b < a.B ? b : a.B - 1;
    const long long row = a.idx[b];
    const float *xr = a.x + row * (long long)m.p;
    BgmHeadRounds hr;
    if constexpr (NTX == 0) {
      hr.src = a.blob + m.whd; hr.buf = lds + m.stage; hr.tid = threadIdx.x; hr.ntx = m.ntx;
      hr.fetch((int)blockIdx.x * BGM_FIT_S);
      hr.commit();
    }
    f32x4 dh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) dh[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int tx0 = (int)blockIdx.x * BGM_FIT_S; tx0 < m.ntx; tx0 += GS) {
      BGM_NO_HOIST();
      const int tx = tx0 + sp;
      if constexpr (NTX == 0) { if (tx0 + GS < m.ntx) hr.fetch(tx0 + GS); }
      if (tx < m.ntx) {
        const float *wl = NTX == 0 ? hr.tile(sp) : lds + m.whd + tx * BGM_PAIR;
        const f32x4 mu = *reinterpret_cast<const f32x4 *>(ws + a.ws.omean + b * XW + 16 * tx + 4 * g);
        const f32x4 sr = *reinterpret_cast<const f32x4 *>(ws + a.ws.osraw + b * XW + 16 * tx + 4 * g);
        f32x4 dms[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * tx + 4 * g + r;
          float dm = 0.0f, dsr = 0.0f;
          if (c < m.p) {   // d/d(mean), d/d(s_raw) of the batch-mean loss  (bgm/base.py:151-153)
            const float s2 = softplus_acc(sr[r]) + BGM_EPS;
            const float d = xr[c] - mu[r];
            dm = -d / s2 * a.inv_B;
            dsr = (-d * d / (2.0f * s2 * s2) + 0.5f / s2) * sigmoid_f(sr[r]) * a.inv_B;
          }
          dms[0][r] = dm;
          dms[1][r] = dsr;
        }
        if (ok) {
          *reinterpret_cast<f32x4 *>(ws + a.ws.dmean + b * XW + 16 * tx + 4 * g) = dms[0];
          *reinterpret_cast<f32x4 *>(ws + a.ws.dsraw + b * XW + 16 * tx + 4 * g) = dms[1];
        }
        heads_bwd17(wl, j, g, dms, dh);
      }
      if constexpr (NTX == 0) { if (tx0 + GS < m.ntx) hr.commit(); }
    }
    // this wave's share -> part[blockIdx.x * S + sp][slot][64]; the last workgroup adds the shares in index order
    st_tiles<4>(a.part + ((long long)blockIdx.x * BGM_FIT_S + sp) * 32 * 64, 64, slot, true, g, dh);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(a.part_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_wg = old == gridDim.x - 1;
      if (last_wg) __hip_atomic_store(a.part_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
    __syncthreads();
    if (!last_wg || sp != 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#pragma unroll
    for (int t = 0; t < 4; ++t) dh[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < GS; ++k) {
      f32x4 pk[4];
      ld_tiles<4>(a.part + (long long)k * 32 * 64, 64, slot, g, pk);
#pragma unroll
      for (int t = 0; t < 4; ++t) dh[t] += pk[t];
    }
    for (int l = NH - 1; l >= 0; --l) {
      BGM_NO_HOIST();
      f32x4 act[4];
      ld_tiles<4>(ws + a.ws.act + (long long)l * a.ws.B * 64, 64, b, g, act);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[t][r] *= (act[t][r] > 0.0f) ? 1.0f : BGM_LEAK;
      st_tiles<4>(ws + a.ws.dact + (long long)l * a.ws.B * 64, 64, b, ok, g, dh);
      if (l > 0) {
        f32x4 dn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dn[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        bwd17<4, 4>(lds + m.wh + (l - 1) * (4 * 64 * 17), j, g, dh, dn);
#pragma unroll
        for (int t = 0; t < 4; ++t) dh[t] = dn[t];
      }
    }
    f32x4 dzn[KTQ];
#pragma unroll
    for (int t = 0; t < KTQ; ++t) dzn[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    bwd17<KTQ, 4>(lds + m.w1, j, g, dh, dzn);
    if (ok) {
#pragma unroll
      for (int t = 0; t < KTQ; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) ws[a.ws.dzn + b * KQ + 16 * t + 4 * r + g] = dzn[t][r];
    }
    return;
  }
  BgmHeadStream hs;
  if constexpr (NTX == 0) hs.begin(a.blob, m, lds);
  const long long n_tiles = (a.B + 15) / 16, passes = bgm_block_passes(n_tiles, WAVES);
  for (long long ps = 0; ps < passes; ++ps) {
    BGM_NO_HOIST();
    long long tile = (ps * gridDim.x + blockIdx.x) * WAVES + wave;
    const bool tile_ok = tile < n_tiles;
    if (NTX > 0 && !tile_ok) break;
    tile = tile_ok ? tile : n_tiles - 1;
    long long b = tile * 16 + j;
    const bool ok = tile_ok && b < a.B;
    b = b < a.B ? b : a.B - 1;
    const long long row = a.idx[b];
    const float *xr = a.x + row * (long long)m.p;
    f32x4 dh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) dh[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    auto head = [&](int tx, const float *wl) {
      const f32x4 mu = *reinterpret_cast<const f32x4 *>(ws + a.ws.omean + b * XW + 16 * tx + 4 * g);
      const f32x4 sr = *reinterpret_cast<const f32x4 *>(ws + a.ws.osraw + b * XW + 16 * tx + 4 * g);
      f32x4 dms[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * tx + 4 * g + r;
        float dm = 0.0f, dsr = 0.0f;
        if (c < m.p) {   // d/d(mean), d/d(s_raw) of the batch-mean loss  (bgm/base.py:151-153)
          const float s2 = softplus_acc(sr[r]) + BGM_EPS;
          const float d = xr[c] - mu[r];
          dm = -d / s2 * a.inv_B;
          dsr = (-d * d / (2.0f * s2 * s2) + 0.5f / s2) * sigmoid_f(sr[r]) * a.inv_B;
        }
        dms[0][r] = dm;
        dms[1][r] = dsr;
      }
      if (ok) {
        *reinterpret_cast<f32x4 *>(ws + a.ws.dmean + b * XW + 16 * tx + 4 * g) = dms[0];
        *reinterpret_cast<f32x4 *>(ws + a.ws.dsraw + b * XW + 16 * tx + 4 * g) = dms[1];
      }
      heads_bwd17(wl, j, g, dms, dh);
    };
    if constexpr (NTX > 0) {
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) {
        BGM_NO_HOIST();
        head(tx, lds + m.whd + tx * BGM_PAIR);
      }
    } else {
#pragma unroll 1
      for (int tx = 0; tx < m.ntx; ++tx) {
        BGM_NO_HOIST();
        hs.fetch(tx + 1 < m.ntx ? tx + 1 : 0);
        head(tx, hs.tile());
        hs.commit();
      }
    }
    for (int l = NH - 1; l >= 0; --l) {
      BGM_NO_HOIST();
      f32x4 act[4];
      ld_tiles<4>(ws + a.ws.act + (long long)l * a.ws.B * 64, 64, b, g, act);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[t][r] *= (act[t][r] > 0.0f) ? 1.0f : BGM_LEAK;
      st_tiles<4>(ws + a.ws.dact + (long long)l * a.ws.B * 64, 64, b, ok, g, dh);
      if (l > 0) {
        f32x4 dn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dn[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        bwd17<4, 4>(lds + m.wh + (l - 1) * (4 * 64 * 17), j, g, dh, dn);
#pragma unroll
        for (int t = 0; t < 4; ++t) dh[t] = dn[t];
      }
    }
    f32x4 dzn[KTQ];
#pragma unroll
    for (int t = 0; t < KTQ; ++t) dzn[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    bwd17<KTQ, 4>(lds + m.w1, j, g, dh, dzn);
    if (ok) {
#pragma unroll
      for (int t = 0; t < KTQ; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) ws[a.ws.dzn + b * KQ + 16 * t + 4 * r + g] = dzn[t][r];
    }
  }
}

// ---- batch-norm backward (single block): dgamma, dbeta into grad[], dz (incl. the prior z/B) into ws.dz
static __global__ __launch_bounds__(256) void bgm_bn_bwd_kernel(const float *wsp, BgmFitWs w, const float *bn, int B, int q,
                                                        int KQ, float inv_B, const float *data_z, const int *idx,
                                                        float *grad_gamma_beta /* [2q] or NULL */, float *dz_out,
                                                        int add_prior, float *z_rw = nullptr, float lr_t = 0.0f, float b1 = 0.0f,
                                                        float b2 = 0.0f, float eps = 0.0f) {
  // 16 features at a time, 16 lanes per feature, double sums through shuffles (no block barrier)
  const int fl = threadIdx.x >> 4, l = threadIdx.x & 15;
  for (int f = fl; f < q; f += 16) {
    double a = 0.0, c = 0.0;   // sum dzn, sum dzn*zhat
    for (int b = l; b < B; b += 16) {
      const double d = wsp[w.dzn + (long long)b * KQ + f];
      a += d; c += d * wsp[w.zhat + (long long)b * KQ + f];
    }
#pragma unroll
    for (int o = 8; o; o >>= 1) { a += __shfl_xor(a, o, 16); c += __shfl_xor(c, o, 16); }
    if (l == 0 && grad_gamma_beta) { grad_gamma_beta[f] = (float)c; grad_gamma_beta[q + f] = (float)a; }
    const float m1s = (float)(a / B);   // mean over the batch of dzn
    const float m2s = (float)(c / B);   // mean of dzn*zhat
    if (dz_out) {
      const float gam = bn[2 * KQ + f], inv = bn[KQ + f];
      for (int b = l; b < B; b += 16) {
        const float zh = wsp[w.zhat + (long long)b * KQ + f];
        const float dzh_centered = gam * (wsp[w.dzn + (long long)b * KQ + f] - m1s - zh * m2s);
        float v = inv * dzh_centered;
        if (add_prior) v += data_z[(long long)idx[b] * q + f] * inv_B;   // d/dz of mean(|z|^2/2)
        dz_out[(long long)b * q + f] = v;
        if (z_rw) {      // update_latent_variable_sgd's optimizer step on a FRESH variable (see the note below the kernel)
          const float m = (1.0f - b1) * v, vv = (1.0f - b2) * v * v;
          z_rw[(long long)idx[b] * q + f] -= lr_t * m / (sqrtf(vv) + eps);
        }
      }
    }
  }
}

// update_latent_variable_sgd's optimizer step: `batch_z` is a FRESH tf.Variable every minibatch
// (bgm/base.py:402), so the Adam slots start at zero while `iterations` keeps counting:
//   m = (1-b1) g, v = (1-b2) g^2, z -= lr_t * m / (sqrt(v) + eps)
// (applied in bgm_bn_bwd_kernel, on the gradient it has just formed)
