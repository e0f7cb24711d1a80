// gx_causal_kernels.h -- CausalBGM sampling on the general-width engine (gx_device.h): log-posterior, persistent
// Metropolis-Hastings chains with the fused effect pass, stand-alone effects, evaluate, encoder -- for deterministic networks of
// ANY hidden widths / depths (params['g_units'], ['e_units'], ['f_units'], ['h_units'] of causalbgm/base.py:64-81).
//
// replaces (src/bayesgm/models/causalbgm/base.py):
//   get_log_posterior            :765-817   -> gx_causal_logp (device), gx_causal_logpost_kernel
//   metropolis_hastings_sampler  :820-904   -> gx_causal_mh_kernel (persistent over iterations)
//   infer_from_latent_posterior  :671-763   -> gx_causal_effects (device), fused for retained draws; gx_causal_effects_kernel
//   evaluate                     :534-570   -> gx_causal_eval_kernel
//   e_net(data_v)                :479,538   -> gx_encode_kernel
// RNG streams, counters and output layouts are those of the resident kernels (causal_kernels.h, oracle/rng.py), so a chain run here
// and a chain run there see the same proposals and uniforms.
//
// One workgroup (4 waves) owns 32 chains.  Per transition: proposal (Philox, one call per 4 features) -> g on the 32 proposals,
// last layer fused with the Gaussian likelihood of the V rows (read from HBM / L2 in the epilogue, never materialised) -> f -> h ->
// one thread per chain assembles the log posterior and accepts.  Weights are read from the padded pack in L2 (gx_device.h).
#pragma once
#include "gx_device.h"

// -D GX_PHASE_CLOCK (development): cycle stamps of thread 0 of every workgroup behind the phases of a transition, summed into
// gx_phase_clk[] (0 proposal, 1 stage g, 2 g hidden, 3 g last, 4 f, 5 h, 6 assemble, 7 accept / copy, 8 effects); gx_api.hip prints them
#ifdef GX_PHASE_CLOCK
__device__ unsigned long long gx_phase_clk[16];
#define GX_PC_DECL unsigned long long gx_pc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, gx_pl = clock64();
#define GX_PC(k) { const unsigned long long t__ = clock64(); gx_pc[k] += t__ - gx_pl; gx_pl = t__; }
#define GX_PC_PARAM , unsigned long long (&gx_pc)[12], unsigned long long &gx_pl
#define GX_PC_ARG , gx_pc, gx_pl
#define GX_PC_FLUSH() { if (threadIdx.x == 0) for (int k__ = 0; k__ < 12; ++k__) atomicAdd(&gx_phase_clk[k__], gx_pc[k__]); }
#else
#define GX_PC_DECL
#define GX_PC(k)
#define GX_PC_PARAM
#define GX_PC_ARG
#define GX_PC_FLUSH()
#endif

struct GxCausalModel {
  GxNet g, f, h, e;
  const float *pack;                 // forward pack of all four networks
  const unsigned *packx;             // split-precision pack of g, f, h (gx_dense_x3; NULL unless bgm_causal_set_precision(2))
  int q, p, z0, z1, z2, binary;
  float sig2_v, sig2_x, sig2_y;      // fixed variances (params['sigma_*'] ** 2) or < 0: learned heads
  int ld;                            // LDS row stride of the activation buffers
  int ldf, db;                       // row stride of the outcome net's activations; doses evaluated per pass of the effect routine (their rows
                                     // stacked in the activation buffers: db x 32 rows x ldf floats fit a buffer)
  int ncg;                           // 32-column groups of g's output
  const int *prior_seg;              // conditional latent prior of IdentifiableCausalBGM (bgm_causal_set_prior) or NULL
  const float *prior_tab;            // [n_segments x (q + 2)]: mu [q], 1 / sigma^2, (q / 2) log sigma^2
};

// LDS carve-up (floats): two activation buffers, the 32 chains' current and proposed states, likelihood partial sums
struct GxLds {
  float *bufA, *bufB, *zc, *zp, *ssep, *sraw, *fo, *ho, *lpn, *lpc, *red;
};
#define GX_MAXDB 4
__host__ __device__ inline int gx_buf_floats(int ld, int ldf, int db) { return GX_ROWS * (ld > db * ldf ? ld : db * ldf); }
__host__ __device__ inline int gx_causal_lds_floats(int ld, int q, int ncg, int ldf = 0, int db = 1) {
  return 2 * gx_buf_floats(ld, ldf, db) + 2 * GX_ROWS * q + ncg * GX_ROWS + (6 + 2 * GX_MAXDB) * GX_ROWS + 64;
}
__device__ __forceinline__ GxLds gx_carve(float *lds, int ld, int q, int ncg, int ldf = 0, int db = 1) {
  GxLds L;
  const int bf = gx_buf_floats(ld, ldf, db);
  L.bufA = lds; L.bufB = L.bufA + bf;
  L.zc = L.bufB + bf; L.zp = L.zc + GX_ROWS * q;
  L.ssep = L.zp + GX_ROWS * q; L.sraw = L.ssep + ncg * GX_ROWS;
  L.fo = L.sraw + GX_ROWS; L.ho = L.fo + 2 * GX_MAXDB * GX_ROWS; L.lpn = L.ho + 2 * GX_ROWS; L.lpc = L.lpn + GX_ROWS; L.red = L.lpc + GX_ROWS;
  return L;
}

// Likelihood epilogue of g's last layer: (v - mu)^2 summed over the lane's columns < p, the variance column p kept aside.  The bias is in
// the accumulators (gx_dense_ld); the lane's data values of a unit are requested one unit ahead (pre / rotate hooks of the engine): read
// in the epilogue itself, each unit ended on an exposed L2 / HBM round trip.
struct GxGLastEpi {
  const float *v; long long row0, n; int p; float *ssep; float *sraw;
  float vc[4][2], vn[4][2];
  __device__ __forceinline__ void pre(int rt, int n0) {
    const int lane = gx_lane(), j = lane & 15, g = lane >> 4;
    const int c0 = n0 + 2 * j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      long long gr = row0 + 16 * rt + 4 * g + r; gr = gr < n ? gr : n - 1;
      const float *vr = v + gr * (long long)p;
      // unconditional requests at clamped columns (a request under a lane condition is an exec-mask branch around it, eight per unit);
      // the epilogue masks the columns >= p.  p even: the lane's two columns are one aligned 8-byte request
      if ((p & 1) == 0) {
        const f32x2 t = *reinterpret_cast<const f32x2 *>(vr + min(c0, p - 2));
        vn[r][0] = t[0]; vn[r][1] = t[1];
      } else {
        vn[r][0] = vr[min(c0, p - 1)];
        vn[r][1] = vr[min(c0 + 1, p - 1)];
      }
    }
  }
  __device__ __forceinline__ void rotate() {
#pragma unroll
    for (int r = 0; r < 4; ++r) { vc[r][0] = vn[r][0]; vc[r][1] = vn[r][1]; }
  }
  __device__ __forceinline__ void operator()(int rt, int n0, const f32x4 &a0, const f32x4 &a1) const {
    const int lane = gx_lane(), j = lane & 15, g = lane >> 4;
    const int c0 = n0 + 2 * j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rt + 4 * g + r;
      const float m0 = a0[r], m1 = a1[r];
      const float d0 = (c0 < p) ? vc[r][0] - m0 : 0.0f, d1 = (c0 + 1 < p) ? vc[r][1] - m1 : 0.0f;
      if (c0 == p) sraw[row] = m0;
      if (c0 + 1 == p) sraw[row] = m1;
      const float s = gx_sum_j(fmaf(d0, d0, d1 * d1));
      if (j == 0) ssep[(n0 >> 5) * GX_ROWS + row] = s;
    }
  }
};

// Stage the 32 rows of a network input into an LDS activation buffer (zero beyond the true width); src(row, col) gives the value.
template <class Src>
__device__ __forceinline__ void gx_stage(float *buf, int ld, int width_pad, Src src, int rows = GX_ROWS) {
  for (int i = threadIdx.x; i < rows * width_pad; i += GX_THREADS) {
    const int r = i / width_pad, c = i - r * width_pad;
    buf[r * ld + c] = src(r, c);
  }
}

// f on `rows` rows (a multiple of 16) of the activation buffers: row R is the latent of chain src(R) (LDS [32][q]) at treatment value xv(R)
// -> fo[2 R] = (mu_y, raw_y).  Collective; ends after a barrier.
template <class Src, class XV>
__device__ __forceinline__ void gx_f_rows(const GxCausalModel &m, const GxLds &L, const float *z, Src src, XV xv, int rows) {
  const int zf = m.z0 + m.z1, q = m.q, ld = m.ldf, nrt = rows >> 4;
  GxPre pre = gx_prefetch(m.pack + m.f.w[0], m.f.pad[1], m.f.pad[1], m.pack + m.f.b[0], nrt);
  gx_stage(L.bufA, ld, m.f.pad[0], [&](int r, int c) { return c < zf ? z[src(r) * q + c] : (c == zf ? xv(r) : 0.0f); }, rows);
  __syncthreads();
  float *cur = gx_hidden(m.f, m.pack, 0, m.f.L - 1, L.bufA, L.bufB, ld, pre, nrt, nrt);
  float *oth = (cur == L.bufA) ? L.bufB : L.bufA;
  const int l = m.f.L - 1;
  gx_dense(m.pack + m.f.w[l], m.f.pad[l], m.f.pad[l + 1], cur, ld, GxStore<false>{oth, ld, nullptr, gx_store_off(ld)}, nrt, m.pack + m.f.b[l], &pre);
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * rows; i += GX_THREADS) L.fo[i] = oth[(i >> 1) * ld + (i & 1)];
  __syncthreads();
}
// f at treatment values xin(row, dose) for the tile's 32 latents; nd <= m.db doses per call, their rows stacked (row 32 d + r)
template <class XIn>
__device__ __forceinline__ void gx_f_forward(const GxCausalModel &m, const GxLds &L, const float *z, XIn xin, int nd = 1) {
  gx_f_rows(m, L, z, [](int r) { return r & (GX_ROWS - 1); }, [&](int r) { return xin(r & (GX_ROWS - 1), r / GX_ROWS); }, GX_ROWS * nd);
}

// log p(z | x, y, v) + const for the tile's 32 rows, z in LDS [32][q]; result in L.lpn[row].  base.py:765-817.
#ifndef GX_PHASE_CLOCK
__device__ __forceinline__ void gx_causal_logp(const GxCausalModel &m, const GxLds &L, const float *z, const float *x, const float *y,
                                               const float *v, long long row0, long long n) {
#else
__device__ __forceinline__ void gx_causal_logp(const GxCausalModel &m, const GxLds &L, const float *z, const float *x, const float *y,
                                               const float *v, long long row0, long long n, unsigned long long (&gx_pc)[12], unsigned long long &gx_pl) {
#endif
  const int q = m.q, ld = m.ld;
  // ---- g: z -> (mu_v [p], raw_v)
  GxPre pre = gx_prefetch(m.pack + m.g.w[0], m.g.pad[1], m.g.pad[1], m.pack + m.g.b[0]);
  gx_stage(L.bufA, ld, m.g.pad[0], [&](int r, int c) { return c < q ? z[r * q + c] : 0.0f; });
  __syncthreads();
  GX_PC(1);
  {
    float *cur = gx_hidden(m.g, m.pack, 0, m.g.L - 1, L.bufA, L.bufB, ld, pre);
    GX_PC(2);
    const int l = m.g.L - 1;
    GxGLastEpi ge;
    ge.v = v; ge.row0 = row0; ge.n = n; ge.p = m.p; ge.ssep = L.ssep; ge.sraw = L.sraw;
    gx_dense(m.pack + m.g.w[l], m.g.pad[l], m.g.pad[l + 1], cur, ld, ge, 2, m.pack + m.g.b[l], &pre);
    __syncthreads();
    GX_PC(3);
  }
  // ---- f: (z0, z1, x) -> (mu_y, raw_y)
  gx_f_forward(m, L, z, [&](int r, int) { long long gr = row0 + r; gr = gr < n ? gr : n - 1; return x[gr]; });
  GX_PC(4);
  // ---- h: (z0, z2) -> (mu_x | logit, raw_x)
  {
    const int z0 = m.z0, z1 = m.z1, z2 = m.z2;
    GxPre ph = gx_prefetch(m.pack + m.h.w[0], m.h.pad[1], m.h.pad[1], m.pack + m.h.b[0]);
    gx_stage(L.bufA, ld, m.h.pad[0], [&](int r, int c) { return c < z0 ? z[r * q + c] : (c < z0 + z2 ? z[r * q + z1 + c] : 0.0f); });
    __syncthreads();
    float *cur = gx_hidden(m.h, m.pack, 0, m.h.L - 1, L.bufA, L.bufB, ld, ph);
    float *oth = (cur == L.bufA) ? L.bufB : L.bufA;
    const int l = m.h.L - 1;
    gx_dense(m.pack + m.h.w[l], m.h.pad[l], m.h.pad[l + 1], cur, ld, GxStore<false>{oth, ld, nullptr, gx_store_off(ld)}, 2, m.pack + m.h.b[l], &ph);
    __syncthreads();
    if (threadIdx.x < 2 * GX_ROWS) L.ho[threadIdx.x] = oth[(threadIdx.x >> 1) * ld + (threadIdx.x & 1)];
    __syncthreads();
  }
  GX_PC(5);
  // ---- assemble -(loss_v + loss_x + loss_y + prior)   (base.py:800-816)
  if (threadIdx.x < GX_ROWS) {
    const int r = threadIdx.x;
    long long gr = row0 + r; gr = gr < n ? gr : n - 1;
    float sse = 0.0f;
    for (int c = 0; c < m.ncg; ++c) sse += L.ssep[c * GX_ROWS + r];
    const float s2v = (m.sig2_v > 0.0f) ? m.sig2_v : softplus_f(L.sraw[r]) + BGM_EPS;
    const float xr = x[gr], yr = y[gr];
    const float mu_x = L.ho[2 * r], mu_y = L.fo[2 * r];
    float loss_x;
    if (m.binary) {
      const float l = mu_x, e = fast_exp(-fabsf(l));
      loss_x = vmax(l, 0.0f) - l * xr + ((e < 2.44140625e-4f) ? e * (1.0f - 0.5f * e) : fast_log(1.0f + e));
    } else {
      const float s2x = (m.sig2_x > 0.0f) ? m.sig2_x : softplus_f(L.ho[2 * r + 1]) + BGM_EPS;
      const float dx = xr - mu_x;
      loss_x = 0.5f * (dx * dx * fast_rcp(s2x) + fast_log(s2x));
    }
    const float s2y = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(L.fo[2 * r + 1]) + BGM_EPS;
    const float dy = yr - mu_y;
    const float loss_y = 0.5f * (dy * dy * fast_rcp(s2y) + fast_log(s2y));
    float prior;
    if (m.prior_seg) {            // Z | U ~ N(mu(U), sigma^2(U) I), identifiable.py:541-551
      const float *t = m.prior_tab + (long long)m.prior_seg[gr] * (q + 2);
      float s = 0.0f;
      for (int c = 0; c < q; ++c) { const float d = z[r * q + c] - t[c]; s = fmaf(d, d, s); }
      prior = 0.5f * s * t[q] + t[q + 1];
    } else {
      float s = 0.0f;
      for (int c = 0; c < q; ++c) { const float zz = z[r * q + c]; s = fmaf(zz, zz, s); }
      prior = 0.5f * s;
    }
    L.lpn[r] = -(0.5f * sse * fast_rcp(s2v) + 0.5f * (float)m.p * fast_log(s2v) + loss_x + loss_y + prior);
  }
  __syncthreads();
  GX_PC(6);
}

// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GX_THREADS) void gx_causal_logpost_kernel(GxCausalModel m, const float *x, const float *y, const float *v,
                                                                       const float *z, long long n, float *out) {
  extern __shared__ float lds[];
  const GxLds L = gx_carve(lds, m.ld, m.q, m.ncg, m.ldf, m.db);
  const long long tiles = (n + GX_ROWS - 1) / GX_ROWS;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long row0 = t * GX_ROWS;
    for (int i = threadIdx.x; i < GX_ROWS * m.q; i += GX_THREADS) {
      long long gr = row0 + i / m.q; gr = gr < n ? gr : n - 1;
      L.zc[i] = z[gr * m.q + i % m.q];
    }
    __syncthreads();
    { GX_PC_DECL gx_causal_logp(m, L, L.zc, x, y, v, row0, n GX_PC_ARG); }
    if (threadIdx.x < GX_ROWS && row0 + threadIdx.x < n) out[row0 + threadIdx.x] = L.lpn[threadIdx.x];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Effects of one retained draw (latents z in LDS) -- infer_from_latent_posterior, base.py:671-763.  EFFECT 1: dose-response sums
// over the tile's valid rows into the workgroup's slot [n_keep][n_doses] (draw-major); EFFECT 2: ITE [n][n_keep].  Outcome noise:
// normals_seq(row, iteration, dose) of oracle/rng.py (tag 3), one Philox call per four doses.
// ---------------------------------------------------------------------------------------------------------------------------
struct GxEffArgs {
  int n_keep, sample_y, n_doses;
  const float *x_values;
  float *adrf_slot;      // this workgroup's [n_keep][n_doses] block (EFFECT 1)
  float *ite;            // [n][n_keep] (EFFECT 2)
  unsigned k0, k1;
  // outcome-net cache of the fused sampler (see causal_effects_cached in causal_kernels.h; here per workgroup = 32 chains):
  float2 *cache;         // this workgroup's [n_doses][GX_ROWS] (mean, sd), or NULL (stand-alone effects, evaluate)
  int eff_skip;          // 1: a retained iteration in which none of the workgroup's chains moved reuses them
  unsigned long long *stats;   // [0] += retained tile-iterations served from the cache (2 row tiles per workgroup), or NULL
};

// Outcome-net cache (cached, block-uniform: e.cache != NULL and e.eff_skip): (mean, sd) of every (dose, chain) persist in e.cache; `stale`
// (bit r: chain r moved since its entries were formed; all 32 at a tile's first retained iteration) names the chains whose (chain, dose)
// pairs go through f -- packed densely as the rows of as few passes of 32 m.db rows as hold them.  A row of a pass depends on its own
// operands only: the entries are the bits a pass over all chains at that dose gives, and the sums below run over the cache with the same
// noise in the same order as without it (see gw_kernels.h, whose wave-local form this is).
template <int EFFECT>
__device__ __forceinline__ void gx_causal_effects(const GxCausalModel &m, const GxLds &L, const float *z, long long row0, long long n,
                                                  long long row_base, unsigned it, long long d, const GxEffArgs &e, unsigned stale = 0xFFFFFFFFu,
                                                  bool cached = false) {
  const int nd = (EFFECT == 2) ? 2 : e.n_doses;
  float ykeep = 0.0f;        // EFFECT 2: y(x = 1) of thread `row`
  auto xval = [&](int k) { return (EFFECT == 2) ? (k == 0 ? 1.0f : 0.0f) : e.x_values[k]; };
  if (cached) {
    const int nmoved = __popc(stale);
    if (nmoved) {
      int *list = reinterpret_cast<int *>(L.ssep);        // the moved chains in ascending order (the likelihood's partial sums are free here)
      if (threadIdx.x < GX_ROWS && ((stale >> threadIdx.x) & 1u)) list[__popc(stale & ((1u << threadIdx.x) - 1u))] = threadIdx.x;
      __syncthreads();
      const int npairs = nmoved * nd, cap = GX_ROWS * m.db;
      for (int p0 = 0; p0 < npairs; p0 += cap) {
        const int rows = min(cap, (npairs - p0 + 15) & ~15);
        gx_f_rows(m, L, z, [&](int r) { const int pi = min(p0 + r, npairs - 1); return list[pi / nd]; },
                  [&](int r) { const int pi = min(p0 + r, npairs - 1); return xval(pi % nd); }, rows);
        for (int r = threadIdx.x; r < rows; r += GX_THREADS) {
          const int pi = p0 + r;
          if (pi < npairs) {
            const int ci = pi / nd, k = pi - ci * nd;
            const float s2y = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(L.fo[2 * r + 1]) + BGM_EPS;
            e.cache[k * GX_ROWS + list[ci]] = make_float2(L.fo[2 * r], __builtin_sqrtf(s2y));
          }
        }
        __syncthreads();
      }
      __threadfence_block();
      __syncthreads();
    }
  }
  for (int k0 = 0; k0 < nd; k0 += m.db) {
    const int nb = min(m.db, nd - k0);
    if (!cached) gx_f_forward(m, L, z, [&](int, int dd) { return xval(k0 + dd); }, nb);
    if (threadIdx.x < GX_ROWS) {
      const int r = threadIdx.x;
      const bool valid = row0 + r < n;
      for (int dd = 0; dd < nb; ++dd) {
        const int k = k0 + dd;
        float mean, sd;
        if (!cached) {
          mean = L.fo[2 * (GX_ROWS * dd + r)];
          const float s2y = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(L.fo[2 * (GX_ROWS * dd + r) + 1]) + BGM_EPS;
          sd = __builtin_sqrtf(s2y);
        } else {
          const float *cp = reinterpret_cast<const float *>(e.cache + k * GX_ROWS + r);
          mean = __builtin_nontemporal_load(cp); sd = __builtin_nontemporal_load(cp + 1);
        }
        float yv = mean;
        if (e.sample_y) {
          const unsigned rowid = (unsigned)(row_base + row0 + r);
          const f32x4 nz = box_muller4(philox4x32_10(rowid, it, (unsigned)(k >> 2), TAG_YNOISE, e.k0, e.k1));
          const int w = k & 3;
          const float eps = w == 0 ? nz[0] : (w == 1 ? nz[1] : (w == 2 ? nz[2] : nz[3]));
          yv = fmaf(sd, eps, yv);
        }
        if (EFFECT == 1) {
          float tot = valid ? yv : 0.0f;
          tot += __shfl_xor(tot, 1); tot += __shfl_xor(tot, 2); tot += __shfl_xor(tot, 4); tot += __shfl_xor(tot, 8); tot += __shfl_xor(tot, 16);
          if (r == 0) e.adrf_slot[(long long)d * nd + k] += tot;        // the slot is private to this workgroup: no atomics, fixed order
        } else {
          if (k == 0) ykeep = yv;
          else if (valid) e.ite[(row0 + r) * (long long)e.n_keep + d] = ykeep - yv;
        }
      }
    }
    __syncthreads();
  }
}

struct GxMhArgs {
  GxCausalModel m;
  const float *x, *y, *v;
  long long n, row_base;
  float *state, *logp;
  int init, it_begin, n_iters, burn_in;
  float q_sd;
  unsigned k0, k1;
  unsigned *acc_count;     // [>= it_begin + n_iters] (+=) or NULL
  float *draws;            // [n_keep][n][q] or NULL
  GxEffArgs e;
  float *adrf_partial;     // [gridDim.x][n_keep][n_doses]
};

template <int EFFECT>
__global__ __launch_bounds__(GX_THREADS) void gx_causal_mh_kernel(GxMhArgs a) {
  extern __shared__ float lds[];
  const GxCausalModel &m = a.m;
  const GxLds L = gx_carve(lds, m.ld, m.q, m.ncg, m.ldf, m.db);
  const int q = m.q;
  const long long n = a.n, tiles = (n + GX_ROWS - 1) / GX_ROWS;
  GxEffArgs e = a.e;
  if (EFFECT == 1) e.adrf_slot = a.adrf_partial + (long long)blockIdx.x * e.n_keep * e.n_doses;
  if (EFFECT != 0 && e.cache) e.cache += (long long)blockIdx.x * ((EFFECT == 2) ? 2 : e.n_doses) * GX_ROWS;
  unsigned n_served = 0u;
  GX_PC_DECL
  const int ncall = (q + 15) >> 4;            // Philox calls per lane group: features 16 t + 4 e + g  <-  call g + 4 t, output e
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long row0 = t * GX_ROWS;
    unsigned stale = 0xFFFFFFFFu; // chains whose entries of e.cache are not those of their current state (all, until the tile's first retained iteration)
    // ---- chain state
    if (a.init) {            // current_state ~ N(0, 1), base.py:842 (tag 0, iteration 0)
      for (int i = threadIdx.x; i < GX_ROWS * 4 * ncall; i += GX_THREADS) {
        const int r = i / (4 * ncall), c = i - r * 4 * ncall, g = c & 3, tt = c >> 2;
        const f32x4 nz = box_muller4(philox4x32_10((unsigned)(a.row_base + row0 + r), 0u, (unsigned)(g + 4 * tt), TAG_INIT, a.k0, a.k1));
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int f = 16 * tt + 4 * w + g; if (f < q) L.zc[r * q + f] = nz[w]; }
      }
      __syncthreads();
      gx_causal_logp(m, L, L.zc, a.x, a.y, a.v, row0, n GX_PC_ARG);
      if (threadIdx.x < GX_ROWS) L.lpc[threadIdx.x] = L.lpn[threadIdx.x];
    } else {
      for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
        long long gr = row0 + i / q; gr = gr < n ? gr : n - 1;
        L.zc[i] = a.state[gr * q + i % q];
      }
      if (threadIdx.x < GX_ROWS) { long long gr = row0 + threadIdx.x; gr = gr < n ? gr : n - 1; L.lpc[threadIdx.x] = a.logp[gr]; }
    }
    __syncthreads();
    for (int it = a.it_begin; it < a.it_begin + a.n_iters; ++it) {
      // ---- proposal  z' = z + q_sd * eps   (base.py:862)
      for (int i = threadIdx.x; i < GX_ROWS * 4 * ncall; i += GX_THREADS) {
        const int r = i / (4 * ncall), c = i - r * 4 * ncall, g = c & 3, tt = c >> 2;
        const f32x4 nz = box_muller4(philox4x32_10((unsigned)(a.row_base + row0 + r), (unsigned)it, (unsigned)(g + 4 * tt), TAG_PROP, a.k0, a.k1));
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int f = 16 * tt + 4 * w + g; if (f < q) L.zp[r * q + f] = fmaf(a.q_sd, nz[w], L.zc[r * q + f]); }
      }
      __syncthreads();
      GX_PC(0);
      gx_causal_logp(m, L, L.zp, a.x, a.y, a.v, row0, n GX_PC_ARG);
      // ---- accept / reject   (base.py:868-871).  u(it) = word (it & 3) of Philox(row, it >> 2, 0, TAG_ACC)
      if (threadIdx.x < 64) {
        const int r = threadIdx.x & 31;
        bool acc = false;
        if (threadIdx.x < GX_ROWS) {
          const uint4 uw = philox4x32_10((unsigned)(a.row_base + row0 + r), (unsigned)it >> 2, 0u, TAG_ACC, a.k0, a.k1);
          const unsigned w = (it & 2) ? ((it & 1) ? uw.w : uw.z) : ((it & 1) ? uw.y : uw.x);
          const float u = u01_open(w);
          const float ratio = fast_exp(fminf(L.lpn[r] - L.lpc[r], 0.0f));
          acc = u < ratio;
          L.red[r] = acc ? 1.0f : 0.0f;
          if (acc) L.lpc[r] = L.lpn[r];
        }
        const unsigned long long bal = __ballot(acc && (row0 + r < n) && threadIdx.x < GX_ROWS);
        if (a.acc_count && threadIdx.x == 0) atomicAdd(&a.acc_count[it], (unsigned)__popcll(bal));
        const unsigned long long any = __ballot(acc && threadIdx.x < GX_ROWS);
        if (threadIdx.x == 0) { L.red[GX_ROWS] = bal != 0ull ? 1.0f : 0.0f; reinterpret_cast<unsigned *>(L.red)[GX_ROWS + 1] = (unsigned)any; }      // which chains moved
      }
      __syncthreads();
      stale |= reinterpret_cast<const unsigned *>(L.red)[GX_ROWS + 1];
      for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS)
        if (L.red[i / q] != 0.0f) L.zc[i] = L.zp[i];
      __syncthreads();
      GX_PC(7);
      if (it >= a.burn_in) {
        const long long d = it - a.burn_in;
        if (a.draws) {            // samples.append(current_state.copy())  (base.py:896)
          for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
            const long long gr = row0 + i / q;
            if (gr < n) a.draws[(d * n + gr) * q + i % q] = L.zc[i];
          }
        }
        if (EFFECT != 0) {
          const bool cached = e.cache != nullptr && e.eff_skip;      // block-uniform
          n_served += (cached && stale == 0u) ? 2u : 0u;            // retained tile-iterations (2 row tiles) that needed no pass of the outcome net
          gx_causal_effects<EFFECT>(m, L, L.zc, row0, n, a.row_base, (unsigned)it, d, e, stale, cached);
          stale = 0u;
        }
        GX_PC(8);
      }
    }
    // ---- write the chain state back
    for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
      const long long gr = row0 + i / q;
      if (gr < n) a.state[gr * q + i % q] = L.zc[i];
    }
    if (threadIdx.x < GX_ROWS && row0 + threadIdx.x < n) a.logp[row0 + threadIdx.x] = L.lpc[threadIdx.x];
    __syncthreads();
  }
  if (EFFECT != 0 && e.stats != nullptr && threadIdx.x == 0 && n_served != 0u) atomicAdd(&e.stats[0], (unsigned long long)n_served);
  GX_PC_FLUSH();
}

// stand-alone effects on a tensor of draws [n_keep][n][q]
struct GxEffKArgs {
  GxCausalModel m;
  const float *draws;
  long long n, row_base;
  int burn_in;
  GxEffArgs e;
  float *adrf_partial;
};
template <int EFFECT>
__global__ __launch_bounds__(GX_THREADS) void gx_causal_effects_kernel(GxEffKArgs a) {
  extern __shared__ float lds[];
  const GxCausalModel &m = a.m;
  const GxLds L = gx_carve(lds, m.ld, m.q, m.ncg, m.ldf, m.db);
  const int q = m.q;
  const long long n = a.n, tiles = (n + GX_ROWS - 1) / GX_ROWS;
  GxEffArgs e = a.e;
  if (EFFECT == 1) e.adrf_slot = a.adrf_partial + (long long)blockIdx.x * e.n_keep * e.n_doses;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long row0 = t * GX_ROWS;
    for (int d = 0; d < e.n_keep; ++d) {
      for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
        long long gr = row0 + i / q; gr = gr < n ? gr : n - 1;
        L.zc[i] = a.draws[((long long)d * n + gr) * q + i % q];
      }
      __syncthreads();
      gx_causal_effects<EFFECT>(m, L, L.zc, row0, n, a.row_base, (unsigned)(a.burn_in + d), d, e);
    }
  }
}

// evaluate (base.py:534-570): squared reconstruction errors (fp64 sums) + plug-in effects at the given latents
struct GxEvalArgs {
  GxCausalModel m;
  const float *x, *y, *v, *z;
  long long n;
  const float *x_values; int n_doses;
  double *sums;            // [3] += sum |v - mu_v|^2, sum (x - x_pred)^2, sum (y - mu_y)^2
  float *adrf_partial;     // [gridDim.x][n_doses] (continuous) += per-workgroup sums over rows of f(z0, z1, x_k)
  float *ite;              // [n] (binary)
};
__global__ __launch_bounds__(GX_THREADS) void gx_causal_eval_kernel(GxEvalArgs a) {
  extern __shared__ float lds[];
  const GxCausalModel &m = a.m;
  const GxLds L = gx_carve(lds, m.ld, m.q, m.ncg, m.ldf, m.db);
  const int q = m.q;
  const long long n = a.n, tiles = (n + GX_ROWS - 1) / GX_ROWS;
  double sv = 0.0, sx = 0.0, sy = 0.0;       // thread r < 32 accumulates its rows
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long row0 = t * GX_ROWS;
    for (int i = threadIdx.x; i < GX_ROWS * q; i += GX_THREADS) {
      long long gr = row0 + i / q; gr = gr < n ? gr : n - 1;
      L.zc[i] = a.z[gr * q + i % q];
    }
    __syncthreads();
    { GX_PC_DECL gx_causal_logp(m, L, L.zc, a.x, a.y, a.v, row0, n GX_PC_ARG); }       // leaves sse, f / h outputs in LDS
    if (threadIdx.x < GX_ROWS && row0 + threadIdx.x < n) {
      const int r = threadIdx.x;
      float sse = 0.0f;
      for (int c = 0; c < m.ncg; ++c) sse += L.ssep[c * GX_ROWS + r];
      const float xr = a.x[row0 + r], yr = a.y[row0 + r];
      const float xp = m.binary ? sigmoid_f(L.ho[2 * r]) : L.ho[2 * r];
      sv += (double)sse; sx += (double)((xr - xp) * (xr - xp)); sy += (double)((yr - L.fo[2 * r]) * (yr - L.fo[2 * r]));
    }
    __syncthreads();
    if (m.binary) {
      float y1 = 0.0f;
      for (int k = 0; k < 2; ++k) {
        const float xv = k == 0 ? 1.0f : 0.0f;
        gx_f_forward(m, L, L.zc, [&](int, int) { return xv; });
        if (threadIdx.x < GX_ROWS) {
          if (k == 0) y1 = L.fo[2 * threadIdx.x];
          else if (row0 + threadIdx.x < n) a.ite[row0 + threadIdx.x] = y1 - L.fo[2 * threadIdx.x];
        }
        __syncthreads();
      }
    } else {
      for (int k = 0; k < a.n_doses; ++k) {
        const float xv = a.x_values[k];
        gx_f_forward(m, L, L.zc, [&](int, int) { return xv; });
        if (threadIdx.x < GX_ROWS) {
          float tot = (row0 + threadIdx.x < n) ? L.fo[2 * threadIdx.x] : 0.0f;
          tot += __shfl_xor(tot, 1); tot += __shfl_xor(tot, 2); tot += __shfl_xor(tot, 4); tot += __shfl_xor(tot, 8); tot += __shfl_xor(tot, 16);
          if (threadIdx.x == 0) a.adrf_partial[(long long)blockIdx.x * a.n_doses + k] += tot;
        }
        __syncthreads();
      }
    }
  }
  if (threadIdx.x < GX_ROWS) {
#pragma unroll
    for (int o = 16; o; o >>= 1) { sv += __shfl_xor(sv, o); sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); }
    if (threadIdx.x == 0) { atomicAdd(&a.sums[0], sv); atomicAdd(&a.sums[1], sx); atomicAdd(&a.sums[2], sy); }
  }
}

// encoder  z = e(v)  (base.py:479, 538): the V rows are staged through LDS in chunks of the first layer's input width
struct GxEncArgs {
  GxNet e; const float *pack; int p, q, ld, kc;     // kc: columns of V staged per chunk (multiple of 32)
  const float *v; long long n; float *z;
};
__global__ __launch_bounds__(GX_THREADS) void gx_encode_kernel(GxEncArgs a) {
  extern __shared__ float lds[];
  float *bufA = lds, *bufB = lds + GX_ROWS * a.ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const long long tiles = (a.n + GX_ROWS - 1) / GX_ROWS;
  const int N1 = a.e.pad[1], K0 = a.e.pad[0];
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long row0 = t * GX_ROWS;
    // ---- first layer, K chunked: each wave keeps the accumulators of its (at most 4) units across the chunks
    f32x4 acc[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) { acc[s][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[s][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    const int units = 2 * (N1 >> 5);
    for (int ub = 0; ub < units; ub += 4 * GX_WAVES) {          // passes of 16 units (first layers wider than 256 outputs)
      for (int c0 = 0; c0 < K0; c0 += a.kc) {
        const int kc = min(a.kc, K0 - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < GX_ROWS * kc; i += GX_THREADS) {
          const int r = i / kc, c = i - r * kc;
          long long gr = row0 + r; gr = gr < a.n ? gr : a.n - 1;
          bufA[r * a.ld + c] = (c0 + c < a.p) ? a.v[gr * (long long)a.p + c0 + c] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int u = ub + wave + GX_WAVES * s;
          if (u < units) {
            const int rt = u & 1, n0 = (u >> 1) << 5;
            const float *ap = bufA + (16 * rt + j) * a.ld + 4 * g;
            const float *wp = a.pack + a.e.w[0] + (size_t)(c0 + 4 * g) * N1 + n0 + 2 * j;
            for (int k0 = 0; k0 < kc; k0 += 16) {
              const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + k0);
              const float *wk = wp + (size_t)k0 * N1;
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                const f32x2 b = *reinterpret_cast<const f32x2 *>(wk + (size_t)w * N1);
                acc[s][0] = BGM_MFMA(av[w], b[0], acc[s][0]); acc[s][1] = BGM_MFMA(av[w], b[1], acc[s][1]);
              }
            }
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int u = ub + wave + GX_WAVES * s;
        if (u < units) {
          if (a.e.L > 1) GxStore<true>{bufB, a.ld, a.pack + a.e.b[0]}(u & 1, (u >> 1) << 5, acc[s][0], acc[s][1]);
          else GxStore<false>{bufB, a.ld, a.pack + a.e.b[0]}(u & 1, (u >> 1) << 5, acc[s][0], acc[s][1]);
        }
        acc[s][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[s][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
    __syncthreads();
    float *cur = bufB, *oth = bufA;
    if (a.e.L > 1) {
      GxPre pe;
      pe.valid = 0;
      cur = gx_hidden(a.e, a.pack, 1, a.e.L - 1, bufB, bufA, a.ld, pe);
      oth = (cur == bufA) ? bufB : bufA;
      const int l = a.e.L - 1;
      gx_dense(a.pack + a.e.w[l], a.e.pad[l], a.e.pad[l + 1], cur, a.ld, GxStore<false>{oth, a.ld, nullptr}, 2, a.pack + a.e.b[l], &pe);
      __syncthreads();
      cur = oth;
    }
    for (int i = threadIdx.x; i < GX_ROWS * a.q; i += GX_THREADS) {
      const long long gr = row0 + i / a.q;
      if (gr < a.n) a.z[gr * a.q + i % a.q] = cur[(i / a.q) * a.ld + i % a.q];
    }
    __syncthreads();
  }
}
