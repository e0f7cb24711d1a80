// gw_kernels.h -- CausalBGM sampling on the general-width engine with ONE ROW TILE PER WAVE ("gw"): log posterior, persistent
// Metropolis-Hastings chains with the fused effect pass, stand-alone effects -- for deterministic networks of any hidden widths up to
// the LDS budget below (hidden layers up to 128 wide; wider ones keep the workgroup-tile kernels of gx_causal_kernels.h).
//
// replaces (src/bayesgm/models/causalbgm/base.py): get_log_posterior :765-817, metropolis_hastings_sampler :820-904,
// infer_from_latent_posterior :671-763 -- the same functions, Philox streams, accept rule and outputs as gx_causal_kernels.h.
//
// Why a second mapping.  gx_causal_kernels.h deals the (row tile, column group) units of a layer to the four waves of a workgroup that
// owns 32 rows and separates layers by workgroup barriers: at narrow layers a wave gets ONE unit of 32 MFMAs per layer, 25 barriers per
// transition, and the code between the products (staging loops over run-time widths, per-unit epilogues, thread-0 phases) issues 7
// VALU instructions per MFMA; the time per iteration does not move with the occupancy (profiles/r05_gx_phases.txt,
// r05_pmc_sq_gx_default_widths.txt).  Here a WAVE owns 16 chains and walks every layer of g, f, h on them alone:
//   * activations [16][ld] in the wave's own LDS region (same conflict-free stride as gx); a layer's output is written and read back by
//     the same wave -- LDS serves a wave's requests in order, so NO barrier separates layers, and there is none in a transition;
//   * the dense routine is gx_dense_ld with (first unit 0, stride 1): the wave walks all column groups, each unit's first weight
//     block and bias pair requested under the previous unit, the next LAYER's under the current layer's last unit (gx_prefetch);
//     the weights come from the FRAGMENT pack (m.pack here: per K block and column group the 64 lanes' 8 floats contiguous, biases at
//     the padded pack's offsets): two 16-byte requests per lane and block, one address step;
//   * proposal, accept step, log-posterior assembly and the effect sums are wave-local (lane r < 16 owns chain r of the tile);
//   * one ADRF slot, acceptance counter contribution and outcome-net cache per WAVE (the resident kernels' granularity).
#pragma once
#include "gx_causal_kernels.h"

#define GW_ROWS 16
// K extent of layer l as the row-tile-per-wave kernels contract it: the true input width rounded up to the 16 of one K block (the padded
// packs round to 32: a layer of at most 16 inputs -- first layers, the 8 -> 2 heads -- is one block instead of two)
__host__ __device__ inline int gw_k16(const GxNet &n, int l) { return (n.dim[l] + 15) & ~15; }
#define GW_WAVES 4
#define GW_THREADS 256

// LDS row stride of a wave's activation buffers: = 4 (mod 64) floats -- the 16 lanes of a lane group read 16 B at banks 4 j .. 4 j + 3, all
// distinct, like gx_ld's = 8 (mod 64), and a 64-wide net's region comes to 10 112 B: 16 waves per CU instead of 12
__host__ __device__ inline int gw_ld(int width) { return ((width + 63) / 64) * 64 + 4; }
struct GwLds { float *bufA, *bufB, *zc, *zp, *ssq, *sraw; };
// floats of one wave's LDS region: two activation buffers of GW_ROWS x max(ld, db * ldf), the 16 chains' current / proposed states, sums
__host__ __device__ inline int gw_buf_floats(int ld, int ldf, int db) { return GW_ROWS * (ld > db * ldf ? ld : db * ldf); }
__host__ __device__ inline int gw_wave_floats(int ld, int q, int ldf, int db) {
  return 2 * gw_buf_floats(ld, ldf, db) + ((2 * GW_ROWS * q + 3) & ~3) + 2 * GW_ROWS;
}
__device__ __forceinline__ GwLds gw_carve(float *w, int ld, int q, int ldf, int db) {
  GwLds L;
  const int bf = gw_buf_floats(ld, ldf, db);
  L.bufA = w; L.bufB = w + bf;
  L.zc = L.bufB + bf; L.zp = L.zc + GW_ROWS * q;
  L.ssq = L.zp + GW_ROWS * q; L.sraw = L.ssq + GW_ROWS;
  return L;
}

// hidden layers l_begin .. l_end - 1 of `net` on the wave's nrt row tiles (no barriers); pre as in gx_hidden
// One dense layer l of `net` on the wave's nrt row tiles, and the request of a layer's first block ahead of it.  X3: split precision
// (gx_dense_x3 on the split pack; it requests a unit's blocks itself, no first-block prefetch).
// X3: 0 = fp32; split precision: 2 = every layer input within 2 K blocks of 32 (hidden widths <= 64): 16 registers of split input, no
// cross-layer request -- within 128 VGPRs, four waves per SIMD (the split kernels are wait-bound: 1.14 -> 0.83 ms per iteration at the
// default widths against the three-wave form); 4 = up to 4 blocks (hidden widths <= 128): 32 registers, ~158 VGPRs, three waves --
// the LDS of 128-wide rows allows two waves per SIMD anyway, and the once-per-layer split beats re-splitting per column group
// (0.852 against 0.889 ms at [128,128]; gx_dense_x3<2, RESPLIT>).
template <int X3> struct GwPreOf { typedef GxPreX T; };
template <> struct GwPreOf<0> { typedef GxPre T; };
template <int X3, class Epi>
__device__ __forceinline__ void gw_layer(const GxCausalModel &m, const GxNet &net, int l, const float *cur, int ld, Epi epi, int nrt,
                                         typename GwPreOf<X3>::T &pre) {
  if constexpr (X3 == 2) gx_dense_x3<2>(m.packx + net.wx[l], net.pad[l], net.pad[l + 1], cur, ld, epi, nrt, m.pack + net.b[l], nullptr);
  else if constexpr (X3 == 4) gx_dense_x3<4>(m.packx + net.wx[l], net.pad[l], net.pad[l + 1], cur, ld, epi, nrt, m.pack + net.b[l], &pre);
  else gx_dense<false, true>(m.pack + net.w[l], gw_k16(net, l), net.pad[l + 1], cur, ld, epi, nrt, m.pack + net.b[l], &pre, 0, 1);
}
template <int X3>
__device__ __forceinline__ typename GwPreOf<X3>::T gw_pre(const GxCausalModel &m, const GxNet &net, int l, int nrt) {
  if constexpr (X3 == 2) { GxPreX p; p.valid = 0; return p; }      // (no cross-layer request: its 16 registers cost the fourth wave)
  else if constexpr (X3 == 4) return gx_prefetch_x3(m.packx + net.wx[l]);
  else return gx_prefetch<true>(m.pack + net.w[l], net.pad[l + 1], net.pad[l + 1], m.pack + net.b[l], nrt, 0, gw_k16(net, l));
}
template <int X3 = 0>
__device__ __forceinline__ float *gw_hidden(const GxCausalModel &m, const GxNet &net, int l_begin, int l_end, float *cur, float *oth, int ld,
                                            typename GwPreOf<X3>::T &pre, int nrt) {
  const int soff = gx_store_off(ld);
  for (int l = l_begin; l < l_end; ++l) {
    typename GwPreOf<X3>::T nx;
    nx.valid = 0;
    if (l + 1 < net.L) nx = gw_pre<X3>(m, net, l + 1, nrt);
    gw_layer<X3>(m, net, l, cur, ld, GxStore<true>{oth, ld, nullptr, soff}, nrt, pre);
    pre = nx;
    float *t = cur; cur = oth; oth = t;
  }
  return cur;
}

// likelihood epilogue of g's last layer for the wave's 16 rows: the lane's partial sums over its columns stay in registers (acc[r] = rows
// 4 g + r) across the units; data values one unit ahead (pre / rotate hooks of gx_dense_ld)
struct GwGLastEpi {
  const float *v; long long row0, n; int p; float *sraw; float *acc;
  float vc[4][2], vn[4][2];
  const float *vb;            // the tile's first data row (uniform)
  unsigned vo[4];             // the lane's four rows' offsets from it (floats; rows past the panel's end clamped to its last row): formed once
  __device__ __forceinline__ void init() {
    const int lane = gx_lane(), g = lane >> 4;
    vb = v + row0 * (long long)p;
    const int last = (int)min((long long)(GW_ROWS - 1), n - 1 - row0);
#pragma unroll
    for (int r = 0; r < 4; ++r) vo[r] = (unsigned)min(4 * g + r, last) * (unsigned)p;
  }
  __device__ __forceinline__ void pre(int, int n0) {
    const int lane = gx_lane(), j = lane & 15;
    const int c0 = n0 + 2 * j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float *vr = vb + vo[r];
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
  __device__ __forceinline__ void operator()(int, int n0, const f32x4 &a0, const f32x4 &a1) const {
    const int lane = gx_lane(), j = lane & 15, g = lane >> 4;
    const int c0 = n0 + 2 * j;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m0 = a0[r], m1 = a1[r];
      const float d0 = (c0 < p) ? vc[r][0] - m0 : 0.0f, d1 = (c0 + 1 < p) ? vc[r][1] - m1 : 0.0f;
      if (c0 == p) sraw[4 * g + r] = m0;
      if (c0 + 1 == p) sraw[4 * g + r] = m1;
      acc[r] = fmaf(d0, d0, fmaf(d1, d1, acc[r]));
    }
  }
};

// f on nd row tiles of the wave: row R (0 .. 16 nd - 1) is the latent of chain src(R) (LDS [16][q]) at treatment value xin(R).  Returns the
// buffer whose columns 0, 1 of row R hold (mu_y, raw_y).  Lane (r = lane & 15, cq = lane >> 4) stages columns cq, cq + 4, ... of rows r, r + 16, ...
template <int X3 = 0, class Src, class XIn>
__device__ __forceinline__ const float *gw_f_rows(const GxCausalModel &m, const GwLds &L, const float *z, Src src, XIn xin, int nd) {
  const int lane = gx_lane(), zf = m.z0 + m.z1, q = m.q, ld = m.ldf, wp = m.f.pad[0];
  auto pre = gw_pre<X3>(m, m.f, 0, nd);
  for (int R = lane & (GW_ROWS - 1); R < GW_ROWS * nd; R += GW_ROWS) {
    const int ch = src(R);
    const float xv = xin(R);
    for (int c = lane >> 4; c < wp; c += 4) {
      const float zc_ = z[ch * q + min(c, q - 1)];          // unconditional read, then select (no exec-mask branch)
      L.bufA[R * ld + c] = c < zf ? zc_ : (c == zf ? xv : 0.0f);
    }
  }
  float *cur = gw_hidden<X3>(m, m.f, 0, m.f.L - 1, L.bufA, L.bufB, ld, pre, nd);
  float *oth = (cur == L.bufA) ? L.bufB : L.bufA;
  gw_layer<X3>(m, m.f, m.f.L - 1, cur, ld, GxStore<false>{oth, ld, nullptr, gx_store_off(ld)}, nd, pre);
  return oth;
}
// the wave's 16 chains at treatment values xin(row, dose), nd <= m.db doses stacked as row tiles (row 16 d + r)
template <int X3 = 0, class XIn>
__device__ __forceinline__ const float *gw_f_forward(const GxCausalModel &m, const GwLds &L, const float *z, XIn xin, int nd) {
  return gw_f_rows<X3>(m, L, z, [](int R) { return R & (GW_ROWS - 1); }, [&](int R) { return xin(R & (GW_ROWS - 1), R >> 4); }, nd);
}

// log p(z | x, y, v) + const of the wave's 16 rows, z in LDS [16][q]; returned in lane r < 16 for row r.  base.py:765-817.
template <int X3 = 0>
__device__ __forceinline__ float gw_causal_logp(const GxCausalModel &m, const GwLds &L, const float *z, const float *x, const float *y,
                                                const float *v, long long row0, long long n) {
  const int lane = gx_lane(), j = lane & 15, g = lane >> 4, q = m.q, ld = m.ld;
  // ---- g: z -> (mu_v [p], raw_v), fused with the Gaussian likelihood of the V rows
  {
    auto pre = gw_pre<X3>(m, m.g, 0, 1);
    const int wp = m.g.pad[0];
    for (int c = lane >> 4; c < wp; c += 4) {          // lane (r = lane & 15, c = lane >> 4, + 4, ...)
      const int r = lane & (GW_ROWS - 1);
      const float zc_ = z[r * q + min(c, q - 1)];
      L.bufA[r * ld + c] = c < q ? zc_ : 0.0f;
    }
    float *cur = gw_hidden<X3>(m, m.g, 0, m.g.L - 1, L.bufA, L.bufB, ld, pre, 1);
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    GwGLastEpi ge;
    ge.v = v; ge.row0 = row0; ge.n = n; ge.p = m.p; ge.sraw = L.sraw; ge.acc = acc;
    ge.init();
    gw_layer<X3>(m, m.g, m.g.L - 1, cur, ld, ge, 1, pre);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = gx_sum_j(acc[r]);
      if (j == 0) L.ssq[4 * g + r] = s;
    }
  }
  // ---- f: (z0, z1, x) -> (mu_y, raw_y)
  float mu_y = 0.0f, raw_y = 0.0f;
  {
    const float *fo = gw_f_forward<X3>(m, L, z, [&](int r, int) { long long gr = row0 + r; gr = gr < n ? gr : n - 1; return x[gr]; }, 1);
    if (lane < GW_ROWS) { mu_y = fo[lane * m.ldf]; raw_y = fo[lane * m.ldf + 1]; }
  }
  // ---- h: (z0, z2) -> (mu_x | logit, raw_x)
  float mu_x = 0.0f, raw_x = 0.0f;
  {
    const int z0 = m.z0, z1 = m.z1, z2 = m.z2, wp = m.h.pad[0];
    auto ph = gw_pre<X3>(m, m.h, 0, 1);
    for (int c = lane >> 4; c < wp; c += 4) {
      const int r = lane & (GW_ROWS - 1);
      const float zc_ = z[r * q + min(c < z0 ? c : z1 + c, q - 1)];
      L.bufA[r * ld + c] = c < z0 + z2 ? zc_ : 0.0f;
    }
    float *cur = gw_hidden<X3>(m, m.h, 0, m.h.L - 1, L.bufA, L.bufB, ld, ph, 1);
    float *oth = (cur == L.bufA) ? L.bufB : L.bufA;
    gw_layer<X3>(m, m.h, m.h.L - 1, cur, ld, GxStore<false>{oth, ld, nullptr, gx_store_off(ld)}, 1, ph);
    if (lane < GW_ROWS) { mu_x = oth[lane * ld]; raw_x = oth[lane * ld + 1]; }
  }
  // ---- assemble -(loss_v + loss_x + loss_y + prior)   (base.py:800-816), lane r = row r
  float lp = 0.0f;
  if (lane < GW_ROWS) {
    const int r = lane;
    long long gr = row0 + r; gr = gr < n ? gr : n - 1;
    const float sse = L.ssq[r];
    const float s2v = (m.sig2_v > 0.0f) ? m.sig2_v : softplus_f(L.sraw[r]) + BGM_EPS;
    const float xr = x[gr], yr = y[gr];
    float loss_x;
    if (m.binary) {
      const float l = mu_x, e = fast_exp(-fabsf(l));
      loss_x = vmax(l, 0.0f) - l * xr + ((e < 2.44140625e-4f) ? e * (1.0f - 0.5f * e) : fast_log(1.0f + e));
    } else {
      const float s2x = (m.sig2_x > 0.0f) ? m.sig2_x : softplus_f(raw_x) + BGM_EPS;
      const float dx = xr - mu_x;
      loss_x = 0.5f * (dx * dx * fast_rcp(s2x) + fast_log(s2x));
    }
    const float s2y = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(raw_y) + BGM_EPS;
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
    lp = -(0.5f * sse * fast_rcp(s2v) + 0.5f * (float)m.p * fast_log(s2v) + loss_x + loss_y + prior);
  }
  return lp;
}

template <int X3 = 0>
__global__ __launch_bounds__(GW_THREADS, X3 == 4 ? 3 : 4) void gw_causal_logpost_kernel(GxCausalModel m, const float *x, const float *y, const float *v, const float *z,
                                                                       long long n, float *out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), q = m.q;      // (wave: provably uniform, so that the region's pointers and the tile walk live in SGPRs)
  const GwLds L = gw_carve(lds + wave * gw_wave_floats(m.ld, q, m.ldf, m.db), m.ld, q, m.ldf, m.db);
  const long long tiles = (n + GW_ROWS - 1) / GW_ROWS, n_slots = (long long)gridDim.x * GW_WAVES;
  for (long long t = (long long)blockIdx.x * GW_WAVES + wave; t < tiles; t += n_slots) {
    const long long row0 = t * GW_ROWS;
    for (int i = lane; i < GW_ROWS * q; i += 64) {
      long long gr = row0 + i / q; gr = gr < n ? gr : n - 1;
      L.zc[i] = z[gr * q + i % q];
    }
    const float lp = gw_causal_logp<X3>(m, L, L.zc, x, y, v, row0, n);
    if (lane < GW_ROWS && row0 + lane < n) out[row0 + lane] = lp;
  }
}

// Effects of one retained draw of the wave's 16 chains (latents z in LDS) -- infer_from_latent_posterior, base.py:671-763; EFFECT 1:
// dose-response sums over the tile's valid rows into the WAVE's slot [n_keep][n_doses]; EFFECT 2: ITE [n][n_keep].  Outcome noise:
// normals_seq(row, iteration, dose) of oracle/rng.py (tag 3), one Philox call per four doses.
// Outcome-net cache (e.cache != NULL, e.eff_skip): (mean, sd) of every (dose, chain) of the tile persist in e.cache; `stale` (bits 0..15,
// wave-uniform) names the chains that moved since their entries were formed (all 16 at a tile's first retained iteration).  Only THEIR
// (chain, dose) pairs go through f, packed densely as the rows of as few 16-row passes as hold them -- a row of a pass depends on that
// row's operands only, so the values are the bits a pass over all 16 chains at that dose gives; the sums then run over the cache with the
// same noise and in the same order as without it.  At the bench's acceptance rate (0.08: 1.3 chains of 16 move per iteration) that is
// 2-3 passes instead of 20.
template <int EFFECT, int X3 = 0>
__device__ __forceinline__ void gw_causal_effects(const GxCausalModel &m, const GwLds &L, const float *z, long long row0, long long n,
                                                  long long row_base, unsigned it, long long d, const GxEffArgs &e, unsigned stale = 0xFFFFu,
                                                  bool cached = false) {
  const int lane = gx_lane();
  const int nd = (EFFECT == 2) ? 2 : e.n_doses;
  float ykeep = 0.0f;
  f32x4 nz = {0.0f, 0.0f, 0.0f, 0.0f};
  const bool valid = lane < GW_ROWS && row0 + lane < n;
  const unsigned rowid = (unsigned)(row_base + row0 + (lane & (GW_ROWS - 1)));
  auto xval = [&](int k) { return (EFFECT == 2) ? (k == 0 ? 1.0f : 0.0f) : e.x_values[k]; };
  if (cached) {
    const int nmoved = __popc(stale & 0xFFFFu);
    if (nmoved) {
      int *list = reinterpret_cast<int *>(L.ssq);          // the moved chains in ascending order (the sums' slots are free outside the log posterior)
      if (lane < GW_ROWS && ((stale >> lane) & 1u)) list[__popc(stale & ((1u << lane) - 1u))] = lane;
      const int npairs = nmoved * nd;
      for (int p0 = 0; p0 < npairs; p0 += GW_ROWS) {
        const int pi = p0 + (lane & (GW_ROWS - 1));
        const bool okp = pi < npairs;
        const int ci = okp ? pi / nd : 0, k = okp ? pi - ci * nd : 0;
        const int ch = list[ci];
        const float *fo = gw_f_rows<X3>(m, L, z, [&](int) { return ch; }, [&](int) { return xval(k); }, 1);
        if (lane < GW_ROWS && okp) {
          const float mean = fo[lane * m.ldf];
          const float s2y = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(fo[lane * m.ldf + 1]) + BGM_EPS;
          e.cache[k * GW_ROWS + ch] = make_float2(mean, __builtin_sqrtf(s2y));
        }
      }
      __threadfence_block();      // the entries are read back below, by other lanes of this wave
    }
  }
  for (int k0 = 0; k0 < nd; k0 += m.db) {
    const int nb = min(m.db, nd - k0);
    const float *fo = nullptr;
    if (!cached) fo = gw_f_forward<X3>(m, L, z, [&](int, int dd) { return xval(k0 + dd); }, nb);
    for (int dd = 0; dd < nb; ++dd) {
      const int k = k0 + dd;
      float mean = 0.0f, sd = 0.0f;
      if (lane < GW_ROWS) {
        if (!cached) {
          mean = fo[(GW_ROWS * dd + lane) * m.ldf];
          const float s2y = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(fo[(GW_ROWS * dd + lane) * m.ldf + 1]) + BGM_EPS;
          sd = __builtin_sqrtf(s2y);
        } else {
          const float *cp = reinterpret_cast<const float *>(e.cache + k * GW_ROWS + lane);
          mean = __builtin_nontemporal_load(cp); sd = __builtin_nontemporal_load(cp + 1);
        }
      }
      float yv = mean;
      if (e.sample_y) {
        if ((k & 3) == 0 || k == k0) nz = box_muller4(philox4x32_10(rowid, it, (unsigned)(k >> 2), TAG_YNOISE, e.k0, e.k1));
        const int w = k & 3;
        const float eps = w == 0 ? nz[0] : (w == 1 ? nz[1] : (w == 2 ? nz[2] : nz[3]));
        yv = fmaf(sd, eps, yv);
      }
      if (EFFECT == 1) {
        float tot = valid ? yv : 0.0f;
        tot += __shfl_xor(tot, 1); tot += __shfl_xor(tot, 2); tot += __shfl_xor(tot, 4); tot += __shfl_xor(tot, 8);
        if (lane == 0) e.adrf_slot[(long long)d * nd + k] += tot;        // the slot is private to this wave: no atomics, fixed order
      } else {
        if (k == 0) ykeep = yv;
        else if (valid) e.ite[(row0 + lane) * (long long)e.n_keep + d] = ykeep - yv;
      }
    }
  }
}

template <int EFFECT, int X3 = 0>
__global__ __launch_bounds__(GW_THREADS, X3 == 4 ? 3 : 4) void gw_causal_mh_kernel(GxMhArgs a) {      // (X3: the split operands of a layer's input live in 32 more registers)
  extern __shared__ float lds[];
  const GxCausalModel &m = a.m;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), q = m.q;      // (wave: provably uniform, so that the region's pointers and the tile walk live in SGPRs)
  const GwLds L = gw_carve(lds + wave * gw_wave_floats(m.ld, q, m.ldf, m.db), m.ld, q, m.ldf, m.db);
  const long long n = a.n, tiles = (n + GW_ROWS - 1) / GW_ROWS;
  const long long slot = (long long)blockIdx.x * GW_WAVES + wave, n_slots = (long long)gridDim.x * GW_WAVES;
  GxEffArgs e = a.e;
  if (EFFECT == 1) e.adrf_slot = a.adrf_partial + slot * e.n_keep * e.n_doses;
  if (EFFECT != 0 && e.cache) e.cache += slot * ((EFFECT == 2) ? 2 : e.n_doses) * GW_ROWS;
  unsigned n_served = 0u;
  const int ncall = (q + 15) >> 4;            // Philox calls per lane group: features 16 t + 4 w + c  <-  call c + 4 t, output w
  const int r_ = lane & (GW_ROWS - 1), c_ = lane >> 4;
  for (long long t = slot; t < tiles; t += n_slots) {
    const long long row0 = t * GW_ROWS;
    const unsigned rowid = (unsigned)(a.row_base + row0 + r_);
    const bool rvalid = lane < GW_ROWS && row0 + lane < n;
    unsigned stale = 0xFFFFu;     // chains whose entries of e.cache are not those of their current state (all, until the tile's first retained iteration)
    float lpc = 0.0f;             // lane r < 16: log posterior of chain r's current state
    // ---- chain state
    if (a.init) {            // current_state ~ N(0, 1), base.py:842 (tag 0, iteration 0)
      for (int tt = 0; tt < ncall; ++tt) {
        const f32x4 nz = box_muller4(philox4x32_10(rowid, 0u, (unsigned)(c_ + 4 * tt), TAG_INIT, a.k0, a.k1));
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int f = 16 * tt + 4 * w + c_; if (f < q) L.zc[r_ * q + f] = nz[w]; }
      }
      lpc = gw_causal_logp<X3>(m, L, L.zc, a.x, a.y, a.v, row0, n);
    } else {
      for (int c = c_; c < q; c += 4) {
        long long gr = row0 + r_; gr = gr < n ? gr : n - 1;
        L.zc[r_ * q + c] = a.state[gr * q + c];
      }
      if (lane < GW_ROWS) { long long gr = row0 + lane; gr = gr < n ? gr : n - 1; lpc = a.logp[gr]; }
    }
    for (int it = a.it_begin; it < a.it_begin + a.n_iters; ++it) {
      // ---- proposal  z' = z + q_sd * eps   (base.py:862)
      for (int tt = 0; tt < ncall; ++tt) {
        const f32x4 nz = box_muller4(philox4x32_10(rowid, (unsigned)it, (unsigned)(c_ + 4 * tt), TAG_PROP, a.k0, a.k1));
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int f = 16 * tt + 4 * w + c_; if (f < q) L.zp[r_ * q + f] = fmaf(a.q_sd, nz[w], L.zc[r_ * q + f]); }
      }
      const float lpn = gw_causal_logp<X3>(m, L, L.zp, a.x, a.y, a.v, row0, n);
      // ---- accept / reject   (base.py:868-871).  u(it) = word (it & 3) of Philox(row, it >> 2, 0, TAG_ACC)
      bool acc = false;
      if (lane < GW_ROWS) {
        const uint4 uw = philox4x32_10(rowid, (unsigned)it >> 2, 0u, TAG_ACC, a.k0, a.k1);
        const unsigned w = (it & 2) ? ((it & 1) ? uw.w : uw.z) : ((it & 1) ? uw.y : uw.x);
        acc = u01_open(w) < fast_exp(fminf(lpn - lpc, 0.0f));
        if (acc) lpc = lpn;
      }
      const unsigned long long bal = __ballot(acc && rvalid), any = __ballot(acc);
      if (a.acc_count && lane == 0 && bal) atomicAdd(&a.acc_count[it], (unsigned)__popcll(bal));
      stale |= (unsigned)(any & 0xFFFFull);
      if ((any >> r_) & 1ull)
        for (int c = c_; c < q; c += 4) L.zc[r_ * q + c] = L.zp[r_ * q + c];
      if (it >= a.burn_in) {
        const long long d = it - a.burn_in;
        if (a.draws) {            // samples.append(current_state.copy())  (base.py:896)
          const long long gr = row0 + r_;
          if (gr < n)
            for (int c = c_; c < q; c += 4) a.draws[(d * n + gr) * q + c] = L.zc[r_ * q + c];
        }
        if (EFFECT != 0) {
          const bool cached = e.cache != nullptr && e.eff_skip;
          n_served += (cached && stale == 0u) ? 1u : 0u;          // retained tile-iterations that needed no pass of the outcome net
          gw_causal_effects<EFFECT, X3>(m, L, L.zc, row0, n, a.row_base, (unsigned)it, d, e, stale, cached);
          stale = 0u;
        }
      }
    }
    // ---- write the chain state back
    {
      const long long gr = row0 + r_;
      if (gr < n)
        for (int c = c_; c < q; c += 4) a.state[gr * q + c] = L.zc[r_ * q + c];
    }
    if (rvalid) a.logp[row0 + lane] = lpc;
  }
  if (EFFECT != 0 && e.stats != nullptr && lane == 0 && n_served != 0u) atomicAdd(&e.stats[0], (unsigned long long)n_served);
}

// stand-alone effects on a tensor of draws [n_keep][n][q]
template <int EFFECT, int X3 = 0>
__global__ __launch_bounds__(GW_THREADS) void gw_causal_effects_kernel(GxEffKArgs a) {
  extern __shared__ float lds[];
  const GxCausalModel &m = a.m;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), q = m.q;      // (wave: provably uniform, so that the region's pointers and the tile walk live in SGPRs)
  const GwLds L = gw_carve(lds + wave * gw_wave_floats(m.ld, q, m.ldf, m.db), m.ld, q, m.ldf, m.db);
  const long long n = a.n, tiles = (n + GW_ROWS - 1) / GW_ROWS;
  const long long slot = (long long)blockIdx.x * GW_WAVES + wave, n_slots = (long long)gridDim.x * GW_WAVES;
  GxEffArgs e = a.e;
  if (EFFECT == 1) e.adrf_slot = a.adrf_partial + slot * e.n_keep * e.n_doses;
  const int r_ = lane & (GW_ROWS - 1), c_ = lane >> 4;
  for (long long t = slot; t < tiles; t += n_slots) {
    const long long row0 = t * GW_ROWS;
    for (int d = 0; d < e.n_keep; ++d) {
      long long gr = row0 + r_; gr = gr < n ? gr : n - 1;
      for (int c = c_; c < q; c += 4) L.zc[r_ * q + c] = a.draws[((long long)d * n + gr) * q + c];
      gw_causal_effects<EFFECT, X3>(m, L, L.zc, row0, n, a.row_base, (unsigned)(a.burn_in + d), d, e);
    }
  }
}
